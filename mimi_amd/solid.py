"""Host-side callers of the hot path with the reference's Python surface, so that the solver
calls of the reference's tests / examples run against the HIP integrators:

    nl = mimi_amd.NonlinearSolid(); nl.read_mesh(...); nl.elevate_degrees(2); nl.subdivide(1)
    nl.set_material(mat); nl.boundary_condition = bc; nl.runtime_communication = rc
    nl.setup(1); nl.configure_newton("nonlinear_solid", 1e-12, 1e-8, 10, False)
    nl.time_step_size = 0.05; u = nl.solution_view("displacement", "x"); nl.step_time2()

These layers are CALLERS of the path (out of scope for acceleration); they are
restated in plain numpy / scipy only as far as the path's tests need them:
  PySolid / PyNonlinearSolid::Setup        src/mimi/py/py_solid.cpp:9-68, py_nonlinear_solid.cpp:15-387
  operators::NonlinearSolid                src/mimi/operators/nonlinear_solid.cpp:124-292
  forms::Nonlinear::AddMult[Grad]          src/mimi/forms/nonlinear.hpp:53-116
  solvers::LineSearchNewton::Mult          src/mimi/solvers/newton.cpp:10-218
  solvers::GeneralizedAlpha2               src/mimi/solvers/ode.cpp:5-79
The linear solves (UMFPack / CG in the reference) use scipy's sparse LU.  The element
integration itself always goes through libmimi_hip (no CPU fallback).

Differences a user must know: dofs are numbered lexicographically (the reference exposes MFEM's
NURBS numbering); meshes must be single-cell degree-1 descriptions with unit weights (any quadrilateral / hexahedron:
the refined control net is the cell's multilinear map at the Greville abscissae), which
covers every mesh the reference's solver tests use.
"""
import re

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import nurbs_mesh, splines
from .integrators import CSRPattern, MortarContact, NonlinearSolid as NonlinearSolidIntegrator
from .linear import LinearSolver
from .splines import BSplinePatch


# ---- utils/runtime_communication.hpp:48-198 (the keys that reach the path) -----------------
class RuntimeCommunication:
    """utils/runtime_communication.hpp:48-200 with the pybind11 names of py_runtime_communication.cpp:14-31: runtime
    switches, the save cadence of solution vectors and their .npz output (cnpy::npz_save(..., "a"): one array per call
    appended to the archive `fname`)."""

    def __init__(self):
        self.reals, self.ints = {}, {}
        self.fname = ""
        self._save_every, self._real_history, self._latest = {}, {}, {}
        self.i_timestep_, self.t_ = 0, 0.0

    def set_fname(self, fname):
        self.fname = str(fname)

    def set_real(self, key, value):
        self.reals[key] = float(value)

    def set_int(self, key, value):
        self.ints[key] = int(value)

    def get_real(self, key, default):
        return self.reals.get(key, default)

    def get_int(self, key, default):
        return self.ints.get(key, default)

    # -- time step counter (InitializeTimeStep / NextTimeStep, :71-80) ---------------------------
    def initialize_time_step(self):
        self.i_timestep_, self.t_ = 0, 0.0

    def next_time_step(self, dt):
        self.i_timestep_ += 1
        self.t_ += dt

    # -- save cadence (:115-130) -----------------------------------------------------------------
    def append_should_save(self, name, every):
        self._save_every[str(name)] = int(every)

    def should_save(self, name):
        every = self._save_every.get(name)
        return every is not None and self.i_timestep_ % every == 0

    # -- histories (:132-161) ----------------------------------------------------------------------
    def setup_real_history(self, name, n_reserve):
        self._real_history[name] = []

    def record_real_history(self, name, value):
        self._real_history[name].append(float(value))

    def get_real_history(self, name):
        return self._real_history[name]

    def get_real_history_at(self, name, at):
        return self._real_history[name][at]

    def save_real_history(self, name):
        self.save_vector(name + "_history", np.asarray(self._real_history[name]))

    # -- npz output (:163-197) -------------------------------------------------------------------------
    def save_vector(self, vector_name, vector):
        """append the array `vector_name` to the archive (an .npz is a zip of .npy members)"""
        import zipfile
        if not self.fname:
            raise RuntimeError("Save requested, but fname not set in RuntimeCommunication")
        with zipfile.ZipFile(self.fname, "a", allowZip64=True) as z:
            with z.open(vector_name + ".npy", "w", force_zip64=True) as f:
                np.lib.format.write_array(f, np.ascontiguousarray(vector, dtype=np.float64), allow_pickle=False)

    def save_dynamic_vector(self, vector_name, vector):
        self.save_vector(vector_name + str(self.i_timestep_), vector)
        self._latest[vector_name] = np.array(vector, dtype=np.float64)

    def latest_vector(self, vector_name):
        return self._latest[vector_name]


# ---- utils/boundary_conditions.hpp: BCMarker / BoundaryConditions ----------------------------
class BoundaryMarker:
    def __init__(self):
        self.dirichlet_, self.body_force_, self.contact_ = [], {}, {}

    def dirichlet(self, bid, dim):
        self.dirichlet_.append((int(bid), int(dim)))
        return self

    def body_force(self, dim, value):
        self.body_force_[int(dim)] = float(value)
        return self

    def contact(self, bid, nearest_distance_coeff):
        self.contact_[int(bid)] = nearest_distance_coeff
        return self


class BoundaryConditions:
    def __init__(self):
        self.initial = BoundaryMarker()
        self.current = BoundaryMarker()


# ---- mesh: MFEM NURBS mesh v1.0, one patch (mimi_amd/nurbs_mesh.py) ------------------------------
class Solid:
    """PySolid (src/mimi/py/py_solid.cpp:9-68): mesh handling."""

    def __init__(self):
        self._dim = None
        self.runtime_communication = None
        self.boundary_condition = None
        self.time_step_size = 0.0
        self.current_time = 0.0

    def read_mesh(self, fname):
        self._nurbs = nurbs_mesh.read_mfem_nurbs(fname)                               # py_solid.cpp:70-95
        self._dim = self._nurbs.dim
        self._faces = self._nurbs.faces

    def elevate_degrees(self, degrees, max_degrees=50):
        if int(degrees) > 0:                                                          # py_solid.cpp:148-168
            self._nurbs = self._nurbs.elevate(int(degrees), int(max_degrees))

    def subdivide(self, n_subdivision):
        for _ in range(int(n_subdivision)):                                           # py_solid.cpp:170-183
            self._nurbs = self._nurbs.refine()

    def mesh_dim(self):
        return self._dim

    def mesh_degrees(self):
        return list(self._nurbs.degrees)

    # counts as py_solid.hpp:130-157
    def n_elements(self):
        return self._nurbs.n_elements()

    def n_vertices(self):
        return self._nurbs.n_vertices()

    def n_boundary_elements(self):
        return self._nurbs.n_boundary_elements()

    def n_subelements(self):
        return self._nurbs.n_subelements()

    def mfem_node_order(self):
        """lexicographic node index of every dof of the reference's (MFEM's) numbering: vectors of the reference, such as
        its golden files, are `v_lexicographic.reshape(-1, dim)[order] = v_reference.reshape(-1, dim)`"""
        return self._nurbs.mfem_order()

    def patch(self):
        nb = self._nurbs
        return BSplinePatch(nb.degrees, nb.knots, nb.ctrl, nb.weights if nb.is_rational() else None)


def _element_tables(patch, quadrature_order=-1, with_gradients=False, elements=None):
    """N[e,q,a], w*det[e,q], conn[e,a] (and dN/dX[e,q,i,a]) for the mass matrix / damping / body force; elements: a slice
    of the element range (setup walks large meshes in chunks)."""
    dim = patch.dim
    pmax = max(patch.degrees)
    order = 2 * pmax + 3 if quadrature_order < 0 else quadrature_order
    nq = order // 2 + 1
    tabs = [splines._tables_1d(patch.knots[d], patch.degrees[d], nq) for d in range(dim)]
    m = [len(t[0]) for t in tabs]
    e = np.arange(int(np.prod(m)))
    if elements is not None:
        e = e[elements]
    em = []
    for s in m:
        em.append(e % s)
        e = e // s
    def outer(f):
        """[e, (z,) y, x, (c,) b, a] = product of the 1-D tables f[d][e, a_d, q_d] (plain broadcasting: the three-operand
        einsum of this takes ten times as long)"""
        t = [np.ascontiguousarray(g.transpose(0, 2, 1)) for g in f]            # [e, q_d, a_d]
        r = t[1][:, :, None, :, None] * t[0][:, None, :, None, :]               # [e, y, x, b, a]
        if dim == 3:
            r = t[2][:, :, None, None, :, None, None] * r[:, None, :, :, None, :, :]   # [e, z, y, x, c, b, a]
        return r

    if dim == 2:
        w = np.einsum("y,x->yx", tabs[1][3], tabs[0][3]).ravel()
    else:
        w = np.einsum("z,y,x->zyx", tabs[2][3], tabs[1][3], tabs[0][3]).ravel()
    N = outer([tabs[d][1][em[d]] for d in range(dim)])
    ne = len(em[0])
    N = N.reshape(ne, w.size, -1)
    # connectivity
    conn = np.zeros((ne, N.shape[2]), dtype=np.int64)
    a = np.arange(N.shape[2])
    stride = 1
    for d in range(dim):
        ad = (a // int(np.prod([pp + 1 for pp in patch.degrees[:d]]))) % (patch.degrees[d] + 1)
        conn += ((tabs[d][0][em[d]] - patch.degrees[d])[:, None] + ad[None, :]) * stride
        stride *= patch.n_ctrl[d]
    # geometry Jacobian per point from the control net (derivatives wrt the element's reference coordinates)
    B = [t[1] for t in tabs]
    D = [t[2] for t in tabs]
    dN = []
    for k in range(dim):
        f = [D[d] if d == k else B[d] for d in range(dim)]
        dN.append(outer([f[d][em[d]] for d in range(dim)]).reshape(ne, w.size, -1))
    if getattr(patch, "weights", None) is not None:
        # rational basis: N = B w / sum(B w), with the quotient rule for the derivatives (precomputed.cpp:295-321 via MFEM)
        wa = patch.weights[conn]                                        # [e, a]
        Ws = np.einsum("eqa,ea->eq", N, wa)
        dWs = [np.einsum("eqa,ea->eq", g, wa) for g in dN]
        dN = [(g * wa[:, None, :] * Ws[:, :, None] - (N * wa[:, None, :]) * dW[:, :, None]) / (Ws ** 2)[:, :, None]
              for g, dW in zip(dN, dWs)]
        N = N * wa[:, None, :] / Ws[:, :, None]
    X = patch.control_points[conn]                                      # [e, a, i]
    J = np.stack([np.matmul(g, X) for g in dN], axis=-1)                # [e, q, i, k]
    det = np.linalg.det(J)
    if not np.all(det > 0):
        raise RuntimeError("geometry map has a non-positive Jacobian determinant")
    if with_gradients:
        Jinv = np.linalg.inv(J)                                             # dxi_k / dX_i  [e, q, k, i]
        dN_dX = np.einsum("keqa,eqki->eqia", np.stack(dN), Jinv)            # [e, q, i, a]
        return N, w[None, :] * det, conn, dN_dX
    return N, w[None, :] * det, conn


class NonlinearSolid(Solid):
    """PyNonlinearSolid (src/mimi/py/py_nonlinear_solid.cpp:15-387) on top of the HIP integrators."""

    def __init__(self, device=0):
        super().__init__()
        self.material = None
        self.device = device
        self._newton = dict(rel_tol=1e-8, abs_tol=1e-12, max_iter=None, iterative_mode=False)
        self.tangent_mode = 0
        self.newton_history = []

    def set_material(self, material):
        self.material = material

    # -- Setup (py_nonlinear_solid.cpp:15-387) -------------------------------------------------
    def setup(self, nthreads=-1):
        dim = self._dim
        self.patch_ = patch = self.patch()
        n = patch.n_vdofs
        self.pattern_ = CSRPattern.of_bspline_patch(patch, device=self.device)
        rowptr, col = self.pattern_.rowptr, self.pattern_.col
        self.x = np.zeros(n)        # displacement (py_nonlinear_solid.cpp:119)
        self.x_dot = np.zeros(n)
        if self.runtime_communication is None:                       # PySolid::RuntimeCommunication(): created on demand
            self.runtime_communication = RuntimeCommunication()
        rc = self.runtime_communication
        rc.initialize_time_step()                                    # py_solid.cpp:360
        bc = self.boundary_condition or BoundaryConditions()
        # Dirichlet dofs (FindBoundaryDofIds, py_solid.cpp:185-235): bid -> attribute bid+1
        dofs = []
        for bid, comp in bc.initial.dirichlet_:
            axis, side = self._faces[bid + 1]
            dofs.append(patch.boundary_nodes(axis, side) * dim + comp)
        self.dirichlet_ = np.unique(np.concatenate(dofs)) if dofs else np.zeros(0, dtype=np.int64)
        # mass (VectorMassIntegrator(rho), FormSystemMatrix(zero_dofs); :155-173), damping (:176-192:
        # VectorDiffusionIntegrator(viscosity): C_(a,i),(b,j) = d_ij nu int grad N_a . grad N_b, integrated with the same
        # rule as the mass matrix -- exact on affine patches; mfem's own default rule for this integrator cannot be read
        # here and no reference fixture sets a viscosity: parity unpinned) and rhs (:221-283), in chunks of elements
        viscosity = getattr(self.material, "viscosity", -1.0)
        mass, visc, rhs = _assemble_mass_viscosity_rhs(patch, rowptr, self.material.density, viscosity,
                                                       bc.initial.body_force_)
        self.mass_ = mass
        _eliminate_row_col(rowptr, col, self.mass_, self.dirichlet_)
        self.visc_ = visc
        if visc is not None:
            _eliminate_row_col(rowptr, col, self.visc_, self.dirichlet_)
        rhs[self.dirichlet_] = 0.0
        self.rhs_ = rhs
        # integrators (py_nonlinear_solid.cpp:197-218, 286-326)
        q_order = rc.get_int("nonlinear_solid_quadrature_order", -1)
        try:
            self.domain_ = NonlinearSolidIntegrator("nonlinear_solid", self.material, self.pattern_, patch=patch,
                                                    device=self.device, quadrature_order=q_order).Prepare()
        except RuntimeError as exc:
            if "not a tensor product" not in str(exc):
                raise
            # NURBS weights that do not factorise: the reference's flat per-point tables (general kernels)
            _, wd_t, conn_t, dN_dX = _element_tables(patch, q_order, with_gradients=True)
            tables = dict(dim=dim, n_nodes=patch.n_nodes, dofs=conn_t.astype(np.int32), dN_dX=np.ascontiguousarray(dN_dX),
                          weight_det=np.ascontiguousarray(wd_t))
            self.domain_ = NonlinearSolidIntegrator("nonlinear_solid", self.material, self.pattern_, tables=tables,
                                                    device=self.device).Prepare()
        self.domain_.SetTangentMode(self.tangent_mode)
        self.contacts_ = []
        for bid, body in bc.current.contact_.items():
            axis, side = self._faces[bid + 1]
            self.contacts_.append(MortarContact(body, "contact", self.pattern_, patch, axis, side, device=self.device,
                                                quadrature_order=rc.get_int("contact_quadrature_order", -1)).Prepare())
        if self._newton["max_iter"] is None:
            self._newton["max_iter"] = 10 * dim                                       # :346-361
        rho_inf = min(max(rc.get_real("ode_coefficient", 0.25), 0.0), 1.0)            # :367-370
        am = (2.0 - rho_inf) / (1.0 + rho_inf)
        af = 1.0 / (1.0 + rho_inf)
        beta = 0.25 * (1.0 + am - af) ** 2
        gamma = 0.5 + am - af
        self._fac = (0.5 - beta / am, af, af * (1.0 - gamma / am), beta * af / am, gamma * af / am, am)  # ode.cpp:5-14
        self._nstate = 0
        # linear solver (py_nonlinear_solid.cpp:327-343): "use_iterative_solver" -> GMRES + Jacobi on the device
        # (mimi_amd/linear.py); else a sparse direct solve on the host (UMFPack in the reference, SuperLU here)
        self.linear_ = LinearSolver(self.pattern_, self.dirichlet_, device=self.device)
        self.use_iterative_solver_ = bool(rc.get_int("use_iterative_solver", 0))
        self._to_device()

    def configure_newton(self, name, rel_tol, abs_tol, max_iter, iterative_mode):   # py_solid.cpp:334-346
        self._newton = dict(rel_tol=rel_tol, abs_tol=abs_tol, max_iter=int(max_iter), iterative_mode=bool(iterative_mode))

    def solution_view(self, fe_space, component):
        """"x" (the displacement), "x_dot", "x_ref" (the nodes' reference positions, py_nonlinear_solid.cpp:91-114): host
        arrays in this facade's node order, the solver's own storage for the first two (write prescribed values in place)"""
        if component == "x_ref":
            return np.ascontiguousarray(self.patch_.control_points, dtype=np.float64).reshape(-1).copy()
        return {"x": self.x, "x_dot": self.x_dot}[component]

    # -- operators::NonlinearSolid ----------------------------------------------------------------
    # ---- device-resident state ---------------------------------------------------------------------
    # x, v, a, the residual, the mass / viscosity / Jacobian values and the right-hand side live in HBM (torch tensors
    # as the memory container); the integrators, the eliminations, the matrix-vector products and -- on the iterative
    # route -- the linear solves use them in place.  The host sees: norms (scalars), the solution after a step
    # (n_vdofs doubles into the arrays `solution_view` hands out) and, on the reference's default direct-solve route
    # only, the Jacobian values for the host factorisation.
    def _to_device(self):
        import torch
        self._torch = torch
        dev = torch.device("cuda", self.device)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self.d_mass_, self.d_rhs_ = f(self.mass_), f(self.rhs_)
        self.d_visc_ = f(self.visc_) if self.visc_ is not None else None
        self.d_jac_ = torch.zeros_like(self.d_mass_)
        self.d_dirichlet_ = torch.from_numpy(np.asarray(self.dirichlet_, dtype=np.int64)).to(dev)
        self.d_x_, self.d_v_ = f(self.x), f(self.x_dot)
        self.pcie_csr_bytes_ = 0          # CSR values that crossed PCIe since setup (direct-solve route only)

    def _csr(self, vals):
        n = len(self.x)
        return sp.csr_matrix((vals, self.pattern_.col, self.pattern_.rowptr), shape=(n, n))

    def _push(self, integ):
        integ.dt_, integ.first_effective_dt_, integ.second_effective_dt_ = self.time_step_size, self._fac0, self._fac1

    def _add_mult(self, xt, y):                       # forms/nonlinear.hpp:53-81
        self._push(self.domain_)
        self.domain_.AddDomainResidual(xt, y)
        for c in self.contacts_:
            c.AddBoundaryResidual(xt, y)
        self.linear_.Eliminate(y, None)               # y[ess] = 0

    def _linear_part(self, a, y):
        """y = M a [+ C (v_alpha + fac1 a)]   (operators/nonlinear_solid.cpp:177-187,249-255)"""
        y.zero_()
        self.linear_.AddMult(self.d_mass_, a, y)
        if self.d_visc_ is not None:
            self.linear_.AddMult(self.d_visc_, self._va + self._fac1 * a, y)

    def _mult(self, a):                               # operators/nonlinear_solid.cpp:172-205
        xt = self._xa + self._fac0 * a
        y = self._torch.empty_like(a)
        self._linear_part(a, y)
        self._add_mult(xt, y)
        y -= self.d_rhs_
        self.linear_.Eliminate(y, None)
        return y

    def _residual_and_grad(self, a):                  # operators/nonlinear_solid.cpp:240-283
        xt = self._xa + self._fac0 * a
        y = self._torch.empty_like(a)
        self._linear_part(a, y)
        # std::copy_n(mass_A_, ...) followed by AddMultGrad, as one pass: J = M + fac0 K with the row gathers reading M
        # where "+=" would read J (mimi_hip_domain_add_residual_and_grad_from) -- no 2 x nnz copy, and the domain
        # integrator of a single-patch solid touches every row
        self._push(self.domain_)
        self.domain_.AddDomainResidualAndGradFrom(xt, self._fac0, y, self.d_mass_, self.d_jac_)
        for c in self.contacts_:
            c.AddBoundaryResidualAndGrad(xt, self._fac0, y, self.d_jac_)
        self.linear_.Eliminate(y, self.d_jac_)        # forms/nonlinear.hpp:76-80,112-115
        if self.d_visc_ is not None:
            self.d_jac_.add_(self.d_visc_, alpha=self._fac1)     # jacobian_->Add(fac1_, viscosity_->SpMat())
        y -= self.d_rhs_
        self.linear_.Eliminate(y, None)
        return y, self.d_jac_

    def _solve(self, J, r):
        """the linear solve of a Newton iteration (py_nonlinear_solid.cpp:327-343)"""
        if self.use_iterative_solver_:
            return self.linear_.Mult(J, r, self._torch.zeros_like(r))        # GMRES + Jacobi, all in HBM
        # the reference's default: a sparse direct solve (UMFPack there, SuperLU here) -- on the host
        Jh = J.cpu().numpy()
        self.pcie_csr_bytes_ += Jh.nbytes
        c = spla.splu(self._csr(Jh).tocsc()).solve(r.cpu().numpy())
        return self._torch.from_numpy(c).to(r.device)

    def _newton_solve(self, x0):                      # solvers/newton.cpp:10-218
        torch = self._torch
        o = self._newton
        nrm = lambda t: float(torch.linalg.vector_norm(t))
        x = x0.clone() if o["iterative_mode"] else torch.zeros_like(x0)
        improved, i_improved = [True] * 5, 0
        best_res, best_x = np.finfo(float).max, x.clone()
        r, J = self._residual_and_grad(x)
        norm0 = norm = nrm(r)
        goal = max(o["rel_tol"] * norm, o["abs_tol"])
        it, converged = 0, False
        while True:
            if norm <= goal:
                converged = True
                break
            if it >= o["max_iter"]:
                if it != 0:
                    x = best_x.clone()
                break
            if not any(improved):
                x = best_x.clone()
                break
            c = self._solve(J, r)
            q1 = norm
            q3 = nrm(self._mult(x - c))
            q2 = nrm(self._mult(x - 0.5 * c))
            den = q1 - 2.0 * q2 + q3
            eps = (3.0 * q1 - 4.0 * q2 + q3) / (4.0 * den) if den != 0 else np.inf
            scale = eps if (den > 0 and 0 < eps < 1) else (1.0 if q3 < q1 else 0.05)
            if abs(scale) < 1e-12:
                break
            x = x - scale * c
            if it == o["max_iter"] - 1:
                r = self._mult(x)
            else:
                r, J = self._residual_and_grad(x)
            norm = nrm(r)
            if norm < best_res:
                best_x, best_res = x.clone(), norm
                improved[i_improved % 5] = True
            else:
                improved[i_improved % 5] = False
            i_improved += 1
            it += 1
        self.newton_history.append(dict(converged=converged, iterations=it, norm=norm, norm0=norm0))
        return x

    # -- GeneralizedAlpha2::StepTime2 (solvers/ode.cpp:16-79) -------------------------------------
    def step_time2(self):
        torch = self._torch
        dt = self.time_step_size
        f0, f1, f2, f3, f4, f5 = self._fac
        # what the caller may have written through solution_view since the last step (n_vdofs doubles)
        self.d_x_.copy_(torch.from_numpy(self.x))
        self.d_v_.copy_(torch.from_numpy(self.x_dot))
        x, v = self.d_x_, self.d_v_
        self._fac0, self._fac1 = f3 * dt * dt, f4 * dt
        if self._nstate == 0:
            z = torch.zeros_like(x)                    # operators/nonlinear_solid.cpp:124-156
            self._add_mult(x, z)
            if self.d_visc_ is not None:
                self.linear_.AddMult(self.d_visc_, v, z)
            z = self.d_rhs_ - z
            if self.use_iterative_solver_:
                # mass_inv_: mfem::CGSolver + DSmoother (operators/nonlinear_solid.cpp:39-50,155)
                self._a = self.linear_.MultCG(self.d_mass_, z, torch.zeros_like(z))
            else:
                self._a = torch.from_numpy(spla.splu(self._csr(self.mass_).tocsc()).solve(z.cpu().numpy())).to(x.device)
            self._aa = torch.zeros_like(x)
            self._nstate = 1
        a = self._a
        self._xa = x + (v + f0 * dt * a) * (f1 * dt)
        self._va = v + f2 * dt * a
        self._aa = self._newton_solve(self._aa)
        aa = self._aa
        xa = self._xa + self._fac0 * aa
        va = self._va + self._fac1 * aa
        prev = 1.0 - 1.0 / f1
        x.mul_(prev).add_(xa, alpha=1.0 / f1)
        v.mul_(prev).add_(va, alpha=1.0 / f1)
        self._a = a * prev + aa / f5
        # PostTimeAdvance (operators/nonlinear_solid.cpp:285-292)
        self._push(self.domain_)
        self.domain_.DomainPostTimeAdvance(x)
        for c in self.contacts_:
            c.BoundaryPostTimeAdvance(x)
        self.current_time += dt
        # the arrays solution_view handed out (zero-copy views of the reference: py_solid.cpp:379-388)
        self.x[:] = x.cpu().numpy()
        self.x_dot[:] = v.cpu().numpy()
        x, v = self.x, self.x_dot
        # PySolid::StepTime2 (py_solid.cpp:433-440): save cadence, in the reference's (MFEM's) dof numbering
        rc = self.runtime_communication
        if rc is not None:
            if rc.should_save("x"):
                rc.save_dynamic_vector("x_", self.in_reference_numbering(x))
            if rc.should_save("v"):
                rc.save_dynamic_vector("v_", self.in_reference_numbering(v))
            rc.next_time_step(dt)

    def in_reference_numbering(self, vec):
        """byVDIM vector of this facade (lexicographic nodes) -> the reference's dof order (MFEM's NURBS numbering)"""
        order = self._nurbs.mfem_order()
        return np.ascontiguousarray(np.asarray(vec).reshape(-1, self._dim)[order]).reshape(-1)

    def from_reference_numbering(self, vec):
        order = self._nurbs.mfem_order()
        out = np.zeros(len(order) * self._dim)
        out.reshape(-1, self._dim)[order] = np.asarray(vec).reshape(-1, self._dim)
        return out


def _structured_positions(patch, rowptr, conn):
    """Position in the CSR value array of (row = node conn[e, a] component 0, column = node conn[e, b] component 0) in the
    structured pattern of a lexicographically numbered patch (CSRPattern.of_bspline_patch): the columns of a node's row are
    the nodes of its window [A_d - p_d, A_d + p_d] (clipped), lexicographic, times the components -- so the position is
    arithmetic, no search through the 10^8 column indices of a large mesh.  Returns pos0[e, a, b] and the row lengths."""
    dim = patch.dim
    row0 = conn * dim
    n_e, n_a = conn.shape
    rank = np.zeros((n_e, n_a, n_a), dtype=np.int64)
    width = np.ones((n_e, n_a), dtype=np.int64)
    rem = conn.copy()
    for d in range(dim):
        n_d, p_d = patch.n_ctrl[d], patch.degrees[d]
        x = rem % n_d                       # coordinate of every node of the element in direction d
        rem = rem // n_d
        lo = np.maximum(x - p_d, 0)         # the window of a ROW node
        w = np.minimum(x + p_d, n_d - 1) - lo + 1
        rank += (x[:, None, :] - lo[:, :, None]) * width[:, :, None]
        width = width * w
    return rowptr[row0][:, :, None] + dim * rank, rowptr[row0 + 1] - rowptr[row0]


def _assemble_mass_viscosity_rhs(patch, rowptr, density, viscosity, body_force, chunk=8192):
    """mass (VectorMassIntegrator(rho), py_nonlinear_solid.cpp:155-173), damping (:176-192: VectorDiffusionIntegrator(
    viscosity): C_(a,i),(b,j) = d_ij nu int grad N_a . grad N_b, integrated with the rule of the mass matrix -- exact on
    affine patches; mfem's own default rule for this integrator cannot be read here and no reference fixture sets a
    viscosity: parity unpinned) and the body-force vector (:221-283) of the whole patch, on the host.
    Elements are taken COLOUR by colour (element index modulo p + 1 per direction): two elements of a colour share no
    node, so their entries go to distinct positions and one vectorised `+=` adds them -- in a fixed order of the colours,
    the same bits every run."""
    dim = patch.dim
    n = patch.n_vdofs
    nnz = int(rowptr[-1])
    mass = np.zeros(nnz)
    visc = np.zeros(nnz) if viscosity > 0.0 else None
    rhs = np.zeros(n)
    spans = list(patch.n_spans)
    e_all = np.arange(patch.n_elements)
    em, rem = [], e_all
    for d in range(dim):
        em.append(rem % spans[d])
        rem = rem // spans[d]
    colour = np.zeros(patch.n_elements, dtype=np.int64)
    mult = 1
    for d in range(dim):
        colour += (em[d] % (patch.degrees[d] + 1)) * mult
        mult *= patch.degrees[d] + 1
    order = np.argsort(colour, kind="stable")
    bounds = np.concatenate([[0], np.cumsum(np.bincount(colour, minlength=mult))])
    for cidx in range(mult):
        els_c = order[bounds[cidx]:bounds[cidx + 1]]
        for s0 in range(0, len(els_c), chunk):
            els = els_c[s0:s0 + chunk]
            if visc is not None:
                N, wd, conn, dN_dX = _element_tables(patch, with_gradients=True, elements=els)
                G = dN_dX.reshape(dN_dX.shape[0], -1, dN_dX.shape[3])                    # [e, (q, i), a]
                Ce = viscosity * np.matmul((G * np.repeat(wd, dim, axis=1)[:, :, None]).transpose(0, 2, 1), G)
            else:
                N, wd, conn = _element_tables(patch, elements=els)
            Me = density * np.matmul((N * wd[:, :, None]).transpose(0, 2, 1), N)           # [e, a, b]
            pos0, row_len = _structured_positions(patch, rowptr, conn)
            # the rows of a node's other components follow with the same column pattern: + c (row length) for the row,
            # + c for the column
            for c in range(dim):
                pos = (pos0 + c * row_len[:, :, None] + c).ravel()
                mass[pos] += Me.ravel()
                if visc is not None:
                    visc[pos] += Ce.ravel()
            fe = np.einsum("eq,eqa->ea", wd, N)
            for comp, value in body_force.items():
                rhs[(conn * dim + comp).ravel()] += (fe * value).ravel()
    return mass, visc, rhs


def _eliminate_row_col(rowptr, col, vals, dofs):
    """SparseMatrix::EliminateRowCol(rc, DIAG_ONE) for each rc (forms/nonlinear.hpp:112-115)."""
    n = len(rowptr) - 1
    mask = np.zeros(n, dtype=bool)
    mask[dofs] = True
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    kill = mask[rows] | mask[col]
    vals[kill] = 0.0
    vals[kill & (rows == col) & mask[rows]] = 1.0
