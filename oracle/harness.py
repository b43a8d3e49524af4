"""TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

CPU restatement of the CALLERS of the hot path, just enough to reproduce the
reference's golden time series (tests/data/ref/*_h1_p2/x_{0..9}.txt):

  operators::NonlinearSolid::{Mult, ResidualAndGrad}   operators/nonlinear_solid.cpp:124-283
  forms::Nonlinear::{AddMult, AddMultGrad}             forms/nonlinear.hpp:53-116
  solvers::LineSearchNewton::Mult                      solvers/newton.cpp:10-218
  solvers::GeneralizedAlpha2::{ComputeFactors,StepTime2} solvers/ode.cpp:5-79
  setup of mass / rhs / Dirichlet                      py/py_nonlinear_solid.cpp:138-283

The linear solves (UMFPack / CG in the reference, both MFEM/SuiteSparse, absent
here) use scipy's sparse LU.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def assemble_mass(patch, tables, density, rowptr, col):
    """VectorMassIntegrator(rho) assembled on the PrepareSparsity pattern
    (py_nonlinear_solid.cpp:155-173): rho * int N_a N_b per component."""
    dim = patch.dim
    N = tables["N"]
    wd = tables["weight"] * tables["det"]
    Me = density * np.einsum("eq,eqa,eqb->eab", wd, N, N)
    conn = tables["conn"].astype(np.int64)
    n = patch.n_vdofs
    # CSR keys are sorted (rows ascending, columns sorted within a row)
    keys = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr)) * n + col
    vals = np.zeros(len(col))
    nd = conn.shape[1]
    for c in range(dim):
        r = conn * dim + c
        k = (r[:, :, None] * n + r[:, None, :]).ravel()
        pos = np.searchsorted(keys, k)
        assert np.array_equal(keys[pos], k)
        np.add.at(vals, pos, Me.ravel())
    return vals


def assemble_viscosity(patch, tables, viscosity, rowptr, col):
    """VectorDiffusionIntegrator(viscosity) on the same pattern (py_nonlinear_solid.cpp:176-192): nu * int grad N_a . grad N_b
    per component, with the quadrature rule of the tables (exact on affine patches).  PARITY UNPINNED: no reference fixture
    sets a viscosity, and MFEM's default rule for this integrator is not visible without MFEM."""
    dim = patch.dim
    wd = tables["weight"] * tables["det"]
    g = tables["dN_dX"]                                   # [e, q, a, J]
    Ce = viscosity * np.einsum("eq,eqaJ,eqbJ->eab", wd, g, g)
    conn = tables["conn"].astype(np.int64)
    n = patch.n_vdofs
    keys = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr)) * n + col
    vals = np.zeros(len(col))
    for c in range(dim):
        r = conn * dim + c
        pos = np.searchsorted(keys, (r[:, :, None] * n + r[:, None, :]).ravel())
        np.add.at(vals, pos, Ce.ravel())
    return vals


def assemble_body_force(patch, tables, b):
    """VectorDomainLFIntegrator with a constant vector coefficient
    (py_nonlinear_solid.cpp:221-240): no density factor."""
    dim = patch.dim
    wd = tables["weight"] * tables["det"]
    fe = np.einsum("eq,eqa->ea", wd, tables["N"])
    rhs = np.zeros(patch.n_vdofs)
    conn = tables["conn"].astype(np.int64)
    for c in range(dim):
        np.add.at(rhs, (conn * dim + c).ravel(), (fe * b[c]).ravel())
    return rhs


def eliminate_row_col(rowptr, col, vals, dofs):
    """SparseMatrix::EliminateRowCol(rc, DIAG_ONE) for every rc (forms/nonlinear.hpp:112-115)."""
    n = len(rowptr) - 1
    mask = np.zeros(n, dtype=bool)
    mask[dofs] = True
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    kill = mask[rows] | mask[col]
    vals[kill] = 0.0
    diag = kill & (rows == col) & mask[rows]
    vals[diag] = 1.0


class Operator:
    """operators::NonlinearSolid for one domain integrator (+ optional contact)."""

    def __init__(self, integ, rowptr, col, mass_vals, rhs, dirichlet, contact=None, visc_vals=None):
        self.integ, self.contact = integ, contact
        self.visc = None
        if visc_vals is not None:                      # FormSystemMatrix(zero_dofs) (py_nonlinear_solid.cpp:191)
            self.visc = visc_vals.copy()
            eliminate_row_col(rowptr, col, self.visc, np.asarray(dirichlet, dtype=np.int64))
        self.rowptr, self.col = rowptr, col
        self.n = len(rowptr) - 1
        self.dirichlet = np.asarray(dirichlet, dtype=np.int64)
        # FormSystemMatrix(zero_dofs): mass rows/cols eliminated, diag 1 (py_nonlinear_solid.cpp:173)
        self.mass = mass_vals.copy()
        eliminate_row_col(rowptr, col, self.mass, self.dirichlet)
        self.rhs = rhs.copy()
        self.rhs[self.dirichlet] = 0.0       # py_nonlinear_solid.cpp:277-283
        self.jac = np.zeros_like(self.mass)
        self.dt = 0.0
        self.fac0 = self.fac1 = 0.0
        self.x = None
        self.n_res = self.n_resgrad = 0
        self.tangent_mode = 0

    def _csr(self, vals):
        return sp.csr_matrix((vals, self.col, self.rowptr), shape=(self.n, self.n))

    def set_parameters(self, fac0, fac1, x, v):           # operators/nonlinear_solid.cpp:5-20
        self.fac0, self.fac1, self.x, self.v = fac0, fac1, x, v

    def _push_dt(self):                                    # forms/nonlinear.hpp:63-65
        self.integ.set_dt(self.dt)

    def explicit_accel(self, x):                           # operators/nonlinear_solid.cpp:124-156
        self._push_dt()
        z = np.zeros(self.n)
        self.integ.add_domain_residual(x, z)
        if self.contact is not None:
            self.contact.add_boundary_residual(x, z)
        z[self.dirichlet] = 0.0
        if self.visc is not None:                          # viscosity_->AddMult(dx_dt, z)
            z += self._csr(self.visc) @ self.v0
        z = -z + self.rhs
        return spla.splu(self._csr(self.mass).tocsc()).solve(z)

    def mult(self, a):                                     # operators/nonlinear_solid.cpp:172-205
        self._push_dt()
        self.n_res += 1
        xt = self.x + self.fac0 * a
        y = self._csr(self.mass) @ a
        self.integ.add_domain_residual(xt, y)
        if self.visc is not None:
            y += self._csr(self.visc) @ (self.v + self.fac1 * a)
        if self.contact is not None:
            self.contact.add_boundary_residual(xt, y)
        y[self.dirichlet] = 0.0
        y -= self.rhs
        y[self.dirichlet] = 0.0
        return y

    def residual_and_grad(self, a):                        # operators/nonlinear_solid.cpp:240-283
        self._push_dt()
        self.n_resgrad += 1
        xt = self.x + self.fac0 * a
        y = self._csr(self.mass) @ a
        if self.visc is not None:
            y += self._csr(self.visc) @ (self.v + self.fac1 * a)
        self.jac[:] = self.mass
        self.integ.add_domain_residual_and_grad(xt, self.fac0, y, self.jac, self.tangent_mode)
        if self.contact is not None:
            self.contact.add_boundary_residual_and_grad(xt, self.fac0, y, self.jac, self.tangent_mode)
        y[self.dirichlet] = 0.0
        eliminate_row_col(self.rowptr, self.col, self.jac, self.dirichlet)
        if self.visc is not None:                          # jacobian_->Add(fac1_, viscosity_->SpMat())
            self.jac += self.fac1 * self.visc
        y -= self.rhs
        y[self.dirichlet] = 0.0
        return y, self.jac

    def post_time_advance(self, x):                        # operators/nonlinear_solid.cpp:285-292
        self.integ.domain_post_time_advance(x)


def line_search_newton(op, x0, rel_tol, abs_tol, max_iter, iterative_mode=False):
    """solvers/newton.cpp:10-218 (have_b == false)."""
    x = x0.copy() if iterative_mode else np.zeros_like(x0)
    improved = [True] * 5
    i_improved = 0
    best_res, best_x, best_it = np.finfo(float).max, x.copy(), 0
    r, J = op.residual_and_grad(x)
    norm0 = norm = np.linalg.norm(r)
    norm_goal = max(rel_tol * norm, abs_tol)
    it = 0
    converged = False
    while True:
        if norm <= norm_goal:
            converged = True
            break
        if it >= max_iter:
            if it != 0:
                x = best_x.copy()
            break
        if not any(improved):
            x = best_x.copy()
            break
        c = spla.splu(op._csr(J).tocsc()).solve(r)
        q1 = norm
        q3 = np.linalg.norm(op.mult(x - c))
        q2 = np.linalg.norm(op.mult(x - 0.5 * c))
        den = q1 - 2.0 * q2 + q3
        eps = (3.0 * q1 - 4.0 * q2 + q3) / (4.0 * den) if den != 0 else np.inf
        if den > 0 and 0 < eps < 1:
            scale = eps
        elif q3 < q1:
            scale = 1.0
        else:
            scale = 0.05
        if abs(scale) < 1e-12:
            break
        x = x - scale * c
        if it == max_iter - 1:
            r = op.mult(x)
        else:
            r, J = op.residual_and_grad(x)
        norm = np.linalg.norm(r)
        if norm < best_res:
            best_x, best_res, best_it = x.copy(), norm, it
            improved[i_improved % 5] = True
        else:
            improved[i_improved % 5] = False
        i_improved += 1
        it += 1
    return x, dict(converged=converged, iterations=it, norm=norm, norm0=norm0)


class GeneralizedAlpha2:
    """solvers/ode.cpp:5-79 with MFEM's SetRhoInf rule."""

    def __init__(self, op, rho_inf, newton_opts):
        rho_inf = min(max(rho_inf, 0.0), 1.0)
        am = (2.0 - rho_inf) / (1.0 + rho_inf)
        af = 1.0 / (1.0 + rho_inf)
        beta = 0.25 * (1.0 + am - af) ** 2
        gamma = 0.5 + am - af
        self.fac0 = 0.5 - beta / am
        self.fac1 = af
        self.fac2 = af * (1.0 - gamma / am)
        self.fac3 = beta * af / am
        self.fac4 = gamma * af / am
        self.fac5 = am
        self.op = op
        self.nstate = 0
        self.newton_opts = newton_opts
        self.a = None
        self.aa = None
        self.history = []

    def step(self, x, v, t, dt):
        op = self.op
        op.dt = dt
        if self.nstate == 0:
            op.v0 = v
            self.a = op.explicit_accel(x)
            self.nstate = 1
            self.aa = np.zeros_like(x)
        a = self.a
        xa = x + (v + self.fac0 * dt * a) * (self.fac1 * dt)
        va = v + self.fac2 * dt * a
        fac3dtdt = self.fac3 * dt * dt
        fac4dt = self.fac4 * dt
        op.set_parameters(fac3dtdt, fac4dt, xa, va)        # ImplicitSolve
        self.aa, info = line_search_newton(op, self.aa, **self.newton_opts)
        self.history.append(info)
        aa = self.aa
        xa = xa + fac3dtdt * aa
        va = va + fac4dt * aa
        prev = 1.0 - 1.0 / self.fac1
        x[:] = x * prev + xa / self.fac1
        v[:] = v * prev + va / self.fac1
        self.a = a * prev + aa / self.fac5
        op.post_time_advance(x)
        return t + dt


# golden dof k -> lexicographic node i + 5 j for the 5x5 p=3 balken case:
# MFEM numbers the 4 vertices, then edge interiors (bottom x-up, top x-down (stored
# reversed), left, right y-up), then the interior lexicographically.
GOLDEN_NODE_ORDER_5x5 = np.array(
    [0, 4, 24, 20, 1, 2, 3, 23, 22, 21, 5, 10, 15, 9, 14, 19, 6, 7, 8, 11, 12, 13, 16, 17, 18])


def golden_to_lexicographic(vec):
    """50-vector in the reference's (MFEM) dof order -> byVDIM lexicographic."""
    out = np.zeros(50)
    g = np.asarray(vec).reshape(25, 2)
    out.reshape(25, 2)[GOLDEN_NODE_ORDER_5x5] = g
    return out
