"""Host-side mirror of the reference's integrator interface
(src/mimi/integrators/nonlinear_base.hpp:14-154) on top of the C ABI of libmimi_hip.so.

Method names, argument meaning and accumulate-into semantics are the reference's:
  Prepare(); AddDomainResidual(u, r); AddDomainResidualAndGrad(u, grad_factor, r, A);
  DomainPostTimeAdvance(u); public members dt_, first_effective_dt_, second_effective_dt_
(snake_case aliases are provided as well).  `u`, `r`, `A` may be numpy arrays (host) or
torch tensors on the handle's device (used in place, call is asynchronous on the stream).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, fptr, ptr

TANGENT_ANALYTIC = 0
TANGENT_REFERENCE_FD = 1


class CSRPattern:
    """(rowptr int64[n_vdofs+1], col int32[nnz]) of PrepareSparsity (utils/precomputed.cpp:151-174)."""

    def __init__(self, rowptr, col, nnz):
        self.rowptr, self.col, self.nnz = rowptr, col, int(nnz)

    @classmethod
    def of_bspline_patch(cls, patch, device=0, on_device=False, node_box=None):
        """Structured pattern of a lexicographically numbered patch, built on the GPU.

        node_box = (begin[dim], end[dim]): the row slice of a rank that owns that box of nodes -- rowptr keeps its full
        length, rows of other nodes are empty, col and the value array hold only the slice (mimi_hip.h:
        mimi_hip_bspline_sparsity_rows)."""
        L = _capi.lib()
        n = (C.c_int32 * 3)(*(patch.n_ctrl + [1] * (3 - patch.dim)))
        p = (C.c_int32 * 3)(*(patch.degrees + [0] * (3 - patch.dim)))
        nnz = C.c_int64(0)
        nrows = patch.n_vdofs
        if node_box is None:
            def build(rowptr, col):
                check(L.mimi_hip_bspline_sparsity(patch.dim, n, p, device, ptr(rowptr), ptr(col) if col is not None else None,
                                                  C.byref(nnz)))
        else:
            lo = (C.c_int32 * 3)(*(list(node_box[0]) + [0] * (3 - patch.dim)))
            hi = (C.c_int32 * 3)(*(list(node_box[1]) + [1] * (3 - patch.dim)))

            def build(rowptr, col):
                check(L.mimi_hip_bspline_sparsity_rows(patch.dim, n, p, lo, hi, device, ptr(rowptr),
                                                       ptr(col) if col is not None else None, C.byref(nnz)))
        if on_device:
            import torch
            dev = torch.device("cuda", device)
            rowptr = torch.empty(nrows + 1, dtype=torch.int64, device=dev)
            build(rowptr, None)
            col = torch.empty(nnz.value, dtype=torch.int32, device=dev)
            build(rowptr, col)
        else:
            rowptr = np.empty(nrows + 1, dtype=np.int64)
            build(rowptr, None)
            col = np.empty(nnz.value, dtype=np.int32)
            build(rowptr, col)
        return cls(rowptr, col, nnz.value)


class NonlinearBase:
    """integrators/nonlinear_base.hpp:14-154"""
    dt_ = 0.0
    first_effective_dt_ = 0.0
    second_effective_dt_ = 0.0

    def __init__(self, name):
        self.name_ = name

    def Name(self):
        return self.name_


class NonlinearSolid(NonlinearBase):
    """integrators::NonlinearSolid (integrators/nonlinear_solid.hpp:15-128) on one MI355X.

    Either `patch` (a splines.BSplinePatch: tables are generated on the device and the
    tensor-product kernels are used) or `tables` (dict with the reference's flattened
    PrecomputedData: dim, n_nodes, dofs[e,a], dN_dX[e,q,J,a], weight_det[e,q]) must be given.
    """

    def __init__(self, name, material, pattern, patch=None, tables=None, device=0, quadrature_order=-1,
                 element_box=None, node_ids=None):
        super().__init__(name)
        self.material_ = material
        self.pattern_ = pattern
        self.patch_, self.tables_ = patch, tables
        self.device_ = device
        self.quadrature_order_ = quadrature_order
        self.element_box_ = element_box
        self.node_ids_ = node_ids
        self._h = None
        self._keep = []

    # -- NonlinearSolid::Prepare (nonlinear_solid.cpp:31-46) --------------------------
    def Prepare(self):
        L = _capi.lib()
        mat = self.material_._c_struct()
        h = C.c_void_p()
        if self.patch_ is not None:
            p = self.patch_
            d = _capi.BSplinePatch()
            d.dim = p.dim
            for i in range(p.dim):
                d.degree[i] = p.degrees[i]
                d.n_knots[i] = len(p.knots[i])
                d.knots[i] = p.knots[i].ctypes.data
            d.control_points = p.control_points.ctypes.data
            if getattr(p, "weights", None) is not None:
                d.weights = p.weights.ctypes.data
            if self.node_ids_ is not None:
                ids = np.ascontiguousarray(self.node_ids_, dtype=np.int64)
                self._keep.append(ids)
                d.node_ids = ids.ctypes.data
            d.quadrature_order = self.quadrature_order_
            if self.element_box_ is not None:
                b, e = self.element_box_
                for i in range(3):
                    d.element_begin[i] = b[i]
                    d.element_end[i] = e[i]
            d.csr_rowptr = ptr(self.pattern_.rowptr, "int64").value
            d.csr_col = ptr(self.pattern_.col, "int32").value
            check(L.mimi_hip_domain_create_bspline(C.byref(d), C.byref(mat), self.device_, C.byref(h)))
        else:
            t = self.tables_
            d = _capi.DomainTables()
            dofs = np.ascontiguousarray(t["dofs"], dtype=np.int32)
            g = np.ascontiguousarray(t["dN_dX"], dtype=np.float64)
            wd = np.ascontiguousarray(t["weight_det"], dtype=np.float64)
            self._keep += [dofs, g, wd]
            d.dim = t["dim"]
            d.n_elements, d.n_dof = dofs.shape
            d.n_quad = wd.shape[1]
            d.n_nodes = t["n_nodes"]
            assert g.shape == (d.n_elements, d.n_quad, d.dim, d.n_dof)
            d.dofs, d.dN_dX, d.weight_det = dofs.ctypes.data, g.ctypes.data, wd.ctypes.data
            d.csr_rowptr = ptr(self.pattern_.rowptr, "int64").value
            d.csr_col = ptr(self.pattern_.col, "int32").value
            check(L.mimi_hip_domain_create(C.byref(d), C.byref(mat), self.device_, C.byref(h)))
        self._h = h
        self.n_elements_ = int(L.mimi_hip_domain_info(h, 0))
        self.n_quad_ = int(L.mimi_hip_domain_info(h, 1))
        self.n_dof_ = int(L.mimi_hip_domain_info(h, 2))
        self.nnz_ = int(L.mimi_hip_domain_info(h, 3))
        self.n_vdofs_ = int(L.mimi_hip_domain_info(h, 4))
        self.path_ = int(L.mimi_hip_domain_info(h, 5))
        self.has_states_ = self.material_._kind == 1
        return self

    def _handle(self):
        if self._h is None:
            raise RuntimeError("Prepare() has not been called")
        return self._h

    def _push_dt(self):
        # forms::Nonlinear pushes these public members before each call (forms/nonlinear.hpp:63-65)
        check(_capi.lib().mimi_hip_domain_set_dt(self._handle(), self.dt_, self.first_effective_dt_,
                                                 self.second_effective_dt_))

    def SetTangentMode(self, mode):
        check(_capi.lib().mimi_hip_domain_set_tangent_mode(self._handle(), mode))

    def SetStream(self, stream):
        """launch on `stream` (a hipStream_t as an int); 0 / None: back to following torch's current stream for CUDA
        tensors and the handle's own stream for host buffers"""
        self._user_stream = bool(stream)
        check(_capi.lib().mimi_hip_domain_set_stream(self._handle(), C.c_void_p(stream) if stream else None))

    def Integrate(self, current_u):
        """phase 1 of a tangent assembly on its own (mimi_hip.h: mimi_hip_domain_integrate): the element pieces stay in the
        handle's scratch until Gather() adds them into r / A.  Two-phase tensor paths, device tensors."""
        self._push_dt()
        self._follow_torch(current_u)
        check(_capi.lib().mimi_hip_domain_integrate(self._handle(), fptr(current_u)))

    def Gather(self, grad_factor, residual, grad, node_begin, node_end):
        """phase 2 over the nodes [node_begin, node_end) (global node indices per direction): r and A += the rows of those
        nodes.  Every node the handle's elements touch is to be gathered exactly once per Integrate()."""
        self._follow_torch(residual, grad)
        lo = (C.c_int32 * 3)(*[int(v) for v in node_begin])
        hi = (C.c_int32 * 3)(*[int(v) for v in node_end])
        check(_capi.lib().mimi_hip_domain_gather(self._handle(), float(grad_factor), fptr(residual), fptr(grad), lo, hi))

    def _follow_torch(self, *buffers):
        # ordering with the caller's torch work (zero fills of r / A, copies of u): see _capi.torch_stream_of
        if getattr(self, "_user_stream", False):
            return
        s = _capi.torch_stream_of(*buffers)
        if s is not None or getattr(self, "_followed", None):
            check(_capi.lib().mimi_hip_domain_set_stream(self._handle(), C.c_void_p(s) if s else None))
            self._followed = s

    def Synchronize(self):
        check(_capi.lib().mimi_hip_domain_synchronize(self._handle()))

    # -- nonlinear_solid.cpp:151-160 ---------------------------------------------------
    def AddDomainResidual(self, current_u, residual):
        self._push_dt()
        self._follow_torch(current_u, residual)
        check(_capi.lib().mimi_hip_domain_add_residual(self._handle(), fptr(current_u), fptr(residual)))

    # -- nonlinear_solid.cpp:162-177 ---------------------------------------------------
    def AddDomainResidualAndGrad(self, current_u, grad_factor, residual, grad_values):
        self._push_dt()
        self._follow_torch(current_u, residual, grad_values)
        check(_capi.lib().mimi_hip_domain_add_residual_and_grad(self._handle(), fptr(current_u), float(grad_factor),
                                                                fptr(residual), fptr(grad_values)))

    def AddDomainResidualAndGradFrom(self, current_u, grad_factor, residual, base_values, grad_values):
        """r += R(u); grad_values = base_values + grad_factor K(u) on the rows of the handle's nodes: the operator's
        "jacobian <- mass values, then AddMultGrad" (operators/nonlinear_solid.cpp:257-258) as ONE pass -- the row gathers
        read base_values where "+=" would read grad_values (mimi_hip.h: mimi_hip_domain_add_residual_and_grad_from)."""
        self._push_dt()
        self._follow_torch(current_u, residual, grad_values, base_values)
        check(_capi.lib().mimi_hip_domain_add_residual_and_grad_from(self._handle(), fptr(current_u), float(grad_factor),
                                                                     fptr(residual), fptr(base_values), fptr(grad_values)))

    # -- nonlinear_solid.cpp:179-199 ---------------------------------------------------
    def DomainPostTimeAdvance(self, converged_u):
        # the reference's material keeps the dt_ of the latest Add* call (nonlinear_solid.cpp:154,167)
        self._push_dt()
        self._follow_torch(converged_u)
        check(_capi.lib().mimi_hip_domain_post_time_advance(self._handle(), fptr(converged_u)))

    def AddDomainGrad(self, current_u, grad):
        raise RuntimeError("Currently not implemented, use AddDomainResidualAndGrad")  # nonlinear_solid.hpp:108-113

    # -- material state (MaterialState, materials.hpp:278-286) --------------------------
    KERNEL_FAMILIES = {0: "none", 1: "tensor_p2_two_phase", 2: "tensor_p3_two_phase", 3: "tensor_small", 4: "general"}

    def LastKernelFamily(self):
        """which kernel family the last assembly on this handle ran on (tests assert it; see mimi_hip_domain_info)"""
        return self.KERNEL_FAMILIES[int(_capi.lib().mimi_hip_domain_info(self._handle(), 7))]

    def State(self, what):
        # "plastic_strain" = the material's first state matrix (J2 / J2Linear: plastic strain, J2Simo: be_old, J2Log:
        # Fp_inv); "state2" = its second one (J2Linear: beta, J2Simo: F_old)
        ids = {"accumulated_plastic_strain": 0, "temperature": 1, "plastic_strain": 2, "state2": 3}
        n = self.n_elements_ * self.n_quad_
        dim = self.patch_.dim if self.patch_ is not None else self.tables_["dim"]
        shape = (self.n_elements_, self.n_quad_, dim * dim) if ids[what] >= 2 else (self.n_elements_, self.n_quad_)
        out = np.empty(shape)
        check(_capi.lib().mimi_hip_domain_get_state(self._handle(), ids[what], ptr(out), out.size))
        return out

    def SetPhaseTiming(self, on=True):
        check(_capi.lib().mimi_hip_domain_set_phase_timing(self._handle(), 1 if on else 0))

    def PhaseMs(self):
        """(phase 1, phase 2) milliseconds of the last two-phase tangent assembly (events on the launch stream)"""
        a, b = C.c_double(0.0), C.c_double(0.0)
        check(_capi.lib().mimi_hip_domain_phase_ms(self._handle(), C.byref(a), C.byref(b)))
        return a.value, b.value

    def PhaseMsDetail(self):
        """(material pre-pass, integration / contraction kernel, row gather) milliseconds of the last two-phase tangent assembly"""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(_capi.lib().mimi_hip_domain_phase_ms_detail(self._handle(), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def ResetState(self):
        check(_capi.lib().mimi_hip_domain_reset_state(self._handle()))

    def __del__(self):
        try:
            if self._h is not None:
                _capi.lib().mimi_hip_domain_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # snake_case aliases
    prepare = Prepare
    add_domain_residual = AddDomainResidual
    add_domain_residual_and_grad = AddDomainResidualAndGrad
    domain_post_time_advance = DomainPostTimeAdvance


class RigidSphere:
    """Analytic rigid body standing in for NearestDistanceToSplines
    (coefficients/nearest_distance.hpp:215-288; splinepy's proximity query is not available)."""
    kind = 0

    def __init__(self, center, radius, coefficient=1.0e4):
        self.center, self.radius = [float(c) for c in center], float(radius)
        self.coefficient = float(coefficient)   # NearestDistanceBase::coefficient_ (nearest_distance.hpp:18)

    def params(self, dim):
        return self.center + [0.0] * (3 - dim) + [self.radius]


class RigidPlane:
    kind = 1

    def __init__(self, point, normal, coefficient=1.0e4):
        n = np.asarray(normal, dtype=np.float64)
        self.point, self.normal = [float(c) for c in point], list(n / np.linalg.norm(n))
        self.coefficient = float(coefficient)

    def params(self, dim):
        return self.point + [0.0] * (3 - dim) + self.normal + [0.0] * (3 - dim)


class RigidSpline:
    """One rigid boundary spline (curve in 2-D, surface in 3-D): what NearestDistanceToSplines holds
    (coefficients/nearest_distance.hpp:215-288).  Orientation: the normal (t_y, -t_x) / S_u x S_v must point out of the
    rigid body (nearest_distance.hpp:139-184)."""
    kind = 2

    def __init__(self, degrees, knots, control_points, weights=None, resolution=100, coefficient=1.0e4, max_iterations=-1):
        self.degrees = [int(p) for p in degrees]
        self.knots = [np.ascontiguousarray(k, dtype=np.float64) for k in knots]
        n = int(np.prod([len(k) - p - 1 for k, p in zip(self.knots, self.degrees)]))
        self.control_points = np.ascontiguousarray(control_points, dtype=np.float64).reshape(n, -1)
        self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64).reshape(n)
        self.resolution, self.max_iterations = int(resolution), int(max_iterations)
        self.coefficient = float(coefficient)

    def params(self, dim):
        return []

    def c_struct(self):
        s = _capi.SplineBody()
        s.para_dim = len(self.degrees)
        for k in range(s.para_dim):
            s.degree[k], s.n_knots[k] = self.degrees[k], len(self.knots[k])
            s.knots[k] = self.knots[k].ctypes.data
        s.control_points = self.control_points.ctypes.data
        s.weights = self.weights.ctypes.data if self.weights is not None else None
        s.kdtree_resolution, s.max_iterations = self.resolution, self.max_iterations
        return s


class NearestDistanceToSplines:
    """coefficients::NearestDistanceToSplines as bound in py/py_nearest_distance.cpp: add_spline / plant_kd_tree /
    coefficient.  `spline` is anything with the attributes degrees, knot_vectors, control_points and (optionally)
    weights -- a splinepy spline has them."""
    kind = 2

    def __init__(self):
        self._coefficient = 1.0e4       # nearest_distance.hpp:18
        self.tolerance = 1.0e-24        # nearest_distance.hpp:20 (the search here stops on the step size)
        self._splines, self._resolution = [], 100
        self._attached = []             # MortarContact integrators built on this scene

    @property
    def coefficient(self):
        return self._coefficient

    @coefficient.setter
    def coefficient(self, value):
        """the reference reads coefficient_ at every evaluation: a change after setup reaches the device handles"""
        self._coefficient = float(value)
        for c in getattr(self, "_attached", []):
            c.UpdateBody(penalty=self._coefficient)

    def add_spline(self, spline):
        self._splines.append(spline)
        return self

    def clear(self):
        self._splines.clear()

    def plant_kd_tree(self, resolution, nthreads=1):
        """PlantKdTree (nearest_distance.hpp:243-255).  examples/nl_contact.py moves the spline's control points and
        calls this before every step: the attached device handles get the moved body."""
        self._resolution = int(resolution)
        for c in self._attached:
            c.UpdateBody(spline=True)

    def size(self):
        return len(self._splines)

    def _body(self):
        if len(self._splines) != 1:
            raise RuntimeError("exactly one boundary spline is supported (nearest_distance.hpp:262-263)")
        sp = self._splines[0]
        w = getattr(sp, "weights", None)
        return RigidSpline(sp.degrees, sp.knot_vectors, sp.control_points, None if w is None else np.ravel(w),
                           resolution=self._resolution, coefficient=self.coefficient)

    def params(self, dim):
        return []

    def c_struct(self):
        self._rs = self._body()
        return self._rs.c_struct()


class MortarContact(NonlinearBase):
    """integrators::MortarContact (integrators/mortar_contact.hpp:23-172) against an analytic
    rigid body, on one face of a B-spline patch."""

    def __init__(self, nearest_distance_coeff, name, pattern, patch, axis, side, device=0, quadrature_order=-1,
                 element_box=None):
        super().__init__(name)
        self.element_box_ = element_box          # multi-GPU: only the faces of the elements of this slab
        self.nearest_distance_coeff_ = nearest_distance_coeff
        self.pattern_, self.patch_ = pattern, patch
        self.axis_, self.side_ = axis, side
        self.device_, self.quadrature_order_ = device, quadrature_order
        self._h = None
        self.last_area_ = 0.0
        self.last_pressure_ = 0.0
        self.last_force_ = np.zeros(patch.dim)

    def Prepare(self):
        from . import splines
        L = _capi.lib()
        p = self.patch_
        dofs, N, dN, weight = splines.face_tables(p, self.axis_, self.side_, self.quadrature_order_, self.element_box_)
        if len(dofs) == 0:
            raise RuntimeError("no marked boundary faces in this element box")
        t = _capi.ContactTables()
        t.dim = p.dim
        t.n_faces, t.n_dof = dofs.shape
        t.n_quad = weight.shape[1]
        t.n_nodes = p.n_nodes
        self._keep = [dofs, N, dN, weight, p.control_points]
        t.dofs, t.N, t.dN_dxi, t.weight = dofs.ctypes.data, N.ctypes.data, dN.ctypes.data, weight.ctypes.data
        t.x_ref = p.control_points.ctypes.data
        body = self.nearest_distance_coeff_
        t.body_kind = body.kind
        for i, v in enumerate(body.params(p.dim)):
            t.body[i] = v
        t.penalty = body.coefficient
        if body.kind == 2:
            self._spline_struct = body.c_struct()
            self._keep.append(body)
            t.spline = C.cast(C.pointer(self._spline_struct), C.c_void_p)
            if hasattr(body, "_attached"):
                body._attached.append(self)
        t.csr_rowptr = ptr(self.pattern_.rowptr, "int64").value
        t.csr_col = ptr(self.pattern_.col, "int32").value
        h = C.c_void_p()
        check(L.mimi_hip_contact_create(C.byref(t), self.device_, C.byref(h)))
        self._h = h
        self.n_marked_boundaries_ = int(t.n_faces)
        return self

    def _handle(self):
        if self._h is None:
            raise RuntimeError("Prepare() has not been called")
        return self._h

    def Synchronize(self):
        check(_capi.lib().mimi_hip_contact_synchronize(self._handle()))

    def UpdateBody(self, spline=False, penalty=-1.0):
        """the rigid body moved (spline=True: re-read it from nearest_distance_coeff_) and / or the penalty changed"""
        sp = None
        if spline:
            self._spline_struct = self.nearest_distance_coeff_.c_struct()
            sp = C.cast(C.pointer(self._spline_struct), C.c_void_p)
        check(_capi.lib().mimi_hip_contact_update_body(self._handle(), sp, float(penalty)))

    # -- the two halves of an evaluation, for element slabs on several GPUs (mimi_amd/parallel.py ShardedContact) -----
    def GapArea(self, current_u):
        """pass 1 only: nodal area / gap of this handle's faces"""
        self._follow_torch(current_u)
        check(_capi.lib().mimi_hip_contact_gap_area(self._handle(), fptr(current_u)))

    def MarkedNodes(self):
        n = C.c_int64(0)
        check(_capi.lib().mimi_hip_contact_marked_nodes(self._handle(), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.int32)
        check(_capi.lib().mimi_hip_contact_marked_nodes(self._handle(), ptr(out), out.size, C.byref(n)))
        return out

    def GetNodal(self, area, gap):
        self._follow_torch(area, gap)
        check(_capi.lib().mimi_hip_contact_nodal(self._handle(), 0, fptr(area), fptr(gap)))

    def SetNodal(self, area, gap):
        self._follow_torch(area, gap)
        check(_capi.lib().mimi_hip_contact_nodal(self._handle(), 1, fptr(area), fptr(gap)))

    def AddBoundaryResidualFromNodal(self, current_u, grad_factor, residual, grad=None):
        """pressure from the (summed) nodal area / gap, then pass 2"""
        self._follow_torch(current_u, residual, grad)
        check(_capi.lib().mimi_hip_contact_add_residual_from_nodal(self._handle(), fptr(current_u), float(grad_factor),
                                                                   fptr(residual), fptr(grad)))

    def SetTangentMode(self, mode):
        check(_capi.lib().mimi_hip_contact_set_tangent_mode(self._handle(), mode))

    def SetStream(self, stream):
        self._user_stream = bool(stream)
        self._user_stream_value = int(stream) if stream else 0
        check(_capi.lib().mimi_hip_contact_set_stream(self._handle(), C.c_void_p(stream) if stream else None))

    def _follow_torch(self, *buffers):
        if getattr(self, "_user_stream", False):
            return
        s = _capi.torch_stream_of(*buffers)
        if s is not None or getattr(self, "_followed", None):
            check(_capi.lib().mimi_hip_contact_set_stream(self._handle(), C.c_void_p(s) if s else None))
            self._followed = s

    def _history(self):
        out = np.zeros(5)
        check(_capi.lib().mimi_hip_contact_last_history(self._handle(), ptr(out)))
        self.last_area_, self.last_pressure_ = out[0], out[1]
        self.last_force_ = out[2:2 + self.patch_.dim].copy()

    # mortar_contact.cpp:297-351
    def AddBoundaryResidual(self, current_u, residual):
        self._follow_torch(current_u, residual)
        check(_capi.lib().mimi_hip_contact_add_residual(self._handle(), fptr(current_u), fptr(residual)))

    # mortar_contact.cpp:353-421
    def AddBoundaryResidualAndGrad(self, current_u, grad_factor, residual, grad_values):
        self._follow_torch(current_u, residual, grad_values)
        check(_capi.lib().mimi_hip_contact_add_residual_and_grad(self._handle(), fptr(current_u), float(grad_factor),
                                                                 fptr(residual), fptr(grad_values)))

    # mortar_contact.cpp:423-467
    def GapNorm(self, test_u, nthreads=-1):
        out = C.c_double(0.0)
        check(_capi.lib().mimi_hip_contact_gap_norm(self._handle(), ptr(test_u), C.byref(out)))
        return out.value

    # mortar_contact.cpp:469-488: record last_area_, last_force_, last_pressure_
    def BoundaryPostTimeAdvance(self, converged_u):
        self._history()

    def AveragePressure(self):
        n = C.c_int64(0)
        check(_capi.lib().mimi_hip_contact_get_pressure(self._handle(), None, 0, C.byref(n)))
        out = np.zeros(n.value)
        check(_capi.lib().mimi_hip_contact_get_pressure(self._handle(), ptr(out), out.size, C.byref(n)))
        return out

    def AddBoundaryGrad(self, current_u, grad):
        raise RuntimeError("Currently not implemented, use AddDomainResidualAndGrad")  # mortar_contact.hpp:142-149

    def __del__(self):
        try:
            if self._h is not None:
                _capi.lib().mimi_hip_contact_destroy(self._h)
                self._h = None
        except Exception:
            pass
