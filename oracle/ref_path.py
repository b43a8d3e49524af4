"""TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

ctypes front-end of oracle/ref_path.c and oracle/contact_path.c (the plain-C
restatement of the reference's integrators) plus the table plumbing from
oracle/iga.py.  Nothing in ``mimi_amd`` imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import iga

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle_ref.so")

MAT_NEOHOOKEAN, MAT_J2, MAT_STVK, MAT_J2LINEAR, MAT_J2SIMO, MAT_J2LOG = 0, 1, 2, 3, 4, 5
MAT_KINDS = dict(neohookean=0, j2=1, stvk=2, j2linear=3, j2simo=4, j2log=5)
HARD = dict(PowerLaw=0, Voce=1, JohnsonCook=2, JohnsonCookRate=3, JohnsonCookTempRate=4,
            JohnsonCookConstTemp=5)
TANGENT_FD, TANGENT_EXACT, TANGENT_NONE = 0, 1, 2   # NONE: timing probe, zeroing + reduction passes only


def build(force=False):
    """gcc-compile the C restatement (a few seconds).  Building the checker is not
    using it: __graft_entry__.build() calls this too."""
    srcs = [os.path.join(_HERE, f) for f in ("ref_path.c", "contact_path.c")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


class _Material(C.Structure):
    _fields_ = [("kind", C.c_int),
                ("density", C.c_double), ("lambda_", C.c_double), ("mu", C.c_double),
                ("K", C.c_double), ("G", C.c_double),
                ("heat_fraction", C.c_double), ("specific_heat", C.c_double),
                ("initial_temperature", C.c_double), ("melting_temperature", C.c_double),
                ("hard_kind", C.c_int),
                ("sigma_y", C.c_double), ("n", C.c_double), ("eps0", C.c_double),
                ("sigma_sat", C.c_double), ("strain_constant", C.c_double),
                ("A", C.c_double), ("B", C.c_double), ("C", C.c_double), ("eps0_dot", C.c_double),
                ("reference_temperature", C.c_double), ("m", C.c_double),
                ("const_temperature_contribution", C.c_double),
                ("lin_isotropic_hardening", C.c_double), ("lin_kinematic_hardening", C.c_double),
                ("lin_sigma_y", C.c_double)]


class _Domain(C.Structure):
    _fields_ = [("dim", C.c_int), ("n_el", C.c_int), ("n_dof", C.c_int), ("n_q", C.c_int),
                ("n_vdofs", C.c_int),
                ("v_dofs", C.c_void_p), ("a_ids", C.c_void_p), ("dN_dX", C.c_void_p),
                ("weight", C.c_void_p), ("det", C.c_void_p),
                ("mat", _Material),
                ("plastic_strain", C.c_void_p), ("eqps", C.c_void_p), ("temperature", C.c_void_p),
                ("dt", C.c_double), ("state2", C.c_void_p)]


class _Contact(C.Structure):
    _fields_ = [("dim", C.c_int), ("n_faces", C.c_int), ("n_dof", C.c_int), ("n_q", C.c_int),
                ("n_vdofs", C.c_int), ("n_marked", C.c_int),
                ("v_dofs", C.c_void_p), ("a_ids", C.c_void_p), ("local_dofs", C.c_void_p),
                ("N", C.c_void_p), ("dN_dxi", C.c_void_p), ("weight", C.c_void_p),
                ("x_ref", C.c_void_p),
                ("body_kind", C.c_int), ("body", C.c_double * 8), ("penalty", C.c_double),
                ("area", C.c_void_p), ("gap", C.c_void_p), ("pressure", C.c_void_p),
                ("last_area", C.c_double), ("last_pressure", C.c_double),
                ("last_force", C.c_double * 3),
                ("sp_para_dim", C.c_int), ("sp_p", C.c_int * 2), ("sp_n_knots", C.c_int * 2),
                ("sp_knots", C.c_void_p * 2), ("sp_ctrl", C.c_void_p), ("sp_weights", C.c_void_p),
                ("sp_resolution", C.c_int), ("sp_max_iterations", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_contact_gap_norm.restype = C.c_double
    return _lib


def lame(young, poisson):
    """MaterialBase::SetYoungPoisson (materials.cpp:7-14)."""
    lam = young * poisson / ((1 + poisson) * (1 - 2 * poisson))
    mu = young / (2.0 * (1.0 + poisson))
    K = young / (3.0 * (1.0 - (2.0 * poisson)))
    return lam, mu, K, mu


def make_material(kind, young, poisson, density=1.0, hardening=None, heat_fraction=0.9,
                  specific_heat=450.0, initial_temperature=20.0, melting_temperature=1500.0,
                  isotropic_hardening=0.0, kinematic_hardening=0.0, sigma_y=0.0):
    """kind: 'neohookean' | 'j2' | 'stvk' | 'j2linear' | 'j2simo' | 'j2log'.  hardening: dict(kind=..., A=..., ...);
    isotropic_hardening / kinematic_hardening / sigma_y: J2Linear only."""
    m = _Material()
    lam, mu, K, G = lame(young, poisson)
    m.kind = MAT_KINDS[kind]
    m.lin_isotropic_hardening, m.lin_kinematic_hardening, m.lin_sigma_y = isotropic_hardening, kinematic_hardening, sigma_y
    m.density, m.lambda_, m.mu, m.K, m.G = density, lam, mu, K, G
    m.heat_fraction, m.specific_heat = heat_fraction, specific_heat
    m.initial_temperature, m.melting_temperature = initial_temperature, melting_temperature
    if hardening is not None:
        h = dict(hardening)
        m.hard_kind = HARD[h.pop("kind")]
        # JohnsonCookRateDependentHardening::C_ is uninitialised unless set
        # (material_hardening.hpp:156); the golden fixtures correspond to C = 0.
        h.setdefault("C", 0.0)
        for k, v in h.items():
            setattr(m, k, v)
        if m.hard_kind == HARD["JohnsonCookConstTemp"]:
            # material_hardening.hpp:310-318 SetTemperature
            m.const_temperature_contribution = 1.0 - (
                (initial_temperature - m.reference_temperature)
                / (melting_temperature - m.reference_temperature)) ** m.m
    return m


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DomainOracle:
    """integrators::NonlinearSolid restated (nonlinear_solid.{hpp,cpp}) over the
    tables of a Patch."""

    def __init__(self, patch, material, quadrature_order=-1, n_threads=1, with_a_ids=True, elements=None, with_sparsity=True):
        """elements: integrate only this subset (tables and state of those elements); with_sparsity=False: no CSR pattern at
        all (element-level calls only: element_residual_and_grad) -- for sampled checks on meshes whose pattern is too big
        to build on the host"""
        self.patch = patch
        self.n_threads = n_threads
        t = patch.tables(quadrature_order, elements=elements)
        self.tables = t
        self.conn = np.ascontiguousarray(t["conn"], dtype=np.int32)
        self.v_dofs = np.ascontiguousarray(patch.vdofs(self.conn), dtype=np.int32)
        # (n_dof x dim) column-major per point == [e][q][J][a]
        self.dN_dX = np.ascontiguousarray(np.transpose(t["dN_dX"], (0, 1, 3, 2)))
        self.weight = np.ascontiguousarray(t["weight"])
        self.det = np.ascontiguousarray(t["det"])
        if with_sparsity:
            self.rowptr, self.col = patch.sparsity()
            self.nnz = int(self.rowptr[-1])
        else:
            self.rowptr = self.col = None
            self.nnz = 0
        self.a_ids = None
        if with_a_ids and with_sparsity:
            ids = patch.a_ids(self.rowptr, self.col)
            if elements is not None:
                ids = ids[elements]
            assert ids.max() < 2 ** 31
            self.a_ids = np.ascontiguousarray(ids, dtype=np.int32)
        ne, nq = self.weight.shape
        dim = patch.dim
        self.plastic_strain = np.zeros((ne, nq, dim * dim))
        self.state2 = np.zeros((ne, nq, dim * dim))
        eye = np.eye(dim).ravel()
        if material.kind == MAT_J2SIMO:      # materials.cpp:199-200: be_old = F_old = I
            self.plastic_strain[:] = eye
            self.state2[:] = eye
        elif material.kind == MAT_J2LOG:     # materials.cpp:244: Fp_inv = I
            self.plastic_strain[:] = eye
        self.eqps = np.zeros((ne, nq))
        self.temperature = np.full((ne, nq), material.initial_temperature)
        d = _Domain()
        d.dim, d.n_el, d.n_dof, d.n_q, d.n_vdofs = dim, ne, patch.n_dof, nq, patch.n_vdofs
        d.v_dofs, d.dN_dX, d.weight, d.det = map(_ptr, (self.v_dofs, self.dN_dX, self.weight, self.det))
        d.a_ids = _ptr(self.a_ids) if self.a_ids is not None else None
        d.mat = material
        d.plastic_strain, d.eqps, d.temperature = map(_ptr, (self.plastic_strain, self.eqps, self.temperature))
        d.dt = 0.0
        d.state2 = _ptr(self.state2)
        self.d = d
        self.material = material
        self.has_states = material.kind not in (MAT_NEOHOOKEAN, MAT_STVK)

    def set_dt(self, dt):
        self.d.dt = float(dt)

    def _check(self, status):
        if status == -1:
            raise MemoryError("oracle allocation failed")
        if status != 0:
            # print.hpp:47-56 PrintAndThrowError -> std::runtime_error
            raise RuntimeError("ScalarSolve: root not bracketed / failed to converge")

    def add_domain_residual(self, u, r):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert r.dtype == np.float64 and r.flags.c_contiguous
        self._check(lib().oracle_add_domain_residual(C.byref(self.d), _ptr(u), _ptr(r), self.n_threads))

    def add_domain_residual_and_grad(self, u, grad_factor, r, A, mode=TANGENT_FD):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert A.dtype == np.float64 and A.flags.c_contiguous and A.size == self.nnz
        self._check(lib().oracle_add_domain_residual_and_grad(
            C.byref(self.d), _ptr(u), C.c_double(grad_factor), _ptr(r), _ptr(A), C.c_long(self.nnz),
            self.n_threads, mode))

    def domain_post_time_advance(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        self._check(lib().oracle_domain_post_time_advance(C.byref(self.d), _ptr(u), self.n_threads))

    def element_residual_and_grad(self, e, u, mode=TANGENT_FD):
        nt = self.patch.n_dof * self.patch.dim
        R = np.zeros(nt)
        K = np.zeros(nt * nt)
        u = np.ascontiguousarray(u, dtype=np.float64)
        self._check(lib().oracle_element_residual_and_grad(C.byref(self.d), int(e), _ptr(u), mode, _ptr(R), _ptr(K)))
        return R, K.reshape(nt, nt).T.copy()   # K[r, c]


def point_pk1(material, F, dt=1.0, plastic_strain=None, eqps=0.0, temperature=20.0, state2=None):
    """(P, A) at one point; F, P row-major [i, J]; A[i, J, j, L] = dP_iJ/dF_jL.  plastic_strain / state2: the
    material's first / second state matrix (row-major), see oracle_domain in ref_path.c."""
    dim = F.shape[0]
    Fc = np.ascontiguousarray(F.T)            # column-major storage
    ps = np.zeros(dim * dim) if plastic_strain is None else np.ascontiguousarray(plastic_strain.T).ravel()
    s2 = np.zeros(dim * dim) if state2 is None else np.ascontiguousarray(state2.T).ravel()
    P = np.zeros(dim * dim)
    A = np.zeros(dim ** 4)
    st = lib().oracle_point_pk1(C.byref(material), dim, C.c_double(dt), _ptr(Fc), _ptr(ps),
                                C.c_double(eqps), C.c_double(temperature), _ptr(P), _ptr(A), _ptr(s2))
    if st:
        raise RuntimeError("ScalarSolve failed")
    return P.reshape(dim, dim).T.copy(), A.reshape(dim, dim, dim, dim)


class ContactOracle:
    """integrators::MortarContact restated (mortar_contact.{hpp,cpp}) against an
    analytic rigid body on one face of a Patch."""

    def __init__(self, patch, axis, side, body, penalty=1.0e4, quadrature_order=-1,
                 rowptr=None, col=None):
        self.patch = patch
        dim = patch.dim
        ft = patch.face_tables(axis, side, quadrature_order)
        self.conn = np.ascontiguousarray(ft["conn"], dtype=np.int32)
        nf, nd = self.conn.shape
        self.v_dofs = np.ascontiguousarray(
            np.concatenate([self.conn * dim + c for c in range(dim)], axis=1), dtype=np.int32)
        self.N = np.ascontiguousarray(ft["N"])
        self.dN_dxi = np.ascontiguousarray(np.transpose(ft["dN_dxi"], (0, 1, 3, 2)))
        self.weight = np.ascontiguousarray(ft["weight"])
        self.x_ref = np.ascontiguousarray(np.transpose(patch.ctrl[self.conn], (0, 2, 1)))
        # mortar_contact.cpp:41-76: sorted unique marked dofs -> dense local numbering
        marked = np.unique(self.conn)
        lut = -np.ones(patch.n_nodes, dtype=np.int64)
        lut[marked] = np.arange(marked.size)
        self.marked_nodes = marked
        self.local_dofs = np.ascontiguousarray(lut[self.conn], dtype=np.int32)
        self.n_marked = int(marked.size)
        self.area = np.zeros(self.n_marked)
        self.gap = np.zeros(self.n_marked)
        self.pressure = np.zeros(self.n_marked)
        self.a_ids = None
        if rowptr is not None:
            nt = nd * dim
            ids = np.zeros((nf, nt * nt), dtype=np.int64)
            vd = self.v_dofs.astype(np.int64)
            for f in range(nf):
                for ir in range(nt):
                    s, t = rowptr[vd[f, ir]], rowptr[vd[f, ir] + 1]
                    ids[f, np.arange(nt) * nt + ir] = s + np.searchsorted(col[s:t], vd[f])
            self.a_ids = np.ascontiguousarray(ids, dtype=np.int32)
        c = _Contact()
        c.dim, c.n_faces, c.n_dof, c.n_q = dim, nf, nd, self.weight.shape[1]
        c.n_vdofs, c.n_marked = patch.n_vdofs, self.n_marked
        c.v_dofs, c.local_dofs = _ptr(self.v_dofs), _ptr(self.local_dofs)
        c.a_ids = _ptr(self.a_ids) if self.a_ids is not None else None
        c.N, c.dN_dxi, c.weight, c.x_ref = map(_ptr, (self.N, self.dN_dxi, self.weight, self.x_ref))
        if body["kind"] == "spline":
            # dict(kind="spline", degrees=[..], knots=[..], control_points=[n, dim], weights=None|[n], resolution=..)
            c.body_kind = 2
            vals = []
            self._spline = [np.ascontiguousarray(k, dtype=np.float64) for k in body["knots"]]
            ctrl = np.ascontiguousarray(body["control_points"], dtype=np.float64)
            w = body.get("weights")
            w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
            self._spline += [ctrl, w]
            c.sp_para_dim = len(body["degrees"])
            for k, (p_, kn) in enumerate(zip(body["degrees"], self._spline[:c.sp_para_dim])):
                c.sp_p[k], c.sp_n_knots[k] = int(p_), len(kn)
                c.sp_knots[k] = kn.ctypes.data
            c.sp_ctrl = ctrl.ctypes.data
            c.sp_weights = w.ctypes.data if w is not None else None
            c.sp_resolution = int(body.get("resolution", 100))
            c.sp_max_iterations = int(body.get("max_iterations", -1))
        elif body["kind"] == "sphere":
            c.body_kind = 0
            vals = list(body["center"]) + [0.0] * (3 - dim) + [body["radius"]]
        else:
            c.body_kind = 1
            vals = list(body["point"]) + [0.0] * (3 - dim) + list(body["normal"]) + [0.0] * (3 - dim)
        for i, v in enumerate(vals):
            c.body[i] = v
        c.penalty = penalty
        c.area, c.gap, c.pressure = map(_ptr, (self.area, self.gap, self.pressure))
        self.c = c

    def add_boundary_residual(self, u, r):
        u = np.ascontiguousarray(u, dtype=np.float64)
        lib().oracle_contact_add_residual(C.byref(self.c), _ptr(u), _ptr(r))

    def add_boundary_residual_and_grad(self, u, grad_factor, r, A, mode=TANGENT_FD):
        u = np.ascontiguousarray(u, dtype=np.float64)
        lib().oracle_contact_add_residual_and_grad(C.byref(self.c), _ptr(u), C.c_double(grad_factor),
                                                   _ptr(r), _ptr(A), mode)

    def gap_norm(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        return float(lib().oracle_contact_gap_norm(C.byref(self.c), _ptr(u)))

    @property
    def last_area(self):
        return self.c.last_area

    @property
    def last_force(self):
        return np.array(self.c.last_force[:self.patch.dim])

    @property
    def last_pressure(self):
        return self.c.last_pressure
