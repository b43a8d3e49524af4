"""The callers' steps around the assembly, device-resident: essential-dof elimination (forms/nonlinear.hpp:76-80,
112-115) and the reference's iterative linear solver (py/py_nonlinear_solid.cpp:329-339: mfem::GMRESSolver with an
mfem::DSmoother preconditioner) -- ctypes front-end of csrc/krylov.hip."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, fptr, ptr


class LinearSolver:
    """One per CSR pattern and list of essential dofs.  Vectors / CSR values may be numpy arrays (staged) or torch
    device tensors (in place)."""

    # mfem::GMRESSolver as configured by the reference (py_nonlinear_solid.cpp:331-336); kdim: mfem's default m
    rel_tol = 1e-8
    abs_tol = 1e-12
    max_iter = 300
    kdim = 50
    use_jacobi = True          # mfem::DSmoother

    def __init__(self, pattern, essential_dofs=None, device=0):
        self.pattern_ = pattern
        self.n_ = int(len(pattern.rowptr) - 1) if not hasattr(pattern.rowptr, "numel") else int(pattern.rowptr.numel() - 1)
        ess = np.ascontiguousarray(essential_dofs if essential_dofs is not None else np.zeros(0), dtype=np.int64)
        self._keep = (pattern.rowptr, pattern.col, ess)   # device arrays are used in place by the library
        h = C.c_void_p()
        check(_capi.lib().mimi_hip_linear_create(self.n_, ptr(pattern.rowptr, "int64"), ptr(pattern.col, "int32"),
                                                 ptr(ess, "int64") if ess.size else None,
                                                 ess.size, device, C.byref(h)))
        self._h = h
        self.final_iter_, self.final_norm_, self.converged_ = 0, 0.0, False

    def SetStream(self, stream):
        self._user_stream = bool(stream)
        check(_capi.lib().mimi_hip_linear_set_stream(self._h, C.c_void_p(stream) if stream else None))

    def _follow_torch(self, *buffers):
        # a handle that was never given a stream launches on torch's current stream when it gets CUDA tensors
        if getattr(self, "_user_stream", False):
            return
        s = _capi.torch_stream_of(*buffers)
        if s is not None or getattr(self, "_followed", None):
            check(_capi.lib().mimi_hip_linear_set_stream(self._h, C.c_void_p(s) if s else None))
            self._followed = s

    def RowGroup(self):
        """consecutive rows sharing one column list that the products read once (3, 2 or 1)"""
        return int(_capi.lib().mimi_hip_linear_info(self._h, 2))

    def NodeColumns(self):
        """True when the shared column list of a node's rows is made of node triples and is read as one index per node"""
        return bool(_capi.lib().mimi_hip_linear_info(self._h, 3))

    def Eliminate(self, r=None, A_values=None):
        """r[ess] = 0; A.EliminateRowCol(ess, DIAG_ONE)"""
        self._follow_torch(r, A_values)
        check(_capi.lib().mimi_hip_linear_eliminate(self._h, fptr(r), fptr(A_values)))

    def AddMult(self, A_values, x, y, alpha=1.0):
        """y += alpha A x (mfem::SparseMatrix::AddMult)"""
        self._follow_torch(A_values, x, y)
        check(_capi.lib().mimi_hip_linear_add_mult(self._h, fptr(A_values), fptr(x), float(alpha), fptr(y)))
        return y

    def Mult(self, A_values, b, x):
        """x = A^-1 b to the configured tolerances (x is overwritten: iterative_mode false)"""
        it, conv, nrm = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        self._follow_torch(A_values, b, x)
        check(_capi.lib().mimi_hip_linear_gmres(self._h, fptr(A_values), fptr(b), fptr(x), self.rel_tol, self.abs_tol,
                                                int(self.max_iter), int(self.kdim), 1 if self.use_jacobi else 0,
                                                C.byref(it), C.byref(nrm), C.byref(conv)))
        self.final_iter_, self.final_norm_, self.converged_ = it.value, nrm.value, bool(conv.value)
        return x

    def MultCG(self, A_values, b, x, rel_tol=1e-8, abs_tol=1e-12, max_iter=1000):
        """x = A^-1 b by preconditioned conjugate gradients: the mass solve of operators::NonlinearSolid
        (operators/nonlinear_solid.cpp:39-50,155)"""
        it, conv, nrm = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        self._follow_torch(A_values, b, x)
        check(_capi.lib().mimi_hip_linear_cg(self._h, fptr(A_values), fptr(b), fptr(x), rel_tol, abs_tol, int(max_iter),
                                             1 if self.use_jacobi else 0, C.byref(it), C.byref(nrm), C.byref(conv)))
        self.final_iter_, self.final_norm_, self.converged_ = it.value, nrm.value, bool(conv.value)
        return x

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _capi.lib().mimi_hip_linear_destroy(self._h)
                self._h = None
        except Exception:
            pass
