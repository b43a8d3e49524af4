"""ctypes binding of libmimi_hip.so (include/mimi_hip.h).  There is no CPU fallback:
if the library is missing or no HIP device is visible, calls raise."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


class Material(C.Structure):
    """mimi_hip_material"""
    _fields_ = [("kind", C.c_int32), ("hardening", C.c_int32),
                ("density", C.c_double), ("lambda_", C.c_double), ("mu", C.c_double),
                ("K", C.c_double), ("G", C.c_double),
                ("heat_fraction", C.c_double), ("specific_heat", C.c_double),
                ("initial_temperature", C.c_double), ("melting_temperature", C.c_double),
                ("sigma_y", C.c_double), ("n", C.c_double), ("eps0", C.c_double),
                ("sigma_sat", C.c_double), ("strain_constant", C.c_double),
                ("A", C.c_double), ("B", C.c_double), ("C", C.c_double), ("eps0_dot", C.c_double),
                ("reference_temperature", C.c_double), ("m", C.c_double),
                ("lin_isotropic_hardening", C.c_double), ("lin_kinematic_hardening", C.c_double)]


class DomainTables(C.Structure):
    """mimi_hip_domain_tables"""
    _fields_ = [("dim", C.c_int32), ("n_elements", C.c_int32), ("n_dof", C.c_int32), ("n_quad", C.c_int32),
                ("n_nodes", C.c_int64),
                ("dofs", C.c_void_p), ("dN_dX", C.c_void_p), ("weight_det", C.c_void_p),
                ("csr_rowptr", C.c_void_p), ("csr_col", C.c_void_p)]


class BSplinePatch(C.Structure):
    """mimi_hip_bspline_patch"""
    _fields_ = [("dim", C.c_int32), ("degree", C.c_int32 * 3), ("n_knots", C.c_int32 * 3),
                ("knots", C.c_void_p * 3), ("control_points", C.c_void_p), ("node_ids", C.c_void_p),
                ("quadrature_order", C.c_int32),
                ("element_begin", C.c_int32 * 3), ("element_end", C.c_int32 * 3),
                ("csr_rowptr", C.c_void_p), ("csr_col", C.c_void_p), ("weights", C.c_void_p)]


class ContactTables(C.Structure):
    """mimi_hip_contact_tables"""
    _fields_ = [("dim", C.c_int32), ("n_faces", C.c_int32), ("n_dof", C.c_int32), ("n_quad", C.c_int32),
                ("n_nodes", C.c_int64),
                ("dofs", C.c_void_p), ("N", C.c_void_p), ("dN_dxi", C.c_void_p), ("weight", C.c_void_p),
                ("x_ref", C.c_void_p),
                ("body_kind", C.c_int32), ("body", C.c_double * 8), ("penalty", C.c_double),
                ("csr_rowptr", C.c_void_p), ("csr_col", C.c_void_p), ("spline", C.c_void_p)]


class SplineBody(C.Structure):
    """mimi_hip_spline_body"""
    _fields_ = [("para_dim", C.c_int32), ("degree", C.c_int32 * 2), ("n_knots", C.c_int32 * 2),
                ("knots", C.c_void_p * 2), ("control_points", C.c_void_p), ("weights", C.c_void_p),
                ("kdtree_resolution", C.c_int32), ("max_iterations", C.c_int32)]


EXPORTS = [
    "mimi_hip_last_error", "mimi_hip_abi_version", "mimi_hip_device_count",
    "mimi_hip_material_set_young_poisson",
    "mimi_hip_domain_create", "mimi_hip_domain_create_bspline", "mimi_hip_domain_destroy",
    "mimi_hip_domain_set_dt", "mimi_hip_domain_set_tangent_mode", "mimi_hip_domain_set_stream",
    "mimi_hip_domain_synchronize", "mimi_hip_domain_add_residual",
    "mimi_hip_domain_add_residual_and_grad", "mimi_hip_domain_add_residual_and_grad_from",
    "mimi_hip_domain_post_time_advance",
    "mimi_hip_domain_get_state", "mimi_hip_domain_reset_state", "mimi_hip_domain_info",
    "mimi_hip_domain_set_phase_timing", "mimi_hip_domain_phase_ms", "mimi_hip_domain_phase_ms_detail",
    "mimi_hip_bspline_sparsity", "mimi_hip_bspline_sparsity_rows",
    "mimi_hip_contact_create", "mimi_hip_contact_destroy", "mimi_hip_contact_set_tangent_mode",
    "mimi_hip_contact_set_stream", "mimi_hip_contact_synchronize", "mimi_hip_contact_add_residual",
    "mimi_hip_contact_add_residual_and_grad", "mimi_hip_contact_gap_norm",
    "mimi_hip_contact_last_history", "mimi_hip_contact_get_pressure",
    "mimi_hip_contact_update_body", "mimi_hip_contact_gap_area", "mimi_hip_contact_marked_nodes", "mimi_hip_contact_nodal",
    "mimi_hip_contact_add_residual_from_nodal",
    "mimi_hip_linear_create", "mimi_hip_linear_destroy", "mimi_hip_linear_set_stream", "mimi_hip_linear_info", "mimi_hip_linear_eliminate",
    "mimi_hip_linear_add_mult",
    "mimi_hip_linear_gmres", "mimi_hip_linear_cg",
    "mimi_hip_domain_integrate", "mimi_hip_domain_gather",
    "mimi_hip_rows_zero", "mimi_hip_rows_pack", "mimi_hip_rows_unpack_add",
    "mimi_hip_entries_pack", "mimi_hip_entries_unpack_add",
]


def _header_abi_version():
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mimi_hip.h")) as f:
            for line in f:
                if line.startswith("#define MIMI_HIP_ABI_VERSION"):
                    return int(line.split()[2])
    except OSError:
        pass
    return None


def library_path():
    return _build.LIB


def lib():
    """Load libmimi_hip.so (building it if hipcc is around and the file is stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MIMI_HIP_LIBRARY") or _build.LIB      # (another build of the same sources: timing experiments)
    import shutil
    if path == _build.LIB and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        _build.build()          # no-op unless a source or the header is newer than the library
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: run `python -m mimi_amd.build` (hipcc --offload-arch=gfx950). "
            "mimi_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7; when it is
    # loaded first our NEEDED entry resolves to that same copy (same soname), and device
    # pointers / streams can be shared with torch.  Loading ours first would hand torch a
    # runtime it was not built against.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(path)
    expected = _header_abi_version()
    if expected is not None and L.mimi_hip_abi_version() != expected:
        raise RuntimeError(f"{path} has ABI version {L.mimi_hip_abi_version()}, include/mimi_hip.h declares {expected}: "
                           "rebuild with `python -m mimi_amd.build`")
    L.mimi_hip_last_error.restype = C.c_char_p
    L.mimi_hip_domain_info.restype = C.c_int64
    L.mimi_hip_domain_info.argtypes = [C.c_void_p, C.c_int]
    L.mimi_hip_domain_add_residual.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_add_residual_and_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_add_residual_and_grad_from.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_post_time_advance.argtypes = [C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_set_dt.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    L.mimi_hip_domain_set_tangent_mode.argtypes = [C.c_void_p, C.c_int]
    L.mimi_hip_domain_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_synchronize.argtypes = [C.c_void_p]
    L.mimi_hip_domain_destroy.argtypes = [C.c_void_p]
    L.mimi_hip_domain_get_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.mimi_hip_domain_reset_state.argtypes = [C.c_void_p]
    L.mimi_hip_domain_set_phase_timing.argtypes = [C.c_void_p, C.c_int]
    L.mimi_hip_domain_phase_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_phase_ms_detail.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mimi_hip_domain_create_bspline.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.mimi_hip_bspline_sparsity.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_bspline_sparsity_rows.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_integrate.argtypes = [C.c_void_p, C.c_void_p]
    L.mimi_hip_domain_gather.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_rows_zero.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.mimi_hip_rows_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_rows_unpack_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
    L.mimi_hip_entries_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_entries_unpack_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
    L.mimi_hip_material_set_young_poisson.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.mimi_hip_material_set_young_poisson.restype = None
    L.mimi_hip_contact_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.mimi_hip_contact_destroy.argtypes = [C.c_void_p]
    L.mimi_hip_contact_set_tangent_mode.argtypes = [C.c_void_p, C.c_int]
    L.mimi_hip_contact_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.mimi_hip_contact_synchronize.argtypes = [C.c_void_p]
    L.mimi_hip_contact_add_residual.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_contact_add_residual_and_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    L.mimi_hip_contact_gap_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_contact_last_history.argtypes = [C.c_void_p, C.c_void_p]
    L.mimi_hip_contact_get_pressure.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.mimi_hip_contact_update_body.argtypes = [C.c_void_p, C.c_void_p, C.c_double]
    L.mimi_hip_contact_gap_area.argtypes = [C.c_void_p, C.c_void_p]
    L.mimi_hip_contact_marked_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.mimi_hip_contact_nodal.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.mimi_hip_contact_add_residual_from_nodal.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    L.mimi_hip_linear_create.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.mimi_hip_linear_destroy.argtypes = [C.c_void_p]
    L.mimi_hip_linear_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.mimi_hip_linear_info.argtypes = [C.c_void_p, C.c_int]
    L.mimi_hip_linear_info.restype = C.c_int64
    L.mimi_hip_linear_eliminate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_linear_add_mult.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    L.mimi_hip_linear_gmres.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mimi_hip_linear_cg.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def check(status):
    """non-zero status -> RuntimeError carrying the library's message (the reference throws
    std::runtime_error -> Python RuntimeError through pybind11, utils/print.hpp:47-56)."""
    if status != 0:
        raise RuntimeError(lib().mimi_hip_last_error().decode())


def ptr(x, dtype=None):
    """void* of a numpy array (host) or a torch tensor (host or device) or None.  dtype: what the C side reads
    ("float64", "int32", "int64"); a buffer of another type raises instead of being reinterpreted."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        assert x.flags.c_contiguous
        if dtype is not None and x.dtype != np.dtype(dtype):
            raise TypeError(f"expected a {dtype} array, got {x.dtype}")
        return x.ctypes.data_as(C.c_void_p)
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous()
        if dtype is not None and str(x.dtype).split(".")[-1] != dtype:
            raise TypeError(f"expected a {dtype} tensor, got {x.dtype}")
        return C.c_void_p(x.data_ptr())
    if isinstance(x, int):
        return C.c_void_p(x)
    raise TypeError(type(x))


def fptr(x):
    """ptr() of a float64 buffer (u, r, CSR values)"""
    return ptr(x, "float64")


def torch_stream_of(*buffers):
    """cuda_stream of torch's current stream on the device of the first CUDA tensor among `buffers`, else None: a handle
    that was never given a stream launches on it, so that what torch enqueued on these tensors before the call (zero
    fills, copies) is ordered before the kernels, and what it enqueues afterwards behind them."""
    for b in buffers:
        if b is not None and hasattr(b, "is_cuda") and b.is_cuda:
            import torch
            s = torch.cuda.current_stream(b.device).cuda_stream
            return s if s else STREAM_NULL       # torch's default stream is the device's null stream (handle 0)
    return None


STREAM_NULL = 2 ** 64 - 1        # MIMI_HIP_STREAM_NULL of include/mimi_hip.h
