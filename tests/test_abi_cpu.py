"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/mimi_hip.h declares; the product refuses to compute without a device
(no CPU fallback); the product never imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mimi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(mimi_hip_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("mimi_hip_domain_create", "mimi_hip_domain_create_bspline", "mimi_hip_domain_add_residual",
                 "mimi_hip_domain_add_residual_and_grad", "mimi_hip_domain_add_residual_and_grad_from",
                 "mimi_hip_domain_post_time_advance",
                 "mimi_hip_contact_add_residual", "mimi_hip_contact_add_residual_and_grad",
                 "mimi_hip_contact_gap_norm", "mimi_hip_bspline_sparsity", "mimi_hip_last_error"):
        assert must in names


def test_library_builds_loads_and_exports_every_declared_symbol():
    from mimi_amd import build, _capi
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_capi.EXPORTS) == declared_functions()
    lib.mimi_hip_abi_version.restype = ctypes.c_int
    assert lib.mimi_hip_abi_version() == 12
    # plain C ABI: no C++ / torch symbols leak through the public names
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    public = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert all(not n.startswith("mimi_hip_") or "_Z" not in n for n in public)


def test_no_cpu_fallback_without_a_device():
    """On a host without a GPU every compute entry point must fail loudly."""
    from mimi_amd import _capi
    L = _capi.lib()
    if L.mimi_hip_device_count() > 0:
        pytest.skip("a HIP device is visible")
    import numpy as np
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    patch = mimi_amd.BSplinePatch.block((2, 2), 2)
    with pytest.raises(RuntimeError, match="no HIP device"):
        CSRPattern.of_bspline_patch(patch)
    mat = mimi_amd.CompressibleOgdenNeoHookean()
    mat.set_young_poisson(2100, 0.3)
    pat = CSRPattern(np.zeros(patch.n_vdofs + 1, dtype=np.int64), np.zeros(1, dtype=np.int32), 0)
    with pytest.raises(RuntimeError, match="no HIP device"):
        NonlinearSolid("domain", mat, pat, patch=patch).Prepare()


def test_product_does_not_import_the_oracle():
    code = ("import sys; sys.path.insert(0, %r); import mimi_amd, mimi_amd.integrators, mimi_amd.parallel, "
            "mimi_amd.splines, mimi_amd.materials; "
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; "
            "assert not bad, bad" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mimi_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f


def test_material_surface_matches_reference_names():
    """attribute names of the reference's pybind11 classes (py_material.cpp, py_hardening.cpp)"""
    import mimi_amd
    m = mimi_amd.J2()
    for attr in ("density", "viscosity", "hardening", "heat_fraction", "specific_heat", "initial_temperature",
                 "melting_temperature", "set_young_poisson", "set_lame"):
        assert hasattr(m, attr)
    h = mimi_amd.JohnsonCookTemperatureAndRateDependentHardening()
    for attr in ("A", "B", "n", "C", "eps0_dot", "reference_temperature", "m"):
        assert hasattr(h, attr)
    m.set_young_poisson(2100, 0.3)
    assert abs(m.lambda_ - 2100 * 0.3 / (1.3 * 0.4)) < 1e-9 and abs(m.mu - 2100 / 2.6) < 1e-9
    m2 = mimi_amd.CompressibleOgdenNeoHookean()
    m2.set_lame(m.lambda_, m.mu)
    assert abs(m2.young - 2100) < 1e-9 and abs(m2.poisson - 0.3) < 1e-12
