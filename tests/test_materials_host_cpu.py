"""The device constitutive routines (mimi_amd/csrc/materials.hpp, materials_other.hpp) compiled for the host
(tests/host_materials.hip, hipcc) against the oracle, point by point: PK1 stress <= 1e-12, tangent <= 1e-11 for the
closed forms (neo-Hookean, J2) and <= 1e-10 for the dual-number tangents of the other materials (the oracle's
difference-quotient tangent and the return map's own tolerance bound that comparison, not the device code)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from _cases import oracle_material
from test_domain_gpu import product_material

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def host_lib():
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = os.path.join(HERE, "_build", "libhost_materials.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "include"), "-o", out, os.path.join(HERE, "host_materials.hip")])
    return C.CDLL(out)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def sym(a):
    return 0.5 * (a + a.T)


def random_state(name, dim, rng, fresh):
    """(first matrix, second matrix, eqps): the material's initial state or a plausible advanced one"""
    eye = np.eye(dim)
    if fresh:
        m1 = eye.copy() if name in ("j2simo", "j2log") else np.zeros((dim, dim))
        m2 = eye.copy() if name == "j2simo" else np.zeros((dim, dim))
        return m1, m2, 0.0
    if name in ("j2", "j2linear"):
        ep = sym(0.01 * rng.standard_normal((dim, dim)))
        ep -= np.trace(ep) / dim * eye
        beta = sym(0.5 * rng.standard_normal((dim, dim))) if name == "j2linear" else np.zeros((dim, dim))
        beta -= np.trace(beta) / dim * eye      # the back stress evolves along the (trace-free) relative stress
        return ep, beta, 0.02
    if name == "j2simo":
        g = eye + 0.02 * rng.standard_normal((dim, dim))
        return g @ g.T, eye + 0.02 * rng.standard_normal((dim, dim)), 0.02
    if name == "j2log":
        return eye + 0.02 * rng.standard_normal((dim, dim)), np.zeros((dim, dim)), 0.02
    return np.zeros((dim, dim)), np.zeros((dim, dim)), 0.0


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("name", ["neohook", "j2", "stvk", "j2linear", "j2simo", "j2log"])
def test_device_materials_on_host_vs_oracle(host_lib, name, dim):
    from oracle import ref_path as rp
    mo = oracle_material(name)
    mp = product_material(name)._c_struct()
    sigma_y_ref = 70.0
    rng = np.random.default_rng(7 + dim)
    tol_A = 1e-11 if name in ("neohook", "j2") else 1e-10
    n_plastic = 0
    for trial in range(120):
        scale = 10 ** rng.uniform(-2.5, -0.9)
        F = np.eye(dim) + scale * rng.standard_normal((dim, dim))
        m1, m2, eqps = random_state(name, dim, rng, fresh=trial % 2 == 0)
        Po, Ao = rp.point_pk1(mo, F, dt=0.5, plastic_strain=m1, eqps=eqps, temperature=20.0, state2=m2)
        Fc = np.ascontiguousarray(F.T).ravel()
        P, A = np.zeros(dim * dim), np.zeros(dim ** 4)
        a1, a2 = np.ascontiguousarray(m1.T).ravel().copy(), np.ascontiguousarray(m2.T).ravel().copy()
        st = host_lib.host_point(C.byref(mp), C.c_double(sigma_y_ref), dim, C.c_double(0.5), ptr(Fc), ptr(a1), ptr(a2),
                                 C.c_double(eqps), C.c_double(20.0), ptr(P), ptr(A))
        assert st == 0
        Pg, Ag = P.reshape(dim, dim).T, A.reshape(dim, dim, dim, dim)
        assert np.abs(Pg - Po).max() <= 1e-12 * max(np.abs(Po).max(), 1.0), (trial, scale)
        assert np.abs(Ag - Ao).max() <= tol_A * np.abs(Ao).max(), (trial, scale, np.abs(Ag - Ao).max() / np.abs(Ao).max())
        # the committed state: DomainPostTimeAdvance at this point
        if name in ("neohook", "stvk"):
            continue
        e, T = C.c_double(eqps), C.c_double(20.0)
        st = host_lib.host_accumulate(C.byref(mp), C.c_double(sigma_y_ref), dim, C.c_double(0.5), ptr(Fc), ptr(a1), ptr(a2),
                                      C.byref(e), C.byref(T))
        assert st == 0
        n_plastic += e.value > eqps
    if name not in ("neohook", "stvk"):
        assert n_plastic > 20      # the plastic branch was exercised


def test_pow_positive_against_long_double_pow(host_lib):
    """csrc/materials.hpp pow_positive: exp(q ln x) written out for the return-map Newton (hardening laws with eqps^n,
    material_hardening.hpp:79-346).  Against numpy's long-double power over the strains and exponents that occur: the error
    stays under 3 ulp x (|q ln x| + 1) -- the conditioning of exp(q ln x) -- i.e. under 3e-14 relative everywhere here."""
    rng = np.random.default_rng(11)
    n = 200000
    x = 10.0 ** rng.uniform(-13.0, 1.0, n)
    q = rng.uniform(-0.9, 4.0, n)
    # the exponents of the fixtures' hardening laws, at strains from the first Newton iterate to large ones
    x[:6], q[:6] = [1e-13, 1e-9, 1e-6, 1e-3, 0.05, 0.5], 0.2835 - 1.0
    out = np.empty(n)
    host_lib.host_pow_positive(n, x.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    ref = np.power(x.astype(np.longdouble), q.astype(np.longdouble))
    rel = np.abs((out.astype(np.longdouble) - ref) / ref).astype(np.float64)
    ulp = 1.1102230246251565e-16
    assert np.all(rel <= 3.0 * ulp * (np.abs(q * np.log(x)) + 1.0))
    assert rel.max() < 3e-14 and rel[:6].max() < 2e-15


def test_pow_positive_outside_its_reduction_range(host_lib):
    """Arguments a diverging return-map Newton can hand to pow_positive: x = +inf, and |q ln x| far
    beyond the exponent range -- answered as the library pow answers them (inf / 0 / 1, NaN through), never by an
    out-of-range double -> int conversion."""
    x = np.array([np.inf, np.inf, np.inf, 1e300, 1e300, 1e-300, 1e-300, 2.0, 0.5, 1e308, np.nan, 3.0])
    q = np.array([0.5, -0.5, 0.0, 1e10, -1e10, 1e10, -1e10, 1e300, 1e300, 2.0, 0.3, np.nan])
    out = np.empty(x.size)
    host_lib.host_pow_positive(x.size, x.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    with np.errstate(all="ignore"):
        ref = np.power(x, q)
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(out[ok], ref[ok])


def test_pow_any_answers_non_positive_bases_as_pow_does(host_lib):
    """csrc/materials.hpp pow_any (round 5): the hardening laws no longer inline the library's pow for the bases a return-map
    equation never produces -- zero (exactly), negative, NaN.  Same answers as pow: the limits at zero (signed for odd
    integer exponents), +- |x|^q for an integer q (to the few ulp of pow_positive), NaN for a negative base with a
    non-integer exponent, 1 for q = 0 whatever the base; positive bases are pow_positive's to the bit."""
    x = np.array([0.0, 0.0, 0.0, -0.0, -0.0, 0.0, -2.0, -2.0, -2.0, -2.0, -1.5, -1e-3, np.nan, np.nan, -np.inf, 3.0, 1e-9])
    q = np.array([2.5, -0.7165, 0.0, -1.0, 3.0, -2.0, 2.0, 3.0, -1.0, 0.5, -0.7165, 4.0, 0.0, 1.5, 0.0, -0.7165, -0.7165])
    out, pos = np.empty(x.size), np.empty(x.size)
    host_lib.host_pow_any(x.size, x.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    with np.errstate(all="ignore"):
        ref = np.power(x, q)
    assert np.array_equal(np.isnan(out), np.isnan(ref)), (out, ref)
    ok = ~np.isnan(ref)
    exact = ok & (np.isinf(ref) | (ref == 0.0) | (q == 0.0))
    assert np.array_equal(out[exact], ref[exact]) and np.array_equal(np.signbit(out[exact]), np.signbit(ref[exact]))
    rest = ok & ~exact
    assert np.all(np.abs(out[rest] - ref[rest]) <= 4e-15 * np.abs(ref[rest]))
    xp = np.abs(x[-2:]); qp = q[-2:]
    host_lib.host_pow_positive(2, xp.ctypes.data_as(C.c_void_p), qp.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out[-2:], pos[:2])
