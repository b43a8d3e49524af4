"""Parity of the sum-factorised two-phase kernels for 3-D p = 3 patches (mimi_amd/csrc/tensor_p3.hip; BASELINE
configuration 3 at oracle size) against the oracle, through the C ABI.  Bars as tests/test_domain_gpu.py:
residual <= 1e-12, analytic tangent vs the oracle's exact tangent <= 1e-11 (relative, max-norm), J2 state 1e-9."""
import numpy as np
import pytest

from _cases import synthetic_u
from test_domain_gpu import make_pair, product_material, relmax

pytestmark = pytest.mark.gpu

# (2,2,1): one element layer; (3,3,5) / (5,4,4): interior nodes with the full 4 x 4 x 4 element neighbourhood along one
# axis; lengths != counts: a non-unit (still affine) geometry map
CASES = [((2, 2, 1), None), ((3, 3, 5), None), ((5, 4, 4), [2.5, 3.0, 1.0]), ((1, 1, 1), None)]


@pytest.mark.parametrize("matname", ["neohook", "j2"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[0])))
def test_p3_residual_and_tangent_parity(case, matname):
    from oracle import ref_path as rp
    n_el, lengths = case
    P, D, G = make_pair(n_el, 3, lengths, matname, "bspline")
    assert G.path_ == 1          # the tensor kernels, not the general ones
    dt = 0.5
    D.set_dt(dt)
    G.dt_ = dt
    hmin = min((lengths[i] if lengths else n_el[i]) / n_el[i] for i in range(3))
    u = synthetic_u(P, scale=0.05 if matname == "neohook" else 0.02 * hmin)
    if matname == "j2":
        u0 = synthetic_u(P, scale=0.03 * hmin, seed=7)
        D.domain_post_time_advance(u0)
        G.DomainPostTimeAdvance(u0)
        assert D.eqps.max() > 1e-4
        assert np.allclose(G.State("accumulated_plastic_strain"), D.eqps, rtol=1e-9, atol=1e-13)
        assert np.allclose(G.State("temperature"), D.temperature, rtol=1e-12, atol=1e-12)
        assert np.allclose(G.State("plastic_strain"), D.plastic_strain, rtol=1e-9, atol=1e-13)
    r0 = np.random.default_rng(3).standard_normal(P.n_vdofs)
    r_o, r_g = r0.copy(), r0.copy()
    D.add_domain_residual(u, r_o)
    G.AddDomainResidual(u, r_g)
    assert relmax(r_g - r0, r_o - r0) < 1e-12
    gf = 0.37
    A0 = np.random.default_rng(4).standard_normal(D.nnz)
    r_o, r_g, A_o, A_g = r0.copy(), r0.copy(), A0.copy(), A0.copy()
    D.add_domain_residual_and_grad(u, gf, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, gf, r_g, A_g)
    assert relmax(r_g - r0, r_o - r0) < 1e-12
    assert relmax(A_g - A0, A_o - A0) < 1e-11
    # bitwise reproducible: no atomics anywhere on this path
    r_2, A_2 = r0.copy(), A0.copy()
    G.AddDomainResidualAndGrad(u, gf, r_2, A_2)
    assert np.array_equal(r_2, r_g) and np.array_equal(A_2, A_g)


@pytest.mark.parametrize("matname", ["neohook", "j2"])
@pytest.mark.parametrize("n_el", [(1, 1, 1), (2, 2, 1), (3, 2, 2), (2, 3, 5), (5, 4, 7)], ids=lambda c: "x".join(map(str, c)))
def test_p3_hand_scheduled_contraction_equals_the_compiler_scheduled_one_bitwise(n_el, matname, monkeypatch):
    """Round 5: the column loop of the contraction is one generated asm statement (tp3_contract_asm_kernel,
    csrc/gen_tp3_contract.py: fixed register map, memory / LDS / scalar instructions in the shadows of the matrix
    instructions, double-buffered accumulator tiles, stand-in stores for the first element).  Per value it performs the
    operations of the compiler-scheduled C++ form (tp3_contract_kernel, MIMI_HIP_P3_CONTRACT=cxx) in the same order, so
    every assembled value must be the same BITS -- on columns of 1, 2, 5 and 7 elements (the stand-in stores of element 0,
    the clamped prefetch of the last element, both tile-set parities at the column end).  Same reference lines as the
    kernels: integrators/nonlinear_solid.cpp:48-76 at n_dof 64."""
    P, D, G = make_pair(n_el, 3, None, matname, "bspline")
    assert G.path_ == 1
    G.dt_ = 0.5
    u = synthetic_u(P, scale=0.05 if matname == "neohook" else 0.02)
    out = {}
    for variant in ("cxx", "asm", "cxx", "asm"):
        monkeypatch.setenv("MIMI_HIP_P3_CONTRACT", variant)
        r, A = np.zeros(P.n_vdofs), np.full(D.nnz, 0.25)
        G.AddDomainResidualAndGrad(u, 0.37, r, A)
        assert np.abs(A - 0.25).max() > 0
        if variant in out:
            assert np.array_equal(out[variant][0], r) and np.array_equal(out[variant][1], A)
        out[variant] = (r, A)
    assert np.array_equal(out["asm"][0], out["cxx"][0])
    assert np.array_equal(out["asm"][1], out["cxx"][1])


@pytest.mark.parametrize("matname", ["stvk", "j2simo"])
def test_p3_other_materials(matname):
    """the materials without a closed-form tangent run through the same record (materials_other.hpp)"""
    from oracle import ref_path as rp
    n_el = (2, 3, 2)
    P, D, G = make_pair(n_el, 3, None, matname, "bspline")
    assert G.path_ == 1
    D.set_dt(0.5)
    G.dt_ = 0.5
    u = synthetic_u(P, scale=0.02)
    if matname != "stvk":
        u0 = synthetic_u(P, scale=0.03, seed=7)
        D.domain_post_time_advance(u0)
        G.DomainPostTimeAdvance(u0)
    r_o, r_g = np.zeros(P.n_vdofs), np.zeros(P.n_vdofs)
    A_o, A_g = np.zeros(D.nnz), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-10      # bar of tests/test_materials_gpu.py for these tangents


def test_p3_element_slabs_add_up():
    """two handles on complementary element slabs (what two GPUs would hold) give the whole-patch assembly"""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    n_el = (6, 3, 4)
    patch = mimi_amd.BSplinePatch.block(n_el, 3)
    pattern = CSRPattern.of_bspline_patch(patch, device=0)
    mat = product_material("neohook")
    whole = NonlinearSolid("whole", mat, pattern, patch=patch).Prepare()
    lo = NonlinearSolid("lo", mat, pattern, patch=patch, element_box=((0, 0, 0), (2, 3, 4))).Prepare()
    hi = NonlinearSolid("hi", mat, pattern, patch=patch, element_box=((2, 0, 0), (6, 3, 4))).Prepare()
    assert whole.path_ == 1 and lo.path_ == 1 and hi.path_ == 1
    u = 0.05 * np.random.default_rng(11).standard_normal(patch.n_vdofs)
    r_w, A_w = np.zeros(patch.n_vdofs), np.zeros(pattern.nnz)
    r_s, A_s = np.zeros(patch.n_vdofs), np.zeros(pattern.nnz)
    whole.AddDomainResidualAndGrad(u, 1.0, r_w, A_w)
    lo.AddDomainResidualAndGrad(u, 1.0, r_s, A_s)
    hi.AddDomainResidualAndGrad(u, 1.0, r_s, A_s)
    assert relmax(r_s, r_w) < 1e-13
    assert relmax(A_s, A_w) < 1e-13


@pytest.mark.parametrize("matname", ["neohook", "stvk"])
def test_p3_frame_indifference(matname):
    """Objectivity of the hyperelastic laws, P(Q F) = Q P(F), through the degree-3 kernels with no oracle in the loop: the
    assembled tangent maps an infinitesimal rigid rotation of x = X + u to the rotated residual, K (omega x x) = omega x r,
    and a rigid translation to zero (16 x 12 x 9 elements: interior nodes with the full 4 x 4 x 4 neighbourhood, several
    columns per direction, the z-carry over nine elements)."""
    import scipy.sparse as sp
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    n_el = (16, 12, 9)
    patch = mimi_amd.BSplinePatch.block(n_el, 3)
    pattern = CSRPattern.of_bspline_patch(patch)
    G = NonlinearSolid("domain", product_material(matname), pattern, patch=patch).Prepare()
    assert G.path_ == 1
    u = 0.05 * np.random.default_rng(20241008).standard_normal(patch.n_vdofs)
    r, A = np.zeros(patch.n_vdofs), np.zeros(pattern.nnz)
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    M = sp.csr_matrix((A, np.asarray(pattern.col), np.asarray(pattern.rowptr)), shape=(patch.n_vdofs, patch.n_vdofs))
    x = patch.control_points + u.reshape(-1, 3)
    rr = r.reshape(-1, 3)
    scale = np.abs(A).max() * np.abs(x).max() * 100
    assert np.abs(rr).max() > 0
    for k in range(3):
        t = np.zeros((patch.n_nodes, 3))
        t[:, k] = 1.0
        assert np.abs(M @ t.ravel()).max() < 1e-11 * scale
        omega = np.eye(3)[k]
        y = M @ np.cross(omega, x).ravel()
        assert np.abs(y - np.cross(omega, rr).ravel()).max() < 1e-11 * scale
        assert np.abs(y).max() > 1e-3 * np.abs(rr).max()


def test_p3_committed_plastic_strain_is_symmetric_to_the_bit():
    """What T3Park (csrc/tensor_p3.hip) relies on: every J2 mode of the degree-3 pre-pass hands F^-1 and the trial deviator
    to LDS over the return mapping and keeps six of the deviator's nine entries -- exact as long as
    s = 2G dev(0.5 (F + F^T) - I - eps_p) is symmetric TO THE BIT, i.e. as long as the plastic strain is: it starts at zero
    and grows by delta n with n built from s.  Two committed steps (the second one starts from a non-trivial eps_p) leave a
    plastic strain that equals its transpose bitwise at every point, and agrees with the oracle's (which parks nothing)."""
    n_el = (3, 3, 4)
    P, D, G = make_pair(n_el, 3, None, "j2", "bspline")
    D.set_dt(0.5)
    G.dt_ = 0.5
    for seed, scale in ((7, 0.03), (8, 0.045)):
        u0 = synthetic_u(P, scale=scale, seed=seed)
        D.domain_post_time_advance(u0)
        G.DomainPostTimeAdvance(u0)
        ep = G.State("plastic_strain").reshape(-1, 3, 3)
        assert np.abs(ep).max() > 1e-4
        assert np.array_equal(ep, np.swapaxes(ep, 1, 2))
        assert np.allclose(G.State("plastic_strain"), D.plastic_strain, rtol=1e-9, atol=1e-13)
    # and the assemblies from that state (parked in all three modes) against the oracle
    u = synthetic_u(P, scale=0.02)
    r_o, r_g = np.zeros(P.n_vdofs), np.zeros(P.n_vdofs)
    D.add_domain_residual(u, r_o)
    G.AddDomainResidual(u, r_g)
    assert relmax(r_g, r_o) < 1e-12
