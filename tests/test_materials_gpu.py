"""The reference's other materials (StVenantKirchhoff, J2Linear, J2Simo, J2Log) through the C ABI against
the oracle: residual <= 1e-12; tangent (dual-number consistent tangent on the device) against the oracle's point-level
sixth-order difference-quotient tangent of the same stress <= 1e-10 (round 3: the oracle's quotient was second
order, good to ~1e-9, and the bar 1e-6; measured now <= 1.2e-11 point by point, tests/test_materials_host_cpu.py); state after DomainPostTimeAdvance
<= 1e-10; and again from the advanced state.  J2Simo / J2Log are pinned by the reference's golden series in
test_nonlinear_solid.py."""
import numpy as np
import pytest

from _cases import synthetic_u
from test_domain_gpu import make_pair, relmax

pytestmark = pytest.mark.gpu

MATS = ["stvk", "j2linear", "j2simo", "j2log"]
BLOCKS = [((2, 2), 3, [5.0, 1.0]), ((3, 2, 2), 2, None), ((2, 2, 1), 3, None)]


@pytest.mark.parametrize("block", BLOCKS, ids=lambda c: "x".join(map(str, c[0])) + f"p{c[1]}")
@pytest.mark.parametrize("matname", MATS)
def test_other_materials_parity(matname, block):
    from oracle import ref_path as rp
    n_el, p, lengths = block
    P, D, G = make_pair(n_el, p, lengths, matname, "bspline")
    # 3-D p = 2: the two-phase tensor kernels through the tangent record of the material pre-pass; else general kernels
    assert G.path_ == 1     # tensor kernels (3-D p = 2, 3 through the tangent record; 2-D: the small-element kernel)
    D.set_dt(0.5)
    G.dt_ = 0.5
    u = synthetic_u(P, scale=0.04)
    for round_ in range(2):
        r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
        r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
        D.add_domain_residual_and_grad(u, 0.7, r_o, A_o, rp.TANGENT_EXACT)
        G.AddDomainResidualAndGrad(u, 0.7, r_g, A_g)
        assert relmax(r_g, r_o) < 1e-12, (round_, relmax(r_g, r_o))
        assert relmax(A_g, A_o) < 1e-10, (round_, relmax(A_g, A_o))
        r2 = np.zeros(P.n_vdofs)
        G.AddDomainResidual(u, r2)
        assert relmax(r2, r_o) < 1e-12
        if matname == "stvk":
            break
        # commit the state at u, then look again from a further state
        D.domain_post_time_advance(u)
        G.DomainPostTimeAdvance(u)
        eq = G.State("accumulated_plastic_strain")
        assert eq.max() > 1e-3              # plasticity active
        assert np.abs(eq - D.eqps).max() < 1e-10
        m1 = G.State("plastic_strain")
        assert np.abs(m1 - D.plastic_strain).max() < 1e-10
        if matname in ("j2linear", "j2simo"):
            assert np.abs(G.State("state2") - D.state2).max() < 1e-10
        if matname != "j2linear":
            assert np.abs(G.State("temperature") - D.temperature).max() < 1e-9
        u = 1.25 * u


@pytest.mark.parametrize("matname", MATS)
def test_other_materials_medium_block(matname):
    """1536 elements, columns of 8, through the record path of the nine-block tensor kernel (every lane / carry /
    gather case), then the same handle family on the general kernels (MIMI_HIP_FORCE_GENERAL is read at create time)."""
    import os
    from oracle import ref_path as rp
    P, D, G = make_pair((16, 12, 8), 2, [4.0, 3.0, 2.0], matname, "bspline")
    assert G.path_ == 1
    D.set_dt(0.25)
    G.dt_ = 0.25
    u = synthetic_u(P, scale=0.01)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    os.environ["MIMI_HIP_FORCE_GENERAL"] = "1"
    try:
        _, _, G2 = make_pair((16, 12, 8), 2, [4.0, 3.0, 2.0], matname, "bspline")
    finally:
        del os.environ["MIMI_HIP_FORCE_GENERAL"]
    assert G2.path_ == 0
    G2.dt_ = 0.25
    for g in (G, G2):
        r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
        g.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
        assert relmax(r_g, r_o) < 1e-12
        assert relmax(A_g, A_o) < 1e-10
        r2 = np.zeros(P.n_vdofs)
        g.AddDomainResidual(u, r2)
        assert relmax(r2, r_o) < 1e-12
    if matname != "stvk":
        D.domain_post_time_advance(u)
        for g in (G, G2):
            g.DomainPostTimeAdvance(u)
            assert np.abs(g.State("accumulated_plastic_strain") - D.eqps).max() < 1e-10
            assert np.abs(g.State("plastic_strain") - D.plastic_strain).max() < 1e-10
        assert D.eqps.max() > 1e-3


@pytest.mark.parametrize("matname", ["j2simo", "j2log"])
def test_other_materials_reference_fd_mode(matname):
    """like for like with the reference: its element-level forward difference on the device against the restated one"""
    from oracle import ref_path as rp
    P, D, G = make_pair((2, 2), 3, [5.0, 1.0], matname, "bspline")
    D.set_dt(0.5)
    G.dt_ = 0.5
    G.SetTangentMode(1)
    u = synthetic_u(P, scale=0.04)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_FD)
    G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 5e-4
