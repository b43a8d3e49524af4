"""Parity of the HIP domain integrator (through the C ABI) against the oracle on the same
seeded inputs.  Bars: residual <= 1e-12 relative (max-norm); analytic tangent
vs the oracle's exact tangent <= 1e-11 relative; reference-FD mode vs the oracle's
reference-FD restatement within the FD round-off amplification (5e-4 relative, measured 1e-5)."""
import os

import numpy as np
import pytest

from _cases import JC_TEST, oracle_material, synthetic_u

pytestmark = pytest.mark.gpu

# (5,6,4) and (4,4,5): columns of 4 / 5 elements along the walked (third) axis, so that the two-step carry
# (2,2) -> (1,1) -> (0,0) of the two-phase kernels is exercised; (4,5,6): the walked (third) axis is the longest one
CASES = [((2, 2), 3, [5.0, 1.0]), ((3, 4), 2, None), ((3, 2, 2), 2, None), ((2, 2, 1), 3, None),
         ((4, 3, 2), 1, None), ((8, 8, 2), 2, None), ((5, 6, 4), 2, [2.5, 3.0, 1.0]), ((5, 5, 5), 2, None),
         ((4, 5, 6), 2, None)]


def product_material(name):
    import mimi_amd
    if name == "neohook":
        m = mimi_amd.CompressibleOgdenNeoHookean()
        m.density = 1.0
        m.set_young_poisson(2100, 0.3)
        return m
    if name == "stvk":
        m = mimi_amd.StVenantKirchhoff()
        m.density = 1.0
        m.set_young_poisson(2100, 0.3)
        return m
    if name == "j2linear":
        m = mimi_amd.J2Linear()
        m.density = 1.0
        m.set_young_poisson(2100, 0.3)
        m.isotropic_hardening, m.kinematic_hardening, m.sigma_y = 40.0, 25.0, 70.0
        return m
    m = {"j2": mimi_amd.J2, "j2simo": mimi_amd.J2Simo, "j2log": mimi_amd.J2Log}[name]()
    m.density = 1.0
    m.set_young_poisson(2100, 0.3)
    m.heat_fraction, m.specific_heat = 0.9, 450
    m.initial_temperature, m.melting_temperature = 20, 1500
    h = mimi_amd.JohnsonCookTemperatureAndRateDependentHardening()
    for k, v in JC_TEST.items():
        if k != "kind":
            setattr(h, k, v)
    m.hardening = h
    return m


def make_pair(n_el, p, lengths, matname, creator):
    """(oracle integrator, product integrator) on the same patch."""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga, ref_path as rp
    P = iga.Patch.block(n_el, p, lengths)
    D = rp.DomainOracle(P, oracle_material(matname), n_threads=2)
    pattern = CSRPattern(D.rowptr.astype(np.int64), D.col.astype(np.int32), D.nnz)
    if creator == "tables":
        tables = dict(dim=P.dim, n_nodes=P.n_nodes, dofs=D.conn, dN_dX=D.dN_dX, weight_det=D.weight * D.det)
        G = NonlinearSolid("domain", product_material(matname), pattern, tables=tables).Prepare()
    else:
        patch = mimi_amd.BSplinePatch.block(n_el, p, lengths)
        G = NonlinearSolid("domain", product_material(matname), pattern, patch=patch).Prepare()
    return P, D, G


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("creator", ["tables", "bspline"])
@pytest.mark.parametrize("matname", ["neohook", "j2"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[0])) + f"p{c[1]}")
def test_residual_and_tangent_parity(case, matname, creator):
    from oracle import ref_path as rp
    n_el, p, lengths = case
    P, D, G = make_pair(n_el, p, lengths, matname, creator)
    # B-spline patches (2-D and 3-D, degree <= 3) must take the sum-factorised (tensor) kernels, flat tables the general ones
    # (MIMI_HIP_NO_STRUCTURED=1, test_fallback_kernel_families: a 3-D degree-2 / 3 patch whose CSR is not recognised as the
    # structured pattern takes the general kernels)
    unrecognised = os.environ.get("MIMI_HIP_NO_STRUCTURED") == "1" and len(n_el) == 3 and p >= 2
    assert G.path_ == (1 if creator == "bspline" and not unrecognised else 0)
    dt = 0.5
    D.set_dt(dt)
    G.dt_ = dt
    hmin = min((lengths[i] if lengths else n_el[i]) / n_el[i] for i in range(len(n_el)))
    scale = 0.05 if matname == "neohook" else 0.02 * hmin
    u = synthetic_u(P, scale=scale)
    if matname == "j2":
        # commit a plastic state first so that eps_p, eqps, T are non-trivial (20-77 % of the
        # points yield with these amplitudes)
        u0 = synthetic_u(P, scale=0.03 * hmin, seed=7)
        D.domain_post_time_advance(u0)
        G.DomainPostTimeAdvance(u0)
        assert D.eqps.max() > 1e-4
        assert np.allclose(G.State("accumulated_plastic_strain"), D.eqps, rtol=1e-9, atol=1e-13)
        assert np.allclose(G.State("temperature"), D.temperature, rtol=1e-12, atol=1e-12)
        assert np.allclose(G.State("plastic_strain"), D.plastic_strain, rtol=1e-9, atol=1e-13)
    # residual only, accumulate semantics (+=)
    r0 = np.random.default_rng(3).standard_normal(P.n_vdofs)
    r_o, r_g = r0.copy(), r0.copy()
    D.add_domain_residual(u, r_o)
    G.AddDomainResidual(u, r_g)
    assert relmax(r_g - r0, r_o - r0) < 1e-12
    # residual + analytic tangent
    gf = 0.37
    A0 = np.random.default_rng(4).standard_normal(D.nnz)
    r_o, r_g, A_o, A_g = r0.copy(), r0.copy(), A0.copy(), A0.copy()
    D.add_domain_residual_and_grad(u, gf, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, gf, r_g, A_g)
    assert relmax(r_g - r0, r_o - r0) < 1e-12
    assert relmax(A_g - A0, A_o - A0) < 1e-11
    # reference finite-difference mode, like for like
    if P.n_el <= 32:
        G.SetTangentMode(1)
        r_o, r_g, A_o, A_g = r0.copy(), r0.copy(), A0.copy(), A0.copy()
        D.add_domain_residual_and_grad(u, gf, r_o, A_o, rp.TANGENT_FD)
        G.AddDomainResidualAndGrad(u, gf, r_g, A_g)
        assert relmax(r_g - r0, r_o - r0) < 1e-12
        # both sides are forward differences with steps down to 1e-10: the round-off of the two
        # residual evaluations (1e-16 |R|) is amplified by 1/h, so they agree to ~1e-5..1e-4
        assert relmax(A_g - A0, A_o - A0) < 5e-4
        G.SetTangentMode(0)


@pytest.mark.parametrize("matname", ["j2", "j2simo", "j2log"])
@pytest.mark.parametrize("case", [((3, 4), 2, None, "bspline"), ((3, 2, 2), 2, None, "bspline"), ((2, 3, 2), 3, None, "bspline"),
                                  ((3, 2, 2), 2, None, "tables")], ids=lambda c: "x".join(map(str, c[0])) + f"p{c[1]}-{c[3]}")
def test_rate_dependent_johnson_cook_parity(case, matname, monkeypatch):
    """The RATE term of the Johnson-Cook law, 1 + C ln(rate / eps0_dot) (material_hardening.hpp:261-279): the reference's own
    tests never set C, so with their parameters the factor is 1 and the logarithm inside every trip of the
    return-map Newton -- and its derivative C / rate in the hardening slope -- never runs.  Here C = 0.04 and a time step
    for which most yielding points exceed the reference rate: committed state, residual, residual + analytic tangent
    against the oracle, on the tensor kernels of degree 2 and 3 (2-D, 3-D) and the general kernels, for the three J2 models
    that take a hardening law."""
    import _cases as cases
    from oracle import ref_path as rp
    monkeypatch.setitem(cases.JC_TEST, "C", 0.04)
    n_el, p, lengths, creator = case
    P, D, G = make_pair(n_el, p, lengths, matname, creator)
    dt = 0.05
    D.set_dt(dt)
    G.dt_ = dt
    u0 = synthetic_u(P, scale=0.03, seed=7)
    D.domain_post_time_advance(u0)
    G.DomainPostTimeAdvance(u0)
    assert D.eqps.max() / dt > 10 * cases.JC_TEST["eps0_dot"]            # (the rate term is active)
    assert np.allclose(G.State("accumulated_plastic_strain"), D.eqps, rtol=1e-9, atol=1e-13)
    assert np.allclose(G.State("temperature"), D.temperature, rtol=1e-12, atol=1e-12)
    u = synthetic_u(P, scale=0.02)
    r_o, r_g = np.zeros(P.n_vdofs), np.zeros(P.n_vdofs)
    D.add_domain_residual(u, r_o)
    G.AddDomainResidual(u, r_g)
    assert relmax(r_g, r_o) < 1e-12
    r_o, r_g, A_o, A_g = np.zeros(P.n_vdofs), np.zeros(P.n_vdofs), np.zeros(D.nnz), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 0.37, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, 0.37, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11
    # and the factor is not 1: the same assembly with C = 0 differs
    monkeypatch.setitem(cases.JC_TEST, "C", 0.0)
    P2, D2, G2 = make_pair(n_el, p, lengths, matname, creator)
    D2.set_dt(dt)
    D2.domain_post_time_advance(u0)
    r_c0 = np.zeros(P.n_vdofs)
    D2.add_domain_residual(u, r_c0)
    assert relmax(r_c0, r_o) > 1e-4


def test_device_pointers_and_stream():
    """u / r / A as torch tensors on the GPU: used in place, asynchronous on the given stream."""
    import torch
    from oracle import ref_path as rp
    P, D, G = make_pair((3, 2, 2), 2, None, "neohook", "bspline")
    u = synthetic_u(P)
    dev = torch.device("cuda", 0)
    tu = torch.from_numpy(u).to(dev)
    tr = torch.zeros(P.n_vdofs, dtype=torch.float64, device=dev)
    tA = torch.zeros(D.nnz, dtype=torch.float64, device=dev)
    G.SetStream(torch.cuda.current_stream().cuda_stream)
    G.AddDomainResidualAndGrad(tu, 1.0, tr, tA)
    G.AddDomainResidual(tu, tr)
    G.Synchronize()
    r_o = np.zeros(P.n_vdofs)
    A_o = np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    assert relmax(tr.cpu().numpy(), 2 * r_o) < 1e-12
    assert relmax(tA.cpu().numpy(), A_o) < 1e-11


def test_structured_sparsity_matches_oracle():
    import mimi_amd
    from mimi_amd.integrators import CSRPattern
    from oracle import iga
    for n_el, p in [((3, 4), 2), ((2, 2), 3), ((3, 2, 2), 2), ((2, 3, 2), 3), ((5, 4, 3), 1)]:
        P = iga.Patch.block(n_el, p)
        rowptr, col = P.sparsity()
        pat = CSRPattern.of_bspline_patch(mimi_amd.BSplinePatch.block(n_el, p))
        assert pat.nnz == rowptr[-1]
        assert np.array_equal(pat.rowptr, rowptr)
        assert np.array_equal(pat.col, col)


def test_errors_surface_as_runtime_error():
    """ScalarSolve's throw paths / bad input arrive as RuntimeError (utils/print.hpp:47-56)."""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    patch = mimi_amd.BSplinePatch.block((2, 2), 2)
    pat = CSRPattern.of_bspline_patch(patch)
    m = mimi_amd.J2()
    m.set_young_poisson(2100, 0.3)
    with pytest.raises(RuntimeError, match="hardening missing"):
        NonlinearSolid("domain", m, pat, patch=patch).Prepare()
    bad = CSRPattern(pat.rowptr, np.zeros_like(pat.col), pat.nnz)
    with pytest.raises(RuntimeError, match="CSR pattern"):
        NonlinearSolid("domain", product_material("neohook"), bad, patch=patch).Prepare()


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("matname", ["neohook", "j2"])
def test_element_boxes_add_up_to_the_whole(axis, matname):
    """Shards (element boxes, as the multi-GPU path uses them) along every axis, including the walked one:
    the sum of the boxes' assemblies equals the oracle's assembly of the whole patch."""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga, ref_path as rp
    n_el = (5, 4, 6)
    P = iga.Patch.block(n_el, 2)
    D = rp.DomainOracle(P, oracle_material(matname), n_threads=2)
    D.set_dt(0.5)
    pattern = CSRPattern(D.rowptr.astype(np.int64), D.col.astype(np.int32), D.nnz)
    patch = mimi_amd.BSplinePatch.block(n_el, 2)
    u = synthetic_u(P, scale=0.05 if matname == "neohook" else 0.02)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    r_g, A_g, r_only = np.zeros(P.n_vdofs), np.zeros(D.nnz), np.zeros(P.n_vdofs)
    cuts = [0, 2, n_el[axis]] if axis != 2 else [0, 2, 3, n_el[axis]]   # a one-element slab along the walked axis too
    for b, e in zip(cuts[:-1], cuts[1:]):
        begin, end = [0, 0, 0], list(n_el)
        begin[axis], end[axis] = b, e
        G = NonlinearSolid("domain", product_material(matname), pattern, patch=patch, element_box=(begin, end)).Prepare()
        assert G.path_ == (0 if os.environ.get("MIMI_HIP_NO_STRUCTURED") == "1" else 1)     # (see test_fallback_kernel_families)
        G.dt_ = 0.5
        G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
        # the residual-only assembly of the same box (neo-Hookean: one wave per element column of the BOX, the columns cut
        # where the box ends along the walked axis; J2: one wave per element)
        G.AddDomainResidual(u, r_only)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11
    assert relmax(r_only, r_o) < 1e-12


def test_element_boxes_p3_general_path(monkeypatch):
    """element boxes on the general path with 64-node elements (store + gather assembly: nodes no element of a box
    touches are skipped by the gather kernel); the boxes add up to the oracle's whole-patch assembly.  (Structured p = 3
    patches take the tensor kernels since round 2: MIMI_HIP_FORCE_GENERAL keeps this path covered.)"""
    import mimi_amd
    monkeypatch.setenv("MIMI_HIP_FORCE_GENERAL", "1")
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga, ref_path as rp
    n_el = (3, 5, 2)
    P = iga.Patch.block(n_el, 3)
    D = rp.DomainOracle(P, oracle_material("neohook"), n_threads=2)
    pattern = CSRPattern(D.rowptr.astype(np.int64), D.col.astype(np.int32), D.nnz)
    patch = mimi_amd.BSplinePatch.block(n_el, 3)
    u = synthetic_u(P, scale=0.04)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    for b, e in ((0, 1), (1, 5)):
        G = NonlinearSolid("domain", product_material("neohook"), pattern, patch=patch,
                           element_box=([0, b, 0], [3, e, 2])).Prepare()
        assert G.path_ == 0
        G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11


@pytest.mark.parametrize("matname", ["neohook", "j2"])
@pytest.mark.parametrize("p", [2, 3])
def test_permuted_node_numbering(matname, p):
    """node_ids = lexicographic -> caller's node id (what MFEM's NURBS dof map is for the reference): u, r and
    the CSR live in the caller's numbering; the two-phase kernels (degree 2 and degree 3) must still be taken (info 6 == 2)."""
    import scipy.sparse as sp
    import mimi_amd
    from mimi_amd import _capi
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga, ref_path as rp
    n_el = (5, 4, 4) if p == 2 else (4, 3, 5)
    P = iga.Patch.block(n_el, p)
    D = rp.DomainOracle(P, oracle_material(matname), n_threads=2)
    D.set_dt(0.5)
    u = synthetic_u(P, scale=0.05 if matname == "neohook" else 0.02)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 0.5, r_o, A_o, rp.TANGENT_EXACT)
    perm = np.random.default_rng(11).permutation(P.n_nodes).astype(np.int64)
    dofperm = (perm[:, None] * 3 + np.arange(3)[None, :]).ravel()          # lexicographic dof -> caller's dof
    rows_o = np.repeat(np.arange(P.n_vdofs), np.diff(D.rowptr))
    # the caller's CSR: same matrix, rows / columns renumbered, columns sorted; slot k of the oracle lands at dst[k]
    S = sp.coo_matrix((np.arange(1, D.nnz + 1, dtype=np.float64), (dofperm[rows_o], dofperm[D.col])),
                      shape=(P.n_vdofs, P.n_vdofs)).tocsr()
    S.sort_indices()
    dst = np.empty(D.nnz, dtype=np.int64)
    dst[(S.data - 1).astype(np.int64)] = np.arange(D.nnz)
    pattern = CSRPattern(S.indptr.astype(np.int64), S.indices.astype(np.int32), D.nnz)
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    G = NonlinearSolid("domain", product_material(matname), pattern, patch=patch, node_ids=perm).Prepare()
    assert G.path_ == 1
    assert _capi.lib().mimi_hip_domain_info(G._h, 6) == 2
    G.dt_ = 0.5
    u_p = np.empty_like(u)
    u_p[dofperm] = u
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    G.AddDomainResidualAndGrad(u_p, 0.5, r_g, A_g)
    assert relmax(r_g[dofperm], r_o) < 1e-12
    assert relmax(A_g[dst], A_o) < 1e-11
    # residual-only call
    r_g[:] = 0.0
    G.AddDomainResidual(u_p, r_g)
    assert relmax(r_g[dofperm], r_o) < 1e-12


@pytest.mark.parametrize("matname", ["neohook", "j2"])
def test_medium_block_parity(matname):
    """1536 elements, columns of 8: every lane / carry / gather case of the two-phase kernels, against the oracle."""
    P, D, G = make_pair((16, 12, 8), 2, [4.0, 3.0, 2.0], matname, "bspline")
    from oracle import ref_path as rp
    D.set_dt(0.25)
    G.dt_ = 0.25
    u = synthetic_u(P, scale=0.05 if matname == "neohook" else 0.005)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11


@pytest.mark.parametrize("n_el", [(3, 4, 16), (5, 3, 12), (2, 2, 24), (7, 1, 8)], ids=lambda n: "x".join(map(str, n)))
def test_column_segments(n_el):
    """Few, long element columns: the symmetric kernel cuts them into segments with their own workgroup (4 x 4, 2 x 6,
    4 x 6, 2 x 4 elements here); the carried rows at every segment end, against the oracle; and a sub-box of the same."""
    from oracle import ref_path as rp
    P, D, G = make_pair(n_el, 2, None, "neohook", "bspline")
    u = synthetic_u(P, scale=0.04)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11


def test_many_columns_per_workgroup():
    """65 x 65 columns of 2 elements: the symmetric kernel walks two columns per workgroup here (4225 columns, so the
    last workgroup has only one) -- the column boundary inside a workgroup (carry stored and reset), against the oracle."""
    P, D, G = make_pair((65, 65, 2), 2, [4.0, 4.0, 1.0], "neohook", "bspline")
    from oracle import ref_path as rp
    u = synthetic_u(P, scale=0.005)   # elements are 0.06 wide: larger random displacements nearly invert some of them
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11


@pytest.mark.parametrize("n_el", [(1, 1, 1), (2, 1, 3), (1, 3, 1), (1, 1, 4), (3, 1, 2)], ids=lambda n: "x".join(map(str, n)))
def test_tiny_blocks(n_el):
    """Degenerate sizes of the two-phase kernels: single columns, single elements per column, windows cut on
    both sides (fewer than 2p+1 nodes per direction)."""
    P, D, G = make_pair(n_el, 2, None, "neohook", "bspline")
    from oracle import ref_path as rp
    assert G.path_ == 1
    u = synthetic_u(P, scale=0.05)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11
    r_g[:] = 0.0
    G.AddDomainResidual(u, r_g)
    assert relmax(r_g, r_o) < 1e-12


@pytest.mark.parametrize("n_el,p", [((3, 4), 2), ((3, 2, 2), 2), ((2, 2), 3)], ids=["3x4p2", "3x2x2p2", "2x2p3"])
def test_rational_weights_general_path(n_el, p):
    """True NURBS (non-unit weights, curved geometry): the reference's flattened per-point tables go through the
    general kernels (mimi_hip_domain_create); parity against the oracle's rational basis."""
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga, ref_path as rp
    P0 = iga.Patch.block(n_el, p)
    rng = np.random.default_rng(21)
    weights = 1.0 + 0.3 * rng.uniform(-1, 1, P0.n_nodes)
    ctrl = np.asarray(P0.ctrl, dtype=np.float64).reshape(P0.n_nodes, -1) + 0.05 * rng.standard_normal((P0.n_nodes, len(n_el)))
    P = iga.Patch(P0.p, P0.knots, ctrl, weights)
    D = rp.DomainOracle(P, oracle_material("neohook"), n_threads=2)
    pattern = CSRPattern(D.rowptr.astype(np.int64), D.col.astype(np.int32), D.nnz)
    tables = dict(dim=P.dim, n_nodes=P.n_nodes, dofs=D.conn, dN_dX=D.dN_dX, weight_det=D.weight * D.det)
    G = NonlinearSolid("domain", product_material("neohook"), pattern, tables=tables).Prepare()
    assert G.path_ == 0
    u = synthetic_u(P, scale=0.03)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11


@pytest.mark.parametrize("matname", ["neohook", "j2"])
@pytest.mark.parametrize("n_el,p", [((4, 3, 3), 2), ((3, 4), 2), ((2, 2), 3), ((3, 2, 5), 3)], ids=["4x3x3p2", "3x4p2", "2x2p3", "3x2x5p3"])
def test_tensor_product_nurbs_weights(n_el, p, matname):
    """True NURBS whose weights are a tensor product of 1-D weights (arcs, cylinders, extrusions ...): the rational
    basis factorises, so the B-spline creator takes them and the 3-D p = 2 / p = 3 cases run on the two-phase tensor kernels.
    Curved geometry (perturbed control net); parity against the oracle's rational basis."""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga, ref_path as rp
    P0 = iga.Patch.block(n_el, p)
    rng = np.random.default_rng(5)
    w1d = [1.0 + 0.4 * rng.uniform(-1, 1, n) for n in P0.n]
    w = w1d[0]
    for d in range(1, len(n_el)):
        w = np.multiply.outer(w1d[d], w)            # first direction fastest
    weights = 0.7 * w.ravel()
    ctrl = np.asarray(P0.ctrl, dtype=np.float64).reshape(P0.n_nodes, -1) + 0.05 * rng.standard_normal((P0.n_nodes, len(n_el)))
    P = iga.Patch(P0.p, P0.knots, ctrl, weights)
    D = rp.DomainOracle(P, oracle_material(matname), n_threads=2)
    D.set_dt(0.5)
    pattern = CSRPattern(D.rowptr.astype(np.int64), D.col.astype(np.int32), D.nnz)
    patch = mimi_amd.BSplinePatch(P.p, P.knots, ctrl, weights)
    G = NonlinearSolid("domain", product_material(matname), pattern, patch=patch).Prepare()
    G.dt_ = 0.5
    assert G.path_ == 1
    u = synthetic_u(P, scale=0.03)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert relmax(r_g, r_o) < 1e-12
    assert relmax(A_g, A_o) < 1e-11
    r_g[:] = 0.0
    G.AddDomainResidual(u, r_g)
    assert relmax(r_g, r_o) < 1e-12
    # weights that do not factorise are refused on this route (they go through mimi_hip_domain_create as flat tables)
    bad = weights.copy()
    bad[P.n_nodes // 2] *= 1.01
    with pytest.raises(RuntimeError, match="tensor product"):
        NonlinearSolid("domain", product_material(matname), pattern,
                       patch=mimi_amd.BSplinePatch(P.p, P.knots, ctrl, bad)).Prepare()


@pytest.mark.parametrize("env", [{"MIMI_HIP_TENSOR_VARIANT": "wgs"}, {"MIMI_HIP_NO_STRUCTURED": "1"}],
                         ids=["nine-block", "csr-not-structured"])
def test_fallback_kernel_families(env):
    """The kernels behind the default route (selected by environment variables the library reads once per process): the
    nine-block workgroup kernel for a hyperelastic material, and what a degree-2 patch runs on when its CSR is not
    recognised as the structured pattern -- since round 5 the general kernels (store + row gather through the
    pair-position tables; the colour-partitioned kernel of round 1, 938 spilled registers, is gone)."""
    import os
    import subprocess
    import sys
    e = dict(os.environ, **env)
    select = "(5x5x5p2 and bspline) or (boxes and neohook)"
    if "MIMI_HIP_NO_STRUCTURED" not in env:
        # (these assert the structured-pattern detection / the two-phase kernels, which that variable takes away)
        select += " or tiny or permuted"
    cmd = [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
           "-k", select]
    res = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]


def test_default_stream_follows_torch():
    """No SetStream: with CUDA tensors the handle launches on torch's CURRENT stream, so torch's zero fills before the
    call and torch's reads after it are ordered with the kernels without any explicit synchronisation."""
    import torch
    from oracle import ref_path as rp
    P, D, G = make_pair((6, 5, 4), 2, None, "neohook", "bspline")
    u = synthetic_u(P)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    tu = torch.from_numpy(u).to(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        tr = torch.full((P.n_vdofs,), 7.0, dtype=torch.float64, device=dev)
        tA = torch.full((D.nnz,), 7.0, dtype=torch.float64, device=dev)
        tr.zero_()
        tA.zero_()
        G.AddDomainResidualAndGrad(tu, 1.0, tr, tA)
        r2 = tr * 2.0                      # torch work behind the kernels, same stream, no Synchronize() in between
        A2 = tA * 2.0
    side.synchronize()
    r_o = np.zeros(P.n_vdofs)
    A_o = np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    assert relmax(r2.cpu().numpy(), 2 * r_o) < 1e-12
    assert relmax(A2.cpu().numpy(), 2 * A_o) < 1e-11
    with pytest.raises(TypeError):         # a float32 buffer is refused, not reinterpreted
        G.AddDomainResidual(tu.float(), tr)


@pytest.mark.parametrize("path", ["general", "tensor"])
@pytest.mark.parametrize("case", [((40, 30), 2), ((24, 20), 3), ((12, 10, 8), 1), ((9, 7), 1)], ids=["2d-p2", "2d-p3", "3d-p1", "2d-p1"])
def test_small_elements_are_bitwise_reproducible(case, path, monkeypatch):
    """2-D meshes and degree 1 (what the reference's own examples are): on the general kernels (flat tables) and on the
    small-element tensor kernel (1-D tables, sum factorisation) alike the element blocks are stored and gathered per CSR
    row: no atomics, so two assemblies of the same input agree bit for bit -- in both tangent modes and for the residual-only
    call -- and match the oracle"""
    from oracle import ref_path as rp
    n_el, p = case
    if path == "general":
        monkeypatch.setenv("MIMI_HIP_FORCE_GENERAL", "1")
    P, D, G = make_pair(n_el, p, None, "neohook", "bspline")
    assert G.path_ == (0 if path == "general" else 1)
    u = synthetic_u(P)
    outs = []
    for rep in range(3):
        r, A = np.zeros(P.n_vdofs), np.zeros(D.nnz)
        G.AddDomainResidualAndGrad(u, 0.9, r, A)
        r2 = np.zeros(P.n_vdofs)
        G.AddDomainResidual(u, r2)
        outs.append((r, A, r2))
    for r, A, r2 in outs[1:]:
        assert np.array_equal(r, outs[0][0]) and np.array_equal(A, outs[0][1]) and np.array_equal(r2, outs[0][2])
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 0.9, r_o, A_o, rp.TANGENT_EXACT)
    assert relmax(outs[0][0], r_o) < 1e-12 and relmax(outs[0][1], A_o) < 1e-11 and relmax(outs[0][2], r_o) < 1e-12
    G.SetTangentMode(1)
    fd = []
    for rep in range(2):
        r, A = np.zeros(P.n_vdofs), np.zeros(D.nnz)
        G.AddDomainResidualAndGrad(u, 0.9, r, A)
        fd.append(A)
    assert np.array_equal(fd[0], fd[1])


@pytest.mark.parametrize("case", [((5, 7, 4), 2, "neohook", None), ((4, 6, 5), 2, "j2", None), ((3, 5, 4), 3, "j2", None),
                                  ((6, 8, 5), 2, "neohook", ([1, 2, 0], [5, 7, 5]))],
                         ids=["p2-symmetric", "p2-nine-block", "p3", "p2-element-box"])
def test_integrate_then_gather_in_parts_equals_one_call(case):
    """the two-step form: Integrate() once, Gather() over node windows that partition the handle's nodes -- bitwise the
    one-call assembly (same kernels, same order of additions per row); a window outside the handle's nodes is refused"""
    import torch
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    n_el, p, matname, box = case
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    dev = torch.device("cuda", 0)
    pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
    G = NonlinearSolid("domain", product_material(matname), pattern, patch=patch, element_box=box).Prepare()
    G.dt_ = 0.5
    from oracle import iga
    u = torch.from_numpy(synthetic_u(iga.Patch.block(n_el, p), scale=0.05 if matname == "neohook" else 0.02)).to(dev)
    r1 = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
    A1 = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    G.AddDomainResidualAndGrad(u, 0.7, r1, A1)
    G.Synchronize()
    eb, ee = box if box else ([0, 0, 0], list(n_el))
    lo, hi = list(eb), [ee[d] + p for d in range(3)]                    # the nodes the handle's elements touch
    r2, A2 = torch.zeros_like(r1), torch.zeros_like(A1)
    with pytest.raises(RuntimeError):
        G.Gather(0.7, r2, A2, lo, hi)                                   # nothing integrated yet
    G.Integrate(u)
    cuts = [lo[1], lo[1] + 1, lo[1] + 1 + (hi[1] - lo[1]) // 2, hi[1]]   # three windows along y: one plane, a half, the rest
    for y0, y1 in zip(cuts[:-1], cuts[1:]):
        G.Gather(0.7, r2, A2, [lo[0], y0, lo[2]], [hi[0], y1, hi[2]])
    G.Synchronize()
    assert float(A1.abs().max()) > 0 and torch.equal(r1, r2) and torch.equal(A1, A2)
    with pytest.raises(RuntimeError):
        G.Gather(0.7, r2, A2, [lo[0], lo[1] - 1, lo[2]], hi)            # a window outside the handle's nodes
    # a handle on the general kernels has no two-step form
    P2d = mimi_amd.BSplinePatch.block((4, 3), 2)
    pat2 = CSRPattern.of_bspline_patch(P2d, on_device=True)
    g2 = NonlinearSolid("domain", product_material("neohook"), pat2, patch=P2d).Prepare()
    with pytest.raises(RuntimeError):
        g2.Integrate(torch.zeros(P2d.n_vdofs, dtype=torch.float64, device=dev))


# (shape, degree, lengths, material, creator): one case per kernel family that assembles a tangent -- two-phase tensor
# degree 2 (symmetric-half and nine-block), degree 3, the small-element tensor kernel, the general kernels
FROM_CASES = [((4, 3, 5), 2, None, "neohook", "bspline"), ((3, 3, 4), 2, None, "j2", "bspline"),
              ((3, 2, 3), 3, None, "neohook", "bspline"), ((3, 4), 2, None, "neohook", "bspline"),
              ((2, 2), 3, [5.0, 1.0], "j2", "tables"), ((3, 2, 2), 2, None, "neohook", "tables")]


@pytest.mark.parametrize("residence", ["device", "host", "mixed", "mixed_base_on_host"])
@pytest.mark.parametrize("case", FROM_CASES, ids=lambda c: "x".join(map(str, c[0])) + f"p{c[1]}-{c[3]}-{c[4]}")
def test_residual_and_grad_from_a_base_array(case, residence):
    """mimi_hip_domain_add_residual_and_grad_from: A_out = A_base + gf K, r += R -- the operator's "J <- M, then
    AddMultGrad" (operators/nonlinear_solid.cpp:257-258) as one pass.  Random A_base, A_out pre-filled with garbage that
    must not survive; compared (1e-11 / 1e-12) with the oracle's assembly on top of the same base and -- bitwise -- with
    the plain "+=" entry applied to a copy of the base; the base array itself is left untouched."""
    import torch
    from oracle import ref_path as rp
    n_el, p, lengths, matname, creator = case
    P, D, G = make_pair(n_el, p, lengths, matname, creator)
    D.set_dt(0.5)
    G.dt_ = 0.5
    u = synthetic_u(P, scale=0.05 if matname == "neohook" else 0.02)
    rng = np.random.default_rng(3)
    base = rng.standard_normal(D.nnz) * 50.0
    r0 = rng.standard_normal(P.n_vdofs)
    gf = 0.37
    r_o, A_o = r0.copy(), base.copy()
    D.add_domain_residual_and_grad(u, gf, r_o, A_o, rp.TANGENT_EXACT)

    dev = torch.device("cuda", 0)
    to = lambda a, on_dev: torch.from_numpy(a.copy()).to(dev) if on_dev else a.copy()
    base_dev, out_dev = {"device": (True, True), "host": (False, False), "mixed": (True, False),
                         "mixed_base_on_host": (False, True)}[residence]
    u_x, r_x = to(u, out_dev), to(r0, out_dev)
    base_x = to(base, base_dev)
    out_x = to(np.full(D.nnz, 1e30), out_dev)            # garbage: every entry must be overwritten
    G.AddDomainResidualAndGradFrom(u_x, gf, r_x, base_x, out_x)
    G.Synchronize()
    host = lambda t: t.cpu().numpy() if isinstance(t, torch.Tensor) else t
    assert relmax(host(r_x), r_o) < 1e-12
    assert relmax(host(out_x), A_o) < 1e-11
    assert np.array_equal(host(base_x), base)            # the base is read, never written
    # the same bits as "copy, then +="
    r_p, A_p = to(r0, out_dev), to(base, out_dev)
    G.AddDomainResidualAndGrad(u_x, gf, r_p, A_p)
    G.Synchronize()
    assert np.array_equal(host(A_p), host(out_x)) and np.array_equal(host(r_p), host(r_x))
    # A_base == A_out is the plain "+="
    A_q, r_q = to(base, out_dev), to(r0, out_dev)
    G.AddDomainResidualAndGradFrom(u_x, gf, r_q, A_q, A_q)
    G.Synchronize()
    assert np.array_equal(host(A_q), host(out_x))


def test_from_a_base_array_is_refused_on_an_element_box():
    """"A_out = A_base + gf K" does not compose over element boxes as "+=" does (a second box would overwrite the rows it
    shares with the first): the entry is for whole-patch handles, a slab handle refuses it -- and still
    accepts A_base == A_out, which is the plain "+="."""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    patch = mimi_amd.BSplinePatch.block((4, 4, 3), 2)
    pattern = CSRPattern.of_bspline_patch(patch)
    G = NonlinearSolid("domain", product_material("neohook"), pattern, patch=patch, element_box=([0, 0, 0], [4, 2, 3])).Prepare()
    u = np.zeros(patch.n_vdofs)
    r, base, out = np.zeros(patch.n_vdofs), np.ones(pattern.nnz), np.zeros(pattern.nnz)
    with pytest.raises(RuntimeError, match="whole-patch handle"):
        G.AddDomainResidualAndGradFrom(u, 1.0, r, base, out)
    G.AddDomainResidualAndGradFrom(u, 1.0, r, base, base)
    assert np.abs(base - 1.0).max() > 0


def test_residual_only_column_kernel_is_bitwise_reproducible_and_agrees_with_the_tangent_assembly():
    """AddDomainResidual of the neo-Hookean law (round 4: one wave per element column, plane accumulators, 9-column node
    gather -- nonlinear_solid.cpp:151-160): the same bits in every run, and the residual the residual+Jacobian assembly
    produces for the same u to rounding (two different kernels: 1e-13 of the largest entry)."""
    P, D, G = make_pair((7, 5, 6), 2, None, "neohook", "bspline")
    u = synthetic_u(P)
    runs = []
    for _ in range(3):
        r = np.zeros(P.n_vdofs)
        G.AddDomainResidual(u, r)
        runs.append(r)
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
    r2, A2 = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    G.AddDomainResidualAndGrad(u, 1.0, r2, A2)
    assert relmax(runs[0], r2) < 1e-13
    r_o = np.zeros(P.n_vdofs)
    D.add_domain_residual(u, r_o)
    assert relmax(runs[0], r_o) < 1e-12
