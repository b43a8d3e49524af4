"""VALUES at BASELINE.json's full sizes: complete CSR rows and residual entries of sampled nodes, compared with the
oracle.  The oracle cannot assemble these meshes in reasonable time, but it can integrate the <= (p+1)^3 elements around
a node: for ~50 nodes (corners, edges, faces, interior, and the LAST rows of the matrix -- for configuration 5 those lie
beyond 2^31 in the value array) it computes every element block exactly (oracle/ref_path.c element_residual_and_grad,
exact tangent), tests/_sampling.py sums the rows of the sampled nodes out of them (that algebra is checked against the
oracle's own whole assembly in tests/test_sampled_rows_cpu.py) and the rows are compared with the full GPU assembly.
Bars as everywhere: residual 1e-12, tangent 1e-11 (relative to the largest sampled entry)."""
import numpy as np
import pytest

from _sampling import SampledRows, sample_nodes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,n_random", [("cfg2", 24), ("northstar", 24), ("cfg3", 4), ("cfg4_domain", 24), ("cfg5", 24)])
def test_sampled_rows_match_the_oracle(workload, n_random):
    import torch
    import bench
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga
    n_el, p, material = bench.WORKLOADS[workload]
    dev = torch.device("cuda", 0)
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
    G = NonlinearSolid("domain", bench.make_material(material), pattern, patch=patch).Prepare()
    G.dt_ = 0.5
    assert G.path_ == 1
    u_host = bench.synthetic_u(patch)
    u = torch.from_numpy(u_host).to(dev)
    r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
    A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    G.Synchronize()
    # the residual-only assembly is a different set of kernels (tensor_residual_col_kernel at degree 2, tp3_point_kernel<0, 0>
    # at degree 3: integrators/nonlinear_solid.cpp:151-160, two of the three assemblies of a Newton iteration): its entries at
    # the sampled nodes are compared with the same oracle values
    r_only = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
    G.AddDomainResidual(u, r_only)
    G.Synchronize()

    P = iga.Patch.block(n_el, p)
    nodes = sample_nodes(P.n, n_random, seed=5)
    S = SampledRows(P, bench._oracle_material(material), nodes, u_host)
    rowptr = pattern.rowptr if isinstance(pattern.rowptr, torch.Tensor) else torch.from_numpy(np.asarray(pattern.rowptr))
    col = pattern.col if isinstance(pattern.col, torch.Tensor) else torch.from_numpy(np.asarray(pattern.col))
    scale_r = float(r.abs().max())
    worst_r = worst_r_only = worst_A = scale_A = 0.0
    beyond_int32 = rows_checked = 0
    for k, node in enumerate(S.node_ids):
        for i in range(3):
            row = node * 3 + i
            lo, hi = int(rowptr[row]), int(rowptr[row + 1])
            beyond_int32 += lo > 2 ** 31
            exp, r_exp = S.row(k, i, col[lo:hi].cpu().numpy())
            got = A[lo:hi].cpu().numpy()
            scale_A = max(scale_A, float(np.abs(exp).max()))
            worst_A = max(worst_A, float(np.abs(got - exp).max()))
            worst_r = max(worst_r, abs(float(r[row]) - r_exp))
            worst_r_only = max(worst_r_only, abs(float(r_only[row]) - r_exp))
            rows_checked += 1
    assert rows_checked >= 3 * 30
    assert worst_r_only / scale_r < 1e-12
    if pattern.nnz > 2 ** 31:
        assert beyond_int32 >= 12       # rows whose values start beyond what an int32 offset reaches
    assert worst_r / scale_r < 1e-12
    assert worst_A / scale_A < 1e-11


def test_stateful_j2_at_cfg3_size():
    """BASELINE configuration 3 is "implicit dynamics": between time steps `DomainPostTimeAdvance` commits the
    return-mapped state (integrators/nonlinear_solid.cpp:179-199, materials/materials.hpp:311-391 accumulate branch) and
    the next assembly starts from it.  On the FULL 128 x 128 x 16 p = 3 mesh: commit at u0 (32.8 M points x 11 state
    doubles), assemble at u; the oracle commits the same u0 on the elements around ~35 sampled nodes.  Compared: the
    committed eqps / plastic strain / temperature of every point of those elements (1e-9, the bar of the small-mesh
    tests), then complete CSR rows (1e-11) and residual entries (1e-12) integrated from the committed state."""
    import torch
    import bench
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga
    n_el, p, material = bench.WORKLOADS["cfg3"]
    dev = torch.device("cuda", 0)
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
    G = NonlinearSolid("domain", bench.make_material(material), pattern, patch=patch).Prepare()
    G.dt_ = 0.5
    assert G.path_ == 1
    u0_host = bench.synthetic_u(patch, scale=0.04, seed=7)
    u_host = bench.synthetic_u(patch)
    G.DomainPostTimeAdvance(torch.from_numpy(u0_host).to(dev))
    u = torch.from_numpy(u_host).to(dev)
    r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
    A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    G.Synchronize()
    assert G.LastKernelFamily() == "tensor_p3_two_phase"

    P = iga.Patch.block(n_el, p)
    nodes = sample_nodes(P.n, 2, seed=11)
    S = SampledRows(P, bench._oracle_material(material), nodes, u_host, u_commit=u0_host)
    # the committed state of the sampled elements
    eqps = G.State("accumulated_plastic_strain")
    eps_p = G.State("plastic_strain")
    temp = G.State("temperature")
    D = S.D
    assert D.eqps.max() > 1e-3 and np.count_nonzero(D.eqps) > 0.5 * D.eqps.size       # the commit did yield
    el = S.elements
    assert np.allclose(eqps[el], D.eqps, rtol=1e-9, atol=1e-13)
    assert np.allclose(eps_p[el], D.plastic_strain, rtol=1e-9, atol=1e-13)
    assert np.allclose(temp[el], D.temperature, rtol=1e-12, atol=1e-12)
    # and the virgin state must NOT reproduce these rows: the test would pass vacuously if the commit were ignored
    rowptr = pattern.rowptr if isinstance(pattern.rowptr, torch.Tensor) else torch.from_numpy(np.asarray(pattern.rowptr))
    col = pattern.col if isinstance(pattern.col, torch.Tensor) else torch.from_numpy(np.asarray(pattern.col))
    scale_r = float(r.abs().max())
    worst_r = worst_A = scale_A = 0.0
    for k, node in enumerate(S.node_ids):
        for i in range(3):
            row = node * 3 + i
            lo, hi = int(rowptr[row]), int(rowptr[row + 1])
            exp, r_exp = S.row(k, i, col[lo:hi].cpu().numpy())
            got = A[lo:hi].cpu().numpy()
            scale_A = max(scale_A, float(np.abs(exp).max()))
            worst_A = max(worst_A, float(np.abs(got - exp).max()))
            worst_r = max(worst_r, abs(float(r[row]) - r_exp))
    assert worst_r / scale_r < 1e-12
    assert worst_A / scale_A < 1e-11
    G.ResetState()
    r.zero_()
    A.zero_()
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    G.Synchronize()
    node = S.node_ids[len(S.node_ids) // 2]
    lo, hi = int(rowptr[node * 3]), int(rowptr[node * 3 + 1])
    exp, _ = S.row(len(S.node_ids) // 2, 0, col[lo:hi].cpu().numpy())
    assert np.abs(A[lo:hi].cpu().numpy() - exp).max() / scale_A > 1e-6


def test_contact_at_cfg4_size_matches_the_oracle():
    """BASELINE configuration 4's contact face at full size (96 x 96 x 12 p = 2: 9 216 faces on the top face, rigid sphere
    of SURVEY 8d): the face integrals are cheap enough for the oracle to do ALL of them, so residual, nodal pressures and
    the complete frozen-pressure tangent are compared (the CSR pattern comes from the library: positions are checked by the
    oracle's own column search).  PARITY UNPINNED like every contact check (oracle/contact_path.c)."""
    import bench
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, MortarContact, RigidSphere
    from oracle import iga, ref_path as rp
    from test_contact import sphere_over_top
    n_el, p, _ = bench.WORKLOADS["cfg4_domain"]
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern.of_bspline_patch(patch)              # host arrays
    rowptr, col = np.asarray(pattern.rowptr), np.asarray(pattern.col)
    P = iga.Patch.block(n_el, p)
    body = sphere_over_top(P, 2)
    Cn = rp.ContactOracle(P, 2, 1, body, penalty=1e4, rowptr=rowptr, col=col)
    G = MortarContact(RigidSphere(body["center"], body["radius"], 1e4), "contact", pattern, patch, 2, 1).Prepare()
    u = bench.synthetic_u(patch, scale=0.05)
    r_o, r_g = np.zeros(P.n_vdofs), np.zeros(P.n_vdofs)
    A_o, A_g = np.zeros(pattern.nnz), np.zeros(pattern.nnz)
    Cn.add_boundary_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    G.AddBoundaryResidualAndGrad(u, 1.0, r_g, A_g)
    assert np.abs(r_o).max() > 0 and np.count_nonzero(Cn.pressure) > 100        # the sphere does press on the face
    assert np.abs(r_g - r_o).max() / np.abs(r_o).max() < 1e-12
    assert np.allclose(G.AveragePressure(), Cn.pressure, rtol=1e-12, atol=1e-12)
    assert np.abs(A_g - A_o).max() / np.abs(A_o).max() < 1e-11


def test_from_base_entry_at_cfg2_size():
    """mimi_hip_domain_add_residual_and_grad_from (ABI 11: the operator's J = M + fac0 K in one pass,
    operators/nonlinear_solid.cpp:257-258) at BASELINE configuration 2's size, device-resident: complete sampled CSR rows of
    A_out equal base + gf K (oracle element blocks, 1e-11), every entry of A_out was written (pre-filled with 1e30), the base
    array is untouched, and the result is the bits of "copy, then +="."""
    import torch
    import bench
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga
    n_el, p, material = bench.WORKLOADS["cfg2"]
    dev = torch.device("cuda", 0)
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
    G = NonlinearSolid("domain", bench.make_material(material), pattern, patch=patch).Prepare()
    G.dt_ = 0.5
    u_host = bench.synthetic_u(patch)
    u = torch.from_numpy(u_host).to(dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    base = torch.randn(pattern.nnz, dtype=torch.float64, device=dev, generator=gen) * 100.0
    base_copy = base.clone()
    gf = 0.37
    r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
    A = torch.full((pattern.nnz,), 1e30, dtype=torch.float64, device=dev)
    G.AddDomainResidualAndGradFrom(u, gf, r, base, A)
    r2 = torch.zeros_like(r)
    A2 = base.clone()
    G.AddDomainResidualAndGrad(u, gf, r2, A2)
    G.Synchronize()
    assert torch.equal(base, base_copy)
    assert torch.equal(A, A2) and torch.equal(r, r2)
    assert float(A.abs().max()) < 1e29
    P = iga.Patch.block(n_el, p)
    nodes = sample_nodes(P.n, 8, seed=5)
    S = SampledRows(P, bench._oracle_material(material), nodes, u_host)
    rowptr = pattern.rowptr if isinstance(pattern.rowptr, torch.Tensor) else torch.from_numpy(np.asarray(pattern.rowptr))
    col = pattern.col if isinstance(pattern.col, torch.Tensor) else torch.from_numpy(np.asarray(pattern.col))
    worst = scale = 0.0
    for k, node in enumerate(S.node_ids):
        for i in range(3):
            row = node * 3 + i
            lo, hi = int(rowptr[row]), int(rowptr[row + 1])
            exp, _ = S.row(k, i, col[lo:hi].cpu().numpy())
            got = (A[lo:hi] - base[lo:hi]).cpu().numpy() / gf
            scale = max(scale, float(np.abs(exp).max()))
            worst = max(worst, float(np.abs(got - exp).max()))
    assert worst / scale < 1e-11
