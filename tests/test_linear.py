"""The callers' steps after the assembly: essential-dof elimination and the reference's
iterative linear solver (py_nonlinear_solid.cpp:329-339: mfem GMRES + Jacobi).  CPU: the numpy restatement
(oracle/krylov.py) against a sparse direct solve.  GPU: the HIP solver (csrc/krylov.hip, through the C ABI) against the
restatement -- same iteration counts, solutions to 1e-9 relative (floating point, iterative: not bitwise)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from _cases import oracle_material, synthetic_u


def newton_system(n_el=(4, 3, 2), p=2, matname="neohook", fac0=2.5e-4):
    """J = M + fac0 K with the x = 0 face clamped (operators/nonlinear_solid.cpp:240-283), and a right-hand side"""
    from oracle import iga, ref_path as rp, harness as hz
    P = iga.Patch.block(n_el, p)
    D = rp.DomainOracle(P, oracle_material(matname), n_threads=4)
    mass = hz.assemble_mass(P, D.tables, 1.0, D.rowptr, D.col)
    u = synthetic_u(P, scale=0.02)
    r = np.zeros(P.n_vdofs)
    J = mass.copy()
    D.add_domain_residual_and_grad(u, fac0, r, J, rp.TANGENT_EXACT)
    nodes = P.boundary_nodes(0, 0)
    ess = np.sort(np.concatenate([nodes * P.dim + c for c in range(P.dim)])).astype(np.int64)
    return P, D, J, r, ess


def eliminate_host(rowptr, col, vals, r, ess):
    n = len(rowptr) - 1
    mask = np.zeros(n, dtype=bool)
    mask[ess] = True
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    kill = mask[rows] | mask[col]
    vals = vals.copy()
    vals[kill] = 0.0
    vals[kill & (rows == col)] = 1.0
    r = r.copy()
    r[ess] = 0.0
    return vals, r


@pytest.mark.parametrize("jacobi", [True, False])
@pytest.mark.parametrize("kdim", [50, 7])
def test_restated_gmres_against_direct_solve(kdim, jacobi):
    from oracle import krylov
    P, D, J, r, ess = newton_system()
    Jv, rv = eliminate_host(D.rowptr, D.col, J, r, ess)
    A = sp.csr_matrix((Jv, D.col, D.rowptr), shape=(P.n_vdofs, P.n_vdofs))
    x_ref = spla.splu(A.tocsc()).solve(rv)
    x, it, nrm, conv = krylov.gmres(A, rv, kdim=kdim, jacobi=jacobi)
    assert conv and 1 < it <= 300
    assert np.abs(x - x_ref).max() <= 1e-6 * np.abs(x_ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("kdim,jacobi", [(50, True), (7, True), (50, False)])
def test_hip_gmres_and_eliminate_against_restatement(kdim, jacobi, where):
    import torch
    from mimi_amd.integrators import CSRPattern
    from mimi_amd.linear import LinearSolver
    from oracle import krylov
    P, D, J, r, ess = newton_system()
    Jv, rv = eliminate_host(D.rowptr, D.col, J, r, ess)
    A = sp.csr_matrix((Jv, D.col, D.rowptr), shape=(P.n_vdofs, P.n_vdofs))
    x_o, it_o, nrm_o, conv_o = krylov.gmres(A, rv, kdim=kdim, jacobi=jacobi)
    if where == "host":
        pattern = CSRPattern(D.rowptr.astype(np.int64), D.col.astype(np.int32), D.nnz)
        S = LinearSolver(pattern, ess)
        Jg, rg = J.copy(), r.copy()
        S.Eliminate(rg, Jg)
        assert np.array_equal(Jg, Jv) and np.array_equal(rg, rv)
        S.kdim, S.use_jacobi = kdim, jacobi
        x = S.Mult(Jg, rg, np.empty_like(rg))
    else:
        dev = torch.device("cuda", 0)
        pattern = CSRPattern(torch.from_numpy(D.rowptr.astype(np.int64)).to(dev), torch.from_numpy(D.col.astype(np.int32)).to(dev),
                             D.nnz)
        S = LinearSolver(pattern, ess)
        Jg, rg = torch.from_numpy(J).to(dev), torch.from_numpy(r).to(dev)
        S.Eliminate(rg, Jg)
        torch.cuda.synchronize()
        assert np.array_equal(Jg.cpu().numpy(), Jv) and np.array_equal(rg.cpu().numpy(), rv)
        S.kdim, S.use_jacobi = kdim, jacobi
        xg = torch.empty_like(rg)
        S.Mult(Jg, rg, xg)
        x = xg.cpu().numpy()
    assert S.converged_ == conv_o
    assert abs(S.final_iter_ - it_o) <= 1
    # (without the preconditioner the system is worse conditioned: rounding differences of the two Arnoldi processes show
    # at the level of the stopping tolerance, 1e-8 in the residual)
    assert np.abs(x - x_o).max() <= (1e-9 if jacobi else 1e-6) * np.abs(x_o).max()
    # the true residual of the answer
    assert np.linalg.norm(A @ x - rv) <= 1e-6 * np.linalg.norm(rv)


@pytest.mark.gpu
def test_hip_gmres_is_reproducible():
    from mimi_amd.integrators import CSRPattern
    from mimi_amd.linear import LinearSolver
    P, D, J, r, ess = newton_system()
    S = LinearSolver(CSRPattern(D.rowptr.astype(np.int64), D.col.astype(np.int32), D.nnz), ess)
    S.Eliminate(r, J)
    xs = [S.Mult(J, r, np.empty_like(r)).copy() for _ in range(3)]
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(xs[0], xs[2])


def mass_system():
    from oracle import iga, ref_path as rp, harness as hz
    P = iga.Patch.block((4, 3, 2), 2)
    D = rp.DomainOracle(P, oracle_material("neohook"), n_threads=2)
    mass = hz.assemble_mass(P, D.tables, 1.0, D.rowptr, D.col)
    nodes = P.boundary_nodes(0, 0)
    ess = np.sort(np.concatenate([nodes * 3 + c for c in range(3)])).astype(np.int64)
    b = np.random.default_rng(3).standard_normal(P.n_vdofs)
    Mv, bv = eliminate_host(D.rowptr, D.col, mass, b, ess)
    return P, D, Mv, bv, ess


def test_restated_cg_against_direct_solve():
    from oracle import krylov
    P, D, Mv, bv, ess = mass_system()
    A = sp.csr_matrix((Mv, D.col, D.rowptr), shape=(P.n_vdofs, P.n_vdofs))
    x_ref = spla.splu(A.tocsc()).solve(bv)
    x, it, nrm, conv = krylov.cg(A, bv)
    assert conv and 1 < it < 1000
    assert np.abs(x - x_ref).max() <= 1e-6 * np.abs(x_ref).max()


@pytest.mark.gpu
def test_hip_cg_against_restatement():
    """the mass solve of operators::NonlinearSolid (CG + Jacobi) on the device against the restatement"""
    from mimi_amd.integrators import CSRPattern
    from mimi_amd.linear import LinearSolver
    from oracle import krylov
    P, D, Mv, bv, ess = mass_system()
    A = sp.csr_matrix((Mv, D.col, D.rowptr), shape=(P.n_vdofs, P.n_vdofs))
    x_o, it_o, nrm_o, conv_o = krylov.cg(A, bv)
    S = LinearSolver(CSRPattern(D.rowptr.astype(np.int64), D.col.astype(np.int32), D.nnz), ess)
    x = S.MultCG(Mv, bv, np.empty_like(bv))
    # (conjugate gradients lose orthogonality in floating point: the two summation orders differ by a few iterations)
    assert S.converged_ and conv_o and abs(S.final_iter_ - it_o) <= max(3, it_o // 20)
    assert np.abs(x - x_o).max() <= 1e-7 * np.abs(x_o).max()     # both stop at rel 1e-8, a few iterations apart
    assert np.linalg.norm(A @ x - bv) <= 1e-6 * np.linalg.norm(bv)
    x2 = S.MultCG(Mv, bv, np.empty_like(bv))
    assert np.array_equal(x, x2)              # deterministic reductions


def _vector_pattern(nx, ny, nz, vdim):
    """CSR pattern of a vdim-vector field on an nx x ny (x nz) node grid, every node coupled to its 3 x 3 (x 3)
    neighbourhood, byVDIM numbering (node * vdim + c): the vdim rows of a node share their column list"""
    import itertools
    dims = [nx, ny] + ([nz] if nz else [])
    nodes = np.arange(int(np.prod(dims))).reshape(dims)
    rows = []
    for idx in itertools.product(*[range(d) for d in dims]):
        sl = tuple(slice(max(i - 1, 0), min(i + 2, d)) for i, d in zip(idx, dims))
        nb = np.sort(nodes[sl].ravel())
        cols = (nb[:, None] * vdim + np.arange(vdim)[None, :]).ravel()
        rows.extend([cols] * vdim)
    rowptr = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int64)
    return rowptr, np.concatenate(rows).astype(np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,group", [("3d", 3), ("2d", 2), ("2d_of_3n_rows", 2), ("ragged", 1)])
def test_hip_products_on_every_row_grouping(kind, group):
    """the products read a node's column list once for its dofs when the rows of a node share it (csrc/krylov.hip
    kr_row_products): the three forms against scipy, and which form the pattern got"""
    from mimi_amd.integrators import CSRPattern
    from mimi_amd.linear import LinearSolver
    rng = np.random.default_rng(17)
    if kind == "3d":
        rowptr, col = _vector_pattern(9, 7, 6, 3)          # rows of 24 .. 81 entries: the tail and the two-per-trip loop
    elif kind == "2d":
        rowptr, col = _vector_pattern(23, 19, 0, 2)
    elif kind == "2d_of_3n_rows":
        rowptr, col = _vector_pattern(24, 15, 0, 2)        # 720 rows: divisible by 3, but grouped in pairs
    else:
        n = 601
        A0 = sp.random(n, n, density=0.3, random_state=5, format="csr") + sp.eye(n, format="csr")
        A0.sort_indices()
        rowptr, col = A0.indptr.astype(np.int64), A0.indices.astype(np.int32)
    n = len(rowptr) - 1
    val = rng.standard_normal(len(col))
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    S = LinearSolver(CSRPattern(rowptr, col, len(col)))
    assert S.RowGroup() == group
    assert S.NodeColumns() == (kind == "3d")      # (node triples: read as one index per node)
    x = rng.standard_normal(n)
    y0 = rng.standard_normal(n)
    y = S.AddMult(val, x, y0.copy(), alpha=-0.75)
    exp = y0 - 0.75 * (A @ x)
    assert np.abs(y - exp).max() <= 1e-13 * np.abs(exp).max()
    # the solver's own product: diagonally dominant values, GMRES without restart converges and the answer solves A x = b
    val2 = val * 0.02
    diag = np.flatnonzero(col == np.repeat(np.arange(n), np.diff(rowptr)))
    assert len(diag) == n
    val2[diag] = 4.0 + rng.random(n)
    A2 = sp.csr_matrix((val2, col, rowptr), shape=(n, n))
    b = rng.standard_normal(n)
    xs = S.Mult(val2, b, np.empty(n))
    assert S.converged_ and S.final_iter_ < 50
    assert np.linalg.norm(A2 @ xs - b) <= 1e-7 * np.linalg.norm(b)


@pytest.mark.gpu
def test_hip_products_without_node_triples():
    """rows in groups of three with one column list that is NOT made of node triples (every third column dropped): the
    group form without the node-column shortcut"""
    from mimi_amd.integrators import CSRPattern
    from mimi_amd.linear import LinearSolver
    rng = np.random.default_rng(23)
    rowptr0, col0 = _vector_pattern(6, 5, 4, 3)
    n = len(rowptr0) - 1
    rows = []
    for r in range(0, n, 3):
        c = col0[rowptr0[r]:rowptr0[r + 1]]
        keep = c[(np.arange(len(c)) % 3 != 1) | (c // 3 == r // 3)]        # drops the middle dof of the other nodes
        rows.extend([keep] * 3)
    rowptr = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int64)
    col = np.concatenate(rows).astype(np.int32)
    val = rng.standard_normal(len(col))
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    S = LinearSolver(CSRPattern(rowptr, col, len(col)))
    assert S.RowGroup() == 3 and not S.NodeColumns()
    x, y0 = rng.standard_normal(n), rng.standard_normal(n)
    y = S.AddMult(val, x, y0.copy(), alpha=1.25)
    exp = y0 + 1.25 * (A @ x)
    assert np.abs(y - exp).max() <= 1e-13 * np.abs(exp).max()
