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
