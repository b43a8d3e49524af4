"""The algebra of tests/_sampling.py (rows of sampled nodes out of the oracle's element blocks) against the oracle's own
whole assembly, on meshes small enough for both: what the full-size GPU checks (tests/test_fullsize_sampled_gpu.py) rest on."""
import numpy as np
import pytest

from _cases import oracle_material
from _sampling import SampledRows, sample_nodes


@pytest.mark.parametrize("n_el,p,matname", [((6, 5, 4), 2, "neohook"), ((4, 4, 3), 3, "j2")])
def test_sampled_rows_equal_the_assembled_rows(n_el, p, matname):
    from oracle import iga, ref_path as rp
    P = iga.Patch.block(n_el, p)
    mat = oracle_material(matname)
    D = rp.DomainOracle(P, mat, n_threads=2)
    D.set_dt(0.5)
    u = 0.05 * np.random.default_rng(1).standard_normal(P.n_vdofs)
    r = np.zeros(P.n_vdofs)
    A = np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r, A, rp.TANGENT_EXACT)
    nodes = sample_nodes(P.n, 4, seed=5)
    S = SampledRows(P, mat, nodes, u)
    for k, node in enumerate(S.node_ids):
        for i in range(3):
            row = node * 3 + i
            lo, hi = D.rowptr[row], D.rowptr[row + 1]
            exp, r_exp = S.row(k, i, D.col[lo:hi])
            assert np.abs(exp - A[lo:hi]).max() <= 1e-13 * np.abs(A).max()
            assert abs(r_exp - r[row]) <= 1e-13 * np.abs(r).max()
