"""Two-phase path with a permuted node numbering (node_ids) vs lexicographic, cfg2 (64x64x8 p=2)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp, torch, mimi_amd, bench
from mimi_amd.integrators import CSRPattern, NonlinearSolid
n_el = (64, 64, 8)
patch = mimi_amd.BSplinePatch.block(n_el, 2)
pat = CSRPattern.of_bspline_patch(patch, on_device=False)
rowptr, col = np.asarray(pat.rowptr), np.asarray(pat.col)
n = patch.n_vdofs
dev = torch.device('cuda', 0)
def rate(pattern, node_ids, u):
    G = NonlinearSolid("d", bench.make_material("neohookean"), pattern, patch=patch, node_ids=node_ids).Prepare()
    tu = torch.from_numpy(u).to(dev)
    r = torch.zeros(n, dtype=torch.float64, device=dev); A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    for _ in range(3): G.AddDomainResidualAndGrad(tu, 1.0, r, A)
    G.Synchronize(); t0 = time.perf_counter()
    for _ in range(20): G.AddDomainResidualAndGrad(tu, 1.0, r, A)
    G.Synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
u = bench.synthetic_u(patch)
print("lexicographic: %.3f ms" % rate(pat, None, u))
for name, perm in [("random permutation", np.random.default_rng(1).permutation(patch.n_nodes).astype(np.int64)),
                   ("reversed", np.arange(patch.n_nodes - 1, -1, -1, dtype=np.int64))]:
    dofperm = (perm[:, None] * 3 + np.arange(3)[None, :]).ravel()
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    S = sp.coo_matrix((np.ones(len(col)), (dofperm[rows], dofperm[col])), shape=(n, n)).tocsr(); S.sort_indices()
    pp = CSRPattern(S.indptr.astype(np.int64), S.indices.astype(np.int32), len(col))
    up = np.empty_like(u); up[dofperm] = u
    print("%s: %.3f ms" % (name, rate(pp, perm, up)))
