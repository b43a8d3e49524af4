"""which entries of the degree-3 tangent differ from the oracle's (debugging aid): python scratch/p3_debug.py NX NY NZ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_domain_gpu import make_pair
from _cases import synthetic_u
from oracle import ref_path as rp
n_el = tuple(int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 1, 3)
P, D, G = make_pair(n_el, 3, None, "neohook", "bspline")
u = synthetic_u(P, scale=0.05)
r_o, r_g, A_o, A_g = np.zeros(P.n_vdofs), np.zeros(P.n_vdofs), np.zeros(D.nnz), np.zeros(D.nnz)
D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
G.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
A_2 = np.zeros(D.nnz); r_2 = np.zeros(P.n_vdofs)
G.AddDomainResidualAndGrad(u, 1.0, r_2, A_2)
print("residual err", np.abs(r_g - r_o).max() / np.abs(r_o).max(), " tangent err", np.abs(A_g - A_o).max() / np.abs(A_o).max(),
      " run-to-run", np.abs(A_2 - A_g).max())
bad = np.nonzero(np.abs(A_g - A_o) > 1e-9 * np.abs(A_o).max())[0]
print(len(bad), "bad entries of", D.nnz)
rowptr, col = P.sparsity()
rows = np.searchsorted(rowptr, bad, side="right") - 1
n = [n_el[d] + 3 for d in range(3)]
import collections
c = collections.Counter()
for k, row in zip(bad[:200000], rows[:200000]):
    A, i = divmod(int(row), 3)
    B, j = divmod(int(col[k]), 3)
    Az, Bz = A // (n[0] * n[1]), B // (n[0] * n[1])
    c[(i, j, Az, Bz)] += 1
for key in sorted(c)[:80]:
    print("  (i, j, node-plane of row, of column) =", key, c[key])
