"""Throughput of the general kernels on a large 2-D block (the reference's own examples are 2-D)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimi_amd, bench
from mimi_amd.integrators import CSRPattern, NonlinearSolid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = int(sys.argv[2]) if len(sys.argv) > 2 else 2
patch = mimi_amd.BSplinePatch.block((n, n), p)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
G = NonlinearSolid("domain", bench.make_material("neohookean"), pattern, patch=patch).Prepare()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(); G.SetStream(stream.cuda_stream)
rng = np.random.default_rng(1)
u = torch.from_numpy(0.05 * rng.standard_normal(patch.n_vdofs)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
for grad in (True, False):
    for _ in range(2):
        G.AddDomainResidualAndGrad(u, 1.0, r, A) if grad else G.AddDomainResidual(u, r)
    G.Synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        G.AddDomainResidualAndGrad(u, 1.0, r, A) if grad else G.AddDomainResidual(u, r)
    G.Synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"2-D {n}x{n} p={p} {'R+J' if grad else 'R'}: {dt*1e3:.2f} ms, {patch.n_elements/dt/1e6:.1f} M element-integrations/s, path {G.path_}")
