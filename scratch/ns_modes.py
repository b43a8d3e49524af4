"""north-star (or another degree-2 workload) phases for the library named by MIMI_HIP_LIBRARY (same-box A/B): residual+Jacobian phase 1 /
phase 2, residual-only assembly, checksums"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
n_el, p, material = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "northstar"]
patch = mimi_amd.BSplinePatch.block(n_el, p)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
G = NonlinearSolid("d", bench.make_material(material), pattern, patch=patch).Prepare()
G.dt_ = 0.5
dev = torch.device("cuda", 0)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
for _ in range(3):
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
G.SetPhaseTiming(True)
acc = np.zeros(2)
for _ in range(10):
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    acc += G.PhaseMs()
acc /= 10
G.SetPhaseTiming(False)
G.Synchronize()
t = time.perf_counter()
for _ in range(20):
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
G.Synchronize()
t_all = (time.perf_counter() - t) / 20 * 1e3
for _ in range(2):
    G.AddDomainResidual(u, r)
G.Synchronize()
t = time.perf_counter()
for _ in range(20):
    G.AddDomainResidual(u, r)
G.Synchronize()
t_r = (time.perf_counter() - t) / 20 * 1e3
r.zero_(); A.zero_(); G.AddDomainResidualAndGrad(u, 1.0, r, A); G.Synchronize()
print(os.path.basename(os.environ.get("MIMI_HIP_LIBRARY", "default")), "R+J %.3f ms (phase 1 %.3f + phase 2 %.3f) | residual-only %.3f ms | sum|r| %.17e sum|A| %.17e" % (
    t_all, acc[0], acc[1], t_r, float(r.abs().sum()), float(A.abs().sum())))
