"""the three modes of the degree-3 pre-pass + the whole step for the library named by MIMI_HIP_LIBRARY (same-box A/B): residual+Jacobian
phases, residual-only assembly, DomainPostTimeAdvance, at BASELINE configuration 3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
n_el, p, material = bench.WORKLOADS["cfg3"]
patch = mimi_amd.BSplinePatch.block(n_el, p)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
G = NonlinearSolid("d", bench.make_material(material), pattern, patch=patch).Prepare()
G.dt_ = 0.5
dev = torch.device("cuda", 0)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
for _ in range(2):
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
G.SetPhaseTiming(True)
acc = np.zeros(3)
for _ in range(5):
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    acc += G.PhaseMsDetail()
acc /= 5
G.SetPhaseTiming(False)
for _ in range(2):
    G.AddDomainResidual(u, r)
G.Synchronize()
t = time.perf_counter()
for _ in range(10):
    G.AddDomainResidual(u, r)
G.Synchronize()
t_r = (time.perf_counter() - t) / 10 * 1e3
cur = torch.cuda.current_stream(dev)
t_c = 0.0
for k in range(4):
    G.ResetState()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur); G.DomainPostTimeAdvance(u); e1.record(cur); e1.synchronize()
    if k: t_c += e0.elapsed_time(e1) / 3
G.ResetState(); r.zero_(); G.AddDomainResidual(u, r); G.Synchronize()
s0 = float(r.abs().sum())
# a second checksum behind a state commit (what DomainPostTimeAdvance wrote is read by this assembly)
G.DomainPostTimeAdvance(u); r.zero_(); G.AddDomainResidual(1.1 * u, r); G.Synchronize()
s1 = float(r.abs().sum())
G.ResetState()
print(os.path.basename(os.environ.get("MIMI_HIP_LIBRARY", "default")), "R+J: pre-pass %.3f contraction %.3f gather %.3f = %.3f ms | residual-only %.3f ms | commit %.3f ms | sum|r| %.17e | after a commit %.17e" % (*acc, acc.sum(), t_r, t_c, s0, s1))
