"""same-box A/B of the two contraction kernels of the p = 3 path (MIMI_HIP_P3_CONTRACT = cxx | asm): phase times by the
library's HIP events, checksums of the assembled values (must be equal to the bit)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
n_el, p, material = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
patch = mimi_amd.BSplinePatch.block(n_el, p)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
G = NonlinearSolid("d", bench.make_material(material), pattern, patch=patch).Prepare()
G.dt_ = 0.5
dev = torch.device("cuda", 0)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
sums = {}
for variant in ("cxx", "asm", "cxx", "asm"):
    os.environ["MIMI_HIP_P3_CONTRACT"] = variant
    G.SetPhaseTiming(False)
    for _ in range(2):
        G.AddDomainResidualAndGrad(u, 1.0, r, A)
    G.SetPhaseTiming(True)
    acc = np.zeros(3)
    for _ in range(reps):
        G.AddDomainResidualAndGrad(u, 1.0, r, A)
        acc += G.PhaseMsDetail()
    acc /= reps
    r.zero_(); A.zero_()
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    G.Synchronize()
    sums[variant] = (float(r.abs().sum()), float(A.abs().sum()), float(A.sum()))
    print(variant, "pre-pass %.3f ms  contraction %.3f ms  gather %.3f ms  sum %.3f ms" % (*acc, acc.sum()), "checksums %.17e %.17e" % sums[variant][:2], flush=True)
    r.zero_(); A.zero_()
print("bitwise equal checksums:", sums["cxx"] == sums["asm"])
