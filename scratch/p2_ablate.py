"""phase times of the degree-2 two-phase path for the library named by MIMI_HIP_LIBRARY (timing experiments: scratch/ab_lib.sh)"""
import os, sys
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
n = 10
for _ in range(n):
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    acc += G.PhaseMs()
acc /= n
print(os.environ.get("MIMI_HIP_LIBRARY", "default"), "phase 1 %.3f ms  phase 2 %.3f ms  sum %.3f ms" % (*acc, acc.sum()))
print("   checksum r %.15e  A %.15e" % (float(r.abs().sum()), float(A.abs().sum())))
