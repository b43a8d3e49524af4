"""phase times of the p = 3 path for the library named by MIMI_HIP_LIBRARY (timing experiments: scratch/p3_variants.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
n_el, p, material = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
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
print(os.environ.get("MIMI_HIP_LIBRARY", "default"), "pre-pass %.2f ms  integration/contraction %.2f ms  gather %.2f ms  sum %.2f ms" % (*acc, acc.sum()))
# a checksum of the result: variants must agree to rounding
print("   checksum r %.15e  A %.15e" % (float(r.abs().sum()), float(A.abs().sum())))

