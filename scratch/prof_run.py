"""In-kernel stage stamps of the colour-partitioned tensor kernel: build the library with -DMH_PROFILE
(scratch/mklib.sh PROF -DMH_PROFILE), put it in place of mimi_amd/lib/libmimi_hip.so and run with
MIMI_HIP_TENSOR_VARIANT=valu."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import torch, mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
from mimi_amd import _capi
import bench
n_el = (128,128,16) if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1].split('x'))
patch = mimi_amd.BSplinePatch.block(n_el, 2)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
G = NonlinearSolid("d", bench.make_material("neohookean"), pattern, patch=patch).Prepare()
dev = torch.device('cuda', 0)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
L = _capi.lib()
out = (C.c_ulonglong * 12)()
G.AddDomainResidualAndGrad(u, 1.0, r, A); G.Synchronize()
L.mimi_hip_debug_profile(G._h, out, 1)
G.AddDomainResidualAndGrad(u, 1.0, r, A); G.Synchronize()
L.mimi_hip_debug_profile(G._h, out, 1)
v = np.array(list(out), dtype=np.float64)
names = ['loop-top/prev-scatter-tail', 'stage0 LDS fill+prefetch', 'stageA material', 'stageR residual', 'issue old loads', 'S12', 'S3', 'carry', 'Kv select', 'scatter stores', '', '']
n_items = patch.n_elements * 3
print('shader cycles per (element,i):')
for k in range(12):
    print('  %-28s %10.1f  (%.1f%%)' % (names[k], v[k] / n_items, 100 * v[k] / v.sum()))
print('  total %.1f per (el,i)' % (v.sum() / n_items))
