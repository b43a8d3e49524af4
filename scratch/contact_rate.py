"""cfg4 on one GPU: 96 x 96 x 12 p = 2 block, domain assembly + mortar contact (rigid sphere over the top face, and
the same sphere as a NURBS surface patch is not needed here: analytic body) -- times of the contact integrator."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimi_amd, bench
from mimi_amd.integrators import CSRPattern, NonlinearSolid, MortarContact, RigidSphere
dev = torch.device("cuda", 0)
n_el = (96, 96, 12)
patch = mimi_amd.BSplinePatch.block(n_el, 2)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
G = NonlinearSolid("domain", bench.make_material("neohookean"), pattern, patch=patch).Prepare(); G.SetStream(stream.cuda_stream)
L = patch.control_points.max(axis=0); R = 0.25 * L[0]; c = 0.5 * L; c[2] = L[2] + 0.9 * R
Cn = MortarContact(RigidSphere(list(c), R, 1e4), "contact", pattern, patch, 2, 1).Prepare(); Cn.SetStream(stream.cuda_stream)
u = torch.from_numpy(bench.synthetic_u(patch, scale=0.01)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
td = timed(lambda: G.AddDomainResidualAndGrad(u, 1.0, r, A))
tc = timed(lambda: Cn.AddBoundaryResidualAndGrad(u, 1.0, r, A))
tr = timed(lambda: Cn.AddBoundaryResidual(u, r))
print(f"cfg4 96x96x12 p=2: domain R+J {td:.3f} ms ({patch.n_elements / td / 1e3:.1f} M element-integrations/s); "
      f"contact ({Cn.n_marked_boundaries_} faces) R+J {tc:.3f} ms, R {tr:.3f} ms")
