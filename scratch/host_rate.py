"""PCIe-inclusive rate: the same assembly with HOST u / r / A buffers (the reference passes mfem host data;
the library stages them, the call is synchronous).  cfg2-sized and north-star-sized."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, mimi_amd, bench
from mimi_amd.integrators import CSRPattern, NonlinearSolid
for n_el in [(64, 64, 8), (128, 128, 16)]:
    patch = mimi_amd.BSplinePatch.block(n_el, 2)
    pattern = CSRPattern.of_bspline_patch(patch, on_device=False)
    G = NonlinearSolid("d", bench.make_material("neohookean"), pattern, patch=patch).Prepare()
    u = bench.synthetic_u(patch)
    r = np.zeros(patch.n_vdofs)
    A = np.zeros(pattern.nnz)
    G.AddDomainResidualAndGrad(u, 1.0, r, A)
    t0 = time.perf_counter(); n = 3
    for _ in range(n):
        G.AddDomainResidualAndGrad(u, 1.0, r, A)
    dt = (time.perf_counter() - t0) / n
    print(f"{n_el}: host buffers {dt*1e3:.1f} ms per assembly = {patch.n_elements/dt/1e6:.2f} M element-integrations/s "
          f"(A values {pattern.nnz*8/1e9:.2f} GB each way)")
