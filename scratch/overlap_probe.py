"""Do two independent assemblies overlap when issued on two streams (phase 1 is fp64-pipe-bound, phase 2 HBM-bound)?"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, mimi_amd, bench
from mimi_amd.integrators import CSRPattern, NonlinearSolid
dev = torch.device('cuda', 0)
patch = mimi_amd.BSplinePatch.block((128, 128, 16), 2)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
hs = []
for k in range(2):
    st = torch.cuda.Stream(device=dev)
    G = NonlinearSolid("d", bench.make_material("neohookean"), pattern, patch=patch).Prepare()
    G.SetStream(st.cuda_stream)
    r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev); A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    hs.append((G, r, A))
def run(n, both):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for k, (G, r, A) in enumerate(hs):
            if both or k == 0:
                G.AddDomainResidualAndGrad(u, 1.0, r, A)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(3, True)
print("one handle: %.2f ms per assembly" % run(10, False))
print("two handles on two streams: %.2f ms per pair (%.2f ms per assembly)" % (run(10, True), run(10, True) / 2))
