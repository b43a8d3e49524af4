"""Boundary-layer thickness of the overlapped multi-GPU step (per-rank cost on one GPU, no exchange)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, mimi_amd, bench
from mimi_amd.integrators import CSRPattern, NonlinearSolid
from mimi_amd import parallel
dev = torch.device('cuda', 0)
patch = mimi_amd.BSplinePatch.block((128, 128, 16), 2)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
def mk(box):
    g = NonlinearSolid("d", bench.make_material("neohookean"), pattern, patch=patch, element_box=box).Prepare(); g.SetStream(stream.cuda_stream); return g
def timed(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for world in (4, 8):
    shard = parallel.SlabShard(patch, pattern, world // 2, world)
    for layers in (2, 3, 4, 5, 6):
        bb, ib = shard.overlap_boxes(layers)
        hb = [mk(b) for b in bb]; hi = mk(ib)
        tb = timed(lambda: [g.AddDomainResidualAndGrad(u, 1.0, r, A) for g in hb])
        ti = timed(lambda: hi.AddDomainResidualAndGrad(u, 1.0, r, A))
        print(f"world {world} layers {layers}: boxes {[b[1][1]-b[0][1] for b in bb]} + {ib[1][1]-ib[0][1]}: boundary {tb:.3f} ms + interior {ti:.3f} ms = {tb+ti:.3f} ms", flush=True)
        del hb, hi
