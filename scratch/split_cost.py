"""Cost of splitting a rank's slab into boundary layers + interior (for overlapping the exchange), one GPU."""
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
for world in (2, 4, 8):
    shard = parallel.SlabShard(patch, pattern, world // 2, world)
    whole = mk(shard.element_box)
    bb, ib = shard.overlap_boxes()
    hb = [mk(b) for b in bb]; hi = mk(ib)
    t1 = timed(lambda: whole.AddDomainResidualAndGrad(u, 1.0, r, A))
    tb = timed(lambda: [g.AddDomainResidualAndGrad(u, 1.0, r, A) for g in hb])
    ti = timed(lambda: hi.AddDomainResidualAndGrad(u, 1.0, r, A))
    print(f"world {world}: whole slab {t1:.2f} ms; boundary layers {tb:.2f} ms + interior {ti:.2f} ms = {tb+ti:.2f} ms")
    del whole, hb, hi

# the two boundary-layer handles of an interior rank on two streams
print("--- boundary layers on two streams ---")
for world in (4, 8):
    shard = parallel.SlabShard(patch, pattern, world // 2, world)
    bb, ib = shard.overlap_boxes()
    s2 = torch.cuda.Stream(device=dev)
    h0 = mk(bb[0]); h1 = mk(bb[1]); h1.SetStream(s2.cuda_stream)
    def both():
        s2.wait_stream(stream)
        h0.AddDomainResidualAndGrad(u, 1.0, r, A)
        h1.AddDomainResidualAndGrad(u, 1.0, r, A)
        stream.wait_stream(s2)
    print(f"world {world}: boundary layers on two streams {timed(both):.2f} ms")
    del h0, h1
