"""What one rank of an N-GPU bench does besides the RCCL transfer itself, timed on one GPU."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, mimi_amd, bench
from mimi_amd.integrators import CSRPattern, NonlinearSolid
from mimi_amd import parallel
dev = torch.device('cuda', 0)
patch = mimi_amd.BSplinePatch.block((128, 128, 16), 2)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
for world in (2, 4, 8):
    rank = world // 2
    shard = parallel.SlabShard(patch, pattern, rank, world)
    G = NonlinearSolid("d", bench.make_material("neohookean"), pattern, patch=patch, element_box=shard.element_box).Prepare()
    r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
    A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    ex = parallel.InterfaceExchange(shard, r, A, dev, mode="owner")
    def timed(f, n=10):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_c = timed(lambda: G.AddDomainResidualAndGrad(u, 1.0, r, A))
    t_z = timed(lambda: ex.zero_interface(True))
    def pack_unpack():     # (the library's row kernels since ABI 8; torch index ops before)
        for s in ex.sides:
            ex._pack(s, True)
        for s in ex.sides:
            ex._unpack_add(s, True)
    t_p = timed(pack_unpack)
    vol = sum(s["send"].numel() for s in ex.sides) * 8 / 1e6
    print(f"world {world}: slab {shard.element_box}: assembly {t_c:.2f} ms, zero_interface {t_z:.2f} ms, pack+unpack {t_p:.2f} ms, "
          f"send volume {vol:.1f} MB to {len(ex.sides)} neighbours")
    del G, ex
