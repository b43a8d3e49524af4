"""The RCCL transport of the interface exchange on ONE GPU: a one-rank `nccl` communicator whose sends and receives
go to the rank itself.  It runs what the gloo rehearsals cannot: `init_process_group("nccl", device_id=...)`, the
all-reduce on device tensors, and `InterfaceExchange.start / finish` on device buffers with no host staging, ordered
against the integration kernels by the stream alone (pack -> send/recv -> interior integration -> unpack), exactly as
`bench.py` drives it for N > 1.  Neighbour-to-neighbour traffic over xGMI needs two GPUs and is not covered here."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(port, n_el, p, fake_rank, fake_world, mode, scheme, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        import bench
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=bench._rccl_options())
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)
        assert int(ones.item()) == 1
        import bench
        import mimi_amd
        from mimi_amd import parallel
        from mimi_amd.integrators import CSRPattern, NonlinearSolid
        patch = mimi_amd.BSplinePatch.block(n_el, p)
        shard = parallel.SlabShard(patch, None, fake_rank, fake_world)       # the slab rank `fake_rank` of `fake_world` would own
        pattern = CSRPattern.of_bspline_patch(patch, on_device=True, node_box=shard.node_box())
        shard.pattern = pattern
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
        r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
        A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
        ex = parallel.InterfaceExchange(shard, r, A, dev, mode=mode, loopback=True)
        if scheme == "gather":
            # one handle: integrate, gather the rows that leave first, send them while the rest is gathered
            g = NonlinearSolid("domain", bench.make_material("neohookean"), pattern, patch=patch, element_box=shard.element_box).Prepare()
            g.SetStream(stream.cuda_stream)
            handles = [g]
            early, rest = ex.gather_windows()
            assert len(early) == len(ex.sides)
            for _ in range(3):
                ex.zero_interface(True)
                g.Integrate(u)
                for w in early:
                    g.Gather(1.0, r, A, *w)
                ready = torch.cuda.Event()
                ready.record(stream)
                g.Gather(1.0, r, A, *rest)
                ex.start(True, ready=ready)
                ex.finish()
        else:
            boundary_boxes, interior_box = shard.overlap_boxes(mode=mode)
            assert boundary_boxes
            handles = []
            for box in boundary_boxes + [interior_box]:
                g = NonlinearSolid("domain", bench.make_material("neohookean"), pattern, patch=patch, element_box=box).Prepare()
                g.SetStream(stream.cuda_stream)
                handles.append(g)
            # a middle rank has two boundary boxes that share no node: the second runs beside the first on its own stream
            side = None
            if len(boundary_boxes) == 2 and shard.boxes_share_no_node(boundary_boxes):
                side = torch.cuda.Stream(device=dev)
                handles[1].SetStream(side.cuda_stream)
            for _ in range(3):                       # several steps back to back: buffers are reused without a host sync
                ex.zero_interface(True)
                if side:
                    side.wait_stream(stream)
                for g in handles[:-1]:
                    g.AddDomainResidualAndGrad(u, 1.0, r, A)
                if side:
                    stream.wait_stream(side)
                ready = torch.cuda.Event()            # the interface rows are complete here; the sends wait for this only
                ready.record(stream)
                handles[-1].AddDomainResidualAndGrad(u, 1.0, r, A)
                ex.start(True, ready=ready)
                ex.finish()
        torch.cuda.synchronize()
        for g in handles:
            g.Synchronize()
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
        # expected: one local assembly of the slab; then, per side, what was sent added into the rows it was received into
        r1 = torch.zeros_like(r)
        A1 = torch.zeros_like(A)
        G = NonlinearSolid("domain", bench.make_material("neohookean"), pattern, patch=patch, element_box=shard.element_box).Prepare()
        G.SetStream(stream.cuda_stream)
        G.AddDomainResidualAndGrad(u, 1.0, r1, A1)
        G.Synchronize()
        torch.cuda.synchronize()
        def positions(rows):
            # positions in the value array of all entries of `rows`, row after row
            start = pattern.rowptr[rows]
            length = pattern.rowptr[rows + 1] - start
            offs = torch.cumsum(length, 0) - length
            return torch.repeat_interleave(start - offs, length) + torch.arange(int(length.sum()), device=dev)

        r_exp, A_exp = r1.clone(), A1.clone()
        for i, j in ex.loopback_pairs():           # side i's message lands in side j's receive buffer (same length)
            s, t = ex.sides[i], ex.sides[j]
            assert s["peer"] == 0 and s["srows"].numel() == t["rrows"].numel() and s["sidx"].numel() == t["ridx"].numel()
            r_exp.index_add_(0, t["rrows"], r1[s["srows"]])
            if ex.trim:
                # (trimmed messages, round 5: the entries the sender lists land where the receiving side lists them -- in loop-back
                # those are this rank's own two lists, cut for different neighbours; that the lists of two REAL neighbours
                # name the same pairs is what tests/test_parallel_{cpu,gpu}.py check against a whole-patch assembly)
                A_exp.index_add_(0, t["ridx"], A1[s["sidx"]])
            else:
                A_exp.index_add_(0, positions(t["rrows"]), A1[positions(s["srows"])])
        shared_rows = torch.cat([s["rrows"] for s in ex.sides])
        shared_idx = positions(shared_rows)
        er = float((r[shared_rows] - r_exp[shared_rows]).abs().max() / r1.abs().max())
        eA = float((A[shared_idx] - A_exp[shared_idx]).abs().max() / A1.abs().max())
        moved = float(r1[torch.cat([s["srows"] for s in ex.sides])].abs().max())
        q.put((True, er, eA, moved, len(ex.sides)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((False, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("scheme", ["boundary", "gather"])
@pytest.mark.parametrize("n_el,p,fake_rank,fake_world,mode", [((4, 4, 18), 2, 1, 3, "replicate"), ((4, 4, 12), 2, 0, 2, "owner"),
                                                               ((3, 3, 24), 3, 1, 3, "replicate"),
                                                               # owner mode at odd degree: a side sends 1 plane and receives 2
                                                               ((3, 3, 24), 3, 1, 3, "owner")])
def test_interface_exchange_over_rccl_loopback(n_el, p, fake_rank, fake_world, mode, scheme):
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_worker, args=(port, n_el, p, fake_rank, fake_world, mode, scheme, q))
    pr.start()
    try:
        res = q.get(timeout=300)
        pr.join(timeout=60)
    finally:
        if pr.is_alive():
            pr.terminate()
            pr.join(timeout=10)
    assert res[0] is True, res[1]
    _, er, eA, moved, n_sides = res
    assert n_sides == (2 if 0 < fake_rank < fake_world - 1 else 1)
    assert moved > 0.0                      # the rows on the wire were not trivially zero
    assert er < 1e-13 and eA < 1e-13, (er, eA)
