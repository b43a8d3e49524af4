"""N > 1 path with the HIP kernels: two (three) processes share the one GPU of the test box, each integrates
its element slab, the interface rows travel through `gloo` (staged through the host; RCCL needs one GPU per
rank); every rank then checks the rows it owns against its own whole-patch assembly."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_ranks(ctx, procs, q, world):
    """start the ranks, collect one result each; whatever happens, no rank is left holding the GPU"""
    for pr in procs:
        pr.start()
    try:
        results = [q.get(timeout=300) for _ in range(world)]
        for pr in procs:
            pr.join(timeout=60)
        assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
        return results
    finally:
        for pr in procs:
            if pr.is_alive():
                pr.terminate()
                pr.join(timeout=10)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_el, mode, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        import mimi_amd
        from mimi_amd import parallel
        from mimi_amd.integrators import CSRPattern, NonlinearSolid
        dev = torch.device("cuda", 0)
        patch = mimi_amd.BSplinePatch.block(n_el, 2)
        full = CSRPattern.of_bspline_patch(patch, on_device=True)
        shard = parallel.SlabShard(patch, None, rank, world)
        # "+slice": this rank's value array holds only the rows of the nodes its slab touches (as bench.py does)
        pattern = CSRPattern.of_bspline_patch(patch, on_device=True, node_box=shard.node_box()) if "+slice" in mode else full
        shard.pattern = pattern
        if "+slice" in mode:
            assert pattern.nnz < full.nnz
        # the library enqueues on the stream it is given: the same (non-default: a null handle means "the
        # handle's own stream") one the exchange's torch ops use, as in bench.py
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        overlap = "+overlap" in mode
        gather = "+gather" in mode
        mode = mode.split("+")[0]
        boundary_boxes, interior_box = shard.overlap_boxes(mode=mode) if overlap else ([], shard.element_box)
        handles = []
        for box in boundary_boxes + [interior_box]:
            g = NonlinearSolid("domain", bench.make_material("neohookean"), pattern, patch=patch, element_box=box).Prepare()
            g.SetStream(stream.cuda_stream)
            handles.append(g)
        assert sum(g.n_elements_ for g in handles) == shard.n_local_elements
        u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
        r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
        A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
        ex = parallel.InterfaceExchange(shard, r, A, dev, mode=mode)
        for _ in range(2):                        # twice: zero_interface must reset the shared rows
            ex.zero_interface(True)
            if gather:
                # two-step assembly: one handle, the rows that leave the rank first, the others while they travel
                early, rest = ex.gather_windows()
                handles[0].Integrate(u)
                for w in early:
                    handles[0].Gather(1.0, r, A, *w)
                ex.start(True)
                handles[0].Gather(1.0, r, A, *rest)
                ex.finish()
            elif len(handles) > 1:
                for g in handles[:-1]:            # the element layers next to the neighbours first ...
                    g.AddDomainResidualAndGrad(u, 1.0, r, A)
                ex.start(True)                    # ... their interface rows travel ...
                handles[-1].AddDomainResidualAndGrad(u, 1.0, r, A)   # ... while the interior is integrated
                ex.finish()
            else:
                handles[0].AddDomainResidualAndGrad(u, 1.0, r, A)
                ex.sum_residual_and_grad()
        for g in handles:
            g.Synchronize()
        Gf = NonlinearSolid("domain", bench.make_material("neohookean"), full, patch=patch).Prepare()
        rf = torch.zeros_like(r)
        Af = torch.zeros(full.nnz, dtype=torch.float64, device=dev)
        Gf.AddDomainResidualAndGrad(u, 1.0, rf, Af)
        Gf.Synchronize()
        # rows of the owned node planes that are interface planes hold one step's sum; interior rows two steps'
        mi_axis = patch.node_multi_index()[shard.axis]
        owned = np.nonzero(np.isin(mi_axis, ex.owned_node_planes()))[0]
        shared_planes = set()
        for nb in (rank - 1, rank + 1):
            if 0 <= nb < world:
                shared_planes.update(shard.interface_node_planes(nb))
        rowptr, rowptr_f = pattern.rowptr.cpu().numpy(), full.rowptr.cpu().numpy()
        col, col_f = pattern.col.cpu().numpy(), full.col.cpu().numpy()
        r_h, A_h, rf_h, Af_h = r.cpu().numpy(), A.cpu().numpy(), rf.cpu().numpy(), Af.cpu().numpy()
        ok = True
        worst = [0.0, 0.0, 0.0, 0.0]   # r / A error on shared planes, on interior planes
        for node in owned:
            shared = mi_axis[node] in shared_planes
            mult = 1.0 if shared else 2.0
            for i in range(3):
                row = node * 3 + i
                s, t = rowptr[row], rowptr[row + 1]
                sf, tf = rowptr_f[row], rowptr_f[row + 1]
                assert np.array_equal(col[s:t], col_f[sf:tf])
                worst[0 if shared else 2] = max(worst[0 if shared else 2], abs(r_h[row] - mult * rf_h[row]))
                worst[1 if shared else 3] = max(worst[1 if shared else 3], np.abs(A_h[s:t] - mult * Af_h[sf:tf]).max())
        scale_r, scale_A = np.abs(rf_h).max(), np.abs(Af_h).max()
        ok = worst[0] < 1e-12 * scale_r and worst[2] < 1e-12 * scale_r and worst[1] < 1e-12 * scale_A and worst[3] < 1e-12 * scale_A
        q.put((rank, bool(ok), len(owned) if ok else repr(worst)))
    except Exception as exc:  # pragma: no cover
        q.put((rank, False, repr(exc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_el,mode", [(2, (4, 6, 3), "owner"), (3, (3, 4, 9), "owner"), (2, (5, 4, 3), "replicate"),
                                             (3, (3, 4, 15), "owner+overlap"), (2, (4, 10, 3), "owner+overlap"),
                                             (3, (3, 4, 15), "owner+overlap+slice"), (2, (4, 10, 3), "owner+slice"),
                                             (3, (3, 4, 15), "owner+gather+slice"), (2, (4, 10, 3), "replicate+gather"),
                                             (2, (3, 4, 6), "replicate+slice")])
def test_slabs_on_one_gpu(world, n_el, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_el, mode, q)) for r in range(world)]
    results = _run_ranks(ctx, procs, q, world)
    assert all(ok is True for _, ok, _ in results), results
    if mode.startswith("owner"):
        assert sum(n for _, _, n in results) == int(np.prod([n + 2 for n in n_el]))


def _local_worker(rank, world, port, n_el, p, scheme, q):
    """the LOCAL layout (round 5, parallel.SlabShard.localized): every rank holds its slab (+ p ghost layers either side) as a
    patch of its own -- u, r of local length, the local patch's whole structured matrix, set-up of local size."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        import mimi_amd
        from mimi_amd import parallel
        from mimi_amd.integrators import CSRPattern, NonlinearSolid
        from mimi_amd.splines import PatchShape
        dev = torch.device("cuda", 0)
        shape = PatchShape.block(n_el, p)
        shard_g = parallel.SlabShard(shape, None, rank, world)
        b, e = shard_g.element_box
        ax = shard_g.axis
        below, above = shard_g.ghost_layers()
        lp = mimi_amd.BSplinePatch.block_slab(n_el, p, ax, b[ax] - below, e[ax] + above)
        pattern = CSRPattern.of_bspline_patch(lp, on_device=True)
        shard = shard_g.localized(lp, pattern, ghost=(below, above))
        gn = shard.global_nodes()
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        mat = bench.make_material("neohookean")
        g = NonlinearSolid("domain", mat, pattern, patch=lp, element_box=shard.element_box).Prepare()
        g.SetStream(stream.cuda_stream)
        assert g.n_elements_ == shard_g.n_local_elements and g.path_ == 1
        # the whole patch on the same GPU: the checker, and the source of u
        patch = mimi_amd.BSplinePatch.block(n_el, p)
        full = CSRPattern.of_bspline_patch(patch, on_device=True)
        u_g = bench.synthetic_u(patch)
        gdofs = (gn[:, None] * 3 + np.arange(3)[None, :]).ravel()
        u = torch.from_numpy(np.ascontiguousarray(u_g[gdofs])).to(dev)
        r = torch.zeros(lp.n_vdofs, dtype=torch.float64, device=dev)
        A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
        assert lp.n_vdofs <= patch.n_vdofs and pattern.nnz <= full.nnz     # (equal when slab + ghost layers are the whole patch)
        # (1) before any exchange: the same bits as a slab handle on the whole patch (same tables: the knot slice has the
        # whole knot vector's values; same kernels) on every row of the slab's nodes
        g.AddDomainResidualAndGrad(u, 1.0, r, A)
        g.Synchronize()
        Gs = NonlinearSolid("slab", mat, full, patch=patch, element_box=shard_g.element_box).Prepare()
        r_s = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
        A_s = torch.zeros(full.nnz, dtype=torch.float64, device=dev)
        Gs.AddDomainResidualAndGrad(torch.from_numpy(u_g).to(dev), 1.0, r_s, A_s)
        Gs.Synchronize()
        rowptr, rowptr_f = pattern.rowptr.cpu().numpy(), full.rowptr.cpu().numpy()
        r_h, A_h, rs_h, As_h = r.cpu().numpy(), A.cpu().numpy(), r_s.cpu().numpy(), A_s.cpu().numpy()
        nb, ne = shard.node_box()
        mi = lp.node_multi_index()
        touched = np.nonzero((mi[ax] >= nb[ax]) & (mi[ax] < ne[ax]))[0]
        same_bits = True
        for node in touched:
            for i in range(3):
                lrow, grow = node * 3 + i, gn[node] * 3 + i
                ls, lt, s, t = rowptr[lrow], rowptr[lrow + 1], rowptr_f[grow], rowptr_f[grow + 1]
                same_bits = same_bits and lt - ls == t - s and np.array_equal(A_h[ls:lt], As_h[s:t]) and r_h[lrow] == rs_h[grow]
        # (2) the exchange in local coordinates, twice (zero_interface must reset the shared rows)
        ex = parallel.InterfaceExchange(shard, r, A, dev, mode="owner")
        r.zero_()
        A.zero_()
        for _ in range(2):
            ex.zero_interface(True)
            if scheme == "gather":
                early, rest = ex.gather_windows()
                g.Integrate(u)
                for w in early:
                    g.Gather(1.0, r, A, *w)
                ex.start(True)
                g.Gather(1.0, r, A, *rest)
                ex.finish()
            else:
                g.AddDomainResidualAndGrad(u, 1.0, r, A)
                ex.sum_residual_and_grad()
        g.Synchronize()
        stream.synchronize()
        Gf = NonlinearSolid("whole", mat, full, patch=patch).Prepare()
        rf = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
        Af = torch.zeros(full.nnz, dtype=torch.float64, device=dev)
        Gf.AddDomainResidualAndGrad(torch.from_numpy(u_g).to(dev), 1.0, rf, Af)
        Gf.Synchronize()
        r_h, A_h, rf_h, Af_h = r.cpu().numpy(), A.cpu().numpy(), rf.cpu().numpy(), Af.cpu().numpy()
        owned = np.nonzero(np.isin(mi[ax], ex.owned_node_planes()))[0]
        shared_planes = set()
        for nbr in (rank - 1, rank + 1):
            if 0 <= nbr < world:
                shared_planes.update(shard.interface_node_planes(nbr))
        worst_r = worst_A = 0.0
        for node in owned:
            mult = 1.0 if mi[ax][node] in shared_planes else 2.0
            for i in range(3):
                lrow, grow = node * 3 + i, gn[node] * 3 + i
                ls, lt, s, t = rowptr[lrow], rowptr[lrow + 1], rowptr_f[grow], rowptr_f[grow + 1]
                assert lt - ls == t - s          # an owned row is complete in its columns
                worst_r = max(worst_r, abs(r_h[lrow] - mult * rf_h[grow]))
                worst_A = max(worst_A, np.abs(A_h[ls:lt] - mult * Af_h[s:t]).max())
        ok = same_bits and worst_r < 1e-12 * np.abs(rf_h).max() and worst_A < 1e-12 * np.abs(Af_h).max()
        q.put((rank, bool(ok), len(owned) if ok else repr((same_bits, worst_r, worst_A))))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_el,p,scheme", [(2, (4, 6, 3), 2, "plain"), (3, (3, 4, 15), 2, "gather"), (2, (3, 8, 2), 3, "gather"),
                                                 (3, (2, 3, 9), 3, "plain")])
def test_local_layout_slabs_on_one_gpu(world, n_el, p, scheme):
    """per-rank vectors and matrix of local size (the slab and its halo as a patch of its own) with
    the HIP kernels, degree 2 and 3: the same bits as a slab handle on the whole patch before the exchange, and after it
    every owned row -- complete in its columns -- equals the whole-patch assembly."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_local_worker, args=(r, world, port, n_el, p, scheme, q)) for r in range(world)]
    results = _run_ranks(ctx, procs, q, world)
    assert all(ok is True for _, ok, _ in results), results
    assert sum(n for _, _, n in results) == int(np.prod([n + p for n in n_el]))


def _contact_worker(rank, world, port, n_el, sliced, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        import mimi_amd
        from mimi_amd import parallel
        from mimi_amd.integrators import CSRPattern, MortarContact, RigidSphere
        dev = torch.device("cuda", 0)
        patch = mimi_amd.BSplinePatch.block(n_el, 2)
        full = CSRPattern.of_bspline_patch(patch, on_device=True)
        shard = parallel.SlabShard(patch, None, rank, world)
        pattern = CSRPattern.of_bspline_patch(patch, on_device=True, node_box=shard.node_box()) if sliced else full
        shard.pattern = pattern
        L = patch.control_points.max(axis=0)
        R = 0.25 * L[0]
        centre = 0.5 * L
        centre[2] = L[2] + 0.9 * R                                   # sphere over the top face
        body = RigidSphere(list(centre), R, 1e4)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        sc = parallel.ShardedContact(shard, body, pattern, 2, 1)
        sc.SetStream(stream.cuda_stream)
        u = torch.from_numpy(bench.synthetic_u(patch, scale=0.01)).to(dev)
        r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
        A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
        ex = parallel.InterfaceExchange(shard, r, A, dev, mode="replicate")
        sc.AddBoundaryResidualAndGrad(u, 0.7, r, A)
        if sc.contact is not None:
            sc.contact.Synchronize()
        stream.synchronize()
        ex.sum_residual_and_grad()
        stream.synchronize()
        # the whole face on one handle
        Gf = MortarContact(body, "contact", full, patch, 2, 1).Prepare()
        rf, Af = torch.zeros_like(r), torch.zeros(full.nnz, dtype=torch.float64, device=dev)
        Gf.AddBoundaryResidualAndGrad(u, 0.7, rf, Af)
        Gf.Synchronize()
        # after the replicate exchange this rank holds the complete rows of every node its elements touch
        b, e = shard.element_box
        mi_axis = patch.node_multi_index()[shard.axis]
        mine = np.nonzero((mi_axis >= b[shard.axis]) & (mi_axis < e[shard.axis] + 2))[0]
        rowptr, rowptr_f = pattern.rowptr.cpu().numpy(), full.rowptr.cpu().numpy()
        r_h, A_h, rf_h, Af_h = r.cpu().numpy(), A.cpu().numpy(), rf.cpu().numpy(), Af.cpu().numpy()
        er = eA = 0.0
        for node in mine:
            for i in range(3):
                row = node * 3 + i
                s, t = rowptr[row], rowptr[row + 1]
                sf, tf = rowptr_f[row], rowptr_f[row + 1]
                er = max(er, abs(r_h[row] - rf_h[row]))
                eA = max(eA, np.abs(A_h[s:t] - Af_h[sf:tf]).max())
        ok = np.abs(rf_h).max() > 0 and er < 1e-11 * np.abs(rf_h).max() and eA < 1e-11 * np.abs(Af_h).max()
        q.put((rank, bool(ok), (er, eA, float(np.abs(rf_h).max()))))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_el,sliced", [(2, (4, 6, 2), False), (3, (3, 9, 2), False), (2, (4, 6, 2), True)])
def test_sharded_contact_on_one_gpu(world, n_el, sliced):
    """cfg4 in miniature: contact faces follow their element slab; the nodal area / gap of the nodes shared between
    slabs are summed over the ranks before the pressure is formed; rows then travel with the interface exchange."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_contact_worker, args=(r, world, port, n_el, sliced, q)) for r in range(world)]
    results = _run_ranks(ctx, procs, q, world)
    assert all(ok is True for _, ok, _ in results), results


def test_row_slice_must_cover_the_handles_nodes():
    """a handle whose elements touch a node the sliced pattern does not hold is refused at create time"""
    import mimi_amd
    import bench
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    patch = mimi_amd.BSplinePatch.block((3, 3, 6), 2)
    full = CSRPattern.of_bspline_patch(patch, on_device=True)
    part = CSRPattern.of_bspline_patch(patch, on_device=True, node_box=([0, 0, 0], [5, 5, 5]))   # element layers 0..2 in z
    assert 0 < part.nnz < full.nnz
    lens_f = (full.rowptr[1:] - full.rowptr[:-1]).cpu().numpy()
    lens_p = (part.rowptr[1:] - part.rowptr[:-1]).cpu().numpy()
    inside = np.repeat(patch.node_multi_index()[2] < 5, 3)
    assert np.array_equal(lens_p[inside], lens_f[inside]) and not lens_p[~inside].any()
    ok = NonlinearSolid("domain", bench.make_material("neohookean"), part, patch=patch, element_box=([0, 0, 0], [3, 3, 3])).Prepare()
    assert ok.path_ == 1
    with pytest.raises(RuntimeError):
        NonlinearSolid("domain", bench.make_material("neohookean"), part, patch=patch, element_box=([0, 0, 0], [3, 3, 4])).Prepare()
