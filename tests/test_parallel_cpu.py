"""N > 1 path on CPU: world_size-2 (and 3) `gloo` processes, each holding the assembly of its
element slab (produced here by the oracle, which is only the data source / checker), run the
product's interface exchange; every rank must end up with the fully assembled rows of all
nodes its slab touches."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_el, p, mode, q, sliced=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mimi_amd
        from mimi_amd import parallel
        from mimi_amd.integrators import CSRPattern
        from oracle import iga, ref_path as rp
        from _cases import oracle_material, synthetic_u
        P = iga.Patch.block(n_el, p)
        rowptr, col = P.sparsity()
        patch = mimi_amd.BSplinePatch.block(n_el, p)
        pattern = CSRPattern(rowptr, col, rowptr[-1])
        shard = parallel.SlabShard(patch, pattern, rank, world)
        (b, e) = shard.element_box
        em = P.element_multi_index()
        sel = np.ones(P.n_el, dtype=bool)
        for d in range(P.dim):
            sel &= (em[d] >= b[d]) & (em[d] < e[d])
        elements = np.nonzero(sel)[0]
        assert len(elements) == shard.n_local_elements
        u = synthetic_u(P)
        D = rp.DomainOracle(P, oracle_material("neohook"), elements=elements)
        r = np.zeros(P.n_vdofs)
        A = np.zeros(D.nnz)
        D.add_domain_residual_and_grad(u, 1.0, r, A, rp.TANGENT_EXACT)
        lrowptr = rowptr
        if sliced:
            # this rank's value array holds only the rows of the nodes its slab touches (SlabShard.node_box): the
            # pattern keeps its length, the other rows are empty
            nb, ne = shard.node_box()
            mi = patch.node_multi_index()
            inside = np.ones(P.n_nodes, dtype=bool)
            for d in range(P.dim):
                inside &= (mi[d] >= nb[d]) & (mi[d] < ne[d])
            keep = np.repeat(inside, P.dim)
            lengths = np.diff(rowptr)
            lrowptr = np.concatenate([[0], np.cumsum(lengths * keep)]).astype(np.int64)
            entries = np.repeat(keep, lengths)
            assert np.all(A[~entries] == 0.0)          # the slab's elements write nothing outside the slice
            A = np.ascontiguousarray(A[entries])
            shard.pattern = CSRPattern(lrowptr, np.ascontiguousarray(col[entries]), lrowptr[-1])
            assert 0 < lrowptr[-1] < rowptr[-1]
        tr, tA = torch.from_numpy(r), torch.from_numpy(A)
        ex = parallel.InterfaceExchange(shard, tr, tA, mode=mode)
        ex.sum_residual_and_grad()
        # full assembly for comparison
        Dfull = rp.DomainOracle(P, oracle_material("neohook"))
        rf = np.zeros(P.n_vdofs)
        Af = np.zeros(D.nnz)
        Dfull.add_domain_residual_and_grad(u, 1.0, rf, Af, rp.TANGENT_EXACT)
        # "replicate": all nodes the slab touches; "owner": the nodes of the planes this rank owns
        mi_axis = patch.node_multi_index()[shard.axis]
        touched = np.nonzero(np.isin(mi_axis, ex.owned_node_planes()))[0]
        if mode == "replicate":
            assert np.array_equal(touched, np.unique(D.conn))
        rows = (touched[:, None] * P.dim + np.arange(P.dim)[None, :]).ravel()
        ok = np.allclose(r[rows], rf[rows], rtol=1e-12, atol=1e-12)
        for row in rows:
            s, t = rowptr[row], rowptr[row + 1]
            ls, lt = lrowptr[row], lrowptr[row + 1]
            ok = ok and lt - ls == t - s and np.allclose(A[ls:lt], Af[s:t], rtol=1e-12, atol=1e-10)
        q.put((rank, bool(ok), len(elements), len(touched)))
    except Exception as exc:  # pragma: no cover
        q.put((rank, False, repr(exc), 0))
    finally:
        dist.destroy_process_group()


def _worker_local(rank, world, port, n_el, p, mode, q):
    """the LOCAL layout (round 5; parallel.SlabShard.localized): every rank holds a patch that consists of its slab alone --
    u, r of local length, the whole structured matrix of the local patch -- and the interface exchange runs in local
    coordinates.  The oracle integrates the LOCAL patch (the data source) and, as the checker, the whole one."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mimi_amd
        from mimi_amd import parallel
        from mimi_amd.integrators import CSRPattern
        from mimi_amd.splines import PatchShape
        from oracle import iga, ref_path as rp
        from _cases import oracle_material, synthetic_u
        shard_g = parallel.SlabShard(PatchShape.block(n_el, p), None, rank, world)
        b, e = shard_g.element_box
        ax = shard_g.axis
        below, above = shard_g.ghost_layers()
        lp = mimi_amd.BSplinePatch.block_slab(n_el, p, ax, b[ax] - below, e[ax] + above)
        P_loc = iga.Patch(lp.degrees, lp.knots, lp.control_points)         # the oracle on the local patch
        assert P_loc.n_nodes == lp.n_nodes
        rowptr, col = P_loc.sparsity()
        shard = shard_g.localized(lp, CSRPattern(rowptr, col, rowptr[-1]), ghost=(below, above))
        em = P_loc.element_multi_index()
        own = np.nonzero((em[ax] >= shard.element_box[0][ax]) & (em[ax] < shard.element_box[1][ax]))[0]
        assert len(own) == shard_g.n_local_elements
        gn = shard.global_nodes()
        P = iga.Patch.block(n_el, p)                                        # (the checker's whole patch)
        assert np.allclose(P.ctrl[gn], lp.control_points, atol=1e-14, rtol=0)
        u = synthetic_u(P)
        dim = P.dim
        gdofs = (gn[:, None] * dim + np.arange(dim)[None, :]).ravel()
        u_loc = np.ascontiguousarray(u[gdofs])
        D = rp.DomainOracle(P_loc, oracle_material("neohook"), elements=own)        # (the ghost layers are the neighbours')
        r = np.zeros(P_loc.n_vdofs)
        A = np.zeros(D.nnz)
        D.add_domain_residual_and_grad(u_loc, 1.0, r, A, rp.TANGENT_EXACT)
        tr, tA = torch.from_numpy(r), torch.from_numpy(A)
        ex = parallel.InterfaceExchange(shard, tr, tA, mode=mode)
        ex.sum_residual_and_grad()
        Dfull = rp.DomainOracle(P, oracle_material("neohook"))
        rf = np.zeros(P.n_vdofs)
        Af = np.zeros(Dfull.nnz)
        Dfull.add_domain_residual_and_grad(u, 1.0, rf, Af, rp.TANGENT_EXACT)
        frowptr, fcol = P.sparsity()
        # rows this rank owns after the exchange, local -> global
        mi_axis = lp.node_multi_index()[ax]
        owned = np.nonzero(np.isin(mi_axis, ex.owned_node_planes()))[0]
        ok = True
        local_of = -np.ones(P.n_nodes, dtype=np.int64)
        local_of[gn] = np.arange(len(gn))
        for node in owned:
            for i in range(dim):
                lrow, grow = node * dim + i, gn[node] * dim + i
                ok = ok and np.isclose(r[lrow], rf[grow], rtol=1e-12, atol=1e-12)
                ls, lt = rowptr[lrow], rowptr[lrow + 1]
                s, t = frowptr[grow], frowptr[grow + 1]
                # an owned row is COMPLETE on this rank, columns included: the ghost layers make room for them
                gc = fcol[s:t]
                lc = local_of[gc // dim] * dim + gc % dim
                ok = ok and np.all(local_of[gc // dim] >= 0)
                ok = ok and np.array_equal(lc, col[ls:lt]) and np.allclose(A[ls:lt], Af[s:t], rtol=1e-12, atol=1e-10)
        q.put((rank, bool(ok), len(own), len(owned)))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, False, repr(exc) + traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(world, n_el, p, mode, sliced, local=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    if local:
        procs = [ctx.Process(target=_worker_local, args=(r, world, port, n_el, p, mode, q)) for r in range(world)]
    else:
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_el, p, mode, q, sliced)) for r in range(world)]
    for pr in procs:
        pr.start()
    try:
        results = [q.get(timeout=240) for _ in range(world)]
        for pr in procs:
            pr.join(timeout=60)
    finally:
        for pr in procs:            # whatever happened, no rank is left waiting in a receive
            if pr.is_alive():
                pr.terminate()
                pr.join(timeout=10)
    assert all(ok is True for _, ok, _, _ in results), results
    assert sum(n for _, _, n, _ in results) == int(np.prod(n_el))
    if mode == "owner":
        # the owned planes partition the nodes
        assert sum(n for _, _, _, n in results) == int(np.prod([n_el[d] + p for d in range(len(n_el))]))


@pytest.mark.parametrize("mode", ["replicate", "owner"])
@pytest.mark.parametrize("world,n_el,p", [(2, (3, 2, 6), 2), (3, (2, 9), 3), (2, (4, 5, 3), 1)])
def test_interface_exchange_gloo(world, n_el, p, mode):
    _run(world, n_el, p, mode, False)


@pytest.mark.parametrize("world,n_el,p,mode", [(2, (3, 2, 6), 2, "owner"), (3, (2, 9), 3, "replicate")])
def test_interface_exchange_gloo_row_slices(world, n_el, p, mode):
    """each rank holds only its row slice of the matrix (what bench.py does for N > 1)"""
    _run(world, n_el, p, mode, True)


@pytest.mark.parametrize("world,n_el,p,mode", [(2, (3, 2, 6), 2, "owner"), (3, (2, 9), 3, "owner"), (3, (3, 7, 2), 2, "replicate"),
                                               (2, (4, 5, 3), 1, "owner")])
def test_interface_exchange_gloo_local_layout(world, n_el, p, mode):
    """every rank holds its slab as a patch of its own: vectors of local length, the local patch's whole matrix
    (what bench.py does for N > 1 since round 5)"""
    _run(world, n_el, p, mode, False, local=True)


def test_local_slab_patch_has_the_whole_patch_tables_and_points():
    """BSplinePatch.block_slab: the knot slice of a slab gives the whole block's basis functions on those layers (the oracle's
    1-D tables of the local patch are the rows of the whole patch's tables, to the bit) and the control points of its
    node planes; LocalSlabShard maps local nodes to global ones."""
    import mimi_amd
    from mimi_amd import parallel
    from mimi_amd.splines import PatchShape
    from oracle import iga
    n_el, p = (3, 7, 2), 3
    whole = mimi_amd.BSplinePatch.block(n_el, p)
    Pw = iga.Patch.block(n_el, p)
    Bw, Dw, Ww, _ = Pw.tables_1d()
    seen = np.zeros(whole.n_nodes, dtype=int)
    for rank in range(2):
        g = parallel.SlabShard(PatchShape.block(n_el, p), None, rank, 2)
        b, e = g.element_box
        assert g.ghost_layers() == ((0, 3) if rank == 0 else (3, 0))
        lp = mimi_amd.BSplinePatch.block_slab(n_el, p, g.axis, b[g.axis], e[g.axis])
        loc = g.localized(lp)
        gn = loc.global_nodes()
        assert np.array_equal(lp.control_points, whole.control_points[gn])
        seen[gn] += 1
        Pl = iga.Patch(lp.degrees, lp.knots, lp.control_points)
        Bl, Dl, Wl, _ = Pl.tables_1d()
        for d in range(3):
            sl = slice(b[d], e[d]) if d == g.axis else slice(None)
            # (the oracle's recursion is not the library's: rounding-level agreement here; the library's own tables are
            # held to the bit through the assembled values, tests/test_parallel_gpu.py)
            assert np.allclose(Bl[d], Bw[d][sl], rtol=0, atol=1e-14) and np.allclose(Dl[d], Dw[d][sl], rtol=0, atol=1e-13)
        assert loc.interface_node_planes(1 - rank) == ([e[g.axis] - b[g.axis] + k for k in range(p)] if rank == 0 else list(range(p)))
    assert seen.min() == 1 and seen.max() == 2 and (seen == 2).sum() == p * whole.n_ctrl[0] * whole.n_ctrl[2]


def test_slab_shard_boxes():
    import mimi_amd
    from mimi_amd import parallel
    patch = mimi_amd.BSplinePatch.block((128, 128, 16), 2)
    boxes = [parallel.SlabShard(patch, None, r, 8).element_box for r in range(8)]
    assert boxes[0] == ([0, 0, 0], [128, 16, 16]) and boxes[7] == ([0, 112, 0], [128, 128, 16])
    with pytest.raises(RuntimeError):
        parallel.SlabShard(mimi_amd.BSplinePatch.block((2, 2, 3), 2), None, 0, 3)


def test_contact_faces_follow_their_element_slab():
    """the faces of a patch face split between the slabs without overlap or loss (mimi_amd/splines.py face_tables)"""
    import mimi_amd
    from mimi_amd import parallel, splines
    from mimi_amd.integrators import CSRPattern
    patch = mimi_amd.BSplinePatch.block((3, 7, 2), 2)
    full = splines.face_tables(patch, 2, 1)
    pattern = CSRPattern(np.zeros(patch.n_vdofs + 1, dtype=np.int64), np.zeros(0, dtype=np.int32), 0)
    rows = []
    for rank in range(3):
        shard = parallel.SlabShard(patch, pattern, rank, 3)
        part = splines.face_tables(patch, 2, 1, element_box=shard.element_box)
        rows.append(part[0])
        assert part[1].shape[0] == part[0].shape[0] == part[2].shape[0] == part[3].shape[0]
    got = np.concatenate(rows)
    assert got.shape == full[0].shape
    assert sorted(map(tuple, got)) == sorted(map(tuple, full[0]))
    # a face on the far side of the slab axis belongs to the last slab only
    patch2 = mimi_amd.BSplinePatch.block((3, 2, 6), 2)
    n = [len(splines.face_tables(patch2, 2, 1, element_box=parallel.SlabShard(patch2, pattern, r, 2).element_box)[0]) for r in range(2)]
    assert n == [0, 6]


def test_boundary_boxes_that_may_run_concurrently():
    """two boundary boxes may assemble concurrently only when no node is touched by both"""
    import mimi_amd
    from mimi_amd import parallel
    patch = mimi_amd.BSplinePatch.block((3, 3, 18), 2)
    thick = parallel.SlabShard(patch, None, 1, 3)                   # 6 layers: boundary 2 + 2, interior 2
    boundary, interior = thick.overlap_boxes()
    assert len(boundary) == 2 and thick.boxes_share_no_node(boundary)
    assert not thick.boxes_share_no_node(boundary + [interior])     # the interior shares node planes with both
    thin = parallel.SlabShard(mimi_amd.BSplinePatch.block((3, 3, 15), 2), None, 1, 3)   # 5 layers: 2 + 1 + 2
    boundary, _ = thin.overlap_boxes()
    assert len(boundary) == 2 and not thin.boxes_share_no_node(boundary)


def test_owner_mode_needs_fewer_boundary_layers():
    """owner mode sends upward only the planes the upper rank owns (touched by the last p - p // 2 element layers) and
    downward only those the lower rank owns (first p // 2 layers); replicate sends all p shared planes (p layers each)"""
    import mimi_amd
    from mimi_amd import parallel
    for p, lower, upper in ((1, 0, 1), (2, 1, 1), (3, 1, 2)):
        patch = mimi_amd.BSplinePatch.block((3, 3, 30), p)
        sh = parallel.SlabShard(patch, None, 1, 3)              # layers 10..19
        boundary, interior = sh.overlap_boxes(mode="owner")
        sizes = sorted((b[2], e[2]) for b, e in boundary)
        expect = ([(10, 10 + lower)] if lower else []) + [(20 - upper, 20)]
        assert sizes == expect and (interior[0][2], interior[1][2]) == (10 + lower, 20 - upper)
        # the planes that leave upward, [20 + p // 2, 20 + p), are touched by the upper boundary box only
        first_plane_sent_up = 20 + p // 2
        assert interior[1][2] - 1 + p < first_plane_sent_up          # last interior layer's highest node plane
        # ... and those that leave downward, [10, 10 + p // 2), by the lower one only
        assert lower == 0 or interior[0][2] >= 10 + p // 2
        boundary_r, interior_r = sh.overlap_boxes(mode="replicate")
        assert sorted((b[2], e[2]) for b, e in boundary_r) == [(10, 10 + p), (20 - p, 20)]


def test_gather_windows_partition_the_slab_nodes():
    """two-step assembly: the planes that leave the rank first, the rest afterwards; together every node plane of the slab once"""
    import torch
    import mimi_amd
    from mimi_amd import parallel
    from mimi_amd.integrators import CSRPattern
    from oracle import iga
    for p, mode, rank, world in ((2, "owner", 1, 3), (2, "replicate", 0, 2), (3, "owner", 2, 3), (1, "owner", 1, 3)):
        n_el = (2, 2, 18)
        P = iga.Patch.block(n_el, p)
        rowptr, col = P.sparsity()
        patch = mimi_amd.BSplinePatch.block(n_el, p)
        sh = parallel.SlabShard(patch, CSRPattern(rowptr, col, rowptr[-1]), rank, world)
        ex = parallel.InterfaceExchange.__new__(parallel.InterfaceExchange)      # (no communicator needed for the plan)
        ex.shard, ex.mode = sh, mode
        early, rest = ex.gather_windows()
        nb, ne = sh.node_box()
        planes = sorted(x for b, e in early + [rest] for x in range(b[2], e[2]))
        assert planes == list(range(nb[2], ne[2]))
        for b, e in early + [rest]:
            assert b[:2] == nb[:2] and e[:2] == ne[:2]
        b0, e0 = int(sh.starts[rank]), int(sh.starts[rank + 1])
        sent = []
        if rank > 0:
            sent += list(range(b0, b0 + (p if mode == "replicate" else p // 2)))
        if rank < world - 1:
            sent += list(range(e0 + (0 if mode == "replicate" else p // 2), e0 + p))
        assert sorted(x for b, e in early for x in range(b[2], e[2])) == sent


@pytest.mark.parametrize("p,mode,expect", [(2, "owner", [(0, 0), (1, 1)]), (3, "replicate", [(0, 0), (1, 1)]),
                                           (3, "owner", [(0, 1), (1, 0)])])
def test_loopback_pairs_sends_with_receives_of_their_own_length(p, mode, expect, monkeypatch):
    """in loop-back (a middle rank talking to itself, bench.py --rehearse-rccl) owner mode at odd degree
    sends 1 plane and receives 2 on one side and the reverse on the other: a send must meet the OTHER side's receive."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import mimi_amd
    from mimi_amd import parallel
    from mimi_amd.integrators import CSRPattern
    from oracle import iga
    monkeypatch.setattr(dist, "get_rank", lambda *a, **k: 0)
    n_el = (2, 2, 18)
    P = iga.Patch.block(n_el, p)
    rowptr, col = P.sparsity()
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    shard = parallel.SlabShard(patch, CSRPattern(rowptr, col, rowptr[-1]), 1, 3)
    ex = parallel.InterfaceExchange(shard, torch.zeros(P.n_vdofs, dtype=torch.float64),
                                    torch.zeros(int(rowptr[-1]), dtype=torch.float64), mode=mode, loopback=True)
    pairs = ex.loopback_pairs()
    assert pairs == expect
    for i, j in pairs:
        assert ex.sides[i]["srows"].numel() == ex.sides[j]["rrows"].numel()
        assert ex.sides[i]["sidx"].numel() == ex.sides[j]["ridx"].numel()
    # an end rank in owner mode at odd degree has one side whose lengths differ: nothing to pair it with
    if (p, mode) == (3, "owner"):
        end = parallel.SlabShard(patch, CSRPattern(rowptr, col, rowptr[-1]), 0, 3)
        ex0 = parallel.InterfaceExchange(end, torch.zeros(P.n_vdofs, dtype=torch.float64),
                                         torch.zeros(int(rowptr[-1]), dtype=torch.float64), mode=mode, loopback=True)
        with pytest.raises(RuntimeError, match="cannot be paired"):
            ex0.loopback_pairs()


@pytest.mark.parametrize("p,mode", [(2, "owner"), (2, "replicate"), (3, "owner")])
def test_trimmed_messages_drop_structural_zeros_only(p, mode, monkeypatch):
    """round 5: a message carries of every shared row only the entries the sender's element layers can have written
    (InterfaceExchange(trim=True), the default).  On an oracle-integrated slab: every entry of the whole-row message that the
    trimmed one leaves out IS zero, the trimmed one is the expected fraction of it (degree 2, owner mode: 3 of the 5 column
    planes of a row), and the receiving side lists as many entries as the neighbour sends."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import mimi_amd
    from mimi_amd import parallel
    from mimi_amd.integrators import CSRPattern
    from oracle import iga, ref_path as rp
    from _cases import oracle_material, synthetic_u
    monkeypatch.setattr(dist, "get_rank", lambda *a, **k: 0)
    n_el = (2, 2, 12)
    P = iga.Patch.block(n_el, p)
    rowptr, col = P.sparsity()
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    exs = {}
    for rank in (1, 2):
        shard = parallel.SlabShard(patch, CSRPattern(rowptr, col, rowptr[-1]), rank, 3)
        em = P.element_multi_index()
        b, e = shard.element_box
        own = np.nonzero((em[shard.axis] >= b[shard.axis]) & (em[shard.axis] < e[shard.axis]))[0]
        D = rp.DomainOracle(P, oracle_material("neohook"), elements=own)
        r, A = np.zeros(P.n_vdofs), np.zeros(D.nnz)
        D.add_domain_residual_and_grad(synthetic_u(P), 1.0, r, A, rp.TANGENT_EXACT)
        tr, tA = torch.from_numpy(r), torch.from_numpy(A)
        full = parallel.InterfaceExchange(shard, tr, tA, mode=mode, loopback=True, trim=False)
        trimmed = parallel.InterfaceExchange(shard, tr, tA, mode=mode, loopback=True)
        assert trimmed.trim and not full.trim
        for sf, st in zip(full.sides, trimmed.sides):
            assert torch.equal(sf["srows"], st["srows"]) and torch.equal(sf["rrows"], st["rrows"])
            kept = torch.zeros(tA.numel(), dtype=torch.bool)
            kept[st["sidx"]] = True
            assert bool(kept[sf["sidx"]].sum() == st["sidx"].numel())            # a subset of the whole rows, ...
            dropped = sf["sidx"][~kept[sf["sidx"]]]
            assert dropped.numel() > 0 and bool((tA[dropped] == 0.0).all())        # ... what it leaves out is zero, ...
            assert bool((tA[st["sidx"]] != 0.0).float().mean() > 0.9)             # ... and what it keeps (mostly) is not
            if (p, mode) == (2, "owner"):
                assert st["sidx"].numel() * 5 == sf["sidx"].numel() * 3
        exs[rank] = trimmed
    # rank 1's upper side talks to rank 2's lower side: what one lists to send the other lists to receive
    up, down = exs[1].sides[1], exs[2].sides[0]
    assert up["sidx"].numel() == down["ridx"].numel() and up["ridx"].numel() == down["sidx"].numel()
    assert up["srows"].numel() == down["rrows"].numel()
    # (both ranks index the whole patch's pattern here: the same pairs in the same order are the same positions)
    assert torch.equal(up["sidx"], down["ridx"]) and torch.equal(up["ridx"], down["sidx"])
