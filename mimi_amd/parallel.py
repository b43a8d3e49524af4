"""Element sharding of one patch across the GPUs of a node (one process per GPU) and the
one exchange step of an assembly: the sum of the shared-dof rows of the residual / Jacobian
between neighbouring slabs.

The reference has no distributed path at all; its only "collective" is the in-process
reduction of per-thread global arrays (integrators/nonlinear_base.hpp:90-151).  Here each rank
integrates a contiguous slab of elements into its own CSR value array; the rows of the `p`
node layers shared with each neighbour are then summed pairwise with RCCL send/recv
(`torch.distributed` backend "nccl"; a slab has at most two neighbours, so each exchange rides
one xGMI link instead of a ring over all ranks).  After the exchange every rank holds the
fully assembled rows of every node its elements touch.
"""
import numpy as np


class SlabShard:
    """Contiguous element slabs along one axis (the longest; ties -> the slowest-varying
    one, whose node planes are contiguous in the lexicographic numbering).

    `pattern`: the CSR pattern this rank assembles into -- the whole patch's, or the row slice of node_box() (then
    the value array holds only this rank's rows; pass None here and set `.pattern` once node_box() is known)."""

    def __init__(self, patch, pattern, rank, world_size, axis=None):
        self.patch, self.pattern = patch, pattern
        self.rank, self.world_size = rank, world_size
        spans = np.asarray(patch.n_spans)
        if axis is None:
            axis = int(np.flatnonzero(spans == spans.max())[-1])
        self.axis = axis
        m = int(spans[axis])
        base, extra = divmod(m, world_size)
        counts = [base + (1 if k < extra else 0) for k in range(world_size)]
        starts = np.concatenate([[0], np.cumsum(counts)])
        self.starts = starts
        p = patch.degrees[axis]
        if world_size > 1 and min(counts) < p:
            raise RuntimeError(f"slabs thinner than the degree ({min(counts)} < {p}): more than two ranks would share a node")
        begin = [0, 0, 0]
        end = [1, 1, 1]
        for d in range(patch.dim):
            end[d] = int(spans[d])
        begin[axis], end[axis] = int(starts[rank]), int(starts[rank + 1])
        self.element_box = (begin, end)
        self.n_local_elements = int(np.prod([end[d] - begin[d] for d in range(patch.dim)]))

    def ghost_layers(self):
        """(below, above): element layers of the neighbours a localized slab carries along -- p on either side (none at an end
        of the patch).  The rows of node plane n hold columns of the planes n - p .. n + p; the rows that travel are those
        of the shared planes [e, e + p) between the slabs [b, e) and [e, ...): with p ghost layers the local patches of
        BOTH sides contain every column of every shared row, in the same (lexicographic) order -- so a row is the same run
        of values on either side and the exchange packs / unpacks whole rows (csrc/exchange.hip) exactly as with global
        row slices -- and the rows a rank owns after the exchange are complete in their columns."""
        p = self.patch.degrees[self.axis]
        b, e = self.element_box[0][self.axis], self.element_box[1][self.axis]
        total = int(self.patch.n_spans[self.axis])
        return (min(p, b) if self.rank > 0 else 0), (min(p, total - e) if self.rank < self.world_size - 1 else 0)

    def localized(self, local_patch, pattern=None, ghost=(0, 0)):
        """This rank's slab as a patch of its OWN (round 5): `local_patch` holds the slab's element layers and `ghost`
        = ghost_layers() more on either side (BSplinePatch.block_slab(n_el, p, axis, b - below, e + above), or any patch cut
        the same way); the handle integrates the slab's own layers (element_box of the returned shard).  Vectors, CSR rows
        and columns, material state and the handle's set-up are then of LOCAL size -- u and r over the node planes of the
        slab and its halo, the matrix = the structured pattern of the local patch -- instead of whole-patch vectors and a row
        slice of the whole pattern.  Returns the shard in local coordinates: InterfaceExchange, gather_windows and
        overlap_boxes work on it unchanged; global_nodes() maps back."""
        return LocalSlabShard(self, local_patch, pattern, ghost)

    def node_box(self):
        """(begin, end) of the nodes this slab's elements touch: the rows a rank has to hold
        (CSRPattern.of_bspline_patch(..., node_box=...) builds exactly that row slice of the pattern)."""
        b, e = self.element_box
        dim = self.patch.dim
        return [int(b[d]) for d in range(dim)], [int(e[d]) + int(self.patch.degrees[d]) for d in range(dim)]

    def overlap_boxes(self, layers=None, mode="replicate"):
        """Split of this slab for overlapping the exchange with compute: (boundary boxes, interior box).
        The rows that go on the wire only receive contributions from the element layers next to the neighbour, so those
        layers are integrated first, their interface rows travel, and the interior is integrated meanwhile.
        How many layers: of the shared node planes [e, e + p) between ranks k and k + 1, mode "replicate" sends all, which
        the last / first p element layers touch; mode "owner" sends upward only the planes the upper rank owns,
        [e + p // 2, e + p) -- touched by the last p - p // 2 layers -- and downward only [e, e + p // 2) -- touched by the
        first p // 2 layers (InterfaceExchange: same split).
        layers: element layers per boundary box at least (more layers give the boundary launch more element columns to
        fill the chip with, at no cost in exchanged rows).
        ([], whole slab) when the slab is too thin or has no neighbour."""
        if mode not in ("replicate", "owner"):
            raise ValueError(mode)
        b, e = self.element_box
        lo, hi = b[self.axis], e[self.axis]
        p = self.patch.degrees[self.axis]
        has_lower, has_upper = self.rank > 0, self.rank < self.world_size - 1
        n_lower = (p if mode == "replicate" else p // 2) if has_lower else 0
        n_upper = (p if mode == "replicate" else p - p // 2) if has_upper else 0
        n_sides = int(n_lower > 0) + int(n_upper > 0)
        if layers is not None and n_sides:
            extra = min(int(layers), (hi - lo - 1) // n_sides)
            n_lower = max(n_lower, extra) if n_lower else 0
            n_upper = max(n_upper, extra) if n_upper else 0
        need = n_lower + n_upper
        if need == 0 or hi - lo < need + 1:
            return [], (list(b), list(e))

        def box(x0, x1):
            bb, ee = list(b), list(e)
            bb[self.axis], ee[self.axis] = x0, x1
            return bb, ee

        boundary = []
        if n_lower:
            boundary.append(box(lo, lo + n_lower))
        if n_upper:
            boundary.append(box(hi - n_upper, hi))
        return boundary, box(lo + n_lower, hi - n_upper)

    def boxes_share_no_node(self, boxes):
        """True when no node is touched by the elements of two of `boxes` (element boxes that differ along the sharding
        axis only): their handles may then assemble into the same `r` / `A` concurrently, on different streams."""
        p = self.patch.degrees[self.axis]
        spans = sorted((b[self.axis], e[self.axis] + p) for b, e in boxes)      # node planes [begin, end) of each box
        return all(spans[k][1] <= spans[k + 1][0] for k in range(len(spans) - 1))

    def interface_node_planes(self, neighbour):
        """Node-plane indices along `axis` shared with rank `neighbour` (= rank +- 1)."""
        p = self.patch.degrees[self.axis]
        if neighbour == self.rank + 1:
            e = int(self.starts[self.rank + 1])
            return list(range(e, e + p))
        if neighbour == self.rank - 1:
            b = int(self.starts[self.rank])
            return list(range(b, b + p))
        raise ValueError(neighbour)

    def interface_nodes(self, neighbour):
        planes = self.interface_node_planes(neighbour)
        mi = self.patch.node_multi_index()
        return np.nonzero(np.isin(mi[self.axis], planes))[0]


class LocalSlabShard(SlabShard):
    """SlabShard.localized: the same slab, seen from a patch that consists of it and its ghost layers alone."""

    def __init__(self, parent, local_patch, pattern=None, ghost=(0, 0)):
        ax = parent.axis
        b, e = parent.element_box
        n_own = e[ax] - b[ax]
        below, above = int(ghost[0]), int(ghost[1])
        if local_patch.n_spans[ax] != below + n_own + above or any(local_patch.n_spans[d] != parent.patch.n_spans[d]
                                                                    for d in range(parent.patch.dim) if d != ax):
            raise RuntimeError(f"local patch has {local_patch.n_spans} spans; the slab has {n_own} layers along axis {ax} and "
                               f"{below} + {above} ghost layers")
        self.parent = parent
        self.patch, self.pattern = local_patch, pattern
        self.rank, self.world_size, self.axis = parent.rank, parent.world_size, ax
        self.ghost = (below, above)
        self.origin = int(b[ax]) - below               # first element layer = first node plane of the local patch, global
        self.starts = parent.starts - self.origin      # (entries rank, rank + 1 are what the methods read: below, below + n_own)
        begin, end = [0, 0, 0], [1, 1, 1]
        for d in range(local_patch.dim):
            end[d] = int(local_patch.n_spans[d])
        begin[ax], end[ax] = below, below + n_own
        self.element_box = (begin, end)
        self.n_local_elements = parent.n_local_elements

    def global_nodes(self):
        """global (whole-patch, lexicographic) id of every local node, in local order"""
        g = self.parent.patch
        mi = self.patch.node_multi_index()
        out = np.zeros(self.patch.n_nodes, dtype=np.int64)
        stride = 1
        for d in range(g.dim):
            out += (mi[d] + (self.origin if d == self.axis else 0)) * stride
            stride *= g.n_ctrl[d]
        return out

    def global_planes(self, local_planes):
        return [int(k) + self.origin for k in local_planes]


class _RowRuns:
    """the values of a set of rows inside a message: how many there are and where each row's run starts"""

    def __init__(self, total, offsets):
        self.total, self.offsets = int(total), offsets

    def numel(self):
        return self.total


class InterfaceExchange:
    """Packed neighbour exchange of interface rows.  `r` and `A` are this rank's (partial)
    residual and CSR value tensors (torch, any device); rowptr comes from the pattern.

    mode "replicate": both neighbours exchange the rows of all `p` shared node planes and add, so
    every rank ends with the fully assembled rows of every node its elements touch.
    mode "owner" (row-partitioned matrix): of the shared planes [e, e+p) between ranks k and k+1 the
    first p//2 belong to k and the rest to k+1; each rank sends only the rows the neighbour owns and
    adds what it receives into the rows it owns -- half the traffic of "replicate"."""

    def __init__(self, shard, r, A, device=None, mode="replicate", loopback=False, trim=True):
        """loopback: every message goes to this process itself (send and receive on the caller's own rank of the
        communicator) -- a transport check on one GPU: what a neighbour would have received is added into the rows
        this rank would have received into (tests/test_rccl_loopback_gpu.py).

        trim (round 5): a message carries of every row only the entries the SENDER's elements can have written -- an element
        layer l touches the node planes l .. l + p, so entry (row plane a, column plane c) of a rank whose layers are
        [b, e) is zero by construction unless some l in [b, e) has max(a, c) - p <= l <= min(a, c).  Of a degree-2 row
        in owner mode 3 of the 5 column planes remain: 11.7 instead of 19.5 MB per neighbour and direction at the
        north-star size on 8 ranks -- what has to hide behind the last gather.  Sender and receiver list the
        same (row, column) pairs in the same order, each from the slab bounds alone.  False: whole rows (csrc/exchange.hip's
        row kernels)."""
        import torch
        import torch.distributed as dist
        if mode not in ("replicate", "owner"):
            raise ValueError(mode)
        self.torch, self.dist = torch, dist
        self.shard, self.r, self.A, self.mode = shard, r, A, mode
        self.loopback = bool(loopback)
        device = r.device if device is None else device
        rowptr = shard.pattern.rowptr
        if not isinstance(rowptr, torch.Tensor):
            rowptr = torch.from_numpy(np.ascontiguousarray(rowptr, dtype=np.int64))
        rowptr = rowptr.to(device)
        dim = shard.patch.dim
        mi_axis = shard.patch.node_multi_index()[shard.axis]

        # device tensors: the library's row kernels pack / unpack / zero (one wave per row, no per-value index array);
        # host tensors (the gloo tests on oracle data): torch index ops over precomputed positions
        self._hip = bool(r.is_cuda)
        self._rowptr = rowptr
        self._comm_stream = None
        self.trim = bool(trim)
        p_ax = int(shard.patch.degrees[shard.axis])
        if self.trim:
            col = shard.pattern.col
            if not isinstance(col, torch.Tensor):
                col = torch.from_numpy(np.ascontiguousarray(col))
            col = col.to(device)
            plane_of_node = torch.from_numpy(np.ascontiguousarray(mi_axis, dtype=np.int64)).to(device)

        whole_rows = []

        def row_sets(planes, layers):
            """rows of the node planes `planes`, and where their travelling values are: `layers` = (b, e), the element layers
            of the rank whose contributions the message carries (the sender's own, or the neighbour's on the receiving side)"""
            nodes = torch.from_numpy(np.nonzero(np.isin(mi_axis, planes))[0]).to(device)
            rows = (nodes[:, None] * dim + torch.arange(dim, device=device)[None, :]).reshape(-1).contiguous()
            start = rowptr[rows]
            length = rowptr[rows + 1] - start
            total = int(length.sum().item()) if rows.numel() else 0
            offs = torch.cumsum(length, 0) - length
            if self._hip and not self.trim:
                # message layout: [rows.numel() residual entries][values row after row]
                return rows, _RowRuns(total, (offs + rows.numel()).contiguous())
            # positions of all values of those rows, row after row
            idx = torch.repeat_interleave(start - offs, length) + torch.arange(total, device=device)
            if not self._hip:
                whole_rows.append(idx)             # (zero_interface zeroes the shared rows whole, trimmed messages or not)
            if self.trim and total:
                a = torch.repeat_interleave(plane_of_node[rows // dim], length)
                c = plane_of_node[col[idx].long() // dim]
                first = torch.clamp(torch.maximum(a, c) - p_ax, min=int(layers[0]))
                last = torch.clamp(torch.minimum(a, c), max=int(layers[1]) - 1)
                idx = idx[first <= last].contiguous()
            return rows, idx

        self.sides = []
        for nb in (shard.rank - 1, shard.rank + 1):
            if nb < 0 or nb >= shard.world_size:
                continue
            planes = shard.interface_node_planes(nb)
            if mode == "replicate":
                send_planes = recv_planes = planes
            else:
                k = len(planes) // 2
                lower, upper = planes[:k], planes[k:]          # owned by the lower / the upper rank
                send_planes, recv_planes = (upper, lower) if nb > shard.rank else (lower, upper)
            srows, sidx = row_sets(send_planes, (shard.starts[shard.rank], shard.starts[shard.rank + 1]))
            rrows, ridx = row_sets(recv_planes, (shard.starts[nb], shard.starts[nb + 1]))
            self.sides.append(dict(peer=dist.get_rank() if loopback else nb, srows=srows, sidx=sidx, rrows=rrows, ridx=ridx,
                                   send=torch.empty(srows.numel() + sidx.numel(), dtype=r.dtype, device=device),
                                   recv=torch.empty(rrows.numel() + ridx.numel(), dtype=r.dtype, device=device)))
        # all shared rows / values of this rank in one index set each (zero_interface: one or two kernels per step)
        if self.sides:
            if mode == "replicate":
                self._zero_rows = torch.cat([s["srows"] for s in self.sides])
            else:
                self._zero_rows = torch.cat([t for s in self.sides for t in (s["srows"], s["rrows"])])
            # (host tensors: the positions of those rows' values -- in "replicate" mode the sent and the received rows are the same)
            self._zero_idx = None if self._hip else torch.unique(torch.cat(whole_rows))

    def _stream(self):
        from ._capi import torch_stream_of
        return torch_stream_of(self.r)

    def _pack(self, s, with_grad):
        from ._capi import check, lib, ptr
        if self.trim:
            check(lib().mimi_hip_entries_pack(self._stream(), ptr(s["srows"], "int64"), s["srows"].numel(), ptr(s["sidx"], "int64"),
                                              s["sidx"].numel(), ptr(self.r, "float64"),
                                              ptr(self.A, "float64") if with_grad else None, ptr(s["send"], "float64")))
            return
        check(lib().mimi_hip_rows_pack(self._stream(), ptr(self._rowptr, "int64"), ptr(s["srows"], "int64"),
                                       ptr(s["sidx"].offsets, "int64"), s["srows"].numel(), ptr(self.r, "float64"),
                                       ptr(self.A, "float64") if with_grad else None, ptr(s["send"], "float64")))

    def _unpack_add(self, s, with_grad):
        from ._capi import check, lib, ptr
        if self.trim:
            check(lib().mimi_hip_entries_unpack_add(self._stream(), ptr(s["rrows"], "int64"), s["rrows"].numel(),
                                                    ptr(s["ridx"], "int64"), s["ridx"].numel(), ptr(s["recv"], "float64"),
                                                    ptr(self.r, "float64"), ptr(self.A, "float64") if with_grad else None))
            return
        check(lib().mimi_hip_rows_unpack_add(self._stream(), ptr(self._rowptr, "int64"), ptr(s["rrows"], "int64"),
                                             ptr(s["ridx"].offsets, "int64"), s["rrows"].numel(), ptr(s["recv"], "float64"),
                                             ptr(self.r, "float64"), ptr(self.A, "float64") if with_grad else None))

    def owned_node_planes(self):
        """Node planes along the sharding axis whose rows are complete on this rank after an exchange."""
        sh = self.shard
        p = sh.patch.degrees[sh.axis]
        n_planes = int(sh.patch.n_spans[sh.axis]) + p
        if self.mode == "replicate":
            return list(range(int(sh.starts[sh.rank]), int(sh.starts[sh.rank + 1]) + p))
        lo = int(sh.starts[sh.rank]) + p // 2 if sh.rank > 0 else 0
        hi = int(sh.starts[sh.rank + 1]) + p // 2 if sh.rank < sh.world_size - 1 else n_planes
        return list(range(lo, hi))

    def gather_windows(self):
        """(early, rest) for the two-step assembly (NonlinearSolid.Integrate / Gather): node boxes (begin, end) of this slab.
        `early`: the node planes whose rows leave this rank -- gathered first, so that they can travel while `rest`,
        everything else the slab's elements touch, is gathered."""
        sh = self.shard
        p = sh.patch.degrees[sh.axis]
        nb, ne = sh.node_box()
        b, e = int(sh.starts[sh.rank]), int(sh.starts[sh.rank + 1])
        k_lo = p if self.mode == "replicate" else p // 2            # planes sent downward: [b, b + k_lo)
        k_hi = p if self.mode == "replicate" else p - p // 2        # planes sent upward:   [e + p - k_hi, e + p)
        lo, hi = nb[sh.axis], ne[sh.axis]

        def box(x0, x1):
            bb, ee = list(nb), list(ne)
            bb[sh.axis], ee[sh.axis] = x0, x1
            return bb, ee

        early = []
        if sh.rank > 0 and k_lo:
            early.append(box(b, b + k_lo))
            lo = b + k_lo
        if sh.rank < sh.world_size - 1 and k_hi:
            early.append(box(e + p - k_hi, e + p))
            hi = e + p - k_hi
        if lo >= hi:
            raise RuntimeError("slab too thin for the two-step assembly")
        return early, box(lo, hi)

    def start(self, with_grad, ready=None):
        """Pack the rows the neighbours need and put them on the wire (asynchronous with "nccl").

        Device tensors: packing and the sends run on the exchange's own stream, ordered behind `ready` -- an event the
        caller recorded when the interface rows were complete (default: everything enqueued on the current stream so
        far).  A caller that records `ready`, enqueues its interior kernels and only then calls start() keeps the GPU
        busy while the host issues the sends (bench.py)."""
        torch, dist = self.torch, self.dist
        if not self._hip:
            self._start(with_grad)
            return
        if self._comm_stream is None:
            # high priority: its own hardware queue (streams of equal priority share a few queues round-robin, and a
            # pack kernel queued behind the interior kernels of the caller's stream would wait for them), and the small
            # pack kernels get workgroup slots as soon as any free up
            self._comm_stream = torch.cuda.Stream(device=self.r.device, priority=-1)
        if ready is None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.r.device))
        self._comm_stream.wait_event(ready)
        with torch.cuda.stream(self._comm_stream):
            self._start(with_grad)

    def _start(self, with_grad):
        torch, dist = self.torch, self.dist
        staged = self.r.is_cuda and dist.get_backend() == "gloo"
        ops = []
        sizes = []
        for s in self.sides:
            ns, nr = s["srows"].numel(), s["rrows"].numel()
            n_send = ns + (s["sidx"].numel() if with_grad else 0)
            n_recv = nr + (s["ridx"].numel() if with_grad else 0)
            if ns and self._hip:
                self._pack(s, with_grad)
            elif ns:
                torch.index_select(self.r, 0, s["srows"], out=s["send"][:ns])
                if with_grad:
                    torch.index_select(self.A, 0, s["sidx"], out=s["send"][ns:n_send])
            sizes.append((n_send, n_recv))
            if staged:
                # gloo moves host memory only: stage device buffers through the host (test rigs without RCCL)
                s["send_host"] = s["send"][:n_send].cpu()
                s["recv_host"] = torch.empty(n_recv, dtype=self.r.dtype)
                send_buf, recv_buf = s["send_host"], s["recv_host"]
            else:
                send_buf, recv_buf = s["send"][:n_send], s["recv"][:n_recv]
            ops.append((dist.P2POp(dist.isend, send_buf, s["peer"]) if n_send else None,
                        dist.P2POp(dist.irecv, recv_buf, s["peer"]) if n_recv else None))
        if self.loopback:
            # messages to oneself are matched in issue order: a side's send must meet a receive of the SAME length
            pairs = self.loopback_pairs(with_grad)
            ops = [(ops[i][0], ops[j][1]) for i, j in pairs]
        ops = [op for pair in ops for op in pair if op is not None]
        self._pending = (dist.batch_isend_irecv(ops) if ops else [], staged, with_grad)

    def loopback_pairs(self, with_grad=True):
        """loop-back only: [(i, j)] = the message side i sends arrives in the receive buffer of side j.  Both neighbours
        being this rank itself, a send can only meet a receive of its own length: side i's own when they agree; in owner
        mode at odd degree a side sends p // 2 planes and receives p - p // 2 (or the reverse), and the send of one side
        is paired with the receive of the OTHER side, which has its plane count.  Raises when nothing pairs."""
        sizes = [(s["srows"].numel() + (s["sidx"].numel() if with_grad else 0),
                  s["rrows"].numel() + (s["ridx"].numel() if with_grad else 0)) for s in self.sides]
        if all(ns == nr for ns, nr in sizes):
            return [(i, i) for i in range(len(sizes))]
        if len(sizes) == 2 and sizes[0][0] == sizes[1][1] and sizes[1][0] == sizes[0][1]:
            return [(0, 1), (1, 0)]
        raise RuntimeError(f"loop-back exchange: send / receive lengths {sizes} cannot be paired on one rank")

    def finish(self):
        """Wait for the neighbours' rows and add them into the rows this rank owns."""
        reqs, staged, with_grad = self._pending
        self._pending = None
        for req in reqs:
            req.wait()
        for s in self.sides:
            nr = s["rrows"].numel()
            if nr == 0:
                continue
            if staged:
                s["recv"][:s["recv_host"].numel()] = s["recv_host"].to(self.r.device)
            if self._hip:
                self._unpack_add(s, with_grad)
                continue
            self.r.index_add_(0, s["rrows"], s["recv"][:nr])          # the indices are unique
            if with_grad:
                self.A.index_add_(0, s["ridx"], s["recv"][nr:nr + s["ridx"].numel()])

    def _exchange(self, with_grad):
        self.start(with_grad)
        self.finish()

    def zero_interface(self, with_grad=True):
        """Zero the rows of all shared node planes (sent and received ones)."""
        if not self.sides:
            return
        if self._hip:
            from ._capi import check, lib, ptr
            check(lib().mimi_hip_rows_zero(self._stream(), ptr(self._rowptr, "int64"), ptr(self._zero_rows, "int64"),
                                           self._zero_rows.numel(), ptr(self.r, "float64"),
                                           ptr(self.A, "float64") if with_grad else None))
            return
        self.r.index_fill_(0, self._zero_rows, 0.0)
        if with_grad:
            self.A.index_fill_(0, self._zero_idx, 0.0)

    def sum_residual(self):
        self._exchange(False)

    def sum_residual_and_grad(self):
        self._exchange(True)


class ShardedContact:
    """MortarContact over the faces of this rank's element slab.  The one extra exchange of the contact
    path: the nodal area / gap of the nodes shared between slabs are summed over the ranks before the pressure is
    formed -- a small all-reduce over the nodes of the contact face (two doubles per node); the residual / Jacobian rows
    then travel with the domain integrator's InterfaceExchange like any other contribution."""

    def __init__(self, shard, body, pattern, axis, side, device=0, quadrature_order=-1, name="contact", loopback=False):
        import torch
        import torch.distributed as dist
        from .integrators import MortarContact
        self.torch, self.dist = torch, dist
        self.shard = shard
        self.body_ = body
        patch = shard.patch
        self.contact = None
        local = isinstance(shard, LocalSlabShard)
        # (a localized slab is a patch of its own: its end along the sharding axis is the patch face only on the first / last rank)
        has_face = not local or axis != shard.axis or (shard.rank == (0 if side == 0 else shard.world_size - 1))
        try:
            if has_face:
                self.contact = MortarContact(body, name, pattern, patch, axis, side, device=device,
                                             quadrature_order=quadrature_order, element_box=shard.element_box).Prepare()
        except RuntimeError as e:                      # this slab does not touch the contact face
            if "no marked boundary faces" not in str(e):
                raise
        # sorted global node ids of the whole face (the slots of the nodal sum over the ranks)
        face_nodes = (shard.parent.patch if local else patch).boundary_nodes(axis, side)
        self.n_face = len(face_nodes)
        backend = dist.get_backend() if dist.is_initialized() else None
        self.comm_device = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
        self.device = torch.device("cuda", device)
        # Only the ranks whose slab touches the contact face take part in the nodal sum (the others have no face, no
        # pressure, nothing to add): a communicator of their own, made once (collective over ALL ranks: every rank
        # reports whether it has faces); one rank alone needs no exchange at all.
        self.group, self.group_size = None, 1
        if dist.is_initialized() and dist.get_world_size() > 1:
            has = [None] * dist.get_world_size()
            dist.all_gather_object(has, self.contact is not None)
            ranks = [r for r, f in enumerate(has) if f]
            self.group_size = len(ranks)
            if 1 < len(ranks) < dist.get_world_size():
                self.group = dist.new_group(ranks)              # (collective: every rank calls it with the same list)
        if loopback and self.group_size == 1:
            self.group_size = 2          # (one-rank rehearsal: the nodal sum runs over the one-rank communicator, to itself)
        if self.contact is not None:
            n = len(self.contact.MarkedNodes())
            marked = np.asarray(self.contact.MarkedNodes())
            if local:
                marked = shard.global_nodes()[marked]
            self.slot = torch.from_numpy(np.searchsorted(face_nodes, marked).astype(np.int64)).to(self.comm_device)
            # persistent buffers: the step allocates nothing and -- with RCCL -- never waits on the host
            self.buf = torch.zeros(2, self.n_face, dtype=torch.float64, device=self.comm_device)
            self.nodal = torch.empty(2, n, dtype=torch.float64, device=self.device)

    def SetStream(self, stream):
        if self.contact is not None:
            self.contact.SetStream(stream)

    def _sum_nodal(self, u):
        """pass 1 on this rank's faces, then the nodal area / gap summed over the ranks that share the face.  Device
        tensors and RCCL: everything is enqueued -- the kernels on the handle's stream, the copies on torch's current
        stream, which the caller keeps identical to it (bench.py: torch.cuda.set_stream + SetStream), the all-reduce
        ordered behind them by torch.distributed -- and the host does not wait.  gloo (tests): staged through the host."""
        if self.contact is None:
            return
        c = self.contact
        c.GapArea(u)
        c.GetNodal(self.nodal[0], self.nodal[1])
        if self.group_size > 1:
            staged = self.comm_device.type == "cpu"
            if staged:
                c.Synchronize()
            else:
                # the copies and the all-reduce below run on torch's current stream, the kernels on the handle's: the two
                # must be one stream (a handle without SetStream follows torch's by itself) -- refused otherwise, rather
                # than summing nodal values pass 1 has not written yet
                from ._capi import STREAM_NULL
                cur = self.torch.cuda.current_stream(self.device).cuda_stream or STREAM_NULL
                if getattr(c, "_user_stream", False) and c._user_stream_value != cur:
                    raise RuntimeError("ShardedContact: the contact handle was given a stream (SetStream) that is not torch's "
                                       "current stream on its device; set both to the same stream")
            self.buf.zero_()
            self.buf[:, self.slot] = self.nodal.to(self.comm_device)
            self.dist.all_reduce(self.buf, group=self.group)
            self.nodal.copy_(self.buf[:, self.slot])
            c.SetNodal(self.nodal[0], self.nodal[1])

    def AddBoundaryResidual(self, u, r):
        self._sum_nodal(u)
        if self.contact is not None:
            self.contact.AddBoundaryResidualFromNodal(u, 0.0, r, None)

    def AddBoundaryResidualAndGrad(self, u, grad_factor, r, A):
        self._sum_nodal(u)
        if self.contact is not None:
            self.contact.AddBoundaryResidualFromNodal(u, grad_factor, r, A)
