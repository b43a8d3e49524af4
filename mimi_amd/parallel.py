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
    one, whose node planes are contiguous in the lexicographic numbering)."""

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


class InterfaceExchange:
    """Packed neighbour exchange of interface rows.  `r` and `A` are this rank's (partial)
    residual and CSR value tensors (torch, any device); rowptr comes from the pattern."""

    def __init__(self, shard, r, A, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.shard, self.r, self.A = shard, r, A
        device = r.device if device is None else device
        rowptr = shard.pattern.rowptr
        if not isinstance(rowptr, torch.Tensor):
            rowptr = torch.from_numpy(np.ascontiguousarray(rowptr, dtype=np.int64))
        rowptr = rowptr.to(device)
        dim = shard.patch.dim
        self.sides = []
        for nb in (shard.rank - 1, shard.rank + 1):
            if nb < 0 or nb >= shard.world_size:
                continue
            nodes = torch.from_numpy(shard.interface_nodes(nb)).to(device)
            rows = (nodes[:, None] * dim + torch.arange(dim, device=device)[None, :]).reshape(-1)
            start = rowptr[rows]
            length = rowptr[rows + 1] - start
            total = int(length.sum().item())
            # positions of all values of those rows, row after row
            offs = torch.cumsum(length, 0) - length
            idx = torch.repeat_interleave(start - offs, length) + torch.arange(total, device=device)
            self.sides.append(dict(peer=nb, rows=rows, idx=idx,
                                   send=torch.empty(rows.numel() + total, dtype=r.dtype, device=device),
                                   recv=torch.empty(rows.numel() + total, dtype=r.dtype, device=device)))

    def _exchange(self, with_grad):
        torch, dist = self.torch, self.dist
        ops = []
        for s in self.sides:
            nr = s["rows"].numel()
            n = nr + (s["idx"].numel() if with_grad else 0)
            s["send"][:nr] = self.r[s["rows"]]
            if with_grad:
                s["send"][nr:] = self.A[s["idx"]]
            ops.append(dist.P2POp(dist.isend, s["send"][:n], s["peer"]))
            ops.append(dist.P2POp(dist.irecv, s["recv"][:n], s["peer"]))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for s in self.sides:
            nr = s["rows"].numel()
            self.r[s["rows"]] += s["recv"][:nr]
            if with_grad:
                self.A[s["idx"]] += s["recv"][nr:nr + s["idx"].numel()]

    def zero_interface(self, with_grad=True):
        for s in self.sides:
            self.r[s["rows"]] = 0.0
            if with_grad:
                self.A[s["idx"]] = 0.0

    def sum_residual(self):
        self._exchange(False)

    def sum_residual_and_grad(self):
        self._exchange(True)
