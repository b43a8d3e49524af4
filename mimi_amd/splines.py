"""Host-side description of a single tensor-product B-spline patch (the discretisation the
hot path integrates over) and the synthetic blocks of the benchmark configurations.

Only what the product needs to DESCRIBE a patch lives here (knots, control net, index
conventions); basis evaluation, geometry Jacobians and tables are built inside
libmimi_hip (csrc/bspline_host.hpp, csrc/kernels_setup.hpp).

Lexicographic conventions, first parametric direction fastest:
  node A = A0 + n0*(A1 + n1*A2);  element e = e0 + m0*(e1 + m1*e2)
"""
import numpy as np


class BSplinePatch:
    def __init__(self, degrees, knots, control_points):
        self.dim = len(degrees)
        self.degrees = [int(p) for p in degrees]
        self.knots = [np.ascontiguousarray(k, dtype=np.float64) for k in knots]
        self.n_ctrl = [len(k) - p - 1 for k, p in zip(self.knots, self.degrees)]
        self.n_nodes = int(np.prod(self.n_ctrl))
        self.control_points = np.ascontiguousarray(control_points, dtype=np.float64).reshape(self.n_nodes, self.dim)
        self.n_spans = [int(np.count_nonzero(np.diff(k[p:len(k) - p]) > 0)) for k, p in zip(self.knots, self.degrees)]
        self.n_elements = int(np.prod(self.n_spans))
        self.n_dof = int(np.prod([p + 1 for p in self.degrees]))
        self.n_vdofs = self.n_nodes * self.dim

    @classmethod
    def block(cls, n_el, degree, lengths=None):
        """Open-uniform B-spline box with control points at the Greville abscissae (affine
        map); unit cells unless `lengths` is given (the synthetic workloads of BASELINE.json)."""
        dim = len(n_el)
        degrees = [degree] * dim if np.isscalar(degree) else list(degree)
        lengths = [float(m) for m in n_el] if lengths is None else [float(x) for x in lengths]
        knots, grev = [], []
        for m, p, L in zip(n_el, degrees, lengths):
            k = np.concatenate([np.zeros(p), np.arange(m + 1) / m, np.ones(p)])
            knots.append(k)
            n = len(k) - p - 1
            grev.append(L * np.array([k[i + 1:i + p + 1].sum() / p for i in range(n)]))
        pts = np.zeros([len(g) for g in grev][::-1] + [dim])
        for d in range(dim):
            shape = [1] * dim
            shape[dim - 1 - d] = -1
            pts[..., d] = grev[d].reshape(shape)
        return cls(degrees, knots, pts.reshape(-1, dim))

    def node_multi_index(self, nodes=None):
        idx = np.arange(self.n_nodes) if nodes is None else np.asarray(nodes)
        out = []
        for n in self.n_ctrl:
            out.append(idx % n)
            idx = idx // n
        return out

    def boundary_nodes(self, axis, side):
        mi = self.node_multi_index()
        return np.nonzero(mi[axis] == (0 if side == 0 else self.n_ctrl[axis] - 1))[0]

    def slab(self, rank, world_size, axis=None):
        """Element box [begin, end) of `rank` when the elements are sharded in contiguous slabs
        along the longest axis."""
        axis = int(np.argmax(self.n_spans)) if axis is None else axis
        m = self.n_spans[axis]
        chunk = (m + world_size - 1) // world_size
        b = min(rank * chunk, m)
        e = min((rank + 1) * chunk, m)
        begin = [0] * 3
        end = [1] * 3
        for d in range(self.dim):
            end[d] = self.n_spans[d]
        begin[axis], end[axis] = b, e
        return begin, end
