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
    def __init__(self, degrees, knots, control_points, weights=None):
        """weights: NURBS weights per control point (lexicographic) or None; the library takes tensor-product weights
        (w[a0,a1,a2] = w0[a0] w1[a1] w2[a2]) on this route, see include/mimi_hip.h."""
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
        self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64).reshape(self.n_nodes)

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

    @classmethod
    def block_slab(cls, n_el, degree, axis, begin, end, lengths=None):
        """The element layers [begin, end) along `axis` of block(n_el, degree, lengths) as a patch of its OWN: what one rank
        of a multi-GPU job holds (parallel.SlabShard.localized).  B-splines depend on their own knots only, so the patch
        with the knot slice k[begin : end + 2 p + 1] along `axis` (not an open knot vector at a cut) has exactly the
        basis functions of the whole block that live on those layers -- same knot values, hence the same tables to the
        bit -- on the node planes [begin, end + p); its control points are those planes' Greville points.  Nothing of
        whole-block size is built."""
        dim = len(n_el)
        degrees = [degree] * dim if np.isscalar(degree) else list(degree)
        lengths = [float(m) for m in n_el] if lengths is None else [float(x) for x in lengths]
        knots, grev = [], []
        for d, (m, p, L) in enumerate(zip(n_el, degrees, lengths)):
            k = np.concatenate([np.zeros(p), np.arange(m + 1) / m, np.ones(p)])
            n = len(k) - p - 1
            g = L * np.array([k[i + 1:i + p + 1].sum() / p for i in range(n)])
            if d == axis:
                k = k[begin:end + 2 * p + 1]
                g = g[begin:end + p]
            knots.append(k)
            grev.append(g)
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


class PatchShape:
    """The index space of a patch without its control net: what SlabShard needs to cut slabs (dim, degrees, spans and
    control points per direction).  PatchShape.block(n_el, p) describes BSplinePatch.block(n_el, p)."""

    def __init__(self, degrees, n_spans):
        self.dim = len(degrees)
        self.degrees = [int(p) for p in degrees]
        self.n_spans = [int(m) for m in n_spans]
        self.n_ctrl = [m + p for m, p in zip(self.n_spans, self.degrees)]
        self.n_nodes = int(np.prod(self.n_ctrl))
        self.n_elements = int(np.prod(self.n_spans))
        self.n_vdofs = self.n_nodes * self.dim

    @classmethod
    def block(cls, n_el, degree):
        dim = len(n_el)
        return cls([degree] * dim if np.isscalar(degree) else list(degree), n_el)

    node_multi_index = BSplinePatch.node_multi_index
    boundary_nodes = BSplinePatch.boundary_nodes


# ---- boundary (face) tables for the contact integrator ---------------------------------
def _gauss_01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def _basis_and_derivative(knots, p, span, xi):
    """All p+1 non-zero B-splines of degree p on knot span `span` and their derivatives,
    by the Cox-de Boor recursion written as a triangular table."""
    N = np.zeros((p + 1, p + 1))
    N[0, 0] = 1.0
    for d in range(1, p + 1):
        for j in range(d + 1):
            i = span - d + j
            v = 0.0
            if j >= 1 and knots[i + d] > knots[i]:
                v += (xi - knots[i]) / (knots[i + d] - knots[i]) * N[d - 1, j - 1]
            if j <= d - 1 and knots[i + d + 1] > knots[i + 1]:
                v += (knots[i + d + 1] - xi) / (knots[i + d + 1] - knots[i + 1]) * N[d - 1, j]
            N[d, j] = v
    dN = np.zeros(p + 1)
    for j in range(p + 1):
        i = span - p + j
        if j >= 1 and knots[i + p] > knots[i]:
            dN[j] += p / (knots[i + p] - knots[i]) * N[p - 1, j - 1]
        if j <= p - 1 and knots[i + p + 1] > knots[i + 1]:
            dN[j] -= p / (knots[i + p + 1] - knots[i + 1]) * N[p - 1, j]
    return N[p], dN


def _tables_1d(knots, p, nq):
    x, w = _gauss_01(nq)
    spans = [s for s in range(p, len(knots) - p - 1) if knots[s + 1] > knots[s]]
    B = np.zeros((len(spans), p + 1, nq))
    D = np.zeros_like(B)
    for ie, s in enumerate(spans):
        h = knots[s + 1] - knots[s]
        for iq in range(nq):
            B[ie, :, iq], dn = _basis_and_derivative(knots, p, s, knots[s] + x[iq] * h)
            D[ie, :, iq] = dn * h
    return np.array(spans), B, D, w


def face_tables(patch, axis, side, quadrature_order=-1, element_box=None):
    """Boundary-element tables of the patch face {xi_axis = side}: what the reference's
    MortarContact reads from PrecomputedData for its marked boundary elements
    (QuadData::N, dN_dxi, integration_weight; src/mimi/utils/precomputed.cpp:100-143,295-311).
    The face parametrisation is oriented so that the surface normal of
    ComputeUnitNormal (integrators/integrator_utils.hpp:216-251) points out of the body.
    Returns dofs[f,a] (int32), N[f,q,a], dN_dxi[f,q,dim-1,a], weight[f,q].
    element_box = (begin, end): only the faces of the elements in that box (multi-GPU element slabs)."""
    if getattr(patch, "weights", None) is not None:
        raise RuntimeError("face tables of a rational patch are not generated here: pass the reference's boundary tables")
    dim = patch.dim
    pmax = max(patch.degrees)
    order = 2 * pmax + 3 if quadrature_order < 0 else quadrature_order
    nq = order // 2 + 1
    if dim == 3:
        cyc = [(axis + 1) % 3, (axis + 2) % 3]
        tang = cyc if side == 1 else cyc[::-1]
        flip = False
    else:
        tang = [1 - axis]
        # 2-D normal of tangent t is (t_y, -t_x): outward needs the tangent reversed on these faces
        flip = (side == 1) if axis == 1 else (side == 0)
    tabs = [_tables_1d(patch.knots[t], patch.degrees[t], nq) for t in tang]
    strides = [int(np.prod(patch.n_ctrl[:d])) for d in range(dim)]
    fixed = (0 if side == 0 else patch.n_ctrl[axis] - 1) * strides[axis]
    if dim == 2:
        spans, B, D, w = tabs[0]
        t = tang[0]
        p = patch.degrees[t]
        dofs = fixed + (spans[:, None] - p + np.arange(p + 1)[None, :]) * strides[t]
        N = np.transpose(B, (0, 2, 1))
        dN = np.transpose(D, (0, 2, 1))[:, :, None, :]
        wq = w.copy()
        if flip:
            N, dN, wq = N[:, ::-1, :], -dN[:, ::-1, :, :], wq[::-1]
        weight = np.broadcast_to(wq, (len(spans), nq)).copy()
        face_el = [np.arange(len(spans))]
    else:
        (s0, B0, D0, w0), (s1, B1, D1, w1) = tabs
        t0, t1 = tang
        p0, p1 = patch.degrees[t0], patch.degrees[t1]
        f0, f1 = np.meshgrid(np.arange(len(s0)), np.arange(len(s1)), indexing="ij")
        f0, f1 = f0.T.ravel(), f1.T.ravel()          # t0 fastest
        a0 = np.tile(np.arange(p0 + 1), p1 + 1)
        a1 = np.repeat(np.arange(p1 + 1), p0 + 1)
        dofs = (fixed + (s0[f0][:, None] - p0 + a0[None, :]) * strides[t0]
                + (s1[f1][:, None] - p1 + a1[None, :]) * strides[t1])
        nf = len(f0)
        N = np.einsum("fax,fby->fyxba", B0[f0], B1[f1]).reshape(nf, nq * nq, -1)
        d0 = np.einsum("fax,fby->fyxba", D0[f0], B1[f1]).reshape(nf, nq * nq, -1)
        d1 = np.einsum("fax,fby->fyxba", B0[f0], D1[f1]).reshape(nf, nq * nq, -1)
        dN = np.stack([d0, d1], axis=2)
        weight = np.broadcast_to(np.einsum("y,x->yx", w1, w0).ravel(), (nf, nq * nq)).copy()
        face_el = [f0, f1]
    if element_box is not None:
        b, e = element_box
        parent = 0 if side == 0 else patch.n_spans[axis] - 1
        keep = np.full(len(dofs), b[axis] <= parent < e[axis])
        for t, fe in zip(tang, face_el):
            keep &= (fe >= b[t]) & (fe < e[t])
        dofs, N, dN, weight = dofs[keep], N[keep], dN[keep], weight[keep]
    return (np.ascontiguousarray(dofs, dtype=np.int32), np.ascontiguousarray(N), np.ascontiguousarray(dN),
            np.ascontiguousarray(weight))
