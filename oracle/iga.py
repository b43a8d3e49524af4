"""TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Single-patch B-spline / NURBS tables, restating the *values* that the reference
asks MFEM for in ``PrecomputedData`` (reference: src/mimi/utils/precomputed.cpp).

MFEM itself is an un-vendored, un-pinned submodule of the reference
(.gitmodules:4-6, third_party/mfem is empty), so this file restates the
published algorithms it implements for this path:

* Cox-de Boor basis functions and first derivatives (Piegl & Tiller, The NURBS
  Book, A2.2/A2.3) -- what ``NURBSFiniteElement::CalcShape/CalcDShape`` return
  (precomputed.cpp:307-311);
* Gauss-Legendre tensor rules with ``order/2 + 1`` points per direction, order
  defaulting to ``2 p + 3`` (precomputed.cpp:284-290);
* ``IsoparametricTransformation::Jacobian/Weight`` (precomputed.cpp:302,320) for
  an element = one non-empty knot span mapped from the reference cube [0,1]^d.

The composite is pinned by the reference's golden time series
(tests/data/ref/*_h1_p2) through tests/test_oracle_golden.py.

Conventions (everything lexicographic, first parametric direction fastest):
  node  A = A0 + n0*(A1 + n1*A2)
  elem  e = e0 + m0*(e1 + m1*e2)
  local a = a0 + (p0+1)*(a1 + (p1+1)*a2)
  quad  q = q0 + nq0*(q1 + nq1*q2)
"""
import numpy as np


def gauss_legendre_01(n):
    """n-point Gauss-Legendre rule on [0, 1] (weights sum to 1)."""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def find_span(knots, p, xi):
    """Index i with knots[i] <= xi < knots[i+1] (last non-empty span at the end)."""
    n = len(knots) - p - 1
    if xi >= knots[n]:
        i = n - 1
        while knots[i] == knots[i + 1]:
            i -= 1
        return i
    i = int(np.searchsorted(knots, xi, side="right")) - 1
    return i


def basis_ders(knots, p, span, xi):
    """Values and first derivatives of the p+1 non-zero B-splines on `span`.

    Piegl & Tiller A2.3 restricted to first derivatives.  Returns (N[p+1], dN[p+1]).
    """
    U = knots
    left = np.zeros(p + 1)
    right = np.zeros(p + 1)
    ndu = np.zeros((p + 1, p + 1))
    ndu[0, 0] = 1.0
    for j in range(1, p + 1):
        left[j] = xi - U[span + 1 - j]
        right[j] = U[span + j] - xi
        saved = 0.0
        for r in range(j):
            ndu[j, r] = right[r + 1] + left[j - r]
            temp = ndu[r, j - 1] / ndu[j, r]
            ndu[r, j] = saved + right[r + 1] * temp
            saved = left[j - r] * temp
        ndu[j, j] = saved
    N = ndu[:, p].copy()
    dN = np.zeros(p + 1)
    if p >= 1:
        for r in range(p + 1):
            d = 0.0
            if r >= 1:
                d += ndu[r - 1, p - 1] / ndu[p, r - 1]
            if r <= p - 1:
                d -= ndu[r, p - 1] / ndu[p, r]
            dN[r] = p * d
    return N, dN


def open_uniform_knots(n_el, p):
    return np.concatenate([np.zeros(p), np.linspace(0.0, 1.0, n_el + 1), np.ones(p)])


def greville(knots, p):
    n = len(knots) - p - 1
    return np.array([np.sum(knots[i + 1:i + p + 1]) / p for i in range(n)])


class Patch:
    """One tensor-product NURBS patch of parametric = physical dimension `dim`."""

    def __init__(self, degrees, knots, ctrl, weights=None):
        self.dim = len(degrees)
        self.p = [int(x) for x in degrees]
        self.knots = [np.asarray(k, dtype=np.float64) for k in knots]
        self.n = [len(k) - p - 1 for k, p in zip(self.knots, self.p)]
        self.n_nodes = int(np.prod(self.n))
        self.ctrl = np.asarray(ctrl, dtype=np.float64).reshape(self.n_nodes, self.dim)
        self.weights = (np.ones(self.n_nodes) if weights is None
                        else np.asarray(weights, dtype=np.float64).reshape(self.n_nodes))
        # non-empty spans per direction
        self.spans = [np.array([i for i in range(p, len(k) - p - 1) if k[i + 1] > k[i]], dtype=np.int64)
                      for k, p in zip(self.knots, self.p)]
        self.m = [len(s) for s in self.spans]
        self.n_el = int(np.prod(self.m))
        self.n_dof = int(np.prod([p + 1 for p in self.p]))
        self.n_vdofs = self.n_nodes * self.dim

    # -- factories -----------------------------------------------------------
    @classmethod
    def block(cls, n_el, p, lengths=None):
        """Open-uniform B-spline block, control points at Greville abscissae
        (affine geometry map), unit cells by default (SURVEY 8d workload)."""
        dim = len(n_el)
        degrees = [p] * dim if np.isscalar(p) else list(p)
        lengths = [float(m) for m in n_el] if lengths is None else lengths
        knots = [open_uniform_knots(m, q) for m, q in zip(n_el, degrees)]
        g = [greville(k, q) * L for k, q, L in zip(knots, degrees, lengths)]
        grids = np.meshgrid(*g, indexing="ij")
        ctrl = np.stack([gr.ravel(order="F") for gr in grids], axis=1)
        return cls(degrees, knots, ctrl)

    # -- index helpers -------------------------------------------------------
    def _unravel(self, idx, shape):
        out = []
        for s in shape:
            out.append(idx % s)
            idx = idx // s
        return out

    def element_multi_index(self):
        e = np.arange(self.n_el)
        return self._unravel(e, self.m)

    def connectivity(self):
        """ElementData::dofs (precomputed.cpp:83): global node ids per element,
        lexicographic local order.  int32 [n_el, n_dof]."""
        em = self.element_multi_index()
        conn = np.zeros((self.n_el, self.n_dof), dtype=np.int64)
        a = np.arange(self.n_dof)
        am = self._unravel(a, [p + 1 for p in self.p])
        stride = 1
        for d in range(self.dim):
            first = self.spans[d][em[d]] - self.p[d]          # first non-zero basis
            conn += (first[:, None] + am[d][None, :]) * stride
            stride *= self.n[d]
        return conn.astype(np.int32)

    def vdofs(self, conn=None):
        """ElementData::v_dofs (precomputed.cpp:84): component-grouped
        [dofs*dim+0 ..., dofs*dim+1 ..., ...] for the byVDIM space."""
        conn = self.connectivity() if conn is None else conn
        return np.concatenate([conn * self.dim + c for c in range(self.dim)], axis=1).astype(np.int32)

    # -- 1-D tables ----------------------------------------------------------
    def quad_points_per_dir(self, quadrature_order=-1):
        out = []
        for p in self.p:
            # precomputed.cpp:284-286 (order = 2*p+3 when negative), Gauss-Legendre
            # with order/2+1 points per direction
            order = 2 * max(self.p) + 3 if quadrature_order < 0 else quadrature_order
            out.append(order // 2 + 1)
        return out

    def tables_1d(self, quadrature_order=-1):
        """Per direction d: B[d][m_d, p+1, nq], D[d][m_d, p+1, nq] (derivative wrt the
        element reference coordinate in [0,1]), gauss weights w[d][nq]."""
        nq = self.quad_points_per_dir(quadrature_order)
        B, D, W, X = [], [], [], []
        for d in range(self.dim):
            x, w = gauss_legendre_01(nq[d])
            k, p = self.knots[d], self.p[d]
            Bd = np.zeros((self.m[d], p + 1, nq[d]))
            Dd = np.zeros_like(Bd)
            for ie, s in enumerate(self.spans[d]):
                h = k[s + 1] - k[s]
                for iq in range(nq[d]):
                    N, dN = basis_ders(k, p, s, k[s] + x[iq] * h)
                    Bd[ie, :, iq] = N
                    Dd[ie, :, iq] = dN * h
            B.append(Bd); D.append(Dd); W.append(w); X.append(x)
        return B, D, W, X

    # -- full per-(element, quad point) tables ------------------------------
    def tables(self, quadrature_order=-1, elements=None):
        """QuadData (precomputed.hpp:58-71) for every element and quadrature point:
          N[e,q,a], dN_dxi[e,q,a,d], dN_dX[e,q,a,J], weight[e,q], det[e,q]
        following precomputed.cpp:295-322."""
        B, D, W, _ = self.tables_1d(quadrature_order)
        em = self.element_multi_index()
        if elements is not None:
            em = [x[elements] for x in em]
        ne = len(em[0])
        conn = self.connectivity()
        if elements is not None:
            conn = conn[elements]
        dim = self.dim
        # tensor products: value and per-direction derivative
        letters = "xyz"[:dim]
        def tp(factors):
            # factors[d]: [ne, p+1, nq_d] -> [ne, q(lexi), a(lexi)]
            if dim == 2:
                t = np.einsum("eax,eby->eyxba", factors[0], factors[1])
            else:
                t = np.einsum("eax,eby,ecz->ezyxcba", factors[0], factors[1], factors[2])
            return t.reshape(ne, -1, self.n_dof)
        Bf = [B[d][em[d]] for d in range(dim)]
        Df = [D[d][em[d]] for d in range(dim)]
        Nb = tp(Bf)
        dNb = np.stack([tp([Df[k] if k == d else Bf[k] for k in range(dim)]) for d in range(dim)], axis=-1)
        # rational (NURBS) weighting
        wa = self.weights[conn]                                  # [ne, a]
        Wsum = np.einsum("eqa,ea->eq", Nb, wa)
        dWsum = np.einsum("eqad,ea->eqd", dNb, wa)
        N = Nb * wa[:, None, :] / Wsum[:, :, None]
        dN_dxi = (dNb * wa[:, None, :, None] * Wsum[:, :, None, None]
                  - (Nb * wa[:, None, :])[..., None] * dWsum[:, :, None, :]) / (Wsum ** 2)[:, :, None, None]
        # geometry Jacobian dX/dxi[e,q,I,d] = sum_a X[a,I] dN_dxi[a,d]
        Xe = self.ctrl[conn]                                     # [ne, a, I]
        J = np.einsum("eaI,eqad->eqId", Xe, dN_dxi)
        det = np.linalg.det(J)
        Jinv = np.linalg.inv(J)                                  # dxi/dX [e,q,d,I]
        dN_dX = np.einsum("eqad,eqdJ->eqaJ", dN_dxi, Jinv)       # precomputed.cpp:320-321
        if dim == 2:
            wq = np.einsum("y,x->yx", W[1], W[0]).ravel()
        else:
            wq = np.einsum("z,y,x->zyx", W[2], W[1], W[0]).ravel()
        weight = np.broadcast_to(wq, (ne, wq.size)).copy()
        return dict(N=N, dN_dxi=dN_dxi, dN_dX=dN_dX, weight=weight, det=det, Jinv=Jinv, conn=conn)

    # -- sparsity ------------------------------------------------------------
    def sparsity(self):
        """CSR pattern of PrepareSparsity (precomputed.cpp:151-174): union of dense
        element blocks over v_dofs, rows finalised with sorted column indices.
        Returns (rowptr int64[n_vdofs+1], col int32[nnz])."""
        vd = self.vdofs().astype(np.int64)
        n = self.n_vdofs
        nt = vd.shape[1]
        keys = (vd[:, :, None] * n + vd[:, None, :]).reshape(-1)
        keys = np.unique(keys)
        rows = keys // n
        cols = (keys % n).astype(np.int32)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(rowptr, rows + 1, 1)
        rowptr = np.cumsum(rowptr)
        return rowptr, cols

    def a_ids(self, rowptr, col):
        """ElementData::A_ids (precomputed.cpp:185-199): flat CSR position of every
        (row r, col c) of the element block, stored column-major: A_ids[e, c*n + r]."""
        vd = self.vdofs().astype(np.int64)
        ne, nt = vd.shape
        out = np.zeros((ne, nt * nt), dtype=np.int64)
        for e in range(ne):
            r = vd[e]
            for ir in range(nt):
                s, t = rowptr[r[ir]], rowptr[r[ir] + 1]
                pos = s + np.searchsorted(col[s:t], vd[e])
                out[e, np.arange(nt) * nt + ir] = pos
        return out

    # -- boundary faces ------------------------------------------------------
    def face_tables(self, axis, side, quadrature_order=-1):
        """Boundary-element QuadData for the patch face {xi_axis = side} (side 0/1):
        what LoadBE / CalcShape / CalcDShape give MortarContact
        (precomputed.cpp:100-143, 295-311; mortar_contact.cpp:100-133).

        The face parametrisation is ordered so that the deformable-surface normal of
        ComputeUnitNormal (integrator_utils.hpp:216-251) points out of the body.
        Returns conn[f, a], N[f,q,a], dN_dxi[f,q,a,dim-1], weight[f,q].
        """
        dim = self.dim
        B, D, W, _ = self.tables_1d(quadrature_order)
        tang = [d for d in range(dim) if d != axis]
        # orientation: normal = t0 x t1 (3-D) or (t_y, -t_x) (2-D) must be outward
        if dim == 3:
            # e_axis = e_t0 x e_t1 for (t0,t1) cyclic after axis
            cyc = [(axis + 1) % 3, (axis + 2) % 3]
            tang = cyc if side == 1 else cyc[::-1]
        else:
            # 2-D: tangent d -> normal (d1, -d0); tangent +x -> -y ; tangent +y -> +x
            t = tang[0]
            outward_sign = +1 if side == 1 else -1
            # normal along axis with sign s requires tangent sign:
            # axis=1 (y): n = (d1,-d0) = (0,-d0) -> d0 = -s ; axis=0 (x): n=(d1,0) -> d1 = s
            flip = (outward_sign > 0) if axis == 1 else (outward_sign < 0)
            tang = [t]
            flip2d = flip
        mt = [self.m[t] for t in tang]
        nf = int(np.prod(mt))
        f = np.arange(nf)
        fm = self._unravel(f, mt)
        pt = [self.p[t] + 1 for t in tang]
        ndf = int(np.prod(pt))
        a = np.arange(ndf)
        am = self._unravel(a, pt)
        node_fixed = 0 if side == 0 else self.n[axis] - 1
        strides = [int(np.prod(self.n[:d])) for d in range(dim)]
        conn = np.full((nf, ndf), node_fixed * strides[axis], dtype=np.int64)
        Bf, Df = [], []
        for k, t in enumerate(tang):
            first = self.spans[t][fm[k]] - self.p[t]
            conn += (first[:, None] + am[k][None, :]) * strides[t]
            Bf.append(B[t][fm[k]]); Df.append(D[t][fm[k]])
        flip = dim == 2 and flip2d
        if dim == 2:
            N = np.einsum("eax->exa", Bf[0])
            dN = np.einsum("eax->exa", Df[0])[..., None]
            wq = W[tang[0]].copy()
            if flip:
                # reverse the parametrisation: xi' = 1 - xi
                N = N[:, ::-1, :]
                dN = -dN[:, ::-1, :, :]
                wq = wq[::-1]
        else:
            N = np.einsum("eax,eby->eyxba", Bf[0], Bf[1]).reshape(nf, -1, ndf)
            d0 = np.einsum("eax,eby->eyxba", Df[0], Bf[1]).reshape(nf, -1, ndf)
            d1 = np.einsum("eax,eby->eyxba", Bf[0], Df[1]).reshape(nf, -1, ndf)
            dN = np.stack([d0, d1], axis=-1)
            wq = np.einsum("y,x->yx", W[tang[1]], W[tang[0]]).ravel()
        wa = self.weights[conn]
        if not np.all(wa == 1.0):
            Ws = np.einsum("fqa,fa->fq", N, wa)
            dWs = np.einsum("fqad,fa->fqd", dN, wa)
            Nr = N * wa[:, None, :] / Ws[:, :, None]
            dN = (dN * wa[:, None, :, None] * Ws[:, :, None, None]
                  - (N * wa[:, None, :])[..., None] * dWs[:, :, None, :]) / (Ws ** 2)[:, :, None, None]
            N = Nr
        weight = np.broadcast_to(wq, (nf, wq.size)).copy()
        return dict(conn=conn.astype(np.int32), N=np.ascontiguousarray(N),
                    dN_dxi=np.ascontiguousarray(dN), weight=weight)

    def boundary_nodes(self, axis, side):
        idx = np.arange(self.n_nodes)
        mi = self._unravel(idx, self.n)
        sel = mi[axis] == (0 if side == 0 else self.n[axis] - 1)
        return idx[sel]
