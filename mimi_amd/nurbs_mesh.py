"""MFEM NURBS mesh v1.0, single patch: reader, MFEM's dof numbering, degree elevation and uniform refinement.

What PySolid does with mfem::Mesh (src/mimi/py/py_solid.cpp:70-95 ReadMesh, :148-168 ElevateDegrees ->
Mesh::DegreeElevate, :170-183 Subdivide -> Mesh::UniformRefinement; counts in py_solid.hpp:130-157), for the
files the reference ships (tests/data/*.mesh: one patch of any degree, with weights).  MFEM itself is absent;
the file format and the dof numbering are restated from the data files themselves:

  control points / weights are listed in the order of NURBSExtension's dofs: the patch's corner vertices, then the
  interior control points of every patch EDGE in the order of the file's `edges` section, each edge walked from its
  lower-numbered to its higher-numbered vertex, then (3-D) the interiors of the six patch FACES in the hexahedron's
  local face order (3 2 1 0), (0 1 5 4), (1 2 6 5), (2 3 7 6), (3 0 4 7), (4 5 6 7), first along v0 -> v1 (fast) then
  along v0 -> v3, then the patch interior lexicographically.
  (tests/data/square-nurbs-3.mesh and cube-nurbs-3.mesh list the 16 / 64 points of the uniform net in exactly this
  order, and the golden vectors of tests/data/ref follow it: tests/test_mesh_refinement.py here.)

Inside this package nodes are numbered lexicographically (first direction fastest); `mfem_order[k]` is the
lexicographic index of MFEM's dof k.
"""
import re

import numpy as np

_QUAD_REF = [(0, 0), (1, 0), (1, 1), (0, 1)]
_HEX_REF = [r + (0,) for r in _QUAD_REF] + [r + (1,) for r in _QUAD_REF]
_HEX_FACES = [(3, 2, 1, 0), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7)]


# ---- 1-D spline algebra ---------------------------------------------------------------------------
def find_span(knots, p, x):
    n = len(knots) - p - 1
    if x >= knots[n]:
        s = n - 1
        while knots[s] == knots[s + 1]:
            s -= 1
        return s
    return int(np.searchsorted(knots, x, side="right") - 1)


def basis_row(knots, p, x):
    """(first non-zero index, the p + 1 non-zero B-spline values at x) -- Cox-de Boor"""
    s = find_span(knots, p, x)
    N = np.zeros(p + 1)
    N[0] = 1.0
    left, right = np.zeros(p + 1), np.zeros(p + 1)
    for j in range(1, p + 1):
        left[j] = x - knots[s + 1 - j]
        right[j] = knots[s + j] - x
        saved = 0.0
        for r in range(j):
            t = N[r] / (right[r + 1] + left[j - r])
            N[r] = saved + right[r + 1] * t
            saved = left[j - r] * t
        N[j] = saved
    return s - p, N


def greville(knots, p):
    n = len(knots) - p - 1
    return np.array([knots[i + 1:i + p + 1].sum() / p for i in range(n)])


def collocation(knots, p, xs):
    n = len(knots) - p - 1
    C = np.zeros((len(xs), n))
    for r, x in enumerate(xs):
        f, N = basis_row(knots, p, x)
        C[r, f:f + p + 1] = N
    return C


def transfer_matrix(old_knots, p_old, new_knots, p_new):
    """T [n_new, n_old]: control points of the SAME function in a spline space that contains the old one (knot insertion,
    degree elevation): collocation at the new space's Greville abscissae, which is unisolvent (Schoenberg-Whitney)."""
    g = greville(new_knots, p_new)
    return np.linalg.solve(collocation(new_knots, p_new, g), collocation(old_knots, p_old, g))


def elevated_knots(knots, t):
    """degree + t: the multiplicity of every distinct knot grows by t (continuity unchanged)"""
    vals, mult = np.unique(knots, return_counts=True)
    return np.repeat(vals, mult + t)


def refined_knots(knots):
    """KnotVector::UniformRefinement: one new knot in the middle of every non-empty span"""
    vals = np.unique(knots)
    return np.sort(np.concatenate([knots, 0.5 * (vals[1:] + vals[:-1])]))


# ---- MFEM's dof numbering of one patch --------------------------------------------------------------
def mfem_dof_order(n_ctrl, edges, element_vertices=None):
    """lexicographic node index of every MFEM dof.  n_ctrl: control points per direction; edges: the file's `edges`
    section [(knotvector, v_a, v_b)]; element_vertices: the patch's vertex list (default 0..2^dim-1)."""
    dim = len(n_ctrl)
    ref = _QUAD_REF if dim == 2 else _HEX_REF
    ev = list(range(2 ** dim)) if element_vertices is None else list(element_vertices)
    ref_of = {v: np.array(rc) for v, rc in zip(ev, ref)}
    n = np.asarray(n_ctrl)

    def lex(mi):
        idx, stride = 0, 1
        for d in range(dim):
            idx += int(mi[d]) * stride
            stride *= int(n[d])
        return idx

    def along(r0, d, t):           # t-th interior point walking direction d away from a vertex with reference coords r0
        return t if r0[d] == 0 else n[d] - 1 - t

    order = [lex(ref_of[v] * (n - 1)) for v in sorted(ref_of)]
    for _, va, vb in edges:
        lo, hi = min(va, vb), max(va, vb)
        d = int(np.flatnonzero(ref_of[lo] != ref_of[hi])[0])
        base = ref_of[lo] * (n - 1)
        for t in range(1, n[d] - 1):
            mi = base.copy()
            mi[d] = along(ref_of[lo], d, t)
            order.append(lex(mi))
    if dim == 3:
        for face in _HEX_FACES:
            w0, w1, w3 = ref_of[ev[face[0]]], ref_of[ev[face[1]]], ref_of[ev[face[3]]]
            d1 = int(np.flatnonzero(w0 != w1)[0])
            d2 = int(np.flatnonzero(w0 != w3)[0])
            base = w0 * (n - 1)
            for t in range(1, n[d2] - 1):
                for s in range(1, n[d1] - 1):
                    mi = base.copy()
                    mi[d1] = along(w0, d1, s)
                    mi[d2] = along(w0, d2, t)
                    order.append(lex(mi))
    for mi in np.ndindex(*[int(k) - 2 for k in n[::-1]]):           # interior, first direction fastest
        order.append(lex(np.array(mi[::-1]) + 1))
    order = np.array(order, dtype=np.int64)
    assert len(order) == int(np.prod(n)) and len(np.unique(order)) == len(order)
    return order


class NurbsPatch:
    """One NURBS patch: degrees, knot vectors, control net and weights in lexicographic order, the boundary attributes
    (attribute -> (axis, side)) and what is needed to reproduce MFEM's numbering."""

    def __init__(self, degrees, knots, ctrl, weights, faces, edges, element_vertices):
        self.dim = len(degrees)
        self.degrees = [int(p) for p in degrees]
        self.knots = [np.asarray(k, dtype=np.float64) for k in knots]
        self.ctrl = np.asarray(ctrl, dtype=np.float64)
        self.weights = np.asarray(weights, dtype=np.float64)
        self.faces, self.edges, self.element_vertices = faces, edges, element_vertices

    @property
    def n_ctrl(self):
        return [len(k) - p - 1 for k, p in zip(self.knots, self.degrees)]

    @property
    def n_spans(self):
        return [len(np.unique(k)) - 1 for k in self.knots]

    def mfem_order(self):
        return mfem_dof_order(self.n_ctrl, self.edges, self.element_vertices)

    def is_rational(self):
        return not np.allclose(self.weights, 1.0)

    # -- counts (py_solid.hpp:130-157) -------------------------------------------------------------
    def n_vertices(self):           # Mesh()->GetNodes()->Size() / dim: the control points
        return int(np.prod(self.n_ctrl))

    def n_elements(self):
        return int(np.prod(self.n_spans))

    def n_boundary_elements(self):
        m = self.n_spans
        return int(2 * sum(np.prod([m[k] for k in range(self.dim) if k != d]) for d in range(self.dim)))

    def n_subelements(self):        # Mesh::GetNumFaces: edges of the element mesh in 2-D, faces in 3-D
        m = self.n_spans
        return int(sum((m[d] + 1) * np.prod([m[k] for k in range(self.dim) if k != d]) for d in range(self.dim)))

    # -- Mesh::DegreeElevate / Mesh::UniformRefinement ----------------------------------------------
    def _apply(self, new_knots, new_degrees):
        shape = self.n_ctrl[::-1]
        hom = np.concatenate([self.ctrl * self.weights[:, None], self.weights[:, None]], axis=1)
        hom = hom.reshape(shape + [self.dim + 1])                    # [k, j, i, c]
        for d in range(self.dim):
            T = transfer_matrix(self.knots[d], self.degrees[d], new_knots[d], new_degrees[d])
            axis = self.dim - 1 - d
            hom = np.moveaxis(np.tensordot(T, hom, axes=([1], [axis])), 0, axis)
        hom = hom.reshape(-1, self.dim + 1)
        w = hom[:, -1]
        return NurbsPatch(new_degrees, new_knots, hom[:, :-1] / w[:, None], w, self.faces, self.edges, self.element_vertices)

    def elevate(self, t, max_degree=50):
        new_deg = [min(p + int(t), max_degree) for p in self.degrees]
        return self._apply([elevated_knots(k, q - p) for k, p, q in zip(self.knots, self.degrees, new_deg)], new_deg)

    def refine(self):
        return self._apply([refined_knots(k) for k in self.knots], list(self.degrees))


def read_mfem_nurbs(fname):
    text = open(fname).read()
    if not text.lstrip().startswith("MFEM NURBS mesh v1.0"):
        raise RuntimeError(f"{fname} Does not contain NURBS mesh.")          # py_solid.cpp:81-85
    tok = re.sub(r"#.*", "", text).split()

    def section(name):
        return tok.index(name) + 1

    dim = int(tok[section("dimension")])
    if dim not in (2, 3):
        raise RuntimeError(f"Unsupported Dim: {dim}")
    i = section("elements")
    if int(tok[i]) != 1:
        raise RuntimeError("only single-patch NURBS meshes are supported")
    ev = [int(v) for v in tok[i + 3:i + 3 + 2 ** dim]]
    i = section("boundary")
    nb = int(tok[i])
    i += 1
    nv_b = 2 if dim == 2 else 4
    bdr = []
    for _ in range(nb):
        bdr.append((int(tok[i]), [int(v) for v in tok[i + 2:i + 2 + nv_b]]))
        i += 2 + nv_b
    i = section("edges")
    ne = int(tok[i])
    edges = [(int(tok[i + 1 + 3 * k]), int(tok[i + 2 + 3 * k]), int(tok[i + 3 + 3 * k])) for k in range(ne)]
    i = section("knotvectors")
    nk = int(tok[i])
    i += 1
    kvs = []
    for _ in range(nk):
        p, n = int(tok[i]), int(tok[i + 1])
        kvs.append((p, np.array([float(x) for x in tok[i + 2:i + 2 + n + p + 1]])))
        i += 2 + n + p + 1
    ref = _QUAD_REF if dim == 2 else _HEX_REF
    ref_of = {v: np.array(rc) for v, rc in zip(ev, ref)}
    # direction d follows the knot vector of the patch edge leaving vertex 0 along d (running from v_a to v_b)
    degrees, knots = [None] * dim, [None] * dim
    for kv, va, vb in edges:
        d = int(np.flatnonzero(ref_of[va] != ref_of[vb])[0])
        if ref_of[va][d] != 0:
            raise RuntimeError("patch edges running against their knot vector are not supported")
        if degrees[d] is None:
            degrees[d], knots[d] = kvs[kv]
        elif degrees[d] != kvs[kv][0] or len(knots[d]) != len(kvs[kv][1]) or not np.allclose(knots[d], kvs[kv][1]):
            raise RuntimeError("parallel patch edges with different knot vectors")
    n_ctrl = [len(k) - p - 1 for k, p in zip(knots, degrees)]
    n_nodes = int(np.prod(n_ctrl))
    i = section("weights")
    w_mfem = np.array([float(x) for x in tok[i:i + n_nodes]])
    i = tok.index("Ordering:") + 2
    c_mfem = np.array([float(x) for x in tok[i:i + n_nodes * dim]]).reshape(n_nodes, dim)
    order = mfem_dof_order(n_ctrl, edges, ev)
    ctrl = np.zeros((n_nodes, dim))
    weights = np.zeros(n_nodes)
    ctrl[order] = c_mfem
    weights[order] = w_mfem
    # boundary attribute -> (axis, side): the reference coordinate all vertices of the boundary element share
    faces = {}
    for attr, verts in bdr:
        rc = np.array([ref_of[v] for v in verts])
        for d in range(dim):
            if np.all(rc[:, d] == rc[0, d]):
                faces[attr] = (d, int(rc[0, d]))
    return NurbsPatch(degrees, knots, ctrl, weights, faces, edges, ev)
