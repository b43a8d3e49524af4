"""Mesh reader, degree elevation and subdivision of the facade against the counts the reference's own test asserts
(tests/test_mesh_refinement.py:4-94 in j042/mimi: the numbers below are that test's; the .mesh files under
tests/golden/meshes are the reference's data files), plus what only the data files can pin without MFEM: the dof
numbering (the degree-3 files list a uniform net, which the reader must put back in lexicographic order) and the
exactness of elevation / refinement (the elevated degree-1 file is the degree-3 file)."""
import os

import numpy as np
import pytest

MESHES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes")


def solid(name):
    import mimi_amd
    s = mimi_amd.Solid()
    s.read_mesh(os.path.join(MESHES, name))
    return s


def counts(s):
    return (s.mesh_dim(), s.n_vertices(), s.n_elements(), s.n_boundary_elements(), s.n_subelements(), s.mesh_degrees())


def test_read_2d_mesh():
    assert counts(solid("square-nurbs.mesh")) == (2, 4, 1, 4, 4, [1, 1])
    assert counts(solid("square-nurbs-3.mesh")) == (2, 16, 1, 4, 4, [3, 3])


def test_read_3d_mesh():
    assert counts(solid("cube-nurbs.mesh")) == (3, 8, 1, 6, 6, [1, 1, 1])
    assert counts(solid("cube-nurbs-3.mesh")) == (3, 64, 1, 6, 6, [3, 3, 3])


def test_subdivide():
    s = solid("square-nurbs.mesh")
    s.subdivide(1)
    assert counts(s) == (2, 9, 4, 8, 12, [1, 1])
    s = solid("cube-nurbs.mesh")
    s.subdivide(1)
    assert counts(s) == (3, 27, 8, 24, 36, [1, 1, 1])


@pytest.mark.parametrize("low,high", [("square-nurbs.mesh", "square-nurbs-3.mesh"), ("cube-nurbs.mesh", "cube-nurbs-3.mesh")])
def test_elevate_degrees(low, high):
    first, second = solid(low), solid(high)
    first.elevate_degrees(2)
    assert counts(first) == counts(second)
    # beyond the counts: same knot vectors; and where the two files describe the same cell (the cube; square-nurbs.mesh is a
    # skewed quadrilateral, square-nurbs-3.mesh the unit square) the elevated patch IS the degree-3 file
    a, b = first.patch(), second.patch()
    for ka, kb in zip(a.knots, b.knots):
        assert np.allclose(ka, kb)
    if np.allclose(a.control_points.min(0), b.control_points.min(0)) and np.allclose(a.control_points.max(0), b.control_points.max(0)) \
            and a.dim == 3:
        assert np.allclose(a.control_points, b.control_points, atol=1e-11)


@pytest.mark.parametrize("name", ["square-nurbs-3.mesh", "cube-nurbs-3.mesh"])
def test_degree_3_files_come_out_lexicographic(name):
    """these files list the uniform net {0, 1/3, 2/3, 1}^dim in MFEM's dof order; after the reader's renumbering node
    (i, j, k) must sit at (i, j, k) / 3"""
    s = solid(name)
    p = s.patch()
    grid = np.stack(np.meshgrid(*[np.arange(4) / 3.0] * p.dim, indexing="ij"), axis=-1)
    lex = grid.reshape(-1, p.dim, order="F") if p.dim == 1 else np.array(
        [[(i % 4) / 3.0, ((i // 4) % 4) / 3.0] + ([(i // 16) / 3.0] if p.dim == 3 else []) for i in range(4 ** p.dim)])
    assert np.allclose(p.control_points, lex, atol=1e-11)
    order = s.mfem_node_order()
    assert sorted(order) == list(range(4 ** p.dim))


def test_mfem_numbering_of_the_golden_case():
    """balken.mesh + elevate_degrees(2) + subdivide(1) (the reference's solver tests): the numbering the golden vectors
    of tests/data/ref are in, recovered independently during the survey (oracle/harness.py)"""
    from oracle import harness
    s = solid("balken.mesh")
    s.elevate_degrees(2)
    s.subdivide(1)
    assert counts(s)[:3] == (2, 25, 4)
    assert np.array_equal(s.mfem_node_order(), harness.GOLDEN_NODE_ORDER_5x5)


def test_refinement_keeps_the_geometry():
    """knot insertion and degree elevation do not move the map: a skewed cell evaluated before and after"""
    from mimi_amd import nurbs_mesh as nm
    s = solid("sqn.mesh")
    before = s._nurbs
    s.elevate_degrees(1)
    s.subdivide(2)
    after = s._nurbs
    assert counts(s) == (2, 36, 16, 16, 40, [2, 2])
    rng = np.random.default_rng(0)
    for xi in rng.random((20, 2)):
        def point(nb):
            rows = [nm.basis_row(k, p, x) for k, p, x in zip(nb.knots, nb.degrees, xi)]
            n0 = nb.n_ctrl[0]
            num, den = np.zeros(2), 0.0
            for a1 in range(nb.degrees[1] + 1):
                for a0 in range(nb.degrees[0] + 1):
                    node = rows[0][0] + a0 + n0 * (rows[1][0] + a1)
                    b = rows[0][1][a0] * rows[1][1][a1] * nb.weights[node]
                    num += b * nb.ctrl[node]
                    den += b
            return num / den
        assert np.allclose(point(before), point(after), atol=1e-12)


def test_runtime_communication_npz(tmp_path):
    """save cadence and npz output of RuntimeCommunication (runtime_communication.hpp:115-130,163-193): arrays
    `x_<i_timestep>` appended to one archive, loadable the way scripts/npz_to_txt.py does"""
    import mimi_amd
    rc = mimi_amd.RuntimeCommunication()
    rc.set_fname(str(tmp_path / "out.npz"))
    rc.append_should_save("x", 2)
    rc.initialize_time_step()
    assert not rc.should_save("v")
    saved = []
    for step in range(5):
        if rc.should_save("x"):
            rc.save_dynamic_vector("x_", np.full(7, float(step)))
            saved.append(step)
        rc.next_time_step(0.1)
    assert saved == [0, 2, 4]
    rc.setup_real_history("gap", 10)
    rc.record_real_history("gap", 0.5)
    rc.record_real_history("gap", 0.25)
    rc.save_real_history("gap")
    npz = np.load(rc.fname)
    assert sorted(npz.keys()) == ["gap_history", "x_0", "x_2", "x_4"]
    assert np.array_equal(npz["x_4"], np.full(7, 4.0)) and np.array_equal(npz["gap_history"], [0.5, 0.25])
    assert np.array_equal(rc.latest_vector("x_"), np.full(7, 4.0))
    assert rc.get_real_history_at("gap", 1) == 0.25


def _write_mfem_nurbs(path, nb):
    """a NurbsPatch as an 'MFEM NURBS mesh v1.0' file in MFEM's dof order (what SaveMesh of the reference writes)"""
    dim = nb.dim
    order = nb.mfem_order()
    quad_edges = [(0, 0, 1), (0, 3, 2), (1, 0, 3), (1, 1, 2)]
    hex_edges = [(0, 0, 1), (0, 3, 2), (0, 4, 5), (0, 7, 6), (1, 0, 3), (1, 1, 2), (1, 4, 7), (1, 5, 6), (2, 0, 4), (2, 1, 5),
                 (2, 2, 6), (2, 3, 7)]
    edges = quad_edges if dim == 2 else hex_edges
    lines = ["MFEM NURBS mesh v1.0", "", "dimension", str(dim), "", "elements", "1",
             "1 3 0 1 2 3" if dim == 2 else "1 5 0 1 2 3 4 5 6 7", "", "boundary"]
    if dim == 2:
        lines += ["4", "1 1 0 1", "2 1 2 3", "3 1 3 0", "4 1 1 2"]
    else:
        lines += ["6", "1 3 3 2 1 0", "2 3 4 5 6 7", "3 3 0 1 5 4", "4 3 1 2 6 5", "5 3 2 3 7 6", "6 3 3 0 4 7"]
    lines += ["", "edges", str(len(edges))] + [" ".join(map(str, e)) for e in edges]
    lines += ["", "vertices", str(2 ** dim), "", "knotvectors", str(dim)]
    for k, p in zip(nb.knots, nb.degrees):
        lines.append(f"{p} {len(k) - p - 1} " + " ".join(repr(float(x)) for x in k))
    lines += ["", "weights"] + [repr(float(nb.weights[n])) for n in order]
    lines += ["", "FiniteElementSpace", f"FiniteElementCollection: NURBS{nb.degrees[0]}", f"VDim: {dim}", "Ordering: 1", ""]
    lines += [" ".join(repr(float(x)) for x in nb.ctrl[n]) for n in order]
    open(path, "w").write("\n".join(lines) + "\n")


@pytest.mark.parametrize("name,elev,sub", [("sqn.mesh", 2, 1), ("cube-nurbs.mesh", 1, 1), ("cube-nurbs-3.mesh", 0, 1)])
def test_refined_rational_patch_survives_a_file_round_trip(tmp_path, name, elev, sub):
    """a curved, rational, multi-span patch of degree > 1 written in MFEM's dof order and read back: same knots, control
    net and weights (the reader's renumbering is the inverse of the writer's), and the same geometry as before refinement"""
    from mimi_amd import nurbs_mesh as nm
    s = solid(name)
    nb = s._nurbs
    rng = np.random.default_rng(3)
    nb.ctrl = nb.ctrl + 0.05 * rng.standard_normal(nb.ctrl.shape)           # a skewed cell
    nb.weights = nb.weights * (1.0 + 0.3 * rng.random(nb.weights.shape))    # truly rational
    coarse = nm.NurbsPatch(nb.degrees, nb.knots, nb.ctrl.copy(), nb.weights.copy(), nb.faces, nb.edges, nb.element_vertices)
    fine = coarse.elevate(elev) if elev else coarse
    for _ in range(sub):
        fine = fine.refine()
    f = tmp_path / "patch.mesh"
    _write_mfem_nurbs(f, fine)
    back = nm.read_mfem_nurbs(str(f))
    assert back.degrees == fine.degrees
    for a, b in zip(back.knots, fine.knots):
        assert np.allclose(a, b, atol=1e-15)
    assert np.allclose(back.ctrl, fine.ctrl, atol=1e-13) and np.allclose(back.weights, fine.weights, atol=1e-13)
    assert back.faces == fine.faces

    def point(pt, xi):
        rows = [nm.basis_row(k, p, x) for k, p, x in zip(pt.knots, pt.degrees, xi)]
        n = pt.n_ctrl
        num, den = np.zeros(pt.dim), 0.0
        for idx in np.ndindex(*[p + 1 for p in pt.degrees]):
            node, stride, b = 0, 1, 1.0
            for d in range(pt.dim):
                node += (rows[d][0] + idx[d]) * stride
                stride *= n[d]
                b *= rows[d][1][idx[d]]
            b *= pt.weights[node]
            num += b * pt.ctrl[node]
            den += b
        return num / den

    for xi in rng.random((10, coarse.dim)):
        assert np.allclose(point(coarse, xi), point(back, xi), atol=1e-12)
