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
