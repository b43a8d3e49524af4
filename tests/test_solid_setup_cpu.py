"""The facade's host-side set-up of mass matrix, damping matrix and body-force vector (mimi_amd/solid.py
_assemble_mass_viscosity_rhs: py_nonlinear_solid.cpp:155-192,221-283): arithmetic CSR positions in the structured pattern
and colour-wise accumulation, against the oracle's element-by-element assembly with a column search."""
import numpy as np
import pytest


@pytest.mark.parametrize("n_el,p", [((5, 4, 3), 2), ((4, 3, 3), 3), ((7, 5), 2), ((3, 2, 2), 1)])
def test_mass_and_rhs_against_the_oracle(n_el, p):
    import mimi_amd
    from mimi_amd import solid
    from oracle import iga, ref_path as rp, harness as hz
    from _cases import oracle_material
    P = iga.Patch.block(n_el, p)
    D = rp.DomainOracle(P, oracle_material("neohook"), n_threads=2)
    mass_o = hz.assemble_mass(P, D.tables, 1.7, D.rowptr, D.col)
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    rowptr = np.asarray(D.rowptr, dtype=np.int64)
    # positions: every (row node, column node) pair of every element lands on its column in the pattern
    N, wd, conn = solid._element_tables(patch)
    pos0, row_len = solid._structured_positions(patch, rowptr, conn)
    dim = patch.dim
    assert np.array_equal(np.asarray(D.col)[pos0], np.broadcast_to(conn[:, None, :] * dim, pos0.shape))
    assert np.array_equal(row_len, np.diff(rowptr)[conn * dim])
    mass, visc, rhs = solid._assemble_mass_viscosity_rhs(patch, rowptr, 1.7, -1.0, {1: -9.81}, chunk=7)
    assert visc is None
    assert np.abs(mass - mass_o).max() <= 1e-13 * np.abs(mass_o).max()
    # body force: component 1 of node a carries -9.81 int N_a; the integrals sum to the volume of the block
    f = rhs.reshape(-1, dim)
    assert np.all(f[:, 0] == 0.0) and np.isclose(f[:, 1].sum(), -9.81 * np.prod(patch.control_points.max(axis=0) - patch.control_points.min(axis=0)), rtol=1e-12)
    # the same bits every run and for every chunking
    mass2, _, rhs2 = solid._assemble_mass_viscosity_rhs(patch, rowptr, 1.7, -1.0, {1: -9.81}, chunk=1000)
    assert np.array_equal(mass, mass2) and np.array_equal(rhs, rhs2)


def test_damping_matrix_rows_sum_to_zero():
    """C = nu int grad N_a . grad N_b: constants are in its null space (partition of unity), and it is symmetric"""
    import scipy.sparse as sp
    import mimi_amd
    from mimi_amd import solid
    from oracle import iga
    n_el, p = (4, 3, 2), 2
    P = iga.Patch.block(n_el, p)
    rowptr, col = P.sparsity()
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    mass, visc, rhs = solid._assemble_mass_viscosity_rhs(patch, np.asarray(rowptr, dtype=np.int64), 1.0, 0.3, {})
    C = sp.csr_matrix((visc, col, rowptr), shape=(patch.n_vdofs, patch.n_vdofs))
    assert np.abs(C @ np.ones(patch.n_vdofs)).max() <= 1e-12 * np.abs(visc).max()
    assert abs(C - C.T).max() <= 1e-13 * np.abs(visc).max()
    assert np.all(rhs == 0.0)
