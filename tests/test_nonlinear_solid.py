"""The reference's own solver test (tests/test_nonlinear_solid.py:55-98 in j042/mimi), run
against the HIP integrators through the mimi_amd facade: 10 implicit generalized-alpha steps of
the 2-D p=3 2x2-element beam, displacement compared with the reference's golden files after
every step with the reference's criterion (np.allclose defaults) -- and a tighter 1e-8.

Dof order: the golden files are in MFEM's NURBS numbering, the facade numbers lexicographically;
the permutation is `oracle.harness.GOLDEN_NODE_ORDER_5x5`."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
MESH = os.path.join(HERE, "golden", "meshes", "balken.mesh")


# The inputs that DEFINE the golden fixtures (the reference's tests/test_nonlinear_solid.py sets the same
# numbers): kept as data, applied by one generic builder.
JOHNSON_COOK = dict(A=70, B=140, n=0.2835, m=1.3558, eps0_dot=0.004, reference_temperature=20)
THERMAL = dict(melting_temperature=1500, initial_temperature=20, specific_heat=450, heat_fraction=0.9)
GOLDEN_CASES = {
    # name: material class, extra material attributes, hardening, body force, dt, fixture directory
    "neohook": dict(material="CompressibleOgdenNeoHookean", attrs={}, hardening=None, body_force=(1, -5), dt=0.05,
                    refdir="neohook_h1_p2"),
    "j2": dict(material="J2", attrs=THERMAL, hardening=JOHNSON_COOK, body_force=(1, -3), dt=0.5, refdir="j2_h1_p2"),
    "j2_simo": dict(material="J2Simo", attrs=THERMAL, hardening=JOHNSON_COOK, body_force=(1, -3), dt=0.5,
                    refdir="j2_simo_h1_p2"),
    "j2_log": dict(material="J2Log", attrs=THERMAL, hardening=JOHNSON_COOK, body_force=(1, -3), dt=0.5,
                   refdir="j2_log_h1_p2"),
}
BEAM = dict(mesh=MESH, elevate=2, subdivide=1, young=2100, poisson=0.3, density=1, viscosity=-1, ode_coefficient=0.5,
            clamped=[(2, 0), (2, 1)], newton=("nonlinear_solid", 1e-12, 1e-8, 10, False))


def beam(case=None, elevate=BEAM["elevate"], subdivide=BEAM["subdivide"], viscosity=BEAM["viscosity"], runtime=(),
         tangent_mode=0, finish=True):
    """the facade object of a golden case: mesh, refinement, material, runtime flags, boundary conditions, setup"""
    import mimi_amd as mimi
    nl = mimi.NonlinearSolid()
    nl.read_mesh(BEAM["mesh"])
    if elevate > 0:
        nl.elevate_degrees(elevate)
    if subdivide > 0:
        nl.subdivide(subdivide)
    if case is None:
        return nl
    c = GOLDEN_CASES[case]
    mat = getattr(mimi, c["material"])()
    for k, v in dict(density=BEAM["density"], viscosity=viscosity, **c["attrs"]).items():
        setattr(mat, k, v)
    mat.set_young_poisson(BEAM["young"], BEAM["poisson"])
    if c["hardening"]:
        mat.hardening = mimi.JohnsonCookTemperatureAndRateDependentHardening()
        for k, v in c["hardening"].items():
            setattr(mat.hardening, k, v)
    nl.set_material(mat)
    rc = mimi.RuntimeCommunication()
    rc.set_real("ode_coefficient", BEAM["ode_coefficient"])
    for k, v in runtime:
        rc.set_int(k, v)
    nl.runtime_communication = rc
    bc = mimi.BoundaryConditions()
    for bid, comp in BEAM["clamped"]:
        bc.initial.dirichlet(bid, comp)
    bc.initial.body_force(*c["body_force"])
    nl.boundary_condition = bc
    if finish:
        nl.tangent_mode = tangent_mode
        nl.setup(1)
        nl.configure_newton(*BEAM["newton"])
        nl.time_step_size = c["dt"]
    return nl


def follow_golden_series(nl, case, golden_dir, tol=1e-8):
    from oracle import harness as hz
    u = nl.solution_view("displacement", "x").ravel()
    for i in range(10):
        nl.step_time2()
        ref = hz.golden_to_lexicographic(np.genfromtxt(os.path.join(golden_dir, "ref", GOLDEN_CASES[case]["refdir"], f"x_{i}.txt")))
        assert np.allclose(u, ref)                     # the reference's criterion
        if tol:
            assert np.abs(u - ref).max() < tol, (i, np.abs(u - ref).max())


# which kernels a golden series runs on: analytic tangent -> the sum-factorised small-element tensor kernel,
# reference-FD tangent -> the general kernels
FAMILY_OF_MODE = {0: "tensor_small", 1: "general"}


def test_mesh_counts_cpu():
    """reference tests/test_mesh_refinement.py: counts after read / subdivide / elevate."""
    import mimi_amd as mimi
    s = mimi.Solid()
    s.read_mesh(MESH)
    assert (s.mesh_dim(), s.n_elements(), s.n_vertices(), s.mesh_degrees()) == (2, 1, 4, [1, 1])
    s.elevate_degrees(2)
    s.subdivide(1)
    # n_vertices = control points (py_solid.hpp:132-136: GetNodes()->Size() / dim), as the reference asserts for its degree-3 files
    assert (s.n_elements(), s.n_vertices(), s.mesh_degrees(), s.n_boundary_elements()) == (4, 25, [3, 3], 8)
    p = s.patch()
    assert p.n_ctrl == [5, 5] and np.allclose(p.control_points.max(axis=0), [5.0, 1.0])
    c = mimi.Solid()
    c.read_mesh(os.path.join(HERE, "golden", "meshes", "cube-nurbs.mesh"))
    c.subdivide(2)
    assert (c.mesh_dim(), c.n_elements(), c.n_vertices()) == (3, 64, 125)


@pytest.mark.gpu
@pytest.mark.parametrize("tangent_mode", [0, 1], ids=["analytic", "referenceFD"])
@pytest.mark.parametrize("case", sorted(GOLDEN_CASES))
def test_nonlinear_solid_golden_series(golden_dir, case, tangent_mode):
    """reference tests/test_nonlinear_solid.py:55-114: the four golden series through the HIP integrators, on the kernel
    family DESIGN 4.3 names for each tangent mode"""
    nl = beam(case, tangent_mode=tangent_mode)
    assert nl.domain_.path_ == 1
    follow_golden_series(nl, case, golden_dir)
    assert nl.domain_.LastKernelFamily() == FAMILY_OF_MODE[tangent_mode]
    if GOLDEN_CASES[case]["hardening"]:
        assert nl.domain_.State("accumulated_plastic_strain").max() > 0.05


@pytest.mark.gpu
def test_nonlinear_solid_neohook_iterative_solver(golden_dir):
    """the reference's "use_iterative_solver" route (py_nonlinear_solid.cpp:329-339): GMRES + Jacobi, here on the device;
    the golden series is reached through inexact Newton steps as well"""
    nl = beam("neohook", runtime=[("use_iterative_solver", 1)])
    assert nl.use_iterative_solver_
    follow_golden_series(nl, "neohook", golden_dir, tol=None)
    assert nl.linear_.final_iter_ > 1
    assert nl.domain_.LastKernelFamily() == "tensor_small"


@pytest.mark.gpu
def test_3d_cantilever_runs_through_tensor_kernels():
    """cfg1-like plumbing: 3-D p=2 block from cube-nurbs.mesh, one implicit step converges and
    agrees with the same step taken with the reference-FD tangent."""
    import mimi_amd as mimi
    sols = []
    for mode in (0, 1):
        nl = mimi.NonlinearSolid()
        nl.read_mesh(os.path.join(HERE, "golden", "meshes", "cube-nurbs.mesh"))
        nl.elevate_degrees(1)
        nl.subdivide(1)
        mat = mimi.CompressibleOgdenNeoHookean()
        mat.density = 1
        mat.set_young_poisson(2100, 0.3)
        nl.set_material(mat)
        bc = mimi.BoundaryConditions()
        bid = [a - 1 for a, f in nl._faces.items() if f == (0, 0)][0]
        bc.initial.dirichlet(bid, 0).dirichlet(bid, 1).dirichlet(bid, 2)
        bc.initial.body_force(2, -20)
        nl.boundary_condition = bc
        nl.tangent_mode = mode
        nl.setup(1)
        assert nl.domain_.path_ == 1
        nl.configure_newton("nonlinear_solid", 1e-12, 1e-9, 20, False)
        nl.time_step_size = 0.05
        for _ in range(2):
            nl.step_time2()
        assert nl.newton_history[-1]["converged"]
        sols.append(nl.solution_view("displacement", "x").copy())
    assert np.abs(sols[0]).max() > 1e-4
    assert np.allclose(sols[0], sols[1], rtol=1e-6, atol=1e-9)


def test_skewed_quadrilateral_mesh_cpu():
    """square-nurbs.mesh (the mesh of examples/nl_contact.py): a skewed quadrilateral; after refinement the control net
    is the bilinear map at the Greville abscissae, the quadrature weights integrate its area (2.5) exactly"""
    import mimi_amd as mimi
    from mimi_amd import solid
    s = mimi.Solid()
    s.read_mesh(os.path.join(HERE, "golden", "meshes", "square-nurbs.mesh"))
    s.elevate_degrees(1)
    s.subdivide(3)
    assert s.mesh_degrees() == [2, 2] and s.n_elements() == 64
    patch = s.patch()
    corners = patch.control_points[[0, patch.n_ctrl[0] - 1, patch.n_nodes - patch.n_ctrl[0], patch.n_nodes - 1]]
    assert np.allclose(corners, [[0, 0], [2, 0], [-1, 2], [1, 1]])
    N, wd, conn = solid._element_tables(patch)
    assert np.isclose(wd.sum(), 2.5, rtol=1e-13)
    assert np.allclose(N.sum(axis=2), 1.0)
    # boundary attributes: 1 bottom (eta = 0), 2 top, 3 left (xi = 0), 4 right
    assert s._faces == {1: (1, 0), 2: (1, 1), 3: (0, 0), 4: (0, 1)}


@pytest.mark.gpu
def test_skewed_quadrilateral_steps_and_matches_the_oracle():
    """the HIP integrators on the curved-cell geometry against the oracle on the same control net (tables creator), and
    two implicit steps of the facade converge"""
    import mimi_amd as mimi
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from oracle import iga, ref_path as rp
    from _cases import oracle_material, synthetic_u
    nl = mimi.NonlinearSolid()
    nl.read_mesh(os.path.join(HERE, "golden", "meshes", "square-nurbs.mesh"))
    nl.elevate_degrees(1)
    nl.subdivide(2)
    mat = mimi.CompressibleOgdenNeoHookean()
    mat.density = 1
    mat.viscosity = -1
    mat.set_young_poisson(2100, 0.3)
    nl.set_material(mat)
    bc = mimi.BoundaryConditions()
    bc.initial.dirichlet(0, 0).dirichlet(0, 1)
    bc.initial.body_force(1, -5)
    nl.boundary_condition = bc
    nl.setup(1)
    patch = nl.patch_
    P = iga.Patch(patch.degrees, patch.knots, patch.control_points)
    D = rp.DomainOracle(P, oracle_material("neohook"), n_threads=2)
    u = synthetic_u(P, scale=0.01)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    nl.domain_.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert np.abs(r_g - r_o).max() < 1e-12 * np.abs(r_o).max()
    assert np.abs(A_g - A_o).max() < 1e-11 * np.abs(A_o).max()
    nl.configure_newton("nonlinear_solid", 1e-10, 1e-8, 20, False)
    nl.time_step_size = 0.01
    for _ in range(2):
        nl.step_time2()
        assert nl.newton_history[-1]["converged"]
    assert np.abs(nl.solution_view("displacement", "x")).max() > 0


@pytest.mark.gpu
def test_contact_with_rigid_spline_through_the_facade():
    """examples/nl_contact.py in miniature: a rigid NURBS circle (NearestDistanceToSplines().add_spline, plant_kd_tree)
    pressed into the top edge of the beam by its own penalty; the steps converge and the contact carries load."""
    import types
    import mimi_amd as mimi
    nl = beam(elevate=1, subdivide=2)
    mat = mimi.CompressibleOgdenNeoHookean()
    mat.density = 1
    mat.viscosity = -1
    mat.set_young_poisson(2100, 0.3)
    nl.set_material(mat)
    s = np.sqrt(0.5)
    pts = np.array([[1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, -1], [1, -1], [1, 0]], dtype=float)
    circle = types.SimpleNamespace(degrees=[2], knot_vectors=[[0, 0, 0, .25, .25, .5, .5, .75, .75, 1, 1, 1]],
                                   control_points=np.array([4.0, 1.0 + 0.5 - 0.1]) + 0.5 * pts,
                                   weights=np.array([1, s, 1, s, 1, s, 1, s, 1]))
    nd = mimi.NearestDistanceToSplines()
    nd.add_spline(circle)
    nd.plant_kd_tree(200, 1)
    nd.coefficient = 1e4
    bc = mimi.BoundaryConditions()
    bc.initial.dirichlet(2, 0).dirichlet(2, 1)
    top = [a - 1 for a, f in nl._faces.items() if f == (1, 1)][0]
    bc.current.contact(top, nd)
    nl.boundary_condition = bc
    nl.setup(1)
    nl.configure_newton("nonlinear_solid", 1e-10, 1e-8, 20, False)
    nl.time_step_size = 0.01
    for _ in range(3):
        nl.step_time2()
        assert nl.newton_history[-1]["converged"]
    u = nl.solution_view("displacement", "x").reshape(-1, 2)
    assert u[:, 1].min() < -1e-6                     # pushed down under the circle
    c = nl.contacts_[0]
    c.BoundaryPostTimeAdvance(nl.x)
    assert c.last_force_[1] != 0.0
    # the body moves between steps and the penalty changes, as in examples/nl_contact.py: lifted clear of the beam the
    # contact force vanishes; back down with a stiffer penalty it returns
    f0 = abs(c.last_force_[1])
    circle.control_points[:, 1] += 5.0
    nd.plant_kd_tree(200, 1)
    r = np.zeros_like(nl.x)
    c.AddBoundaryResidual(nl.x, r)
    assert np.all(r == 0.0)
    circle.control_points[:, 1] -= 5.0
    nd.plant_kd_tree(200, 1)
    nd.coefficient = 4e4
    c.AddBoundaryResidual(nl.x, r)
    c.BoundaryPostTimeAdvance(nl.x)
    assert abs(c.last_force_[1]) > 2.0 * f0


@pytest.mark.gpu
def test_viscosity_term():
    """material.viscosity > 0: the damping form of py_nonlinear_solid.cpp:176-192 (C v in the residual, fac1 C in the
    Jacobian).  No reference fixture sets a viscosity (parity unpinned): the facade is compared with the oracle's
    restatement of the same operator, and the damped series must differ from the undamped one."""
    import mimi_amd as mimi
    from oracle import harness as hz, iga, ref_path as rp
    from _cases import oracle_material
    series = {}
    for nu in (-1.0, 40.0):
        nl = beam("neohook", viscosity=nu)
        out = []
        for _ in range(4):
            nl.step_time2()
            out.append(nl.solution_view("displacement", "x").ravel().copy())
        series[nu] = np.array(out)
    assert np.abs(series[40.0] - series[-1.0]).max() > 1e-3
    # oracle: same operator with the damping matrix
    P = iga.Patch.block((2, 2), 3, [5.0, 1.0])
    D = rp.DomainOracle(P, oracle_material("neohook"), n_threads=1)
    mass = hz.assemble_mass(P, D.tables, 1.0, D.rowptr, D.col)
    visc = hz.assemble_viscosity(P, D.tables, 40.0, D.rowptr, D.col)
    rhs = hz.assemble_body_force(P, D.tables, [0.0, -5.0])
    nodes = P.boundary_nodes(0, 0)
    dirichlet = np.sort(np.concatenate([nodes * 2, nodes * 2 + 1]))
    op = hz.Operator(D, D.rowptr, D.col, mass, rhs, dirichlet, visc_vals=visc)
    op.tangent_mode = rp.TANGENT_EXACT
    ode = hz.GeneralizedAlpha2(op, 0.5, dict(rel_tol=1e-12, abs_tol=1e-8, max_iter=10, iterative_mode=False))
    x, v, t = np.zeros(P.n_vdofs), np.zeros(P.n_vdofs), 0.0
    for i in range(4):
        op.dt = 0.05
        t = ode.step(x, v, t, 0.05)
        assert np.abs(x - series[40.0][i]).max() < 1e-8, (i, np.abs(x - series[40.0][i]).max())


@pytest.mark.gpu
def test_facade_on_a_truly_rational_patch(tmp_path):
    """a mesh file whose NURBS weights are NOT a tensor product of 1-D weights: the facade hands the integrator the
    reference's flat per-point tables instead of the patch (general kernels); the assembled residual / tangent equal the
    oracle's on the same rational patch, and a time step converges"""
    import mimi_amd as mimi
    from mimi_amd import nurbs_mesh as nm
    from oracle import iga, ref_path as rp
    from _cases import oracle_material
    from test_mesh_refinement import _write_mfem_nurbs
    nb = nm.read_mfem_nurbs(MESH).elevate(1).refine().refine()
    rng = np.random.default_rng(9)
    nb.weights = 1.0 + 0.3 * rng.random(nb.weights.shape)
    nb.ctrl = nb.ctrl + 0.02 * rng.standard_normal(nb.ctrl.shape)
    f = tmp_path / "rational.mesh"
    _write_mfem_nurbs(f, nb)
    nl = mimi.NonlinearSolid()
    nl.read_mesh(str(f))
    mat = mimi.CompressibleOgdenNeoHookean()
    mat.density = 1
    mat.viscosity = -1
    mat.set_young_poisson(2100, 0.3)
    nl.set_material(mat)
    rc = mimi.RuntimeCommunication()
    rc.set_real("ode_coefficient", 0.5)
    nl.runtime_communication = rc
    bc = mimi.BoundaryConditions()
    bc.initial.dirichlet(2, 0).dirichlet(2, 1)
    bc.initial.body_force(1, -5)
    nl.boundary_condition = bc
    nl.setup(1)
    assert nl.domain_.path_ == 0
    P = iga.Patch(nb.degrees, nb.knots, nb.ctrl, nb.weights)
    D = rp.DomainOracle(P, oracle_material("neohook"), n_threads=2)
    u = 0.02 * rng.standard_normal(P.n_vdofs)
    r_o, A_o = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r_o, A_o, rp.TANGENT_EXACT)
    r_g, A_g = np.zeros(P.n_vdofs), np.zeros(D.nnz)
    nl.domain_.AddDomainResidualAndGrad(u, 1.0, r_g, A_g)
    assert np.abs(r_g - r_o).max() < 1e-12 * np.abs(r_o).max()
    assert np.abs(A_g - A_o).max() < 1e-11 * np.abs(A_o).max()
    nl.configure_newton("nonlinear_solid", 1e-10, 1e-8, 10, False)
    nl.time_step_size = 0.05
    nl.step_time2()
    assert nl.newton_history[-1]["converged"] and np.abs(nl.x).max() > 0
