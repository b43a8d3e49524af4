"""Shared problem definitions for the tests (inputs follow the reference's own tests:
tests/test_nonlinear_solid.py:6-81 in j042/mimi; synthetic blocks follow SURVEY 8d)."""
import numpy as np

JC_TEST = dict(kind="JohnsonCookTempRate", A=70, B=140, n=0.2835, m=1.3558, eps0_dot=0.004,
               reference_temperature=20)


def oracle_material(name):
    from oracle import ref_path as rp
    if name == "neohook":
        return rp.make_material("neohookean", 2100, 0.3, density=1.0)
    if name == "stvk":
        return rp.make_material("stvk", 2100, 0.3, density=1.0)
    if name == "j2linear":
        return rp.make_material("j2linear", 2100, 0.3, density=1.0, isotropic_hardening=40.0, kinematic_hardening=25.0,
                                sigma_y=70.0)
    kind = {"j2": "j2", "j2simo": "j2simo", "j2log": "j2log"}[name]
    return rp.make_material(kind, 2100, 0.3, density=1.0, hardening=JC_TEST, heat_fraction=0.9,
                            specific_heat=450, initial_temperature=20, melting_temperature=1500)


def synthetic_u(patch, scale=0.05, seed=20241008, clamp_axis=0):
    """u = scale*h*N(0,1), Dirichlet face x=0 zeroed (h = 1 for unit cells)."""
    rng = np.random.default_rng(seed)
    u = scale * rng.standard_normal(patch.n_vdofs)
    nodes = patch.boundary_nodes(clamp_axis, 0)
    u.reshape(-1, patch.dim)[nodes] = 0.0
    return u


def balken_oracle(matname, tangent_mode=0, n_threads=1):
    """2-D 5x1 beam, p=3, 2x2 elements (balken.mesh + elevate_degrees(2) + subdivide(1))."""
    from oracle import iga, ref_path as rp, harness as hz
    P = iga.Patch.block((2, 2), 3, [5.0, 1.0])
    D = rp.DomainOracle(P, oracle_material(matname), n_threads=n_threads)
    force, dt = (-5.0, 0.05) if matname == "neohook" else (-3.0, 0.5)
    mass = hz.assemble_mass(P, D.tables, 1.0, D.rowptr, D.col)
    rhs = hz.assemble_body_force(P, D.tables, [0.0, force])
    nodes = P.boundary_nodes(0, 0)
    dirichlet = np.sort(np.concatenate([nodes * 2, nodes * 2 + 1]))
    op = hz.Operator(D, D.rowptr, D.col, mass, rhs, dirichlet)
    op.tangent_mode = tangent_mode
    ode = hz.GeneralizedAlpha2(op, 0.5, dict(rel_tol=1e-12, abs_tol=1e-8, max_iter=10, iterative_mode=False))
    return P, D, op, ode, dt
