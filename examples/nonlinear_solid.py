"""Cantilever under its own weight: the scenario of the reference's examples/nonlinear_solid.py, driven through
mimi_amd (HIP integrators) and printed instead of plotted.

    python examples/nonlinear_solid.py [--steps 20] [--dt 0.05]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mimi_amd as mimi  # noqa: E402


def build_beam(mesh_file):
    """5 x 1 beam, clamped at x = 0, neo-Hookean, gravity-like body force"""
    solid = mimi.NonlinearSolid()
    solid.read_mesh(mesh_file)
    solid.elevate_degrees(1)          # p = 2
    solid.subdivide(2)                # 4 x 4 elements

    rubber = mimi.CompressibleOgdenNeoHookean()
    rubber.density, rubber.viscosity = 1, -1
    rubber.set_young_poisson(2100, 0.3)
    solid.set_material(rubber)

    conditions = mimi.BoundaryConditions()
    clamp = conditions.initial
    clamp.dirichlet(2, 0)
    clamp.dirichlet(2, 1)
    conditions.initial.body_force(1, -5)
    solid.boundary_condition = conditions

    solid.setup(2)
    solid.configure_newton("nonlinear_solid", 1e-12, 1e-8, 10, False)
    return solid


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dt", type=float, default=0.05)
    args = ap.parse_args()
    beam = build_beam(os.path.join(REPO, "tests", "golden", "meshes", "balken.mesh"))
    beam.time_step_size = args.dt
    displacement = beam.solution_view("displacement", "x").reshape(-1, beam.mesh_dim())
    for k in range(args.steps):
        beam.step_time2()
        info = beam.newton_history[-1]
        print(f"step {k:3d}  t = {beam.current_time:.3f}  Newton iterations {info['iterations']:2d}  |r| = {info['norm']:.2e}  "
              f"tip deflection {displacement[:, 1].min():+.5f}")


if __name__ == "__main__":
    main()
