"""Rigid-curve contact: the scenario of the reference's examples/nl_contact.py (a skewed quadrilateral block, a rigid
cubic Bezier punch that first descends, then slides sideways) on mimi_amd, printed instead of plotted.

The punch is handed over as any object with `degrees`, `knot_vectors`, `control_points` (a splinepy spline qualifies);
a SimpleNamespace is enough.

    python examples/nl_contact.py [--steps 30]
"""
import argparse
import os
import sys
from types import SimpleNamespace

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mimi_amd as mimi  # noqa: E402

DESCEND_STEPS = 100          # after that many steps the punch slides in -x
PUNCH_SPEED = 0.005          # per step


def make_block():
    block = mimi.NonlinearSolid()
    block.read_mesh(os.path.join(REPO, "tests", "golden", "meshes", "square-nurbs.mesh"))
    block.elevate_degrees(1)
    block.subdivide(3)
    steel_like = mimi.CompressibleOgdenNeoHookean()
    steel_like.density, steel_like.viscosity = 7e4, -1
    steel_like.set_young_poisson(1e10, 0.3)
    block.set_material(steel_like)
    return block


def make_punch():
    """cubic Bezier above the top edge; its normal (t_y, -t_x) points down, out of the rigid side"""
    pts = np.array([[-2.5, 1.3], [0.3, 0.7], [0.7, 0.7], [1.5, 1.3]])
    pts += np.array([0.05, 1.0])
    return SimpleNamespace(degrees=[3], knot_vectors=[[0.0] * 4 + [1.0] * 4], control_points=pts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()

    block, punch = make_block(), make_punch()
    rigid_scene = mimi.NearestDistanceToSplines()
    rigid_scene.add_spline(punch)
    rigid_scene.plant_kd_tree(100000, 4)
    rigid_scene.coefficient = 0.5e11

    conditions = mimi.BoundaryConditions()
    conditions.initial.dirichlet(0, 0).dirichlet(0, 1)       # bottom edge clamped
    conditions.current.contact(1, rigid_scene)               # top edge may touch the punch
    block.boundary_condition = conditions
    block.setup(4)
    block.configure_newton("nonlinear_solid", 1e-10, 1e-8, 100, False)
    block.time_step_size = 0.001
    rigid_scene.coefficient = 1e11                           # stiffer penalty once everything is set up

    disp = block.solution_view("displacement", "x").reshape(-1, block.mesh_dim())
    for k in range(args.steps):
        shift = [0.0, -PUNCH_SPEED] if k < DESCEND_STEPS else [-PUNCH_SPEED, 0.0]
        punch.control_points += shift
        rigid_scene.plant_kd_tree(10000, 4)                  # the moved body reaches the device handles here
        block.step_time2()
        info = block.newton_history[-1]
        force = block.contacts_[0].last_force_
        print(f"step {k:3d}  Newton iterations {info['iterations']:2d}  converged {info['converged']}  "
              f"|u|max = {np.abs(disp).max():.3e}  contact force {force}")


if __name__ == "__main__":
    main()
