"""The reference's examples/nl_contact.py on the HIP integrators (headless): a skewed quadrilateral pushed by a rigid cubic
Bezier curve that first moves down, then sideways.  The rigid curve is any object with degrees / knot_vectors /
control_points (a splinepy spline has them); here a plain namespace.

    python examples/nl_contact.py [--steps 30]
"""
import argparse
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimi_amd as mimi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()

HERE = os.path.dirname(os.path.abspath(__file__))

# init, read mesh
nl = mimi.NonlinearSolid()
nl.read_mesh(os.path.join(HERE, "..", "tests", "golden", "meshes", "square-nurbs.mesh"))
# refine
nl.elevate_degrees(1)
nl.subdivide(3)

# mat
mat = mimi.CompressibleOgdenNeoHookean()
mat.density = 7e4
mat.viscosity = -1
mat.set_young_poisson(1e10, 0.3)
nl.set_material(mat)

# the rigid body: cubic Bezier curve above the top edge (its normal (t_y, -t_x) points down, out of the rigid body)
curv = types.SimpleNamespace(
    degrees=[3],
    knot_vectors=[[0, 0, 0, 0, 1, 1, 1, 1]],
    control_points=np.array([[-2.5, 1.3], [0.3, 0.7], [0.7, 0.7], [1.5, 1.3]]) + [0.05, 1.0],
)

scene = mimi.NearestDistanceToSplines()
scene.add_spline(curv)
scene.plant_kd_tree(100000, 4)
scene.coefficient = 0.5e11

bc = mimi.BoundaryConditions()
bc.initial.dirichlet(0, 0).dirichlet(0, 1)
bc.current.contact(1, scene)
nl.boundary_condition = bc

# setup needs to be called this assembles bilinear forms, linear forms
nl.setup(4)
nl.configure_newton("nonlinear_solid", 1e-10, 1e-8, 100, False)
nl.time_step_size = 0.001

u = nl.solution_view("displacement", "x").reshape(-1, nl.mesh_dim())
scene.coefficient = 1e11


def move(i):
    if i < 100:
        curv.control_points[:] -= [0, 0.005]
    else:
        curv.control_points[:] -= [0.005, 0]
    scene.plant_kd_tree(10000, 4)


contact = None
for i in range(args.steps):
    move(i)
    nl.step_time2()
    h = nl.newton_history[-1]
    contact = nl.contacts_[0]
    print(f"step {i:3d}  Newton iterations {h['iterations']:2d}  converged {h['converged']}  |u|max = {np.abs(u).max():.3e}  "
          f"contact force {contact.last_force_}")
