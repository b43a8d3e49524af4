"""The reference's examples/nonlinear_solid.py on the HIP integrators (headless: the reference shows the deforming spline
with splinepy / gustaf, which are not part of this repository).

    python examples/nonlinear_solid.py [--steps 20]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mimi_amd as mimi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()

HERE = os.path.dirname(os.path.abspath(__file__))

# create nl solid
nl = mimi.NonlinearSolid()
nl.read_mesh(os.path.join(HERE, "..", "tests", "golden", "meshes", "balken.mesh"))
# refine
nl.elevate_degrees(1)
nl.subdivide(2)

# create material
mat = mimi.CompressibleOgdenNeoHookean()
mat.density = 1
mat.viscosity = -1
# define material properties (young's modulus, poisson's ratio)
mat.set_young_poisson(2100, 0.3)
nl.set_material(mat)

bc = mimi.BoundaryConditions()
bc.initial.dirichlet(2, 0).dirichlet(2, 1)
bc.initial.body_force(1, -5)
nl.boundary_condition = bc

nl.setup(2)
nl.configure_newton("nonlinear_solid", 1e-12, 1e-8, 10, False)
nl.time_step_size = 0.05

u = nl.solution_view("displacement", "x").reshape(-1, nl.mesh_dim())
for i in range(args.steps):
    nl.step_time2()
    h = nl.newton_history[-1]
    print(f"step {i:3d}  t = {nl.current_time:.3f}  Newton iterations {h['iterations']:2d}  |r| = {h['norm']:.2e}  "
          f"tip deflection {u[:, 1].min():+.5f}")
