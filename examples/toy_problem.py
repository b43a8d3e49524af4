"""The scenario of the reference's examples/toy_problem.py on mimi_amd, printed instead of plotted: a viscous neo-Hookean
strip (tests/data/es.mesh, degree elevated and subdivided three times) is pulled by its right edge along a curved channel
whose two walls are rigid quadratic B-spline curves (mortar contact on the strip's bottom and top edges, penalty 1e10).

The reference builds the channel with splinepy (a 25 x 2 control net; `extract.boundaries`, `extract.spline`, `sample`);
the three things it needs from it are written out here: the two wall curves (the rows of the net), the straight line
between them (the net is linear across), and the de Boor evaluation of a curve.

    python examples/toy_problem.py [--steps 20]
"""
import argparse
import os
import sys
from types import SimpleNamespace

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mimi_amd as mimi  # noqa: E402

# control net of the channel (25 x 2 points, degree 2 x 1, first parametric direction fastest; rows: v = 0 and v = 1)
CHANNEL = np.loadtxt(os.path.join(REPO, "examples", "data", "toy_channel_net.txt"))
KNOTS = [0.0] * 3 + [float(k) for k in range(1, 12) for _ in range(2)] + [12.0] * 3


def curve_points(ctrl, knots, degree, ts):
    """de Boor evaluation of a B-spline curve at the parameters ts"""
    knots = np.asarray(knots)
    out = np.empty((len(ts), ctrl.shape[1]))
    for n, t in enumerate(ts):
        k = min(max(np.searchsorted(knots, t, side="right") - 1, degree), len(ctrl) - 1)
        d = [ctrl[j + k - degree].copy() for j in range(degree + 1)]
        for r in range(1, degree + 1):
            for j in range(degree, r - 1, -1):
                lo, hi = knots[j + k - degree], knots[j + 1 + k - r]
                a = 0.0 if hi == lo else (t - lo) / (hi - lo)
                d[j] = (1.0 - a) * d[j - 1] + a * d[j]
        out[n] = d[degree]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--path-points", type=int, default=500, help="the reference samples the channel's centre lines at 500 points")
    args = ap.parse_args()

    strip = mimi.NonlinearSolid()
    strip.read_mesh(os.path.join(REPO, "tests", "golden", "meshes", "es.mesh"))
    strip.elevate_degrees(1)
    strip.subdivide(3)
    rubber = mimi.CompressibleOgdenNeoHookean()
    rubber.density, rubber.viscosity = 4000, 100
    rubber.set_young_poisson(1e7, 0.3)
    strip.set_material(rubber)

    # the walls: the rows v = 0 and v = 1 of the net; the second one reversed (its normal must point into the channel) and
    # its end pulled away, the start of the first one pulled away, as the reference does
    lower = SimpleNamespace(degrees=[2], knot_vectors=[KNOTS], control_points=CHANNEL[:25].copy())
    upper = SimpleNamespace(degrees=[2], knot_vectors=[KNOTS], control_points=CHANNEL[25:][::-1].copy())
    upper.control_points[24] -= 1.0
    lower.control_points[0] += [-5.0, 0.0]
    # the path of the pulled edge: its two ends run along the lines 1 % and 99 % across the channel
    ts = np.linspace(KNOTS[0], KNOTS[-1], args.path_points)
    row0, row1 = curve_points(CHANNEL[:25], KNOTS, 2, ts), curve_points(CHANNEL[25:], KNOTS, 2, ts)
    up, down = 0.99 * row0 + 0.01 * row1, 0.01 * row0 + 0.99 * row1

    wall_lower, wall_upper = mimi.NearestDistanceToSplines(), mimi.NearestDistanceToSplines()
    for wall, curve in ((wall_lower, lower), (wall_upper, upper)):
        wall.add_spline(curve)
        wall.plant_kd_tree(1001, 4)
        wall.coefficient = 1e3

    conditions = mimi.BoundaryConditions()
    conditions.initial.dirichlet(3, 0).dirichlet(3, 1)       # the right edge is moved by hand
    conditions.current.contact(0, wall_upper)
    conditions.current.contact(1, wall_lower)
    strip.boundary_condition = conditions
    strip.setup(4)
    strip.configure_newton("nonlinear_solid", 1e-10, 1e-8, 100, False)
    strip.time_step_size = 0.0003
    wall_lower.coefficient = wall_upper.coefficient = 1e10

    dim = strip.mesh_dim()
    u = strip.solution_view("displacement", "x").reshape(-1, dim)
    x_ref = strip.solution_view("displacement", "x_ref").reshape(-1, dim)
    axis, side = strip._faces[4]                              # boundary id 3 = attribute 4: the edge u = 1
    edge = strip.patch_.boundary_nodes(axis, side)            # its nodes, in order along the edge
    for k in range(args.steps):
        i = min(k, args.path_points - 1)
        u[edge] = np.linspace(down[i], up[i], len(edge)) - x_ref[edge]
        strip.step_time2()
        info = strip.newton_history[-1]
        forces = [c.last_force_ for c in strip.contacts_]
        print(f"step {k:3d}  Newton iterations {info['iterations']:2d}  converged {info['converged']}  "
              f"|u|max = {np.abs(u).max():.3e}  wall forces {forces}")


if __name__ == "__main__":
    main()
