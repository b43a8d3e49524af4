"""The example scripts (the reference's examples/nonlinear_solid.py, examples/nl_contact.py and examples/toy_problem.py on
the HIP integrators, headless) run and converge."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_example_nonlinear_solid():
    text = run("nonlinear_solid.py", "--steps", "3")
    lines = [l for l in text.splitlines() if l.startswith("step")]
    assert len(lines) == 3
    assert float(lines[-1].split()[-1]) < -0.01          # the beam tip goes down under its body force


def test_example_nl_contact():
    text = run("nl_contact.py", "--steps", "25")
    lines = [l for l in text.splitlines() if l.startswith("step")]
    assert len(lines) == 25 and all("converged True" in l for l in lines)
    assert "contact force [0. 0.]" not in lines[-1]      # the curve has reached the body by then


def test_example_toy_problem():
    """viscous strip pulled through a channel of two rigid B-spline curves: the mesh file of the reference, viscosity,
    a prescribed edge, contact on two boundaries"""
    text = run("toy_problem.py", "--steps", "45")
    lines = [l for l in text.splitlines() if l.startswith("step")]
    assert len(lines) == 45 and all("converged True" in l for l in lines)
    assert "array([0., 0.]), array([0., 0.])" in lines[0]              # nothing touches at the start
    assert any("array([0., 0.]), array([0., 0.])" not in l for l in lines[15:])   # a wall has been reached by then
