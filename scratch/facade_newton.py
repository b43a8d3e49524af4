"""One implicit time step through the facade (mimi_amd.NonlinearSolid.step_time2) at the north-star element count
(cube-nurbs.mesh, elevate_degrees(1), subdivide(6): 64^3 = 262 144 p=2 elements), iterative route: everything of the
Newton iteration stays in HBM.  Prints the timings and the CSR bytes that crossed PCIe after setup."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mimi_amd as mimi

sub = int(sys.argv[1]) if len(sys.argv) > 1 else 6
iterative = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nl = mimi.NonlinearSolid()
nl.read_mesh(os.path.join(ROOT, "tests", "golden", "meshes", "cube-nurbs.mesh"))
nl.elevate_degrees(1)
nl.subdivide(sub)
mat = mimi.CompressibleOgdenNeoHookean()
mat.density = 1
mat.viscosity = -1
mat.set_young_poisson(2100, 0.3)
nl.set_material(mat)
rc = mimi.RuntimeCommunication()
rc.set_real("ode_coefficient", 0.5)
rc.set_int("use_iterative_solver", iterative)
nl.runtime_communication = rc
bc = mimi.BoundaryConditions()
bc.initial.dirichlet(0, 0).dirichlet(0, 1).dirichlet(0, 2)
bc.initial.body_force(2, -0.5)
nl.boundary_condition = bc
t0 = time.perf_counter()
nl.setup(1)
t_setup = time.perf_counter() - t0
nl.configure_newton("nonlinear_solid", 1e-10, 1e-8, 10, False)
nl.time_step_size = 0.01
out = dict(elements=nl.n_elements(), vdofs=len(nl.x), nnz=int(nl.pattern_.nnz), setup_s=t_setup, path=nl.domain_.path_)
for step in range(2):
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nl.step_time2()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = nl.newton_history[-1]
    out[f"step{step}"] = dict(seconds=dt, newton_iterations=h["iterations"], converged=h["converged"],
                              seconds_per_newton_iteration=dt / max(h["iterations"], 1),
                              gmres_iterations_last_solve=nl.linear_.final_iter_ if iterative else None,
                              norm0=h["norm0"], norm=h["norm"])
# the operator's residual+Jacobian (operators/nonlinear_solid.cpp:240-283) on its own: J = M + fac0 K in one pass (ABI 11,
# mimi_hip_domain_add_residual_and_grad_from), against the reference's order "J <- M, then +=" done with a device copy
import torch
a = nl._torch.zeros_like(nl.d_x_)
for _ in range(3):
    nl._residual_and_grad(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    nl._residual_and_grad(a)
torch.cuda.synchronize()
out["residual_and_grad_ms"] = (time.perf_counter() - t0) / 10 * 1e3
xt = nl._xa + nl._fac0 * a
y = nl._torch.zeros_like(a)
for k in range(13):
    if k == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    nl.d_jac_.copy_(nl.d_mass_)
    nl.domain_.AddDomainResidualAndGrad(xt, nl._fac0, y, nl.d_jac_)
torch.cuda.synchronize()
out["copy_then_add_ms"] = (time.perf_counter() - t0) / 10 * 1e3
for k in range(13):
    if k == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    nl.domain_.AddDomainResidualAndGradFrom(xt, nl._fac0, y, nl.d_mass_, nl.d_jac_)
torch.cuda.synchronize()
out["from_base_ms"] = (time.perf_counter() - t0) / 10 * 1e3
out["csr_bytes_over_pcie_after_setup"] = nl.pcie_csr_bytes_
out["max_displacement"] = float(np.abs(nl.x).max())
print(out)
