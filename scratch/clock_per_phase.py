"""engine clock / power per PHASE of the cfg3 (or north-star) assembly: each phase launched back to back for ~12 s while rocm-smi is sampled
(the two-step entry points: Integrate = the integration kernels, Gather = the row gather).  usage: python scratch/clock_per_phase.py [workload]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n_el, p, material = bench.WORKLOADS[wl]
patch = mimi_amd.BSplinePatch.block(n_el, p)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
G = NonlinearSolid("d", bench.make_material(material), pattern, patch=patch).Prepare()
G.dt_ = 0.5
dev = torch.device("cuda", 0)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
nb, ne = [0, 0, 0], [int(n) for n in patch.n_ctrl]


def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    s = [ln.split(":")[-1].strip() for ln in out.splitlines() if "sclk" in ln or "Power (W)" in ln]
    return " ".join(s)


def run(name, fn, seconds=12.0):
    fn(); G.Synchronize()
    t0 = time.perf_counter(); fn(); G.Synchronize(); one = time.perf_counter() - t0
    n = max(1, int(0.5 / one))
    samples, calls, t_start = [], 0, time.perf_counter()
    while time.perf_counter() - t_start < seconds:
        for _ in range(n):
            fn()
        calls += n
        G.Synchronize()          # (the queue holds half a second of work at most; the sample below is taken while the next batch is NOT running,
        # so enqueue the next batch first)
        for _ in range(n):
            fn()
        calls += n
        samples.append(smi())
        G.Synchronize()
    ms = (time.perf_counter() - t_start) / calls * 1e3
    print(f"{wl} {name}: {ms:.3f} ms per call (rocm-smi's own time included: an upper bound) | sclk, W: " + " | ".join(samples[2:8]), flush=True)


G.AddDomainResidualAndGrad(u, 1.0, r, A); G.Synchronize()
run("integration kernels (Integrate)", lambda: G.Integrate(u))
run("row gather (Gather)", lambda: G.Gather(1.0, r, A, nb, ne))
run("residual-only assembly", lambda: G.AddDomainResidual(u, r))
run("whole residual+Jacobian assembly", lambda: G.AddDomainResidualAndGrad(u, 1.0, r, A))
