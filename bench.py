#!/usr/bin/env python3
"""Benchmark of the hot path: element-integrations/sec of the residual+Jacobian assembly
(one `AddDomainResidualAndGrad` over the whole mesh = one "step"), BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 workload: 128x128x16 p=2 neo-Hookean B-spline block (the configuration the north-star
target is quoted on; it fits one MI355X).  For N > 1 the SAME mesh is sharded in element slabs
across the ranks (strong scaling) and the shared-dof rows of the residual / Jacobian are
summed between neighbouring ranks over RCCL inside the timed region.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured with events on the
launch stream) and `cpu_baseline` (the restated reference CPU path = oracle, timed on this
box's host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_el, p, material)
    "cfg1": ((8, 8, 2), 2, "neohookean"),
    "cfg2": ((64, 64, 8), 2, "neohookean"),
    "northstar": ((128, 128, 16), 2, "neohookean"),
    "cfg3": ((128, 128, 16), 3, "j2"),
    "northstar_j2": ((128, 128, 16), 2, "j2"),
    "cfg3_neo": ((128, 128, 16), 3, "neohookean"),
    # the reference's other materials (general kernels, dual-number tangents): measured for DESIGN.md only
    "cfg2_stvk": ((64, 64, 8), 2, "stvk"),
    "cfg2_j2linear": ((64, 64, 8), 2, "j2linear"),
    "cfg2_j2simo": ((64, 64, 8), 2, "j2simo"),
    "cfg2_j2log": ((64, 64, 8), 2, "j2log"),
    "cfg5": ((256, 256, 32), 2, "neohookean"),
    # orientation experiments (same block, short axis first)
    "northstar_zfirst": ((16, 128, 128), 2, "neohookean"),
}

# algorithmic bytes / flops per element integration
def b_alg(dim, p, grad=True, j2=False):
    n_dof = (p + 1) ** dim
    n_tdof = n_dof * dim
    n_q = (p + 2) ** dim
    b = 8 * n_tdof + 8 * n_tdof + 4 * n_dof + 8 * n_q * (dim * dim + 1)
    if grad:
        b += 8 * n_tdof * n_tdof
    if j2:
        b += 8 * 11 * n_q
    return b


def measured_traffic(workload, world, grad, material):
    """HBM bytes per step from the committed rocprofv3 PMC passes (profiles/r01_traffic.json: FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH_SIZE doubled as the gfx950 guide prescribes, calibrated for this
    kernel's 8-byte accesses by scratch/fetch_calib.hip).  None when no measurement exists for this case."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)
    except OSError:
        return None
    key = f"{workload}/{material}/{'grad' if grad else 'residual'}/n{world}"
    return t.get(key, {}).get("bytes_per_step")


def measured_traffic_per_kernel(workload, world, grad, material):
    """(phase 1 bytes, phase 2 bytes) of the same PMC passes, or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f).get(f"{workload}/{material}/{'grad' if grad else 'residual'}/n{world}")
        f2 = sum(v for k, v in t["fetch_size_kb_raw"].items() if "p2" in k)
        w2 = sum(v for k, v in t["write_size_kb"].items() if "p2" in k)
        f1 = sum(t["fetch_size_kb_raw"].values()) - f2
        w1 = sum(t["write_size_kb"].values()) - w2
        return (2 * f1 + w1) * 1024, (2 * f2 + w2) * 1024
    except (OSError, TypeError, KeyError, AttributeError):
        return None


def make_material(kind):
    import mimi_amd
    if kind == "neohookean":
        m = mimi_amd.CompressibleOgdenNeoHookean()
        m.density = 1.0
        m.set_young_poisson(2100, 0.3)
        return m
    if kind == "stvk":
        m = mimi_amd.StVenantKirchhoff()
        m.density = 1.0
        m.set_young_poisson(2100, 0.3)
        return m
    if kind == "j2linear":
        m = mimi_amd.J2Linear()
        m.density = 1.0
        m.set_young_poisson(2100, 0.3)
        m.isotropic_hardening, m.kinematic_hardening, m.sigma_y = 40.0, 25.0, 70.0
        return m
    m = {"j2": mimi_amd.J2, "j2simo": mimi_amd.J2Simo, "j2log": mimi_amd.J2Log}[kind]()
    m.density = 1.0
    m.set_young_poisson(2100, 0.3)
    m.heat_fraction, m.specific_heat = 0.9, 450
    m.initial_temperature, m.melting_temperature = 20, 1500
    h = mimi_amd.JohnsonCookTemperatureAndRateDependentHardening()
    h.A, h.B, h.n, h.m, h.eps0_dot, h.reference_temperature = 70, 140, 0.2835, 1.3558, 0.004, 20
    m.hardening = h
    return m


def synthetic_u(patch, scale=0.05, seed=20241008):
    """u = 0.05*h*N(0,1) (h = 1: unit cells), Dirichlet face x=0 zeroed."""
    rng = np.random.default_rng(seed)
    u = scale * rng.standard_normal(patch.n_vdofs)
    u.reshape(-1, patch.dim)[patch.boundary_nodes(0, 0)] = 0.0
    return u


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(p, material, seconds_hint=12.0):
    """The restated reference CPU path (oracle/ref_path.c: forward-FD element Jacobian,
    per-thread full-size arrays + reduction pass, OpenMP) on a bounded sample of the workload."""
    from oracle import iga, ref_path as rp
    n_el = (32, 32, 8)
    threads = min(os.cpu_count() or 1, 32)
    P = iga.Patch.block(n_el, p)
    if material in ("neohookean", "stvk"):
        mat = rp.make_material(material, 2100, 0.3)
    elif material == "j2linear":
        mat = rp.make_material("j2linear", 2100, 0.3, isotropic_hardening=40.0, kinematic_hardening=25.0, sigma_y=70.0)
    else:
        mat = rp.make_material(material, 2100, 0.3, hardening=dict(kind="JohnsonCookTempRate", A=70, B=140, n=0.2835,
                               m=1.3558, eps0_dot=0.004, reference_temperature=20),
                               specific_heat=450, initial_temperature=20, melting_temperature=1500)
    D = rp.DomainOracle(P, mat, n_threads=threads)
    D.set_dt(0.5)
    rng = np.random.default_rng(20241008)
    u = 0.05 * rng.standard_normal(P.n_vdofs)
    u.reshape(-1, 3)[P.boundary_nodes(0, 0)] = 0.0
    r = np.zeros(P.n_vdofs)
    A = np.zeros(D.nnz)
    D.add_domain_residual_and_grad(u, 1.0, r, A, rp.TANGENT_FD)          # warm-up (page faults)
    reps, t_total = 0, 0.0
    while t_total < seconds_hint and reps < 64:      # about 12 s of CPU work
        t0 = time.perf_counter()
        D.add_domain_residual_and_grad(u, 1.0, r, A, rp.TANGENT_FD)
        t_total += time.perf_counter() - t0
        reps += 1
    # the same path with an ANALYTIC element tangent, so that the GPU/CPU ratio is not credited with the
    # FD -> analytic change of algorithm (a few seconds more)
    D.add_domain_residual_and_grad(u, 1.0, r, A, rp.TANGENT_EXACT)
    reps_a, t_a = 0, 0.0
    while t_a < 4.0 and reps_a < 256:
        t0 = time.perf_counter()
        D.add_domain_residual_and_grad(u, 1.0, r, A, rp.TANGENT_EXACT)
        t_a += time.perf_counter() - t0
        reps_a += 1
    return dict(value=P.n_el * reps / t_total, unit="element-integrations/s", cores=threads, kind="port",
                sample=f"{'x'.join(map(str, n_el))} p={p} {material} block ({P.n_el} elements), {reps} residual+Jacobian "
                       f"assemblies, reference forward-FD element Jacobian, OpenMP {threads} threads "
                       f"(host has {os.cpu_count()} logical cores, {_cpu_model()})",
                analytic_tangent_value=P.n_el * reps_a / t_a,
                analytic_tangent_note="same restated path and threads with the oracle's analytic element tangent instead of "
                                      "the reference's forward differences")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("MIMI_BENCH_WORKLOAD", "northstar"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--residual-only", action="store_true", help="time AddDomainResidual instead (not the headline metric)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from mimi_amd import parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # MIMI_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (interface rows are
    # staged through the host, ranks share the visible GPUs); the measured configuration is "nccl" = RCCL
    backend = os.environ.get("MIMI_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n_el, p, material = WORKLOADS[args.workload]
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern.of_bspline_patch(patch, device=local_rank, on_device=True)
    shard = parallel.SlabShard(patch, pattern, rank, world)
    # a non-default stream: the library launches on it and the events below are recorded on it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    def make_integrator(box):
        g = NonlinearSolid("domain", make_material(material), pattern, patch=patch, device=local_rank,
                           element_box=box).Prepare()
        g.dt_ = 0.5
        g.SetStream(stream.cuda_stream)
        return g

    # N > 1, tangent assembly: the element layers next to a neighbour are integrated first, their interface rows
    # go on the wire, and the interior is integrated while they travel (parallel.SlabShard.overlap_boxes)
    boundary_boxes, interior_box = shard.overlap_boxes() if (world > 1 and not args.residual_only) else ([], shard.element_box)
    integ = make_integrator(interior_box)
    boundary = [make_integrator(b) for b in boundary_boxes]

    u = torch.from_numpy(synthetic_u(patch)).to(dev)
    r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
    A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    exchange = parallel.InterfaceExchange(shard, r, A, dev, mode="owner") if world > 1 else None

    def step():
        if exchange:
            # interface rows hold exactly this step's sum (interior rows just keep accumulating)
            exchange.zero_interface(with_grad=not args.residual_only)
        if args.residual_only:
            integ.AddDomainResidual(u, r)
            if exchange:
                exchange.sum_residual()
        else:
            if boundary:
                for g in boundary:
                    g.AddDomainResidualAndGrad(u, 1.0, r, A)
                exchange.start(True)
                integ.AddDomainResidualAndGrad(u, 1.0, r, A)
                exchange.finish()
            else:
                integ.AddDomainResidualAndGrad(u, 1.0, r, A)
                if exchange:
                    exchange.sum_residual_and_grad()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(args.steps):
        step()
        ev[k + 1].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for g in [integ] + boundary:
        g.Synchronize()   # raises if a kernel reported an error
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]))
    # "also reported per full Newton iteration (= 1 x (R+J) + 2 x (R) assemblies, newton.cpp:142-190)":
    # the residual-only assembly is timed AFTER the timed region above (N = 1 only; informational)
    # per-kernel durations, live: events inside the library around phase 1 and phase 2 (after the timed region)
    phase_ms = None
    if world == 1 and not args.residual_only and integ.path_ == 1:
        integ.SetPhaseTiming(True)
        acc1 = acc2 = 0.0
        for _ in range(5):
            integ.AddDomainResidualAndGrad(u, 1.0, r, A)
            a1, a2 = integ.PhaseMs()
            acc1 += a1
            acc2 += a2
        integ.SetPhaseTiming(False)
        phase_ms = (acc1 / 5, acc2 / 5)
    residual_ms = None
    if world == 1 and not args.residual_only:
        for _ in range(3):
            integ.AddDomainResidual(u, r)
        integ.Synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            integ.AddDomainResidual(u, r)
        integ.Synchronize()
        residual_ms = (time.perf_counter() - t1) / 20 * 1e3

    if rank == 0:
        n_elements = patch.n_elements
        value = n_elements * args.steps / elapsed
        balg = b_alg(patch.dim, p, grad=not args.residual_only, j2=(material == "j2"))
        local_elements = integ.n_elements_ + sum(g.n_elements_ for g in boundary)
        achieved = balg * local_elements / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "element-integrations/sec (residual+Jacobian assembly)" if not args.residual_only
                      else "element-integrations/sec (residual-only assembly)",
            "value": value, "unit": "element-integrations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{'x'.join(map(str, n_el))} p={p} {material} B-spline block, "
                                   f"{n_elements} elements, n_q={(p + 2) ** patch.dim}, nnz={pattern.nnz}",
                       "name": args.workload,
                       "parallelism": f"element slabs x{world}" + (", interface rows summed on their owner rank" if world > 1 else "")
                                      + (", exchange overlapped with the interior elements" if boundary else ""),
                       "kernel_path": "tensor" if integ.path_ == 1 else "general",
                       "u": "0.05*N(0,1), seed 20241008, face x=0 clamped"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0,
                         "traffic": measured_traffic(args.workload, world, not args.residual_only, material),
                         "kernel": "one step = tensor_wgsym_kernel (neo-Hookean; J2: tensor_point_kernel + tensor_wgs_kernel) "
                                   "(integration, phase 1) + tensor_p2_kernel (row gather, phase 2) on rank 0; "
                                   "avg_launch_ms is their sum, measured with events on the launch stream",
                         "algorithmic_bytes_per_element": balg, "elements_per_launch": local_elements,
                         "avg_launch_ms": kernel_ms,
                         "phase_ms": None if phase_ms is None else
                         {"phase1_integration_kernels": phase_ms[0], "phase2_row_gather": phase_ms[1],
                          **({} if measured_traffic_per_kernel(args.workload, world, True, material) is None else
                             {"phase1_hbm_GB_per_s": measured_traffic_per_kernel(args.workload, world, True, material)[0] / phase_ms[0] / 1e6,
                              "phase2_hbm_GB_per_s": measured_traffic_per_kernel(args.workload, world, True, material)[1] / phase_ms[1] / 1e6}),
                          "how": "HIP events recorded by the library on the launch stream around the kernels of each phase, "
                                 "mean of 5 assemblies after the timed region; phase 1 is fp64-pipe-bound, phase 2 HBM-bound "
                                 "(DESIGN.md 4.1)"}},
        }
        if not args.residual_only and patch.dim == 3:
            # second view of the same step (the tangent contraction is fp64-matrix-bound before it is HBM-bound):
            # the ALGORITHMIC flops of the dense B^T A B form (no symmetry, no sum factorisation) over the step time.  The
            # kernels execute several times fewer (sum factorisation; symmetric half for hyperelastic materials), so this
            # ratio may exceed 1 -- it says how far the step is below the 9.7 ms the dense form needs at the matrix peak.
            n_dof, n_q = (p + 1) ** 3, (p + 2) ** 3
            f_alg = n_q * (4 * n_dof * 9 + 200 + 2 * (n_dof * 81 + (3 * n_dof) ** 2 * 3))
            tf = f_alg * local_elements / (kernel_ms * 1e-3) / 1e12
            out["roofline_fp64"] = {"bound": "mfma", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                                    "algorithmic_flops_per_element": f_alg,
                                    "note": "dense-form flops of SURVEY 8d over the same step time; the kernels execute fewer "
                                            "(sum factorisation, symmetric half): see DESIGN.md 4.1 for the issued fp64 work"}
        if residual_ms is not None:
            rj = elapsed / args.steps * 1e3
            out["per_newton_iteration"] = {"ms": rj + 2.0 * residual_ms, "residual_only_ms": residual_ms,
                                           "composition": "1 x (residual+Jacobian) + 2 x (residual) assemblies, the line search of "
                                                          "solvers/newton.cpp:142-190"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, material)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
