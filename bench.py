#!/usr/bin/env python3
"""Benchmark of the hot path: element-integrations/sec of the residual+Jacobian assembly
(one `AddDomainResidualAndGrad` over the whole mesh = one "step"), BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 workload: 128x128x16 p=2 neo-Hookean B-spline block (the configuration the north-star
target is quoted on; it fits one MI355X).  For N > 1 the SAME mesh is sharded in element slabs
across the ranks (strong scaling) and the shared-dof rows of the residual / Jacobian are
summed between neighbouring ranks over RCCL inside the timed region.

`--gpus N` without a launcher (no WORLD_SIZE in the environment): this process never touches a GPU; it checks
that N devices are visible, starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
127.0.0.1 rendezvous) and relays rank 0's line.  Under torchrun it is a rank.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernels, measured with events on the
launch stream) and `cpu_baseline` (the restated reference CPU path = oracle, timed on this
box's host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# the CPUs this process may use, read before any OpenMP runtime is loaded
try:
    _AFFINITY_AT_START = len(os.sched_getaffinity(0))
except (AttributeError, OSError):
    _AFFINITY_AT_START = os.cpu_count() or 1
# OMP_PROC_BIND is NOT set here: an OpenMP runtime loaded under it (import torch) pins the initial thread to one CPU, and
# every child process -- the ranks of --gpus N, their RCCL proxy threads -- inherits that one-CPU mask.  BASELINE.md 2's
# OMP_PROC_BIND=close belongs to the CPU baseline alone, which runs in a child process of its own (cpu_baseline_child).

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
    "cfg3_small": ((32, 32, 16), 3, "j2"),
    "cfg4_domain": ((96, 96, 12), 2, "neohookean"),
    # BASELINE configuration 4: the same block with the rigid sphere of SURVEY 8d pressing on its top face (the step is the
    # domain residual+Jacobian plus the contact residual+Jacobian; for N > 1 the contact faces follow their element slab)
    "cfg4": ((96, 96, 12), 2, "neohookean"),
    # the reference's other materials (tangent-record route): measured for DESIGN.md only
    "cfg2_stvk": ((64, 64, 8), 2, "stvk"),
    "cfg2_j2linear": ((64, 64, 8), 2, "j2linear"),
    "cfg2_j2simo": ((64, 64, 8), 2, "j2simo"),
    "cfg2_j2log": ((64, 64, 8), 2, "j2log"),
    "cfg5": ((256, 256, 32), 2, "neohookean"),
    # orientation experiments (same block, short axis first)
    "northstar_zfirst": ((16, 128, 128), 2, "neohookean"),
}


# algorithmic bytes per element integration
def b_alg(dim, p, grad=True, stateful=False):
    n_dof = (p + 1) ** dim
    n_tdof = n_dof * dim
    n_q = (p + 2) ** dim
    b = 8 * n_tdof + 8 * n_tdof + 4 * n_dof + 8 * n_q * (dim * dim + 1)
    if grad:
        b += 8 * n_tdof * n_tdof
    if stateful:
        b += 8 * 11 * n_q        # J2 family: 11 state doubles per point
    return b


def kernel_sources_sha():
    """hash of the sources of the domain assembly's kernels (what the recorded workloads run; the linear solver, the
    contact integrals and the interface pack kernels are separate translation units outside the timed step): recorded
    PMC numbers are quoted only for the kernels they were measured on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mimi_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f in ("krylov.hip", "contact.hip", "exchange.hip") or not os.path.isfile(os.path.join(d, f)):
            continue
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


def recorded_pmc(workload, world, grad, material):
    """HBM bytes per step and fp64-pipe occupancy from the committed rocprofv3 PMC passes (profiles/*_traffic.json:
    FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as the gfx950 guide prescribes, calibrated for this
    kernel's 8-byte accesses by scratch/fetch_calib.hip).  bench.py cannot run the profiler on itself, so these are
    RECORDED numbers: returned with their source, and dropped (None) when the kernel sources changed since."""
    key = f"{workload}/{material}/{'grad' if grad else 'residual'}/n{world}"
    for name in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f).get(key)
        except OSError:
            continue
        if not t:
            continue
        current = kernel_sources_sha()
        fresh = t.get("kernel_sources_sha") == current
        per_kernel = {k: 2048.0 * t["fetch_size_kb_raw"].get(k, 0.0) + 1024.0 * t["write_size_kb"].get(k, 0.0)
                      for k in t.get("fetch_size_kb_raw", {})} if "write_size_kb" in t else None
        return dict(bytes_per_step=t.get("bytes_per_step") if fresh else None,
                    pipe=t.get("pipe") if fresh else None,
                    per_kernel_bytes=per_kernel if fresh else None,
                    traffic_source=f"profiles/{name} ({t.get('source', 'rocprofv3 --pmc')})",
                    recorded_for_kernel_sources_sha=t.get("kernel_sources_sha"), current_kernel_sources_sha=current,
                    stale=not fresh)
    return None


def useful_flop(p, dim=3):
    """fp64 work of the sum-factorised tangent contraction per element, counted from its three stages (DESIGN.md 4.1 / 4.2;
    SURVEY 8d's F_alg prices the dense B^T A B form instead: 2.9 / 29 MFLOP).  With n = p + 1 nodes and q = p + 2 points per
    direction, per (i, j) block of the element matrix:
      S1 contracts q2:  9 (m, n) terms x q^2 points (q0, q1) x n^2 pairs (a2, b2) x q multiply-adds
      S2 contracts q1:  9 (m, n) terms x q points q0 x n^2 (a1, b1) x n^2 (a2, b2) x q multiply-adds
      S3 contracts q0:  4 table variants x n^2 (a0, b0) x n^4 pairs x q multiply-adds
    times 9 blocks, 2 flop per multiply-add: 0.513 MFLOP at p = 2, 2.84 MFLOP at p = 3 (the point work -- F, the material,
    the pulled-back tangent, the residual -- is < 4 % of that and left out)."""
    n, q = p + 1, p + 2
    return 2.0 * dim * dim * (9 * q ** 3 * n ** 2 + 9 * q ** 2 * n ** 4 + 4 * q * n ** 6)


# the chip's fp64 pipe: 1024 SIMDs x 32 flop per cycle x 2.4 GHz = 78.6 TFLOP/s
FP64_PEAK, N_SIMD, CLOCK_HZ = 78.6e12, 1024, 2.4e9


def binding_rooflines(p, elements, phase_ms, pmc):
    """"state the achieved fraction of whichever bound is binding, do not relabel".  Phase 1 (integration) is
    bound by the fp64 pipe, phase 2 (row gather) by HBM: per phase the live duration (events in the library) and, from the
    RECORDED counters of the same kernel sources (None when they changed since), the issued share of the fp64 pipe
    (matrix instructions at 64 cycles, vector instructions at 4 cycles each -- all vector instructions, so an upper
    bound on the fp64 share) and the HBM bytes over the phase's time."""
    if phase_ms is None:
        return None
    t1, t2 = phase_ms[0] * 1e-3, phase_ms[1] * 1e-3
    pipe = (pmc or {}).get("pipe") or {}
    raw = (pmc or {}).get("per_kernel_bytes") or {}
    integ = {k: v for k, v in pipe.items() if "gather" not in k and "p2_kernel" not in k}
    gath = [k for k in raw if "gather" in k or "p2_kernel" in k]
    cycles = sum(v["mfma_instructions_per_element"] * 64.0 + v["valu_instructions_per_element"] * 4.0 for v in integ.values())
    phase1 = {"resource": "fp64 pipe (matrix and vector fp64 instructions share it on gfx950)", "ms": phase_ms[0],
              "useful_flop_frac": useful_flop(p) * elements / t1 / FP64_PEAK if t1 > 0 else None,
              "useful_flop_per_element": useful_flop(p),
              "issued_frac": cycles * elements / (N_SIMD * CLOCK_HZ * t1) if integ and t1 > 0 else None,
              "issued_cycles_per_element": cycles if integ else None,
              "peak": "78.6 TFLOP/s = 1024 SIMDs x 32 flop/cycle x 2.4 GHz",
              "clock_note": "recorded engine clock under this load (rocm-smi every 3 s over 35-s runs, profiles/r05_clock_under_load.txt): "
                            + ("2.15-2.18 GHz at 1.30 kW -- the chip is at its power limit on this workload, the pipe's own "
                               "peak is 0.90 of the figure above and the issued share correspondingly higher" if p == 3 else
                               "2.35-2.38 GHz at 1.19 kW (0.98-0.99 of the 2.4 GHz the peak above assumes)")}
    bytes2 = sum(raw[k] for k in gath) if gath else None
    phase2 = {"resource": "hbm", "ms": phase_ms[1], "measured_bytes": bytes2,
              "frac": bytes2 / t2 / 8e12 if bytes2 and t2 > 0 else None, "peak": "8 TB/s"}
    return {"phase1": phase1, "phase2": phase2,
            "counters": "recorded rocprofv3 PMC passes of these kernel sources" if pipe else "none for these kernel sources"}


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


def kernel_description(p, material, path):
    if path != 1:
        return "general-table kernels: domain_general_kernel (+ general_gather_kernel for 64-node elements)"
    if p == 3:
        return ("tp3_point_kernel (material + tangent record + residual pieces) + tp3_contract_kernel (sum-factorised "
                "contraction on v_mfma_f64_16x16x4, one wave per element column x (i, j)) = phase 1; tp3_gather_kernel "
                "(row gather) = phase 2")
    if material == "neohookean":
        return "tensor_wgsym_kernel (integration, symmetric half) = phase 1; tensor_p2_kernel (row gather) = phase 2"
    if material == "j2":
        return "tensor_point_kernel (return mapping) + tensor_wgs_kernel (nine-block integration) = phase 1; tensor_p2_kernel = phase 2"
    return "tensor_point_kernel (material + tangent record) + tensor_wgs_kernel = phase 1; tensor_p2_kernel = phase 2"


# ------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def physical_cores():
    """(usable cores, description): distinct (socket, core) pairs of /proc/cpuinfo, clipped to the CPUs this process may
    run on (affinity mask) and to the cgroup CPU quota -- more OpenMP threads than that only get throttled"""
    pairs = set()
    try:
        with open("/proc/cpuinfo") as f:
            phys = core = None
            for line in f:
                key = line.split(":")[0].strip()
                if key == "physical id":
                    phys = line.split(":")[1].strip()
                elif key == "core id":
                    core = line.split(":")[1].strip()
                    pairs.add((phys, core))
    except OSError:
        pass
    physical = len(pairs) or (os.cpu_count() or 1)
    n, why = physical, f"{physical} physical cores"
    aff = _AFFINITY_AT_START
    if aff < n:
        n, why = aff, f"affinity mask of {aff} CPUs ({physical} physical cores)"
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                n, why = q, f"cgroup CPU quota of {q} CPUs ({physical} physical cores on the host)"
    except (OSError, ValueError):
        pass
    return max(1, n), why


def _oracle_material(material):
    from oracle import ref_path as rp
    if material in ("neohookean", "stvk"):
        return rp.make_material(material, 2100, 0.3)
    if material == "j2linear":
        return rp.make_material("j2linear", 2100, 0.3, isotropic_hardening=40.0, kinematic_hardening=25.0, sigma_y=70.0)
    return rp.make_material(material, 2100, 0.3, hardening=dict(kind="JohnsonCookTempRate", A=70, B=140, n=0.2835,
                            m=1.3558, eps0_dot=0.004, reference_temperature=20),
                            specific_heat=450, initial_temperature=20, melting_temperature=1500)


def _median_time(fn, warmup, calls, budget_s):
    for _ in range(warmup):
        fn()
    ts = []
    t_begin = time.perf_counter()
    while len(ts) < calls and (len(ts) < 3 or time.perf_counter() - t_begin < budget_s):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def cpu_baseline(p, material, threads=None, n_el=None, sweep=False):
    """The restated reference CPU path (oracle/ref_path.c: forward-FD element Jacobian, virtual-call-free material per
    point, contiguous element chunk per thread, per-thread full-size arrays zeroed per call + reduction pass, OpenMP) on
    a bounded sample of the workload, by BASELINE.md 2's protocol: threads = physical cores, OMP_PROC_BIND=close,
    3 warm-up + >= 10 timed assemblies, median."""
    from oracle import iga, ref_path as rp
    if not n_el and os.environ.get("MIMI_BENCH_CPU_SAMPLE"):          # (tests: a sample that takes seconds)
        n_el = [int(x) for x in os.environ["MIMI_BENCH_CPU_SAMPLE"].split("x")]
    n_el = tuple(n_el) if n_el else ((64, 64, 8) if p == 2 else (32, 32, 8))
    cores, cores_why = physical_cores()
    threads = threads or cores
    P = iga.Patch.block(n_el, p)
    D = rp.DomainOracle(P, _oracle_material(material), n_threads=threads)
    D.set_dt(0.5)
    rng = np.random.default_rng(20241008)
    u = 0.05 * rng.standard_normal(P.n_vdofs)
    u.reshape(-1, 3)[P.boundary_nodes(0, 0)] = 0.0
    r = np.zeros(P.n_vdofs)
    A = np.zeros(D.nnz)
    # per-thread arrays of the reference design: threads x (n_vdofs + nnz) doubles
    tl_gb = threads * (P.n_vdofs + D.nnz) * 8 / 1e9

    def timed(mode, nt, budget):
        D.n_threads = nt
        return _median_time(lambda: D.add_domain_residual_and_grad(u, 1.0, r, A, mode), 3, 10, budget)

    sys.stderr.write(f"bench.py: CPU baseline on {threads} threads ({'x'.join(map(str, n_el))} sample) ...\n")
    sys.stderr.flush()
    t_fd, n_fd = timed(rp.TANGENT_FD, threads, 20.0)
    t_ex, n_ex = timed(rp.TANGENT_EXACT, threads, 8.0)
    t_none, _ = timed(rp.TANGENT_NONE, threads, 3.0)
    out = dict(value=P.n_el / t_fd, unit="element-integrations/s", cores=threads, kind="port",
               sample=f"{'x'.join(map(str, n_el))} p={p} {material} block ({P.n_el} elements, nnz {D.nnz}), median of {n_fd} "
                      f"residual+Jacobian assemblies after 3 warm-up calls, reference forward-FD element Jacobian, OpenMP "
                      f"{threads} threads = {cores_why}, OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} "
                      f"(host: {os.cpu_count()} logical CPUs, {_cpu_model()}); per-thread arrays "
                      f"{tl_gb:.1f} GB in all",
               seconds_per_assembly=t_fd,
               analytic_tangent_value=P.n_el / t_ex,
               analytic_tangent_note=f"same restated path and threads with the oracle's analytic element tangent instead of the "
                                     f"reference's forward differences (median of {n_ex})",
               zero_and_reduce_seconds=t_none,
               zero_and_reduce_note="the same call with no elements: zeroing the per-thread n_vdofs + nnz arrays and the "
                                    "reduction pass over all of them (nonlinear_solid.cpp:117-121, nonlinear_base.hpp:112-151) "
                                    "-- the part of an assembly that grows with the thread count")
    if sweep:
        out["thread_sweep"] = {}
        for nt in sorted({1, 4, 8, 16, 32, 128, threads}):
            if nt > (os.cpu_count() or 1):
                continue
            sys.stderr.write(f"bench.py: CPU baseline sweep, {nt} threads ...\n")
            sys.stderr.flush()
            if nt == 1 and P.n_el > 4096:
                continue                      # one thread on the full sample takes minutes: see the small-sample sweep
            tf, _ = timed(rp.TANGENT_FD, nt, 12.0)
            te, _ = timed(rp.TANGENT_EXACT, nt, 6.0)
            tn, _ = timed(rp.TANGENT_NONE, nt, 2.0)
            out["thread_sweep"][str(nt)] = dict(fd=P.n_el / tf, analytic=P.n_el / te, zero_and_reduce_seconds=tn)
    rp.lib().oracle_release_thread_local()
    return out


def cpu_baseline_child(workload, sweep=False):
    """cpu_baseline() in a child process of its own with OMP_PROC_BIND=close: the binding (and the one-CPU affinity mask
    an OpenMP runtime leaves on the thread that loaded it) stays out of this process and of everything it starts.  The
    child imports numpy and the oracle only (no torch, no HIP)."""
    env = dict(os.environ, OMP_PROC_BIND="close")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload]
    if sweep:
        cmd.append("--cpu-sweep")
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=3000 if sweep else 900)
    if out.returncode != 0:
        raise RuntimeError(f"CPU baseline child exited with {out.returncode}")
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


# ------------------------------------------------------------------------------------------------
# rank launcher
# ------------------------------------------------------------------------------------------------
def visible_gpu_count():
    """GPUs this process would see, counted WITHOUT loading HIP (the launcher must not touch a GPU before it starts its
    ranks): KFD topology nodes with SIMDs, clipped by a *_VISIBLE_DEVICES list; no KFD, no GPU."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return 0
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """--gpus N without a launcher: N fresh processes, started before anything in this one touches a GPU."""
    backend = os.environ.get("MIMI_BENCH_BACKEND", "nccl")
    visible = visible_gpu_count()                # (sysfs: no HIP, no torch in this process)
    if backend == "nccl" and visible < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this box shows {visible}; "
                         f"(MIMI_BENCH_BACKEND=gloo rehearses the N-rank code path on fewer GPUs, timings meaningless)\n")
        return 2
    env = dict(os.environ)
    env.pop("OMP_PROC_BIND", None)               # (a bound OpenMP runtime would pin each rank's host threads to one CPU)
    env.update(WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as out0:       # rank 0's stdout (a file: nobody has to drain a pipe while we poll)
        for rank in range(args.gpus):
            e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                          stdout=out0 if rank == 0 else subprocess.DEVNULL))
        codes = supervise(procs)
        out0.seek(0)
        for line in out0.read().decode().splitlines():      # (gloo greets on stdout: keep stdout to the one JSON line)
            (sys.stdout if line.startswith("{") and not any(codes) else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        return 1
    return 0


def supervise(procs, poll_s=0.2, grace_s=5.0):
    """Wait for every rank; as soon as ONE exits non-zero the others are terminated (then killed): a rank that died at
    start-up would otherwise leave the rest inside init_process_group / their first collective until the process
    group's timeout.  Returns the exit codes.  (The children are fresh processes; nothing is re-executed.)"""
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            return codes
        if any(c not in (None, 0) for c in codes):
            bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
            sys.stderr.write(f"bench.py: rank(s) {bad} exited with {[codes[i] for i in bad]}; stopping the others\n")
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.monotonic() + grace_s
            for p in procs:
                try:
                    p.wait(timeout=max(0.0, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            return [p.returncode for p in procs]
        time.sleep(poll_s)


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def measure(args, workload, rank, world, local_rank, backend, with_extras, loopback=False):
    """time `args.steps` assemblies of one workload on this rank's slab; returns the result dict (rank 0) or None"""
    import torch
    import torch.distributed as dist
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, NonlinearSolid
    from mimi_amd import parallel

    dev = torch.device("cuda", local_rank)
    n_el, p, material = WORKLOADS[workload]
    t_setup = time.perf_counter()
    free0, _total = torch.cuda.mem_get_info(dev)
    if world > 1:
        # N > 1 (round 5): every rank holds its slab and p ghost layers either side as a patch of its OWN -- u, r, the matrix (the
        # local patch's whole structured pattern), the material state and the handle's set-up are of local size; nothing
        # of whole-patch size is built on the host or the device (parallel.SlabShard.localized)
        from mimi_amd.splines import PatchShape
        whole_shape = PatchShape.block(n_el, p)
        shard_g = parallel.SlabShard(whole_shape, None, rank, world)
        below, above = shard_g.ghost_layers()
        gb, ge = shard_g.element_box
        patch = mimi_amd.BSplinePatch.block_slab(n_el, p, shard_g.axis, gb[shard_g.axis] - below, ge[shard_g.axis] + above)
        pattern = CSRPattern.of_bspline_patch(patch, device=local_rank, on_device=True)
        shard = shard_g.localized(patch, pattern, ghost=(below, above))
    else:
        whole_shape = patch = mimi_amd.BSplinePatch.block(n_el, p)
        shard = parallel.SlabShard(patch, None, rank, world)
        pattern = CSRPattern.of_bspline_patch(patch, device=local_rank, on_device=True)
        shard.pattern = pattern
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
    # (p layers per side although owner mode could do with p // 2 and p - p // 2 -- overlap_boxes(mode="owner"): measured in
    # loop-back at N = 8, the thinner boundary launches cost more than the layers they move to the interior save:
    # 1.29 against 1.27 ms, cfg3 6.36 against 6.15 ms)
    # N > 1, two ways to have the interface rows on the wire while the rest is assembled (--scheme):
    #   "gather": ONE handle for the slab; its integration kernels run, the rows that leave the rank are gathered first, they
    #             travel while the other rows are gathered (two-phase tensor paths: NonlinearSolid.Integrate / Gather)
    #   "boundary": the element layers next to a neighbour as handles of their own, assembled first; the interior meanwhile
    scheme = "none"
    if world > 1 and not args.residual_only:
        scheme = args.scheme
    u_whole = synthetic_u(whole_shape, scale=0.01 if workload == "cfg4" else 0.05)     # (the workload's u: same numbers at every N)
    if world > 1:
        gnodes = shard.global_nodes()
        u_whole = np.ascontiguousarray(u_whole.reshape(-1, patch.dim)[gnodes].reshape(-1))
    u = torch.from_numpy(u_whole).to(dev)
    del u_whole
    integ = None
    if scheme == "gather":
        integ = make_integrator(shard.element_box)
        try:                                   # (handles off the two-phase tensor paths have the one-call form only)
            integ.Integrate(u)
            integ.Synchronize()
        except RuntimeError as exc:
            sys.stderr.write(f"bench.py: rank {rank}: no two-step assembly for this handle ({exc}); --scheme boundary\n")
            scheme, integ = "boundary", None
    boundary_boxes, interior_box = shard.overlap_boxes() if scheme == "boundary" else ([], shard.element_box)
    if integ is None:
        integ = make_integrator(interior_box)
    boundary = [make_integrator(b) for b in boundary_boxes]
    # the two boundary boxes of a middle rank are a few hundred element columns each -- half of the chip's workgroup slots:
    # they share no node, so the second one runs beside the first on its own stream
    side_stream = None
    if len(boundary) == 2 and shard.boxes_share_no_node(boundary_boxes):
        side_stream = torch.cuda.Stream(device=dev)
        boundary[1].SetStream(side_stream.cuda_stream)

    r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
    A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    exchange = parallel.InterfaceExchange(shard, r, A, dev, mode="owner", loopback=loopback) if world > 1 else None
    early = rest = None
    if scheme == "gather":
        early, rest = exchange.gather_windows()
    contact = None
    if workload == "cfg4":
        from mimi_amd.integrators import RigidSphere
        L = np.asarray(n_el, dtype=np.float64)             # (unit cells: the block's extent)
        R = 0.25 * L[0]
        c = 0.5 * L
        c[2] = L[2] + 0.9 * R
        body = RigidSphere(list(c), R, 1e4)
        if world > 1:
            contact = parallel.ShardedContact(shard, body, pattern, 2, 1, device=local_rank, loopback=loopback)
        else:
            from mimi_amd.integrators import MortarContact
            contact = MortarContact(body, "contact", pattern, patch, 2, 1, device=local_rank).Prepare()
            contact.body_ = body
        contact.SetStream(stream.cuda_stream)

    def step():
        if exchange:
            # interface rows hold exactly this step's sum (interior rows just keep accumulating)
            exchange.zero_interface(with_grad=not args.residual_only)
        if args.residual_only:
            integ.AddDomainResidual(u, r)
            if exchange:
                exchange.sum_residual()
        else:
            if scheme == "gather":
                integ.Integrate(u)
                for w in early:                      # the rows the neighbours wait for
                    integ.Gather(1.0, r, A, *w)
                if contact:          # (its rows on shared node planes must be in before they go on the wire)
                    contact.AddBoundaryResidualAndGrad(u, 1.0, r, A)
                ready = torch.cuda.Event()
                ready.record(stream)
                integ.Gather(1.0, r, A, *rest)       # ... travel while everything else is gathered
                exchange.start(True, ready=ready)
                exchange.finish()
            elif boundary:
                if side_stream:
                    side_stream.wait_stream(stream)
                for g in boundary:
                    g.AddDomainResidualAndGrad(u, 1.0, r, A)
                if side_stream:
                    stream.wait_stream(side_stream)
                if contact:          # (its rows on shared node planes must be in before they go on the wire)
                    contact.AddBoundaryResidualAndGrad(u, 1.0, r, A)
                # the interior kernels are enqueued BEFORE the host issues the sends: packing and the sends run on the
                # exchange's stream behind `ready`, beside the interior, and the GPU does not idle while the host talks to RCCL
                ready = torch.cuda.Event()
                ready.record(stream)
                integ.AddDomainResidualAndGrad(u, 1.0, r, A)
                exchange.start(True, ready=ready)
                exchange.finish()
            else:
                integ.AddDomainResidualAndGrad(u, 1.0, r, A)
                if contact:
                    contact.AddBoundaryResidualAndGrad(u, 1.0, r, A)
                if exchange:
                    exchange.sum_residual_and_grad()

    steps, warmup = args.steps, args.warmup
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(steps):
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
    kernel_ms = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]))

    # N > 1 (unless --no-check): after the timed region, the rows this rank owns after the exchange against a whole-patch
    # assembly on this rank's own GPU -- the evidence that what travelled over RCCL is right
    check = None
    if not args.no_check and world > 1 and not args.residual_only and not loopback:
        r.zero_()
        A.zero_()
        step()                                   # (collective: every rank gets here)
        torch.cuda.synchronize()

        def compare():
            gpatch = mimi_amd.BSplinePatch.block(n_el, p)
            full = CSRPattern.of_bspline_patch(gpatch, device=local_rank, on_device=True)
            whole = NonlinearSolid("domain", make_material(material), full, patch=gpatch, device=local_rank).Prepare()
            whole.dt_ = 0.5
            whole.SetStream(stream.cuda_stream)
            u_w = torch.from_numpy(synthetic_u(gpatch, scale=0.01 if workload == "cfg4" else 0.05)).to(dev)
            r_w = torch.zeros(gpatch.n_vdofs, dtype=torch.float64, device=dev)
            A_w = torch.zeros(full.nnz, dtype=torch.float64, device=dev)
            whole.AddDomainResidualAndGrad(u_w, 1.0, r_w, A_w)
            if contact:
                from mimi_amd.integrators import MortarContact
                whole_c = MortarContact(contact.body_, "contact", full, gpatch, 2, 1, device=local_rank).Prepare()
                whole_c.SetStream(stream.cuda_stream)
                whole_c.AddBoundaryResidualAndGrad(u_w, 1.0, r_w, A_w)
                whole_c.Synchronize()
            whole.Synchronize()
            torch.cuda.synchronize()
            # the rows this rank owns: local rows and the whole patch's rows of the same nodes (an owned row is complete in
            # its columns on this rank -- the ghost layers -- so the two runs have the same length and order)
            planes = torch.tensor(exchange.owned_node_planes(), device=dev)
            mi_axis = torch.from_numpy(patch.node_multi_index()[shard.axis]).to(dev)
            nodes = torch.nonzero(torch.isin(mi_axis, planes)).reshape(-1)
            gn = torch.from_numpy(shard.global_nodes()).to(dev)
            rows = (nodes[:, None] * 3 + torch.arange(3, device=dev)[None, :]).reshape(-1)
            grows = (gn[nodes][:, None] * 3 + torch.arange(3, device=dev)[None, :]).reshape(-1)
            er = float((r[rows] - r_w[grows]).abs().max() / r_w.abs().max())

            def positions(rowptr, which):
                # positions in a value array of all entries of the rows `which`, row after row
                start = rowptr[which]
                length = rowptr[which + 1] - start
                offs = torch.cumsum(length, 0) - length
                return torch.repeat_interleave(start - offs, length) + torch.arange(int(length.sum()), device=dev)

            mine, ref = positions(pattern.rowptr, rows), positions(full.rowptr, grows)
            if mine.numel() != ref.numel() or mine.numel() == 0:
                raise RuntimeError("owned rows of the local and the whole pattern differ")
            return er, float((A[mine] - A_w[ref]).abs().max() / A_w.abs().max())

        failed = ""
        try:
            er, eA = compare()
        except Exception as exc:            # (the measurement above stands; the line says that the check did not run)
            er = eA = 0.0
            failed = f"rank {rank}: {type(exc).__name__}: {exc}"
            sys.stderr.write("bench.py: check failed to run -- " + failed + "\n")
        errs = torch.tensor([er, eA, 1.0 if failed else 0.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(errs, op=dist.ReduceOp.MAX)
        what = "owned rows of every rank after the exchange vs a whole-patch assembly on the same GPU (max over ranks)"
        if float(errs[2]) > 0:
            check = dict(error="the comparison did not run on every rank" + (": " + failed if failed else ""), what=what)
        else:
            check = dict(residual_rel_err=float(errs[0]), tangent_rel_err=float(errs[1]), what=what)
        torch.cuda.empty_cache()

    phase_ms = residual_ms = post_ms = None
    if with_extras and world == 1 and not args.residual_only:
        # per-kernel durations, live: events inside the library around phase 1 and phase 2 (after the timed region)
        if integ.path_ == 1:
            integ.SetPhaseTiming(True)
            acc1 = acc2 = 0.0
            for _ in range(5):
                integ.AddDomainResidualAndGrad(u, 1.0, r, A)
                a1, a2 = integ.PhaseMs()
                acc1 += a1
                acc2 += a2
            integ.SetPhaseTiming(False)
            phase_ms = (acc1 / 5, acc2 / 5)
        # "also reported per full Newton iteration (= 1 x (R+J) + 2 x (R) assemblies, newton.cpp:142-190)"
        for _ in range(3):
            integ.AddDomainResidual(u, r)
        integ.Synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            integ.AddDomainResidual(u, r)
        integ.Synchronize()
        residual_ms = (time.perf_counter() - t1) / 10 * 1e3
        # DomainPostTimeAdvance (nonlinear_solid.cpp:179-199): once per time step the converged state is committed -- for
        # BASELINE configuration 3 ("implicit dynamics") a return mapping at every point.  Timed last (it changes the
        # state), from the virgin state each time (a commit on top of an already committed u finds nothing to return).
        if material not in ("neohookean", "stvk"):
            cur = torch.cuda.current_stream(dev)
            acc = 0.0
            for k in range(4):
                integ.ResetState()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                integ.DomainPostTimeAdvance(u)
                e1.record(cur)
                e1.synchronize()
                if k:
                    acc += e0.elapsed_time(e1)
            post_ms = acc / 3
            integ.ResetState()

    result = None
    if rank == 0 or loopback:
        n_elements = whole_shape.n_elements
        grad = not args.residual_only
        stateful = material not in ("neohookean", "stvk")
        balg = b_alg(patch.dim, p, grad=grad, stateful=stateful)
        local_elements = integ.n_elements_ + sum(g.n_elements_ for g in boundary)
        achieved = balg * local_elements / (kernel_ms * 1e-3) / 1e9
        pmc = recorded_pmc(workload, world, grad, material)
        result = dict(
            value=n_elements * steps / elapsed, ms_per_step=elapsed / steps * 1e3, n_elements=n_elements,
            config={"workload": f"{'x'.join(map(str, n_el))} p={p} {material} B-spline block, "
                                f"{n_elements} elements, n_q={(p + 2) ** patch.dim}, nnz={pattern.nnz}",
                    "name": workload,
                    "parallelism": f"element slabs x{world}" + (", interface rows summed on their owner rank; every rank holds its "
                                   "slab and p ghost element layers either side as a patch of its own: u, r, matrix, state "
                                   f"and set-up of local size ({patch.n_vdofs} of {whole_shape.n_vdofs} dofs, {pattern.nnz} stored entries)"
                                   if world > 1 else "")
                                   + (", exchange overlapped with the interior elements" if boundary else "")
                                   + (", rows that leave the rank gathered first and sent while the others are gathered"
                                      if scheme == "gather" else "")
                                   + (f", messages trimmed to the entries the sender's elements can have written "
                                      f"({max(s['sidx'].numel() + s['srows'].numel() for s in exchange.sides) * 8 / 1e6:.1f} MB per neighbour and direction)"
                                      if exchange is not None and exchange.trim and exchange.sides else ""),
                    "kernel_path": "tensor" if integ.path_ == 1 else "general",
                    "u": f"{0.01 if workload == 'cfg4' else 0.05}*N(0,1), seed 20241008, face x=0 clamped"},
            roofline={"bound": "hbm", "bound_note": "the HBM roofline of SURVEY 8d's algorithmic bytes, as the contract asks; the "
                      "resource that actually binds each phase is in `binding` (phase 1: fp64 pipe, phase 2: HBM)",
                      "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                      "binding": binding_rooflines(p, local_elements, phase_ms, pmc),
                      "traffic": pmc["bytes_per_step"] if pmc else None,
                      "traffic_source": None if not pmc else {k: pmc[k] for k in ("traffic_source", "recorded_for_kernel_sources_sha",
                                                                                   "current_kernel_sources_sha", "stale")},
                      "kernel": "one step = " + kernel_description(p, material, integ.path_) + "; avg_launch_ms is their sum on "
                                "rank 0, measured with events on the launch stream",
                      "algorithmic_bytes_per_element": balg, "elements_per_launch": local_elements,
                      "avg_launch_ms": kernel_ms,
                      "phase_ms": None if phase_ms is None else
                      {"phase1_integration_kernels": phase_ms[0], "phase2_row_gather": phase_ms[1],
                       "how": "HIP events recorded by the library on the launch stream around the kernels of each phase, "
                              "mean of 5 assemblies after the timed region"}},
            check=check)
        import resource
        free1, _total = torch.cuda.mem_get_info(dev)
        result["rank_footprint"] = {"setup_s": setup_s, "host_peak_rss_gib": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20,
                                    "device_gib_in_use_after_the_run": (free0 - free1) / 2 ** 30,
                                    "torch_peak_allocated_gib": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
                                    "what": "this rank: seconds from the first patch object to the first step; peak resident host "
                                            "memory of the process; device memory in use (library + torch) after the timed run"}
        if pmc and pmc.get("pipe"):
            result["fp64_pipe"] = {"per_kernel": pmc["pipe"], "source": pmc["traffic_source"],
                                   "note": "rocprofv3 PMC per kernel (recorded, not live): mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES over "
                                           "the kernel's cycles per SIMD (the fp64 matrix and vector instructions share one pipe on "
                                           "gfx950: DESIGN.md 4.1), instruction counts per element, and where the wave-cycles go "
                                           "(issuing / issue stalls / waitcnt)"}
        if residual_ms is not None:
            rj = elapsed / steps * 1e3
            result["per_newton_iteration"] = {"ms": rj + 2.0 * residual_ms, "residual_only_ms": residual_ms,
                                              "composition": "1 x (residual+Jacobian) + 2 x (residual) assemblies, the line search "
                                                             "of solvers/newton.cpp:142-190"}
            # the residual-only assembly against ITS roofline (HBM-bound, B_alg without the n_tdof^2 term)
            b_res = b_alg(patch.dim, p, grad=False, stateful=stateful)
            ach = b_res * local_elements / (residual_ms * 1e-3) / 1e9
            # what binds it, from the RECORDED counters of these kernel sources (None when they changed since): the vector
            # instructions the assembly issues per element against the chip's issue capacity over the measured time -- the J2
            # return mapping makes cfg3's residual-only assembly an fp64-vector-issue problem, not an HBM one
            pmc_r = recorded_pmc(workload, world, False, material)
            pipe_r = (pmc_r or {}).get("pipe") or {}
            valu = sum(v["valu_instructions_per_element"] for v in pipe_r.values()) if pipe_r else None
            binding_r = None if valu is None else {
                "resource": "vector instruction issue (fp64 pipe)", "valu_instructions_per_element": valu,
                "issued_frac": valu * 4.0 * local_elements / (N_SIMD * CLOCK_HZ * residual_ms * 1e-3),
                "issued_frac_at_4.9_cycles": valu * 4.9 * local_elements / (N_SIMD * CLOCK_HZ * residual_ms * 1e-3),
                "how": "recorded SQ_INSTS_VALU per element x 4 cycles (the pipe's rate; 4.9 = what one wave's instruction takes, "
                       "DESIGN 4.2) x elements / (1024 SIMDs x 2.4 GHz x the measured time)",
                "measured_bytes": pmc_r.get("bytes_per_step"), "source": pmc_r.get("traffic_source")}
            result["residual_only"] = {"ms": residual_ms, "value": n_elements / (residual_ms * 1e-3),
                                       "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                                                    "frac": ach / 8000.0, "algorithmic_bytes_per_element": b_res,
                                                    "binding": binding_r,
                                                    "how": "mean of 10 AddDomainResidual calls after the timed region, host clock "
                                                           "around a device synchronisation"}}
        if post_ms is not None:
            result["post_time_advance_ms"] = post_ms
    # release this workload's device memory before the next one
    del integ, boundary, exchange, u, r, A, pattern, contact
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return result


def _rccl_options():
    """RCCL's stream at high priority: streams of equal priority share a few hardware queues, and a send / recv kernel
    queued behind the interior kernels of the compute stream would not overlap with them at all (seen in the kernel
    trace of `--rehearse-rccl`); high-priority streams have queues of their own"""
    import torch.distributed as dist
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    return opts


def _emit(fd, obj):
    """the ONE JSON line, on the process's original stdout"""
    os.write(fd, (json.dumps(obj) + "\n").encode())


def run_rank(args):
    # stdout carries one JSON line and nothing else: RCCL prints a version banner on the stdout of rank 0 when its
    # communicator comes up, so everything but that line is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # MIMI_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (interface rows are
    # staged through the host, ranks share the visible GPUs); the measured configuration is "nccl" = RCCL
    backend = os.environ.get("MIMI_BENCH_BACKEND", "nccl")
    n_visible = torch.cuda.device_count()
    if backend == "gloo":
        local_rank %= max(n_visible, 1)
    elif local_rank >= n_visible:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_visible} GPUs are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_ranks = 1
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a short timeout: a rank that never arrives fails the others within minutes, not after the default half hour
        import datetime
        patience = datetime.timedelta(seconds=int(os.environ.get("MIMI_BENCH_PG_TIMEOUT", "240")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, pg_options=_rccl_options(), timeout=patience)
        else:
            dist.init_process_group(backend, timeout=patience)
        # one sum over the communicator before anything is timed: every rank is there and the transport works
        ones = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        comm_ranks = int(ones.item())
        if comm_ranks != world:
            raise SystemExit(f"communicator sums to {comm_ranks} ranks, expected {world}")

    if args.rehearse_rccl:
        # one slab of an N-rank job on this one GPU, exchange over a one-rank RCCL communicator (sends to itself)
        if world != 1:
            raise SystemExit("--rehearse-rccl: one process")
        n_fake = args.rehearse_rccl
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=_rccl_options())
        res = measure(args, args.workload, n_fake // 2, n_fake, local_rank, "nccl", with_extras=False, loopback=True)
        out = {"rehearsal": f"rank {n_fake // 2} of {n_fake} alone on one GPU: its slab, its boundary / interior split, pack, "
                            "RCCL send / recv (to itself: device-local copies, not xGMI), unpack; `value` is the whole job's "
                            "rate IF every rank took this long -- not a measurement of an N-GPU run",
               "value": res["value"], "unit": "element-integrations/s", "n_gpus": 1, "ranks_rehearsed": n_fake,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "config": res["config"],
               "elements_on_this_rank": res["roofline"]["elements_per_launch"], "rank_footprint": res.get("rank_footprint")}
        _emit(json_fd, out)
        dist.destroy_process_group()
        return

    main = measure(args, args.workload, rank, world, local_rank, backend, with_extras=True)
    other = {}
    if world == 1 and not args.residual_only and not args.no_other_configs and args.workload == "northstar":
        # BASELINE configuration 3 in the same run (fewer steps: 45 ms each), so that its figure is measured, not copied
        saved = (args.steps, args.warmup)
        args.steps, args.warmup = min(args.steps, 5), min(args.warmup, 2)
        try:
            c3 = measure(args, "cfg3", rank, world, local_rank, backend, with_extras=True)
            other["cfg3"] = {k: c3[k] for k in ("value", "ms_per_step", "config", "per_newton_iteration", "residual_only",
                                                "post_time_advance_ms") if k in c3}
            other["cfg3"]["steps"] = args.steps
            other["cfg3"]["roofline"] = {k: c3["roofline"][k] for k in ("achieved", "frac", "algorithmic_bytes_per_element", "phase_ms", "kernel")}
        except Exception as exc:        # (e.g. a smaller GPU: the scratch of cfg3 needs about 70 GB)
            other["cfg3"] = {"error": str(exc)}
        args.steps, args.warmup = saved

    if rank == 0:
        out = {
            "metric": "element-integrations/sec (residual+Jacobian assembly)" if not args.residual_only
                      else "element-integrations/sec (residual-only assembly)",
            "value": main["value"], "unit": "element-integrations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": main["config"], "roofline": main["roofline"],
            "communicator": {"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "ranks": comm_ranks},
        }
        for k in ("fp64_pipe", "per_newton_iteration", "residual_only", "post_time_advance_ms", "check"):
            if main.get(k) is not None:
                out[k] = main[k]
        if other:
            out["other_configs"] = other
        if world == 1 and not args.no_cpu_baseline:
            # a slow or failing host baseline must not throw the finished GPU measurement away
            try:
                out["cpu_baseline"] = cpu_baseline_child(args.workload, sweep=args.cpu_sweep)
                out["cpu_baseline"]["gpu_over_cpu"] = main["value"] / out["cpu_baseline"]["value"]
                # BASELINE.md holds no published figure for this metric; its section 2 names the number to compare with:
                # the restated reference CPU path timed on this box (above), reference forward-FD Jacobian
                out["vs_baseline"] = main["value"] / out["cpu_baseline"]["value"]
                out["vs_baseline_note"] = ("value / cpu_baseline.value (BASELINE.md 2: no published number exists; the "
                                           "restated reference OpenMP path on this box's host cores is the baseline)")
            except (RuntimeError, subprocess.TimeoutExpired, ValueError, IndexError, KeyError, OSError) as exc:
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
                out["vs_baseline"] = None
        else:
            out["cpu_baseline"] = None
        _emit(json_fd, out)
    if world > 1:
        dist.destroy_process_group()
    # a wrong exchange must not exit 0 (N > 1): the check ran on every rank and the owned rows agree
    bad = None
    chk = main.get("check") if main else None
    if rank == 0 and world > 1 and not args.no_check and not args.residual_only:
        if not chk or "error" in chk:
            bad = f"the N > 1 check did not run: {chk}"
        elif not (chk["residual_rel_err"] < 1e-10 and chk["tangent_rel_err"] < 1e-10):
            bad = f"owned rows differ from the whole-patch assembly: {chk}"
    if bad:
        sys.stderr.write("bench.py: " + bad + "\n")
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rehearse-rccl", type=int, default=0, metavar="N",
                    help="time the step of the middle rank of an N-rank job on this one GPU, exchange over RCCL to itself")
    ap.add_argument("--workload", default=os.environ.get("MIMI_BENCH_WORKLOAD", "northstar"), choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="print the CPU baseline of the workload as one JSON line and exit (no GPU, no torch: what "
                         "cpu_baseline_child runs under OMP_PROC_BIND=close)")
    ap.add_argument("--cpu-sweep", action="store_true", help="add a 1/32/64/128-thread sweep of the CPU baseline (minutes)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg3 measurement that follows the north-star one at N = 1")
    ap.add_argument("--scheme", default="gather", choices=["gather", "boundary", "none"],
                    help="N > 1: how the exchange is overlapped with the assembly (see measure()); none = no overlap")
    ap.add_argument("--no-check", action="store_true",
                    help="N > 1: skip the comparison of the owned rows with a whole-patch assembly that follows the timed region")
    ap.add_argument("--check", action="store_true", help="(accepted for older command lines: the check is on by default)")
    ap.add_argument("--residual-only", action="store_true", help="time AddDomainResidual instead (not the headline metric)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.cpu_baseline_only:
        _, p, material = WORKLOADS[args.workload]
        print(json.dumps(cpu_baseline(p, material, sweep=args.cpu_sweep)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
