"""bench.py pieces that need no GPU: the rank launcher refuses to start without enough devices, the algorithmic byte
counts are SURVEY 8d's, the usable-core detection returns something sane."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_refuses_without_enough_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.pop("MIMI_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2
    assert "needs 64 visible GPUs" in out.stderr and out.stdout.strip() == ""


def test_algorithmic_bytes_and_cores():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.b_alg(3, 2) == 59012 and bench.b_alg(3, 2, grad=False) == 6524
    assert bench.b_alg(3, 3) == 308240 and bench.b_alg(3, 3, stateful=True) == 319240
    n, why = bench.physical_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and isinstance(why, str)
    assert len(bench.kernel_sources_sha()) == 16


def test_importing_bench_leaves_the_affinity_mask_alone():
    """OMP_PROC_BIND set at import pinned the process (and every rank it starts) to one CPU."""
    code = ("import os, sys; sys.path.insert(0, %r); before = os.sched_getaffinity(0); import bench, torch; "
            "assert os.sched_getaffinity(0) == before, (before, os.sched_getaffinity(0)); "
            "assert 'OMP_PROC_BIND' not in os.environ" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "OMP_PROC_BIND"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr


def test_cpu_baseline_runs_in_a_child_under_proc_bind():
    sys.path.insert(0, ROOT)
    import bench
    before = os.sched_getaffinity(0)
    os.environ["MIMI_BENCH_CPU_SAMPLE"] = "4x4x2"
    try:
        res = bench.cpu_baseline_child("cfg1")
    finally:
        del os.environ["MIMI_BENCH_CPU_SAMPLE"]
    assert res["kind"] == "port" and res["value"] > 0 and "OMP_PROC_BIND=close" in res["sample"]
    assert os.sched_getaffinity(0) == before


def test_supervisor_stops_the_other_ranks_when_one_dies():
    """a rank that dies at start-up must not leave the others waiting in their first collective.
    Two gloo ranks; rank 1 exits 7 before it joins; the supervisor has to return non-zero within seconds."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    port = bench._free_port()
    rank_code = (
        "import os, sys, datetime\n"
        "if os.environ['RANK'] == '1':\n"
        "    sys.exit(7)\n"
        "import torch.distributed as dist\n"
        "dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=600))\n"
        "dist.barrier()\n")
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    t0 = time.monotonic()
    procs = [subprocess.Popen([sys.executable, "-c", rank_code], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for r in range(2)]
    codes = bench.supervise(procs)
    assert time.monotonic() - t0 < 30.0
    assert codes[1] == 7 and codes[0] not in (0, None)


def test_binding_rooflines_use_the_live_phase_times():
    sys.path.insert(0, ROOT)
    import bench
    pmc = dict(pipe={"tensor_wgsym_kernel<0>": dict(mfma_instructions_per_element=234.0, valu_instructions_per_element=4694.0),
                     "tensor_p2_kernel": dict(mfma_instructions_per_element=0.0, valu_instructions_per_element=1851.0)},
               per_kernel_bytes={"tensor_wgsym_kernel<0>": 9.5e9, "tensor_p2_kernel": 14.0e9})
    b = bench.binding_rooflines(2, 262144, (4.79, 2.77), pmc)
    assert abs(b["phase1"]["issued_cycles_per_element"] - (234 * 64 + 4694 * 4)) < 1e-9
    assert 0.7 < b["phase1"]["issued_frac"] < 0.9 and 0.3 < b["phase1"]["useful_flop_frac"] < 0.4
    assert abs(b["phase2"]["frac"] - 14.0e9 / 2.77e-3 / 8e12) < 1e-12
    assert bench.binding_rooflines(2, 1, None, pmc) is None
    none = bench.binding_rooflines(2, 262144, (4.79, 2.77), None)
    assert none["phase1"]["issued_frac"] is None and none["phase2"]["frac"] is None


def test_algorithmic_counts_of_survey_8d():
    """SURVEY 8d's per-element figures the roofline objects are priced with, and the useful-flop count derived from the three
    contraction stages (bench.useful_flop): 59 012 / 6 524 B at p = 2, 308 240 (+ 11 000 J2 state) / 13 328 B at p = 3;
    0.513 / 2.84 MFLOP."""
    import bench
    assert bench.b_alg(3, 2, grad=True) == 59012 and bench.b_alg(3, 2, grad=False) == 6524
    assert bench.b_alg(3, 3, grad=True) == 308240 and bench.b_alg(3, 3, grad=False) == 13328
    assert bench.b_alg(3, 3, grad=True, stateful=True) == 319240
    assert bench.useful_flop(2) == 513216.0 and bench.useful_flop(3) == 2835360.0
