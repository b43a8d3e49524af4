"""bench.py end to end on the GPU box, at sizes that take seconds: the default line's contract (one JSON line on stdout, the
roofline / cpu_baseline objects) on the small BASELINE configuration, and the one-GPU rehearsal of a rank of an N-GPU job
over a real RCCL communicator."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MIMI_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:2000]            # stdout carries the one JSON line and nothing else
    return json.loads(lines[0])


def test_bench_line_contract_on_the_small_configuration():
    d = run_bench("--workload", "cfg2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f64" and d["config"]["name"] == "cfg2"
    assert d["config"]["kernel_path"] == "tensor"
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and 0.0 < roof["frac"] < 1.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    # whole-job rate = elements x steps / time
    assert abs(d["value"] - 64 * 64 * 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_rehearsal_of_one_rank_over_rccl():
    d = run_bench("--rehearse-rccl", "4", "--workload", "cfg2", "--steps", "3", "--warmup", "1")
    assert d["ranks_rehearsed"] == 4 and d["n_gpus"] == 1 and "rehearsal" in d
    assert d["elements_on_this_rank"] == 64 * 64 * 8 // 4
    assert d["ms_per_step"] > 0.0


@pytest.mark.parametrize("launcher", ["own", "torchrun"])
def test_two_ranks_sharing_the_gpu_over_gloo(launcher):
    """the N > 1 code path of bench.py end to end (slabs, two-step assembly, interface exchange, barrier + max-over-ranks
    timing, the owned-rows check against a whole-patch assembly) with both ways of starting the ranks: bench.py's own
    launcher and the driver's `python -m torch.distributed.run`.  Two ranks share the one GPU and exchange over gloo
    (MIMI_BENCH_BACKEND=gloo): the timings mean nothing, the line's contract and the check do."""
    import socket
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MIMI_BENCH_BACKEND"] = "gloo"
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg2", "--steps", "2", "--warmup", "1"]
    if launcher == "own":
        cmd = [sys.executable] + tail
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["name"] == "cfg2"
    assert abs(d["value"] - 64 * 64 * 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    check = d["check"]
    assert check["residual_rel_err"] < 1e-12 and check["tangent_rel_err"] < 1e-12
