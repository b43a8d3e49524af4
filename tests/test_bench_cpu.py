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
