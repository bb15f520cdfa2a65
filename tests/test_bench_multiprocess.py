"""The N>1 path of bench.py (one process per GPU, replicas only, barrier + max-over-ranks) run
with world_size 2 on the CPU: gloo backend, the oracle standing in for the HIP solver."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--K", "128", "--T", "20", "--selftest-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE json line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["vs_baseline"] is None
    # the timed block is repeated off the headline number: the first block is `ms_per_step`, all give the spread
    assert d["repeats"] == 10 and d["min_ms_per_step"] <= d["median_ms_per_step"] <= d["max_ms_per_step"]
    assert d["min_ms_per_step"] <= d["ms_per_step"] <= d["max_ms_per_step"]
    # whole-job aggregate: units of ALL ranks over the max-over-ranks time
    assert abs(d["value"] - 128 * 3 * 2 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]
    a, b = d["instances"]
    assert a["rank"] == 0 and b["rank"] == 1
    # independent instances: distinct costmaps, start states and (seeded) solutions
    assert a["map_checksum"] != b["map_checksum"] and a["start_state"] != b["start_state"] and a["U0"] != b["U0"]
    assert "NOT a benchmark" in d["data"]


def test_bench_refuses_world_size_mismatch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-cpu", "--steps", "1",
                        "--warmup", "0", "--K", "64", "--T", "10"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
