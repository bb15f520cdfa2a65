"""The N>1 path of bench.py (one process per GPU, replicas only, barrier + max-over-ranks) run
with world_size 2 on the CPU: gloo backend, the oracle standing in for the HIP solver."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_block_basis(d):
    """`value` is built from the MAX over ranks of every rank's OWN time for a block (no closing barrier inside); the
    bracket closed behind the barrier is kept beside it and can only be longer."""
    inst = d["instances"]
    for i, b in enumerate(d["blocks_ms_per_step"]):
        own = max(r["own_blocks_ms_per_step"][i] for r in inst)
        assert abs(b - own) <= 1e-9 * max(1.0, own), (i, b, own)
        assert d["blocks_ms_per_step_incl_closing_barrier"][i] >= b
    assert d["ms_per_step_incl_closing_barrier"] >= d["ms_per_step"]
    assert "closing barrier" in d["value_basis"]


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
    # the timed block is repeated: the MEDIAN of the blocks is `ms_per_step`, the first one `first_block_ms_per_step`
    assert d["repeats"] == 10 and d["min_ms_per_step"] <= d["median_ms_per_step"] <= d["max_ms_per_step"]
    assert d["ms_per_step"] == d["median_ms_per_step"] and d["first_block_ms_per_step"] == d["blocks_ms_per_step"][0]
    _check_block_basis(d)
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


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_real_solver_on_one_gpu():
    """The driver's N > 1 launch with the REAL solver: two ranks under torch.distributed.run, both mapped to
    GPU 0 (--devices 0,0), process group gloo (RCCL refuses two ranks on one device; the group only carries
    barrier / max / gather -- there is no collective on the data path, so nothing else differs from the 8-GPU
    launch: rank-0 build + barrier, capi.Solver(device), all_gather_object, one JSON line)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29613", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "5", "--dist-backend", "gloo", "--devices", "0,0",
           "--repeats", "3", "--latency-solves", "20", "--event-solves", "8", "--prime-ms", "50"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["config"]["process_group"] == "gloo" and "row8w" in d["config"]["rollout_variant"]
    assert abs(d["value"] - 4096 * 20 * 2 / (d["ms_per_step"] * 20 / 1e3)) < 1e-6 * d["value"]
    a, b = d["instances"]
    assert (a["rank"], b["rank"]) == (0, 1) and a["device"] == b["device"] == 0
    assert a["map_checksum"] != b["map_checksum"] and a["start_state"] != b["start_state"] and a["U0"] != b["U0"]
    for inst in (a, b):  # every rank's own time for the timed block (per-GPU spread on the real node)
        assert 0.0 < inst["own_ms_per_step"] <= d["first_block_ms_per_step"] * 1.001  # (own: the first timed block)
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d  # the CPU leg runs at N = 1 only
    _check_block_basis(d)
    for inst in (a, b):  # every rank's own rollout kernel against the roofline (rollouts/s AND achieved GB/s per GPU)
        assert inst["roofline"]["achieved_GBps"] > 0 and 0 < inst["roofline"]["frac_of_f32_peak"] < 1


def test_bench_eight_ranks_gloo():
    """The shape of the driver's 8-GPU launch on the CPU: eight ranks, eight DISTINCT instances (BASELINE configs[4]),
    one JSON line, every rank's own time present."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
           "--gpus", "8", "--steps", "2", "--warmup", "1", "--K", "64", "--T", "10", "--repeats", "2", "--selftest-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and len(d["instances"]) == 8 and [i["rank"] for i in d["instances"]] == list(range(8))
    assert len({i["map_checksum"] for i in d["instances"]}) == 8 and len({tuple(i["start_state"]) for i in d["instances"]}) == 8
    assert all(i["own_ms_per_step"] > 0 for i in d["instances"])
    assert abs(d["value"] - 64 * 2 * 8 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    _check_block_basis(d)


@pytest.mark.gpu
def test_bench_four_ranks_real_solver_on_one_gpu():
    """Rehearsal of the many-rank launch with the REAL solver: four ranks (the box admits at most six processes on its card,
    and the test runner itself is one of them; the 8-GPU node runs eight, one per GPU) mapped to GPU 0, gloo process group:
    four distinct instances, four host threads polling their result blocks at once, one JSON line.  The eight-rank shape
    runs on the CPU (test_bench_eight_ranks_gloo)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
           "--master-addr", "127.0.0.1", "--master-port", "29619", os.path.join(ROOT, "bench.py"),
           "--gpus", "4", "--steps", "10", "--warmup", "3", "--dist-backend", "gloo", "--devices", "0,0,0,0",
           "--repeats", "2", "--latency-solves", "0", "--event-solves", "4", "--prime-ms", "20", "--sustained-s", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["process_group"] == "gloo" and d["data"] == "synthetic"
    inst = d["instances"]
    assert [i["rank"] for i in inst] == list(range(4)) and all(i["device"] == 0 for i in inst)
    assert len({i["map_checksum"] for i in inst}) == 4 and len({tuple(i["U0"]) for i in inst}) == 4
    assert all(0.0 < i["own_ms_per_step"] <= d["first_block_ms_per_step"] * 1.001 for i in inst)
    assert abs(d["value"] - 4096 * 10 * 4 / (d["ms_per_step"] * 10 / 1e3)) < 1e-6 * d["value"]
    assert d["cold"] is not None and d["cold"]["ms_per_step"] > 0
    _check_block_basis(d)
    assert all(i["roofline"]["achieved_GBps"] > 0 and i["roofline"]["rollout_kernel_ms"] > 0 for i in inst)


@pytest.mark.gpu
def test_bench_rccl_process_group_smoke_at_world_size_one():
    """bench.py's `nccl` (= RCCL) path -- init_process_group(device_id=...), barrier, all_reduce(MAX) of a CUDA tensor,
    all_gather_object -- executed on a gfx950 box before any solver call, at world size 1 (--force-process-group): the
    first 8-GPU run is then not the first time these lines run.  RCCL refuses two ranks on one device, so the many-rank
    rehearsals above use gloo."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29621", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3",
           "--force-process-group", "--dist-backend", "nccl", "--repeats", "2", "--latency-solves", "0", "--event-solves", "4",
           "--prime-ms", "20", "--sustained-s", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["process_group"] == "nccl" and d["value"] > 0
    assert d["roofline"]["bound"] == "valu-latency" and d["cold"]["value"] > 0
