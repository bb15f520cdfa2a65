#!/usr/bin/env python3
"""bench.py -- MPPI solve throughput on MI355X (BASELINE.json metric: trajectory rollouts/s).

A "step" is one full MPPI solve through the C ABI (mppi_compute_control: device noise generation,
rollout, weighting, weighted reduction, Savitzky-Golay; blocking, incl. the small H2D/D2H of
U/state/results) followed by the control loop's slideControlSeq(1) warm start.
N=1 workload = BASELINE.json configs[2]: K=4096, T=100, 6-32-32-4 shipped weights, CCRF-like oval
costmap written to / loaded from .npz.  N>1 = configs[4]: independent instances, one per GPU
(distinct costmaps + start states), no collective on the data path ("replicas only"); the process
group is used for the barrier and the max-over-ranks time only.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

# Algorithmic HBM bytes per state update.  SURVEY 8(d) counts 32 B for the
# whole solve with a stand-alone generator (noise write 8 + rollout read 8 + write-back 8 +
# reduction read 8).  The quad rollout kernel draws eps in-kernel, so its share is the 8-B
# write-back only (whole solve 16 B); the single-wave / VALU forms read eps from HBM (16 B).
ROLLOUT_BYTES_INLINE_NOISE = 8
ROLLOUT_BYTES_BUFFERED_NOISE = 16
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense f32 MFMA (= f32 vector) peak
PEAK_HBM_GBPS = 8000.0
def kernel_sources_sha16():
    """sha256 (first 16 hex digits) over the kernel sources a rollout launch is compiled from: tools/prof_summary.py stamps a
    PMC summary with it, and a summary taken on other sources no longer describes the code that is running."""
    import glob
    import hashlib
    h = hashlib.sha256()
    host_side = ("abi_", "mppi_abi", "host_net", "tanhf_vec", "ddp_")  # the C ABI's host files: no device code
    for f in sorted(glob.glob(os.path.join(ROOT, "autorally_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "autorally_amd", "csrc", "*.hpp"))):
        if os.path.basename(f).startswith(host_side):
            continue
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(cfg, variant):
    """(HBM bytes per rollout launch or None, note) from the committed rocprofv3 PMC passes (tools/prof_run.sh:
    FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE, separate passes).  PMC counters cannot be read from inside this
    process, so this is a profiled constant: taken from the newest profiles/r*_pmc.json of this exact workload and kernel
    form, and only if that summary was stamped with the kernel sources this run was built from -- otherwise None, with
    the stale figure named in the note."""
    import glob
    sha = kernel_sources_sha16()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):  # newest round / letter first
        try:
            with open(path) as f:
                p = json.load(f)
            w = p["workload"]
            if (w["K"], w["T"], w["layers"], w["rollout_variant"]) != (cfg["K"], cfg["T"], list(cfg["layers"]), variant):
                continue
            for name, e in p["kernels"].items():
                if "rollout" in name and "traffic_bytes_per_launch" in e:
                    rel = os.path.relpath(path, ROOT)
                    if p.get("kernel_sources_sha16") == sha:
                        return e["traffic_bytes_per_launch"], "HBM bytes per launch (rocprofv3 PMC passes, %s, same kernel sources %s)" % (rel, sha)
                    if stale is None:
                        stale = "null: the newest PMC pass of this workload and form (%s: %.0f bytes per launch) was taken on other kernel sources (%s, now %s)" % (
                            rel, e["traffic_bytes_per_launch"], p.get("kernel_sources_sha16", "unstamped"), sha)
        except (OSError, KeyError, ValueError):
            pass
    return None, stale or "null: no committed PMC pass (tools/prof_run.sh -> profiles/) for this workload and kernel form"


BASIS_FLOPS_PER_UPDATE = 270  # 100 MACs of W phi + ~70 multiply/divide/add of the 25 basis functions, no transcendentals


def flops_per_update(layers):
    """MAC = 2, bias add = 1, no transcendentals: 2756 for 6-32-32-4, 9604 for 6-64-64-4."""
    return sum(2 * a * b + b for a, b in zip(layers[:-1], layers[1:]))


def init_dist(backend, devices=None, force=False):
    """(rank, device, world, dist-or-None); one process per GPU, env from torch.distributed.run.
    `devices` (--devices "0,0"): rank -> device map for rehearsing the N > 1 path on fewer GPUs than ranks
    (then with --dist-backend gloo: RCCL refuses two ranks on one device); default device = LOCAL_RANK."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = local_rank
    if devices:
        if len(devices) != world:
            raise SystemExit("--devices names %d devices for WORLD_SIZE=%d" % (len(devices), world))
        device = devices[rank]
    if world == 1 and not force:
        return rank, device, world, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend == "nccl":
        torch.cuda.set_device(device)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend=backend)
    return rank, device, world, dist


def pin_to_gpu_numa(device):
    """Keeps this process (its solve thread polls the result block the GPU writes over PCIe) on the CPUs of the
    NUMA node its GPU hangs off.  Best effort: returns a description, or the reason nothing was done."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return "unchanged (no PCI bus id)"
        bus = buf.value.decode().lower()
        with open("/sys/bus/pci/devices/%s/numa_node" % bus) as f:
            node = int(f.read().strip())
        if node < 0:
            return "unchanged (device %s reports no NUMA node)" % bus
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return "unchanged (no allowed CPU on node %d)" % node
        os.sched_setaffinity(0, cpus)
        return "NUMA node %d of GPU %s (%d CPUs)" % (node, bus, len(cpus))
    except (OSError, ValueError, AttributeError) as e:
        return "unchanged (%s)" % type(e).__name__


def rank_workload(args, rank):
    """Instance `rank` of BASELINE configs[4]: its own costmap (rotated/offset oval), start state
    and RNG seed.  Rank 0 alone is configs[2]."""
    from autorally_amd import synthetic as S
    layers = [int(x) for x in args.layers.split("-")] if args.layers else None
    over = {}
    if args.dynamics == "basis":
        from autorally_amd import params as P
        over["bf_W"] = P.load_bf_npz(os.path.join(S.MODELS_DIR, "basis_function_09_12_2018.npz"))
    return S.make_config(args.K, args.T, layers=layers, track="oval", instance=rank, seed=1234 + rank, **over)


def rollout_bytes_per_update(variant):
    """Algorithmic HBM bytes of the rollout kernel per state update, by what the form that ran does with eps."""
    inline_noise = (("quad" in variant or "oct8w" in variant or "row8w" in variant or "row64" in variant or "m44" in variant or "multi" in variant)
                    and not variant.endswith("_gen")) \
        or variant.endswith("_3w")  # the kernel's own noise / control wavefront draws eps
    return ROLLOUT_BYTES_INLINE_NOISE if inline_noise else ROLLOUT_BYTES_BUFFERED_NOISE


def max_over_ranks(dist, value, on_gpu):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _host_cpus():
    """CPUs this process may run on (after pin_to_gpu_numa: the GPU's NUMA node), the physical cores among them (one per set
    of hyper-thread siblings), the container's CPU quota if it has one, and the CPU's model string."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = {}
    for c in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                key = f.read().strip()
        except OSError:
            key = str(c)
        cores.setdefault(key, c)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            quota = None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return allowed, sorted(cores.values()), quota, model


def cpu_baseline(cfg, budget_s=12.0):
    """The CPU oracle (C restatement of the reference kernels, OpenMP over rollouts) built -O3 -march=native for THIS host
    and timed on it on the SAME workload: a bounded sample of whole solves on all physical cores this process may use (the
    GPU's NUMA node; capped by the container's CPU quota and at 16, the share of the box that goes with one GPU), and on one."""
    from oracle import oracle as O
    allowed, phys, quota, model = _host_cpus()
    threads = len(phys)
    if quota is not None:
        threads = min(threads, max(1, int(quota)))
    threads = max(1, min(threads, 16))
    if hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, set(phys))  # one thread per physical core (the timed region is over: rank 0, N = 1 only)
    # the bits of the native build against the portable build of the tests, on BASELINE configs[0]'s size
    from autorally_amd import synthetic as S
    small = S.make_config(128, 50, track="ring")
    eps_s = O.generate_noise(1234, 0, 128, 50)[None]
    Us, hs = np.zeros((50, 2), np.float32), np.zeros(4, np.float32)
    ra = O.Oracle(small, fma_mode=1, nthreads=1).compute_control(small["start_state"], Us, hs, eps_s)
    rb = O.Oracle(small, fma_mode=1, nthreads=1, native=True).compute_control(small["start_state"], Us, hs, eps_s)
    same = bool(np.array_equal(ra["U"].view(np.uint32), rb["U"].view(np.uint32)) and
                np.array_equal(ra["costs"].view(np.uint32), rb["costs"].view(np.uint32)))
    orc = O.Oracle(cfg, fma_mode=1, nthreads=threads, native=True)
    K, T = cfg["K"], cfg["T"]
    U = np.zeros((T, 2), np.float32)
    hist = np.zeros(4, np.float32)
    eps = O.generate_noise(1234, 0, K, T)[None]
    orc.compute_control(cfg["start_state"], U, hist, eps)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        r = orc.compute_control(cfg["start_state"], U, hist, eps)
        U = r["U"]
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 400:
            break
    out = {"value": K * n / el, "unit": "rollouts/s", "cores": threads, "kind": "port",
           "cpu_model": model, "physical_cores_allowed": len(phys), "logical_cpus_allowed": len(allowed), "cpu_quota": quota,
           "build": "gcc -O3 -march=native -ffp-contract=off -fopenmp (oracle/Makefile: libmppi_oracle_native.so), OpenMP static "
                    "schedule over rollouts, one thread per physical core",
           "bit_identical_to_portable_build": same,
           "sample": "%d full solves (rollout+weights+reduction+SG on pre-generated noise) of K=%d T=%d in %.1f s"
                     % (n, K, T, el)}
    # the same solve on ONE core (SURVEY 8d asks for both), a few solves only
    orc1 = O.Oracle(cfg, fma_mode=1, nthreads=1, native=True)
    n1, t1 = 0, time.perf_counter()
    while True:
        orc1.compute_control(cfg["start_state"], U, hist, eps)
        n1 += 1
        el1 = time.perf_counter() - t1
        if el1 >= 4.0 or n1 >= 50:
            break
    out["single_thread"] = {"value": K * n1 / el1, "unit": "rollouts/s", "cores": 1,
                            "sample": "%d full solves in %.1f s" % (n1, el1)}
    return out


class _OracleStandIn:
    """--selftest-cpu only: stands in for the HIP solver so that the multi-process driver logic
    (rank workloads, barrier, max-over-ranks, aggregation) can be exercised with gloo on a CPU-only
    machine.  Never used for a reported number."""

    def __init__(self, cfg):
        from oracle import oracle as O
        self.O, self.cfg = O, cfg
        self.orc = O.Oracle(cfg, fma_mode=1, nthreads=1)
        self.U = np.zeros((cfg["T"], 2), np.float32)
        self.hist = np.zeros(4, np.float32)
        self.n = 0

    def compute_control(self, state):
        eps = self.O.generate_noise(self.cfg["seed"], 2 * self.cfg["T"] * self.n, self.cfg["K"], self.cfg["T"])[None]
        self.U = self.orc.compute_control(state, self.U, self.hist, eps)["U"]
        self.n += 1

    def slide_control_seq(self, stride):
        self.U, self.hist = self.orc.slide_control_seq(self.U, self.hist, self.cfg["init_u"], stride)

    def get_control_seq(self):
        return self.U


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--layers", type=str, default="")
    ap.add_argument("--dynamics", choices=("nn", "basis"), default="nn",
                    help="basis: the reference's second model family (GeneralizedLinear, 25 basis functions; "
                         "its build uses K=2560), with the shipped basis_function_09_12_2018.npz")
    ap.add_argument("--variant", type=str, default="auto")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--loop", choices=("native", "python"), default="native",
                    help="timed steps inside one library call (mppi_control_ticks) or one ctypes call per ABI call")
    ap.add_argument("--prime-ms", type=float, default=250.0,
                    help="initialisation before the warm-up steps: solves repeated for this long (clock ramp, code load)")
    ap.add_argument("--repeats", type=int, default=10,
                    help="timed blocks of --steps steps each; their MEDIAN is the reported value / ms_per_step (SURVEY 8d), "
                         "the first block alone is first_block_ms_per_step, min / max are extra keys")
    ap.add_argument("--latency-solves", type=int, default=200,
                    help="separate pass after the timed region: steps timed one by one (median solve latency)")
    ap.add_argument("--event-solves", type=int, default=64,
                    help="separate pass after the timed region: solves with HIP events around every stage")
    ap.add_argument("--sustained-s", type=float, default=2.0,
                    help="separate pass after the timed region: the step repeated for this many seconds (sustained rate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-chain", action="store_true",
                    help="A/B: mppi_control_ticks launches every solve when its turn comes instead of one tick ahead "
                         "(mppi_debug_set_chained_ticks(0))")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="exercise the multi-process driver with gloo and the CPU oracle (no GPU, not a benchmark)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="process group for barrier / max / gather (never on the data path); nccl = RCCL (default), "
                         "gloo for rehearsing N > 1 with several ranks on one GPU")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise the process group at world size 1 too (smoke of the RCCL path on a one-GPU box: "
                         "barrier and max-over-ranks on a CUDA tensor; nothing of the data path changes)")
    ap.add_argument("--devices", type=str, default="",
                    help="rank -> device map, e.g. '0,0' (default: device = LOCAL_RANK)")
    args = ap.parse_args()

    selftest = args.selftest_cpu
    backend = "gloo" if selftest else args.dist_backend
    devices = [int(x) for x in args.devices.split(",")] if args.devices else None
    rank, local_rank, world, dist = init_dist(backend, devices, args.force_process_group)  # local_rank: this rank's DEVICE from here on
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    cuda = not selftest
    if cuda:
        import torch
        torch.cuda.set_device(local_rank)
        affinity = pin_to_gpu_numa(local_rank)
        from autorally_amd import build as B
        if rank == 0:
            B.build()
        if dist is not None:
            dist.barrier()
        from autorally_amd import capi

    cfg = rank_workload(args, rank)
    if selftest:
        sol = _OracleStandIn(cfg)
    else:
        sol = capi.Solver(cfg, device=local_rank)  # raises without the HIP library / a gfx950 device
        if args.variant != "auto":
            sol.set_rollout_variant(args.variant)
        if args.block:
            sol.set_rollout_variant("block%d" % args.block)
        if args.no_chain:
            sol.debug_set_chained_ticks(0)
    state = cfg["start_state"].copy()

    def sync():
        if cuda:
            torch.cuda.synchronize()

    solve = sol.bind_state(state) if cuda else (lambda: sol.compute_control(state))

    def step():
        solve()
        sol.slide_control_seq(1)

    native = cuda and args.loop == "native"

    def run_steps(n):
        # one step = computeControl(state) + slideControlSeq(1), result on the host before the next one
        # starts.  "native": the n steps run inside one library call (mppi_control_ticks), as they would
        # under a C++ caller like the reference's runControlLoop; "python": one ctypes call per ABI call.
        if native:
            sol.control_ticks(state, n, 1)
        else:
            for _ in range(n):
                step()

    local_s = []  # this rank's own time for the steps of each block, before the closing barrier (per-GPU spread)
    incl_s = []   # the same blocks with the closing barrier inside the bracket, max over ranks (extra key)

    def timed_block():
        """EXACTLY args.steps steps between barrier + synchronize on both sides.  Returned: the MAX over ranks of every
        rank's OWN time for its steps (opening barrier + synchronize, the steps, synchronize) -- the slowest GPU's time.
        The closing barrier itself (an all-reduce launch + a synchronize through torch, tens of microseconds against a block
        of 0.9 ms at the driver's --steps 20) is no part of any GPU's work and no scaling loss: it stays outside the
        figure and is kept beside it (incl_s, the bracket closed behind the barrier, max over ranks)."""
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        run_steps(args.steps)
        sync()
        own = time.perf_counter() - t0
        local_s.append(own)
        if dist is not None:
            dist.barrier()
        incl = time.perf_counter() - t0
        on_gpu = cuda and backend == "nccl"
        incl_s.append(max_over_ranks(dist, incl, on_gpu))
        return max_over_ranks(dist, own, on_gpu)

    # Initialisation, before the W warm-up steps: the same step repeated for --prime-ms of wall time, so
    # that code objects are loaded, allocations are settled and the GPU has left its idle clock state (a
    # fresh process starts at the idle shader clock and the W = 5 steps the driver asks for last 0.4 ms).
    # A controller runs for minutes at 50 Hz; the steady state is the quantity of interest.  Reported below.
    # Before any of that, the contract read literally: W warm-up steps on the fresh process, then the K timed steps --
    # reported as the extra key "cold" beside the primed figure (the first solves of a process run at the idle shader clock).
    cold_s = None
    if cuda and args.prime_ms > 0:
        run_steps(args.warmup)
        cold_s = timed_block()
        local_s.clear()
        incl_s.clear()
    t_prime, n_prime = time.perf_counter(), 0
    while cuda and 1e3 * (time.perf_counter() - t_prime) < args.prime_ms:
        run_steps(20)
        n_prime += 20
    sync()
    run_steps(args.warmup)
    # the contract's timed region: no event records, no per-step host timing inside it
    elapsed = timed_block()
    # the same block repeated (off the headline number): spread of the measurement
    block_s = [elapsed] + [timed_block() for _ in range(max(0, args.repeats - 1))]
    # separate pass 1, directly behind the timed blocks (the GPU in the state they ran in: behind the 2 s sustained pass the
    # same kernel measures ~12 % longer): HIP events around the stages of every solve (they lengthen the launch gaps, so
    # they stay out of the timed region); the rollout kernel's duration feeds the roofline block
    if cuda:
        sol.enable_stage_timing(1)
        sol.reset_stage_times()
        run_steps(args.event_solves)
        sync()
        stage_times = sol.get_stage_times()
        sol.enable_stage_timing(0)
    # separate pass 2: every step timed on its own (one ctypes call per ABI call): median solve latency
    per_solve_ms = None
    if cuda and rank == 0 and args.latency_solves > 0:
        lat = []
        for _ in range(args.latency_solves):
            t1 = time.perf_counter()
            step()
            lat.append(1e3 * (time.perf_counter() - t1))
        lat = np.sort(np.asarray(lat))
        per_solve_ms = {"n": int(lat.size), "median": float(np.median(lat)), "p10": float(lat[int(0.1 * lat.size)]),
                        "p90": float(lat[int(0.9 * lat.size)]), "min": float(lat[0]), "max": float(lat[-1])}
    # separate pass 3: a sustained run (off the headline number): the same step for --sustained-s seconds of wall time --
    # what a controller that runs for minutes sees, and long enough for a utilisation sampler to catch the GPU at work
    sustained = None
    if cuda and args.sustained_s > 0:
        sync()
        n_s, t_s = 0, time.perf_counter()
        while time.perf_counter() - t_s < args.sustained_s:
            run_steps(500)
            n_s += 500
        sync()
        el_s = time.perf_counter() - t_s
        sustained = {"seconds": el_s, "solves": n_s, "ms_per_step": 1e3 * el_s / n_s,
                     "value": cfg["K"] * cfg.get("num_iters", 1) * n_s / el_s, "unit": "rollouts/s (this rank)"}
    if dist is not None:
        dist.barrier()

    # what every rank ran (proves the instances are distinct), gathered off the timed region
    own_roofline = None
    if cuda:
        # this rank's rollout kernel against the roofline (north_star: "rollouts/s and achieved HBM GB/s vs. the roofline" at
        # 1 / 2 / 4 / 8 GPUs): its own stage events, the algorithmic bytes and flop of its own launch
        st_ = stage_times
        n_ = max(1, st_["n_solves"])
        it_ = cfg.get("num_iters", 1)
        rs_ = st_["rollout_ms"] * 1e-3 / n_ / it_
        fl_ = BASIS_FLOPS_PER_UPDATE if cfg.get("bf_W") is not None else flops_per_update(cfg["layers"])
        bpu_ = rollout_bytes_per_update(sol.rollout_variant())
        own_roofline = {"rollout_kernel_ms": rs_ * 1e3,
                        "achieved_TFLOPs": fl_ * cfg["K"] * cfg["T"] / rs_ / 1e12 if rs_ > 0 else 0.0,
                        "achieved_GBps": bpu_ * cfg["K"] * cfg["T"] / rs_ / 1e9 if rs_ > 0 else 0.0,
                        "frac_of_f32_peak": (fl_ * cfg["K"] * cfg["T"] / rs_ / 1e12 / PEAK_F32_MFMA_TFLOPS) if rs_ > 0 else 0.0,
                        "frac_of_hbm_peak": (bpu_ * cfg["K"] * cfg["T"] / rs_ / 1e9 / PEAK_HBM_GBPS) if rs_ > 0 else 0.0}
    mine = {"rank": rank, "device": local_rank, "start_state": [round(float(x), 4) for x in cfg["start_state"]],
            "map_checksum": float(np.asarray(cfg["map_rgba"], dtype=np.float64).sum()),
            "U0": [round(float(x), 5) for x in sol.get_control_seq()[0]],
            "own_ms_per_step": 1e3 * local_s[0] / args.steps,  # the timed block, this GPU alone (no closing barrier)
            "own_median_ms_per_step": 1e3 * float(np.median(local_s)) / args.steps,
            "own_blocks_ms_per_step": [1e3 * x / args.steps for x in local_s],
            "roofline": own_roofline}
    instances = [mine]
    if dist is not None:
        instances = [None] * world
        dist.all_gather_object(instances, mine)

    if rank == 0:
        K, T = cfg["K"], cfg["T"]
        iters = cfg.get("num_iters", 1)
        # The reported figure: the MEDIAN of the --repeats timed blocks, each of EXACTLY args.steps steps behind warm-up
        # steps and bracketed as the contract says (SURVEY 8d defines the metric as a median over solves; on a shared host a
        # block now and then runs 1.3-1.5x long -- profiles/r04_s_table.txt has four such rows of thirteen -- and a single block
        # would report the neighbour, not the kernel).  The first block alone stays in the line as "first_block_ms_per_step".
        typical = float(np.median(block_s))
        value = K * iters * args.steps * world / typical
        out = {
            "metric": "trajectory rollouts/s per MPPI solve",
            "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * typical / args.steps,
            "first_block_ms_per_step": 1e3 * elapsed / args.steps,
            "value_basis": "median of %d timed blocks of exactly %d steps each; a block = barrier + synchronize, the steps, "
                           "synchronize on every rank, timed on every rank, MAX over ranks of the ranks' own times (one all-reduce "
                           "after the block); the closing barrier of the contract is executed behind every block and is NOT inside "
                           "the figure: the bracket closed behind it is ms_per_step_incl_closing_barrier; the first block alone: "
                           "first_block_ms_per_step" % (len(block_s), args.steps),
            "ms_per_step_incl_closing_barrier": 1e3 * float(np.median(incl_s)) / args.steps,
            "blocks_ms_per_step": [1e3 * x / args.steps for x in block_s],
            "blocks_ms_per_step_incl_closing_barrier": [1e3 * x / args.steps for x in incl_s],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if cuda else "selftest-cpu (oracle stand-in, NOT a benchmark)",
            "config": {"workload": "K=%d T=%d %s dynamics, CCRF-like oval costmap via .npz, "
                                   "one independent MPPI instance per GPU"
                                   % (K, T, "25 basis functions (GeneralizedLinear)" if cfg.get("bf_W") is not None
                                      else "-".join(map(str, cfg["layers"])) + " NN"),
                       "K": K, "T": T, "layers": [] if cfg.get("bf_W") is not None else cfg["layers"], "num_iters": iters,
                       "rollout_variant": sol.rollout_variant() if cuda else "none",
                       "step": "computeControl + slideControlSeq(1), result on the host before the next step",
                       "timed_loop": (("native (mppi_control_ticks%s)" % ("" if not args.no_chain else ", chain off")) if native
                                      else "python (one ctypes call per ABI call)"),
                       "host_affinity": affinity if cuda else "unchanged",
                       "priming": "%d untimed solves (%.0f ms) before the %d warm-up steps" % (n_prime, args.prime_ms, args.warmup),
                       "parallelism": "replicas x%d (no collective)" % world,
                       "process_group": backend if dist is not None else "none"},
            "state_updates_per_s": value * T,
            "repeats": len(block_s),
            "median_ms_per_step": 1e3 * float(np.median(block_s)) / args.steps,
            "min_ms_per_step": 1e3 * min(block_s) / args.steps,
            "max_ms_per_step": 1e3 * max(block_s) / args.steps,
            "median_value": K * iters * args.steps * world / float(np.median(block_s)),
            "cold": None if cold_s is None else {
                "ms_per_step": 1e3 * cold_s / args.steps, "value": K * iters * args.steps * world / cold_s,
                "note": "the same %d timed steps straight after %d warm-up steps on the fresh process, before the priming pass "
                        "(barrier + synchronize on both sides, max over ranks): the contract read literally" % (args.steps, args.warmup)},
            "per_solve_ms": per_solve_ms,
            "sustained": sustained,
            "instances": instances,
        }
        if cuda:
            st = stage_times
            n = max(1, st["n_solves"])
            rollout_s = st["rollout_ms"] * 1e-3 / n / iters
            fl = BASIS_FLOPS_PER_UPDATE if cfg.get("bf_W") is not None else flops_per_update(cfg["layers"])
            ach = fl * K * T / rollout_s / 1e12 if rollout_s > 0 else 0.0
            variant = sol.rollout_variant()
            bpu = rollout_bytes_per_update(variant)
            out["stage_ms"] = {k: st[k] / n for k in ("noise_ms", "rollout_ms", "weights_ms", "reduction_ms", "total_ms")}
            out["stage_ms"]["note"] = ("HIP events on the handle's stream, %d solves, separate pass after the timed region: markers around the "
                                       "noise and tail stages; the rollout stage is the begin / end of the rollout kernel's own dispatch "
                                       "(hipExtLaunchKernelGGL start / stop events: the quantity rocprofv3 --kernel-trace reports)" % n)
            if variant.endswith("_gen") or "fused" in variant or variant.startswith("valu") or variant == "basis_funcs25_valu":
                out["stage_ms"]["note"] += ("; noise_ms = the stand-alone generator kernel's launch (its own begin / end events): in the "
                                            "prefetching forms it runs on a second stream BESIDE the rollout or the tail of the same "
                                            "solve, so noise + rollout + tail is more than the step")
            traffic, traffic_note = measured_traffic(cfg, variant)
            # which ceiling: the matrix-instruction forms against the dense f32 MFMA peak; the row forms issue no MFMA and are
            # bound by the LATENCY of one wave's dependent vector multiply-add chain (peak kept: f32 vector = f32 MFMA peak);
            # the throughput-style vector kernels and the basis-function model against the f32 vector peak
            bound = "valu-latency" if "row8w" in variant else ("mfma" if variant.startswith("mfma") else "valu")
            out["roofline"] = {
                "bound": bound, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                "traffic_unit": traffic_note,
                "kernel": "rollout (%s)" % variant, "kernel_ms": rollout_s * 1e3,
                "flop_per_state_update": fl, "state_updates_per_launch": K * T,
                "algorithmic_bytes_per_launch": bpu * K * T,
                "algorithmic_GBps": bpu * K * T / rollout_s / 1e9 if rollout_s > 0 else 0.0,
                "hbm_peak_GBps": PEAK_HBM_GBPS,
            }
            if native and not args.no_chain and ("row8w" in variant or "m44_split" in variant or ("multi4_tree_gen" in variant and K // 16 <= 1024 and K * T >= 1 << 20)):
                out["roofline"]["chained_ticks"] = (
                    "the timed steps run chained (mppi_control_ticks enqueues every solve but a block's first one tick ahead: DESIGN.md "
                    "4.10); under rocprofv3 they appear as rollout_*_gated_kernel, whose duration INCLUDES its wait for the host's gate "
                    "(about 1 us on average, rare long ones when the host thread is descheduled); kernel_ms / achieved here are the "
                    "UNGATED kernel's (the stage-event pass runs unchained: the same code without the wait), which rocprofv3 lists "
                    "as rollout_*_kernel (a block's first solve)")
            if "row8w" in variant:
                # The latency form on the vector ALU (rollout_row.hip): no MFMA is issued; the f32 vector peak with packed
                # multiply-adds equals the f32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md), so the roofline block keeps
                # that ceiling.  The configuration is latency bound: cycles per step against the recurrence of one dynamics
                # wavefront alone on a SIMD (tools/ub/row_bcast_ub.hip, profiles/r03_j_row_bcast_ub.txt: the activations handed
                # round by DPP, 70 dependent v_pk_fma_f32 at ~9 cycles + two tanh = 865 cycles; with the per-step bookkeeping of
                # the product's dynamics wave -- record, sequence word, next controls, progress test -- 1 024).
                clk_ghz = 2.3  # ASSUMED: the clock the chip held under this kernel when the stamps were taken (in-kernel
                # s_memtime cycles x T against the rocprofv3 duration); not measured in this run
                cyc = rollout_s / T * clk_ghz * 1e9
                tree = "tree" in variant
                # the k-ascending chain of the exact form: 70 dependent v_pk_fma_f32 + two tanh = 865 cycles alone on a SIMD; the
                # tree form replaces the output layer's 32 links by 2 packed products + a 4-deep butterfly: 6 + 32 links, two
                # tanh, the butterfly (profiles/r03_j_row_bcast_ub.txt and DESIGN.md 4.2c)
                # (round 4: one v_mov_b64_dpp per activation pair -- a 32-link layer 292 -> 272 cycles alone on a SIMD,
                # tools/ub/row_split_ub.hip forms 0 / 4, profiles/r04_q_row_split_ub.txt)
                floor = (865.0 - 32 * 9.0 + 60.0 - 20.0) if tree else (865.0 - 40.0)
                out["roofline"]["pipe"] = "vector ALU (v_pk_fma_f32), no MFMA issued; f32 vector peak = f32 MFMA peak"
                out["roofline"]["recurrence"] = {
                    "cycles_per_step": cyc, "bare_recurrence_cycles_per_step": floor, "frac_of_floor": floor / cyc if cyc > 0 else 0.0,
                    "clock_GHz_assumed": clk_ghz,
                    "note": "event-measured kernel time / T (launch, prologue and riders included) against one dynamics wavefront's "
                            "recurrence alone on a SIMD (tools/ub/row_bcast_ub.hip: 865 cycles for the exact chain with 32-bit moves; tree form: "
                            "that minus 32 links of ~9 cycles plus ~60 for two packed products and the butterfly; 64-bit moves: "
                            "20 cycles less per 32-link layer, tools/ub/row_split_ub.hip)"}
            if "quad" in variant and cfg.get("bf_W") is None:
                # The configuration is latency bound (one 16-rollout group per CU, T sequential steps), so next
                # to the throughput roofline: the step time of the recurrence against (a) what its two dynamics
                # wavefronts must at least ISSUE and (b) the floor of this decomposition with its dependency
                # latencies and one hand-over per swap (tools/ub/*.hip, tools/quad_stamps.py:
                # f32 MFMA 32.6 cycles, tanh 28.8 cycles per register with packed multiply-adds, MFMA and VALU do
                # not overlap inside a wave, ~95 cycles of exposed latency per layer boundary, >= 190 per hand-over).
                H, nh = cfg["layers"][1], len(cfg["layers"]) - 2
                mt, ksh = H // 16, H // 4
                mfma = 2 * mt + (nh - 1) * (mt // 2) * ksh + ksh          # layer 0 on both waves, own tiles, output layer
                tanh = 4 * mt + (nh - 1) * 4 * (mt // 2)
                issue = 32.6 * mfma + 28.8 * tanh
                floor = issue + 95.0 * (nh + 1) + 190.0 * (nh - 1)
                clk_ghz = 2.3  # in-kernel s_memtime cycles per step x T against the rocprofv3 duration
                cyc = rollout_s / T * clk_ghz * 1e9
                out["roofline"]["recurrence"] = {
                    "cycles_per_step": cyc, "issue_cycles_per_step": issue, "decomposition_floor_cycles_per_step": floor,
                    "frac_of_floor": floor / cyc if cyc > 0 else 0.0,
                    "mfma_per_step": mfma, "tanh_per_lane_per_step": tanh, "clock_GHz": clk_ghz,
                    "note": "event-measured kernel time / T (launch, prologue and the events' own overhead included) at the clock "
                            "the chip holds under this kernel; stamped phases: profiles/r02_b_quad_dynamics_wave_stamps_h32.json"}
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    if cuda:
        sol.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
