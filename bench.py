#!/usr/bin/env python3
"""bench.py -- MPPI solve throughput on MI355X (BASELINE.json metric: trajectory rollouts/s).

A "step" is one full MPPI solve through the C ABI (mppi_compute_control: device noise generation,
rollout, weighting, weighted reduction, Savitzky-Golay; blocking, incl. the small H2D/D2H of
U/state/results) followed by the control-loop's slideControlSeq(1) warm start.
N=1 workload = BASELINE.json configs[2]: K=4096, T=100, 6-32-32-4 shipped weights, CCRF-like oval
costmap written to / loaded from .npz.  N>1 = configs[4]: independent instances, one per GPU
(distinct costmaps + start states), no collective on the data path ("replicas only").
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

FLOP_PER_UPDATE = {(32, 2): 2756, (64, 2): 9604}   # MAC=2, bias add=1
BYTES_PER_UPDATE = 32                              # SURVEY 8(d) algorithmic HBM bytes
PEAK_F32_MFMA_TFLOPS = 157.3                       # MI355X_MICROARCH.md, dense f32 MFMA/vector
PEAK_HBM_GBPS = 8000.0


def flops_per_update(layers):
    f = 0
    for a, b in zip(layers[:-1], layers[1:]):
        f += 2 * a * b + b
    return f


def cpu_baseline(cfg, budget_s=12.0):
    """The CPU oracle (a C port of the reference kernels, OpenMP over rollouts) timed on this
    host on the SAME workload: a bounded sample of whole solves."""
    from oracle import oracle as O
    # the GPU box gives one GPU's share of host cores (16); never oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    orc = O.Oracle(cfg, fma_mode=1, nthreads=threads)
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
        if el >= budget_s or n >= 200:
            break
    return {"value": K * n / el, "unit": "rollouts/s", "cores": threads, "kind": "port",
            "sample": "%d full solves (rollout+weights+reduction+SG, explicit noise) of K=%d T=%d in %.1f s"
                      % (n, K, T, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--layers", type=str, default="")
    ap.add_argument("--variant", type=str, default="auto")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from autorally_amd import build as B
    if rank == 0:
        B.build()
    if dist is not None:
        dist.barrier()
    from autorally_amd import capi
    from autorally_amd import synthetic as S

    layers = [int(x) for x in args.layers.split("-")] if args.layers else None
    cfg = S.make_config(args.K, args.T, layers=layers, track="oval", instance=rank, seed=1234 + rank)
    sol = capi.Solver(cfg, device=local_rank)
    if args.variant != "auto":
        sol.set_rollout_variant(args.variant)
    if args.block:
        sol.set_rollout_variant("block%d" % args.block)
    state = cfg["start_state"].copy()

    def step():
        sol.compute_control(state)
        sol.slide_control_seq(1)

    for _ in range(args.warmup):
        step()
    sol.enable_stage_timing(True)
    sol.reset_stage_times()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    st = sol.get_stage_times()
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        K, T = cfg["K"], cfg["T"]
        iters = cfg.get("num_iters", 1)
        value = K * iters * args.steps * world / elapsed
        rollout_s = st["rollout_ms"] * 1e-3 / max(1, st["n_solves"]) / iters
        fl = flops_per_update(cfg["layers"])
        ach_tflops = fl * K * T / rollout_s / 1e12 if rollout_s > 0 else 0.0
        out = {
            "metric": "trajectory rollouts/s per MPPI solve",
            "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "K=%d T=%d %s NN dynamics, CCRF-like oval costmap via .npz, "
                                   "one independent MPPI instance per GPU" % (K, T, "-".join(map(str, cfg["layers"]))),
                       "K": K, "T": T, "layers": cfg["layers"], "num_iters": iters,
                       "rollout_variant": sol.rollout_variant(), "parallelism": "replicas x%d (no collective)" % world},
            "state_updates_per_s": value * T,
            "stage_ms": {k: (st[k] / max(1, st["n_solves"])) for k in
                         ("noise_ms", "rollout_ms", "weights_ms", "reduction_ms", "total_ms")},
            "roofline": {"bound": "mfma", "achieved": ach_tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach_tflops / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "kernel": "rollout", "kernel_ms": rollout_s * 1e3,
                         "flop_per_state_update": fl,
                         "hbm_algorithmic_GBps": BYTES_PER_UPDATE * K * T / rollout_s / 1e9 if rollout_s > 0 else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    sol.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
