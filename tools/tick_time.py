#!/usr/bin/env python3
"""tools/tick_time.py: what the two controllers of runControlLoop (run_control_loop.cuh:218-219; K = 1920 each by
default, the reference's rollout count) cost per tick on this GPU:
  one     -- one controller alone: solve + slide
  streams -- both controllers, each solve on its own handle's stream, enqueued before either is waited for (round 2)
  batch   -- both controllers in one launch (mppi_compute_control_batch)
Times are per tick (both solves + both slides), median of --repeats blocks of --ticks ticks, inside one library call
per block where the library has one (one, batch)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autorally_amd import capi, synthetic as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=1920)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=9)
    ap.add_argument("--modes", type=str, default="one,streams,batch")
    a = ap.parse_args()
    cfg = S.make_config(a.K, a.T, track="oval")
    st = cfg["start_state"]
    st2 = st.copy()
    st2[0] += 0.3
    out = {"K": a.K, "T": a.T, "ticks": a.ticks}
    for mode in a.modes.split(","):
        sols = [capi.Solver(cfg) for _ in range(1 if mode == "one" else 2)]

        def block():
            if mode == "one":
                sols[0].control_ticks(st, a.ticks, 1)
            elif mode == "batch":
                capi.control_ticks_batch(sols, [st, st2], a.ticks, 1)
            else:
                for _ in range(a.ticks):
                    sols[0].compute_control_async(st)
                    sols[1].compute_control_async(st2)
                    sols[0].synchronize()
                    sols[1].synchronize()
                    sols[0].slide_control_seq(1)
                    sols[1].slide_control_seq(1)
        for _ in range(3):
            block()
        ts = []
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            block()
            ts.append(1e3 * (time.perf_counter() - t0) / a.ticks)
        out[mode] = {"ms_per_tick_median": float(np.median(ts)), "min": float(min(ts)), "max": float(max(ts)),
                     "variant": sols[0].rollout_variant()}
        for s in sols:
            s.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
