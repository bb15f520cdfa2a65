#!/usr/bin/env python3
"""tools/coresidency_probe.py: what the stand-alone generator kernel costs the rollout kernel it runs beside (BASELINE config
4: the next solve's noise_kernel starts with this solve's rollout, on a second stream, and fills the dependency bubbles of the
dynamics waves -- VERDICT round 4, weak 7).  The rollout kernel's own dispatch time (hipExtLaunchKernelGGL events) with the
generator beside it (generator mode, prefetching) and alone (explicit noise: no generator runs), same handle shape."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autorally_amd import capi, synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

K, T = 16384, 150
cfg = S.make_config(K, T, track="oval", layers=[6, 64, 64, 4])
st = cfg["start_state"]
out = {"workload": {"K": K, "T": T, "layers": cfg["layers"]}}
sol = capi.Solver(cfg)
out["variant"] = sol.rollout_variant()
for _ in range(100):
    sol.compute_control(st); sol.slide_control_seq(1)
sol.enable_stage_timing(1); sol.reset_stage_times()
for _ in range(100):
    sol.compute_control(st); sol.slide_control_seq(1)
t = sol.get_stage_times(); sol.enable_stage_timing(0)
out["generator_beside_the_rollout"] = {"rollout_kernel_us": 1e3 * t["rollout_ms"] / t["n_solves"], "generator_kernel_us": 1e3 * t["noise_ms"] / t["n_solves"], "solves": t["n_solves"]}
eps = O.generate_noise(1234, 0, K, T)[None]
sol2 = capi.Solver(cfg)
for _ in range(3):
    sol2.set_noise(eps); sol2.compute_control(st); sol2.slide_control_seq(1)
sol2.enable_stage_timing(1); sol2.reset_stage_times()
for _ in range(20):
    sol2.set_noise(eps); sol2.compute_control(st); sol2.slide_control_seq(1)
t2 = sol2.get_stage_times(); sol2.enable_stage_timing(0)
out["rollout_alone_explicit_noise"] = {"rollout_kernel_us": 1e3 * t2["rollout_ms"] / t2["n_solves"], "solves": t2["n_solves"], "variant": sol2.rollout_variant()}
print(json.dumps(out, indent=1))
sol.close(); sol2.close()
