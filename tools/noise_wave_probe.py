#!/usr/bin/env python3
"""tools/noise_wave_probe.py [K] [T]: what does the in-kernel noise wave cost the row form's dynamics waves?  The same solve with eps
drawn by the rollout kernel's own noise wave (MRG32k3a steps: quarter-rate 32 x 32 multiplies on the SIMD it shares with a dynamics
wave) and with eps uploaded beforehand (the noise wave idles, the control wave reads eps from memory): median host time of a blocking
compute_control, 400 solves each, alternating blocks."""
import sys
import time
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autorally_amd import capi, synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = S.make_config(K, T, track="oval")
st = np.asarray(cfg["start_state"], np.float32)
eps = O.generate_noise(7, 0, K, T)[None]
sol = capi.Solver(cfg)
res = {"generator": [], "uploaded": []}
for rep in range(4):
    for mode in ("generator", "uploaded"):
        sol.seed(11 + rep)
        for i in range(120):
            sol.reset_controls()
            if mode == "uploaded":
                sol.set_noise(eps)
            t0 = time.perf_counter()
            sol.compute_control(st)
            t1 = time.perf_counter()
            if i >= 20:
                res[mode].append(1e3 * (t1 - t0))
print("variant", sol.rollout_variant())
for mode, v in res.items():
    v = np.sort(v)
    print("%-10s median %.4f ms  p10 %.4f  p90 %.4f  (%d solves)" % (mode, np.median(v), v[len(v) // 10], v[9 * len(v) // 10], len(v)))
sol.close()
