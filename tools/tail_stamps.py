#!/usr/bin/env python3
"""tools/tail_stamps.py <lib built with -DMPPI_TAIL_STAMPS> [K]: where a row workgroup of solve_tail_kernel spends the time
between its first instruction and the publication of its row (s_memrealtime stamps of the workgroup of row T/2, 100 MHz;
diagnostic build)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
os.environ["MPPI_LIB_PATH"] = lib
from autorally_amd import capi, synthetic as S
K, T = (int(sys.argv[2]) if len(sys.argv) > 2 else 4096), 100
cfg = S.make_config(K, T, track="oval")
sol = capi.Solver(cfg)
L = C.CDLL(lib)
names = ["first instructions", "loads requested", "costs arrived + thread minimum", "beta (block minimum)", "exps", "eta (block sum)",
         "row staged in LDS (its loads arrived)", "weights normalised + barrier", "64-link chains", "partials added", "row published (store issued)"]
acc = [0.0] * 11
n = 0
for i in range(300):
    sol.compute_control(cfg["start_state"])
    sol.slide_control_seq(1)
    if i >= 100:
        buf = (C.c_ulonglong * 16)()
        assert L.mppi_debug_read_tail_stamps(buf) == 0
        for j in range(11):
            acc[j] += (buf[j] - buf[0]) * 10.0  # ns
        n += 1
out = {"workload": {"K": K, "T": T, "variant": sol.rollout_variant()},
       "ns_since_first_instruction": {names[j]: round(acc[j] / n, 1) for j in range(11)}}
print(json.dumps(out, indent=1))
