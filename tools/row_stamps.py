#!/usr/bin/env python3
"""tools/row_stamps.py <lib built with -DMPPI_ROW_STAMPS> [T]: where one launch of the row rollout kernel spends its time
outside the T loop (s_memtime stamps of workgroup 0, cycles since the kernel's first instruction; diagnostic build)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
os.environ["MPPI_LIB_PATH"] = lib
from autorally_amd import capi, synthetic as S
K, T = 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = S.make_config(K, T, track="oval")
sol = capi.Solver(cfg)
for _ in range(50):
    sol.compute_control(cfg["start_state"])
    sol.slide_control_seq(1)
sol.enable_stage_timing(1); sol.reset_stage_times()
for _ in range(20):
    sol.compute_control(cfg["start_state"])
st = sol.get_stage_times()
L = C.CDLL(lib)
buf = (C.c_ulonglong * 16)()
assert L.mppi_debug_read_row_stamps(buf) == 0
t0 = buf[0]
names = ["first instruction", "behind the start barrier", "dynamics wave 0: weights in registers", "first controls published: T loop starts",
         "T loop done", "cost wave done (costs stored)", "control wave done", "pose wave done", "noise wave done"]
out = {"workload": {"K": K, "T": T, "variant": sol.rollout_variant()}, "rollout_kernel_us": 1e3 * st["rollout_ms"] / max(1, st["n_solves"]),
       "cycles_since_first_instruction": {names[i]: int(buf[i] - t0) for i in range(9)}}
c = out["cycles_since_first_instruction"]
out["cycles_per_step_in_loop"] = (c[names[4]] - c[names[3]]) / max(1, T - 1)
print(json.dumps(out, indent=1))
