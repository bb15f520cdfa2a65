#!/usr/bin/env python3
"""tools/stream_stamps.py <lib built with -DMPPI_TAIL_STAMPS> [K [T]]: where two workgroups of solve_tail_stream_kernel (K > 8192)
spend the time between their first instruction and the publication of the row -- the row-closing workgroup of row T/2 and
workgroup (T/2, chunk 0); s_memrealtime stamps, 100 MHz; diagnostic build (tools/build_variant.sh stamps solve_kernels.hip
-DMPPI_TAIL_STAMPS)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
os.environ["MPPI_LIB_PATH"] = lib
from autorally_amd import capi, synthetic as S
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cfg = S.make_config(K, T, track="oval")
sol = capi.Solver(cfg)
L = C.CDLL(lib)
names = ["first instructions", "loads requested", "before the poll of {beta, eta} (a leader: chunk minimum)", "beta known (a leader: after the exchange of minima)", "exps (a leader: + chunk sum)",
         "eta known (a leader: after the exchange of sums, replicas stored)", "weights + row staged in LDS", "barrier", "64-link chains (+ granules stored)",
         "other chunks' chain results collected", "row published (store issued)"]
acc = [[0.0] * 11 for _ in range(2)]
skew = 0.0
n = 0
for i in range(300):
    sol.compute_control(cfg["start_state"])
    sol.slide_control_seq(1)
    if i >= 100:
        buf = (C.c_ulonglong * 32)()
        assert L.mppi_debug_read_stream_stamps(buf) == 0
        for w in range(2):
            for j in range(11 if w == 0 else 9):
                acc[w][j] += (buf[16 * w + j] - buf[16 * w]) * 10.0  # ns
        skew += (buf[0] - buf[16]) * 10.0
        n += 1
out = {"workload": {"K": K, "T": T, "variant": sol.rollout_variant()},
       "closer_ns_since_first_instruction": {names[j]: round(acc[0][j] / n, 1) for j in range(11)},
       "chunk0_ns_since_first_instruction": {names[j]: round(acc[1][j] / n, 1) for j in range(9)},
       "closer_started_after_chunk0_ns": round(skew / n, 1)}
print(json.dumps(out, indent=1))
