#!/usr/bin/env python3
"""tools/quad_stamps.py <lib built with -DMPPI_STAMPS> [--layers ..] : phases of one step of the two dynamics waves
of the quad rollout kernel (s_memtime, workgroup 0, averaged over steps 16..T-1 of the last solve).
Diagnostic build only: the stamps perturb the kernel (each is an s_memtime + s_waitcnt lgkmcnt(0))."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
os.environ["MPPI_LIB_PATH"] = lib
from autorally_amd import capi, synthetic as S, params as P
layers = None
if "--layers" in sys.argv:
    layers = [int(x) for x in sys.argv[sys.argv.index("--layers") + 1].split("-")]
K, T = 4096, 100
cfg = S.make_config(K, T, layers=layers, track="oval")
sol = capi.Solver(cfg)
sol.set_rollout_variant("quad")
for _ in range(20):
    sol.compute_control(cfg["start_state"])
    sol.slide_control_seq(1)
sol.enable_stage_timing(1); sol.reset_stage_times()
for _ in range(20):
    sol.compute_control(cfg["start_state"])
st = sol.get_stage_times()
L = C.CDLL(lib)
buf = (C.c_ulonglong * 16)()
assert L.mppi_debug_read_quad_stamps(buf) == 0
out = {"workload": {"K": K, "T": T, "layers": cfg["layers"], "variant": sol.rollout_variant()},
       "rollout_kernel_ms_with_stamps": st["rollout_ms"] / max(1, st["n_solves"]),
       "note": "cycles per step (s_memtime), averaged over steps 16..T-1 of one solve, workgroup 0; diagnostic build, "
               "every stamp is s_memtime + s_waitcnt lgkmcnt(0)"}
names = ["layer0_mfma_tanh", "own_tile_mfma_tanh_store_publish", "early_output_mfma_and_wait_for_partner", "rest_of_output_layer_and_euler"]
for w in range(2):
    n = buf[w * 8 + 4]
    out["dynamics_wave_%d" % w] = {names[i]: buf[w * 8 + i] / n for i in range(4)}
    out["dynamics_wave_%d" % w]["loop_back_edge_and_record"] = buf[w * 8 + 5] / n
    out["dynamics_wave_%d" % w]["sum"] = (sum(buf[w * 8 + i] for i in range(4)) + buf[w * 8 + 5]) / n
print(json.dumps(out, indent=1))
