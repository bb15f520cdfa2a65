#!/usr/bin/env python3
"""tools/multi_stamps.py <lib built with -DMPPI_STAMPS> [--layers ..] [--variant multi2] [--K 4096]: phases of one
step of dynamics wave 0 of the multi rollout kernel (s_memtime, workgroup 0, averaged over steps 16..T-2 of the
last solve).  Diagnostic build only: every stamp is an s_memtime + s_waitcnt lgkmcnt(0)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
os.environ["MPPI_LIB_PATH"] = lib
from autorally_amd import capi, synthetic as S, params as P


def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


layers = [int(x) for x in opt("--layers", "6-32-32-4").split("-")]
K, T, variant = int(opt("--K", "4096")), int(opt("--T", "100")), opt("--variant", "multi2")
kw = {}
if layers != [6, 32, 32, 4]:
    l, th = P.synthetic_model(layers, seed=4)
    kw = dict(layers=l, theta=th)
cfg = S.make_config(K, T, track="oval", **kw)
sol = capi.Solver(cfg)
sol.set_rollout_variant(variant)
for _ in range(20):
    sol.compute_control(cfg["start_state"])
    sol.slide_control_seq(1)
sol.enable_stage_timing(1); sol.reset_stage_times()
for _ in range(20):
    sol.compute_control(cfg["start_state"])
st = sol.get_stage_times()
L = C.CDLL(lib)
buf = (C.c_ulonglong * 8)()
assert L.mppi_debug_read_multi_stamps(buf) == 0
n = max(1, buf[4])
out = {"workload": {"K": K, "T": T, "layers": cfg["layers"], "variant": sol.rollout_variant()},
       "rollout_kernel_ms_with_stamps": st["rollout_ms"] / max(1, st["n_solves"]),
       "cycles_per_step": {"record_publish_requests": buf[0] / n, "network_and_euler": buf[1] / n,
                           "operand_of_next_step_and_counters": buf[2] / n, "loop_back_edge": buf[3] / n,
                           "sum": (buf[0] + buf[1] + buf[2] + buf[3]) / n}}
print(json.dumps(out, indent=1))
