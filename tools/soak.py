#!/usr/bin/env python3
"""tools/soak.py [ticks]: long runs of the latency forms for rare hand-over failures.  Two handles with the same seed run
`ticks` solve + slide ticks each -- one through mppi_control_ticks, the other through batched ticks with a partner -- and must
end with bit-identical control sequences; further pairs run the other latency forms (K = 8192, 64-wide nets).  Any starved wave would surface as MPPI_ERR_HIP
(NaN normaliser); any race in the rings as a difference.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autorally_amd import capi, params as P, synthetic as S  # noqa: E402

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
out = {"ticks": ticks}
def net(layers):
    return dict(zip(("layers", "theta"), P.synthetic_model(layers, seed=4)))


# (the automatic forms: row-tree for 6-32-32-4 up to two groups per CU, the 4x4x1-MFMA form for 64-wide nets)
for name, kw, K in (("row", {}, 4096), ("row_k1920", {}, 1920), ("row_k8192", {}, 8192), ("multi4_tree_k16384", {}, 16384), ("h64", net([6, 64, 64, 4]), 2048),
                    ("h64x4_k1920", net([6, 64, 64, 64, 64, 4]), 1920)):
    cfg = S.make_config(K, 100, track="oval", **kw)
    a, b, partner = capi.Solver(cfg), capi.Solver(cfg), capi.Solver(dict(cfg, seed=77))
    st = cfg["start_state"]
    t0 = time.perf_counter()
    a.control_ticks(st, ticks, 1)
    t1 = time.perf_counter()
    chunk = 1000
    for _ in range(ticks // chunk):
        capi.control_ticks_batch([b, partner], [st, st], chunk, 1)
    t2 = time.perf_counter()
    ua, ub = a.get_control_seq(), b.get_control_seq()
    out[name] = {"variant": a.rollout_variant(), "identical": bool(np.array_equal(ua.view(np.uint32), ub.view(np.uint32))),
                 "finite": bool(np.all(np.isfinite(ua))), "single_ms_per_tick": 1e3 * (t1 - t0) / ticks,
                 "batched_pair_ms_per_tick": 1e3 * (t2 - t1) / ticks}
    for s in (a, b, partner):
        s.close()
print(json.dumps(out))
ok = all(v["identical"] and v["finite"] for k, v in out.items() if isinstance(v, dict))
sys.exit(0 if ok else 1)
