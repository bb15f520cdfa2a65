#!/usr/bin/env python3
"""tools/stream_soak.py [seconds] [first seed]: randomised many-chunk solves (K > 4096: the one-launch streaming tail with its
in-launch granule hand-overs, beta out of the rollout kernel where the form is one of rollout_multi.hip's (beyond 8192 rollouts),
the generator kernel beside or behind the rollout, chained ticks)
for spending idle GPU minutes.  Every draw takes a shape -- K a multiple of 64 in (4096, 49152], T in [4, 160], 32- or 64-wide
net, stride 1 or 2 -- and runs the SAME seeded tick sequence on three handles:
  A  the product's way: mppi_control_ticks (chained where the form allows), beta from the rollout kernel;
  B  the same ticks one by one (compute_control + slide), beta from the rollout kernel;
  C  one by one, beta reduced by the tail kernel itself (mppi_debug_min_cost off: the first version's hand-overs).
The control sequences after the last tick, and every tick's trajectory cost on B and C, must agree bit for bit; anything a race, a
stale tag or an arrival-order dependence could change would show as a difference, a starved wait as MPPI_ERR_HIP.  Prints one line
per mismatch and one JSON summary; exit code 1 on any mismatch."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autorally_amd import capi, params as P, synthetic as S  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
t_note = time.time() + 60.0
nets = {32: None, 64: dict(zip(("layers", "theta"), P.synthetic_model([6, 64, 64, 4], seed=4)))}
draws = bad = ticks_total = 0
base = {}
forms, chunks = {}, {}
seed = seed0
while time.time() < t_end:
    rng = np.random.RandomState(seed)
    K = 64 * int(rng.randint(4096 // 64 + 1, 49152 // 64 + 1))
    T = int(rng.randint(4, 161))
    width = 32 if rng.rand() < 0.6 else 64
    if width == 64 and K * T > 16384 * 150:  # keep a draw to a fraction of a second
        K = 64 * int(rng.randint(65, 16384 // 64 + 1))
    stride = int(rng.randint(1, 3))
    n = int(rng.randint(2, 7))
    kw = nets[width] or {}
    track = "oval" if rng.rand() < 0.7 else "ring"
    if track not in base:
        base[track] = S.make_config(64, 4, track=track)  # the map once per track
    cfg = dict(base[track], K=K, T=T, opt_stride=stride, gamma=float(rng.choice([0.15, 0.05, 0.5])), **kw)
    rng_seed = int(rng.randint(1, 1 << 30))
    st = np.asarray(cfg["start_state"], np.float32)
    a, b, c = capi.Solver(cfg), capi.Solver(cfg), capi.Solver(cfg)
    c.debug_min_cost(0)
    for s_ in (a, b, c):
        s_.seed(rng_seed)
    try:
        a.control_ticks(st, n, stride)
        tb, tc = [], []
        for i in range(n):
            for s, acc in ((b, tb), (c, tc)):
                s.compute_control(st)
                acc.append(s.get_results(with_vectors=False)["traj_cost"])
                s.slide_control_seq(stride)
        publishes = "multi" in b.rollout_variant()  # (the row / m44 forms of K <= 8192 leave beta to the tail kernel)
        assert b.debug_min_cost() == publishes and not c.debug_min_cost(), "beta source"
        ua, ub, uc = a.get_control_seq(), b.get_control_seq(), c.get_control_seq()
        ok = np.array_equal(ua.view(np.uint32), ub.view(np.uint32)) and np.array_equal(ub.view(np.uint32), uc.view(np.uint32)) and tb == tc \
            and bool(np.all(np.isfinite(ua)))
    except Exception as e:  # MPPI_ERR_HIP from a starved wait, or the assertion above
        ok = False
        print("seed %d K=%d T=%d width=%d stride=%d n=%d: %s" % (seed, K, T, width, stride, n, e))
    if not ok:
        bad += 1
        print("MISMATCH seed %d K=%d T=%d width=%d stride=%d n=%d form %s" % (seed, K, T, width, stride, n, a.rollout_variant()))
    forms[a.rollout_variant()] = forms.get(a.rollout_variant(), 0) + 1
    nchunks = (K + 4095) // 4096
    chunks[nchunks] = chunks.get(nchunks, 0) + 1
    for s in (a, b, c):
        s.close()
    draws += 1
    ticks_total += 3 * n
    seed += 1
    if time.time() > t_note:  # a line a minute: a silent run is taken for a hung one on the GPU pool
        print("# %d draws, %d mismatches" % (draws, bad), file=sys.stderr, flush=True)
        t_note = time.time() + 60.0
print(json.dumps({"draws": draws, "first_seed": seed0, "solves": ticks_total, "mismatches": bad, "forms": forms,
                  "chunks_per_row": {str(k): v for k, v in sorted(chunks.items())}}))
sys.exit(1 if bad else 0)
