#!/usr/bin/env python3
"""tools/fuzz_sweep.py <first seed> <last seed>: the randomised whole-solve parity of tests/test_fuzz_gpu.py over OTHER seeds
(the committed test draws seeds 0..149), for spending idle GPU minutes.  Same generator, same criterion per draw:
applied controls bit-exact, flipped rollouts <= 3 %, |dU| <= 2e-4 + 4 x (weight mass of the flipped rollouts) x iterations.
Prints one line per failing draw (a candidate for a named regression test) and a summary; exit code 1 if any failed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz_gpu as F  # noqa: E402

gd = os.path.join(ROOT, "tests", "golden")
lo, hi = int(sys.argv[1]), int(sys.argv[2])
# a third argument "row": only the draws the vector-ALU row form serves (6-32-32-4, at most one group per CU), on that form
only_row = len(sys.argv) > 3 and sys.argv[3] == "row"
bad, explained, conditioned, illcond, worst_clean, forms = 0, 0, 0, 0, 0.0, {}
for seed in range(lo, hi):
    cfg, variant, hist = F._draw(gd, seed)
    if only_row:
        if cfg.get("bf_W") is not None or list(cfg["layers"]) != [6, 32, 32, 4] or cfg["K"] > 4096:
            continue
        variant = "row"
    iters = cfg["num_iters"]
    eps = F.noise_for(cfg)
    U0 = F.warm_U(cfg, seed=seed)
    ref = F.O.Oracle(cfg, fma_mode=1, nthreads=16).compute_control(cfg["start_state"], U0, hist, eps, num_iters=iters)
    sol = F.capi.Solver(cfg)
    try:
        sol.set_rollout_variant(variant)
    except F.capi.MppiError:
        pass
    sol.set_control_seq(U0)
    sol.set_control_hist(hist)
    sol.set_noise(eps)
    sol.compute_control(cfg["start_state"])
    got = sol.get_results()
    V = sol.get_applied_controls()
    name = sol.rollout_variant()
    sol.close()
    forms[name] = forms.get(name, 0) + 1
    err = F.rel_err(got["costs"], ref["costs"])
    flipped = err > 1e-4
    w = ref["w"] / ref["w"].sum()
    wg = got["w"] / got["w"].sum()
    mass = float(np.sum(np.maximum(w, wg)[flipped]))
    bound = 2e-4 + 4.0 * mass * iters
    dU = float(np.max(np.abs(got["U"] - ref["U"])))
    ok = float(np.mean(flipped)) <= 0.03 and dU <= bound
    if iters == 1:
        ok = ok and np.array_equal(V.view(np.uint32), ref["V"][-1].view(np.uint32))
    if mass == 0.0:
        worst_clean = max(worst_clean, dU)
    if not ok and iters > 1:
        # Several iterations: the flipped-rollout count is taken on the LAST iteration's costs, and those all move once
        # a weight-bearing rollout flipped in an EARLIER iteration (its dU feeds the next iteration's nominal controls).
        # Such a draw is explained if its first iteration alone meets the single-iteration criterion with flipped weight.
        c1 = dict(cfg, num_iters=1)
        r1 = F.O.Oracle(c1, fma_mode=1, nthreads=16).compute_control(c1["start_state"], U0, hist, eps[:1], num_iters=1)
        s1 = F.capi.Solver(c1)
        try:
            s1.set_rollout_variant(variant)
        except F.capi.MppiError:
            pass
        s1.set_control_seq(U0)
        s1.set_control_hist(hist)
        s1.set_noise(eps[:1])
        s1.compute_control(c1["start_state"])
        g1 = s1.get_results()
        V1 = s1.get_applied_controls()
        s1.close()
        f1 = F.rel_err(g1["costs"], r1["costs"]) > 1e-4
        m1 = float(np.sum(np.maximum(r1["w"] / r1["w"].sum(), g1["w"] / g1["w"].sum())[f1]))
        d1 = float(np.max(np.abs(g1["U"] - r1["U"])))
        if (np.array_equal(V1.view(np.uint32), r1["V"][-1].view(np.uint32)) and float(np.mean(f1)) <= 0.03 and m1 > 0.0
                and d1 <= 2e-4 + 4.0 * m1):
            explained += 1
            print("explained seed %d K=%d T=%d %s iters=%d: first iteration alone: %d flipped rollout(s) carrying %.3f of the "
                  "weight, dU=%.3e <= %.3e" % (seed, cfg["K"], cfg["T"], name, iters, int(f1.sum()), m1, d1, 2e-4 + 4.0 * m1), flush=True)
            ok = True
    if not ok and iters == 1 and float(np.mean(flipped)) <= 0.03 and np.array_equal(V.view(np.uint32), ref["V"][-1].view(np.uint32)):
        # Large costs x gamma: a last-digit cost difference (below the 1e-4 "flipped" mark) still moves the softmax.  First
        # order: dw_k/w_k = -gamma (dJ_k - sum_j w_j dJ_j), so |dU| <= 2 gamma sum_k w_k |dJ_k| max_k |V_k - U| -- the draw
        # is explained by its own measured cost differences if dU stays inside that.
        dJ = np.abs(got["costs"].astype(np.float64) - ref["costs"].astype(np.float64))
        S = float(cfg["gamma"]) * float(np.sum(w[~flipped] * dJ[~flipped]))
        R = float(np.max(np.abs(ref["V"][-1] - ref["U"][None])))
        if dU <= bound + 2.0 * S * R:
            conditioned += 1
            print("conditioned seed %d K=%d T=%d %s: gamma sum w|dJ| = %.3e (median cost %.0f), dU=%.3e <= %.3e" % (
                seed, cfg["K"], cfg["T"], name, S, float(np.median(ref["costs"])), dU, bound + 2.0 * S * R), flush=True)
            ok = True
    if not ok:
        # last resort: the oracle against itself -- its two arithmetic modes (explicit fmaf where nvcc contracts / none) on
        # this draw.  A draw whose own restatements disagree by more than the HIP path does is ill-conditioned (seed 14060:
        # eta = 1, two rollouts tie for the minimum cost to 3e-5 relative, the winner takes all the weight).
        r0 = F.O.Oracle(cfg, fma_mode=0, nthreads=16).compute_control(cfg["start_state"], U0, hist, eps, num_iters=iters)
        spread = float(np.max(np.abs(r0["U"] - ref["U"])))
        if dU <= spread:
            illcond += 1
            print("ill-conditioned seed %d K=%d T=%d %s iters=%d: dU=%.3e, the oracle's own two modes differ by %.3e (eta %.3f)" % (
                seed, cfg["K"], cfg["T"], name, iters, dU, spread, float(ref["w"].sum())), flush=True)
            ok = True
    if not ok:
        bad += 1
        print("BAD seed %d K=%d T=%d layers=%s %s iters=%d flipped=%.4f dU=%.3e bound=%.3e eta=%.3f" % (
            seed, cfg["K"], cfg["T"], cfg.get("layers"), name, iters, float(np.mean(flipped)), dU, bound, float(ref["w"].sum())), flush=True)
print("seeds %d..%d: %d draws, %d failed, %d multi-iteration draws explained by a weight-bearing flip in their first iteration, "
      "%d single-iteration draws inside the first-order bound of their own cost differences, %d draws on which the oracle's own two "
      "modes differ by more than the HIP path does, worst |dU| of a draw without flipped weight %.3e" % (
          lo, hi - 1, sum(forms.values()), bad, explained, conditioned, illcond, worst_clean))
print("kernel forms drawn:", ", ".join("%s x%d" % kv for kv in sorted(forms.items())))
sys.exit(1 if bad else 0)
