#!/usr/bin/env python3
"""tools/fuzz_sweep.py <first seed> <last seed> [row | row_tree]: the randomised whole-solve parity of tests/test_fuzz_gpu.py over
OTHER seeds (the committed test draws seeds 0..149), for spending idle GPU minutes.  Same generator, same criterion -- per
ITERATION, teacher-forced: the oracle's iteration i is started from the HIP path's own U after iteration i-1
(mppi_debug_capture_iterations; tests/helpers.py), so every iteration is a comparison on identical inputs:
applied controls bit-exact, flipped rollouts <= 3 %, |dU| <= 2e-4 + 4 x (weight mass of the flipped rollouts).
An iteration outside that is then classified, in this order:
  conditioned      inside the first-order bound of its own cost differences (tests/helpers.py: first_order_bound): each capped
                   at the a-priori 3e-6 relative (large costs x gamma: last-digit cost differences move the softmax), except
                   at most max(2, K/200) "gray-zone" rollouts -- threshold flips of small effect, 1e-5 < relative <= 1e-4;
  granularity      more than 3 % flipped but no more than 4 rollouts (K = 64, 128), controls inside the bound;
  ill-conditioned  the oracle's own two arithmetic modes (fmaf where nvcc contracts / none), teacher-forced the same way,
                   differ from each other by more than the HIP path differs from the nearer one;
  BAD              none of these: a candidate for a named regression test or a fix.
Third argument "row" / "row_tree": only the draws the vector-ALU row form serves (6-32-32-4, at most one group per CU), on
that form ("row_tree": against the NOMINAL oracle, i.e. the north-star tolerance for the re-associated output layer);
"multi4_tree[_gen]": the draws of the shapes the multi form has, forced onto its butterfly-output form, against the nominal oracle;
"m44": only the 64-wide draws (6-64-64-4, 6-64-64-64-64-4, K <= 4096) on the 4x4x1-MFMA form, likewise against the nominal oracle.
Prints one line per draw that needed a classification and a summary; exit code 1 if any draw is BAD.  FUZZ_SECONDS=<s> in the
environment stops the sweep after that time (the summary names the last seed done)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz_gpu as F  # noqa: E402
from tests.helpers import solve_with_iterations, teacher_forced_iterations  # noqa: E402

gd = os.path.join(ROOT, "tests", "golden")
lo, hi = int(sys.argv[1]), int(sys.argv[2])
only_row = sys.argv[3] if len(sys.argv) > 3 else None
assert only_row in (None, "row", "row_tree", "m44", "multi4_tree", "multi4_tree_gen")
WANT = {"m44": ([6, 64, 64, 4], [6, 64, 64, 64, 64, 4]),
        "multi4_tree": ([6, 32, 32, 4], [6, 32, 32, 32, 32, 4], [6, 64, 64, 4]),
        "multi4_tree_gen": ([6, 32, 32, 4], [6, 32, 32, 32, 32, 4], [6, 64, 64, 4])}.get(only_row, ([6, 32, 32, 4],))
KMAX = 1 << 30 if (only_row or "").startswith("multi4") else 4096  # the latency forms serve up to 4096 (8192) rollouts
n_draws = n_iter = bad = conditioned = granular = illcond = 0
worst_clean, worst_any, forms = 0.0, 0.0, {}
t_stop = time.time() + float(os.environ.get("FUZZ_SECONDS", "1e9"))  # optional time budget: stop early, summary over the seeds done
last = lo - 1
for seed in range(lo, hi):
    if time.time() > t_stop:
        break
    last = seed
    if seed % 500 == 0:
        print("# at seed %d: %d draws, %d BAD" % (seed, n_draws, bad), file=sys.stderr, flush=True)
    if only_row:  # the first three draws of F._draw decide whether the row form serves this seed: skip the rest cheaply
        r = np.random.RandomState(seed)
        K_ = 64 * int(r.choice([1, 2, 3, 5, 8, 16, 17, 32, 64, 65, 100]))
        r.choice([2, 3, 5, 9, 16, 20, 33, 47, 60, 100])
        lay = F.LAYERS[r.randint(len(F.LAYERS))]
        if (lay if lay is not None else [6, 32, 32, 4]) not in WANT or K_ > KMAX:
            continue
    cfg, variant, hist = F._draw(gd, seed)
    if only_row:
        if cfg.get("bf_W") is not None or list(cfg["layers"]) not in WANT or cfg["K"] > KMAX:
            continue
        variant = only_row
    iters = cfg["num_iters"]
    eps = F.noise_for(cfg)
    U0 = F.warm_U(cfg, seed=seed)
    got, its, name = solve_with_iterations(cfg, variant, U0, hist, eps)
    forms[name] = forms.get(name, 0) + 1
    n_draws += 1
    ms = teacher_forced_iterations(cfg, got, its, U0, hist, eps, fma_mode=1)
    ms0 = None
    for i, m in enumerate(ms):
        n_iter += 1
        bound = 2e-4 + 4.0 * m["mass"]
        dU = max(m["dU"], m.get("dU_smoothed", 0.0))
        worst_any = max(worst_any, dU)
        if m["mass"] == 0.0 and m["V_equal"] and m["flipped"] <= 0.03 and dU <= bound:
            worst_clean = max(worst_clean, dU)
        if m["V_equal"] and m["flipped"] <= 0.03 and dU <= bound:
            continue
        head = "seed %d K=%d T=%d %s iteration %d of %d (gamma %g, eta %.3f, median cost %.0f)" % (
            seed, cfg["K"], cfg["T"], name, i + 1, iters, cfg["gamma"], m["eta"], m["median_cost"])
        few_gray = m["n_gray"] <= max(2, cfg["K"] // 200)
        if m["V_equal"] and m["flipped"] <= 0.03 and few_gray and dU <= bound + m["first_order"]:
            conditioned += 1
            print("conditioned %s: dU=%.3e <= %.3e = 2e-4 + 4 x %.2e + first-order %.3e (%d gray-zone rollout(s))" % (
                head, dU, bound + m["first_order"], m["mass"], m["first_order"], m["n_gray"]), flush=True)
            continue
        if m["V_equal"] and m["n_flipped"] <= 4 and few_gray and dU <= bound + m["first_order"]:
            granular += 1
            print("granularity %s: %d flipped rollouts = %.1f %% of K, dU=%.3e <= %.3e" % (head, m["n_flipped"], 100 * m["flipped"], dU, bound + m["first_order"]), flush=True)
            continue
        if ms0 is None:
            ms0 = teacher_forced_iterations(cfg, got, its, U0, hist, eps, fma_mode=0)
            # the oracle against itself on the same inputs: mode 0 started from the same U as mode 1 was
            c1 = dict(cfg, num_iters=1)
            o1, o0 = F.O.Oracle(c1, fma_mode=1, nthreads=16), F.O.Oracle(c1, fma_mode=0, nthreads=16)
        U_in = U0 if i == 0 else its["U_raw"][i - 1]
        Us = []
        for o in (o1, o0):
            c_, V_, _ = o.rollouts(cfg["start_state"], U_in, eps[i])
            w_, _, e_, _ = o.weights(c_)
            Us.append(o.weighted_reduction(w_, e_, V_))
        spread = float(np.max(np.abs(Us[0] - Us[1])))
        d0 = max(ms0[i]["dU"], ms0[i].get("dU_smoothed", 0.0))
        if m["V_equal"] and min(dU, d0) <= spread:
            illcond += 1
            print("ill-conditioned %s: dU=%.3e / %.3e against the oracle's two modes, which differ by %.3e from each other" % (head, dU, d0, spread), flush=True)
            continue
        bad += 1
        print("BAD %s: V_equal=%s flipped=%.4f (%d) mass=%.3e dU=%.3e bound=%.3e first-order=%.3e" % (
            head, m["V_equal"], m["flipped"], m["n_flipped"], m["mass"], dU, bound, m["first_order"]), flush=True)
print("seeds %d..%d: %d draws, %d iterations, each compared with the oracle on identical inputs: %d BAD; %d conditioned (first-order bound "
      "of their own cost differences, capped at 3e-6 relative), %d granularity (<= 4 flipped rollouts are > 3 %% of K), %d ill-conditioned "
      "(the oracle's two modes differ by more); worst |dU| of a clean iteration without flipped weight %.3e, of any iteration %.3e" % (
          lo, last, n_draws, n_iter, bad, conditioned, granular, illcond, worst_clean, worst_any))
print("kernel forms drawn:", ", ".join("%s x%d" % kv for kv in sorted(forms.items())))
sys.exit(1 if bad else 0)
