"""Shared helpers for the parity tests (tests/ only)."""
import numpy as np

from autorally_amd import synthetic as S
from oracle import oracle as O


def oracle_mode_for(variant_name):
    """The oracle's arithmetic mode (orc_problem.fma_mode) that states the summation order of a kernel form, by the name
    mppi_rollout_variant gives it: 1 = the reference's order in every layer (every "exact" form); the tree forms re-associate
    the OUTPUT layer only -- 2: row-tree / row64 butterflies, 3: the 4x4x1-MFMA form with one chain per hidden layer ("m44_chain"),
    4: the multi4-tree form; 5: the automatic 4x4x1-MFMA form, whose hidden layers are two accumulation chains as well."""
    if "m44_split" in variant_name:
        return 5
    if "m44" in variant_name:
        return 3
    if "multi4_tree" in variant_name:
        return 4
    return 2 if "_tree" in variant_name else 1


def noise_for(cfg, seed=1234, iters=None):
    """Explicit eps[num_iters][K][T][2] from the oracle's statement of the noise spec."""
    it = int(cfg.get("num_iters", 1)) if iters is None else iters
    K, T = cfg["K"], cfg["T"]
    out = np.zeros((it, K, T, 2), dtype=np.float32)
    for i in range(it):
        out[i] = O.generate_noise(seed, 2 * T * i, K, T)
    return out


def warm_U(cfg, seed=7):
    """A non-trivial nominal control sequence (smooth, inside the limits)."""
    T = cfg["T"]
    t = np.arange(T, dtype=np.float64)
    rng = np.random.RandomState(seed)
    # negative steering turns left; -0.28 / 0.22 keeps the car on the radius-10 ring (probed with
    # the oracle's nominal trajectory); the oval starts on a straight
    base = -0.28 if cfg.get("track") == "ring" else 0.0
    thr = 0.22 if cfg.get("track") == "ring" else 0.3
    U = np.stack([base + 0.05 * np.sin(t / 9.0 + rng.uniform(0, 1)), thr + 0.05 * np.cos(t / 13.0)], axis=1)
    return U.astype(np.float32)


def rel_err(a, b, floor=1.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def load_nn_golden(golden_dir):
    """The reference-generated NN vectors as one dict: the shipped models read through the reference's
    reader (gen_golden.py) and the model written by its training pipeline's writer
    (gen_model_writer_golden.py)."""
    import os
    g = {}
    for f in ("nn_dynamics_golden.npz", "nn_writer_golden.npz"):
        z = np.load(os.path.join(golden_dir, f))
        g.update({k: z[k] for k in z.files})
    return g


COST_RTOL = 3e-6   # a-priori relative tolerance of a rollout cost away from the discontinuities (tanh / sincos / atan forms)
GRAY_RTOL = 1e-5   # beyond this (and up to the 1e-4 "flipped" mark) a cost difference is a threshold flip of SMALL effect


def first_order_bound(gamma, w_norm, dJ, costs_ref, flipped, V_ref, U_ref):
    """|dU| a softmax-weighted mean can move, to first order, when the costs move by dJ: dw_k / w_k = -gamma (dJ_k - sum_j w_j
    dJ_j), hence |dU| <= 2 gamma sum_k w_k |dJ_k| max_k |V_k - U|.  Large costs x gamma make this exceed the 2e-4 of a clean
    draw although every cost agrees to the last digits.  Three classes of rollouts:
      * relative difference <= GRAY_RTOL: smooth arithmetic -- |dJ_k| is the MEASURED difference but never more than
        COST_RTOL |J_k|, so the bound does not widen with the device's error beyond the a-priori tolerance;
      * GRAY_RTOL < relative difference <= 1e-4 ("gray"): a nearest-texel / crash / slip threshold flipped on ONE or a few
        steps of a rollout whose cost is large (crashed rollouts: ~1e3), so the flip stays under the 1e-4 mark of a "flipped"
        rollout although it is one (sweep seeds 201853: 1.5e-5 of 2 262 on a rollout with 1.8 % of the weight; 209729: 5e-5 of
        240 on the rollout with 38 %) -- entered with its measured |dJ_k|; callers limit how many there may be;
      * beyond 1e-4: accounted for by their weight mass (not here).
    What bounds the gray allowance a priori (ADVICE round 4 asked): a gray rollout's |dJ_k| is at most 1e-4 |J_k| by the
    definition of the class, and callers admit at most max(2, K / 200) of them -- so the bound cannot grow with the device's
    error beyond 2 gamma x (their weight) x 1e-4 |J| x max|V - U|.  Requiring instead that the oracle's own two arithmetic modes
    disagree on such a rollout (a "proven" flip) does not work: on both named draws they AGREE to 1e-6 on it
    (test_gray_zone_flip_on_a_weight_bearing_rollout asserts that) -- the flip is libm's tanh / sincos against the device's
    forms, which no mode of the oracle reproduces.  What would a real arithmetic regression look like?  It moves MANY costs by a
    similar relative amount: the p99 < 5e-6 criterion of the fixed cases and the count limit here catch that.
    Returns (bound, number of gray rollouts)."""
    keep = ~flipped
    ref = np.maximum(np.abs(costs_ref.astype(np.float64)), 1.0)
    a = np.abs(dJ)
    gray = keep & (a > GRAY_RTOL * ref)
    d = np.where(gray, a, np.minimum(a, COST_RTOL * ref))
    return 2.0 * float(gamma) * float(np.sum(w_norm[keep] * d[keep])) * float(np.max(np.abs(V_ref - U_ref[None]))), int(gray.sum())


def solve_with_iterations(cfg, variant, U0, hist, eps):
    """One HIP solve with every iteration captured (mppi_debug_capture_iterations): (results, iterations, variant name)."""
    from autorally_amd import capi
    sol = capi.Solver(cfg)
    try:
        sol.set_rollout_variant(variant)
    except capi.MppiError:
        pass  # a form this layer list does not have: the automatic choice stays
    sol.debug_capture_iterations(1)
    sol.set_control_seq(U0)
    sol.set_control_hist(hist)
    sol.set_noise(eps)
    sol.compute_control(cfg["start_state"])
    got = sol.get_results()
    its = sol.debug_get_iterations(with_V=True)
    name = sol.rollout_variant()
    sol.close()
    return got, its, name


def teacher_forced_iterations(cfg, got, its, U0, hist, eps, fma_mode=1, nthreads=16):
    """Per-iteration comparison on IDENTICAL inputs (mppi_controller.cu:609-667 re-uses the raw weighted mean U_ as the next
    iteration's nominal sequence): iteration i of the HIP solve started from ITS OWN U after iteration i-1, so the oracle's
    iteration i is started from that same U.  Returns one dict per iteration: V_equal, flipped (fraction of rollouts beyond
    1e-4 relative), mass (weight they carry), dU (raw weighted mean, L-inf), first_order (2 gamma sum w|dJ| max|V - U| over the
    rollouts that did not flip), eta, and for the last iteration dU_smoothed / d_traj_cost."""
    iters = int(cfg.get("num_iters", 1))
    c1 = dict(cfg, num_iters=1)
    orc = O.Oracle(c1, fma_mode=fma_mode, nthreads=nthreads)
    out = []
    for i in range(iters):
        U_in = np.asarray(U0, np.float32) if i == 0 else its["U_raw"][i - 1]
        costs_o, V_o, _ = orc.rollouts(cfg["start_state"], U_in, eps[i])
        w_o, _, eta, tc = orc.weights(costs_o)
        U_o = orc.weighted_reduction(w_o, eta, V_o)
        w_g, _, eta_g, tc_g = orc.weights(its["costs"][i])
        err = rel_err(its["costs"][i], costs_o)
        fl = err > 1e-4
        wn, wgn = w_o / w_o.sum(), w_g / w_g.sum()
        dJ = np.abs(its["costs"][i].astype(np.float64) - costs_o.astype(np.float64))
        fo, n_gray = first_order_bound(cfg["gamma"], wn, dJ, costs_o, fl, V_o, U_o)
        m = {"V_equal": bool(np.array_equal(its["V"][i].view(np.uint32), V_o.view(np.uint32))),
             "flipped": float(np.mean(fl)), "n_flipped": int(fl.sum()),
             "mass": float(np.sum(np.maximum(wn, wgn)[fl])),
             "dU": float(np.max(np.abs(its["U_raw"][i] - U_o))),
             "first_order": fo, "n_gray": n_gray,
             "K": int(cfg["K"]), "eta": float(eta), "p99": float(np.percentile(err, 99)), "median_cost": float(np.median(costs_o))}
        if i == iters - 1:
            m["dU_smoothed"] = float(np.max(np.abs(got["U"] - orc.savgol(U_o, hist))))
            m["d_traj_cost"] = abs(got["traj_cost"] - tc) / max(abs(tc), 1e-3)
        out.append(m)
    return out


def iteration_ok(m, with_first_order=True):
    """The single-iteration criterion of tests/test_fuzz_gpu.py on one teacher-forced iteration."""
    bound = 2e-4 + 4.0 * m["mass"] + (m["first_order"] if with_first_order else 0.0)
    return m["V_equal"] and m["flipped"] <= 0.03 and m["dU"] <= bound and m.get("dU_smoothed", 0.0) <= bound and \
        (not with_first_order or m["n_gray"] <= max(2, int(m["K"]) // 200))
