"""C-ABI behaviour on the GPU: error convention, determinism, live updates, concurrent handles,
edge sizes, and size-independent properties at BASELINE.json's largest configuration."""
import ctypes as C
import threading

import numpy as np
import pytest

from autorally_amd import capi
from autorally_amd import params as P
from autorally_amd import synthetic as S
from oracle import oracle as O
from tests.helpers import noise_for, rel_err, warm_U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from autorally_amd import build as B
    B.build()
    assert capi.lib().mppi_device_count() >= 1


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_error_convention_and_call_order():
    L = capi.lib()
    cfg = S.make_config(128, 20)
    c = capi.make_config_struct(cfg)
    h = C.c_void_p()
    assert L.mppi_create(C.byref(c), C.byref(h)) == capi.OK
    state = cfg["start_state"].astype(np.float32)
    # solve before the model / costmap / cost parameters were set: refused, nothing computed
    assert L.mppi_compute_control(h, _fp(state)) == capi.ERR_STATE
    assert b"mppi_set_nn_params" in L.mppi_last_error(h)
    theta = np.ascontiguousarray(cfg["theta"], np.float32)
    assert L.mppi_set_nn_params(h, _fp(theta), theta.size - 1) == capi.ERR_INVALID
    assert L.mppi_set_nn_params(h, _fp(theta), theta.size) == capi.OK
    assert L.mppi_compute_control(h, _fp(state)) == capi.ERR_STATE
    U = np.full((20, 2), 7.0, np.float32)
    assert L.mppi_get_control_seq(h, _fp(U), 39) == capi.ERR_INVALID
    assert np.all(U == 7.0)  # outputs untouched on error
    assert L.mppi_set_noise(h, _fp(np.zeros(10, np.float32)), 10) == capi.ERR_INVALID
    assert L.mppi_slide_control_seq(h, -1) == capi.ERR_INVALID
    assert L.mppi_slide_control_seq(h, 21) == capi.ERR_INVALID
    assert L.mppi_set_rollout_variant(h, b"nonsense") == capi.ERR_INVALID
    assert L.mppi_destroy(h) == capi.OK
    assert L.mppi_destroy(None) == capi.ERR_INVALID


def test_determinism_and_seeds():
    cfg = S.make_config(1024, 40, track="ring")
    res = []
    for seed in (1234, 1234, 99):
        sol = capi.Solver(dict(cfg, seed=seed))
        sol.compute_control(cfg["start_state"])
        sol.slide_control_seq(1)
        sol.compute_control(cfg["start_state"])
        res.append(sol.get_results()["U"].copy())
        sol.close()
    np.testing.assert_array_equal(res[0].view(np.uint32), res[1].view(np.uint32))
    assert np.max(np.abs(res[0] - res[2])) > 1e-4


def test_create_destroy_loop_and_many_handles():
    cfg = S.make_config(256, 20)
    sols = [capi.Solver(cfg) for _ in range(6)]
    for s in sols:
        s.compute_control(cfg["start_state"])
    u0 = sols[0].get_results(False)["U"]
    for s in sols[1:]:
        np.testing.assert_array_equal(s.get_results(False)["U"], u0)  # identical generators (seed 1234)
    for s in sols:
        s.close()
    for _ in range(20):
        s = capi.Solver(cfg)
        s.compute_control(cfg["start_state"])
        s.close()


def test_two_handles_driven_from_two_threads():
    """Distinct handles may be driven concurrently from distinct host threads (one per controller /
    per GPU): results equal the sequential ones."""
    cfgs = [S.make_config(2048, 60, track="oval", instance=i, seed=50 + i) for i in range(2)]
    seq = []
    for cfg in cfgs:
        s = capi.Solver(cfg)
        for _ in range(3):
            s.compute_control(cfg["start_state"])
            s.slide_control_seq(1)
        seq.append(s.get_control_seq())
        s.close()
    out = [None, None]

    def work(i):
        s = capi.Solver(cfgs[i])
        for _ in range(3):
            s.compute_control(cfgs[i]["start_state"])
            s.slide_control_seq(1)
        out[i] = s.get_control_seq()
        s.close()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        np.testing.assert_array_equal(out[i].view(np.uint32), seq[i].view(np.uint32))


def test_update_model_and_costmap_and_limits(golden_dir):
    import os
    cfg = S.make_config(512, 40, track="ring")
    eps = noise_for(cfg)
    U0 = warm_U(cfg)
    layers2, theta2 = P.load_model_npz(os.path.join(golden_dir, "models", "gazebo_nnet_09_12_2018.npz"))
    # [W1|W2|W3|b1|b2|b3] layout of updateModel (neural_net_model.cu:152-180)
    z = np.load(os.path.join(golden_dir, "models", "gazebo_nnet_09_12_2018.npz"))
    data = np.concatenate([z["dynamics_W%d" % i].astype(np.float32).ravel() for i in (1, 2, 3)] +
                          [z["dynamics_b%d" % i].astype(np.float32).ravel() for i in (1, 2, 3)])

    def solve(sol):
        sol.set_control_seq(U0)
        sol.set_noise(eps)
        sol.compute_control(cfg["start_state"])
        return sol.get_results()

    a = capi.Solver(cfg)
    r0 = solve(a)
    a.update_model([6, 32, 32, 4], data)
    r1 = solve(a)
    b = capi.Solver(dict(cfg, theta=theta2))
    r2 = solve(b)
    assert np.max(np.abs(r0["costs"] - r1["costs"])) > 1e-2
    np.testing.assert_array_equal(r1["costs"].view(np.uint32), r2["costs"].view(np.uint32))
    with pytest.raises(capi.MppiError):
        a.update_model([6, 16, 32, 4], data)  # structure mismatch: rejected, model unchanged
    np.testing.assert_array_equal(solve(a)["costs"].view(np.uint32), r2["costs"].view(np.uint32))
    # costmap channel 0 swap == fresh solver with that map
    ch0, xb, yb, ppm = S.gaussian_ring_map(radius=11.0)
    a.set_costmap_channel(0, ch0)
    m2 = cfg["map_rgba"].copy()
    m2[:, :, 0] = ch0
    c = capi.Solver(dict(cfg, theta=theta2, map_rgba=m2))
    np.testing.assert_array_equal(solve(a)["costs"].view(np.uint32), solve(c)["costs"].view(np.uint32))
    # cutThrottle (mppi_controller.cu:460-466): max throttle 0 and desired speed 0
    a.set_control_limits([-0.99, -0.99], [0.99, 0.0])
    a.set_cost_params(dict(cfg["cost"], desired_speed=0.0))
    solve(a)
    _, cs = a.nominal_traj(cfg["start_state"])
    assert cs[:, 1].max() <= 0.0
    orc = O.Oracle(dict(cfg, theta=theta2, map_rgba=m2, u_hi=(0.99, 0.0), cost=dict(cfg["cost"], desired_speed=0.0)))
    ref = orc.compute_control(cfg["start_state"], U0, np.zeros(4, np.float32), eps)
    assert np.max(np.abs(a.get_results(False)["U"] - ref["U"])) <= 1e-4
    for s in (a, b, c):
        s.close()


@pytest.mark.parametrize("K,T", [(64, 2), (64, 11), (128, 10), (192, 29), (64, 203)])
def test_edge_sizes(K, T):
    """Smallest K, T = 2, T around the control wave's 4-step chunks and the 16-step LDS rings."""
    cfg = S.make_config(K, T, track="ring")
    eps = noise_for(cfg)
    U0 = warm_U(cfg)
    ref = O.Oracle(cfg).compute_control(cfg["start_state"], U0, np.zeros(4, np.float32), eps)
    for variant in ("row", "quad", "fused", "valu"):
        sol = capi.Solver(cfg)
        sol.set_rollout_variant(variant)
        sol.set_control_seq(U0)
        sol.set_noise(eps)
        sol.compute_control(cfg["start_state"])
        got = sol.get_results()
        np.testing.assert_array_equal(sol.get_applied_controls().view(np.uint32), ref["V"][-1].view(np.uint32))
        assert float(np.percentile(rel_err(got["costs"], ref["costs"]), 95)) < 1e-5, variant
        assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4, variant
        # generator mode on the same handle (in-kernel noise for quad, stand-alone kernel otherwise)
        sol.seed(1234, 0)
        sol.set_control_seq(U0)
        sol.compute_control(cfg["start_state"])
        np.testing.assert_array_equal(sol.get_results()["costs"].view(np.uint32), got["costs"].view(np.uint32))
        sol.close()


def test_baseline_config4_full_size_parity():
    """BASELINE.json configs[3] at FULL size -- K=16384, T=150, 6-64-64-4 (seed-4 weights, oval map, warm U) --
    against the CPU oracle's complete solve (mppi_controller.cu:600-675; it needs a few seconds on 8-16
    threads) with the criteria of test_rollout_costs_and_controls: applied controls bit-exact, cost p99 < 5e-6,
    flipped rollouts <= K/200 moving < 1e-4 of the weight mass, U L-inf <= 1e-4, trajectory cost rel <= 1e-4.
    The multi4 form + generator kernel on explicit noise; the other exact kernel forms must equal it bit for bit; the
    AUTOMATIC form (round 4: multi4 with the output layer as a butterfly, "multi4_tree_gen") against its own oracle mode (4)
    at the same criteria and against the nominal oracle at the north-star ones, on explicit noise AND in generator mode (the
    device's own MRG32k3a draws against O.generate_noise); plus the size-independent properties (softmax, hull, float64
    weighted mean)."""
    K, T = 16384, 150
    layers, theta = P.synthetic_model([6, 64, 64, 4], seed=4)
    cfg = S.make_config(K, T, layers=layers, theta=theta, track="oval")
    U0 = warm_U(cfg)
    hist = np.zeros(4, np.float32)
    rng = np.random.RandomState(5)
    eps = rng.standard_normal((1, K, T, 2)).astype(np.float32)
    orc = O.Oracle(cfg, fma_mode=1, nthreads=16)
    ref = orc.compute_control(cfg["start_state"], U0, hist, eps)
    orc4 = O.Oracle(cfg, fma_mode=4, nthreads=16)  # the butterfly's summation order (tests/helpers.py: oracle_mode_for)
    ref4 = orc4.compute_control(cfg["start_state"], U0, hist, eps)

    def check(got, ref, what):
        np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32), err_msg=what)
        err = rel_err(got["costs"], ref["costs"])
        assert int(np.sum(err > 1e-4)) <= K // 200, (what, int(np.sum(err > 1e-4)), float(err.max()))
        assert float(np.percentile(err, 99)) < 5e-6, what
        assert float(np.abs(got["w"] - ref["w"]).sum()) / float(ref["w"].sum()) < 1e-4, what
        assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4, what
        assert abs(got["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"]), what

    outs = {}
    for variant in ("auto", "multi4_gen", "quad", "fused", "multi4"):
        sol = capi.Solver(cfg)
        sol.set_rollout_variant(variant)
        sol.set_control_seq(U0)
        sol.set_noise(eps)
        sol.compute_control(cfg["start_state"])
        outs[variant] = dict(sol.get_results(), V=sol.get_applied_controls(), name=sol.rollout_variant())
        if variant == "auto":
            # generator mode on the same handle: the stand-alone generator kernel's draws (prefetch path included:
            # two solves, the second one consumes prefetched draws) against the oracle's statement of the spec
            sol.seed(4321, 0)
            for it in range(2):
                sol.set_control_seq(U0)
                sol.compute_control(cfg["start_state"])
                gen = dict(sol.get_results(), V=sol.get_applied_controls())
                eps_g = O.generate_noise(4321, 2 * T * it, K, T)[None]
                check(gen, orc4.compute_control(cfg["start_state"], U0, hist, eps_g), "generator mode, solve %d" % it)
                nom = orc.compute_control(cfg["start_state"], U0, hist, eps_g)
                assert np.max(np.abs(gen["U"] - nom["U"])) <= 1e-4 and abs(gen["traj_cost"] - nom["traj_cost"]) <= 1e-4 * abs(nom["traj_cost"])
        sol.close()
    t = outs["auto"]
    # auto: four dynamics waves + pose / fetch / cost / control waves per 64 rollouts, eps from the generator kernel, butterfly output
    assert "multi4_tree_gen" in t["name"]
    check(t, ref4, "tree form, explicit noise, its own oracle mode")
    assert np.max(np.abs(t["U"] - ref["U"])) <= 1e-4 and abs(t["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"])
    assert int(np.sum(rel_err(t["costs"], ref["costs"]) > 1e-4)) <= K // 200
    a = outs["multi4_gen"]
    assert "multi4_gen" in a["name"] and "quad" in outs["quad"]["name"] and "fused" in outs["fused"]["name"]
    check(a, ref, "explicit noise")
    # non-degenerate: most rollouts stay on the track and the weights are spread
    assert float(np.mean(ref["costs"] < 5000.0)) > 0.5 and float(ref["w"].sum()) > 4.0
    for v in ("quad", "fused", "multi4"):
        np.testing.assert_array_equal(a["costs"].view(np.uint32), outs[v]["costs"].view(np.uint32))
        np.testing.assert_array_equal(a["V"].view(np.uint32), outs[v]["V"].view(np.uint32))
        np.testing.assert_array_equal(a["U"].view(np.uint32), outs[v]["U"].view(np.uint32))
    w = a["w"]
    assert w.max() == 1.0 and np.all(w >= 0) and np.all(np.isfinite(a["U"]))
    eta = float(w.astype(np.float64).sum())
    # weighted mean in float64 from the GPU's own weights and applied controls, then the oracle's filter
    Uw = np.einsum("k,ktj->tj", w.astype(np.float64) / eta, a["V"].astype(np.float64))
    Us = orc.savgol(Uw.astype(np.float32), hist)
    assert np.max(np.abs(Us - a["U"])) < 5e-5
    assert abs(float((w.astype(np.float64) ** 2).sum() / eta) - a["traj_cost"]) < 1e-4 * a["traj_cost"]
    assert a["V"][..., 0].min() <= a["U"][:, 0].min() and a["U"][:, 0].max() <= a["V"][..., 0].max()


def test_debug_cost_raster_matches_oracle():
    """MPPICosts::getDebugDisplay's raster (debug_kernels.cuh:39-88): costmap window around the car with
    the car marker; pixels on a texel edge or on the marker's outline may flip on a last-digit difference."""
    cfg = S.make_config(64, 4, track="oval")
    sol = capi.Solver(cfg)
    orc = O.Oracle(cfg)
    for (x, y, hd, wm, hm, ppm) in [(0.0, -10.0, 0.3, 10, 10, 50), (3.0, -9.0, 2.5, 6, 4, 20), (-30.0, 40.0, -1.0, 3, 5, 7)]:
        got = sol.debug_cost_raster(x, y, hd, wm, hm, ppm)
        ref = orc.debug_cost_raster(x, y, hd, wm, hm, ppm)
        assert got.shape == (hm * ppm, wm * ppm)
        assert int(np.sum(got != ref)) <= max(2, got.size // 500), int(np.sum(got != ref))
        assert np.all(got[-1, :] == 0.0) and got[0, 0] == 0.0  # row yi = 0 and flat index 0 are never written
        if (x, y) == (0.0, -10.0):
            assert np.any(got == 1.0) and got.min() == 0.0  # the car marker is inside the window
    with pytest.raises(capi.MppiError):
        sol._ck(sol.L.mppi_debug_cost_raster(sol.h, 0.0, 0.0, 0.0, 10, 10, 50, capi._fp(np.zeros(4, np.float32)), 4))
    sol.close()


def test_savitsky_golay_as_a_call_of_its_own():
    """MPPIController::savitskyGolay() (mppi_controller.cuh:134, mppi_controller.cu:468-499) is public in the
    reference: mppi_savitsky_golay smooths the handle's U_ in place with control_hist_ as left padding -- equal to
    the oracle's filter bit for bit -- and the next solve perturbs the smoothed sequence."""
    cfg = S.make_config(256, 37, track="ring")
    U0 = warm_U(cfg)
    hist = np.array([0.02, 0.2, -0.01, 0.25], np.float32)
    orc = O.Oracle(cfg, fma_mode=1)
    sol = capi.Solver(cfg)
    sol.set_control_seq(U0)
    sol.set_control_hist(hist)
    sol.savitsky_golay()
    U1 = orc.savgol(U0, hist)
    np.testing.assert_array_equal(sol.get_control_seq().view(np.uint32), U1.view(np.uint32))
    assert np.max(np.abs(U1 - U0)) > 1e-5
    eps = noise_for(cfg)
    sol.set_noise(eps)
    sol.compute_control(cfg["start_state"])
    ref = orc.compute_control(cfg["start_state"], U1, hist, eps)
    np.testing.assert_array_equal(sol.get_applied_controls().view(np.uint32), ref["V"][-1].view(np.uint32))
    assert np.max(np.abs(sol.get_results()["U"] - ref["U"])) <= 1e-4
    sol.close()


def test_control_ticks_equals_the_call_by_call_loop():
    """mppi_control_ticks(n, stride) == n x (compute_control + slide_control_seq(stride)), bit for bit."""
    cfg = S.make_config(1024, 50, track="ring")
    a, b = capi.Solver(cfg), capi.Solver(cfg)
    a.seed(99, 0)
    b.seed(99, 0)
    for _ in range(7):
        a.compute_control(cfg["start_state"])
        a.slide_control_seq(2)
    b.control_ticks(cfg["start_state"], 7, 2)
    np.testing.assert_array_equal(a.get_control_seq().view(np.uint32), b.get_control_seq().view(np.uint32))
    np.testing.assert_array_equal(a.get_control_hist().view(np.uint32), b.get_control_hist().view(np.uint32))
    assert a.get_results(False)["traj_cost"] == b.get_results(False)["traj_cost"]
    b.control_ticks(cfg["start_state"], 0, 1)  # nothing
    b.control_ticks(cfg["start_state"], 1, 0)  # solve without slide
    a.compute_control(cfg["start_state"])
    np.testing.assert_array_equal(a.get_control_seq().view(np.uint32), b.get_control_seq().view(np.uint32))
    with pytest.raises(capi.MppiError):
        b.control_ticks(cfg["start_state"], -1, 1)
    a.close(); b.close()


@pytest.mark.parametrize("family,wave", [("nn", 1), ("nn", 2), ("nn", 3), ("nn", 4), ("nn64", 2),
                                         ("row", 1), ("row", 2), ("row", 3), ("row", 4), ("row", 5), ("row", 6), ("row", 7), ("row", 8), ("row_tree", 1), ("row_tree", 4), ("row_tree", 6), ("bf", 1), ("bf", 2), ("bf", 3), ("bf2", 1), ("bf2", 2),
                                         ("multi4", 1), ("multi4", 4), ("multi4", 5), ("multi4", 6), ("multi4", 7), ("multi4", 8),
                                         ("multi2", 2), ("multi2", 3), ("multi2", 4),
                                         ("oct", 1), ("oct", 2), ("oct", 3), ("oct", 4), ("oct", 5), ("oct", 6), ("oct", 7),
                                         ("oct", 8)])
def test_a_starved_wavefront_of_any_role_fails_the_solve_loudly(golden_dir, family, wave):
    """The multi-wavefront rollout kernels hand data over through LDS sequence words; a wave whose wait runs
    out of its poll budget carries on with whatever the LDS holds.  Whichever role that is -- a dynamics
    wave, the cost wave, the control wave -- the workgroup's fail word must poison the costs, so that the
    solve returns MPPI_ERR_HIP instead of finite, wrong controls (round 1 poisoned from the cost wave only).
    The hook starts one role with an exhausted poll budget: it never waits for its partners."""
    import os
    extra = {}
    if family.startswith("bf"):
        extra["bf_W"] = P.load_bf_npz(os.path.join(golden_dir, "models", "basis_function_09_12_2018.npz"))
    elif family == "nn64" or family.startswith("oct"):
        l, th = P.synthetic_model([6, 64, 64, 4], seed=4)
        extra = dict(layers=l, theta=th)
    cfg = S.make_config(256, 40, track="oval", **extra)
    sol = capi.Solver(cfg)
    if family == "bf2":
        sol.set_rollout_variant("quad")  # dynamics + cost wave; "bf": + control wave (the automatic choice)
    elif family != "bf":  # multi form: roles 1..ND = dynamics waves, then [pose wave (ND = 4),] cost wave, control wave[, fetch wave (ND = 4)];
        # oct form: 1..4 dynamics, 5 pose, 6 cost, 7 control, 8 noise wave
        sol.set_rollout_variant(family if family.startswith(("multi", "oct", "row")) else "quad")
    sol.compute_control(cfg["start_state"])          # healthy
    good = sol.get_results()
    assert np.all(np.isfinite(good["costs"])) and np.all(np.isfinite(good["U"]))
    sol.debug_inject_handover_fault(wave, 32)
    with pytest.raises(capi.MppiError) as e:
        sol.compute_control(cfg["start_state"])
    assert e.value.status == capi.ERR_HIP and "hand-over" in str(e.value)
    costs = sol.rollout_only(cfg["start_state"])     # the kernel alone: every workgroup poisoned its costs
    assert np.all(np.isnan(costs))
    sol.debug_inject_handover_fault(0, 0)            # back to normal: the handle keeps working
    sol.reset_controls()
    sol.seed(cfg.get("seed", 1234), 0)
    sol.compute_control(cfg["start_state"])
    again = sol.get_results()
    np.testing.assert_array_equal(again["costs"].view(np.uint32), good["costs"].view(np.uint32))
    np.testing.assert_array_equal(again["U"].view(np.uint32), good["U"].view(np.uint32))
    sol.close()


@pytest.mark.parametrize("K,T,variant,opt,layers", [(4096, 100, "auto", 1, None), (1920, 100, "row_exact", 1, None), (256, 40, "auto", 2, None),
                                                    (8192, 60, "auto", 1, None), (1920, 50, "auto", 1, [6, 64, 64, 64, 64, 4]), (512, 30, "auto", 1, [6, 64, 64, 4]),
                                                    # the generator-kernel form (K T >= 2^20: its draws are prefetched on a second stream, beside the rollout),
                                                    # 32- and 64-wide nets; at 65 536 rollouts (generator behind the rollout) the ticks are not chained
                                                    (16384, 64, "auto", 1, None), (16384, 70, "auto", 1, [6, 64, 64, 4]), (65536, 20, "auto", 1, None),
                                                    (24576, 50, "auto", 1, None), (32768, 40, "auto", 1, [6, 64, 64, 4])])
def test_chained_control_ticks_equal_the_unchained_loop_bit_for_bit(K, T, variant, opt, layers):
    """mppi_control_ticks on one handle in the row form enqueues every solve but the first one tick AHEAD, gated on a word
    the host writes once it holds the previous result (csrc/abi_solve.hip).  Same bits as launching every solve when its
    turn comes, and as a loop of compute_control + slide_control_seq; the handle's device state afterwards is the same too
    (the next ordinary solve agrees)."""
    cfg = S.make_config(K, T, track="oval", opt_stride=opt, layers=layers)
    st = cfg["start_state"]
    sols = [capi.Solver(cfg) for _ in range(3)]
    for sol in sols:
        sol.set_rollout_variant(variant)
        sol.seed(77, 0)
    assert ("multi4_tree_gen" if K >= 16384 else "row8w" if layers is None else "m44_split") in sols[0].rollout_variant()
    sols[1].debug_set_chained_ticks(0)
    n = 23
    sols[0].control_ticks(st, n, opt)   # chained
    sols[1].control_ticks(st, n, opt)   # every solve launched when its turn comes
    for _ in range(n):                  # one ABI call per step
        sols[2].compute_control(st)
        sols[2].slide_control_seq(opt)
    for other in sols[1:]:
        np.testing.assert_array_equal(sols[0].get_control_seq().view(np.uint32), other.get_control_seq().view(np.uint32))
        np.testing.assert_array_equal(sols[0].get_control_hist().view(np.uint32), other.get_control_hist().view(np.uint32))
    res = []
    for sol in sols:  # the device copies (U, generator states) are where the unchained loop leaves them
        sol.compute_control(st)
        res.append(sol.get_results())
    for r in res[1:]:
        for key in ("U", "costs", "w"):
            np.testing.assert_array_equal(res[0][key].view(np.uint32), r[key].view(np.uint32), err_msg=key)
        assert res[0]["traj_cost"] == r["traj_cost"]
    assert np.all(np.isfinite(res[0]["U"]))
    # a second chained call on the same handle, after an ordinary solve in between
    sols[0].slide_control_seq(opt); sols[1].slide_control_seq(opt)
    sols[0].control_ticks(st, 5, opt); sols[1].control_ticks(st, 5, opt)
    np.testing.assert_array_equal(sols[0].get_control_seq().view(np.uint32), sols[1].get_control_seq().view(np.uint32))
    for sol in sols:
        sol.close()


@pytest.mark.parametrize("block", range(3))
def test_random_call_sequences_with_chained_ticks_equal_unchained_ones(block):
    """Stateful: the same random sequence of ABI calls on two handles -- runs of ticks (mppi_control_ticks: chained on one handle,
    switched off on the other), single solves (blocking / asynchronous / explicit noise), slides by several strides, sequence /
    history / limits / cost updates, resets, kernel-form switches, rollout_only, re-seeding -- with every read-back compared
    bit for bit.  The chain keeps state of its own on the handle (the solve ahead, which copy of U the device holds, which
    generator buffer is whose): whatever is called around it must find the handle where the unchained loop leaves it."""
    for sq in range(block * 6, block * 6 + 6):
        rng = np.random.RandomState(515100 + sq)
        K = int(rng.choice([64, 512, 1920, 4096]))
        T = int(rng.choice([5, 16, 33, 60]))
        layers = [None, None, [6, 64, 64, 4], [6, 32, 32, 32, 32, 4]][rng.randint(4)]
        iters = int(rng.choice([1, 1, 1, 2]))
        opt = min(int(rng.choice([1, 1, 2])), T - 1)
        cfg = S.make_config(K, T, layers=layers, track=str(rng.choice(["ring", "oval"])), num_iters=iters, opt_stride=opt)
        a, b = capi.Solver(cfg), capi.Solver(cfg)
        b.debug_set_chained_ticks(0)
        seed = int(rng.randint(1, 1 << 30))
        for s_ in (a, b):
            s_.seed(seed, 0)
        state = cfg["start_state"].copy()
        log = []

        def same(what):
            tag = (sq, K, T, cfg["layers"], iters, opt, " ".join(log), what)
            np.testing.assert_array_equal(a.get_control_seq().view(np.uint32), b.get_control_seq().view(np.uint32), err_msg=str(tag))
            np.testing.assert_array_equal(a.get_control_hist().view(np.uint32), b.get_control_hist().view(np.uint32), err_msg=str(tag))

        for step in range(16):
            op = str(rng.choice(["ticks", "ticks", "ticks", "solve", "async", "explicit", "slide", "setU", "sethist", "reset", "cost",
                                 "limits", "variant", "rollout_only", "seed"]))
            log.append(op)
            if op == "ticks":
                n, stride = int(rng.choice([1, 2, 3, 7])), int(rng.choice([opt, opt, opt, 1, 0]))
                log[-1] = "ticks(%d,%d)" % (n, stride)
                for s_ in (a, b):
                    s_.control_ticks(state, n, stride)
                same("ticks")
            elif op in ("solve", "async", "explicit"):
                if op == "explicit":
                    eps = np.stack([O.generate_noise(999, 2 * T * i + 7 * step * T, K, T) for i in range(iters)])
                for s_ in (a, b):
                    if op == "explicit":
                        s_.set_noise(eps)
                    if op == "async":
                        s_.compute_control_async(state)
                        s_.synchronize()
                    else:
                        s_.compute_control(state)
                ra, rb = a.get_results(), b.get_results()
                for key in ("U", "costs", "w"):
                    np.testing.assert_array_equal(ra[key].view(np.uint32), rb[key].view(np.uint32), err_msg=str((sq, " ".join(log), key)))
                assert ra["traj_cost"] == rb["traj_cost"]
                gs, _ = a.nominal_traj(state)
                state = gs[min(opt, T - 1)].copy()
            elif op == "slide":
                st = int(rng.choice([opt, opt, 1, 2, T]))
                for s_ in (a, b):
                    s_.slide_control_seq(st)
                same("slide")
            elif op == "setU":
                U = warm_U(cfg, seed=step + sq)
                for s_ in (a, b):
                    s_.set_control_seq(U)
            elif op == "sethist":
                hist = rng.uniform(-0.3, 0.3, 4).astype(np.float32)
                for s_ in (a, b):
                    s_.set_control_hist(hist)
            elif op == "reset":
                for s_ in (a, b):
                    s_.reset_controls()
            elif op == "cost":
                cost = dict(cfg["cost"], desired_speed=float(rng.choice([4.0, 8.0, 12.0])), track_coeff=float(rng.choice([100.0, 200.0])),
                            steering_coeff=float(rng.choice([0.0, 0.5])), l1_cost=bool(rng.rand() < 0.3))
                cfg = dict(cfg, cost=cost)
                for s_ in (a, b):
                    s_.set_cost_params(cost)
            elif op == "limits":
                lo, hi = (-0.8, -0.5), (0.9, float(rng.choice([0.3, 0.65])))
                for s_ in (a, b):
                    s_.set_control_limits(lo, hi)
            elif op == "variant":
                v = str(rng.choice(["auto", "auto", "row", "quad", "fused", "m44", "mfma"]))
                for s_ in (a, b):
                    try:
                        s_.set_rollout_variant(v)
                    except capi.MppiError:
                        pass
                assert a.rollout_variant() == b.rollout_variant()
            elif op == "rollout_only":
                ca, cb = a.rollout_only(state), b.rollout_only(state)
                np.testing.assert_array_equal(ca.view(np.uint32), cb.view(np.uint32), err_msg=str((sq, " ".join(log))))
            elif op == "seed":
                seed += 1
                for s_ in (a, b):
                    s_.seed(seed, 0)
        same("end")
        a.close(); b.close()


@pytest.mark.parametrize("K,T", [(8192, 100), (16384, 60)])  # the row form with its own noise wave / the generator-kernel form
def test_an_error_inside_chained_ticks_calls_the_solve_ahead_off(K, T):
    """A solve of a chain fails on the host side (here: its wait runs out of time) while the NEXT solve is already enqueued
    behind it, gated.  The gate must be opened with the cancel bit (a gated kernel left waiting would hold the queue for its
    100 ms deadline and then poison a solve nobody asked for), the call returns the error, and the handle -- whose device
    copies are stale by then -- works again from the host's copies: the next results are those of a fresh handle fed the same
    control sequence."""
    import time
    cfg = S.make_config(K, T, track="oval")
    st = cfg["start_state"]
    sol = capi.Solver(cfg)
    sol.seed(5, 0)
    sol.control_ticks(st, 4, 1)
    sol.set_wait_timeout(1e-6)
    t0 = time.perf_counter()
    with pytest.raises(capi.MppiError) as e:
        sol.control_ticks(st, 6, 1)
    assert e.value.status == capi.ERR_HIP and "timed out" in str(e.value)
    assert time.perf_counter() - t0 < 0.05  # the solve ahead was called off at once, not left to its deadline
    sol.set_wait_timeout(30.0)
    U, hist = sol.get_control_seq().copy(), sol.get_control_hist().copy()
    deadline = time.perf_counter() + 5.0
    while True:
        try:
            sol.seed(9, 0)
            break
        except capi.MppiError:
            assert time.perf_counter() < deadline
            time.sleep(0.001)
    sol.control_ticks(st, 3, 1)
    ref = capi.Solver(cfg)
    ref.set_control_seq(U)
    ref.set_control_hist(hist)
    ref.seed(9, 0)
    ref.control_ticks(st, 3, 1)
    np.testing.assert_array_equal(sol.get_control_seq().view(np.uint32), ref.get_control_seq().view(np.uint32))
    np.testing.assert_array_equal(sol.get_control_hist().view(np.uint32), ref.get_control_hist().view(np.uint32))
    sol.close(); ref.close()


def test_wait_timeout_is_kept_and_the_lost_solve_is_not_waited_for_again():
    """mppi_set_wait_timeout: a limit far below the length of a solve (K = 65536, T = 150, 6-64-64-4: about a millisecond
    of device work) ends the blocking call in MPPI_ERR_HIP within about the limit; the calls that follow return at once
    while the lost solve's work is still on the device (they do not wait the limit out again), and once that work has
    drained the handle solves again, from the control sequence the host held -- bit for bit what a fresh handle computes."""
    import time
    cfg = S.make_config(65536, 150, track="oval", layers=[6, 64, 64, 4])  # synthetic weights (config 4's)
    ref = capi.Solver(cfg)
    ref.compute_control(cfg["start_state"])
    want = ref.get_results(with_vectors=False)
    ref.close()
    sol = capi.Solver(cfg)
    sol.set_wait_timeout(100e-6)
    t0 = time.perf_counter()
    with pytest.raises(capi.MppiError) as e:
        sol.compute_control(cfg["start_state"])
    first = time.perf_counter() - t0
    assert e.value.status == capi.ERR_HIP and "timed out" in str(e.value)
    assert first < 0.05, first  # the limit (0.1 ms) plus the launches, not the solve's length times anything
    t1 = time.perf_counter()
    n_refused = 0
    for _ in range(3):
        try:
            sol.get_results(with_vectors=False)
        except capi.MppiError as e2:
            assert e2.status == capi.ERR_HIP
            n_refused += 1
    assert n_refused == 3 and time.perf_counter() - t1 < 0.02  # no second wait: the result getters refuse at once
    sol.set_wait_timeout(30.0)
    deadline = time.perf_counter() + 5.0
    while True:  # the next solve is refused (at once) while the lost one still runs, accepted afterwards
        try:
            sol.seed(cfg.get("seed", 1234), 0)
            sol.compute_control(cfg["start_state"])
            break
        except capi.MppiError as e3:
            assert e3.status == capi.ERR_HIP and "timed out" in str(e3), str(e3)
            assert time.perf_counter() < deadline, "the lost solve's device work never drained"
            time.sleep(0.001)
    got = sol.get_results(with_vectors=False)
    np.testing.assert_array_equal(got["U"].view(np.uint32), want["U"].view(np.uint32))
    sol.close()


@pytest.mark.parametrize("K", [6400, 12288 + 64, 20480 + 64])  # two chunks on the row form (beta taken in the tail: from all costs), four with
# beta out of the rollout kernel, six (without it the weights workgroups would exchange chunk minima)
@pytest.mark.parametrize("role", [32, 33, 34])
def test_stream_tail_wait_that_runs_out_of_time_is_an_error(role, K):
    """K > 4096: the tail stage is ONE launch whose workgroups hand chunk sums and chain results (and, where the rollout kernel
    left no beta, chunk minima and beta) over as {value, epoch} granules (solve_kernels.hip: solve_tail_stream_kernel).  A
    workgroup that never publishes -- chunk 0's sum of weights (32), chunk 0's chain results (33), every chunk sum (34) -- must end the solve in MPPI_ERR_HIP within the
    deadline, never in a hang and never in finite controls; the handle keeps working afterwards, bit for bit."""
    cfg = S.make_config(K, 30, track="oval")  # two / four / six chunks, the last one ragged
    sol = capi.Solver(cfg)
    sol.compute_control(cfg["start_state"])
    good = sol.get_results()
    assert np.all(np.isfinite(good["U"])) and np.isfinite(good["traj_cost"])
    sol.debug_inject_handover_fault(role, 200)  # every wait gives up after 200 us
    with pytest.raises(capi.MppiError) as e:
        sol.compute_control(cfg["start_state"])
    assert e.value.status == capi.ERR_HIP and "hand-over" in str(e.value)
    sol.debug_inject_handover_fault(0, 0)
    sol.reset_controls()
    sol.seed(cfg.get("seed", 1234), 0)
    sol.compute_control(cfg["start_state"])
    again = sol.get_results()
    np.testing.assert_array_equal(again["U"].view(np.uint32), good["U"].view(np.uint32))
    assert again["traj_cost"] == good["traj_cost"]
    np.testing.assert_array_equal(again["w"].view(np.uint32), good["w"].view(np.uint32))
    sol.close()


def test_slide_by_zero_is_a_no_op_and_hist_waits_for_a_pending_solve():
    """stride 0 (a tick without a new pose, run_control_loop.cuh:208-216) changes nothing; set_control_hist
    during an asynchronous solve waits for it, so that solve is still smoothed with the history it ran with."""
    cfg = S.make_config(512, 30, track="ring")
    sol = capi.Solver(cfg)
    U0 = warm_U(cfg)
    hist = np.array([0.01, 0.2, -0.02, 0.25], np.float32)
    sol.set_control_seq(U0)
    sol.set_control_hist(hist)
    sol.slide_control_seq(0)
    np.testing.assert_array_equal(sol.get_control_seq(), U0)
    np.testing.assert_array_equal(sol.get_control_hist(), hist)
    eps = noise_for(cfg)
    sol.set_noise(eps)
    sol.compute_control_async(cfg["start_state"])
    sol.set_control_hist(np.zeros(4, np.float32))  # must not leak into the pending solve's smoothing
    got = sol.get_results(with_vectors=False)
    ref = O.Oracle(cfg, fma_mode=1).compute_control(cfg["start_state"], U0, hist, eps)
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4
    sol.close()


def test_prefetched_generator_draws_are_the_right_ones():
    """Handles with K T >= 2^20 draw eps with the stand-alone generator on a stream of its own and request the
    NEXT solve's draws while the current solve's tail runs.  The sequence of draws must be what it always was:
    solve i of a handle consumes draws [2T i, 2T (i+1)) of every rollout's subsequence -- also across
    mppi_generate_noise (hands out the prefetched draws), mppi_seed (discards them), an explicit-noise solve in
    between (consumes none) and a switch to a kernel form with its own noise wavefront."""
    K, T = 16384, 64
    assert K * T >= 1 << 20
    cfg = S.make_config(K, T, track="oval")
    gen = capi.Solver(dict(cfg, seed=77))
    ref = capi.Solver(dict(cfg, seed=77))
    assert "multi4" in gen.rollout_variant() and gen.rollout_variant().endswith("_gen")
    x = cfg["start_state"]

    def both(offset, label):
        gen.compute_control(x)
        ref.set_noise(O.generate_noise(77, offset, K, T)[None])
        ref.compute_control(x)
        a, b = gen.get_control_seq(), ref.get_control_seq()
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=label)
        gen.slide_control_seq(1)
        ref.slide_control_seq(1)

    both(0, "first solve (generated now)")
    both(2 * T, "second solve (prefetched)")
    both(4 * T, "third solve (prefetched)")
    # the API's own view of the stream: the next draws are the prefetched ones
    e = gen.generate_noise()
    np.testing.assert_array_equal(e.view(np.uint32), O.generate_noise(77, 6 * T, K, T).view(np.uint32))
    both(8 * T, "after mppi_generate_noise")
    # an explicit-noise solve consumes no draws
    eps = O.generate_noise(5, 0, K, T)[None]
    for s in (gen, ref):
        s.set_noise(eps)
        s.compute_control(x)
        s.slide_control_seq(1)
    np.testing.assert_array_equal(gen.get_control_seq().view(np.uint32), ref.get_control_seq().view(np.uint32))
    both(10 * T, "after an explicit-noise solve")
    # a form with its own noise wavefront takes over the prefetched draws, then continues the stream itself
    # (the reference handle follows: the quad form keeps the reference's summation order, the automatic form at this K sums
    # the output layer as a butterfly -- equal draws, not equal bits, between the two)
    for s in (gen, ref):
        s.set_rollout_variant("quad")
    both(12 * T, "quad form, prefetched draws")
    both(14 * T, "quad form, in-kernel generator")
    for s in (gen, ref):
        s.set_rollout_variant("auto")
    both(16 * T, "back to the generator kernel")
    # re-seeding discards what was prefetched
    gen.seed(123, 10)
    gen.compute_control(x)
    ref.set_noise(O.generate_noise(123, 10, K, T)[None])
    ref.compute_control(x)
    np.testing.assert_array_equal(gen.get_control_seq().view(np.uint32), ref.get_control_seq().view(np.uint32))
    gen.close(); ref.close()


def test_large_handle_two_iterations_generator_mode():
    """K T >= 2^20 with num_iters = 2: the second iteration's draws cannot be prefetched (they are generated on
    the generator stream between the two iterations, ordered by events against the first iteration's tail kernel
    reading the other buffer).  Generator mode must equal explicit noise fed with the same draws, twice in a row."""
    K, T = 16384, 64
    cfg = S.make_config(K, T, track="oval", num_iters=2)
    gen = capi.Solver(dict(cfg, seed=31))
    ref = capi.Solver(dict(cfg, seed=31))
    x = cfg["start_state"]
    for solve in range(2):
        gen.compute_control(x)
        eps = np.stack([O.generate_noise(31, 2 * T * (2 * solve + it), K, T) for it in range(2)])
        ref.set_noise(eps)
        ref.compute_control(x)
        np.testing.assert_array_equal(gen.get_control_seq().view(np.uint32), ref.get_control_seq().view(np.uint32))
        a, b = gen.get_applied_controls(), ref.get_applied_controls()
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        gen.slide_control_seq(1)
        ref.slide_control_seq(1)
    gen.close(); ref.close()
