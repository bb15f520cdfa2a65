"""Parity of the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (BASELINE.json north_star / SURVEY 8d):
  * applied controls V (the rewritten du_d buffer): bit-exact (only fp32 mul/add of the inputs)
  * NN state derivative vs the reference-derived golden vectors: 1e-5 (fp32 vs fp64)
  * per-rollout costs: rel 1e-4 against max(|cost|, 1) on crash-free maps
  * smoothed control sequence U: L-inf <= 1e-4 ; trajectory cost: rel <= 1e-4
  * generated noise: bit-exact against the oracle's statement of the same spec
"""
import os

import numpy as np
import pytest

from autorally_amd import capi
from autorally_amd import params as P
from autorally_amd import synthetic as S
from oracle import oracle as O
from tests.helpers import load_nn_golden, noise_for, oracle_mode_for, rel_err, warm_U

pytestmark = pytest.mark.gpu

MODELS = ["autorally_nnet_09_12_2018", "gazebo_nnet_09_12_2018", "shallow_network_08_20_2020",
          "wider_deeper_network_08_20_2020", "trained_writer_6_16_24_4"]


@pytest.fixture(scope="module", autouse=True)
def _built():
    from autorally_amd import build as B
    B.build()
    assert capi.lib().mppi_device_count() >= 1, "no gfx950 device: the HIP path cannot run"


def _solve_both(cfg, U0=None, hist=None, seed=1234, variant=None):
    """HIP solve and the oracle on the same inputs.  The oracle runs in the arithmetic mode of the kernel form that served
    the solve: mode 1 (the reference's summation order) for every form but the tree forms, whose output layer is summed as
    a butterfly (tests/helpers.py: oracle_mode_for) -- for those the NOMINAL oracle's U / trajectory cost / costs come along as ref["nominal"] and
    every caller's north-star criteria are checked against them here."""
    eps = noise_for(cfg, seed)
    U0 = np.zeros((cfg["T"], 2), np.float32) if U0 is None else U0
    hist = np.zeros(4, np.float32) if hist is None else hist
    sol = capi.Solver(cfg)
    if variant:
        sol.set_rollout_variant(variant)
    sol.set_control_seq(U0)
    sol.set_control_hist(hist)
    sol.set_noise(eps)
    sol.compute_control(cfg["start_state"])
    got = sol.get_results()
    got["V"] = sol.get_applied_controls()
    got["variant"] = sol.rollout_variant()
    sol.close()
    tree = "_tree" in got["variant"]
    iters = cfg.get("num_iters", 1)
    mode = oracle_mode_for(got["variant"])  # the output layer's summation order of the form that ran
    ref = O.Oracle(cfg, fma_mode=mode, nthreads=8).compute_control(cfg["start_state"], U0, hist, eps, num_iters=iters)
    if tree:
        nom = O.Oracle(cfg, fma_mode=1, nthreads=8).compute_control(cfg["start_state"], U0, hist, eps, num_iters=iters)
        ref["nominal"] = nom
        if iters == 1:  # north star: controls and trajectory cost within 1e-4 of the reference's own summation order
            # the margin at the contract's mark, visible in the log (pytest -s / pytest.log): VERDICT round 4, item 3
            print("nominal margin: K=%d T=%d %s %s: |dU|inf = %.3e, trajectory cost rel = %.3e (bound 1e-4 each), flipped %d of %d" % (
                cfg["K"], cfg["T"], "-".join(map(str, cfg["layers"])), got["variant"], float(np.max(np.abs(got["U"] - nom["U"]))),
                abs(got["traj_cost"] - nom["traj_cost"]) / abs(nom["traj_cost"]), int(np.sum(rel_err(got["costs"], nom["costs"]) > 1e-4)), cfg["K"]))
            assert np.max(np.abs(got["U"] - nom["U"])) <= 1e-4
            assert abs(got["traj_cost"] - nom["traj_cost"]) <= 1e-4 * abs(nom["traj_cost"])
            assert int(np.sum(rel_err(got["costs"], nom["costs"]) > 1e-4)) <= max(cfg["K"] // 200, 1)
    return ref, got


# ---------------------------------------------------------------- dynamics vs golden vectors
@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("variant", ["auto", "valu"])
def test_dynamics_golden_on_gpu(golden_dir, name, variant):
    g = load_nn_golden(golden_dir)
    layers, theta = P.load_model_npz(os.path.join(golden_dir, "models", name + ".npz"))
    cfg = S.make_config(64, 10, layers=layers, theta=theta,
                        negate_yaw_der=bool(g[name + "/negate_yaw_der"][0]))
    sol = capi.Solver(cfg)
    sol.set_rollout_variant(variant)
    ders = sol.debug_dynamics(g[name + "/states"], g[name + "/controls"])
    ref = g[name + "/state_ders"]
    assert np.max(np.abs(ders - ref) / np.maximum(1.0, np.abs(ref))) < 1e-5
    # and against the fp32 oracle: only tanh / sincos differ
    orc = O.Oracle(cfg, fma_mode=1)
    o = np.stack([orc.state_deriv(s, u) for s, u in zip(g[name + "/states"], g[name + "/controls"])])
    assert np.max(np.abs(ders - o) / np.maximum(1.0, np.abs(o))) < 5e-6
    sol.close()


# ---------------------------------------------------------------- noise generator
@pytest.mark.parametrize("K,T", [(128, 50), (2048, 100), (4096, 100), (64, 7)])
def test_noise_generator_bit_exact(K, T):
    cfg = S.make_config(K, T)
    sol = capi.Solver(cfg)
    sol.seed(1234, 0)
    a = sol.generate_noise()
    b = sol.generate_noise()  # continues each rollout's subsequence
    ref = O.generate_noise(1234, 0, K, 2 * T)
    np.testing.assert_array_equal(a.view(np.uint32), ref[:, :T].view(np.uint32))
    np.testing.assert_array_equal(b.view(np.uint32), ref[:, T:].view(np.uint32))
    sol.seed(99, 10)
    c = sol.generate_noise()
    np.testing.assert_array_equal(c.view(np.uint32), O.generate_noise(99, 10, K, T).view(np.uint32))
    sol.close()


# ---------------------------------------------------------------- rollout / solve parity
CASES = [
    # (K, T, track, layers)   -- BASELINE.json configs 1, 2, 3 + the shipped wide net
    (128, 50, "ring", None),
    (2048, 100, "ring", None),
    (4096, 100, "oval", None),
    (256, 60, "ring", [6, 64, 64, 4]),
    (6144, 24, "oval", None),  # rows of the weighted reduction spread over two workgroups (4096 + 2048)
]


@pytest.mark.parametrize("K,T,track,layers", CASES)
@pytest.mark.parametrize("variant", ["auto", "valu"])
def test_rollout_costs_and_controls(K, T, track, layers, variant):
    cfg = S.make_config(K, T, layers=layers, track=track)
    U0 = warm_U(cfg)
    ref, got = _solve_both(cfg, U0=U0, variant=variant)
    # applied controls are plain fp32 arithmetic on identical inputs: bit-exact
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    err = rel_err(got["costs"], ref["costs"])
    bad = int(np.sum(err > 1e-4))
    # Threshold chaos: the nearest-texel lookup and the crash/slip thresholds are
    # discontinuous, so a 1-ulp tanh/sincos difference moves a few grazing rollouts to the
    # neighbouring texel (the oracle's own FMA / no-FMA builds disagree on the same rollouts).
    # Such rollouts must be few and the weight they move must be negligible.
    assert bad <= K // 200, (bad, float(err.max()))
    assert float(np.percentile(err, 99)) < 5e-6  # away from the discontinuities the agreement is tight
    wsum = float(ref["w"].sum())
    assert float(np.abs(got["w"] - ref["w"]).sum()) / wsum < 1e-4
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"])
    # downstream stages alone: oracle weighting + reduction + smoothing fed with the GPU's own
    # costs and applied controls must reproduce the GPU's U (no chaos left in this comparison)
    orc = O.Oracle(cfg, fma_mode=1)
    w, _, eta, tc = orc.weights(got["costs"])
    U2 = orc.savgol(orc.weighted_reduction(w, eta, got["V"]), np.zeros(4, np.float32))
    assert np.max(np.abs(U2 - got["U"])) <= 2e-6
    assert abs(tc - got["traj_cost"]) <= 1e-5 * abs(tc)


@pytest.mark.parametrize("instance", range(8))
def test_config5_eight_independent_instances(instance):
    """BASELINE.json configs[4]: eight MPPI instances (distinct start states + costmaps: the oval rotated
    by 0.35 rad and shifted per instance, each inside its own recentred map), K=4096 T=100 each.  On the
    8-GPU node they run one per GPU with no collective (bench.py --gpus 8 uses exactly
    make_config(..., instance=rank, seed=1234+rank)); here they run one after another on one device, each
    against the oracle with the criteria of test_rollout_costs_and_controls."""
    cfg = S.make_config(4096, 100, track="oval", instance=instance, seed=1234 + instance)
    m = cfg["map_rgba"][:, :, 0]
    on = m < 1.0  # the drivable band must lie inside the map, away from its border
    assert not (on[0].any() or on[-1].any() or on[:, 0].any() or on[:, -1].any())
    U0 = warm_U(cfg, seed=7 + instance)
    ref, got = _solve_both(cfg, U0=U0, seed=1234 + instance)
    assert "row8w" in got["variant"]  # 6-32-32-4 at one group per CU: the vector-ALU row form
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    err = rel_err(got["costs"], ref["costs"])
    assert int(np.sum(err > 1e-4)) <= cfg["K"] // 200, (instance, float(err.max()))
    assert float(np.percentile(err, 99)) < 5e-6
    assert float(np.abs(got["w"] - ref["w"]).sum()) / float(ref["w"].sum()) < 1e-4
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"])
    # non-degenerate: the start pose is on the centre-line (most rollouts stay on the track, weights spread)
    assert float(np.mean(ref["costs"] < 5000.0)) > 0.5 and float(ref["w"].sum()) > 4.0


@pytest.mark.parametrize("layers,variant", [([6, 16, 8, 4], "auto"), ([6, 24, 4], "auto"), (None, "valu_lds"),
                                            ([6, 64, 64, 64, 64, 4], "valu")])
def test_generic_and_register_valu_kernels(layers, variant):
    """Shapes outside 6-HxN-4 (H in {32,64}) run the generic LDS kernel; 'valu_lds' forces it on a
    standard shape; 'valu' on a standard shape is the register / LDS-broadcast kernel."""
    cfg = S.make_config(256, 40, layers=layers, track="ring")
    ref, got = _solve_both(cfg, U0=warm_U(cfg), variant=variant)
    assert got["variant"].startswith("valu")
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    assert float(np.percentile(rel_err(got["costs"], ref["costs"]), 99)) < 5e-6
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4


def test_mfma_and_valu_variants_agree_bitwise():
    """Both kernels use the same k-ascending fmaf chain; the f32 MFMA is documented to be exactly
    that chain, so per-rollout costs must be IDENTICAL between the two arms."""
    cfg = S.make_config(512, 40, track="oval")
    U0 = warm_U(cfg)
    _, a = _solve_both(cfg, U0=U0, variant="mfma")
    _, b = _solve_both(cfg, U0=U0, variant="valu")
    _, c = _solve_both(cfg, U0=U0, variant="valu_lds")
    assert a["variant"].startswith("mfma") and b["variant"] == "valu_reg_lds" and c["variant"] == "valu_lds"
    for o in (b, c):
        np.testing.assert_array_equal(a["costs"].view(np.uint32), o["costs"].view(np.uint32))
        np.testing.assert_array_equal(a["U"].view(np.uint32), o["U"].view(np.uint32))


@pytest.mark.parametrize("K,T,track,layers", [(512, 43, "oval", None), (256, 30, "ring", [6, 64, 64, 4]),
                                              (192, 37, "oval", [6, 32, 32, 32, 32, 4])])
def test_quad_and_fused_mfma_kernels_agree_bitwise(K, T, track, layers):
    """The four-wave form (network split over two wavefronts, cost wave, control wave) does the same
    arithmetic in the same order as the single-wave form; T is not a multiple of the control wave's
    chunk nor of the LDS rings' depth on purpose.  Explicit noise here; generator mode below."""
    kw = {}
    if layers:
        l, th = P.synthetic_model(layers, seed=4)
        kw = dict(layers=l, theta=th)
    cfg = S.make_config(K, T, track=track, **kw)
    U0 = warm_U(cfg)
    _, a = _solve_both(cfg, U0=U0, variant="fused")
    _, c = _solve_both(cfg, U0=U0, variant="quad")
    assert "fused" in a["variant"] and "quad" in c["variant"]
    np.testing.assert_array_equal(a["costs"].view(np.uint32), c["costs"].view(np.uint32))
    np.testing.assert_array_equal(a["V"].view(np.uint32), c["V"].view(np.uint32))
    np.testing.assert_array_equal(a["U"].view(np.uint32), c["U"].view(np.uint32))
    # the multi form (ND dynamics waves + one cost wave + one control wave per 16 ND rollouts), ND = 4, 2, 1
    if layers is None:  # the row form exists for 6-32-32-4
        _, m = _solve_both(cfg, U0=U0, variant="row")
        assert "row8w" in m["variant"]
        np.testing.assert_array_equal(a["costs"].view(np.uint32), m["costs"].view(np.uint32))
        np.testing.assert_array_equal(a["V"].view(np.uint32), m["V"].view(np.uint32))
        np.testing.assert_array_equal(a["U"].view(np.uint32), m["U"].view(np.uint32))
    for v in ("multi4", "multi2", "multi4_gen"):
        _, m = _solve_both(cfg, U0=U0, variant=v)
        assert v in m["variant"]
        np.testing.assert_array_equal(a["costs"].view(np.uint32), m["costs"].view(np.uint32))
        np.testing.assert_array_equal(a["V"].view(np.uint32), m["V"].view(np.uint32))
        np.testing.assert_array_equal(a["U"].view(np.uint32), m["U"].view(np.uint32))


@pytest.mark.parametrize("K,T,layers", [(256, 30, [6, 64, 64, 4]), (64, 2, [6, 64, 64, 4]), (128, 3, [6, 64, 64, 4]),
                                        (192, 37, [6, 64, 64, 64, 64, 4]), (64, 17, [6, 64, 64, 64, 64, 4])])
def test_oct_kernel_agrees_bitwise_with_the_single_wave_form(K, T, layers):
    """The eight-wave form for 64-wide nets (four dynamics waves, one M tile of every layer each, all-to-all
    swap per layer; riders: pose -> cost and noise -> control) keeps every dot product's k-ascending order:
    costs, applied controls and the new control sequence equal the single-wave form's bit for bit -- also for
    horizons shorter than the rings, T not a multiple of the control wave's chunk, and in generator mode."""
    l, th = P.synthetic_model(layers, seed=4)
    cfg = S.make_config(K, T, track="oval", layers=l, theta=th)
    U0 = warm_U(cfg)
    _, a = _solve_both(cfg, U0=U0, variant="fused")
    for v in ("oct",):
        _, m = _solve_both(cfg, U0=U0, variant=v)
        assert "oct8w" in m["variant"]
        np.testing.assert_array_equal(a["costs"].view(np.uint32), m["costs"].view(np.uint32))
        np.testing.assert_array_equal(a["V"].view(np.uint32), m["V"].view(np.uint32))
        np.testing.assert_array_equal(a["U"].view(np.uint32), m["U"].view(np.uint32))
    # generator mode: the noise wave draws in the kernel, like the quad form's control wave; "_gen": eps from the
    # stand-alone generator kernel.  Same streams, same sequence of solves.
    names = ("quad", "oct", "oct_gen", "fused")
    sols = [capi.Solver(cfg) for _ in names]
    for sol, v in zip(sols, names):
        sol.set_rollout_variant(v)
    for it in range(3):
        for sol in sols:
            sol.compute_control(cfg["start_state"])
        r0 = sols[0].get_results()
        for sol, v in zip(sols[1:], names[1:]):
            r1 = sol.get_results()
            assert v.replace("oct", "oct8w") in sol.rollout_variant()
            np.testing.assert_array_equal(r0["costs"].view(np.uint32), r1["costs"].view(np.uint32))
            np.testing.assert_array_equal(r0["U"].view(np.uint32), r1["U"].view(np.uint32))
    for sol in sols:
        sol.close()


def test_oct_form_is_refused_for_narrow_nets():
    cfg = S.make_config(64, 10, track="oval")
    sol = capi.Solver(cfg)
    with pytest.raises(capi.MppiError) as e:
        sol.set_rollout_variant("oct")
    assert e.value.status == capi.ERR_UNSUPPORTED
    sol.close()


def test_cold_start_zero_controls():
    cfg = S.make_config(1024, 100, track="oval")
    ref, got = _solve_both(cfg)
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"])


def test_two_iterations():
    cfg = S.make_config(512, 50, track="ring", num_iters=2)
    ref, got = _solve_both(cfg, U0=warm_U(cfg))
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"])
    # The second iteration's applied controls are (raw weighted mean of iteration 1) + nu * eps: not
    # bit-exact (the mean carries the ulp-level cost differences of iteration 1), but bounded like the
    # fuzz test bounds them for iters > 1: 2e-4 + 4 x the weight mass of the rollouts whose cost flipped.
    err = rel_err(got["costs"], ref["costs"])
    flipped = err > 1e-4
    w, wg = ref["w"] / ref["w"].sum(), got["w"] / got["w"].sum()
    mass = float(np.sum(np.maximum(w, wg)[flipped]))
    assert float(np.max(np.abs(got["V"] - ref["V"][-1]))) <= 2e-4 + 8.0 * mass
    # and the noise-free rollout 0 applies exactly the mean both sides fed back
    assert float(np.max(np.abs(got["V"][0] - ref["V"][-1][0]))) <= 1e-4


def test_cost_branches_l1_control_cost_and_opt_stride():
    cost = dict(P.DEFAULT_COST, l1_cost=True, steering_coeff=0.5, throttle_coeff=0.25, track_slop=0.05)
    cfg = S.make_config(256, 40, track="ring", cost=cost, opt_stride=3)
    ref, got = _solve_both(cfg, U0=warm_U(cfg))
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    assert np.all(rel_err(got["costs"], ref["costs"]) < 1e-4)
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4


def test_crash_and_clamp_paths():
    """Start off-track at high speed with aggressive controls: exercises the sticky crash flag,
    the slip-angle kill, the 1e12 cap is not reached, control clamping and pure-noise rollouts."""
    cfg = S.make_config(512, 60, track="oval", nu=(0.9, 0.9))
    cfg["start_state"] = np.array([0.0, -12.4, 0.3, 0.0, 9.0, 0.5, 0.2], np.float32)
    ref, got = _solve_both(cfg, U0=warm_U(cfg))
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    err = rel_err(got["costs"], ref["costs"])
    assert np.mean(err > 1e-4) < 0.02
    assert ref["costs"].max() > 5000.0  # crash costs were really exercised


def test_warm_start_loop_with_slide():
    """Five consecutive solves with slideControlSeq between them (the control-loop usage)."""
    cfg = S.make_config(1024, 60, track="ring")
    orc = O.Oracle(cfg, fma_mode=1, nthreads=8)
    sol = capi.Solver(cfg)
    U = np.zeros((cfg["T"], 2), np.float32)
    hist = np.zeros(4, np.float32)
    state = cfg["start_state"].copy()
    for it in range(5):
        eps = O.generate_noise(1234, 2 * cfg["T"] * it, cfg["K"], cfg["T"])[None]
        ref = orc.compute_control(state, U, hist, eps)
        sol.set_noise(eps)
        sol.compute_control(state)
        got = sol.get_results(with_vectors=False)
        assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4, it
        # both sides continue from the ORACLE's sequence so errors cannot accumulate silently
        ss, cs = orc.nominal_traj(state, ref["U"])
        gs, gc = sol.nominal_traj(state)
        np.testing.assert_allclose(gs, orc.nominal_traj(state, got["U"])[0], atol=1e-5, rtol=1e-5)
        state = ss[1].copy()
        U, hist = orc.slide_control_seq(ref["U"], hist, cfg["init_u"], 1)
        sol.set_control_seq(ref["U"])
        sol.slide_control_seq(1)
        np.testing.assert_array_equal(sol.get_control_seq(), U)
        np.testing.assert_array_equal(sol.get_control_hist(), hist)
    sol.close()


@pytest.mark.parametrize("stride", [1, 2])
def test_device_resident_loop_matches_oracle(stride):
    """solve -> slideControlSeq -> solve ... without ever touching U from the host: the control
    sequence and control history stay on the device (the tail kernel leaves the slid copy), the
    noise comes from the in-kernel generator.  The oracle runs the same loop on its own state."""
    cfg = S.make_config(1024, 60, track="ring", opt_stride=stride)
    orc = O.Oracle(cfg, fma_mode=1, nthreads=8)
    sol = capi.Solver(cfg)
    sol.seed(1234, 0)
    U = np.zeros((cfg["T"], 2), np.float32)
    hist = np.zeros(4, np.float32)
    state = cfg["start_state"].copy()
    for it in range(5):
        eps = O.generate_noise(1234, 2 * cfg["T"] * it, cfg["K"], cfg["T"])[None]
        ref = orc.compute_control(state, U, hist, eps)
        sol.compute_control(state)
        got = sol.get_results(with_vectors=False)
        assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4, it
        assert abs(got["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"]), it
        ss, _ = orc.nominal_traj(state, ref["U"])
        state = ss[stride].copy()
        U, hist = orc.slide_control_seq(ref["U"], hist, cfg["init_u"], stride)
        sol.slide_control_seq(stride)
        np.testing.assert_allclose(sol.get_control_seq(), U, atol=1e-4)
        np.testing.assert_allclose(sol.get_control_hist(), hist, atol=1e-4)
    # a slide by something else than optimization_stride takes the kernel path; same semantics
    sol.slide_control_seq(3)
    U3, h3 = orc.slide_control_seq(U, hist, cfg["init_u"], 3)
    np.testing.assert_allclose(sol.get_control_seq(), U3, atol=1e-4)
    sol.compute_control(state)  # consumes the device copy produced by the slide kernel
    eps = O.generate_noise(1234, 2 * cfg["T"] * 5, cfg["K"], cfg["T"])[None]
    ref = orc.compute_control(state, U3, h3, eps)
    assert np.max(np.abs(sol.get_results(False)["U"] - ref["U"])) <= 2e-4
    sol.close()


def test_generator_mode_solve_matches_explicit_noise():
    """Default mode (device generator, seed 1234) == explicit mode fed with the oracle's noise."""
    cfg = S.make_config(1024, 50, track="ring")
    a = capi.Solver(cfg)
    a.seed(1234, 0)
    a.compute_control(cfg["start_state"])
    ra = a.get_results()
    b = capi.Solver(cfg)
    b.set_noise(noise_for(cfg, 1234))
    b.compute_control(cfg["start_state"])
    rb = b.get_results()
    np.testing.assert_array_equal(ra["costs"].view(np.uint32), rb["costs"].view(np.uint32))
    np.testing.assert_array_equal(ra["U"].view(np.uint32), rb["U"].view(np.uint32))
    a.close(); b.close()


def test_async_and_stage_timing():
    cfg = S.make_config(1024, 50, track="ring")
    sol = capi.Solver(cfg)
    sol.enable_stage_timing(True)
    for _ in range(3):
        sol.compute_control_async(cfg["start_state"])
        sol.synchronize()
    st = sol.get_stage_times()
    assert st["n_solves"] == 3 and st["rollout_ms"] > 0 and st["total_ms"] >= st["rollout_ms"]
    sol.close()


@pytest.mark.parametrize("K,T,layers", [(4096, 100, None), (64, 37, None), (512, 50, [6, 32, 32, 32, 32, 4])])
def test_quad_kernel_loops_are_reproducible(K, T, layers):
    """The four-wave kernel hands data between wavefronts through LDS sequence words: a lost or early
    hand-over would show up as a control sequence that differs run against run (or from the
    single-wave form on the same stream of draws), or as a poisoned (NaN) solve."""
    cfg = S.make_config(K, T, layers=layers, track="oval")
    sols = [capi.Solver(dict(cfg, seed=7)) for _ in range(5)]
    sols[0].set_rollout_variant("quad")
    sols[1].set_rollout_variant("quad")
    sols[2].set_rollout_variant("multi4")   # in-kernel generator in its control wave, like the quad form
    sols[3].set_rollout_variant("multi2")
    sols[4].set_rollout_variant("fused")    # stand-alone generator kernel: the same stream of draws
    x = cfg["start_state"]
    for it in range(300):
        us = []
        for s in sols[:4] if it >= 40 else sols:
            s.compute_control(x)
            us.append(s.get_control_seq())
            s.slide_control_seq(1)
        assert np.all(np.isfinite(us[0]))
        for u in us[1:]:
            np.testing.assert_array_equal(us[0].view(np.uint32), u.view(np.uint32))
    for s in sols:
        s.close()


@pytest.mark.parametrize("variant", ["row", "quad", "fused", "multi4", "valu"])
def test_projective_costmap_transform(variant):
    """updateTransform (costs.cu:175-188) accepts a full homography: w = r_c1.z x + r_c2.z y + trs.z != 1
    takes the kernels' u/w, v/w path (the shipped maps are affine and skip the two divides)."""
    cfg = S.make_config(256, 40, track="oval")
    r_c1 = np.array(cfg["r_c1"], np.float32)
    r_c2 = np.array(cfg["r_c2"], np.float32)
    trs = np.array(cfg["trs"], np.float32)
    r_c1[2], r_c2[2], trs[2] = 0.004, -0.003, 1.1
    cfg = dict(cfg, r_c1=r_c1, r_c2=r_c2, trs=trs)
    ref, got = _solve_both(cfg, U0=warm_U(cfg), variant=variant)
    plain, _ = _solve_both(dict(cfg, r_c1=S.make_config(256, 40, track="oval")["r_c1"],
                                r_c2=S.make_config(256, 40, track="oval")["r_c2"],
                                trs=S.make_config(256, 40, track="oval")["trs"]), U0=warm_U(cfg), variant=variant)
    assert np.max(np.abs(ref["costs"] - plain["costs"])) > 1.0  # the homography really changes the lookups
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    err = rel_err(got["costs"], ref["costs"])
    assert int(np.sum(err > 1e-4)) <= 3 and float(np.percentile(err, 95)) < 1e-5
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4


@pytest.mark.parametrize("name,negate", [("shallow_network_08_20_2020", False), ("wider_deeper_network_08_20_2020", False),
                                         ("gazebo_nnet_09_12_2018", True)])
def test_solve_with_the_other_shipped_models(golden_dir, name, negate):
    """Whole solves with the other weight files of params/models (6-32-32-4 with negate_yaw_der = false,
    the four-hidden-layer 6-64-64-64-64-4 net, the gazebo net): MFMA forms against the oracle and each other."""
    layers, theta = P.load_model_npz(os.path.join(golden_dir, "models", name + ".npz"))
    cfg = S.make_config(256, 40, layers=layers, theta=theta, track="oval", negate_yaw_der=negate)
    U0 = np.tile(np.array([0.0, 0.25], np.float32), (40, 1))
    ref, q = _solve_both(cfg, U0=U0, variant="quad")
    _, f = _solve_both(cfg, U0=U0, variant="fused")
    assert "quad" in q["variant"] and "fused" in f["variant"]
    np.testing.assert_array_equal(q["costs"].view(np.uint32), f["costs"].view(np.uint32))
    if layers[1] == 64:  # the eight-wave form of the 64-wide nets: the same bits; the automatic choice at this size is the
        # 4x4x1-MFMA form, whose output layer is a butterfly (its own oracle mode + the nominal criteria inside _solve_both)
        _, x = _solve_both(cfg, U0=U0, variant="oct")
        assert "oct8w" in x["variant"]
        np.testing.assert_array_equal(x["costs"].view(np.uint32), f["costs"].view(np.uint32))
        np.testing.assert_array_equal(x["U"].view(np.uint32), f["U"].view(np.uint32))
        r3, m = _solve_both(cfg, U0=U0)
        assert "m44" in m["variant"]
        np.testing.assert_array_equal(m["V"].view(np.uint32), r3["V"][-1].view(np.uint32))
        assert float(np.percentile(rel_err(m["costs"], r3["costs"]), 95)) < 1e-5 and np.max(np.abs(m["U"] - r3["U"])) <= 1e-4
    np.testing.assert_array_equal(q["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    err = rel_err(q["costs"], ref["costs"])
    assert int(np.sum(err > 1e-4)) <= 3 and float(np.percentile(err, 95)) < 1e-5
    assert np.max(np.abs(q["U"] - ref["U"])) <= 1e-4


def test_solve_with_a_model_written_by_the_reference_trainer(golden_dir):
    """The 6-16-24-4 network the reference's torch_model_to_npz wrote (gen_model_writer_golden.py): not an
    MFMA shape, so the whole solve runs on the generic VALU kernel."""
    layers, theta = P.load_model_npz(os.path.join(golden_dir, "models", "trained_writer_6_16_24_4.npz"))
    assert layers == [6, 16, 24, 4]
    cfg = S.make_config(512, 40, layers=layers, theta=theta, track="oval")
    U0 = np.tile(np.array([0.0, 0.25], np.float32), (40, 1))
    ref, got = _solve_both(cfg, U0=U0)
    assert got["variant"] == "valu_lds"
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    err = rel_err(got["costs"], ref["costs"])
    assert int(np.sum(err > 1e-4)) <= 5 and float(np.percentile(err, 95)) < 1e-5
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4


@pytest.mark.parametrize("T", [2, 3, 5, 15, 16, 17, 33, 65, 301])
@pytest.mark.parametrize("family", ["nn", "bf"])
def test_horizon_edge_cases(golden_dir, T, family):
    """Horizons around the kernels' internal granularities (4-step noise chunks, 16-step hand-over rings,
    one-step software pipelining) and a long one, K = 64 (a single wavefront group) and 192: every kernel
    form against the oracle and bit-identical to the others.  T = 1 is rejected by mppi_create."""
    extra = {}
    if family == "bf":
        extra["bf_W"] = P.load_bf_npz(os.path.join(golden_dir, "models", "basis_function_09_12_2018.npz"))
    variants = ["row", "quad", "fused", "multi4", "multi2", "valu", "valu_lds"] if family == "nn" else ["auto", "fused"]
    for K in (64, 192):
        cfg = S.make_config(K, T, track="oval", **extra)
        U0 = warm_U(cfg)
        hist = np.array([0.01, 0.2, -0.02, 0.25], np.float32)
        first = None
        for v in variants:
            ref, got = _solve_both(cfg, U0=U0, hist=hist, variant=v)
            np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
            assert float(np.percentile(rel_err(got["costs"], ref["costs"]), 90)) < 2e-5
            assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4
            if first is None:
                first = got
            else:
                np.testing.assert_array_equal(got["costs"].view(np.uint32), first["costs"].view(np.uint32))
                np.testing.assert_array_equal(got["U"].view(np.uint32), first["U"].view(np.uint32))
    with pytest.raises(capi.MppiError):
        capi.Solver(S.make_config(64, 1, track="oval", **extra))


@pytest.mark.parametrize("opt", [1, 3, 16])
@pytest.mark.parametrize("variant", ["row", "quad", "fused", "multi4", "valu"])
def test_slide_strides_up_to_the_horizon(opt, variant):
    """Device-resident loops (in-kernel generator, slid copy left by the tail kernel or made by the slide
    kernel, host re-upload every other tick) with optimization strides and slide strides from 1 to T,
    T = 17 (one more than the hand-over ring).  A stride beyond T is refused: the reference would index
    before U_ (mppi_controller.cu:545-552)."""
    K, T = 192, 17
    for sl in (opt, 1, T):
        cfg = S.make_config(K, T, track="ring", opt_stride=opt)
        orc = O.Oracle(cfg, fma_mode=1, nthreads=8)
        sol = capi.Solver(cfg)
        sol.set_rollout_variant(variant)
        sol.seed(77, 0)
        U = np.zeros((T, 2), np.float32)
        hist = np.zeros(4, np.float32)
        state = cfg["start_state"].copy()
        for it in range(4):
            eps = O.generate_noise(77, 2 * T * it, K, T)[None]
            ref = orc.compute_control(state, U, hist, eps)
            sol.compute_control(state)
            got = sol.get_results(with_vectors=False)
            assert np.max(np.abs(got["U"] - ref["U"])) <= 2e-4, (sl, it)
            U, hist = orc.slide_control_seq(ref["U"], hist, cfg["init_u"], sl)
            if it % 2 == 1:  # continue from the oracle's sequence: host slide + upload
                sol.set_control_seq(ref["U"])
                sol.slide_control_seq(sl)
                np.testing.assert_array_equal(sol.get_control_seq(), U)
            else:            # continue from the device copy
                sol.slide_control_seq(sl)
                np.testing.assert_allclose(sol.get_control_seq(), U, atol=2e-4)
                U, hist = sol.get_control_seq().copy(), sol.get_control_hist().copy()
        with pytest.raises(capi.MppiError):
            sol.slide_control_seq(T + 1)
        sol.close()


@pytest.mark.parametrize("K", [8256, 65600, 131072])
def test_many_chunk_rows_are_reduced_correctly(K):
    """K far beyond one reduction chunk (4096 rollouts per workgroup, a ragged last chunk at 8256 / 65600,
    700-1300 workgroups at T = 40): several fresh handles, every solve against the oracle.  (A missing
    barrier between staging and normalising the chunk's weights used to corrupt about one row in a
    hundred solves at these sizes -- found by a sweep over K, never by the K <= 16384 cases.)"""
    T = 40
    cfg = S.make_config(K, T, track="oval")
    eps = noise_for(cfg)
    U0 = warm_U(cfg)
    hist = np.array([0.01, 0.2, -0.02, 0.25], np.float32)
    ref = O.Oracle(cfg, fma_mode=1, nthreads=16).compute_control(cfg["start_state"], U0, hist, eps)
    for fresh in range(3):
        sol = capi.Solver(cfg)
        for rep in range(4):
            sol.set_control_seq(U0)
            sol.set_control_hist(hist)
            sol.set_noise(eps)
            sol.compute_control(cfg["start_state"])
            got = sol.get_results(with_vectors=False)
            assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4, (fresh, rep)
            assert abs(got["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"])
        sol.close()
