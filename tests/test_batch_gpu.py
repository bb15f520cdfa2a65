"""mppi_compute_control_batch: the solves of several controllers in one launch (runControlLoop's two controllers,
run_control_loop.cuh:218-219).  Every instance's results must equal its own stand-alone solve bit for bit --
explicit noise, generator mode over successive ticks with slides, different K / costmaps / cost parameters per
instance, two iterations -- and a batch the kernels cannot share a launch for must fall back to per-handle solves
with the same results."""
import numpy as np
import pytest

from autorally_amd import capi
from autorally_amd import params as P
from autorally_amd import synthetic as S
from oracle import oracle as O
from tests.helpers import noise_for, warm_U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from autorally_amd import build as B
    B.build()
    assert capi.lib().mppi_device_count() >= 1


def _same(a, b):
    for k in ("U", "costs", "w"):
        np.testing.assert_array_equal(a[k].view(np.uint32), b[k].view(np.uint32), err_msg=k)
    assert a["traj_cost"] == b["traj_cost"]


def test_two_controllers_one_launch_equal_their_own_solves_and_the_oracle():
    """The reference's deployment shape: two controllers of K=1920, T=100 on one costs / model object, solving
    from the measured and from the predicted state.  Explicit noise: batch == stand-alone == oracle."""
    cfg = S.make_config(1920, 100, track="oval")
    st_a = cfg["start_state"].copy()
    st_p = st_a.copy()
    st_p[0] += 0.4
    st_p[4] -= 0.5
    U0 = warm_U(cfg)
    eps = [noise_for(cfg, seed=11), noise_for(cfg, seed=12)]
    states = [st_a, st_p]
    alone = []
    for e, st in zip(eps, states):
        s = capi.Solver(cfg)
        s.set_control_seq(U0)
        s.set_noise(e)
        s.compute_control(st)
        alone.append(dict(s.get_results(), V=s.get_applied_controls()))
        s.close()
    sols = [capi.Solver(cfg), capi.Solver(cfg)]
    for s, e in zip(sols, eps):
        s.set_control_seq(U0)
        s.set_noise(e)
    capi.compute_control_batch(sols, states)
    for s, ref, e, st in zip(sols, alone, eps, states):
        got = dict(s.get_results(), V=s.get_applied_controls())
        _same(got, ref)
        np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"].view(np.uint32))
        orc = O.Oracle(cfg, fma_mode=1, nthreads=8).compute_control(st, U0, np.zeros(4, np.float32), e)
        np.testing.assert_array_equal(got["V"].view(np.uint32), orc["V"][-1].view(np.uint32))
        assert np.max(np.abs(got["U"] - orc["U"])) <= 1e-4
        assert abs(got["traj_cost"] - orc["traj_cost"]) <= 1e-4 * abs(orc["traj_cost"])
    for s in sols:
        s.close()


@pytest.mark.parametrize("shapes", [
    [(1920, 100, None), (1920, 100, None)],
    [(1024, 60, None), (2048, 37, None), (512, 100, None)],          # different K and T per instance
    [(1024, 50, [6, 64, 64, 4]), (1024, 50, [6, 64, 64, 4])],        # a 64-wide net forced into the quad form
    [(640, 33, None), (640, 33, None), (640, 33, None), (640, 33, None)],
])
@pytest.mark.parametrize("form", ["quad", "row"])
def test_batched_ticks_in_generator_mode_equal_separate_handles(shapes, form):
    """Six ticks of solve + slide per controller, device generators, asynchronous batch + per-handle collection:
    the batched controllers follow stand-alone controllers bit for bit (control sequence, history, costs)."""
    if form == "row" and any(layers for _, _, layers in shapes):
        pytest.skip("the row form exists for 6-32-32-4")
    cfgs = []
    for i, (K, T, layers) in enumerate(shapes):
        kw = {}
        if layers:
            l, th = P.synthetic_model(layers, seed=4)
            kw = dict(layers=l, theta=th)
        cost = dict(P.DEFAULT_COST)
        if i % 2:
            cost.update(steering_coeff=0.3, desired_speed=5.0)  # a control-cost instance next to one without
        cfgs.append(S.make_config(K, T, track="oval", instance=i, seed=77 + i, cost=cost, **kw))
    ref, bat = [capi.Solver(c) for c in cfgs], [capi.Solver(c) for c in cfgs]
    for s in ref + bat:
        s.set_rollout_variant(form)
    states = [c["start_state"].copy() for c in cfgs]
    for tick in range(6):
        for s, st in zip(ref, states):
            s.compute_control(st)
        capi.compute_control_batch(bat, states, blocking=(tick % 2 == 0))
        for r, b, c in zip(ref, bat, cfgs):
            _same(b.get_results(), r.get_results())
            np.testing.assert_array_equal(b.get_control_hist(), r.get_control_hist())
            stride = 1 if tick % 3 else 2
            r.slide_control_seq(stride)
            b.slide_control_seq(stride)
            np.testing.assert_array_equal(b.get_control_seq().view(np.uint32), r.get_control_seq().view(np.uint32))
        states = [st + np.float32(0.01) * np.arange(7, dtype=np.float32) for st in states]
    assert all(("row8w" if form == "row" else "quad") in s.rollout_variant() for s in bat)
    for s in ref + bat:
        s.close()


def test_basis_function_controllers_share_a_launch_too(golden_dir):
    """path_integral_bf's two controllers (K = 2560 each, three wavefronts per 64 rollouts: 240 waves): batched
    ticks in generator mode follow stand-alone handles bit for bit."""
    import os
    W = P.load_bf_npz(os.path.join(golden_dir, "models", "basis_function_09_12_2018.npz"))
    cfgs = [S.make_config(2560, 100, track="oval", bf_W=W, seed=5), S.make_config(2560, 100, track="oval", bf_W=W, seed=6)]
    ref, bat = [capi.Solver(c) for c in cfgs], [capi.Solver(c) for c in cfgs]
    assert all(s.rollout_variant() == "basis_funcs25_valu_3w" for s in bat)
    states = [c["start_state"].copy() for c in cfgs]
    states[1][0] += 0.5
    for tick in range(5):
        for s, st in zip(ref, states):
            s.compute_control(st)
        capi.compute_control_batch(bat, states)
        for r, b in zip(ref, bat):
            _same(b.get_results(), r.get_results())
            r.slide_control_seq(1)
            b.slide_control_seq(1)
    # a mixed pair (network + basis functions) cannot share a kernel: per-handle solves, same results
    mixed_cfg = S.make_config(1024, 100, track="oval")
    m_ref, m_bat = capi.Solver(mixed_cfg), capi.Solver(mixed_cfg)
    m_ref.compute_control(mixed_cfg["start_state"])
    ref[0].compute_control(states[0])
    capi.compute_control_batch([m_bat, bat[0]], [mixed_cfg["start_state"], states[0]])
    _same(m_bat.get_results(), m_ref.get_results())
    _same(bat[0].get_results(), ref[0].get_results())
    for s in ref + bat + [m_ref, m_bat]:
        s.close()


def test_nominal_trajectories_of_a_pair_in_lockstep_equal_the_single_replays(golden_dir):
    """mppi_nominal_traj_pair (the computeNominalTraj of both controllers of a tick, two host replays advancing in
    lockstep) returns bit for bit what two mppi_nominal_traj calls return; pairs that cannot run in lockstep (different
    horizons, the basis-function model) are served one after the other."""
    import os
    cfg = S.make_config(512, 100, track="oval")
    a, b = capi.Solver(cfg), capi.Solver(dict(cfg, seed=9))
    sa, sb = cfg["start_state"].copy(), cfg["start_state"].copy()
    sb[0] += 0.4
    sb[4] -= 1.0
    capi.compute_control_batch([a, b], [sa, sb])
    (xa, ua), (xb, ub) = capi.nominal_traj_pair(a, sa, b, sb)
    ra, rb = a.nominal_traj(sa), b.nominal_traj(sb)
    for got, ref in ((xa, ra[0]), (ua, ra[1]), (xb, rb[0]), (ub, rb[1])):
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.max(np.abs(xa - xb)) > 1e-3
    orc = O.Oracle(cfg, fma_mode=1)
    rs, _ = orc.nominal_traj(sa, a.get_control_seq())
    assert np.max(np.abs(xa - rs)) <= 1e-4
    W = P.load_bf_npz(os.path.join(golden_dir, "models", "basis_function_09_12_2018.npz"))
    c = capi.Solver(S.make_config(256, 60, track="oval", bf_W=W))
    c.compute_control(sa)
    (xa2, _), (xc, uc) = capi.nominal_traj_pair(a, sa, c, sa)
    np.testing.assert_array_equal(xa2.view(np.uint32), ra[0].view(np.uint32))
    rc = c.nominal_traj(sa)
    np.testing.assert_array_equal(xc.view(np.uint32), rc[0].view(np.uint32))
    np.testing.assert_array_equal(uc.view(np.uint32), rc[1].view(np.uint32))
    for s in (a, b, c):
        s.close()


def test_paired_host_work_on_a_helper_thread_equals_the_single_calls():
    """mppi_set_host_threads(2): the two nominal replays / the two DDP passes of a tick run side by side (the caller's thread
    and one helper thread); results bit for bit those of the single calls and of the one-thread setting, tick after tick
    (the helper sleeps between ticks, is woken by the batched solve, and is found asleep again after a pause)."""
    import time
    cfg = S.make_config(512, 100, track="oval")
    a, b = capi.Solver(cfg), capi.Solver(dict(cfg, seed=9))
    sa, sb = cfg["start_state"].copy(), cfg["start_state"].copy()
    sb[0] += 0.4
    sb[4] -= 1.0

    def bits(x):
        return np.ascontiguousarray(x).view(np.uint32)
    try:
        for tick in range(6):
            capi.set_host_threads(2 if tick % 3 else 1)
            capi.compute_control_batch([a, b], [sa, sb])
            (xa, ua), (xb, ub) = capi.nominal_traj_pair(a, sa, b, sb)
            ra, rb = a.nominal_traj(sa), b.nominal_traj(sb)
            for got, ref in ((xa, ra[0]), (ua, ra[1]), (xb, rb[0]), (ub, rb[1])):
                np.testing.assert_array_equal(bits(got), bits(ref))
            capi.compute_feedback_gains_pair(a, sa, b, sa, (xa, ua), (xb, ub))
            ga, gb = a.feedback_gains(), b.feedback_gains()
            ha, hb = a.compute_feedback_gains(sa, xa, ua), b.compute_feedback_gains(sa, xb, ub)
            for g, h in ((ga, ha), (gb, hb)):
                for key in ("feedback", "feedforward", "x", "u"):
                    np.testing.assert_array_equal(bits(g[key]), bits(h[key]))
                assert g["total_cost"] == h["total_cost"]
            assert np.max(np.abs(ga["feedback"] - gb["feedback"])) > 0
            # without targets: each pass replays its own handle's sequence from the given state
            capi.compute_feedback_gains_pair(a, sa, b, sb)
            np.testing.assert_array_equal(bits(a.feedback_gains()["feedback"]), bits(a.compute_feedback_gains(sa)["feedback"]))
            np.testing.assert_array_equal(bits(b.feedback_gains()["feedback"]), bits(b.compute_feedback_gains(sb)["feedback"]))
            for s in (a, b):
                s.slide_control_seq(1)
            if tick == 3:
                time.sleep(0.05)  # the helper goes back to sleep (1 ms after its last job)
        with pytest.raises(capi.MppiError):
            capi.set_host_threads(3)
        with pytest.raises(capi.MppiError):
            capi.compute_feedback_gains_pair(a, sa, a, sa)
    finally:
        capi.set_host_threads(1)
        for s in (a, b):
            s.close()


def test_batch_then_single_then_batch_and_two_iterations():
    """Transitions between the batch stream and a handle's own stream (single solve, result vectors, applied
    controls, set_noise, seed) keep every handle's sequence of results; num_iters = 2 batches both iterations."""
    cfg = S.make_config(1024, 40, track="ring", num_iters=2)
    ref, bat = [capi.Solver(cfg) for _ in range(2)], [capi.Solver(cfg) for _ in range(2)]
    for i in range(2):
        ref[i].seed(500 + i, 0)
        bat[i].seed(500 + i, 0)
    st = [cfg["start_state"].copy(), cfg["start_state"].copy()]
    st[1][1] += 0.3
    seq = ["batch", "single", "batch", "explicit", "batch", "batch"]
    for step, op in enumerate(seq):
        if op == "explicit":
            eps = noise_for(cfg, seed=900 + step)
            for s in ref + bat:
                s.set_noise(eps)
        for i in range(2):
            ref[i].compute_control(st[i])
        if op == "single":
            for i in range(2):
                bat[i].compute_control(st[i])
        else:
            capi.compute_control_batch(bat, st)
        for i in range(2):
            a, b = bat[i].get_results(), ref[i].get_results()
            _same(a, b)
            np.testing.assert_array_equal(bat[i].get_applied_controls().view(np.uint32),
                                          ref[i].get_applied_controls().view(np.uint32))
            ref[i].slide_control_seq(1)
            bat[i].slide_control_seq(1)
    for s in ref + bat:
        s.close()


def test_batches_the_library_cannot_share_a_launch_for_fall_back_to_per_handle_solves(golden_dir):
    """More groups than CUs, different layer lists, the basis-function model, a single handle: the call still
    solves every handle (on its own stream), with its own results."""
    import os
    W = P.load_bf_npz(os.path.join(golden_dir, "models", "basis_function_09_12_2018.npz"))
    l64, th64 = P.synthetic_model([6, 64, 64, 4], seed=4)
    groups = [
        [S.make_config(4096, 30, track="oval"), S.make_config(4096, 30, track="oval", instance=1)],   # 512 groups
        [S.make_config(512, 30, track="oval"), S.make_config(512, 30, track="oval", layers=l64, theta=th64)],
        [S.make_config(640, 30, track="ring", bf_W=W), S.make_config(512, 30, track="ring")],
        [S.make_config(256, 30, track="ring")],
    ]
    for cfgs in groups:
        ref, bat = [capi.Solver(c) for c in cfgs], [capi.Solver(c) for c in cfgs]
        states = [c["start_state"] for c in cfgs]
        for _ in range(2):
            for s, st in zip(ref, states):
                s.compute_control(st)
            capi.compute_control_batch(bat, states)
            for r, b in zip(ref, bat):
                _same(b.get_results(), r.get_results())
        for s in ref + bat:
            s.close()
    # the same handle twice / an empty batch are refused
    s = capi.Solver(groups[3][0])
    with pytest.raises(capi.MppiError):
        capi.compute_control_batch([s, s], [states[0], states[0]])
    s.close()


@pytest.mark.parametrize("state_kind", ["finite", "inf_x", "nan_speed"])
def test_a_mixed_batch_equals_the_single_solves_also_for_non_finite_states(state_kind):
    """One controller on an affine costmap transform without control cost, its partner on a projective transform WITH control
    cost: a kernel specialised for the superset would compute 0 * x where the single solve computes nothing -- different bits
    for a non-finite x.  The library shares a launch only between instances of equal specialisation, so this pair is solved
    one by one and every controller's results are those of its own mppi_compute_control, finite or not."""
    a = S.make_config(512, 30, track="oval")
    cost_b = dict(P.DEFAULT_COST, steering_coeff=0.3, throttle_coeff=0.25)
    b = S.make_config(512, 30, track="oval", cost=cost_b)
    # a projective costmap transform: third components of the columns non-trivial (costs.cu:373-377 divides by w)
    b["r_c1"] = (b["r_c1"][0], b["r_c1"][1], 0.001)
    b["r_c2"] = (b["r_c2"][0], b["r_c2"][1], -0.002)
    sa, sb = a["start_state"].copy(), b["start_state"].copy()
    if state_kind == "inf_x":
        sa[0] = np.inf
    elif state_kind == "nan_speed":
        sa[4] = np.nan
    ref, bat = [capi.Solver(a), capi.Solver(b)], [capi.Solver(a), capi.Solver(b)]
    for s in ref + bat:
        s.seed(77, 0)
    outs = []
    for pair, batched in ((ref, False), (bat, True)):
        try:
            if batched:
                capi.compute_control_batch(pair, [sa, sb])
            else:
                pair[0].compute_control(sa)
                pair[1].compute_control(sb)
            outs.append([s.get_results() for s in pair])
        except capi.MppiError as e:
            outs.append(("error", e.status))
    if isinstance(outs[0], tuple) or isinstance(outs[1], tuple):
        assert outs[0] == outs[1], outs  # the same loud failure both ways
    else:
        for r, g in zip(outs[0], outs[1]):
            for key in ("U", "costs", "w"):
                np.testing.assert_array_equal(g[key].view(np.uint32), r[key].view(np.uint32), err_msg=key)
    for s in ref + bat:
        s.close()
