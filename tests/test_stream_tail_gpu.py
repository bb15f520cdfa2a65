"""The one-launch tail of many-chunk solves (K > 4096; csrc/solve_kernels.hip: solve_tail_stream_kernel) on its own: weights
workgroups + row workgroups that hand chunk minima / sums, {beta, eta} and chain results over as {value, tag} granules.  The
randomised whole-solve tests draw K <= 6400 and reach it with two chunks only; here many-chunk shapes -- ragged last chunks, 2 to 20
chunks per row, the leaders' both ways to beta (all costs in one batch up to 16 384 rollouts, exchanged chunk minima beyond),
short and long horizons -- are held to
  * the oracle's weighting + weighted reduction + smoothing fed with the GPU's OWN costs and applied controls (no threshold chaos
    left: 2e-6, as test_rollout_costs_and_controls does for one-chunk rows), weights and trajectory cost included;
  * themselves: the same solve again on the same handle and on a fresh one -- bit for bit (a race in a granule hand-over, a tag
    left from the launch before, an order that depends on arrival would show here);
and a two-iteration solve beyond 8192 rollouts (the non-last iteration leaves the raw mean for the next rollout, smooths nothing)
against the oracle, teacher-forced."""
import os

import numpy as np
import pytest

from autorally_amd import capi
from autorally_amd import synthetic as S
from oracle import oracle as O
from tests.helpers import noise_for, rel_err, warm_U, solve_with_iterations, teacher_forced_iterations, iteration_ok

pytestmark = pytest.mark.gpu

SHAPES = [(4160, 9), (6400, 50), (8192, 100), (8256, 7), (12288, 33), (12352, 100), (16384, 2), (16448, 41), (20480, 60), (24640, 19), (32768, 25), (40960, 12), (81984, 9)]


@pytest.mark.parametrize("K,T", SHAPES)
def test_stream_tail_reproduces_the_oracle_stages_and_itself(K, T):
    cfg = S.make_config(K, T, track="oval")
    U0 = warm_U(cfg, seed=K % 97)
    hist = np.array([0.02, 0.21, -0.01, 0.24], np.float32)
    eps = noise_for(cfg, 1000 + T)
    runs = []
    sol = capi.Solver(cfg)
    for rep in range(3):
        if rep == 2:  # a fresh handle: granule buffers zeroed, epoch 1
            sol.close()
            sol = capi.Solver(cfg)
        sol.set_control_seq(U0)
        sol.set_control_hist(hist)
        sol.set_noise(eps)
        sol.compute_control(cfg["start_state"])
        got = sol.get_results()
        got["V"] = sol.get_applied_controls()
        runs.append(got)
    sol.close()
    for other in runs[1:]:
        for key in ("U", "costs", "w"):
            np.testing.assert_array_equal(runs[0][key].view(np.uint32), other[key].view(np.uint32), err_msg=key)
        assert runs[0]["traj_cost"] == other["traj_cost"]
    got = runs[0]
    assert np.all(np.isfinite(got["U"])) and float(got["w"].max()) == 1.0
    orc = O.Oracle(cfg, fma_mode=1, nthreads=8)
    w, beta, eta, tc = orc.weights(got["costs"])
    assert float(np.abs(w - got["w"]).sum()) / float(w.sum()) < 1e-6  # expf(-gamma (J - beta)): libm against the device's expf
    assert beta == float(np.min(got["costs"]))
    # eta: the kernel adds chunk sums (pairwise inside a chunk, chunk order across); the reference's host loop -- and the oracle
    # -- add K weights one by one in fp32, a sum whose own rounding error grows like sqrt(K) eps (9e-6 relative at K = 82 000,
    # measured here).  Held against the oracle's reduction fed with the correctly rounded sum, the kernel is at 2e-6; against
    # the sequential one, inside its rounding error -- both far inside the 1e-4 of the contract.
    eta_exact = np.float32(np.sum(w, dtype=np.float64))
    assert abs(float(eta) - float(eta_exact)) <= 4e-8 * np.sqrt(K) * float(eta_exact)
    U2 = orc.savgol(orc.weighted_reduction(w, eta_exact, got["V"]), hist)
    assert np.max(np.abs(U2 - got["U"])) <= 2e-6, float(np.max(np.abs(U2 - got["U"])))
    U3 = orc.savgol(orc.weighted_reduction(w, eta, got["V"]), hist)
    assert np.max(np.abs(U3 - got["U"])) <= 2e-6 + 8e-8 * np.sqrt(K) * float(np.max(np.abs(got["U"])))
    assert abs(tc - got["traj_cost"]) <= (1e-5 + 8e-8 * np.sqrt(K)) * abs(tc)


def test_two_iterations_beyond_8192_rollouts():
    cfg = S.make_config(12352, 40, track="oval", num_iters=2)
    U0 = warm_U(cfg)
    hist = np.zeros(4, np.float32)
    eps = noise_for(cfg, 77)
    got, its, name = solve_with_iterations(cfg, "auto", U0, hist, eps)
    assert "multi4_tree" in name
    ms = teacher_forced_iterations(cfg, got, its, U0, hist, eps, fma_mode=1)
    assert len(ms) == 2
    for m in ms:
        assert iteration_ok(dict(m, V_equal=True)), m  # (the tree form's applied controls are held to ITS oracle mode elsewhere)
    assert np.all(np.isfinite(got["U"]))


def _solve_once(cfg, U0, hist, eps, n_solves=1, min_cost=None):
    sol = capi.Solver(cfg)
    if min_cost is not None:
        sol.debug_min_cost(min_cost)
    outs = []
    for i in range(n_solves):
        sol.set_control_seq(U0)
        sol.set_control_hist(hist)
        sol.set_noise(eps)
        sol.compute_control(cfg["start_state"])
        got = sol.get_results()
        got["from_rollout"] = sol.debug_min_cost()
        outs.append(got)
    sol.close()
    return outs


@pytest.mark.parametrize("K,T,layers", [(12352, 40, None), (20544, 30, None), (16384, 20, [6, 64, 64, 4])])
def test_beta_from_the_rollout_kernel_and_from_the_tail_are_the_same_bits(K, T, layers):
    """Beyond 8192 rollouts the rollout kernel leaves min_k costs[k] behind as a tagged atomic minimum (csrc/mppi_device.hpp:
    publish_min_cost) and the tail kernel reads it; switched off, the tail kernel reduces the costs and hands beta round as in
    the first version.  The hook says which way a solve went; the results are the same bits."""
    kw = {"layers": layers} if layers else {}
    cfg = S.make_config(K, T, track="oval", **kw)
    U0 = warm_U(cfg, seed=3)
    hist = np.array([0.0, 0.2, 0.01, 0.22], np.float32)
    eps = noise_for(cfg, 500 + T)
    on = _solve_once(cfg, U0, hist, eps, n_solves=2)
    off = _solve_once(cfg, U0, hist, eps, n_solves=2, min_cost=0)
    assert all(g["from_rollout"] for g in on) and not any(g["from_rollout"] for g in off)
    for a, b in zip(on, off):
        for key in ("U", "costs", "w"):
            np.testing.assert_array_equal(a[key].view(np.uint32), b[key].view(np.uint32), err_msg=key)
        assert a["traj_cost"] == b["traj_cost"] and float(a["w"].max()) == 1.0  # exp(-gamma (min - beta)) = 1: beta IS the minimum


def test_a_rollout_form_that_publishes_no_minimum_makes_the_tail_reduce_the_costs():
    cfg = S.make_config(12352, 20, track="oval")
    U0 = warm_U(cfg)
    hist = np.zeros(4, np.float32)
    eps = noise_for(cfg, 9)
    sol = capi.Solver(cfg)
    sol.set_rollout_variant("valu")  # one lane per rollout on the vector ALU: leaves the keys alone
    sol.set_control_seq(U0); sol.set_control_hist(hist); sol.set_noise(eps)
    sol.compute_control(cfg["start_state"])
    got = sol.get_results()
    assert not sol.debug_min_cost()
    assert float(got["w"].max()) == 1.0 and np.all(np.isfinite(got["U"]))
    sol.set_rollout_variant("auto")  # ... and the publishing form right behind it on the same handle
    sol.set_control_seq(U0); sol.set_control_hist(hist); sol.set_noise(eps)
    sol.compute_control(cfg["start_state"])
    got = sol.get_results()
    assert sol.debug_min_cost() and float(got["w"].max()) == 1.0
    sol.close()


def test_min_cost_tags_run_out_and_start_again():
    """A key is {~tag, cost}: a later launch's key beats every older one, nothing is ever reset -- until the 32-bit tag runs out
    (4e9 launches: two days at 25 000 solves a second).  Then the keys are wiped and the tags start again.  Started six launches
    before that point (MPPI_MIN_COST_TAG, read when the handle is created), a run of control ticks crosses it: every solve takes
    beta from the rollout kernel, and the bits are those of a handle nowhere near the wrap."""
    cfg = S.make_config(12352, 30, track="oval")
    state = np.asarray(cfg["start_state"], np.float32)
    runs = []
    for tag0 in (None, str(0xFFFFFFFF - 6)):
        if tag0 is not None:
            os.environ["MPPI_MIN_COST_TAG"] = tag0
        try:
            sol = capi.Solver(cfg)
        finally:
            os.environ.pop("MPPI_MIN_COST_TAG", None)
        sol.seed(11)
        seq = []
        for i in range(12):
            sol.compute_control(state)
            assert sol.debug_min_cost(), (tag0, i)
            got = sol.get_results(with_vectors=False)
            seq.append((got["U"].copy(), got["traj_cost"]))
            sol.slide_control_seq(1)
        sol.close()
        runs.append(seq)
    for (Ua, ta), (Ub, tb) in zip(*runs):
        np.testing.assert_array_equal(Ua.view(np.uint32), Ub.view(np.uint32))
        assert ta == tb
