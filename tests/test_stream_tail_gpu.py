"""The one-launch tail of many-chunk solves (K > 8192; csrc/solve_kernels.hip: solve_tail_stream_kernel) on its own: weights
workgroups + row workgroups that hand chunk minima / sums, {beta, eta} and chain results over as {value, tag} granules.  The
randomised whole-solve tests draw K <= 6400 and never reach it; here random many-chunk shapes -- ragged last chunks, 3 to 20
chunks per row, the leaders' both ways to beta (all costs in one batch up to 16 384 rollouts, exchanged chunk minima beyond),
short and long horizons -- are held to
  * the oracle's weighting + weighted reduction + smoothing fed with the GPU's OWN costs and applied controls (no threshold chaos
    left: 2e-6, as test_rollout_costs_and_controls does for one-chunk rows), weights and trajectory cost included;
  * themselves: the same solve again on the same handle and on a fresh one -- bit for bit (a race in a granule hand-over, a tag
    left from the launch before, an order that depends on arrival would show here);
and a two-iteration solve beyond 8192 rollouts (the non-last iteration leaves the raw mean for the next rollout, smooths nothing)
against the oracle, teacher-forced."""
import numpy as np
import pytest

from autorally_amd import capi
from autorally_amd import synthetic as S
from oracle import oracle as O
from tests.helpers import noise_for, rel_err, warm_U, solve_with_iterations, teacher_forced_iterations, iteration_ok

pytestmark = pytest.mark.gpu

SHAPES = [(8256, 7), (12288, 33), (12352, 100), (16384, 2), (16448, 41), (20480, 60), (24640, 19), (32768, 25), (40960, 12), (81984, 9)]


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
