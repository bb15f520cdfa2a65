"""The latency form of 64-wide nets (rollout_row64.hip): 32-lane rollouts on the vector ALU, hidden layers in the
reference's k-ascending order with their weights from LDS, the OUTPUT layer as a butterfly over the 32 lanes.

Like the row-tree form of 32-wide nets (tests/test_row_tree_gpu.py) it is held to two bars:
  * its own oracle mode (fma_mode 2: the butterfly's summation order): applied controls bit for bit, costs p99 < 5e-6,
    flipped rollouts <= K/200, U <= 1e-4;
  * the NOMINAL oracle (fma_mode 1, the reference's order): U L-inf <= 1e-4, trajectory cost rel <= 1e-4, flipped <= K/200.
Models: 6-64-64-4 (synthetic weights, BASELINE config 4's shape) and the reference's shipped 6-64-64-64-64-4
(wider_deeper_network_08_20_2020.npz, negate_yaw_der = false) at the reference's K = 1920.
"""
import os

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
    assert capi.lib().mppi_device_count() >= 1, "no gfx950 device: the HIP path cannot run"


def _cfg(golden_dir, model, K, T, track="oval"):
    if model == "wd":
        layers, theta = P.load_model_npz(os.path.join(golden_dir, "models", "wider_deeper_network_08_20_2020.npz"))
        return S.make_config(K, T, layers=layers, theta=theta, track=track, negate_yaw_der=False)
    layers, theta = P.synthetic_model([6, 64, 64, 4], seed=4)
    return S.make_config(K, T, layers=layers, theta=theta, track=track)


def _gpu(cfg, U0, eps, variant):
    sol = capi.Solver(cfg)
    sol.set_rollout_variant(variant)
    sol.set_control_seq(U0)
    sol.set_control_hist(np.zeros(4, np.float32))
    sol.set_noise(eps)
    sol.compute_control(cfg["start_state"])
    got = sol.get_results()
    got["V"] = sol.get_applied_controls()
    got["variant"] = sol.rollout_variant()
    sol.close()
    return got


CASES = [("h64", 64, 7, "row64"), ("h64", 256, 40, "row64_r16"), ("h64", 4096, 100, "row64"),
         ("h64", 2048, 100, "row64"), ("wd", 1920, 100, "row64"), ("wd", 192, 37, "row64_r16"), ("wd", 4096, 60, "row64"),
         ("h64", 16384, 150, "row64_r16")]  # BASELINE config 4 at full size: the tuned vector-ALU arm of its MFMA-vs-VALU A/B


@pytest.mark.parametrize("model,K,T,variant", CASES)
def test_row64_form_against_its_mode_and_the_nominal_oracle(golden_dir, model, K, T, variant):
    cfg = _cfg(golden_dir, model, K, T)
    U0 = warm_U(cfg)
    eps = noise_for(cfg, 1234)
    hist = np.zeros(4, np.float32)
    got = _gpu(cfg, U0, eps, variant)
    assert "row64_r16" in got["variant"]
    exact = _gpu(cfg, U0, eps, "oct")
    ref2 = O.Oracle(cfg, fma_mode=2, nthreads=8).compute_control(cfg["start_state"], U0, hist, eps)
    ref1 = O.Oracle(cfg, fma_mode=1, nthreads=8).compute_control(cfg["start_state"], U0, hist, eps)
    # ---- its own mode
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref2["V"][-1].view(np.uint32))
    err2 = rel_err(got["costs"], ref2["costs"])
    assert int(np.sum(err2 > 1e-4)) <= max(K // 200, 1), float(err2.max())
    assert float(np.percentile(err2, 99)) < 5e-6
    assert float(np.abs(got["w"] - ref2["w"]).sum()) / float(ref2["w"].sum()) < 1e-4
    assert np.max(np.abs(got["U"] - ref2["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref2["traj_cost"]) <= 1e-4 * abs(ref2["traj_cost"])
    # ---- the nominal oracle: north-star criteria
    err1 = rel_err(got["costs"], ref1["costs"])
    assert int(np.sum(err1 > 1e-4)) <= max(K // 200, 1), float(err1.max())
    assert np.max(np.abs(got["U"] - ref1["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref1["traj_cost"]) <= 1e-4 * abs(ref1["traj_cost"])
    e_exact = rel_err(exact["costs"], ref1["costs"])
    assert float(np.percentile(err1, 99)) < max(4 * float(np.percentile(e_exact, 99)), 2e-5)
    # ---- downstream stages alone
    orc = O.Oracle(cfg, fma_mode=1)
    w, _, eta, tc = orc.weights(got["costs"])
    U2 = orc.savgol(orc.weighted_reduction(w, eta, got["V"]), hist)
    assert np.max(np.abs(U2 - got["U"])) <= 2e-6
    assert abs(tc - got["traj_cost"]) <= 1e-5 * abs(tc)


@pytest.mark.parametrize("model,variant", [("h64", "row64"), ("wd", "row64_r16")])
def test_row64_generator_mode_equals_explicit_noise(golden_dir, model, variant):
    cfg = _cfg(golden_dir, model, 512, 33)
    U0 = warm_U(cfg)
    eps = noise_for(cfg, 1234)
    sol = capi.Solver(cfg)
    sol.set_rollout_variant(variant)
    sol.set_control_seq(U0)
    sol.seed(1234, 0)
    for it in range(2):  # two solves: the generator states carry over
        sol.compute_control(cfg["start_state"])
    gen = sol.get_results()
    ref = capi.Solver(cfg)
    ref.set_rollout_variant("oct_gen")
    ref.set_control_seq(U0)
    ref.seed(1234, 0)
    ref.compute_control(cfg["start_state"])
    first = _gpu(cfg, U0, eps, variant)
    sol2 = capi.Solver(cfg)
    sol2.set_rollout_variant(variant)
    sol2.set_control_seq(U0)
    sol2.seed(1234, 0)
    sol2.compute_control(cfg["start_state"])
    g1 = sol2.get_results()
    np.testing.assert_array_equal(g1["U"].view(np.uint32), first["U"].view(np.uint32))
    np.testing.assert_array_equal(g1["costs"].view(np.uint32), first["costs"].view(np.uint32))
    assert np.all(np.isfinite(gen["U"]))
    for s_ in (sol, ref, sol2):
        s_.close()


@pytest.mark.parametrize("variant,wave", [("row64_r16", 1), ("row64_r16", 2), ("row64_r16", 8), ("row64_r16", 9), ("row64_r16", 10), ("row64_r16", 11), ("row64_r16", 12)])
def test_row64_starved_wave_fails_the_solve_loudly(golden_dir, variant, wave):
    """Roles: 1 .. R/2 dynamics waves, then pose, cost, control, noise wave (mppi_debug_inject_handover_fault)."""
    cfg = _cfg(golden_dir, "h64", 256, 40)
    sol = capi.Solver(cfg)
    sol.set_rollout_variant(variant)
    sol.compute_control(cfg["start_state"])
    good = sol.get_results()
    assert np.all(np.isfinite(good["costs"]))
    sol.debug_inject_handover_fault(wave, 32)
    with pytest.raises(capi.MppiError) as e:
        sol.compute_control(cfg["start_state"])
    assert e.value.status == capi.ERR_HIP
    sol.debug_inject_handover_fault(0, 0)
    sol.reset_controls()
    sol.seed(cfg.get("seed", 1234), 0)
    sol.compute_control(cfg["start_state"])
    again = sol.get_results()
    np.testing.assert_array_equal(again["costs"].view(np.uint32), good["costs"].view(np.uint32))
    sol.close()
