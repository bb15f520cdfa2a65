"""The latency form of 64-wide nets on v_mfma_f32_4x4x1 with A-matrix broadcast (rollout_m44.hip): four rollouts per wave, all
hidden weights in registers, the OUTPUT layer as a butterfly over the lanes; the hidden layers as TWO accumulation chains (even /
odd k: the automatic form "m44", oracle fma_mode 5) or as one chain in the reference's k-ascending order ("m44_chain", fma_mode 3).

Like the row-tree form of 32-wide nets (tests/test_row_tree_gpu.py) it is held to two bars:
  * its own oracle mode (tests/helpers.py: oracle_mode_for): applied controls bit for bit, costs p99 < 5e-6,
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
from tests.helpers import noise_for, oracle_mode_for, rel_err, warm_U

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


CASES = [(m, K, T, v) for v in ("m44", "m44_chain")
         for (m, K, T) in (("h64", 64, 7), ("h64", 256, 40), ("h64", 4096, 100), ("h64", 2048, 100),
                           ("wd", 1920, 100), ("wd", 192, 37), ("wd", 4096, 60))]


@pytest.mark.parametrize("model,K,T,variant", CASES)
def test_m44_form_against_its_mode_and_the_nominal_oracle(golden_dir, model, K, T, variant):
    cfg = _cfg(golden_dir, model, K, T)
    U0 = warm_U(cfg)
    eps = noise_for(cfg, 1234)
    hist = np.zeros(4, np.float32)
    got = _gpu(cfg, U0, eps, variant)
    assert "m44" in got["variant"] and ("m44_split" in got["variant"]) == (variant == "m44")
    exact = _gpu(cfg, U0, eps, "oct")
    ref2 = O.Oracle(cfg, fma_mode=oracle_mode_for(got["variant"]), nthreads=8).compute_control(cfg["start_state"], U0, hist, eps)
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


@pytest.mark.parametrize("model", ["h64", "wd"])
def test_m44_generator_mode_equals_explicit_noise(golden_dir, model):
    cfg = _cfg(golden_dir, model, 512, 33)
    U0 = warm_U(cfg)
    eps = noise_for(cfg, 1234)
    first = _gpu(cfg, U0, eps, "m44")
    sol = capi.Solver(cfg)
    sol.set_rollout_variant("m44")
    sol.set_control_seq(U0)
    sol.seed(1234, 0)
    sol.compute_control(cfg["start_state"])
    g1 = sol.get_results()
    np.testing.assert_array_equal(g1["U"].view(np.uint32), first["U"].view(np.uint32))
    np.testing.assert_array_equal(g1["costs"].view(np.uint32), first["costs"].view(np.uint32))
    sol.close()


def test_m44_chain_hidden_layers_keep_the_reference_order(golden_dir):
    """"m44_chain": only the output layer is re-associated: against the row64 form (the same hidden chains on the vector ALU, another
    output butterfly) and the oct form (everything in the reference's order) the costs agree to the last digits on all
    but the threshold-grazing rollouts."""
    cfg = _cfg(golden_dir, "wd", 512, 60)
    U0 = warm_U(cfg)
    eps = noise_for(cfg, 1234)
    a, b, c = _gpu(cfg, U0, eps, "m44_chain"), _gpu(cfg, U0, eps, "row64"), _gpu(cfg, U0, eps, "oct")
    for other in (b, c):
        assert float(np.percentile(rel_err(a["costs"], other["costs"]), 99)) < 5e-6
        assert np.max(np.abs(a["U"] - other["U"])) <= 1e-4


@pytest.mark.parametrize("wave", range(1, 9))
def test_m44_starved_wave_fails_the_solve_loudly(golden_dir, wave):
    """Roles: 1 .. 4 dynamics waves, then pose, cost, control, noise wave (mppi_debug_inject_handover_fault)."""
    cfg = _cfg(golden_dir, "h64", 256, 40)
    sol = capi.Solver(cfg)
    sol.set_rollout_variant("m44")
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
