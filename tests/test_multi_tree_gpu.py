"""The multi4 form with the output layer as a butterfly over the four lanes of a rollout (mfma_net.hpp: nn_last_tree; variant
"multi4_tree[_gen]", the automatic choice beyond two groups of 16 rollouts per CU): 8 of 28 (6-32-32-4) / 16 of 88 (6-64-64-4)
matrix instructions per step replaced by packed multiply-adds on the lane's own activations + two permlane swaps.

Held to the two bars of every tree form (tests/test_row_tree_gpu.py): its own oracle mode (4) at the bit-level criteria, the
nominal oracle (the reference's order) at the north-star criteria -- including BASELINE config 4 at full size."""
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


def _cfg(layers, K, T):
    if layers is None:
        return S.make_config(K, T, track="oval")
    l, th = P.synthetic_model(layers, seed=4)
    return S.make_config(K, T, layers=l, theta=th, track="oval")


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


CASES = [(None, 256, 40, "multi4_tree"), (None, 16384, 100, "auto"), ([6, 64, 64, 4], 320, 33, "multi4_tree_gen"),
         ([6, 64, 64, 4], 16384, 150, "auto"),  # BASELINE config 4 at full size
         ([6, 32, 32, 32, 32, 4], 8192, 47, "multi4_tree"), (None, 65536, 20, "auto")]


@pytest.mark.parametrize("layers,K,T,variant", CASES)
def test_multi4_tree_against_its_mode_and_the_nominal_oracle(layers, K, T, variant):
    cfg = _cfg(layers, K, T)
    U0 = warm_U(cfg)
    eps = noise_for(cfg, 1234)
    hist = np.zeros(4, np.float32)
    got = _gpu(cfg, U0, eps, variant)
    assert "multi4_tree" in got["variant"], got["variant"]
    exact = _gpu(cfg, U0, eps, "multi4")
    assert "multi4" in exact["variant"] and "tree" not in exact["variant"]
    ref4 = O.Oracle(cfg, fma_mode=4, nthreads=16).compute_control(cfg["start_state"], U0, hist, eps)
    ref1 = O.Oracle(cfg, fma_mode=1, nthreads=16).compute_control(cfg["start_state"], U0, hist, eps)
    # ---- its own mode
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref4["V"][-1].view(np.uint32))
    err4 = rel_err(got["costs"], ref4["costs"])
    assert int(np.sum(err4 > 1e-4)) <= max(K // 200, 1), float(err4.max())
    assert float(np.percentile(err4, 99)) < 5e-6
    assert float(np.abs(got["w"] - ref4["w"]).sum()) / float(ref4["w"].sum()) < 1e-4
    assert np.max(np.abs(got["U"] - ref4["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref4["traj_cost"]) <= 1e-4 * abs(ref4["traj_cost"])
    # ---- the nominal oracle: north-star criteria
    err1 = rel_err(got["costs"], ref1["costs"])
    print("nominal margin: K=%d T=%d %s: |dU|inf = %.3e, trajectory cost rel = %.3e (bound 1e-4 each), flipped %d of %d" % (
        K, T, got["variant"], float(np.max(np.abs(got["U"] - ref1["U"]))),
        abs(got["traj_cost"] - ref1["traj_cost"]) / abs(ref1["traj_cost"]), int(np.sum(err1 > 1e-4)), K))
    assert int(np.sum(err1 > 1e-4)) <= max(K // 200, 1), float(err1.max())
    assert np.max(np.abs(got["U"] - ref1["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref1["traj_cost"]) <= 1e-4 * abs(ref1["traj_cost"])
    e_exact = rel_err(exact["costs"], ref1["costs"])
    assert float(np.percentile(err1, 99)) < max(4 * float(np.percentile(e_exact, 99)), 2e-5)


def test_multi4_tree_generator_mode_equals_explicit_noise_over_several_solves():
    """eps from the stand-alone generator kernel (prefetched on a second stream) or from the control wave's own generator:
    the same streams, the same bits as explicit noise, solve after solve."""
    cfg = _cfg(None, 1024, 33)
    U0 = warm_U(cfg)
    sols = []
    for v in ("multi4_tree", "multi4_tree_gen"):
        s_ = capi.Solver(cfg)
        s_.set_rollout_variant(v)
        s_.set_control_seq(U0)
        s_.seed(1234, 0)
        sols.append(s_)
    ex = capi.Solver(cfg)
    ex.set_rollout_variant("multi4_tree")
    ex.set_control_seq(U0)
    for it in range(3):
        eps = O.generate_noise(1234, 2 * cfg["T"] * it, cfg["K"], cfg["T"])[None]
        ex.set_noise(eps)
        ex.compute_control(cfg["start_state"])
        r0 = ex.get_results()
        for s_ in sols:
            s_.compute_control(cfg["start_state"])
            r1 = s_.get_results()
            np.testing.assert_array_equal(r1["U"].view(np.uint32), r0["U"].view(np.uint32))
            np.testing.assert_array_equal(r1["costs"].view(np.uint32), r0["costs"].view(np.uint32))
    for s_ in sols + [ex]:
        s_.close()


@pytest.mark.parametrize("wave", range(1, 9))
def test_multi4_tree_starved_wave_fails_the_solve_loudly(wave):
    """Roles of the eight-wave multi form: 1 .. 4 dynamics waves, 5 pose, 6 cost, 7 control, 8 fetch wave."""
    cfg = _cfg(None, 256, 40)
    sol = capi.Solver(cfg)
    sol.set_rollout_variant("multi4_tree")
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
    np.testing.assert_array_equal(sol.get_results()["costs"].view(np.uint32), good["costs"].view(np.uint32))
    sol.close()
