"""Second dynamics family: GeneralizedLinear<CarBasisFuncs,7,2,25,CarKinematics,3>
(generalized_linear.cu:169-245, car_bfs.cuh:44-120) behind the same ABI, with the shipped
basis_function_09_12_2018.npz (tests/golden/models, data copied from the reference's params/models).

The reference sums its y-threads' partial products with atomicAdd (no fixed order), so this family has
no bit-exact definition; oracle and kernel both take the thread-order execution and are compared with
the tolerances of the network tests."""
import os

import numpy as np
import pytest

from autorally_amd import params as P
from autorally_amd import synthetic as S
from oracle import oracle as O
from tests.helpers import noise_for, rel_err


def _bf_cfg(golden_dir, K=256, T=40, track="oval", **over):
    W = P.load_bf_npz(os.path.join(golden_dir, "models", "basis_function_09_12_2018.npz"))
    return S.make_config(K, T, track=track, bf_W=W, **over)


def _np_basis(s, u):
    """float64 numpy restatement of CarBasisFuncs::basisFuncX (car_bfs.cuh:44-120), written
    independently of the C one, for the values (not the rounding) of the 25 functions."""
    s4, s5, s6, s3 = float(s[4]), float(s[5]), float(s[6]), float(s[3])
    u0, u1 = float(u[0]), float(u[1])
    big = s4 > .1
    A = np.tan(np.arctan(s5 / s4 + .45 * s6 / s4) - u0) if big else np.tan(-u0)
    B = (s5 / s4 - .35 * s6 / s4) if big else 0.0
    su = np.sin(u0)
    return np.array([
        u1, s4 / 10.0, su * A / 1200.0, su * A * abs(A) / 1440000.0, su * A ** 3 / 1728000000.0,
        s6 * s5 / 25.0, s6 / 10.0, s5 / 10.0, su, (s5 / s4 / 40.0) if big else 0.0,
        A / 1400.0, A * abs(A) / 1960000, A ** 3 / 2744000000,
        B / 40.0 if big else 0.0, B * abs(B) / 1600.0 if big else 0.0, B ** 3 / 64000.0 if big else 0.0,
        s6 * s4 / 50.0, s3, s3 * s6, s3 * s4 / 3.0, s3 * s4 * s6 / 5.0, s4 ** 2 / 100.0, s4 ** 3 / 1000.0,
        u1 ** 2, u1 ** 3])


def _samples(n, seed=0):
    rng = np.random.RandomState(seed)
    s = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    s[:, 4] = rng.uniform(-0.5, 12.0, n)  # u_x on both sides of the 0.1 switch
    s[:, 5] = rng.uniform(-2, 2, n)
    s[:, 6] = rng.uniform(-3, 3, n)
    s[:, 3] = rng.uniform(-0.4, 0.4, n)
    s[0, 4] = np.float32(0.1)             # (double)0.1f > .1 is true
    s[1, 4] = np.nextafter(np.float32(0.1), np.float32(0))
    u = rng.uniform(-0.99, 0.99, (n, 2)).astype(np.float32)
    return s, u


def test_oracle_basis_functions_against_numpy(golden_dir):
    cfg = _bf_cfg(golden_dir)
    orc = O.Oracle(cfg)
    s, u = _samples(200)
    W = cfg["bf_W"].astype(np.float64)
    for i in range(s.shape[0]):
        phi = np.array([orc.L.orc_basis_func(j, O._fp(s[i]), O._fp(u[i])) for j in range(25)])
        ref = _np_basis(s[i], u[i])
        np.testing.assert_allclose(phi, ref, rtol=3e-5, atol=1e-9)
        sd = orc.state_deriv(s[i], u[i])
        np.testing.assert_allclose(sd[3:], W @ ref, rtol=2e-4, atol=2e-4)
        assert sd[2] == -s[i, 6]  # yaw rate always negated (generalized_linear.cu:216)
    # the two sides of the u_x switch
    assert orc.L.orc_basis_func(9, O._fp(s[0]), O._fp(u[0])) != 0.0
    assert orc.L.orc_basis_func(9, O._fp(s[1]), O._fp(u[1])) == 0.0


def test_oracle_solve_with_basis_functions(golden_dir):
    """The whole restated solve runs on the second model and drives the car forward."""
    cfg = _bf_cfg(golden_dir, K=256, T=60)
    orc = O.Oracle(cfg)
    U = np.tile(np.array([0.0, 0.3], np.float32), (cfg["T"], 1))
    r = orc.compute_control(cfg["start_state"], U, np.zeros(4, np.float32), noise_for(cfg))
    assert np.all(np.isfinite(r["U"])) and np.all(np.isfinite(r["costs"]))
    ss, cs = orc.nominal_traj(cfg["start_state"], r["U"])
    assert ss[-1, 0] > ss[0, 0] + 1.0


@pytest.mark.gpu
def test_gpu_basis_dynamics_match_oracle(golden_dir):
    from autorally_amd import capi
    cfg = _bf_cfg(golden_dir)
    orc = O.Oracle(cfg)
    sol = capi.Solver(cfg)
    assert sol.rollout_variant() == "basis_funcs25_valu_3w"
    s, u = _samples(512, seed=3)
    got = sol.debug_dynamics(s, u)
    ref = np.stack([orc.state_deriv(s[i], u[i]) for i in range(s.shape[0])])
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=2e-6)
    scale = np.abs(ref[:, 3:]).max(axis=0)
    assert np.all(np.abs(got[:, 3:] - ref[:, 3:]).max(axis=0) <= 2e-5 * scale)
    with pytest.raises(capi.MppiError):
        sol._ck(sol.L.mppi_set_nn_params(sol.h, O._fp(np.zeros(4, np.float32)), 4))
    sol.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K,T,track", [(256, 40, "oval"), (2560, 100, "oval"), (512, 60, "ring")])
def test_gpu_basis_solve_matches_oracle(golden_dir, K, T, track):
    """BASELINE-style parity of a whole solve on the second model (K = 2560 is the reference's
    MPPI_NUM_ROLLOUTS__ for this build, path_integral_main.cu:71)."""
    from autorally_amd import capi
    cfg = _bf_cfg(golden_dir, K=K, T=T, track=track)
    orc = O.Oracle(cfg, nthreads=8)
    eps = noise_for(cfg)
    U0 = np.tile(np.array([0.0, 0.3], np.float32), (T, 1))
    ref = orc.compute_control(cfg["start_state"], U0, np.zeros(4, np.float32), eps)
    sol = capi.Solver(cfg)
    sol.set_control_seq(U0)
    sol.set_noise(eps)
    sol.compute_control(cfg["start_state"])
    got = dict(sol.get_results(), V=sol.get_applied_controls())
    np.testing.assert_array_equal(got["V"].view(np.uint32), ref["V"][-1].view(np.uint32))
    err = rel_err(got["costs"], ref["costs"])
    assert int(np.sum(err > 1e-4)) <= max(1, K // 100), float(err.max())
    assert float(np.percentile(err, 95)) < 2e-5
    assert np.max(np.abs(got["U"] - ref["U"])) <= 1e-4
    assert abs(got["traj_cost"] - ref["traj_cost"]) <= 1e-4 * abs(ref["traj_cost"])
    # nominal trajectory (host replay) and feedback gains (numerical Jacobian, ddp_dynamics.h:71-84)
    x0 = np.array(cfg["start_state"], np.float32)
    ss, cs = sol.nominal_traj(x0)
    rs, rc = orc.nominal_traj(x0, got["U"])
    np.testing.assert_allclose(ss, rs, rtol=0, atol=5e-5)
    np.testing.assert_allclose(cs, rc, rtol=0, atol=1e-6)
    g = sol.compute_feedback_gains(x0, rs, rc)
    r = orc.ddp_feedback_gains(x0, rs, rc)
    scale = np.abs(r["feedback"]).max()
    assert scale > 1e-3 and np.max(np.abs(g["feedback"] - r["feedback"])) <= 2e-2 * scale
    # the one- and two-wave forms of the kernel do the same arithmetic in the same order as the three-wave form
    assert sol.rollout_variant() == "basis_funcs25_valu_3w"
    for v, name in (("fused", "basis_funcs25_valu"), ("quad", "basis_funcs25_valu_2w")):
        sol.set_rollout_variant(v)
        assert sol.rollout_variant() == name
        sol.set_control_seq(U0)
        sol.set_noise(eps)
        sol.compute_control(cfg["start_state"])
        one = sol.get_results()
        np.testing.assert_array_equal(one["costs"].view(np.uint32), got["costs"].view(np.uint32))
        np.testing.assert_array_equal(one["U"].view(np.uint32), got["U"].view(np.uint32))
    # generator mode: the control wave's in-kernel draws (three-wave form) equal the stand-alone generator
    # kernel's (the other forms), solve after solve; and a re-seeded handle repeats itself
    seq = {}
    for v in ("auto", "quad", "fused"):
        sol.set_rollout_variant(v)
        sol.seed(5, 0)
        sol.reset_controls()
        sol.set_control_seq(U0)
        outs = []
        for it in range(3):
            sol.compute_control(x0)
            outs.append(sol.get_results()["U"].copy())
        seq[v] = outs
    for v in ("quad", "fused"):
        for a, b in zip(seq["auto"], seq[v]):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    sol.close()
