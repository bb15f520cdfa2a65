"""Independent numpy restatements and size-independent properties of the CPU oracle's
"parity unpinned" parts (rollout bookkeeping, weighting, reduction, smoothing, sliding).
Written from the reference source separately from oracle/mppi_oracle.c; float32 throughout."""
import numpy as np
import pytest

from autorally_amd import params as P
from autorally_amd import synthetic as S
from oracle import oracle as O
from tests.helpers import noise_for, warm_U

f32 = np.float32


def np_weights(costs, gamma):
    # mppi_controller.cu:627-652 with sequential fp32 sums
    beta = costs.min()
    w = np.exp((-f32(gamma)) * (costs - beta)).astype(f32)
    eta = f32(0)
    for x in w:
        eta = f32(eta + x)
    tc = f32(0)
    for x in w:
        tc = f32(tc + f32(f32(x * x) / eta))
    return w, beta, eta, tc


def np_reduction(w, eta, V):
    # mppi_controller.cu:219-267, no-FMA order: 64-rollout partials then sequential partial sum
    K, T, _ = V.shape
    U = np.zeros((T, 2), f32)
    wn = (w / f32(eta)).astype(f32)
    for t in range(T):
        for j in range(2):
            tot = f32(0)
            for m in range(K // 64):
                acc = f32(0)
                for i in range(64):
                    k = 64 * m + i
                    acc = f32(acc + f32(wn[k] * V[k, t, j]))
                tot = f32(tot + acc)
            U[t, j] = tot
    return U


def np_savgol(U, hist):
    # mppi_controller.cu:468-499
    T = U.shape[0]
    f = (np.array([-3, 12, 17, 12, -3], f32) / f32(35.0)).astype(f32)
    X = np.concatenate([hist.reshape(2, 2), U, U[-1:], U[-1:]]).astype(f32)
    out = np.zeros_like(U)
    for i in range(T):
        for j in range(2):
            acc = f32(f[0] * X[i, j])
            for m in range(1, 5):
                acc = f32(acc + f32(f[m] * X[i + m, j]))
            out[i, j] = acc
    return out


@pytest.fixture(scope="module")
def small():
    cfg = S.make_config(128, 50, track="ring")
    return cfg, O.Oracle(cfg, fma_mode=0), O.Oracle(cfg, fma_mode=1)


def test_weights_match_numpy(small):
    cfg, o0, _ = small
    costs = (30 + 20 * np.random.RandomState(0).rand(cfg["K"])).astype(f32)
    w, beta, eta, tc = o0.weights(costs)
    w2, beta2, eta2, tc2 = np_weights(costs, cfg["gamma"])
    assert beta == beta2 and w.max() == 1.0
    np.testing.assert_allclose(w, w2, rtol=2e-7)
    np.testing.assert_allclose([eta, tc], [eta2, tc2], rtol=1e-6)


def test_reduction_and_savgol_match_numpy_bitwise(small):
    cfg, o0, _ = small
    rng = np.random.RandomState(1)
    K, T = cfg["K"], cfg["T"]
    V = rng.standard_normal((K, T, 2)).astype(f32)
    w = rng.rand(K).astype(f32)
    eta = f32(w.sum())
    U = o0.weighted_reduction(w, eta, V)
    np.testing.assert_array_equal(U, np_reduction(w, eta, V))  # same order, no FMA: bit-exact
    hist = rng.standard_normal(4).astype(f32)
    np.testing.assert_array_equal(o0.savgol(U, hist), np_savgol(U, hist))


def test_savgol_preserves_linear_sequences(small):
    # a quadratic/cubic-preserving 5-tap filter reproduces straight lines away from the padded ends
    _, o0, _ = small
    T = 40
    U = np.stack([0.01 * np.arange(T), 0.3 - 0.005 * np.arange(T)], 1).astype(f32)
    hist = np.array([U[0, 0] - 0.02, U[0, 1] + 0.01, U[0, 0] - 0.01, U[0, 1] + 0.005], f32)
    out = o0.savgol(U, hist)
    np.testing.assert_allclose(out[:-2], U[:-2], atol=2e-7)


def test_slide_control_seq(small):
    cfg, o0, _ = small
    T = 10
    U = np.arange(2 * T, dtype=f32).reshape(T, 2)
    hist = np.array([100, 101, 102, 103], f32)
    U1, h1 = o0.slide_control_seq(U, hist, [7, 8], 1)
    np.testing.assert_array_equal(h1, [102, 103, 0, 1])
    np.testing.assert_array_equal(U1[:-1], U[1:])
    np.testing.assert_array_equal(U1[-1], [7, 8])
    U2, h2 = o0.slide_control_seq(U, hist, [7, 8], 2)  # stride 2: t = 0, hist = U[0:4] flat (Q15)
    np.testing.assert_array_equal(h2, [0, 1, 2, 3])
    np.testing.assert_array_equal(U2[:-2], U[2:])
    np.testing.assert_array_equal(U2[-2:], [[7, 8], [7, 8]])


def test_rollout_bookkeeping_q3_q4(small):
    cfg, _, o1 = small
    K, T = cfg["K"], cfg["T"]
    eps = noise_for(cfg)[0]
    U = warm_U(cfg)
    for opt_stride in (1, 3):
        cfg2 = dict(cfg, opt_stride=opt_stride)
        o = O.Oracle(cfg2, fma_mode=1)
        costs, V, crash = o.rollouts(cfg["start_state"], U, eps)
        nu = np.array(cfg["nu"], f32)
        # rollout 0 and the first opt_stride steps are noise free (Q4)
        np.testing.assert_array_equal(V[0], U)
        np.testing.assert_array_equal(V[:, :opt_stride], np.broadcast_to(U[:opt_stride], (K, opt_stride, 2)))
        k99 = int(np.ceil(0.99 * K))
        assert k99 == 127
        # ordinary rollouts: U + eps*nu, stored unclamped (Q3); last 1 %: pure noise (Q17)
        np.testing.assert_array_equal(V[1:k99, opt_stride:], (U[None, opt_stride:] + eps[1:k99, opt_stride:] * nu).astype(f32))
        np.testing.assert_array_equal(V[k99:, opt_stride:], (eps[k99:, opt_stride:] * nu).astype(f32))
        assert np.abs(V).max() > max(cfg["u_hi"])  # some stored controls exceed the clamp range


def test_cost_of_noise_free_rollout_equals_manual_replay(small):
    """Rollout 0 = replay of U through update_state + compute_cost with the running mean of Q5."""
    cfg, _, o1 = small
    eps = np.zeros((cfg["K"], cfg["T"], 2), f32)
    U = warm_U(cfg)
    costs, _, _ = o1.rollouts(cfg["start_state"], U, eps)
    s = cfg["start_state"].astype(f32).copy()
    J = f32(0)
    crash = 0
    for t in range(cfg["T"]):
        u = np.clip(U[t], cfg["u_lo"], cfg["u_hi"]).astype(f32)
        if t > 0:
            c, crash = o1.compute_cost(s, u, np.zeros(2, f32), crash)
            J = f32(np.float64(J) + np.float64(f32(f32(c) - J)) / np.float64(t))
        s, _ = o1.update_state(s, U[t])
        if abs(float(s[3])) > 1.57:
            crash = 1
    assert costs[0] == J
    # with zero noise every ordinary rollout equals rollout 0; the last 1 % use u = 0
    assert np.all(costs[:126] == costs[0])


def test_cost_terms(small):
    cfg, _, o1 = small
    c = cfg["cost"]
    s = np.array([10.0, 0.0, np.pi / 2, 0.0, 6.0, 0.6, 0.1], f32)  # on the ring centre-line
    cost, crash = o1.compute_cost(s, [0.1, 0.2], [0.0, 0.0])
    speed = c["speed_coeff"] * (6.0 - c["desired_speed"]) ** 2
    slip = np.arctan(0.6 / 6.0)
    stab = c["slip_penalty"] * slip ** 2
    m = cfg["map_rgba"]
    # front/back points (10, +-0.5): texel columns/rows by the documented transform
    def tex(x, y):
        i = min(max(int(np.floor((x + 20) / 40 * m.shape[1])), 0), m.shape[1] - 1)
        j = min(max(int(np.floor((y + 20) / 40 * m.shape[0])), 0), m.shape[0] - 1)
        return m[j, i, 0]
    track = c["track_coeff"] * (abs(tex(10.0, 0.5)) + abs(tex(10.0, -0.5))) / 2
    assert crash == 0
    assert abs(cost - (speed + track + stab)) < 1e-3 * cost
    # off the map: clamped addressing, boundary crossed -> sticky crash and discounted crash cost
    s_out = np.array([100.0, 100.0, 0.0, 0.0, 6.0, 0.0, 0.0], f32)
    cost2, crash2 = o1.compute_cost(s_out, [0, 0], [0, 0])
    assert crash2 == 1 and cost2 > (1 - c["discount"]) * c["crash_coeff"]
    # slip-angle kill
    s_slip = np.array([10.0, 0.0, np.pi / 2, 0.0, 1.0, 5.0, 0.0], f32)
    cost3, _ = o1.compute_cost(s_slip, [0, 0], [0, 0])
    assert cost3 > c["crash_coeff"]
    # control cost with zero nu -> NaN -> 1e12 cap (Q12)
    cfgz = dict(cfg, nu=(0.0, 0.3))
    oz = O.Oracle(cfgz, fma_mode=1)
    cost4, _ = oz.compute_cost(s, [0.1, 0.2], [0.0, 0.0])
    assert cost4 == np.float32(1e12)


def test_fma_and_nofma_solves_agree(small):
    cfg, o0, o1 = small
    eps = noise_for(cfg)
    U0 = warm_U(cfg)
    a = o0.compute_control(cfg["start_state"], U0, np.zeros(4, f32), eps)
    b = o1.compute_control(cfg["start_state"], U0, np.zeros(4, f32), eps)
    np.testing.assert_array_equal(a["V"], b["V"])
    assert np.max(np.abs(a["U"] - b["U"])) < 1e-4
    assert abs(a["traj_cost"] - b["traj_cost"]) < 1e-4 * abs(b["traj_cost"])


def test_full_solve_composition(small):
    cfg, _, o1 = small
    eps = noise_for(cfg)
    U0 = warm_U(cfg)
    hist = np.array([0.01, 0.2, 0.02, 0.21], f32)
    r = o1.compute_control(cfg["start_state"], U0, hist, eps)
    costs, V, _ = o1.rollouts(cfg["start_state"], U0, eps[0])
    w, _, eta, tc = o1.weights(costs)
    U = o1.savgol(o1.weighted_reduction(w, eta, V), hist)
    np.testing.assert_array_equal(r["costs"], costs)
    np.testing.assert_array_equal(r["U"], U)
    assert r["traj_cost"] == tc
    # weights are a softmin: the normalised weights sum to 1 and the mean control stays in the hull
    assert abs((w / eta).sum() - 1) < 1e-5
    assert V[..., 0].min() - 1e-6 <= r["U"][:, 0].min() and r["U"][:, 0].max() <= V[..., 0].max() + 1e-6
