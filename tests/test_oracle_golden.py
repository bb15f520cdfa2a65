"""Pins the CPU oracle's NN dynamics step against outputs of the reference's own
Python restatement (tests/golden/gen_golden.py, scripts/ml_pipeline/utils.py) and its
RNG against L'Ecuyer's published MRG32k3a constants.  "trained_writer_6_16_24_4" is a model the
reference's training pipeline WROTE (torch_model_to_npz, tests/golden/gen_model_writer_golden.py),
with a layer list none of the shipped files has."""
import os

import numpy as np
import pytest

from autorally_amd import params as P
from autorally_amd import synthetic as S
from oracle import oracle as O
from tests.helpers import load_nn_golden

MODELS = ["autorally_nnet_09_12_2018", "gazebo_nnet_09_12_2018", "shallow_network_08_20_2020",
          "wider_deeper_network_08_20_2020", "trained_writer_6_16_24_4"]


def _oracle_for(golden_dir, name, g, fma_mode):
    layers, theta = P.load_model_npz(os.path.join(golden_dir, "models", name + ".npz"))
    assert layers == list(g[name + "/layers"])
    cfg = S.make_config(64, 10, layers=layers, theta=theta,
                        negate_yaw_der=bool(g[name + "/negate_yaw_der"][0]))
    return O.Oracle(cfg, fma_mode=fma_mode)


def test_package_default_models_equal_the_golden_fixtures(golden_dir):
    """autorally_amd/data/models (the package's default dynamics; the package does not reach into tests/) holds
    the same bytes as the fixtures gen_golden.py copied from the reference's params/models."""
    for f in sorted(os.listdir(S.MODELS_DIR)):
        with open(os.path.join(S.MODELS_DIR, f), "rb") as a, open(os.path.join(golden_dir, "models", f), "rb") as b:
            assert a.read() == b.read(), f
    assert "autorally_nnet_09_12_2018.npz" in os.listdir(S.MODELS_DIR)


def test_num_params(golden_dir):
    # NUM_PARAMS = 1412 for 6-32-32-4, 4868 for 6-64-64-4
    for layers, n in (([6, 32, 32, 4], 1412), ([6, 64, 64, 4], 4868)):
        _, theta = P.synthetic_model(layers)
        assert theta.size == n
    layers, theta = P.load_model_npz(os.path.join(golden_dir, "models", MODELS[0] + ".npz"))
    assert layers == [6, 32, 32, 4] and theta.size == 1412


def test_sample_from_survey(golden_dir):
    g = load_nn_golden(golden_dir)
    orc = _oracle_for(golden_dir, MODELS[0], g, 1)
    out = orc.nn_forward(g["sample_in"])
    np.testing.assert_allclose(out, g["sample_out"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(g["sample_out"], [-0.15874888, 5.03792211, -0.46054682, -0.24567117],
                               atol=1e-8)


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("fma_mode", [0, 1])
def test_state_deriv_matches_reference_python(golden_dir, name, fma_mode):
    g = load_nn_golden(golden_dir)
    orc = _oracle_for(golden_dir, name, g, fma_mode)
    states, ctrls, ders = g[name + "/states"], g[name + "/controls"], g[name + "/state_ders"]
    worst = 0.0
    for s, u, d in zip(states, ctrls, ders):
        sd = orc.state_deriv(s, u)
        scale = np.maximum(1.0, np.abs(d))
        worst = max(worst, float(np.max(np.abs(sd - d) / scale)))
    # fp32 oracle vs fp64 reference: tolerance 1e-5
    assert worst < 1e-5, worst


@pytest.mark.parametrize("name", MODELS)
def test_open_loop_trajectory(golden_dir, name):
    g = load_nn_golden(golden_dir)
    orc = _oracle_for(golden_dir, name, g, 1)
    traj, tctrl = g[name + "/traj_states"], g[name + "/traj_controls"]
    s = traj[0].astype(np.float32)
    for i in range(10):
        s, _ = orc.update_state(s, tctrl[i])
        np.testing.assert_allclose(s, traj[i + 1], atol=2e-5, rtol=1e-5)


def test_fma_and_nofma_variants_agree(golden_dir):
    g = load_nn_golden(golden_dir)
    a = _oracle_for(golden_dir, MODELS[0], g, 0)
    b = _oracle_for(golden_dir, MODELS[0], g, 1)
    for s, u in zip(g[MODELS[0] + "/states"], g[MODELS[0] + "/controls"]):
        np.testing.assert_allclose(a.state_deriv(s, u), b.state_deriv(s, u), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("name", [m for m in MODELS if "wider" not in m and "writer" not in m])
def test_tree_mode_of_the_output_layer(golden_dir, name):
    """fma_mode 2 -- the output layer in the summation order of the row-tree kernel (mppi_oracle.c: out_tree_dot) -- is a
    re-association of 32 products: pinned by the same reference-Python vectors at the same 1e-5, within the spread the
    FMA / no-FMA modes already have between themselves, and equal to a numpy statement of the butterfly."""
    g = load_nn_golden(golden_dir)
    o0, o1, o2 = (_oracle_for(golden_dir, name, g, m) for m in (0, 1, 2))
    states, ctrls, ders = g[name + "/states"], g[name + "/controls"], g[name + "/state_ders"]
    layers, theta = P.load_model_npz(os.path.join(golden_dir, "models", name + ".npz"))
    assert list(layers) == [6, 32, 32, 4]
    W3 = theta[-(4 * 32 + 4):-4].reshape(4, 32).astype(np.float32)
    b3 = theta[-4:].astype(np.float32)
    W1 = theta[:192].reshape(32, 6); b1 = theta[192:224]; W2 = theta[224:224 + 1024].reshape(32, 32); b2 = theta[1248:1280]
    worst21 = worst01 = 0.0
    differs = 0
    for s, u, d in zip(states, ctrls, ders):
        sd2, sd1, sd0 = o2.state_deriv(s, u), o1.state_deriv(s, u), o0.state_deriv(s, u)
        assert float(np.max(np.abs(sd2 - d) / np.maximum(1.0, np.abs(d)))) < 1e-5
        np.testing.assert_array_equal(sd2[:3], sd1[:3])  # kinematics untouched
        worst21 = max(worst21, float(np.max(np.abs(sd2 - sd1))))
        worst01 = max(worst01, float(np.max(np.abs(sd0 - sd1))))
        differs += int(np.any(sd2 != sd1))
        # numpy statement of the tree on the mode-1 hidden activations (float32 products are exact in float64, one rounding
        # per operation): lane p owns activations 2p, 2p+1
        a = np.array([s[3], s[4], s[5], s[6], u[0], u[1]], np.float32)
        h = a
        for W, b in ((W1, b1), (W2, b2)):
            z = np.zeros(W.shape[0], np.float32)
            for k in range(W.shape[1]):
                z = (W[:, k].astype(np.float64) * np.float64(h[k]) + z.astype(np.float64)).astype(np.float32)  # fmaf
            h = np.tanh((z + b.astype(np.float32)).astype(np.float32)).astype(np.float32)
        out = np.zeros(4, np.float32)
        for j in range(4):
            Pp = np.zeros(16, np.float32)
            for q in range(16):
                m = np.float32(W3[j, 2 * q] * h[2 * q])
                Pp[q] = np.float32(np.float64(W3[j, 2 * q + 1]) * np.float64(h[2 * q + 1]) + np.float64(m))
            L1 = np.array([Pp[q] + Pp[q ^ 8] for q in range(16)], np.float32)
            L2 = np.array([L1[q] + L1[q ^ 7] for q in range(16)], np.float32)
            L3 = np.array([L2[q] + L2[q ^ 1] for q in range(16)], np.float32)
            out[j] = np.float32(np.float32(L3[0] + L3[2]) + b3[j])
            # every lane of the kernel's row holds the same bits
            assert all(np.float32(L3[q] + L3[q ^ 2]) == np.float32(L3[0] + L3[2]) for q in range(16))
        # (numpy's tanh and libm's tanhf may differ in the last bit of an activation: compare at 2e-6, the order itself
        # is pinned bit for bit on the GPU, tests/test_row_tree_gpu.py)
        np.testing.assert_allclose(out, sd2[3:], atol=2e-6, rtol=1e-6)
    assert differs > 0          # the mode does change bits ...
    assert worst21 <= max(2 * worst01, 4e-6), (worst21, worst01)  # ... by no more than FMA contraction does


@pytest.mark.parametrize("name", [m for m in MODELS if "wider" in m])
def test_split_hidden_mode_of_the_64_wide_net(golden_dir, name):
    """fma_mode 5 -- the 4x4x1-MFMA kernel's automatic form: every 64-input hidden layer as two accumulation chains (even / odd
    k) added at the end, the output layer as in mode 3 -- is pinned by the same reference-Python vectors of the shipped
    6-64-64-64-64-4 model at the same 1e-5, stays within the spread the FMA / no-FMA modes have between themselves, leaves the
    kinematics alone and within 2e-5 of the one-chain mode 3 (the order itself is pinned bit for bit on the GPU,
    tests/test_m44_gpu.py)."""
    g = load_nn_golden(golden_dir)
    o0, o1, o3, o5 = (_oracle_for(golden_dir, name, g, m) for m in (0, 1, 3, 5))
    states, ctrls, ders = g[name + "/states"], g[name + "/controls"], g[name + "/state_ders"]
    layers, theta = P.load_model_npz(os.path.join(golden_dir, "models", name + ".npz"))
    assert list(layers) == [6, 64, 64, 64, 64, 4]
    worst51 = worst01 = 0.0
    differs = 0
    for s, u, d in zip(states, ctrls, ders):
        sd5, sd3, sd1, sd0 = o5.state_deriv(s, u), o3.state_deriv(s, u), o1.state_deriv(s, u), o0.state_deriv(s, u)
        assert float(np.max(np.abs(sd5 - d) / np.maximum(1.0, np.abs(d)))) < 1e-5
        np.testing.assert_array_equal(sd5[:3], sd1[:3])  # kinematics untouched
        worst51 = max(worst51, float(np.max(np.abs(sd5 - sd1))))
        worst01 = max(worst01, float(np.max(np.abs(sd0 - sd1))))
        differs += int(np.any(sd5 != sd3))
        # against the one-chain form of the same kernel (mode 3): a re-association of 3 x 64 x 64 products per step
        out3 = o3.state_deriv(s, u)[3:]
        assert float(np.max(np.abs(sd5[3:] - out3))) < 2e-5
    assert differs > 0
    assert worst51 <= max(2 * worst01, 8e-6), (worst51, worst01)


# ---------------- MRG32k3a known answers ----------------
# L'Ecuyer, Simard, Chen, Kelton, "An object-oriented random-number package with many long
# streams and substreams" (RngStreams): A1p76, A2p76, A1p127, A2p127.
A1P76 = [[82758667, 1871391091, 4127413238], [3672831523, 69195019, 1871391091],
         [3672091415, 3528743235, 69195019]]
A2P76 = [[1511326704, 3759209742, 1610795712], [4292754251, 1511326704, 3889917532],
         [3859662829, 4292754251, 3708466080]]
A1P127 = [[2427906178, 3580155704, 949770784], [226153695, 1230515664, 3580155704],
          [1988835001, 986791581, 1230515664]]
A2P127 = [[1464411153, 277697599, 1610723613], [32183930, 1464411153, 1022607788],
          [2824425944, 32183930, 2093834863]]


def test_mrg32k3a_published_jump_matrices():
    a1, a2 = O.jump_matrices(76)
    assert a1.tolist() == A1P76 and a2.tolist() == A2P76
    a1, a2 = O.jump_matrices(127)
    assert a1.tolist() == A1P127 and a2.tolist() == A2P127


def test_mrg32k3a_first_outputs_and_skip():
    import ctypes as C
    L = O.lib()
    st = O.MrgState()
    L.orc_mrg_seed(C.byref(st), 0)  # seed 0 = L'Ecuyer's default state 12345 x 6
    assert list(st.s1) == [12345] * 3 and list(st.s2) == [12345] * 3
    u = [L.orc_mrg_next_u01(C.byref(st)) for _ in range(4)]
    # RngStreams' first stream starts 0.12701112, 0.31852757, 0.30918602, 0.82584686
    np.testing.assert_allclose(u, [0.1270111220, 0.3185275654, 0.3091860156, 0.8258468629], atol=5e-10)
    # skip(n) == n single steps
    a, b = O.MrgState(), O.MrgState()
    L.orc_mrg_seed(C.byref(a), 1234)
    L.orc_mrg_seed(C.byref(b), 1234)
    for _ in range(1000):
        L.orc_mrg_next_z(C.byref(a))
    L.orc_mrg_skip(C.byref(b), 1000)
    assert list(a.s1) == list(b.s1) and list(a.s2) == list(b.s2)


def test_noise_moments_and_layout():
    K, T = 256, 50
    e = O.generate_noise(1234, 0, K, T)
    assert e.shape == (K, T, 2) and np.all(np.isfinite(e))
    assert abs(e.mean()) < 0.02 and abs(e.std() - 1.0) < 0.02
    assert abs(np.mean(e ** 3)) < 0.05 and abs(np.mean(e ** 4) - 3.0) < 0.15
    # offset continues the per-rollout stream: draws [2T, 4T) of each subsequence
    e2 = O.generate_noise(1234, 2 * T, K, T)
    e_long = O.generate_noise(1234, 0, K, 2 * T)
    np.testing.assert_array_equal(e_long[:, T:, :], e2)
    np.testing.assert_array_equal(e_long[:, :T, :], e)
    # different rollouts use different subsequences
    assert not np.array_equal(e[0], e[1])


def test_mrg32k3a_long_run_against_bigint():
    """The recurrence must hold over long runs (catches 64-bit wrap in a12*s11 - a13n*s10)."""
    import ctypes as C
    L = O.lib()
    st = O.MrgState()
    L.orc_mrg_seed(C.byref(st), 1234)
    s1, s2 = [int(x) for x in st.s1], [int(x) for x in st.s2]
    m1, m2 = 4294967087, 4294944443
    n = 300000
    zs = np.array([L.orc_mrg_next_z(C.byref(st)) for _ in range(n)], dtype=np.uint64)
    ref = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        p1 = (1403580 * s1[1] - 810728 * s1[0]) % m1
        s1 = [s1[1], s1[2], p1]
        p2 = (527612 * s2[2] - 1370589 * s2[0]) % m2
        s2 = [s2[1], s2[2], p2]
        z = (p1 - p2) % m1
        ref[i] = z if z > 0 else m1
    np.testing.assert_array_equal(zs, ref)


def test_costmap_file_written_by_reference_track_converter(golden_dir):
    """tests/golden/costmap_track_converter.npz was written by the reference's own
    scripts/track_converter.py:gen_costmap (tests/golden/gen_costmap_golden.py): the loader restating
    MPPICosts::loadTrackData (costs.cu:190-232) must read back the values of the input text, row-major
    with x fastest, and derive the transform of :224-229."""
    import os
    from autorally_amd import params as P
    txt = open(os.path.join(golden_dir, "costmap_input.txt")).read().split(" ")
    x0, x1, y0, y1, ppm = [float(v) for v in txt[:5]]
    vals = np.array(txt[5:-1], dtype=np.float32)
    m, r_c1, r_c2, trs = P.load_costmap_npz(os.path.join(golden_dir, "costmap_track_converter.npz"))
    W, H = int((x1 - x0) * ppm), int((y1 - y0) * ppm)
    assert m.shape == (H, W, 4) and vals.size == W * H
    np.testing.assert_array_equal(m[:, :, 0].reshape(-1), vals)
    assert not m[:, :, 1:].any()
    np.testing.assert_allclose(r_c1, [1.0 / (x1 - x0), 0, 0], rtol=1e-7)
    np.testing.assert_allclose(r_c2, [0, 1.0 / (y1 - y0), 0], rtol=1e-7)
    np.testing.assert_allclose(trs, [-x0 / (x1 - x0), -y0 / (y1 - y0), 1], rtol=1e-7)
    # and the restated texture lookup hits the texel the text file holds at that position
    from oracle import oracle as O
    from autorally_amd import synthetic as S
    cfg = S.make_config(64, 4)
    cfg = dict(cfg, map_rgba=m, r_c1=r_c1, r_c2=r_c2, trs=trs)
    orc = O.Oracle(cfg)
    for (px, py) in [(-2.9, -1.9), (1.0, 0.25), (4.9, 2.4), (0.3, -0.6)]:
        s = np.array([px, py, 0, 0, 0, 0, 0], np.float32)
        # only the track term left: cost = (|front texel| + |back texel|) / 2 (costs.cu:359-393)
        c = dict(cfg["cost"], track_coeff=1.0, speed_coeff=0.0, crash_coeff=0.0, slip_penalty=0.0,
                 boundary_threshold=1e9, desired_speed=0.0)
        o2 = O.Oracle(dict(cfg, cost=c))
        # front/back points are +-0.5 m along the heading: use the mean of the two texels
        f = vals[min(H - 1, max(0, int((py - y0) * ppm))) * W + min(W - 1, max(0, int((px + 0.5 - x0) * ppm)))]
        b = vals[min(H - 1, max(0, int((py - y0) * ppm))) * W + min(W - 1, max(0, int((px - 0.5 - x0) * ppm)))]
        got = o2.compute_cost(s, np.zeros(2, np.float32), np.zeros(2, np.float32))[0]
        assert abs(got - (abs(f) + abs(b)) / 2) < 1e-6, (px, py, got, f, b)


def test_costmap_file_written_by_reference_track_generator(golden_dir):
    """tests/golden/costmap_track_generator.npz was written by the reference's image converter
    (scripts/track_generator.py:gen_costmap, run by tests/golden/gen_costmap_image_golden.py) from the
    committed costmap_image.png + costmap_image_config.txt, with all four channels populated.  The
    loader restating MPPICosts::loadTrackData (costs.cu:190-232) must put channel c of the file in
    component c of texel (row, column) -- checked against the image's own pixels."""
    import ast
    import os
    from PIL import Image
    from autorally_amd import params as P
    cfg = ast.literal_eval(open(os.path.join(golden_dir, "costmap_image_config.txt")).read())
    m, r_c1, r_c2, trs = P.load_costmap_npz(os.path.join(golden_dir, "costmap_track_generator.npz"))
    x0, x1 = cfg["xBounds"]
    y0, y1 = cfg["yBounds"]
    W, H = int((x1 - x0) * cfg["pixelsPerMeter"]), int((y1 - y0) * cfg["pixelsPerMeter"])
    assert m.shape == (H, W, 4)
    px = np.array(Image.open(os.path.join(golden_dir, "costmap_image.png")), dtype=np.float32)
    assert px.shape == (H, W, 4) and cfg["imageRotation"] == 180 and cfg["flip"]
    px = px[::-1, ::-1]  # the 180 degree rotation
    off = [cfg[k + "Offset"] for k in "rgba"]
    nrm = [cfg[k + "Normalizer"] for k in "rgba"]
    for i in range(4):  # image channel i lands in costmap channel channelMap[i], flipped vertically
        want = ((px[:, :, i] + np.float32(off[i])) / np.float32(nrm[i]))[::-1]
        np.testing.assert_allclose(m[:, :, cfg["channelMap"][i]], want, rtol=1e-6, atol=0)
    np.testing.assert_allclose(r_c1, [1.0 / (x1 - x0), 0, 0], rtol=1e-7)
    np.testing.assert_allclose(trs, [-x0 / (x1 - x0), -y0 / (y1 - y0), 1], rtol=1e-7)
    # the restated lookup returns component 0 of that texture (costs.cu:359-393 use .x only)
    from oracle import oracle as O
    from autorally_amd import synthetic as S
    base = S.make_config(64, 4)
    c = dict(base["cost"], track_coeff=1.0, speed_coeff=0.0, crash_coeff=0.0, slip_penalty=0.0,
             boundary_threshold=1e9, desired_speed=0.0)
    orc = O.Oracle(dict(base, map_rgba=m, r_c1=r_c1, r_c2=r_c2, trs=trs, cost=c))
    ppm = cfg["pixelsPerMeter"]
    for (qx, qy) in [(-1.3, -0.4), (0.6, 0.9), (2.1, 1.7)]:
        s = np.array([qx, qy, 0, 0, 0, 0, 0], np.float32)
        row = min(H - 1, max(0, int((qy - y0) * ppm)))
        f = m[row, min(W - 1, max(0, int((qx + 0.5 - x0) * ppm))), 0]
        b = m[row, min(W - 1, max(0, int((qx - 0.5 - x0) * ppm))), 0]
        got = orc.compute_cost(s, np.zeros(2, np.float32), np.zeros(2, np.float32))[0]
        assert abs(got - (abs(f) + abs(b)) / 2) < 1e-5 * max(1.0, abs(got)), (qx, qy, got, f, b)


# ------------------------------------------------------------------ third-party known answers (scipy)
def test_savgol_is_the_published_quadratic_5_tap_filter():
    """mppi_controller.cu:468-499 hard-codes [-3, 12, 17, 12, -3] / 35: the Savitzky-Golay smoothing filter of
    window 5 and polynomial order 2 (or 3).  Pinned against scipy's derivation of the same filter: the
    coefficients, and the oracle's output away from the padded ends against scipy.signal.savgol_filter."""
    from scipy.signal import savgol_coeffs, savgol_filter
    np.testing.assert_allclose(savgol_coeffs(5, 2), np.array([-3, 12, 17, 12, -3]) / 35.0, atol=1e-15)
    np.testing.assert_allclose(savgol_coeffs(5, 3), np.array([-3, 12, 17, 12, -3]) / 35.0, atol=1e-15)
    cfg = S.make_config(64, 60, track="ring")
    orc = O.Oracle(cfg)
    rng = np.random.RandomState(3)
    U = np.cumsum(rng.standard_normal((60, 2)) * 0.05, axis=0).astype(np.float32)
    hist = rng.standard_normal(4).astype(np.float32) * 0.1
    got = orc.savgol(U, hist)
    ref = savgol_filter(U.astype(np.float64), 5, 2, axis=0)
    np.testing.assert_allclose(got[2:-2], ref[2:-2], atol=5e-7)
    # the ends follow the reference's padding: [hist0, hist1, U..., U_last, U_last] (:476-489)
    X = np.concatenate([hist.reshape(2, 2), U, U[-1:], U[-1:]]).astype(np.float64)
    c = np.array([-3, 12, 17, 12, -3]) / 35.0
    full = np.stack([np.convolve(X[:, j], c[::-1], mode="valid") for j in range(2)], 1)
    np.testing.assert_allclose(got, full, atol=5e-7)


def test_weights_are_a_softmax_and_trajectory_cost_follows():
    """normExpKernel + the host normaliser (mppi_controller.cu:193-203, 627-652): w / eta is the softmax of
    -gamma * J (scipy.special.softmax), and trajectory_cost_ is sum w^2 / eta (Q8)."""
    from scipy.special import softmax
    cfg = S.make_config(256, 20, track="ring")
    orc = O.Oracle(cfg)
    rng = np.random.RandomState(11)
    J = (40.0 + 30.0 * rng.rand(256)).astype(np.float32)
    J[7] = 1e12  # a capped rollout weighs nothing
    w, beta, eta, tc = orc.weights(J)
    p = softmax(-cfg["gamma"] * J.astype(np.float64))
    np.testing.assert_allclose(w / eta, p, rtol=2e-6, atol=1e-12)
    assert beta == J.min() and w[np.argmin(J)] == 1.0 and w[7] == 0.0
    np.testing.assert_allclose(tc, float((p ** 2).sum() * eta), rtol=1e-5)


def test_generated_noise_is_standard_normal_and_uncorrelated():
    """The noise spec (MRG32k3a + Box-Muller written in IEEE basic operations) against scipy.stats: a
    Kolmogorov-Smirnov test on 200 000 draws, the two members of a Box-Muller pair uncorrelated, successive
    timesteps and neighbouring rollouts (subsequences 2^76 apart) uncorrelated."""
    from scipy import stats
    K, T = 1000 // 64 * 64 + 64, 100
    e = O.generate_noise(4321, 0, K, T).astype(np.float64)
    flat = e.reshape(-1)
    assert flat.size >= 200000
    assert stats.kstest(flat, "norm").pvalue > 1e-3
    assert stats.kstest(e[..., 0].reshape(-1), "norm").pvalue > 1e-3 and stats.kstest(e[..., 1].reshape(-1), "norm").pvalue > 1e-3
    n = e[..., 0].size
    bound = 5.0 / np.sqrt(n)
    assert abs(np.corrcoef(e[..., 0].reshape(-1), e[..., 1].reshape(-1))[0, 1]) < bound
    assert abs(np.corrcoef(e[:, :-1, 0].reshape(-1), e[:, 1:, 0].reshape(-1))[0, 1]) < bound
    assert abs(np.corrcoef(e[:-1, :, 0].reshape(-1), e[1:, :, 0].reshape(-1))[0, 1]) < bound
    # tails exist and are finite: |z| up to ~4.9 expected in 2e5 draws, none absurd
    assert 3.5 < np.abs(flat).max() < 6.5
