"""Launch-file defaults and parameter packing for the MPPI hot path (host-side plumbing).

Reference: autorally_control/launch/path_integral_nn.launch:22-72 (values),
neural_net_model.cu:73-106 (npz keys dynamics_W{i}/dynamics_b{i}, float64, (out,in)),
neural_net_model.cu:120-141 (packed layout [W1|b1|W2|b2|...]),
costs.cu:190-232 (costmap npz keys + transform), path_integral_main.cu:98-116.
"""
import numpy as np

# path_integral_nn.launch:51-62
DEFAULT_COST = dict(
    desired_speed=8.0, speed_coeff=4.25, track_coeff=200.0, max_slip_ang=1.25,
    slip_penalty=10.0, track_slop=0.0, crash_coeff=10000.0, steering_coeff=0.0,
    throttle_coeff=0.0, boundary_threshold=0.65, discount=0.1, l1_cost=False,
)

# path_integral_nn.launch:22-48, path_integral_main.cu:98
DEFAULT_CTRL = dict(
    hz=50, opt_stride=1, gamma=0.15, num_iters=1,
    nu=(0.275, 0.3), init_u=(0.0, 0.0),
    u_lo=(-0.99, -0.99), u_hi=(0.99, 0.65), negate_yaw_der=True,
)


def pack_theta(weights, biases):
    """[W1|b1|W2|b2|...] row-major (out,in) float32 -- neural_net_model.cu:120-141."""
    parts = []
    for W, b in zip(weights, biases):
        W = np.asarray(W, dtype=np.float64).astype(np.float32)  # (float)weight_i[...] :93
        b = np.asarray(b, dtype=np.float64).astype(np.float32)
        assert W.ndim == 2 and b.shape == (W.shape[0],)
        parts.append(W.reshape(-1))
        parts.append(b)
    return np.concatenate(parts).astype(np.float32)


def load_model_npz(path):
    """Returns (layers, theta_packed). Keys dynamics_W1.., dynamics_b1.. (neural_net_model.cu:84-99)."""
    z = np.load(path)
    n = len([k for k in z.files if k.startswith("dynamics_W")])
    Ws = [z["dynamics_W%d" % i] for i in range(1, n + 1)]
    bs = [z["dynamics_b%d" % i].reshape(-1) for i in range(1, n + 1)]
    layers = [Ws[0].shape[1]] + [W.shape[0] for W in Ws]
    return layers, pack_theta(Ws, bs)


def synthetic_model(layers, seed=4):
    """U(-1/sqrt(in), 1/sqrt(in)) weights and biases (cfg 4: no 6-64-64-4 file exists)."""
    rng = np.random.RandomState(seed)
    Ws, bs = [], []
    for nin, nout in zip(layers[:-1], layers[1:]):
        lim = 1.0 / np.sqrt(nin)
        Ws.append(rng.uniform(-lim, lim, size=(nout, nin)))
        bs.append(rng.uniform(-lim, lim, size=(nout,)))
    return list(layers), pack_theta(Ws, bs)


def costmap_transform(x_min, x_max, y_min, y_max):
    """R columns and trs exactly as costs.cu:224-229 / :175-188 build them (float32)."""
    f = np.float32
    x_min, x_max, y_min, y_max = f(x_min), f(x_max), f(y_min), f(y_max)
    r_c1 = np.array([f(1.0 / (x_max - x_min)), 0.0, 0.0], dtype=np.float32)
    r_c2 = np.array([0.0, f(1.0 / (y_max - y_min)), 0.0], dtype=np.float32)
    trs = np.array([f(-x_min / (x_max - x_min)), f(-y_min / (y_max - y_min)), 1.0], dtype=np.float32)
    return r_c1, r_c2, trs


def load_costmap_npz(path):
    """Returns (map_rgba[H,W,4] f32, r_c1, r_c2, trs) -- costs.cu:190-232."""
    z = np.load(path)
    xb = z["xBounds"].astype(np.float32).reshape(-1)
    yb = z["yBounds"].astype(np.float32).reshape(-1)
    ppm = np.float32(z["pixelsPerMeter"].reshape(-1)[0])
    W = int((xb[1] - xb[0]) * ppm)
    H = int((yb[1] - yb[0]) * ppm)
    m = np.zeros((H, W, 4), dtype=np.float32)
    for c in range(4):
        m[:, :, c] = z["channel%d" % c].astype(np.float32).reshape(-1)[: W * H].reshape(H, W)
    r_c1, r_c2, trs = costmap_transform(xb[0], xb[1], yb[0], yb[1])
    return m, r_c1, r_c2, trs


def save_costmap_npz(path, channel0, x_bounds, y_bounds, ppm):
    """Writer in the documented format (scripts/track_generator.py:34-42)."""
    ch0 = np.asarray(channel0, dtype=np.float32)
    z = np.zeros(ch0.size, dtype=np.float32)
    np.savez(path, xBounds=np.array(x_bounds, dtype=np.float32),
             yBounds=np.array(y_bounds, dtype=np.float32),
             pixelsPerMeter=np.array([ppm], dtype=np.float32),
             channel0=ch0.reshape(-1), channel1=z, channel2=z, channel3=z)


def load_bf_npz(path):
    """GeneralizedLinear::loadParams (generalized_linear.cu:95-110): key "W", (4, 25) <f8 -> float32."""
    z = np.load(path)
    W = np.asarray(z["W"], dtype=np.float64)
    if W.shape != (4, 25):
        raise ValueError("W must be (4, 25), got %r" % (W.shape,))
    return W.astype(np.float32)
