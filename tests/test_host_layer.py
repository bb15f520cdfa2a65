"""C++ host layer above the C ABI: cnpy-format .npz reader, launch-XML loader (CPU), and the
ROS-free path_integral_nn binary checked against the same loop driven from Python (GPU)."""
import json
import os
import subprocess

import numpy as np
import pytest

from autorally_amd import build as B
from autorally_amd import params as P
from autorally_amd import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCH = os.path.join(ROOT, "autorally_amd", "host", "launch", "path_integral_nn.launch")


@pytest.fixture(scope="module")
def bins():
    B.build()
    return dict((os.path.basename(p), p) for p in B.build_host())


def test_host_selftest(bins, golden_dir, tmp_path):
    cmap = os.path.join(golden_dir, "costmap_track_converter.npz")
    r = subprocess.run([bins["host_selftest"], os.path.join(golden_dir, "models", "autorally_nnet_09_12_2018.npz"),
                        LAUNCH, str(tmp_path), cmap], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert "host selftest OK" in r.stdout
    # the value the C++ reader printed equals numpy's view of the same file
    z = np.load(os.path.join(golden_dir, "models", "autorally_nnet_09_12_2018.npz"))
    assert "W1[0,0]=%.17g" % z["dynamics_W1"][0, 0] in r.stdout
    # and the map the C++ writer produced loads with numpy (cnpy/numpy interoperability)
    m = np.load(os.path.join(str(tmp_path), "selftest_map.npz"))
    assert m["channel0"].dtype == np.float32 and m["channel0"].shape == (96,)
    assert m["xBounds"].tolist() == [-3.0, 3.0]
    # loadTrackData on the file the reference's own track_converter.py wrote (tests/golden/gen_costmap_golden.py)
    txt = open(os.path.join(golden_dir, "costmap_input.txt")).read().split(" ")
    vals = np.array(txt[5:-1], dtype=np.float32)
    line = [l for l in r.stdout.splitlines() if l.startswith("costmap ")][0]
    assert "W=32 H=18" in line
    assert "r_c1=[0.125 0 0]" in line and "trs=[0.375 0.444444448 1]" in line  # costs.cu:224-229
    assert "ch0[0]=%.9g" % vals[0] in line and "ch0[last]=%.9g" % vals[-1] in line
    assert "sum0=%.9g" % float(np.sum(vals.astype(np.float64))) in line and "ch1max=0" in line
    # and on the four-channel file the reference's track_generator.py wrote (gen_costmap_image_golden.py)
    cmap4 = os.path.join(golden_dir, "costmap_track_generator.npz")
    r = subprocess.run([bins["host_selftest"], os.path.join(golden_dir, "models", "autorally_nnet_09_12_2018.npz"),
                        LAUNCH, str(tmp_path), cmap4], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    z = np.load(cmap4)
    line = [l for l in r.stdout.splitlines() if l.startswith("costmap4 ")][0]
    sums = [float(np.sum(z["channel%d" % c].astype(np.float64))) for c in range(4)]
    assert "sums=[%.9g %.9g %.9g %.9g]" % tuple(sums) in line
    mid = 20 * (12 // 2) + 20 // 3
    assert "texel%d=[%.9g %.9g %.9g %.9g]" % ((mid,) + tuple(float(z["channel%d" % c][mid]) for c in range(4))) in line
    assert "W=20 H=12" in r.stdout


def _params_dir(tmp_path, golden_dir, model_file, map_file):
    d = os.path.join(str(tmp_path), "params")
    os.makedirs(os.path.join(d, "models"))
    os.makedirs(os.path.join(d, "maps"))
    src = os.path.join(golden_dir, "models", model_file)
    with open(src, "rb") as f, open(os.path.join(d, "models", model_file), "wb") as g:
        g.write(f.read())
    ch0, xb, yb, ppm = S.oval_track_map()
    P.save_costmap_npz(os.path.join(d, "maps", map_file), ch0, xb, yb, ppm)
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("binary,launch,model_file,map_file", [
    ("path_integral_nn", "path_integral_nn.launch", "autorally_nnet_09_12_2018.npz", "ccrf_costmap_09_29_2017.npz"),
    ("path_integral_bf", "path_integral_bf.launch", "basis_function_09_12_2018.npz", "marietta_costmap_09_08_2018.npz"),
])
def test_path_integral_binaries_match_python_loop(bins, golden_dir, tmp_path, binary, launch, model_file, map_file):
    """20 ticks of the two-controller loop (debug-mode self-simulation) for both reference builds
    (network / basis-function dynamics): the C++ binary (launch XML -> .npz files -> controller classes)
    and a Python loop over the same C ABI must agree."""
    from autorally_amd import capi
    d = _params_dir(tmp_path, golden_dir, model_file, map_file)
    launch_path = os.path.join(ROOT, "autorally_amd", "host", "launch", launch)
    K, iters = 1024, 20
    start = (0.0, -10.0, 0.0)
    env = dict(os.environ, AR_MPPI_PARAMS_PATH=d)
    r = subprocess.run([bins[binary], launch_path, "--rollouts", str(K), "--max-iter", str(iters), "--no-sleep",
                        "--set", "x_pos=%r" % start[0], "--set", "y_pos=%r" % start[1], "--set", "heading=%r" % start[2],
                        "--trace", os.path.join(str(tmp_path), "trace.txt")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["iterations"] == iters and out["rollouts"] == K

    # the same loop from Python
    m, r_c1, r_c2, trs = P.load_costmap_npz(os.path.join(d, "maps", map_file))
    cfg = dict(K=K, T=100, map_rgba=m, r_c1=r_c1, r_c2=r_c2, trs=trs, cost=dict(P.DEFAULT_COST), seed=1234)
    cfg.update(P.DEFAULT_CTRL)
    if binary == "path_integral_bf":
        layers, theta = P.synthetic_model([6, 4], seed=0)  # unused by the basis-function model
        cfg.update(layers=layers, theta=theta, bf_W=P.load_bf_npz(os.path.join(d, "models", model_file)),
                   init_u=(0.0, -0.01))
        cfg["cost"]["desired_speed"] = 6.0
    else:
        layers, theta = P.load_model_npz(os.path.join(d, "models", model_file))
        cfg.update(layers=layers, theta=theta)
    actual, predicted = capi.Solver(cfg), capi.Solver(cfg)
    state = np.array([start[0], start[1], start[2], 0, 0, 0, 0], np.float32)
    pred_state_seq = np.zeros((100, 7), np.float32)
    pred_state_seq[0] = state
    from oracle import oracle as O  # host twin of model->updateState for the debug-mode self-simulation
    orc = O.Oracle(dict(cfg, start_state=state), fma_mode=1)
    n_actual = 0
    for it in range(iters):
        actual.slide_control_seq(1)
        predicted.slide_control_seq(1)
        pred_state_seq[:-1] = pred_state_seq[1:].copy()
        actual.compute_control(state)
        a_ss, a_cs = actual.nominal_traj(state)
        ps = pred_state_seq[0].copy()
        predicted.compute_control(ps)
        p_ss, p_cs = predicted.nominal_traj(ps)
        ca, cp = actual.get_results(False)["traj_cost"], predicted.get_results(False)["traj_cost"]
        # use_feedback_gains (launch default): both controllers, from the measured state, each tracking
        # its own solution (run_control_loop.cuh:220-225)
        ga = actual.compute_feedback_gains(state, a_ss, a_cs)["feedback"]
        gp = predicted.compute_feedback_gains(state, p_ss, p_cs)["feedback"]
        if ca < cp:
            cs, pred_state_seq, gains = a_cs, a_ss.copy(), ga
            n_actual += 1
        else:
            cs, pred_state_seq, gains = p_cs, p_ss.copy(), gp
        u = cs[0].copy()
        state, _ = orc.update_state(state, u)   # the reference advances the state twice per tick
        state, _ = orc.update_state(state, u)   # (shared model object, run_control_loop.cuh:299-300)
    np.testing.assert_allclose(out["final_state"], state, atol=2e-4, rtol=1e-4)
    assert out["actual_state_used"] == n_actual
    # both model families to the same bound: since the host replays of the basis-function model call powf like
    # car_bfs.cuh the binary's plant and this loop's oracle plant agree to the last digit,
    # so the fp32 central differences of ddp_dynamics.h:71-84 see the same inputs in both loops
    gtol = 2e-3
    print("gain row sums: binary", out["feedback_gain_row_sums_t0"], "python loop", gains[0].sum(axis=1).tolist())
    np.testing.assert_allclose(out["feedback_gain_row_sums_t0"], gains[0].sum(axis=1), rtol=gtol, atol=2e-4)
    assert np.abs(gains[0]).max() > 1e-3
    assert abs(state[4]) > 0.5  # the car actually drove


@pytest.mark.gpu
def test_live_update_hooks_and_debug_raster(bins, golden_dir, tmp_path):
    """runControlLoop's live updates (run_control_loop.cuh:162-204): a dynamic_reconfigure message waiting at
    the first tick reaches both controllers' cost parameters (updateParams_dcfg, costs.cu:75-87), and the debug
    raster around the predicted state is handed to the plant."""
    from autorally_amd import capi
    model_file, map_file = "autorally_nnet_09_12_2018.npz", "ccrf_costmap_09_29_2017.npz"
    d = _params_dir(tmp_path, golden_dir, model_file, map_file)
    launch_path = os.path.join(ROOT, "autorally_amd", "host", "launch", "path_integral_nn.launch")
    env = dict(os.environ, AR_MPPI_PARAMS_PATH=d)
    base = [bins["path_integral_nn"], launch_path, "--rollouts", "512", "--max-iter", "30", "--no-sleep",
            "--set", "x_pos=0.0", "--set", "y_pos=-10.0", "--set", "heading=0.0", "--set", "use_feedback_gains=false"]
    outs = {}
    for tag, extra in (("plain", []), ("slow", ["--dcfg-desired-speed", "2.0", "--debug-image"]),
                       ("poked", ["--poke-desired-speed", "2.0"]), ("cut", ["--poke-max-throttle", "0.0"])):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        outs[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert outs["plain"]["desired_speed"] == 8.0 and outs["plain"]["debug_image_pixels"] == 0  # launch value, no window
    assert outs["slow"]["desired_speed"] == 2.0
    assert outs["slow"]["final_state"][4] < outs["plain"]["final_state"][4] - 0.5  # the car really drives slower
    # plain writes to the public members at tick 5 (no setter, no version bump) reach the next solves, like the
    # unconditional paramsToDevice of mppi_controller.cu:605-606: a lower desired speed, a throttle range cut to 0
    assert outs["poked"]["desired_speed"] == 2.0
    assert outs["poked"]["final_state"][4] < outs["plain"]["final_state"][4] - 0.5
    assert outs["cut"]["final_state"][4] < outs["poked"]["final_state"][4]  # coasting from tick 5 on
    # the raster handed over at the last tick: 10 m x 10 m at 50 px/m around the predicted state
    assert outs["slow"]["debug_image_pixels"] == 500 * 500
    assert 0.0 < outs["slow"]["debug_image_sum"] < 500 * 500 * 10.0  # the synthetic map rises above 1 off the track


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["live", "debug"])
def test_control_loop_against_the_oracle(bins, golden_dir, tmp_path, mode):
    """runControlLoop in the C++ binary against a Python statement of the same loop in which the ORACLE is the
    solver of both controllers (same generator streams: seed 1234, 2T draws per solve) and the model that moves
    the car -- loop parity, not the solver compared with itself.
      live:  the live-pose half (run_control_loop.cuh:140-144, 175-181, 206-216) under a scripted pose clock
             (1, 2, 1, 3, 1 control periods between poses): the plant drives the handed-over solution;
      debug: the self-simulating half (:296-302), stride = optimization_stride, the state advanced TWICE per
             executed control (both controllers share one model object).
    Per tick: the stride slid by, the controller chosen, both trajectory costs, the first control handed over;
    at the end the state."""
    from oracle import oracle as O
    model_file, map_file = "autorally_nnet_09_12_2018.npz", "ccrf_costmap_09_29_2017.npz"
    d = _params_dir(tmp_path, golden_dir, model_file, map_file)
    launch_path = os.path.join(ROOT, "autorally_amd", "host", "launch", "path_integral_nn.launch")
    K, T, hz = 1024, 100, 50
    script = [0.02, 0.04, 0.02, 0.06, 0.02, 0.02]
    iters = len(script)
    start = (0.0, -10.0, 0.0)
    trace = os.path.join(str(tmp_path), "trace.txt")
    extra = ["--pose-script", ",".join("%r" % v for v in script)] if mode == "live" else ["--no-sleep"]
    r = subprocess.run([bins["path_integral_nn"], launch_path, "--rollouts", str(K), "--max-iter", str(iters),
                        "--set", "use_feedback_gains=false", "--set", "x_pos=%r" % start[0], "--set", "y_pos=%r" % start[1],
                        "--set", "heading=%r" % start[2], "--trace", trace] + extra, capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, AR_MPPI_PARAMS_PATH=d))
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    if mode == "live":
        want_strides = [1] + [int(round(v * hz)) for v in script[:-1]]   # tick 1: status still 1 -> optimization_stride
    else:
        want_strides = [1] * iters
    assert out["iterations"] == iters and [int(v) for v in out["strides"].split()] == want_strides
    rows = [l.split() for l in open(trace).read().splitlines()]

    m, r_c1, r_c2, trs = P.load_costmap_npz(os.path.join(d, "maps", map_file))
    layers, theta = P.load_model_npz(os.path.join(d, "models", model_file))
    cfg = dict(K=K, T=T, map_rgba=m, r_c1=r_c1, r_c2=r_c2, trs=trs, cost=dict(P.DEFAULT_COST), seed=1234, layers=layers, theta=theta)
    cfg.update(P.DEFAULT_CTRL)
    state = np.array([start[0], start[1], start[2], 0, 0, 0, 0], np.float32)
    orc = O.Oracle(dict(cfg, start_state=state), fma_mode=1, nthreads=8)
    ctl = {n: dict(U=np.tile(np.array(cfg["init_u"], np.float32), (T, 1)), hist=np.zeros(4, np.float32), off=0,
                   ss=np.zeros((T, 7), np.float32)) for n in ("a", "p")}
    ctl["a"]["ss"][0] = state
    ctl["p"]["ss"][0] = state

    def solve(c, x):
        eps = O.generate_noise(1234, c["off"], K, T)[None]
        c["off"] += 2 * T
        res = orc.compute_control(x, c["U"], c["hist"], eps)
        c["U"] = res["U"]
        c["ss"], c["cs"] = orc.nominal_traj(x, res["U"])
        return res["traj_cost"]

    for it in range(iters):
        stride = want_strides[it]
        for c in ctl.values():
            c["U"], c["hist"] = orc.slide_control_seq(c["U"], c["hist"], cfg["init_u"], stride)
            c["ss"][:T - stride] = c["ss"][stride:].copy()
        ca = solve(ctl["a"], state)
        cp = solve(ctl["p"], ctl["p"]["ss"][0].copy())
        used = "actual" if ca < cp else "predicted"
        chosen = ctl["a"] if ca < cp else ctl["p"]
        if ca < cp:  # run_control_loop.cuh:255-258: only the sequences the controller reports move over, not U_
            ctl["p"]["ss"], ctl["p"]["cs"] = ctl["a"]["ss"].copy(), ctl["a"]["cs"].copy()
        cs = chosen["cs"]
        if mode == "live":
            for t in range(int(round(script[it] * hz))):
                state, _ = orc.update_state(state, cs[t].copy())
        else:
            state, _ = orc.update_state(state, cs[0].copy())  # once through each controller's model_ pointer
            state, _ = orc.update_state(state, cs[0].copy())
        row = rows[it]
        assert row[1] == used, (it, row, ca, cp)
        assert abs(float(row[2]) - ca) <= 2e-3 * abs(ca) and abs(float(row[3]) - cp) <= 2e-3 * abs(cp), (it, row, ca, cp)
        assert abs(float(row[11]) - cs[0][0]) <= 2e-3 and abs(float(row[12]) - cs[0][1]) <= 2e-3, (it, row, cs[0])
        assert int(row[14]) == stride
    got = out["plant_state"] if mode == "live" else out["final_state"]
    np.testing.assert_allclose(got, state, atol=3e-3, rtol=1e-3)
    assert got[4] > 0.2  # the car drove
