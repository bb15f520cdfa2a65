"""ctypes binding of libmppi_hip.so (include/mppi_hip.h) -- plumbing for tests and bench.py.

The product is the shared library; this module only marshals numpy arrays across the C ABI.
It never computes anything itself and has no CPU fallback: if the library is missing or no
gfx950 device is usable, it raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MPPI_LIB_PATH: developer override used for kernel A/B builds (scratch/), never a fallback
LIB_PATH = os.environ.get("MPPI_LIB_PATH") or os.path.join(HERE, "libmppi_hip.so")
MAX_LAYERS = 8

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = range(6)


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int),
        ("num_rollouts", C.c_int),
        ("num_timesteps", C.c_int),
        ("hz", C.c_int),
        ("optimization_stride", C.c_int),
        ("gamma", C.c_float),
        ("num_iters", C.c_int),
        ("n_layers", C.c_int),
        ("layers", C.c_int * MAX_LAYERS),
        ("exploration_std", C.c_float * 2),
        ("init_control", C.c_float * 2),
        ("control_min", C.c_float * 2),
        ("control_max", C.c_float * 2),
        ("negate_yaw_der", C.c_int),
        ("seed", C.c_uint64),
    ]


class CostParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "desired_speed", "speed_coeff", "track_coeff", "max_slip_ang", "slip_penalty", "track_slop",
        "crash_coeff", "steering_coeff", "throttle_coeff", "boundary_threshold", "discount")] + [
        ("l1_cost", C.c_int)]


class StageTimes(C.Structure):
    _fields_ = [("n_solves", C.c_int), ("noise_ms", C.c_float), ("rollout_ms", C.c_float),
                ("weights_ms", C.c_float), ("reduction_ms", C.c_float), ("total_ms", C.c_float)]


# every symbol include/mppi_hip.h declares
SYMBOLS = [
    "mppi_abi_version", "mppi_strerror", "mppi_last_error", "mppi_device_count", "mppi_create",
    "mppi_destroy", "mppi_set_nn_params", "mppi_update_model", "mppi_set_control_limits",
    "mppi_set_costmap", "mppi_set_costmap_transform", "mppi_set_costmap_channel", "mppi_set_cost_params", "mppi_reset_controls",
    "mppi_set_control_seq", "mppi_get_control_seq", "mppi_set_control_hist", "mppi_get_control_hist",
    "mppi_savitsky_golay", "mppi_slide_control_seq", "mppi_seed", "mppi_set_noise", "mppi_generate_noise",
    "mppi_compute_control", "mppi_control_ticks", "mppi_compute_control_async", "mppi_synchronize",
    "mppi_compute_control_batch_async", "mppi_compute_control_batch", "mppi_control_ticks_batch", "mppi_get_results",
    "mppi_get_applied_controls", "mppi_rollout_only", "mppi_nominal_traj", "mppi_nominal_traj_pair",
    "mppi_set_bf_params", "mppi_set_ddp_weights", "mppi_debug_cost_raster", "mppi_compute_feedback_gains", "mppi_get_feedback_gains",
    "mppi_enable_stage_timing", "mppi_reset_stage_times", "mppi_get_stage_times",
    "mppi_rollout_variant", "mppi_set_rollout_variant", "mppi_debug_dynamics",
    "mppi_debug_inject_handover_fault", "mppi_compute_feedback_gains_pair", "mppi_set_host_threads",
    "mppi_debug_capture_iterations", "mppi_debug_get_iterations", "mppi_set_wait_timeout", "mppi_debug_form_candidates",
    "mppi_debug_set_chained_ticks", "mppi_debug_min_cost",
]

ABI2_SYMBOLS = ("mppi_debug_inject_handover_fault", "mppi_savitsky_golay", "mppi_set_costmap_transform",
                "mppi_compute_control_batch", "mppi_compute_control_batch_async", "mppi_control_ticks_batch",
                "mppi_nominal_traj_pair")
ABI3_SYMBOLS = ("mppi_compute_feedback_gains_pair", "mppi_set_host_threads")
ABI4_SYMBOLS = ("mppi_debug_capture_iterations", "mppi_debug_get_iterations", "mppi_set_wait_timeout", "mppi_debug_form_candidates")
ABI5_SYMBOLS = ("mppi_debug_set_chained_ticks", "mppi_debug_min_cost")

_lib = None


class MppiError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("mppi status %d: %s" % (status, msg))
        self.status = status


def lib():
    """Loads libmppi_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmppi_hip.so is missing: run `python -m autorally_amd.build` "
                               "or __graft_entry__.build() first")
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        hp = C.c_void_p
        L.mppi_abi_version.restype = C.c_int
        L.mppi_strerror.restype = C.c_char_p
        L.mppi_strerror.argtypes = [C.c_int]
        L.mppi_last_error.restype = C.c_char_p
        L.mppi_last_error.argtypes = [hp]
        L.mppi_device_count.restype = C.c_int
        L.mppi_create.argtypes = [C.POINTER(Config), C.POINTER(hp)]
        L.mppi_destroy.argtypes = [hp]
        L.mppi_set_nn_params.argtypes = [hp, fp, C.c_size_t]
        L.mppi_update_model.argtypes = [hp, C.POINTER(C.c_int), C.c_int, fp, C.c_size_t]
        L.mppi_set_control_limits.argtypes = [hp, fp, fp]
        L.mppi_set_costmap.argtypes = [hp, C.c_int, C.c_int, fp, fp, fp, fp]
        L.mppi_set_costmap_transform.argtypes = [hp, fp, fp, fp]
        L.mppi_set_costmap_channel.argtypes = [hp, C.c_int, fp, C.c_size_t]
        L.mppi_set_cost_params.argtypes = [hp, C.POINTER(CostParams)]
        L.mppi_reset_controls.argtypes = [hp]
        L.mppi_set_control_seq.argtypes = [hp, fp, C.c_size_t]
        L.mppi_get_control_seq.argtypes = [hp, fp, C.c_size_t]
        L.mppi_set_control_hist.argtypes = [hp, fp]
        L.mppi_get_control_hist.argtypes = [hp, fp]
        L.mppi_slide_control_seq.argtypes = [hp, C.c_int]
        L.mppi_seed.argtypes = [hp, C.c_uint64, C.c_uint64]
        L.mppi_set_noise.argtypes = [hp, fp, C.c_size_t]
        L.mppi_generate_noise.argtypes = [hp, fp, C.c_size_t]
        L.mppi_compute_control.argtypes = [hp, fp]
        L.mppi_compute_control_async.argtypes = [hp, fp]
        L.mppi_control_ticks.argtypes = [hp, fp, C.c_int, C.c_int]
        L.mppi_synchronize.argtypes = [hp]
        L.mppi_get_results.argtypes = [hp, fp, fp, fp, fp]
        L.mppi_get_applied_controls.argtypes = [hp, fp, C.c_size_t]
        L.mppi_rollout_only.argtypes = [hp, fp, fp]
        L.mppi_nominal_traj.argtypes = [hp, fp, fp, fp]
        L.mppi_set_bf_params.argtypes = [hp, fp, C.c_size_t]
        L.mppi_debug_cost_raster.argtypes = [hp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, fp, C.c_size_t]
        L.mppi_set_ddp_weights.argtypes = [hp, fp, fp, fp]
        L.mppi_compute_feedback_gains.argtypes = [hp, fp, fp, fp]
        L.mppi_get_feedback_gains.argtypes = [hp, fp, fp, fp, fp, fp]
        L.mppi_enable_stage_timing.argtypes = [hp, C.c_int]
        L.mppi_reset_stage_times.argtypes = [hp]
        L.mppi_get_stage_times.argtypes = [hp, C.POINTER(StageTimes)]
        L.mppi_rollout_variant.restype = C.c_char_p
        L.mppi_rollout_variant.argtypes = [hp]
        L.mppi_set_rollout_variant.argtypes = [hp, C.c_char_p]
        L.mppi_debug_dynamics.argtypes = [hp, C.c_int, fp, fp, fp]
        # symbols added with ABI version 2 (include/mppi_hip.h): an older library (kernel A/B through
        # MPPI_LIB_PATH) reports version 1 and lacks them
        v2 = L.mppi_abi_version() >= 2
        if v2:
            L.mppi_debug_inject_handover_fault.argtypes = [hp, C.c_int, C.c_int]
            L.mppi_savitsky_golay.argtypes = [hp]
            L.mppi_compute_control_batch_async.argtypes = [C.POINTER(hp), fp, C.c_int]
            L.mppi_compute_control_batch.argtypes = [C.POINTER(hp), fp, C.c_int]
            L.mppi_control_ticks_batch.argtypes = [C.POINTER(hp), fp, C.c_int, C.c_int, C.c_int]
            L.mppi_nominal_traj_pair.argtypes = [hp, fp, fp, fp, hp, fp, fp, fp]
        v3 = L.mppi_abi_version() >= 3
        if v3:
            L.mppi_compute_feedback_gains_pair.argtypes = [hp, fp, fp, fp, hp, fp, fp, fp]
            L.mppi_set_host_threads.argtypes = [C.c_int]
        v4 = L.mppi_abi_version() >= 4
        if v4:
            L.mppi_debug_capture_iterations.argtypes = [hp, C.c_int]
            L.mppi_debug_get_iterations.argtypes = [hp, fp, fp, fp]
            L.mppi_set_wait_timeout.argtypes = [hp, C.c_double]
            L.mppi_debug_form_candidates.argtypes = [hp, C.POINTER(C.c_char_p), C.c_int]
        v5 = L.mppi_abi_version() >= 5
        if v5:
            L.mppi_debug_set_chained_ticks.argtypes = [hp, C.c_int]
            L.mppi_debug_min_cost.argtypes = [hp, C.c_int, C.POINTER(C.c_int)]
        for s in SYMBOLS:  # every declared symbol of the library's ABI version must be there
            if (v2 or s not in ABI2_SYMBOLS) and (v3 or s not in ABI3_SYMBOLS) and (v4 or s not in ABI4_SYMBOLS) and \
                    (v5 or s not in ABI5_SYMBOLS):
                getattr(L, s)
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def make_config_struct(cfg, device=0):
    c = Config()
    c.device = int(device)
    c.num_rollouts = int(cfg["K"])
    c.num_timesteps = int(cfg["T"])
    c.hz = int(cfg["hz"])
    c.optimization_stride = int(cfg["opt_stride"])
    c.gamma = float(cfg["gamma"])
    c.num_iters = int(cfg.get("num_iters", 1))
    # cfg["bf_W"] selects the GeneralizedLinear basis-function dynamics (n_layers = 0)
    layers = [] if cfg.get("bf_W") is not None else [int(x) for x in cfg["layers"]]
    c.n_layers = len(layers)
    for i, v in enumerate(layers):
        c.layers[i] = v
    for i in range(2):
        c.exploration_std[i] = float(cfg["nu"][i])
        c.init_control[i] = float(cfg["init_u"][i])
        c.control_min[i] = float(cfg["u_lo"][i])
        c.control_max[i] = float(cfg["u_hi"][i])
    c.negate_yaw_der = int(bool(cfg["negate_yaw_der"]))
    c.seed = int(cfg.get("seed", 1234))
    return c


def make_cost_struct(cost):
    p = CostParams()
    for n, _ in CostParams._fields_[:-1]:
        setattr(p, n, float(cost[n]))
    p.l1_cost = int(bool(cost.get("l1_cost", False)))
    return p


class Solver:
    """One mppi_handle: a thin object wrapper over the C ABI (method names follow the ABI)."""

    def __init__(self, cfg, device=0):
        self.L = lib()
        self.cfg = cfg
        self.K, self.T = int(cfg["K"]), int(cfg["T"])
        self.num_iters = int(cfg.get("num_iters", 1))
        self.h = C.c_void_p()
        c = make_config_struct(cfg, device)
        rc = self.L.mppi_create(C.byref(c), C.byref(self.h))
        if rc != OK:
            self.h = C.c_void_p()
            raise MppiError(rc, self.L.mppi_strerror(rc).decode())
        if cfg.get("bf_W") is not None:
            W = _f32(cfg["bf_W"], (4, 25))
            self._ck(self.L.mppi_set_bf_params(self.h, _fp(W), W.size))
        else:
            theta = _f32(cfg["theta"])
            self._ck(self.L.mppi_set_nn_params(self.h, _fp(theta), theta.size))
        m = _f32(cfg["map_rgba"])
        H, W = m.shape[0], m.shape[1]
        self._ck(self.L.mppi_set_costmap(self.h, W, H, _fp(m), _fp(_f32(cfg["r_c1"])),
                                         _fp(_f32(cfg["r_c2"])), _fp(_f32(cfg["trs"]))))
        p = make_cost_struct(cfg["cost"])
        self._ck(self.L.mppi_set_cost_params(self.h, C.byref(p)))

    def _ck(self, rc):
        if rc != OK:
            raise MppiError(rc, "%s (%s)" % (self.L.mppi_strerror(rc).decode(),
                                             self.L.mppi_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.L.mppi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- setters / getters ---
    def set_cost_params(self, cost):
        p = make_cost_struct(cost)
        self._ck(self.L.mppi_set_cost_params(self.h, C.byref(p)))

    def set_control_limits(self, lo, hi):
        self._ck(self.L.mppi_set_control_limits(self.h, _fp(_f32(lo)), _fp(_f32(hi))))

    def set_control_seq(self, U):
        U = _f32(U)
        self._ck(self.L.mppi_set_control_seq(self.h, _fp(U), U.size))

    def get_control_seq(self):
        U = np.zeros((self.T, 2), dtype=np.float32)
        self._ck(self.L.mppi_get_control_seq(self.h, _fp(U), U.size))
        return U

    def reset_controls(self):
        self._ck(self.L.mppi_reset_controls(self.h))

    def set_control_hist(self, hist):
        self._ck(self.L.mppi_set_control_hist(self.h, _fp(_f32(hist, (4,)))))

    def get_control_hist(self):
        h = np.zeros(4, dtype=np.float32)
        self._ck(self.L.mppi_get_control_hist(self.h, _fp(h)))
        return h

    def savitsky_golay(self):
        self._ck(self.L.mppi_savitsky_golay(self.h))

    def slide_control_seq(self, stride):
        self._ck(self.L.mppi_slide_control_seq(self.h, int(stride)))

    def seed(self, seed, offset=0):
        self._ck(self.L.mppi_seed(self.h, int(seed), int(offset)))

    def set_noise(self, eps):
        eps = _f32(eps)
        self._ck(self.L.mppi_set_noise(self.h, _fp(eps), eps.size))

    def generate_noise(self):
        e = np.zeros((self.K, self.T, 2), dtype=np.float32)
        self._ck(self.L.mppi_generate_noise(self.h, _fp(e), e.size))
        return e

    def update_model(self, description, data):
        d = (C.c_int * len(description))(*[int(x) for x in description])
        data = _f32(data)
        self._ck(self.L.mppi_update_model(self.h, d, len(description), _fp(data), data.size))

    def set_costmap_transform(self, r_c1, r_c2, trs):
        self._ck(self.L.mppi_set_costmap_transform(self.h, _fp(_f32(r_c1, (3,))), _fp(_f32(r_c2, (3,))), _fp(_f32(trs, (3,)))))

    def set_costmap_channel(self, channel, data):
        data = _f32(data)
        self._ck(self.L.mppi_set_costmap_channel(self.h, int(channel), _fp(data), data.size))

    # --- compute ---
    def compute_control(self, state):
        self._ck(self.L.mppi_compute_control(self.h, _fp(_f32(state, (7,)))))

    def bind_state(self, state):
        """Returns a zero-argument callable running mppi_compute_control on a fixed state buffer
        (skips the per-call numpy/ctypes marshalling; used by bench.py's timed loop)."""
        buf = _f32(state, (7,)).copy()
        ptr = _fp(buf)
        fn, h, ck = self.L.mppi_compute_control, self.h, self._ck

        def run(_keep=buf):
            ck(fn(h, ptr))
        return run

    def control_ticks(self, state, n_ticks, stride=1):
        """n_ticks x (compute_control + slide_control_seq(stride)) inside one library call."""
        self._ck(self.L.mppi_control_ticks(self.h, _fp(_f32(state, (7,))), int(n_ticks), int(stride)))

    def debug_set_chained_ticks(self, on):
        self._ck(self.L.mppi_debug_set_chained_ticks(self.h, int(on)))

    def debug_min_cost(self, on=-1):
        """Switch beta-from-the-rollout-kernel on / off (on < 0: leave it); returns whether the LAST solve's tail kernel used it."""
        got = C.c_int(0)
        self._ck(self.L.mppi_debug_min_cost(self.h, int(on), C.byref(got)))
        return bool(got.value)

    def compute_control_async(self, state):
        self._ck(self.L.mppi_compute_control_async(self.h, _fp(_f32(state, (7,)))))

    def synchronize(self):
        self._ck(self.L.mppi_synchronize(self.h))

    def get_results(self, with_vectors=True):
        U = np.zeros((self.T, 2), dtype=np.float32)
        tc = C.c_float()
        costs = np.zeros(self.K, dtype=np.float32) if with_vectors else None
        w = np.zeros(self.K, dtype=np.float32) if with_vectors else None
        self._ck(self.L.mppi_get_results(self.h, _fp(U), C.byref(tc),
                                         _fp(costs) if with_vectors else None,
                                         _fp(w) if with_vectors else None))
        return dict(U=U, traj_cost=tc.value, costs=costs, w=w)

    def get_applied_controls(self):
        V = np.zeros((self.K, self.T, 2), dtype=np.float32)
        self._ck(self.L.mppi_get_applied_controls(self.h, _fp(V), V.size))
        return V

    def rollout_only(self, state):
        costs = np.zeros(self.K, dtype=np.float32)
        self._ck(self.L.mppi_rollout_only(self.h, _fp(_f32(state, (7,))), _fp(costs)))
        return costs

    def nominal_traj(self, state):
        ss = np.zeros((self.T, 7), dtype=np.float32)
        cs = np.zeros((self.T, 2), dtype=np.float32)
        self._ck(self.L.mppi_nominal_traj(self.h, _fp(_f32(state, (7,))), _fp(ss), _fp(cs)))
        return ss, cs

    def set_ddp_weights(self, Q, R, Qf):
        self._ck(self.L.mppi_set_ddp_weights(self.h, _fp(_f32(Q, (7,))), _fp(_f32(R, (2,))), _fp(_f32(Qf, (7,)))))

    def compute_feedback_gains(self, state, target_x=None, target_u=None):
        """computeFeedbackGains + getFeedbackGains: dict(feedback[T,2,7], feedforward[T,2], x, u, total_cost).
        Targets default to the nominal trajectory of the current control sequence from `state`."""
        tx = _fp(_f32(target_x, (self.T, 7))) if target_x is not None else None
        tu = _fp(_f32(target_u, (self.T, 2))) if target_u is not None else None
        self._ck(self.L.mppi_compute_feedback_gains(self.h, _fp(_f32(state, (7,))), tx, tu))
        return self.feedback_gains()

    def feedback_gains(self):
        """getFeedbackGains of the last successful computeFeedbackGains."""
        fb = np.zeros((self.T, 2, 7), dtype=np.float32)
        ff = np.zeros((self.T, 2), dtype=np.float32)
        x = np.zeros((self.T, 7), dtype=np.float32)
        u = np.zeros((self.T, 2), dtype=np.float32)
        tc = np.zeros(1, dtype=np.float32)
        self._ck(self.L.mppi_get_feedback_gains(self.h, _fp(fb), _fp(ff), _fp(x), _fp(u), _fp(tc)))
        return dict(feedback=fb, feedforward=ff, x=x, u=u, total_cost=float(tc[0]))

    def debug_cost_raster(self, x, y, heading, width_m=10, height_m=10, ppm=50):
        out = np.zeros((height_m * ppm, width_m * ppm), dtype=np.float32)
        self._ck(self.L.mppi_debug_cost_raster(self.h, x, y, heading, width_m, height_m, ppm, _fp(out), out.size))
        return out

    def debug_dynamics(self, states, controls):
        states = _f32(states).reshape(-1, 7)
        controls = _f32(controls).reshape(-1, 2)
        out = np.zeros_like(states)
        self._ck(self.L.mppi_debug_dynamics(self.h, states.shape[0], _fp(states), _fp(controls), _fp(out)))
        return out

    def debug_capture_iterations(self, on=1):
        self._ck(self.L.mppi_debug_capture_iterations(self.h, int(on)))

    def debug_get_iterations(self, with_V=False):
        """{"U_raw": [iters][T][2], "costs": [iters][K], "V": [iters][K][T][2] (explicit-noise solves, with_V)}"""
        it, K, T = int(self.cfg.get("num_iters", 1)), self.cfg["K"], self.cfg["T"]
        U = np.zeros((it, T, 2), np.float32)
        c = np.zeros((it, K), np.float32)
        V = np.zeros((it, K, T, 2), np.float32) if with_V else None
        self._ck(self.L.mppi_debug_get_iterations(self.h, _fp(U), _fp(c), _fp(V) if with_V else None))
        out = {"U_raw": U, "costs": c}
        if with_V:
            out["V"] = V
        return out

    def form_candidates(self):
        """Names of the kernel forms the selection table knows for this model (mppi_debug_form_candidates)."""
        buf = (C.c_char_p * 16)()
        n = self.L.mppi_debug_form_candidates(self.h, buf, 16)
        return [buf[i].decode() for i in range(n)]

    def set_wait_timeout(self, seconds):
        self._ck(self.L.mppi_set_wait_timeout(self.h, float(seconds)))

    def debug_inject_handover_fault(self, wave, spin_budget):
        self._ck(self.L.mppi_debug_inject_handover_fault(self.h, int(wave), int(spin_budget)))

    # --- measurement ---
    def enable_stage_timing(self, on=1):
        """on = 1: every solve, N > 1: every Nth solve, 0: off."""
        self._ck(self.L.mppi_enable_stage_timing(self.h, int(on)))

    def reset_stage_times(self):
        self._ck(self.L.mppi_reset_stage_times(self.h))

    def get_stage_times(self):
        st = StageTimes()
        self._ck(self.L.mppi_get_stage_times(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in StageTimes._fields_}

    def rollout_variant(self):
        return self.L.mppi_rollout_variant(self.h).decode()

    def set_rollout_variant(self, name):
        self._ck(self.L.mppi_set_rollout_variant(self.h, name.encode()))


def compute_control_batch(solvers, states, blocking=True):
    """mppi_compute_control_batch[_async]: the solves of several Solver objects enqueued together (one launch of
    the quad kernel + one of the tail kernel where the library can batch them)."""
    n = len(solvers)
    hs = (C.c_void_p * n)(*[s.h for s in solvers])
    st = np.ascontiguousarray(np.stack([_f32(x, (7,)) for x in states]), dtype=np.float32)
    L = solvers[0].L
    fn = L.mppi_compute_control_batch if blocking else L.mppi_compute_control_batch_async
    rc = fn(hs, _fp(st), n)
    if rc != OK:
        msgs = [s.L.mppi_last_error(s.h).decode() for s in solvers]
        raise MppiError(rc, "; ".join(m for m in msgs if m))


def control_ticks_batch(solvers, states, n_ticks, stride=1):
    """mppi_control_ticks_batch: n_ticks x (batched solve + slide of every controller) inside one library call."""
    n = len(solvers)
    hs = (C.c_void_p * n)(*[s.h for s in solvers])
    st = np.ascontiguousarray(np.stack([_f32(x, (7,)) for x in states]), dtype=np.float32)
    rc = solvers[0].L.mppi_control_ticks_batch(hs, _fp(st), n, int(n_ticks), int(stride))
    if rc != OK:
        raise MppiError(rc, "; ".join(s.L.mppi_last_error(s.h).decode() for s in solvers))


def set_host_threads(n):
    """mppi_set_host_threads: 1 = the caller's thread only, 2 = one helper thread for the paired host work of a tick."""
    rc = lib().mppi_set_host_threads(int(n))
    if rc != OK:
        raise MppiError(rc, "mppi_set_host_threads(%d)" % n)


def compute_feedback_gains_pair(sol_a, state_a, sol_b, state_b, targets_a=None, targets_b=None):
    """mppi_compute_feedback_gains_pair; targets_x = (state_seq [T][7], control_seq [T][2]) or None (the handle's own
    nominal replay from state_x)."""
    def tp(t):
        if t is None:
            return None, None
        return _fp(_f32(t[0], (sol_a.T, 7))), _fp(_f32(t[1], (sol_a.T, 2)))
    ta, tb = tp(targets_a), tp(targets_b)
    rc = sol_a.L.mppi_compute_feedback_gains_pair(sol_a.h, _fp(_f32(state_a, (7,))), ta[0], ta[1],
                                                  sol_b.h, _fp(_f32(state_b, (7,))), tb[0], tb[1])
    if rc != OK:
        raise MppiError(rc, sol_a.L.mppi_last_error(sol_a.h).decode() + " | " + sol_b.L.mppi_last_error(sol_b.h).decode())


def nominal_traj_pair(sol_a, state_a, sol_b, state_b):
    """mppi_nominal_traj_pair: ((state_seq_a, control_seq_a), (state_seq_b, control_seq_b))."""
    out = []
    for s in (sol_a, sol_b):
        out.append((np.zeros((s.T, 7), np.float32), np.zeros((s.T, 2), np.float32)))
    rc = sol_a.L.mppi_nominal_traj_pair(sol_a.h, _fp(_f32(state_a, (7,))), _fp(out[0][0]), _fp(out[0][1]),
                                        sol_b.h, _fp(_f32(state_b, (7,))), _fp(out[1][0]), _fp(out[1][1]))
    if rc != OK:
        raise MppiError(rc, sol_a.L.mppi_last_error(sol_a.h).decode() + " | " + sol_b.L.mppi_last_error(sol_b.h).decode())
    return out

