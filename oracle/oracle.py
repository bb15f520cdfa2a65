"""ctypes loader for the CPU oracle (oracle/libmppi_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (autorally_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmppi_oracle.so")

MAX_LAYERS = 8


class CostParams(C.Structure):
    _fields_ = [
        ("desired_speed", C.c_float),
        ("speed_coeff", C.c_float),
        ("track_coeff", C.c_float),
        ("max_slip_ang", C.c_float),
        ("slip_penalty", C.c_float),
        ("track_slop", C.c_float),
        ("crash_coeff", C.c_float),
        ("steering_coeff", C.c_float),
        ("throttle_coeff", C.c_float),
        ("boundary_threshold", C.c_float),
        ("discount", C.c_float),
        ("num_timesteps", C.c_int),
        ("grid_res", C.c_int),
        ("r_c1", C.c_float * 3),
        ("r_c2", C.c_float * 3),
        ("trs", C.c_float * 3),
        ("l1_cost", C.c_int),
    ]


class Problem(C.Structure):
    _fields_ = [
        ("K", C.c_int),
        ("T", C.c_int),
        ("n_layers", C.c_int),
        ("layers", C.c_int * MAX_LAYERS),
        ("theta", C.POINTER(C.c_float)),
        ("dt", C.c_float),
        ("nu", C.c_float * 2),
        ("u_lo", C.c_float * 2),
        ("u_hi", C.c_float * 2),
        ("negate_yaw_der", C.c_int),
        ("opt_delay", C.c_int),
        ("gamma", C.c_float),
        ("P", CostParams),
        ("map_w", C.c_int),
        ("map_h", C.c_int),
        ("map_rgba", C.POINTER(C.c_float)),
        ("fma_mode", C.c_int),
        ("nthreads", C.c_int),
        ("bf_W", C.POINTER(C.c_float)),
    ]


class MrgState(C.Structure):
    _fields_ = [("s1", C.c_uint32 * 3), ("s2", C.c_uint32 * 3)]


def build(force=False):
    """Compile oracle/libmppi_oracle.so with gcc (idempotent)."""
    deps = [os.path.join(_HERE, f) for f in ("mppi_oracle.c", "ddp_oracle.c", "mppi_oracle.h")]
    if not force and os.path.exists(_LIB_PATH):
        if os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmppi_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


_NATIVE_PATH = os.path.join(_HERE, "libmppi_oracle_native.so")
_lib_native = None


def build_native(force=False):
    """The same sources compiled for THIS host (-O3 -march=native, still -ffp-contract=off with the explicit fmaf's): the
    CPU baseline SURVEY 8(d) specifies.  Built where it runs (bench.py's cpu_baseline leg on the GPU box); the checker of the
    tests stays the portable -O2 build."""
    deps = [os.path.join(_HERE, f) for f in ("mppi_oracle.c", "ddp_oracle.c", "mppi_oracle.h")]
    if not force and os.path.exists(_NATIVE_PATH) and os.path.getmtime(_NATIVE_PATH) >= max(os.path.getmtime(d) for d in deps):
        return _NATIVE_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmppi_oracle_native.so"], stdout=subprocess.DEVNULL)
    return _NATIVE_PATH


def lib(native=False):
    global _lib, _lib_native
    if native:
        if _lib_native is None:
            build_native(force=True)  # -march=native: never trust a file that travelled from another machine
            _lib_native = _bind(C.CDLL(_NATIVE_PATH))
        return _lib_native
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L):
    if True:
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        pp = C.POINTER(Problem)
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_num_params.restype = C.c_int
        L.orc_num_params.argtypes = [ip, C.c_int]
        L.orc_nn_forward.argtypes = [fp, ip, C.c_int, fp, fp, C.c_int]
        L.orc_state_deriv.argtypes = [pp, fp, fp, fp]
        L.orc_update_state.argtypes = [pp, fp, fp]
        L.orc_compute_cost.restype = C.c_float
        L.orc_compute_cost.argtypes = [pp, fp, fp, fp, ip]
        L.orc_debug_cost_raster.argtypes = [pp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, fp]
        L.orc_rollouts.argtypes = [pp, fp, fp, fp, fp, ip]
        L.orc_weights.argtypes = [fp, C.c_int, C.c_float, fp, fp, fp, fp]
        L.orc_weighted_reduction.argtypes = [fp, C.c_float, fp, C.c_int, C.c_int, fp, C.c_int]
        L.orc_savgol.argtypes = [fp, fp, C.c_int]
        L.orc_nominal_traj.argtypes = [pp, fp, fp, fp, fp]
        L.orc_slide_control_seq.argtypes = [fp, fp, fp, C.c_int, C.c_int]
        L.orc_compute_control.argtypes = [pp, C.c_int, fp, fp, fp, fp, fp, fp, fp]
        L.orc_mrg_seed.argtypes = [C.POINTER(MrgState), C.c_uint64]
        L.orc_mrg_skip_subsequences.argtypes = [C.POINTER(MrgState), C.c_uint64]
        L.orc_mrg_skip.argtypes = [C.POINTER(MrgState), C.c_uint64]
        L.orc_mrg_next_u01.restype = C.c_double
        L.orc_mrg_next_u01.argtypes = [C.POINTER(MrgState)]
        L.orc_mrg_next_z.restype = C.c_uint32
        L.orc_mrg_next_z.argtypes = [C.POINTER(MrgState)]
        L.orc_mrg_jump_matrices.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_box_muller.argtypes = [C.c_float, C.c_float, fp, fp]
        L.orc_generate_noise.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, fp]
        L.orc_basis_func.restype = C.c_float
        L.orc_set_bf_pow_products.argtypes = [C.c_int]
        L.orc_basis_func.argtypes = [C.c_int, fp, fp]
        L.orc_ddp_feedback_gains.restype = C.c_int
        L.orc_ddp_feedback_gains.argtypes = [fp, ip, C.c_int, C.c_int, C.c_float, fp, fp, C.c_int, fp, fp, fp, fp, fp, fp,
                                             fp, fp, fp, fp, fp, fp]
    return L


class Oracle:
    """Holds one orc_problem plus the numpy arrays it points into.

    `cfg` is a plain dict (see autorally_amd.synthetic.default_config):
      K, T, layers, theta, hz, nu, u_lo, u_hi, negate_yaw_der, opt_stride, gamma,
      cost (dict of CostParams scalars + l1_cost), map_rgba (H,W,4) f32, r_c1, r_c2, trs
    """

    def __init__(self, cfg, fma_mode=1, nthreads=1, native=False):
        self.L = lib(native)
        self.cfg = cfg
        p = Problem()
        p.K = int(cfg["K"])
        p.T = int(cfg["T"])
        layers = [int(x) for x in cfg["layers"]]
        p.n_layers = len(layers)
        for i, v in enumerate(layers):
            p.layers[i] = v
        self.theta = np.ascontiguousarray(cfg["theta"], dtype=np.float32)
        assert self.theta.size == self.L.orc_num_params(p.layers, p.n_layers)
        p.theta = _fp(self.theta)
        p.dt = np.float32(1.0 / int(cfg["hz"]))
        for i in range(2):
            p.nu[i] = cfg["nu"][i]
            p.u_lo[i] = cfg["u_lo"][i]
            p.u_hi[i] = cfg["u_hi"][i]
        p.negate_yaw_der = int(bool(cfg["negate_yaw_der"]))
        p.opt_delay = int(cfg["opt_stride"])
        p.gamma = cfg["gamma"]
        c = cfg["cost"]
        for name in ("desired_speed", "speed_coeff", "track_coeff", "max_slip_ang", "slip_penalty",
                     "track_slop", "crash_coeff", "steering_coeff", "throttle_coeff",
                     "boundary_threshold", "discount"):
            setattr(p.P, name, c[name])
        p.P.num_timesteps = p.T
        p.P.grid_res = 0
        p.P.l1_cost = int(bool(c.get("l1_cost", False)))
        for i in range(3):
            p.P.r_c1[i] = cfg["r_c1"][i]
            p.P.r_c2[i] = cfg["r_c2"][i]
            p.P.trs[i] = cfg["trs"][i]
        self.map = np.ascontiguousarray(cfg["map_rgba"], dtype=np.float32)
        assert self.map.ndim == 3 and self.map.shape[2] == 4
        p.map_h, p.map_w = self.map.shape[0], self.map.shape[1]
        p.map_rgba = _fp(self.map)
        p.fma_mode = int(fma_mode)
        p.nthreads = int(nthreads)
        # second dynamics family (GeneralizedLinear, generalized_linear.cu): cfg["bf_W"] = W[4][25]
        self.bf_W = None
        if cfg.get("bf_W") is not None:
            self.bf_W = np.ascontiguousarray(cfg["bf_W"], dtype=np.float32).reshape(4, 25)
            p.bf_W = _fp(self.bf_W)
        self.L.orc_set_num_threads(int(nthreads))
        self.p = p

    # -- pieces --
    def nn_forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(self.p.layers[self.p.n_layers - 1], dtype=np.float32)
        self.L.orc_nn_forward(self.p.theta, self.p.layers, self.p.n_layers, _fp(x), _fp(out),
                              self.p.fma_mode)
        return out

    def state_deriv(self, s, u):
        s = np.ascontiguousarray(s, dtype=np.float32)
        u = np.ascontiguousarray(u, dtype=np.float32)
        sd = np.zeros(7, dtype=np.float32)
        self.L.orc_state_deriv(C.byref(self.p), _fp(s), _fp(u), _fp(sd))
        return sd

    def update_state(self, s, u):
        s = np.array(s, dtype=np.float32)
        u = np.array(u, dtype=np.float32)
        self.L.orc_update_state(C.byref(self.p), _fp(s), _fp(u))
        return s, u

    def compute_cost(self, s, u, du, crash=0):
        s = np.ascontiguousarray(s, dtype=np.float32)
        u = np.ascontiguousarray(u, dtype=np.float32)
        du = np.ascontiguousarray(du, dtype=np.float32)
        cr = C.c_int(crash)
        c = self.L.orc_compute_cost(C.byref(self.p), _fp(s), _fp(u), _fp(du), C.byref(cr))
        return float(c), cr.value

    def debug_cost_raster(self, x, y, heading, width_m=10, height_m=10, ppm=50):
        """debugCostKernel (debug_kernels.cuh:39-88): [height_m*ppm, width_m*ppm] f32, unwritten pixels 0."""
        out = np.zeros((height_m * ppm, width_m * ppm), dtype=np.float32)
        self.L.orc_debug_cost_raster(C.byref(self.p), x, y, heading, width_m, height_m, ppm, _fp(out))
        return out

    def rollouts(self, state, U, eps):
        """Returns (costs[K], V[K,T,2], crash[K]); eps is not modified."""
        K, T = self.p.K, self.p.T
        state = np.ascontiguousarray(state, dtype=np.float32)
        U = np.ascontiguousarray(U, dtype=np.float32).reshape(T, 2)
        V = np.array(eps, dtype=np.float32).reshape(K, T, 2).copy()
        costs = np.zeros(K, dtype=np.float32)
        crash = np.zeros(K, dtype=np.int32)
        self.L.orc_rollouts(C.byref(self.p), _fp(state), _fp(U), _fp(V), _fp(costs),
                            crash.ctypes.data_as(C.POINTER(C.c_int)))
        return costs, V, crash

    def weights(self, costs):
        costs = np.ascontiguousarray(costs, dtype=np.float32)
        w = np.zeros_like(costs)
        b, e, tc = C.c_float(), C.c_float(), C.c_float()
        self.L.orc_weights(_fp(costs), costs.size, float(self.p.gamma), _fp(w), C.byref(b), C.byref(e),
                           C.byref(tc))
        return w, b.value, e.value, tc.value

    def weighted_reduction(self, w, eta, V):
        K, T = self.p.K, self.p.T
        w = np.ascontiguousarray(w, dtype=np.float32)
        V = np.ascontiguousarray(V, dtype=np.float32).reshape(K, T, 2)
        U = np.zeros((T, 2), dtype=np.float32)
        self.L.orc_weighted_reduction(_fp(w), float(np.float32(eta)), _fp(V), K, T, _fp(U),
                                      self.p.fma_mode)
        return U

    def savgol(self, U, hist):
        U = np.array(U, dtype=np.float32).reshape(-1, 2).copy()
        hist = np.ascontiguousarray(hist, dtype=np.float32)
        self.L.orc_savgol(_fp(U), _fp(hist), U.shape[0])
        return U

    def nominal_traj(self, state, U):
        T = self.p.T
        state = np.ascontiguousarray(state, dtype=np.float32)
        U = np.ascontiguousarray(U, dtype=np.float32).reshape(T, 2)
        ss = np.zeros((T, 7), dtype=np.float32)
        cs = np.zeros((T, 2), dtype=np.float32)
        self.L.orc_nominal_traj(C.byref(self.p), _fp(state), _fp(U), _fp(ss), _fp(cs))
        return ss, cs

    DDP_Q = (0.5, 0.5, 0.25, 0.0, 0.05, 0.01, 0.01)  # initDDP, mppi_controller.cu:410-417
    DDP_R = (10.0, 10.0)
    DDP_QF = (0.0,) * 7

    def ddp_feedback_gains(self, state, target_x, target_u, Q=DDP_Q, R=DDP_R, Qf=DDP_QF):
        """computeFeedbackGains (mppi_controller.cu:431-441): dict(feedback[T,2,7], feedforward[T,2], x, u, total_cost)."""
        T = self.p.T
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        state, tx, tu = f32(state), f32(target_x).reshape(T, 7), f32(target_u).reshape(T, 2)
        Q, R, Qf = f32(Q), f32(R), f32(Qf)
        lo, hi = f32([self.p.u_lo[0], self.p.u_lo[1]]), f32([self.p.u_hi[0], self.p.u_hi[1]])
        fb = np.zeros((T, 2, 7), np.float32)
        ff = np.zeros((T, 2), np.float32)
        x = np.zeros((T, 7), np.float32)
        u = np.zeros((T, 2), np.float32)
        tc = np.zeros(1, np.float32)
        rc = self.L.orc_ddp_feedback_gains(_fp(self.theta), self.p.layers, self.p.n_layers, T, self.p.dt, _fp(lo), _fp(hi),
                                           self.p.negate_yaw_der, _fp(Q), _fp(R), _fp(Qf), _fp(state), _fp(tx), _fp(tu),
                                           _fp(fb), _fp(ff), _fp(x), _fp(u), _fp(tc),
                                           _fp(self.bf_W) if self.bf_W is not None else None)
        if rc:
            raise RuntimeError("DDP: control Hessian could not be factorised")
        return dict(feedback=fb, feedforward=ff, x=x, u=u, total_cost=float(tc[0]))

    def slide_control_seq(self, U, hist, init_u, stride):
        U = np.array(U, dtype=np.float32).reshape(-1, 2).copy()
        hist = np.array(hist, dtype=np.float32).copy()
        init_u = np.ascontiguousarray(init_u, dtype=np.float32)
        self.L.orc_slide_control_seq(_fp(U), _fp(hist), _fp(init_u), U.shape[0], int(stride))
        return U, hist

    def compute_control(self, state, U, hist, eps, num_iters=1):
        """One full solve. eps: [num_iters,K,T,2]. Returns dict(U, traj_cost, costs, w, V)."""
        K, T = self.p.K, self.p.T
        state = np.ascontiguousarray(state, dtype=np.float32)
        U = np.array(U, dtype=np.float32).reshape(T, 2).copy()
        hist = np.ascontiguousarray(hist, dtype=np.float32)
        V = np.array(eps, dtype=np.float32).reshape(num_iters, K, T, 2).copy()
        tc = C.c_float()
        costs = np.zeros(K, dtype=np.float32)
        w = np.zeros(K, dtype=np.float32)
        self.L.orc_compute_control(C.byref(self.p), int(num_iters), _fp(state), _fp(U), _fp(hist),
                                   _fp(V), C.byref(tc), _fp(costs), _fp(w))
        return dict(U=U, traj_cost=tc.value, costs=costs, w=w, V=V)


def generate_noise(seed, offset, K, T):
    eps = np.zeros((K, T, 2), dtype=np.float32)
    lib().orc_generate_noise(int(seed), int(offset), int(K), int(T), _fp(eps))
    return eps


def jump_matrices(e):
    a1 = (C.c_uint32 * 9)()
    a2 = (C.c_uint32 * 9)()
    lib().orc_mrg_jump_matrices(int(e), a1, a2)
    return np.array(a1, dtype=np.uint64).reshape(3, 3), np.array(a2, dtype=np.uint64).reshape(3, 3)
