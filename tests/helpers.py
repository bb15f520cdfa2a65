"""Shared helpers for the parity tests (tests/ only)."""
import numpy as np

from autorally_amd import synthetic as S
from oracle import oracle as O


def noise_for(cfg, seed=1234, iters=None):
    """Explicit eps[num_iters][K][T][2] from the oracle's statement of the noise spec."""
    it = int(cfg.get("num_iters", 1)) if iters is None else iters
    K, T = cfg["K"], cfg["T"]
    out = np.zeros((it, K, T, 2), dtype=np.float32)
    for i in range(it):
        out[i] = O.generate_noise(seed, 2 * T * i, K, T)
    return out


def warm_U(cfg, seed=7):
    """A non-trivial nominal control sequence (smooth, inside the limits)."""
    T = cfg["T"]
    t = np.arange(T, dtype=np.float64)
    rng = np.random.RandomState(seed)
    # negative steering turns left; -0.28 / 0.22 keeps the car on the radius-10 ring (probed with
    # the oracle's nominal trajectory); the oval starts on a straight
    base = -0.28 if cfg.get("track") == "ring" else 0.0
    thr = 0.22 if cfg.get("track") == "ring" else 0.3
    U = np.stack([base + 0.05 * np.sin(t / 9.0 + rng.uniform(0, 1)), thr + 0.05 * np.cos(t / 13.0)], axis=1)
    return U.astype(np.float32)


def rel_err(a, b, floor=1.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def load_nn_golden(golden_dir):
    """The reference-generated NN vectors as one dict: the shipped models read through the reference's
    reader (gen_golden.py) and the model written by its training pipeline's writer
    (gen_model_writer_golden.py)."""
    import os
    g = {}
    for f in ("nn_dynamics_golden.npz", "nn_writer_golden.npz"):
        z = np.load(os.path.join(golden_dir, f))
        g.update({k: z[k] for k in z.files})
    return g
