"""CPU-side checks of the C ABI: the library builds, loads and exports every symbol that
include/mppi_hip.h declares; argument validation works without a GPU; no compute is attempted."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from autorally_amd import build as B
from autorally_amd import capi
from autorally_amd import synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    B.build()
    return capi.lib()


def test_header_symbols_exported(L):
    hdr = open(os.path.join(ROOT, "include", "mppi_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mppi_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "libmppi_hip.so does not export %s" % name
    assert sorted(capi.SYMBOLS) == declared


def test_version_and_strerror(L):
    assert L.mppi_abi_version() == 5  # include/mppi_hip.h: MPPI_ABI_VERSION
    assert L.mppi_strerror(0) == b"ok"
    assert b"gfx950" in L.mppi_strerror(capi.ERR_NO_DEVICE)


def test_create_validates_arguments(L):
    cfg = S.make_config(128, 50)
    c = capi.make_config_struct(cfg)
    h = C.c_void_p()
    assert L.mppi_create(None, C.byref(h)) == capi.ERR_INVALID
    bad = capi.make_config_struct(dict(cfg, K=100))  # not a multiple of 64 (Q11)
    assert L.mppi_create(C.byref(bad), C.byref(h)) == capi.ERR_INVALID
    bad = capi.make_config_struct(dict(cfg, layers=[5, 32, 32, 4]))
    assert L.mppi_create(C.byref(bad), C.byref(h)) == capi.ERR_INVALID
    if L.mppi_device_count() == 0:
        # the product path must fail loudly without a gfx950 device: no CPU fallback
        assert L.mppi_create(C.byref(c), C.byref(h)) == capi.ERR_NO_DEVICE
        assert not h.value
        with pytest.raises(capi.MppiError):
            capi.Solver(cfg)


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under autorally_amd/ may import or link it."""
    pkg = os.path.join(ROOT, "autorally_amd")
    for d, _, files in os.walk(pkg):
        if os.path.basename(d) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "mppi_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_create_rejects_malformed_configs_without_crashing(L):
    """mppi_create validates before it touches the device: every malformed configuration is refused with
    MPPI_ERR_INVALID (never a crash, never a handle), with or without a GPU in the machine."""
    cfg = S.make_config(128, 50)
    rng = np.random.RandomState(7)
    bad_fields = [
        dict(K=0), dict(K=-64), dict(K=96), dict(K=65), dict(T=1), dict(T=0), dict(T=-5), dict(hz=0), dict(hz=-50),
        dict(num_iters=0), dict(opt_stride=-1),
        dict(layers=[6, 4, 32, 32, 32, 32, 32, 32, 4]),  # more than MPPI_MAX_LAYERS entries cannot even be expressed: truncated below
        dict(layers=[6]), dict(layers=[6, 0, 4]), dict(layers=[6, 300, 4]), dict(layers=[7, 32, 4]), dict(layers=[6, 32, 5]),
        dict(K=1 << 20, T=1 << 12),                      # K*T above the supported product
    ]
    for over in bad_fields:
        c = dict(cfg, **over)
        if len(c["layers"]) > 8:
            c["layers"] = c["layers"][:8]  # ends in 32, not 4: still malformed
        st = capi.make_config_struct(c)
        h = C.c_void_p()
        assert L.mppi_create(C.byref(st), C.byref(h)) == capi.ERR_INVALID, over
        assert not h.value
    # random garbage in every integer field
    for _ in range(200):
        st = capi.make_config_struct(cfg)
        for name in ("num_rollouts", "num_timesteps", "hz", "optimization_stride", "num_iters", "n_layers"):
            if rng.rand() < 0.5:
                setattr(st, name, int(rng.randint(-(1 << 31), (1 << 31) - 1)))
        for i in range(8):
            if rng.rand() < 0.3:
                st.layers[i] = int(rng.randint(-1000, 1000))
        h = C.c_void_p()
        rc = L.mppi_create(C.byref(st), C.byref(h))
        assert rc in (capi.ERR_INVALID, capi.ERR_NO_DEVICE, capi.OK)
        if rc == capi.OK:          # a GPU is present and the draw happened to be valid
            assert L.mppi_destroy(h) == capi.OK
        else:
            assert not h.value
    # every other entry point refuses a null handle
    assert L.mppi_compute_control(None, None) == capi.ERR_INVALID
    assert L.mppi_set_bf_params(None, None, 0) == capi.ERR_INVALID
    assert L.mppi_compute_feedback_gains(None, None, None, None) == capi.ERR_INVALID
    assert L.mppi_debug_cost_raster(None, 0.0, 0.0, 0.0, 1, 1, 1, None, 0) == capi.ERR_INVALID
