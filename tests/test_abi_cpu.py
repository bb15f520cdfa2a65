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
    assert L.mppi_abi_version() == 1
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
