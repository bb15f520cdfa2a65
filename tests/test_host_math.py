"""Host arithmetic that must equal the C library's (CPU only).

csrc/tanhf_vec.hpp restates glibc's tanhf (fdlibm s_tanhf.c / s_expm1f.c) on eight AVX2 lanes for the host replays
of the network (computeNominalTraj, the DDP forward pass); those replays must not change by a bit, so the vector
form has to return libm's own bits for every input.  tools/host/tanhf_exhaustive compares all 2^32 bit patterns
(stride 1, ~15-25 s on 8 cores: run it after touching the header); here every 5th pattern plus the library's
start-up self-check."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vector_tanhf_returns_libm_bits(tmp_path):
    exe = str(tmp_path / "tanhf_exhaustive")
    subprocess.check_call(["g++", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp",
                           os.path.join(ROOT, "tools", "host", "tanhf_exhaustive.cpp"), "-o", exe])
    r = subprocess.run([exe, "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 differ" in r.stdout and "selfcheck ok" in r.stdout, r.stdout
    n = int(r.stdout.split(":")[1].split("inputs")[0])
    assert n >= (1 << 32) // 5
