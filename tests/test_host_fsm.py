"""The demod kernels' squelch state machine (csrc/squelch_fsm.h: lane masks, wave-uniform rare-event branches), compiled as
plain C++ -- one lane per "wavefront" -- and checked sample by sample against the oracle's Squelch restatement and, where
/root/reference exists, the reference's own Squelch (oracle/_ref).  CPU only: this tests the kernel's LOGIC; the GPU
parity tests (test_gpu_parity.py) test the kernel.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import pyoracle
import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "rtlsdr-airband_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostfsm") / "libhostfsm.so")
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-I" + os.path.join(REPO, "include"), "-o", out,
           os.path.join(HERE, "host_fsm_harness.cpp"), os.path.join(CSRC, "params.cpp")]
    subprocess.run(cmd, check=True)
    lib = C.CDLL(out)
    lib.hostfsm_run.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 5
    lib.hostfsm_run.restype = C.c_int
    return lib


def streams(seed, n, kind):
    """Raw magnitude streams that exercise every transition: keyed carrier, marginal (flapping) signal, short bursts that
    abort on low signal, and a filtered stream that sometimes falls below the delayed pre-filter level."""
    rng = np.random.default_rng(seed)
    noise = np.abs(rng.normal(0.02, 0.005, n)).astype(np.float32)
    sig = np.zeros(n, np.float32)
    t = 0
    while t < n:
        if kind == "keyed":
            on, off = int(rng.integers(300, 3000)), int(rng.integers(300, 3000))
        elif kind == "bursty":
            on, off = int(rng.integers(20, 400)), int(rng.integers(20, 1500))
        else:  # flappy: opens again and again within the 1000-sample window
            on, off = int(rng.integers(250, 500)), int(rng.integers(100, 400))
        amp = float(rng.uniform(0.05, 0.6)) if kind != "marginal" else float(rng.uniform(0.055, 0.075))
        sig[t:t + on] = amp * (1.0 + 0.2 * rng.standard_normal(min(on, n - t))).astype(np.float32)
        t += on + off
    raw = (noise + np.abs(sig)).astype(np.float32)
    # filtered magnitude: mostly ~ the raw one, with stretches where the filter removes most of the energy
    filt = (raw * rng.uniform(0.7, 1.0, n)).astype(np.float32)
    for _ in range(max(1, n // 5000)):
        a = int(rng.integers(0, n - 600))
        filt[a:a + int(rng.integers(100, 600))] *= np.float32(0.05)
    return raw, filt


def run_host(lib, snr, manual, mode, chunk, raw, filt):
    n = len(raw)
    flags = np.zeros(n, np.uint8)
    noise = np.zeros(n, np.float32)
    level = np.zeros(n, np.float32)
    st = np.zeros(8, np.int64)
    cnt = np.zeros(5, np.uint64)
    rc = lib.hostfsm_run(snr, manual, mode, chunk, raw.ctypes.data, filt.ctypes.data, n, flags.ctypes.data, noise.ctypes.data, level.ctypes.data, st.ctypes.data,
                         cnt.ctypes.data)
    assert rc == 0
    return flags, noise, level, st, cnt


def run_oracle(L, prefix, snr, manual, lowpass, raw, filt):
    n = len(raw)
    new = getattr(L, prefix + "_squelch_new")
    p = new(snr, manual, 0.0, 8000, 512) if prefix == "orc" else new(snr, manual, 0.0)
    flags = np.zeros(n, np.uint8)
    noise = np.zeros(n, np.float32)
    level = np.zeros(n, np.float32)
    if lowpass:
        getattr(L, prefix + "_squelch_raw_filtered")(p, raw.ctypes.data, filt.ctypes.data, n, flags.ctypes.data, noise.ctypes.data, level.ctypes.data)
    else:
        getattr(L, prefix + "_squelch_raw")(p, raw.ctypes.data, n, flags.ctypes.data, noise.ctypes.data, level.ctypes.data)
    cnt = np.zeros(4, np.uint64)
    getattr(L, prefix + "_squelch_counts")(p, cnt.ctypes.data)
    getattr(L, prefix + "_squelch_free")(p)
    return flags, noise, level, cnt


CASES = [(-1.0, 0), (6.0, 0), (-1.0, -70), (-1.0, -62)]


@pytest.mark.parametrize("kind", ["keyed", "bursty", "flappy", "marginal"])
@pytest.mark.parametrize("snr,manual", CASES)
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 5])
def test_fsm_matches_oracle(harness, kind, snr, manual, mode):
    L = pyoracle.lib()
    n = 60000
    seed = 1000 * ["keyed", "bursty", "flappy", "marginal"].index(kind) + 100 * mode + 10 * int(abs(snr)) + abs(manual)
    raw, filt = streams(seed, n, kind)
    want = run_oracle(L, "orc", snr, manual, mode in (1, 2, 5), raw, filt)
    for chunk in (0, 1000, 37):
        got = run_host(harness, snr, manual, mode, chunk, raw, filt)
        assert np.array_equal(got[0], want[0]), "flags differ at %d" % int(np.argmax(got[0] != want[0]))
        assert np.array_equal(got[1], want[1])
        assert np.array_equal(got[2], want[2])
        assert got[4][0] == want[3][0] and got[4][1] == want[3][1]
    if kind != "marginal":
        assert want[0].max() > 0  # the squelch did open
    if mode == 3 and kind == "keyed":  # the group path is the common one, also through the timed states: (almost) everything but the transitions themselves
        assert int(run_host(harness, snr, manual, mode, 0, raw, filt)[4][4]) > 0.9 * (n // 4)
    # head/tail bookkeeping: every sample advances both by one (mod 102)
    assert int(got[3][4]) == n % 102 and int(got[3][5]) == (n + 1) % 102


@pytest.mark.parametrize("mode", [0, 1, 3])
def test_fsm_matches_reference(harness, mode):
    if not pyref.have_ref(False):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    R = pyref.load_units(False)
    n = 40000
    for kind in ("keyed", "flappy"):
        raw, filt = streams(7 + mode, n, kind)
        want = run_oracle(R, "refh", -1.0, 0, mode == 1, raw, filt)
        got = run_host(harness, -1.0, 0, mode, 1000, raw, filt)
        assert np.array_equal(got[0], want[0])
        assert np.array_equal(got[1], want[1])
        assert np.array_equal(got[2], want[2])
        assert got[4][0] == want[3][0] and got[4][1] == want[3][1]
