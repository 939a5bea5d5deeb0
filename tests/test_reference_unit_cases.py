"""The reference's own unit tests for the hot-path classes (src/test_squelch.cpp:51-281, src/test_ctcss.cpp:122-155),
re-expressed as plain asserts (GoogleTest cannot be fetched here) and run against the CPU oracle -- and against the
real reference classes too where oracle/_ref exists.  These are behavioural pins, not numeric golden vectors."""
import ctypes as C

import numpy as np
import pytest

import pyoracle
import pyref

NO_SIGNAL, SIGNAL = np.float32(0.05), np.float32(0.75)
STANDARD_TONES = [67.0, 69.3, 71.9, 74.4, 77.0, 79.7, 82.5, 85.4, 88.5, 91.5, 94.8, 97.4, 100.0, 103.5, 107.2, 110.9, 114.8, 118.8, 123.0, 127.3, 131.8, 136.5, 141.3,
                  146.2, 150.0, 151.4, 156.7, 159.8, 162.2, 165.5, 167.9, 171.3, 173.8, 177.3, 179.9, 183.5, 186.2, 189.9, 192.8, 196.6, 199.5, 203.5, 206.5, 210.7,
                  218.1, 225.7, 229.1, 233.6, 241.8, 250.3, 254.1]


class Sq:
    """Uniform driver over the oracle's and the reference's Squelch."""

    def __init__(self, impl, ctcss=0.0):
        self.impl = impl
        if impl == "oracle":
            self.L = pyoracle.lib()
            self.p = self.L.orc_squelch_new(-1.0, 0, ctcss, 8000, 512)
        else:
            self.L = pyref.load_units(False)
            self.p = self.L.refh_squelch_new(-1.0, 0, ctcss)

    def raw(self, value, n=1):
        x = np.full(n, value, np.float32)
        f, noise, lvl = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32)
        fn = self.L.orc_squelch_raw if self.impl == "oracle" else self.L.refh_squelch_raw
        fn(self.p, x.ctypes.data, n, f.ctypes.data, noise.ctypes.data, lvl.ctypes.data)
        return f, noise, lvl

    def audio_raw(self, raw_value, audio):
        n = len(audio)
        x = np.full(n, raw_value, np.float32)
        f = np.zeros(n, np.uint8)
        fn = self.L.orc_squelch_audio_raw if self.impl == "oracle" else self.L.refh_squelch_audio_raw
        fn(self.p, x.ctypes.data, np.ascontiguousarray(audio, np.float32).ctypes.data, n, f.ctypes.data)
        return f

    def counts(self):
        c = np.zeros(4, np.uint64)
        (self.L.orc_squelch_counts if self.impl == "oracle" else self.L.refh_squelch_counts)(self.p, c.ctypes.data)
        return dict(open=int(c[0]), flappy=int(c[1]), ctcss=int(c[2]), no_ctcss=int(c[3]))

    def settle_noise_floor(self):
        """send_samples_for_noise_floor(): feed the no-signal level until the floor is within 1 % of it."""
        for _ in range(100000):
            f, noise, lvl = self.raw(NO_SIGNAL, 16)
            if noise[-1] <= 1.01 * NO_SIGNAL:
                assert SIGNAL > lvl[-1]
                return
        raise AssertionError("noise floor never settled")


IMPLS = ["oracle"] + (["reference"] if pyref.have_ref(False) else [])


@pytest.mark.parametrize("impl", IMPLS)
def test_noise_floor_decays_monotonically(built, impl):
    s = Sq(impl)
    assert s.counts()["open"] == 0
    _, noise, _ = s.raw(NO_SIGNAL, 1)
    assert noise[0] > 10 * NO_SIGNAL
    last = noise[0]
    for _ in range(10000):
        _, noise, _ = s.raw(NO_SIGNAL, 25)
        assert noise[-1] <= last
        if noise[-1] == last:
            break
        last = noise[-1]
    assert last < 1.01 * NO_SIGNAL


@pytest.mark.parametrize("impl", IMPLS)
def test_normal_operation_open_hold_close(built, impl):
    s = Sq(impl)
    s.settle_noise_floor()
    f, _, _ = s.raw(SIGNAL, 500)
    assert (f & 1).any(), "squelch must open within 500 signal samples"
    first_open = int(np.argmax(f & 1))
    assert not (f[:first_open] & 2).any() and (f[first_open] & 2), "should_process_audio turns true exactly when the squelch opens"
    f, _, _ = s.raw(SIGNAL, 1000)
    assert (f & 1).all() and (f & 2).all()
    f, _, _ = s.raw(NO_SIGNAL, 100)
    assert not (f[-1] & 1) and not (f[-1] & 2), "squelch must close within 100 no-signal samples"
    closed_at = int(np.argmin(f & 1))
    assert (f[:closed_at] & 2).all()


@pytest.mark.parametrize("impl", IMPLS)
def test_dead_spot_keeps_squelch_open(built, impl):
    s = Sq(impl)
    s.settle_noise_floor()
    s.raw(SIGNAL, 500)
    f, _, _ = s.raw(SIGNAL, 1000)
    assert (f & 1).all()
    f, _, _ = s.raw(NO_SIGNAL, 50)
    assert (f & 1).all() and (f & 2).all()
    f, _, _ = s.raw(SIGNAL, 1000)
    assert (f & 1).all() and (f & 2).all()


def _tone(freq, n, start=1, ampl=0.2, rate=8000.0):
    t = np.arange(start, start + n, dtype=np.float64)
    return (ampl * np.sin(2 * np.pi * t * freq / rate)).astype(np.float32)


@pytest.mark.parametrize("impl", IMPLS)
def test_squelch_ctcss_good_wrong_close(built, impl):
    # good tone: opens and stays open for 100 000 samples, only "found" windows
    s = Sq(impl, ctcss=STANDARD_TONES[5])
    s.settle_noise_floor()
    f, _, _ = s.raw(SIGNAL, 500)
    assert (f & 2).any() and not (f & 1).any(), "audio is processed but the gate waits for the tone"
    audio = _tone(STANDARD_TONES[5], 100500)
    f = s.audio_raw(SIGNAL, audio)
    opened = int(np.argmax(f & 1))
    assert (f & 1).any() and opened < 500 and (f[opened:] & 1).all()
    c = s.counts()
    assert c["ctcss"] > 0 and c["no_ctcss"] == 0
    # wrong tone: never opens
    s = Sq(impl, ctcss=STANDARD_TONES[7])
    s.settle_noise_floor()
    s.raw(SIGNAL, 500)
    f = s.audio_raw(SIGNAL, _tone(STANDARD_TONES[0], 100000))
    assert (f & 2).all() and not (f & 1).any()
    c = s.counts()
    assert c["ctcss"] == 0 and c["no_ctcss"] > 0
    # close tone: the fast detector lets it through, the slow one shuts it again within 3000 samples, for good
    s = Sq(impl, ctcss=STANDARD_TONES[7])
    s.settle_noise_floor()
    s.raw(SIGNAL, 500)
    f = s.audio_raw(SIGNAL, _tone(STANDARD_TONES[5], 103500))
    opened = int(np.argmax(f & 1))
    assert (f & 1).any() and opened < 500
    closed = opened + int(np.argmin(f[opened:] & 1))
    assert closed - opened < 3000 and not (f[closed:] & 1).any()
    c = s.counts()
    assert c["ctcss"] == 0 and c["no_ctcss"] > 0


def _ctcss_run(impl, tone, rate, window, x):
    h = np.zeros(len(x), np.uint8)
    cnt = np.zeros(2, np.uint64)
    if impl == "oracle":
        pyoracle.lib().orc_ctcss_run(tone, rate, window, x.ctypes.data, len(x), h.ctypes.data, cnt.ctypes.data)
    else:
        pyref.load_units(False).refh_ctcss_run(tone, rate, window, x.ctypes.data, len(x), h.ctypes.data, cnt.ctypes.data)
    return h


@pytest.mark.parametrize("impl", IMPLS)
def test_ctcss_each_standard_tone_detected_as_itself_only(built, impl):
    """src/test_ctcss.cpp:122-155: 8 kHz, 0.4 s window, tone 0.2 + noise 0.2 * N(0, 0.1)."""
    rate, window = 8000.0, 3200
    rng = np.random.default_rng(2024)
    for tone in STANDARD_TONES[::3] + [(STANDARD_TONES[0] + STANDARD_TONES[0]) / 2]:
        x = _tone(tone, window) + (0.2 * 0.1 * rng.standard_normal(window)).astype(np.float32)
        h = _ctcss_run(impl, tone, rate, window, x)
        assert h[-1] & 2 and h[-1] & 1, "tone %.1f not found" % tone
        for other in STANDARD_TONES:
            if abs(other - tone) < 5:
                continue
            h = _ctcss_run(impl, other, rate, window, x)
            assert h[-1] & 2 and not (h[-1] & 1), "detector for %.1f fired on %.1f" % (other, tone)
    # no signal at all: no detector fires
    x = np.zeros(window, np.float32)
    for other in STANDARD_TONES[::5]:
        assert not (_ctcss_run(impl, other, rate, window, x)[-1] & 1)
