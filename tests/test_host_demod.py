"""The stage-2 kernel SOURCE (csrc/demod.hip) compiled as plain C++ for the host -- one lane per "wavefront", through the shim in
tests/hostshim/ -- and run on the oracle's stage-1 output: squelch trace, axcindicate, audio and statistics must equal the oracle's,
bit for bit, batch after batch.  Covers the lane-per-channel kinds that have no cross-lane step (AM, NFM, NFM + lowpass; notch, manual
squelch, quadri discriminator, de-emphasis).  CPU only: it checks the ARITHMETIC and the bookkeeping (ring rotation, tail copy, zero rows
left alone, runs flushed through the staging columns) of the code the GPU runs; the GPU parity tests check the kernels themselves, and
are the only ones that reach the CTCSS chain and the channelizer.  Test infrastructure: nothing here is a code path of the library.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

import helpers
import pyoracle

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "rtlsdr-airband_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"  # csrc/demod.hip uses clang's ext_vector_type

pkg = importlib.import_module("rtlsdr-airband_amd")
capi = pkg.capi
sg = pkg.siggen


@pytest.fixture(scope="module")
def hostdemod(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ in this image")
    out = str(tmp_path_factory.mktemp("hostdemod") / "libhostdemod.so")
    # AIRBAND_HOST_DEFINES="-DAB_..." compiles an experiment variant of the kernel source (e.g. -DAB_MASKED_DELAY) instead of the product's
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-I" + os.path.join(HERE, "hostshim"),
           "-I" + os.path.join(REPO, "include")] + os.environ.get("AIRBAND_HOST_DEFINES", "").split() + [
           "-o", out, os.path.join(HERE, "host_demod_harness.cpp"), os.path.join(CSRC, "params.cpp")]
    subprocess.run(cmd, check=True)
    lib = C.CDLL(out)
    vp = C.c_void_p
    lib.hostdemod_create.argtypes = [C.POINTER(capi.Config), C.c_int, C.POINTER(vp)]
    lib.hostdemod_destroy.argtypes = [vp]
    lib.hostdemod_destroy.restype = None
    lib.hostdemod_wave_batch.argtypes = [vp]
    lib.hostdemod_process_bins.argtypes = [vp, vp, vp]
    lib.hostdemod_collect.argtypes = [vp, vp, vp, vp]
    lib.hostdemod_collect.restype = None
    lib.hostdemod_stats.argtypes = [vp, vp]
    lib.hostdemod_stats.restype = None
    return lib


class HostDemod:
    def __init__(self, lib, devices, wave_rate, fm_demod=0):
        self.lib = lib
        cfg, self._keep = pkg.make_config(devices, wave_rate=wave_rate, fm_demod=fm_demod)
        self.h = C.c_void_p()
        rc = lib.hostdemod_create(C.byref(cfg), 1, C.byref(self.h))
        assert rc == 0, rc
        self.B = lib.hostdemod_wave_batch(self.h)
        self.n = sum(len(d["channels"]) for d in devices)

    def process_bins(self, wavein, iqin):
        wavein = np.ascontiguousarray(wavein, np.float32)
        iqin = np.ascontiguousarray(iqin, np.float32)
        assert wavein.shape == (self.n, self.B) and iqin.shape == (self.n, 2 * self.B)
        assert self.lib.hostdemod_process_bins(self.h, wavein.ctypes.data, iqin.ctypes.data) == 0

    def collect(self):
        wave = np.zeros((self.n, self.B), np.float32)
        axc = np.zeros((self.n,), np.uint8)
        trace = np.zeros((self.n, self.B), np.uint8)
        self.lib.hostdemod_collect(self.h, wave.ctypes.data, axc.ctypes.data, trace.ctypes.data)
        return wave, axc, trace

    def stats(self):
        st = (capi.ChannelStats * self.n)()
        self.lib.hostdemod_stats(self.h, C.cast(st, C.c_void_p))
        return [{f[0]: getattr(s, f[0]) for f in capi.ChannelStats._fields_} for s in st]

    def close(self):
        if self.h:
            self.lib.hostdemod_destroy(self.h)
            self.h = None


def _plan(variant):
    """(devices, carriers, wave_rate, fm_demod): plans of the kinds the harness can run."""
    if variant == "am":
        chans, carriers = sg.baseline_plan(mixed=False)
        chans[1]["notch_freq"], chans[1]["notch_q"] = 1000.0, 5.0
        chans[2]["squelch_threshold_dbfs"] = -40
        chans[3]["squelch_snr_threshold_db"] = 6.0
        chans[5]["ampfactor"] = 2.5
        return chans, carriers, 8000, 0
    chans, carriers = sg.baseline_plan(mixed=True)
    for c, ch in enumerate(chans):  # the CTCSS channels of the mixed plan become plain NFM ones: the tone kernel is not a lane-per-channel kernel
        if ch["ctcss_freq"]:
            ch["ctcss_freq"] = 0.0
            ch["notch_freq"] = 0.0
    chans[1]["tau_us"] = 0 if variant == "nfm_quadri" else 100  # NFM, de-emphasis off / on
    chans[5]["notch_freq"], chans[5]["notch_q"] = 150.0, 8.0     # NFM + notch
    chans[7]["bandwidth_hz"] = 6250                              # a second lowpass gain
    chans[0]["squelch_threshold_dbfs"] = -35                     # AM, manual squelch
    return chans, carriers, 16000, (1 if variant == "nfm_quadri" else 0)


def _bursty(carriers):
    out = []
    for k, c in enumerate(carriers):
        period, on = [(0.11, 0.045), (0.31, 0.02), (0.26, 0.19), (0.07, 0.05)][k % 4]
        amp = [0.08, 0.03, 0.05, 0.012][(k // 2) % 4]
        out.append(sg.make_carrier(sg.PLAN_OFFSETS_HZ[k], sg.SAMPLE_RATE, amplitude=amp, kind=c.kind, key_slot=k, key_period_s=period, key_on_s=on, key_slot_s=0.013))
    return out


@pytest.mark.parametrize("style", ["keyed", "bursty"])
@pytest.mark.parametrize("variant", ["am", "nfm_atan2", "nfm_quadri"])
def test_kernel_source_on_the_host_equals_the_oracle(hostdemod, variant, style):
    chans, carriers, wave_rate, fm_demod = _plan(variant)
    if style == "bursty":
        carriers = _bursty(carriers)
    n_dev, n_batches = 2, 7
    devices = [dict(channels=[dict(c) for c in chans]) for _ in range(n_dev)]
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    src = pyoracle.Oracle(devices, wave_rate=wave_rate, fm_demod=fm_demod)
    raw = [src.run_device(d, sg.generate_u8(d, 0, nbytes // 2, carriers), n_batches) for d in range(n_dev)]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fm_demod=fm_demod)
    hd = HostDemod(hostdemod, devices, wave_rate, fm_demod)
    try:
        opened = 0
        for b in range(n_batches):
            wavein = np.concatenate([r["raw_wavein"][b] for r in raw])
            iqin = np.concatenate([r["raw_iq"][b] for r in raw])
            want = [orc.run_bins(d, raw[d]["raw_wavein"][b], raw[d]["raw_iq"][b]) for d in range(n_dev)]
            hd.process_bins(wavein, iqin)
            wave, axc, trace = hd.collect()
            assert np.array_equal(trace, np.concatenate([w["trace"] for w in want])), "batch %d: squelch trace" % b
            assert np.array_equal(axc, np.concatenate([w["axc"] for w in want])), "batch %d: axc" % b
            ww = np.concatenate([w["waveout"] for w in want])
            assert np.array_equal(wave.view(np.uint32), ww.view(np.uint32)), "batch %d: waveout, max diff %g" % (b, np.abs(wave - ww).max())
            opened += int((axc != ord(" ")).sum())
        assert opened > 0
        st = hd.stats()
        k = 0
        for d in range(n_dev):
            for j in range(len(chans)):
                o = orc.stats(d, j)
                for f in ("noise_level", "signal_level", "squelch_level", "agcavgfast", "open_count", "flappy_count", "active_counter", "squelch_state", "signal_outside_filter"):
                    assert o[f] == st[k][f], (d, j, f, o[f], st[k][f])
                k += 1
    finally:
        hd.close()
        src.close()
        orc.close()


def test_harness_refuses_the_kinds_it_cannot_run(hostdemod):
    chans, _ = sg.baseline_plan(mixed=True)  # has CTCSS channels
    cfg, keep = pkg.make_config([dict(channels=chans)], wave_rate=16000)
    h = C.c_void_p()
    assert hostdemod.hostdemod_create(C.byref(cfg), 0, C.byref(h)) == -100


def _fuzz_streams(rng, n_ch, B, n_batches, nfm, dense=False):
    """Stage-1 outputs made up directly (no channelizer): per channel a noise floor, keyed bursts of random strength and length, stretches of exact
    zeros, of values whose squares underflow, and of large values -- the seams of the kernels' short sqrt / division sequences and of the squelch."""
    n = B * n_batches
    wave = np.zeros((n_ch, n), np.float32)
    iq = np.zeros((n_ch, 2 * n), np.float32)
    for c in range(n_ch):
        floor = float(10.0 ** rng.uniform(-3.5, -1.0))
        env = np.full(n, floor, np.float64) * (1.0 + 0.3 * rng.standard_normal(n))
        t = 0
        while t < n:
            gap, on = (int(rng.integers(20, 500)), int(rng.integers(20, 600))) if dense else (int(rng.integers(50, 3000)), int(rng.integers(5, 2500)))
            t += gap
            env[t:t + on] += floor * float(10.0 ** rng.uniform(-0.3, 1.8)) * (1.0 + 0.15 * rng.standard_normal(min(on, max(0, n - t))))
            if dense and rng.random() < 0.5:  # a step inside the transmission: the delayed and the current level differ for ~100 samples
                a = t + int(rng.integers(0, on))
                env[a:t + on] *= float(rng.choice([0.3, 0.6, 1.7, 4.0]))
            t += on
        for _ in range(3):  # the odd stretches
            a, ln = int(rng.integers(0, n - 400)), int(rng.integers(1, 400))
            env[a:a + ln] = [0.0, 1e-25, 3e5, floor * 1e-6][int(rng.integers(0, 4))]
        env = np.abs(env)
        if nfm[c]:
            ph = np.cumsum(rng.normal(0.0, 0.4, n)) + 2 * np.pi * rng.uniform(-0.2, 0.2) * np.arange(n)
            re = (env * np.cos(ph)).astype(np.float32)
            im = (env * np.sin(ph)).astype(np.float32)
            iq[c, 0::2], iq[c, 1::2] = re, im
            wave[c] = np.sqrt(re * re + im * im)  # float32 throughout: what stage 1 hands over (src/rtl_airband.cpp:484-487)
        else:
            wave[c] = env.astype(np.float32)
    return wave, iq


def _aim_at_boundaries(rng, make_oracle, wave, iq, B, n_batches):
    """Moves every channel's stream in time so that one of its squelch transitions (found with a scratch oracle; into OPEN by preference) lands on or within
    three samples of a batch boundary, and -- half of the time -- puts a step in the level ~101 samples before that boundary (what the squelch's delay line
    hands the post-filter gate there).  A random stream puts a transition on a boundary once in WAVE_BATCH tries; the state a kernel carries from batch to
    batch is looked at on exactly those samples (the round-4 delay-line defect sat there, and thousands of unaimed seeds did not see it)."""
    n_ch, n = wave.shape

    def states():
        probe = make_oracle()
        try:
            return np.concatenate([probe.run_bins(0, wave[:, b * B:(b + 1) * B], iq[:, 2 * b * B:2 * (b + 1) * B])["trace"] for b in range(n_batches)], axis=1) & 7
        finally:
            probe.close()

    def shift(c, by):
        wave[c] = np.roll(wave[c], by)
        iq[c] = np.roll(iq[c], 2 * by)

    st = states()
    aim = {}
    for c in range(n_ch):
        tr = np.nonzero(st[c, 1:] != st[c, :-1])[0] + 1
        tr = tr[tr > 150]
        if len(tr) == 0:
            continue
        opens = tr[st[c, tr] == 4]
        t = int(rng.choice(opens if len(opens) and rng.random() < 0.7 else tr))
        at = int(rng.integers(1, n_batches)) * B + (0 if rng.random() < 0.5 else int(rng.integers(-3, 4)))
        aim[c] = (at, int(st[c, t]))
        shift(c, at - t)
        if rng.random() < 0.5:
            a = at - 101 + int(rng.integers(-2, 3))
            ln = int(rng.integers(1, 40))
            f = np.float32(rng.choice([0.02, 0.2, 5.0, 40.0]))
            wave[c, a:a + ln] *= f
            iq[c, 2 * a:2 * (a + ln)] *= f
    for _ in range(2):  # the moved stream has another past: the transition may have moved by a few samples; follow it
        st = states()
        for c, (at, to) in aim.items():
            tr = np.nonzero((st[c, 1:] != st[c, :-1]) & (st[c, 1:] == to))[0] + 1
            if len(tr):
                t = int(tr[np.argmin(np.abs(tr - at))])
                if t != at and abs(t - at) < 200:
                    shift(c, at - t)
    return wave, iq


@pytest.mark.parametrize("seed", range(int(os.environ.get("AIRBAND_FUZZ_SEEDS", "12"))))
def test_random_plans_and_made_up_stage1_output(hostdemod, seed):
    """Random channel settings (squelch mode and thresholds, notch, lowpass bandwidth, de-emphasis, amplification, discriminator) on made-up stage-1
    output with awkward values; the host-compiled kernel source against the oracle, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    nfm_build = bool(seed % 3)  # WAVE_RATE 16000 builds carry NFM channels
    wave_rate = 16000 if nfm_build else 8000
    fm_demod = int(rng.integers(0, 2)) if nfm_build else 0
    chans = []
    for k, off in enumerate(sg.PLAN_OFFSETS_HZ):
        c = dict(frequency=sg.CENTERFREQ + off, modulation=0, afc=0, squelch_threshold_dbfs=0, squelch_snr_threshold_db=-1.0, notch_freq=0.0, notch_q=0.0, ctcss_freq=0.0,
                 bandwidth_hz=0, ampfactor=1.0, tau_us=-1, has_iq_outputs=0)
        if nfm_build and rng.random() < 0.6:
            c["modulation"] = 1
            if rng.random() < 0.5:
                c["bandwidth_hz"] = int(rng.choice([5000, 6250, 12500, 25000]))
            c["tau_us"] = int(rng.choice([-1, 0, 50, 200, 750]))
        mode = rng.random()
        if mode < 0.3:
            c["squelch_threshold_dbfs"] = int(rng.integers(-60, -10))
        elif mode < 0.6:
            c["squelch_snr_threshold_db"] = float(rng.choice([3.0, 6.0, 9.5, 14.0]))
        if rng.random() < 0.3:
            c["notch_freq"], c["notch_q"] = float(rng.choice([100.0, 150.0, 1000.0])), float(rng.choice([0.0, 4.0, 10.0]))
        if rng.random() < 0.3:
            c["ampfactor"] = float(rng.choice([0.25, 2.0, 8.0]))
        chans.append(c)
    devices = [dict(channels=chans)]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fm_demod=fm_demod)
    hd = HostDemod(hostdemod, devices, wave_rate, fm_demod)
    try:
        B, n_batches = hd.B, 4
        wave, iq = _fuzz_streams(rng, len(chans), B, n_batches, [c["modulation"] == 1 for c in chans], dense=seed % 4 >= 2)  # dense: a transition every few hundred samples
        if seed % 2:  # every other seed with its transitions aimed at the batch boundaries
            wave, iq = _aim_at_boundaries(rng, lambda: pyoracle.Oracle(devices, wave_rate=wave_rate, fm_demod=fm_demod), wave, iq, B, n_batches)
        for b in range(n_batches):
            w, q = wave[:, b * B:(b + 1) * B], iq[:, 2 * b * B:2 * (b + 1) * B]
            want = orc.run_bins(0, w, q)
            hd.process_bins(w, q)
            got_w, got_a, got_t = hd.collect()
            assert np.array_equal(got_t, want["trace"]), "seed %d batch %d: squelch trace (channels %s)" % (seed, b, np.nonzero((got_t != want["trace"]).any(axis=1))[0])
            assert np.array_equal(got_a, want["axc"]), "seed %d batch %d: axc" % (seed, b)
            same = (got_w.view(np.uint32) == want["waveout"].view(np.uint32)) | (np.isnan(got_w) & np.isnan(want["waveout"]))
            assert same.all(), "seed %d batch %d: waveout (channels %s)" % (seed, b, np.nonzero((~same).any(axis=1))[0])
    finally:
        hd.close()
        orc.close()


def test_opening_timer_expires_on_the_first_sample_of_a_batch_with_the_post_filter_in_use(hostdemod):
    """NFM + lowpass channels whose OPENING delay runs out on sample 0 of a batch while the post-filter gate is what decides
    (src/squelch.cpp:381-398,467-475): update_current_state() tests post_filter_.capped_ against buffer_[buffer_tail_] BEFORE the tail moves --
    the entry pushed 102 samples earlier -- and the kernel, which recomputes the delay line (SqShadow), has to carry that very value across the
    batch boundary.  The entry one sample later is made much larger (a step in the level 101 samples before the boundary); with the later entry
    the gate closes channels that the reference opens."""
    wave_rate, B, n_dev = 16000, 2000, 8
    devices, level = helpers.boundary_devices(n_dev)
    wave, iq = helpers.boundary_streams(lambda: pyoracle.Oracle(devices, wave_rate=wave_rate), n_dev, B, level)
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    hd = HostDemod(hostdemod, devices, wave_rate, 0)
    try:
        assert hd.B == B
        outcomes = set()
        for b in range(3):
            w, q = wave[:, b * B:(b + 1) * B], iq[:, 2 * b * B:2 * (b + 1) * B]
            want = [orc.run_bins(d, w[8 * d:8 * d + 8], q[8 * d:8 * d + 8]) for d in range(n_dev)]
            want_t, want_a, want_w = (np.concatenate([r[k] for r in want]) for k in ("trace", "axc", "waveout"))
            hd.process_bins(w, q)
            got_w, got_a, got_t = hd.collect()
            if b == 2:
                outcomes = {int(t[1]) & 7 for t in want_t}
            assert np.array_equal(got_t, want_t), "batch %d: squelch trace (channels %s)" % (b, np.nonzero((got_t != want_t).any(axis=1))[0])
            assert np.array_equal(got_a, want_a)
            assert np.array_equal(got_w.view(np.uint32), want_w.view(np.uint32))
            st = hd.stats()
            for q in range(n_dev * 8):  # the TUI's '~' (Squelch::signal_outside_filter) reads the same delay-line entry
                o = orc.stats(q // 8, q % 8)
                assert (st[q]["signal_outside_filter"], st[q]["squelch_state"]) == (o["signal_outside_filter"], o["squelch_state"]), (b, q)
        assert orc.stats(0, 0)["squelch_level"] == level
        assert 4 in outcomes, "no channel went OPEN at the boundary: the streams no longer exercise the case (%s)" % outcomes
    finally:
        hd.close()
        orc.close()
