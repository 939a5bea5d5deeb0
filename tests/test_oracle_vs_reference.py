"""Pins the C restatement (oracle/airband_oracle.c) against the REAL reference compiled in place (oracle/_ref):
whole streams through demodulate() and every stand-alone piece, bit for bit.  Skipped where oracle/_ref is absent."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers
import pyoracle
import pyref

need_ref = pytest.mark.skipif(not (pyref.have_ref(True) and pyref.have_ref(False)), reason="oracle/_ref not built (needs /root/reference)")


def _reference_run(devices, iq_list, n_batches, **kw):
    """pyref.run_reference, again if the harness came back a batch short: its feeding loop ends when the stream is used up and demodulate() has not YET flagged the last batch
    (a scheduling race of the harness, not of what is compared: a complete run is bit-identical every time)."""
    for _ in range(4):
        ref = pyref.run_reference(devices, iq_list, n_batches, **kw)
        if all(r["n_batches"] == n_batches for r in ref):
            break
    return ref


def _tweak(d, ch):
    ch[3]["has_iq_outputs"] = 1
    ch[0]["bandwidth_hz"] = 8000
    ch[2]["squelch_threshold_dbfs"] = -40
    ch[4]["squelch_snr_threshold_db"] = 6.0
    ch[6]["ampfactor"] = 2.5
    ch[7]["tau_us"] = 75
    ch[5]["notch_q"] = 5.0


@need_ref
@pytest.mark.parametrize("mixed,wave_rate,fm_demod", [(False, 8000, 0), (False, 16000, 0), (True, 16000, 0), (True, 16000, 1)])
def test_stream_bit_exact(pkg, built, mixed, wave_rate, fm_demod, tmp_path):
    devices, carriers = helpers.plan_devices(1, mixed, _tweak if mixed else None)
    n_batches = 14
    iq = pkg.siggen.generate_u8(3, 0, helpers.stream_bytes(n_batches, wave_rate) // 2, carriers)
    ref = _reference_run(devices, [iq], n_batches, nfm=wave_rate == 16000, fm_demod=fm_demod, trace_dir=str(tmp_path))[0]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fm_demod=fm_demod)
    got = orc.run_device(0, iq, n_batches)
    assert ref["n_batches"] == got["n_batches"] == n_batches
    assert np.array_equal(ref["waveout"].view(np.uint32), got["waveout"].view(np.uint32))
    assert np.array_equal(ref["iq_out"].view(np.uint32), got["iq_out"].view(np.uint32))
    assert np.array_equal(ref["axc"], got["axc"])
    assert (ref["axc"] == ord("*")).any() and (ref["axc"] == ord(" ")).any()
    B = wave_rate // 8
    for j in range(8):
        a, b = ref["stats"][j], orc.stats(0, j)
        for k in a:
            if k != "squelch_state":
                assert a[k] == b[k], (j, k, a[k], b[k])
        assert ref["consts"][j][0] == orc.constants(0, j)[0]  # bin
        assert ref["consts"][j][1] == orc.constants(0, j)[1]  # dm_dphi
        # per-sample squelch state: the reference's own DEBUG_SQUELCH dump (src/squelch.cpp:593-633) vs our trace byte
        tr = pyref.read_trace(str(tmp_path), 0, j)
        assert len(tr) >= n_batches * B
        mine = got["trace"][:, j, :].reshape(-1)
        # debug_state() runs at the end of update_current_state(), the only place current_state_ changes: record i holds
        # sample i's state; its noise floor / pre-filter columns belong to the previous sample and are not compared
        assert np.array_equal(tr["current_state"][:n_batches * B], (mine & 7).astype(np.intc)), j


# (sample format, fft_size_log, sample rate, WAVE_RATE): every class of configuration the matrix-core channelizer claims (u8 / CS16 /
# s8 at fft 256 ... 8192, hops that are not multiples of 16 bytes) plus f32 -- the restatement's convert x window (src/rtl_airband.cpp:402-455),
# hop (:394) and bin (src/config.cpp:666-667) arithmetic outside u8 / fft 512 / 2.56 MS/s, against the reference itself.
FORMAT_CASES = [("SFMT_S16", 9, 2_560_000, 16000), ("SFMT_U8", 10, 2_560_000, 8000), ("SFMT_U8", 8, 2_560_000, 16000), ("SFMT_S16", 10, 2_560_000, 16000),
                ("SFMT_U8", 9, 2_400_000, 16000), ("SFMT_U8", 9, 2_400_000, 8000), ("SFMT_S8", 9, 2_560_000, 8000), ("SFMT_S8", 10, 2_400_000, 16000),
                ("SFMT_F32", 9, 2_560_000, 16000), ("SFMT_U8", 11, 2_560_000, 16000), ("SFMT_U8", 12, 2_560_000, 8000), ("SFMT_S16", 13, 2_560_000, 8000),
                ("SFMT_U8", 9, 1_024_000, 8000), ("SFMT_S16", 8, 2_048_000, 16000), ("SFMT_U8", 9, 3_200_000, 8000), ("SFMT_S16", 11, 2_400_000, 8000),
                # what tests/test_gpu_wavefront_fft.py adds on the wavefront-FFT path: 2.0 MS/s (hops of 125 samples = 250 bytes), f32 at 2.4 MS/s / fft 1024
                # and f32 at fft 2048 (every GPU case has its oracle configuration pinned to the reference here)
                ("SFMT_U8", 9, 2_000_000, 16000), ("SFMT_F32", 10, 2_400_000, 8000), ("SFMT_F32", 11, 2_560_000, 16000)]


# round 4: the configurations the GPU suite newly runs (CF32 on the float32 matrix pipe, hops of an odd number of samples on the int8 one) -- one dongle,
# four batches each: the arithmetic that differs (convert x window per format, hop, bin) is per hop, not per dongle
FORMAT_CASES_SHORT = [("SFMT_F32", 9, 2_560_000, 8000), ("SFMT_F32", 8, 2_560_000, 16000), ("SFMT_F32", 9, 2_400_000, 16000), ("SFMT_F32", 9, 2_400_000, 8000),
                      ("SFMT_F32", 10, 2_560_000, 16000), ("SFMT_S8", 10, 2_000_000, 16000), ("SFMT_U8", 8, 1_200_000, 16000), ("SFMT_U8", 11, 2_000_000, 16000),
                      ("SFMT_F32", 11, 2_560_000, 8000), ("SFMT_F32", 11, 2_400_000, 16000), ("SFMT_F32", 12, 2_560_000, 16000)]  # round 5: the new GPU cases of CF32 at fft 1024 / 2048 / 4096


@need_ref
@pytest.mark.parametrize("sfmt_name,fft_log,sample_rate,wave_rate,n_dev,n_batches", [c + (2, 6) for c in FORMAT_CASES] + [c + (1, 4) for c in FORMAT_CASES_SHORT])
def test_stream_bit_exact_other_formats(pkg, built, sfmt_name, fft_log, sample_rate, wave_rate, n_dev, n_batches):
    sfmt = getattr(pkg.capi, sfmt_name)
    devices, iq = helpers.format_case(pkg, sfmt, fft_log, sample_rate, wave_rate, n_dev, n_batches, first_dongle=5)
    # one reference process per dongle: demodulate() sleeps 10 ms whenever its round robin meets a device without a full hop
    # (src/rtl_airband.cpp:395-400), so feeding two devices of one instance one after the other would take minutes
    ref = [_reference_run([devices[d]], [iq[d]], n_batches, nfm=wave_rate == 16000, fft_log=fft_log)[0] for d in range(n_dev)]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fft_log=fft_log)
    opened = 0
    for d in range(n_dev):  # dongle 1 of a CS16 case has its own input->fullscale
        got = orc.run_device(d, iq[d], n_batches)
        assert ref[d]["n_batches"] == got["n_batches"] == n_batches
        assert np.array_equal(ref[d]["axc"], got["axc"])
        assert np.array_equal(ref[d]["waveout"].view(np.uint32), got["waveout"].view(np.uint32))
        opened += int((ref[d]["axc"] == ord("*")).sum())
        for j in range(8):
            a, b = ref[d]["stats"][j], orc.stats(d, j)
            for k in a:
                if k != "squelch_state":
                    assert a[k] == b[k], (d, j, k, a[k], b[k])
            assert ref[d]["consts"][j][0] == orc.constants(d, j)[0]
            assert ref[d]["consts"][j][1] == orc.constants(d, j)[1]
    assert opened > 0


@need_ref
@pytest.mark.parametrize("seed", range(int(os.environ.get("AIRBAND_FUZZ_SEEDS_ORACLE", "4"))))
def test_oracle_is_the_reference_on_random_configurations(pkg, built, seed):
    """The GPU fuzz over channelizer configurations (tests/test_gpu_parity.py::test_random_channelizer_configurations) measures the library against the C restatement
    on configurations no fixed list holds -- sample rates like 1.44 / 1.8 / 2.88 MS/s, 1 ... 24 channels per dongle at random frequencies (bins in the upper half, shared
    bins), per-dongle CS16 full scales.  The same generator, the same seeds, here against the REFERENCE itself: audio, axcindicate, statistics and the bin / dm_dphi
    constants of every channel bit for bit, so that the restatement is pinned where the fuzz uses it (AIRBAND_FUZZ_SEEDS_ORACLE=N for more seeds)."""
    import test_gpu_parity
    devices, iq, fft_log, wave_rate, n_batches, _ = test_gpu_parity.random_stage1_case(pkg, seed)
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fft_log=fft_log)
    try:
        for d in range(len(devices)):
            nc = len(devices[d]["channels"])
            ref = _reference_run([devices[d]], [iq[d]], n_batches, nfm=wave_rate == 16000, fft_log=fft_log)[0]
            got = orc.run_device(d, iq[d], n_batches)
            what = "seed %d dongle %d (sfmt %d, fft %d, %d S/s, %d channels)" % (seed, d, devices[d]["sfmt"], 1 << fft_log, devices[d]["sample_rate"], nc)
            assert ref["n_batches"] == got["n_batches"] == n_batches, what
            assert np.array_equal(ref["axc"], got["axc"]), what
            assert np.array_equal(ref["waveout"].view(np.uint32), got["waveout"].view(np.uint32)), what
            for j in range(nc):
                a, b = ref["stats"][j], orc.stats(d, j)
                for k in a:
                    if k != "squelch_state":
                        assert a[k] == b[k], (what, j, k, a[k], b[k])
                assert ref["consts"][j][0] == orc.constants(d, j)[0] and ref["consts"][j][1] == orc.constants(d, j)[1], (what, j)
    finally:
        orc.close()


def random_plan_case(pkg, seed):
    """(device, carriers, wave_rate, fm_demod, n_batches): ONE dongle with a random plan over every kind -- CTCSS on FM and AM channels, lowpass with and without CTCSS,
    notch, manual / SNR squelch, de-emphasis, amplification, raw-I/Q outputs -- and transmitters for it: keyed in random rhythms and strengths (some too weak to open a
    squelch, some that make it flap), sub-tones that are right, a neighbouring standard tone, or absent."""
    sg = pkg.siggen
    rng = np.random.default_rng(61_000 + seed)
    nfm_build = bool(seed % 4)
    wave_rate = 16000 if nfm_build else 8000
    fm_demod = int(rng.integers(0, 2)) if nfm_build else 0
    chans, carriers = [], []
    for k, off in enumerate(sg.PLAN_OFFSETS_HZ):
        c = dict(frequency=sg.CENTERFREQ + off, modulation=0, afc=0, squelch_threshold_dbfs=0, squelch_snr_threshold_db=-1.0, notch_freq=0.0, notch_q=0.0, ctcss_freq=0.0,
                 bandwidth_hz=0, ampfactor=1.0, tau_us=-1, has_iq_outputs=0)
        if nfm_build and rng.random() < 0.6:
            c["modulation"] = 1
            c["tau_us"] = int(rng.choice([-1, 0, 50, 200, 750]))
        if rng.random() < 0.3:
            c["bandwidth_hz"] = int(rng.choice([5000, 6250, 12500, 25000]))
        if rng.random() < 0.4:
            c["ctcss_freq"] = float(rng.choice([67.0, 100.0, 123.0, 254.1]))
        mode = rng.random()
        if mode < 0.3:
            c["squelch_threshold_dbfs"] = int(rng.integers(-60, -25))
        elif mode < 0.6:
            c["squelch_snr_threshold_db"] = float(rng.choice([3.0, 6.0, 9.5, 14.0]))
        if rng.random() < 0.3:
            c["notch_freq"], c["notch_q"] = float(rng.choice([100.0, 150.0, 1000.0])), float(rng.choice([0.0, 4.0, 10.0]))
        if rng.random() < 0.3:
            c["ampfactor"] = float(rng.choice([0.25, 2.0, 8.0]))
        if rng.random() < 0.15:
            c["has_iq_outputs"] = 1
        chans.append(c)
        tone = rng.random()
        ct = c["ctcss_freq"] if tone < 0.6 else (c["ctcss_freq"] * 1.035 if tone < 0.8 else 0.0)
        period, on = [(1.5, 0.75), (0.5, 0.3), (0.11, 0.045), (0.31, 0.02), (0.26, 0.19), (2.0, 1.7)][int(rng.integers(0, 6))]
        carriers.append(sg.make_carrier(off, sg.SAMPLE_RATE, amplitude=float(rng.choice([0.08, 0.05, 0.03, 0.012, 0.004])), kind=c["modulation"],
                                        ctcss_hz=ct if c["modulation"] == 1 else 0.0, key_slot=k, key_period_s=period, key_on_s=on, key_slot_s=float(rng.choice([0.125, 0.04, 0.013]))))
    return dict(channels=chans), carriers, wave_rate, fm_demod, (8 if seed % 3 == 0 else 4)


@need_ref
@pytest.mark.parametrize("seed", range(int(os.environ.get("AIRBAND_FUZZ_SEEDS_ORACLE_PLANS", "4"))))
def test_oracle_is_the_reference_on_random_plans(pkg, built, seed):
    """The stage-2 fuzz (tests/test_host_demod.py, test_host_wave64.py, test_gpu_parity.py::test_random_plans_on_the_gpu) measures kernels against the C restatement on plans
    drawn from the same parameter space as here; this pins the restatement to the REFERENCE over that space: whole streams through demodulate() and through the oracle,
    audio, raw-I/Q output, axcindicate and every statistic (CTCSS counters included) bit for bit (AIRBAND_FUZZ_SEEDS_ORACLE_PLANS=N for more seeds)."""
    device, carriers, wave_rate, fm_demod, n_batches = random_plan_case(pkg, seed)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = pkg.siggen.generate_u8(seed, 0, nbytes // 2, carriers)
    ref = _reference_run([device], [iq], n_batches, nfm=wave_rate == 16000, fm_demod=fm_demod)[0]
    orc = pyoracle.Oracle([device], wave_rate=wave_rate, fm_demod=fm_demod)
    try:
        got = orc.run_device(0, iq, n_batches)
        assert ref["n_batches"] == got["n_batches"] == n_batches
        assert np.array_equal(ref["axc"], got["axc"]), "seed %d: axcindicate" % seed
        for key in ("waveout", "iq_out"):
            same = (ref[key].view(np.uint32) == got[key].view(np.uint32)) | (np.isnan(ref[key]) & np.isnan(got[key]))
            assert same.all(), "seed %d: %s differs on channels %s" % (seed, key, sorted(set(np.nonzero(~same)[1].tolist())))
        for j in range(8):
            a, b = ref["stats"][j], orc.stats(0, j)
            for k in a:
                if k != "squelch_state":  # (an unstable lowpass -- bandwidth above WAVE_RATE, which the reference's parser accepts -- leaves NaN in agcavgfast / signal_level: NaN on both sides is agreement)
                    assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), (seed, j, k, a[k], b[k])
    finally:
        orc.close()


@need_ref
def test_tone_coefficients_all_standard_tones(built):
    ref = pyref.load_units(True)
    L = pyoracle.lib()
    tones = [67.0, 69.3, 71.9, 74.4, 77.0, 79.7, 82.5, 85.4, 88.5, 91.5, 94.8, 97.4, 100.0, 103.5, 107.2, 110.9, 114.8, 118.8, 123.0, 127.3, 131.8, 136.5, 141.3, 146.2,
             150.0, 151.4, 156.7, 159.8, 162.2, 165.5, 167.9, 171.3, 173.8, 177.3, 179.9, 183.5, 186.2, 189.9, 192.8, 196.6, 199.5, 203.5, 206.5, 210.7, 218.1, 225.7,
             229.1, 233.6, 241.8, 250.3, 254.1, 68.15, 123.45]
    for rate in (8000.0, 16000.0):
        for win in (int(rate * 0.05), int(rate * 0.4)):
            for t in tones:
                assert ref.refh_tone_coeff(t, rate, win) == L.orc_tone_coeff(t, rate, win), (t, rate, win)


@need_ref
def test_filters_bit_exact(built):
    ref = pyref.load_units(True)
    L = pyoracle.lib()
    rng = np.random.default_rng(7)
    x = rng.standard_normal(4000).astype(np.float32)
    for freq, rate, q in [(100.0, 16000.0, 10.0), (67.0, 8000.0, 10.0), (254.1, 16000.0, 3.0), (1000.0, 8000.0, 25.0)]:
        a, b = x.copy(), x.copy()
        ref.refh_notch_run(freq, rate, q, a.ctypes.data, len(a))
        L.orc_notch_run(freq, rate, q, b.ctypes.data, len(b))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (freq, rate, q)
        assert not np.array_equal(a, x)
    for freq, rate in [(6250.0, 16000.0), (4000.0, 16000.0), (2500.0, 8000.0), (500.0, 16000.0)]:
        re1, im1 = x.copy(), x[::-1].copy()
        re2, im2 = re1.copy(), im1.copy()
        ref.refh_lowpass_run(freq, rate, re1.ctypes.data, im1.ctypes.data, len(re1))
        L.orc_lowpass_run(freq, rate, re2.ctypes.data, im2.ctypes.data, len(re2))
        assert np.array_equal(re1.view(np.uint32), re2.view(np.uint32)) and np.array_equal(im1.view(np.uint32), im2.view(np.uint32)), (freq, rate)


@need_ref
def test_small_math_bit_exact(built):
    ref = pyref.load_units(True)
    L = pyoracle.lib()
    s1, c1, s2, c2 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    for phi in list(range(0, 1 << 24, 65536 + 4099)) + [0, 1, 0xFFFF, 0x10000, 0xFFFFFF]:
        ref.refh_sincos_lut(phi, C.byref(s1), C.byref(c1))
        L.orc_sincos_lut(phi, C.byref(s2), C.byref(c2))
        assert (s1.value, c1.value) == (s2.value, c2.value), phi
    for dbfs in (-1, -20, -40, -55, -90):
        assert ref.refh_dbfs_to_level(float(dbfs)) == L.orc_dbfs_to_level(float(dbfs), 512)
    rng = np.random.default_rng(3)
    v = rng.standard_normal((2000, 4)).astype(np.float32)
    v[0] = 0
    v[1] = [1, 0, -1, 0]
    for a, b, c, d in v:
        assert ref.refh_fast_atan2(a, b) == L.orc_fast_atan2(a, b)
        assert ref.refh_polar_disc_fast(a, b, c, d) == L.orc_polar_disc_fast(a, b, c, d)
        assert ref.refh_fm_quadri_demod(a, b, c, d) == L.orc_fm_quadri_demod(a, b, c, d)


@need_ref
@pytest.mark.parametrize("ctcss", [0.0, 100.0])
def test_squelch_object_bit_exact(built, ctcss):
    ref = pyref.load_units(False)  # AM build: WAVE_RATE 8000
    L = pyoracle.lib()
    rng = np.random.default_rng(11)
    n = 60000
    lvl = np.where((np.arange(n) // 7000) % 2 == 1, 0.75, 0.05).astype(np.float32)
    raw = (lvl * (1 + 0.2 * rng.standard_normal(n))).astype(np.float32)
    raw[20000:20050] = 0.01  # dead spot
    t = np.arange(1, n + 1)
    audio = (0.2 * np.sin(2 * np.pi * t * 100.0 / 8000) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    a = ref.refh_squelch_new(-1.0, 0, ctcss)
    b = L.orc_squelch_new(-1.0, 0, ctcss, 8000, 512)
    f1, f2 = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    ref.refh_squelch_raw_audio(a, raw.ctypes.data, audio.ctypes.data, n, f1.ctypes.data)
    L.orc_squelch_raw_audio(b, raw.ctypes.data, audio.ctypes.data, n, f2.ctypes.data)
    assert np.array_equal(f1, f2)
    c1, c2 = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
    ref.refh_squelch_counts(a, c1.ctypes.data)
    L.orc_squelch_counts(b, c2.ctypes.data)
    assert np.array_equal(c1, c2) and c1[0] > 0
    ref.refh_squelch_free(a)
    L.orc_squelch_free(b)


@need_ref
def test_afc_bit_exact(pkg, built):
    """AFC on: bins move at squelch-open edges ('<' / '>' in axcindicate) and return afterwards -- oracle == reference."""
    devices, carriers = helpers.afc_case(1)
    n_batches = 14
    iq = pkg.siggen.generate_u8(1, 0, helpers.stream_bytes(n_batches, 8000) // 2, carriers)
    ref = _reference_run(devices, [iq], n_batches, nfm=False)[0]
    orc = pyoracle.Oracle(devices, wave_rate=8000)
    got = orc.run_device(0, iq, n_batches)
    assert np.array_equal(ref["axc"], got["axc"])
    assert np.array_equal(ref["waveout"].view(np.uint32), got["waveout"].view(np.uint32))
    assert (ref["axc"] == ord(">")).any() and (ref["axc"] == ord("<")).any(), "test signal never triggered AFC"
    for j in range(8):
        assert ref["stats"][j]["bin"] == orc.stats(0, j)["bin"]
