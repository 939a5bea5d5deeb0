"""Stage 2 with its wavefront semantics, on the CPU: csrc/demod.hip compiled for the host with tests/hostshim_wave64/ -- every work-item a fiber, the
cross-lane operations (ballots, v_readlane, barriers, the lockstep LDS exchanges) rendezvous points of a wavefront's fibers -- and driven by the
library's own launch_demod().  All kinds run: the CTCSS chain (front -> tone -> back kernels), the cooperative stores of full 64-channel blocks, real
64-bit lane masks with lanes in different squelch states.  Squelch trace (tone bit included), axcindicate, audio and statistics (CTCSS counters
included) must equal the oracle's bit for bit.  A cross-lane operation reached by only part of a wavefront would hang the emulation and is reported
as a deadlock -- the kernels' rule "lane masks are only assigned in wave-uniform control flow" is checked on the way.
Test infrastructure: it tests the logic of the code the GPU runs; the GPU parity tests test the kernels.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers
import pyoracle
from test_host_demod import CLANG, CSRC, HERE, REPO, HostDemod, _bursty, capi, pkg, sg


@pytest.fixture(scope="module")
def wave64(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ in this image")
    out = str(tmp_path_factory.mktemp("hostwave64") / "libhostwave64.so")
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-DAB_WAVE64_EMU",
           "-I" + os.path.join(HERE, "hostshim_wave64"), "-I" + os.path.join(REPO, "include")] + os.environ.get("AIRBAND_HOST_DEFINES", "").split() + [
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


@pytest.fixture(params=[0, 1, 2, 3], ids=["slot_order", "regrouped", "regrouped_line_groups", "permuted_line_groups"])
def regroup(request, monkeypatch):
    """Round 6: regrouped stage 2 (AIRBAND_HIP_FLAG_REGROUP on the GPU): workgroups of four wavefronts deal their 256 slots out among themselves by squelch state at
    every batch start and walk the batch in step -- which channels share a wavefront changes from batch to batch, the results must not.  2: the second form, the
    slots that share a ring line move together and the wavefronts do not wait for each other."""
    if request.param:
        monkeypatch.setenv("AB_HOST_REGROUP", str(request.param))
    else:
        monkeypatch.delenv("AB_HOST_REGROUP", raising=False)
    return request.param


def _tweak(d, ch):
    """Different dongles, different settings: the lanes of one wavefront then sit in different squelch states and take different per-lane paths."""
    if d % 2:
        ch[2]["squelch_snr_threshold_db"] = 6.0
        ch[3]["bandwidth_hz"] = 6250
    if d % 3 == 1:
        ch[0]["notch_freq"], ch[0]["notch_q"] = 1000.0, 5.0
        ch[5]["ctcss_freq"] = 88.5  # a CTCSS tone nobody sends: the gate stays shut
    if d % 4 == 2:
        ch[4]["squelch_threshold_dbfs"] = -38
        ch[7]["ctcss_freq"] = 100.0   # lowpass + CTCSS: the generic kind (pairs hand-off)
        ch[6]["bandwidth_hz"] = 8000  # an AM channel with a lowpass filter: raw I/Q on an AM lane (the generic kind again)
        ch[1]["has_iq_outputs"] = 1   # raw-I/Q output rows


@pytest.mark.parametrize("style,n_dev,mixed,wave_rate,n_batches", [("keyed", 3, True, 16000, 6), ("long", 3, True, 16000, 16), ("bursty", 3, True, 16000, 4), ("keyed", 8, False, 8000, 3),
                                                                    ("keyed", 32, True, 16000, 2)])
def test_all_kinds_with_wavefront_semantics(wave64, regroup, style, n_dev, mixed, wave_rate, n_batches):
    """3 dongles: partial blocks (lane-private stores).  8 AM dongles: one full 64-channel block (cooperative stores).  32 mixed dongles: full blocks of
    the AM, NFM + lowpass and NFM + CTCSS kinds -- cooperative stores in the fused kinds, the front's hand-off rows, the back kernel."""
    devices, carriers = helpers.plan_devices(n_dev, mixed, _tweak if mixed else None)
    if style == "bursty":
        carriers = _bursty(carriers) if not mixed else [sg.make_carrier(sg.PLAN_OFFSETS_HZ[k], sg.SAMPLE_RATE, amplitude=[0.08, 0.03, 0.05, 0.012][(k // 2) % 4], kind=c.kind,
                                                                        ctcss_hz=100.0 if c.step_ctcss else 0.0, key_slot=k,
                                                                        key_period_s=[0.11, 0.31, 0.26, 0.07][k % 4], key_on_s=[0.045, 0.02, 0.19, 0.05][k % 4], key_slot_s=0.013)
                                                        for k, c in enumerate(carriers)]
    if style == "long":  # keyed for 1.7 of 2 s: the slow CTCSS detector (0.4 s windows) completes several windows, the tone kernel's slow-only steady path decides
        carriers = [sg.make_carrier(sg.PLAN_OFFSETS_HZ[k], sg.SAMPLE_RATE, kind=c.kind, ctcss_hz=100.0 if c.step_ctcss else 0.0, key_slot=k, key_period_s=2.0, key_on_s=1.7)
                    for k, c in enumerate(carriers)]
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    src = pyoracle.Oracle(devices, wave_rate=wave_rate)
    raw = [src.run_device(d, sg.generate_u8(d, 0, nbytes // 2, carriers), n_batches) for d in range(n_dev)]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    hd = HostDemod(wave64, devices, wave_rate)
    try:
        tone_seen = 0
        for b in range(n_batches):
            wavein = np.concatenate([r["raw_wavein"][b] for r in raw])
            iqin = np.concatenate([r["raw_iq"][b] for r in raw])
            want = [orc.run_bins(d, raw[d]["raw_wavein"][b], raw[d]["raw_iq"][b]) for d in range(n_dev)]
            hd.process_bins(wavein, iqin)
            wave, axc, trace = hd.collect()
            wt = np.concatenate([w["trace"] for w in want])
            assert np.array_equal(trace, wt), "batch %d: squelch trace (channels %s)" % (b, np.nonzero((trace != wt).any(axis=1))[0])
            assert np.array_equal(axc, np.concatenate([w["axc"] for w in want])), "batch %d: axc" % b
            ww = np.concatenate([w["waveout"] for w in want])
            assert np.array_equal(wave.view(np.uint32), ww.view(np.uint32)), "batch %d: waveout (channels %s)" % (b, np.nonzero((wave.view(np.uint32) != ww.view(np.uint32)).any(axis=1))[0])
            tone_seen += int(((wt >> 5) & 1).sum())
        if mixed and style != "bursty":
            assert tone_seen > 0  # the CTCSS gate did open somewhere (the short bursts of the other style never fill a detector window)
        st = hd.stats()
        k = 0
        for d in range(n_dev):
            for j in range(len(devices[d]["channels"])):
                o = orc.stats(d, j)
                for f in ("noise_level", "signal_level", "squelch_level", "agcavgfast", "open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter",
                          "squelch_state"):
                    assert o[f] == st[k][f], (d, j, f, o[f], st[k][f])
                k += 1
        if style == "long":
            assert max(s["ctcss_count"] for s in st) >= 3
    finally:
        hd.close()
        src.close()
        orc.close()


def _fuzz_streams_ct(rng, chans, wave_rate, B, n_batches, dense=False):
    """test_host_demod._fuzz_streams with a CTCSS sub-tone on the FM channels that ask for one: the right tone, a neighbouring standard tone, or none."""
    from test_host_demod import _fuzz_streams
    nfm = [c["modulation"] == 1 for c in chans]
    wave, iq = _fuzz_streams(rng, len(chans), B, n_batches, nfm, dense)
    n = B * n_batches
    t = np.arange(n) / wave_rate
    for c, ch in enumerate(chans):
        if not (nfm[c] and ch["ctcss_freq"]):
            continue
        if rng.random() < 0.5:  # transmissions long enough for several windows of the fast detector and one or two of the slow one (0.4 s)
            floor = float(10.0 ** rng.uniform(-3.0, -1.5))
            env = np.full(n, floor) * (1.0 + 0.2 * rng.standard_normal(n))
            pos = int(rng.integers(0, 2000))
            while pos < n:
                on = int(rng.integers(2000, 12000))
                env[pos:pos + on] += floor * float(10.0 ** rng.uniform(0.6, 1.6))
                pos += on + int(rng.integers(300, 4000))
            ph = np.cumsum(rng.normal(0.0, 0.3, n)) + 2 * np.pi * rng.uniform(-0.1, 0.1) * np.arange(n)
            z = np.abs(env) * np.exp(1j * ph)
            iq[c, 0::2], iq[c, 1::2] = z.real.astype(np.float32), z.imag.astype(np.float32)
            wave[c] = np.sqrt(iq[c, 0::2] * iq[c, 0::2] + iq[c, 1::2] * iq[c, 1::2])
        kind = rng.random()
        f = ch["ctcss_freq"] if kind < 0.6 else (ch["ctcss_freq"] * 1.035 if kind < 0.8 else 0.0)
        if f:
            beta = float(rng.uniform(0.5, 4.0))
            rot = np.exp(1j * beta * np.sin(2 * np.pi * f * t))
            z = (iq[c, 0::2] + 1j * iq[c, 1::2]) * rot
            iq[c, 0::2], iq[c, 1::2] = z.real.astype(np.float32), z.imag.astype(np.float32)
            re, im = iq[c, 0::2], iq[c, 1::2]
            wave[c] = np.sqrt(re * re + im * im)
    return wave, iq


def random_scenario(seed, max_dev=9):
    """(devices, wave_rate, fm_demod, B, n_batches, streams): a random plan over EVERY kind and made-up stage-1 output for it -- shared with the GPU twin
    (tests/test_gpu_parity.py::test_random_plans_on_the_gpu), so a seed means the same scenario on the emulated and on the real wavefront."""
    rng = np.random.default_rng(5000 + seed)
    nfm_build = bool(seed % 4)
    wave_rate = 16000 if nfm_build else 8000
    fm_demod = int(rng.integers(0, 2)) if nfm_build else 0
    n_dev = int(rng.integers(1, max_dev + 1))
    devices = []
    for d in range(n_dev):
        chans = []
        for k, off in enumerate(sg.PLAN_OFFSETS_HZ):
            c = dict(frequency=sg.CENTERFREQ + off, modulation=0, afc=0, squelch_threshold_dbfs=0, squelch_snr_threshold_db=-1.0, notch_freq=0.0, notch_q=0.0, ctcss_freq=0.0,
                     bandwidth_hz=0, ampfactor=1.0, tau_us=-1, has_iq_outputs=0)
            if nfm_build and rng.random() < 0.6:
                c["modulation"] = 1
                c["tau_us"] = int(rng.choice([-1, 0, 50, 200, 750]))
            if rng.random() < 0.3:
                c["bandwidth_hz"] = int(rng.choice([5000, 6250, 12500, 25000]))  # on an AM channel too: raw I/Q on an AM lane (generic kind)
            if rng.random() < 0.4:
                c["ctcss_freq"] = float(rng.choice([67.0, 100.0, 123.0, 254.1]))
            mode = rng.random()
            if mode < 0.3:
                c["squelch_threshold_dbfs"] = int(rng.integers(-60, -10))
            elif mode < 0.6:
                c["squelch_snr_threshold_db"] = float(rng.choice([3.0, 6.0, 9.5, 14.0]))
            if rng.random() < 0.3:
                c["notch_freq"], c["notch_q"] = float(rng.choice([100.0, 150.0, 1000.0])), float(rng.choice([0.0, 4.0, 10.0]))
            if rng.random() < 0.3:
                c["ampfactor"] = float(rng.choice([0.25, 2.0, 8.0]))
            if rng.random() < 0.15:
                c["has_iq_outputs"] = 1
            chans.append(c)
        devices.append(dict(channels=chans))
    B, n_batches = wave_rate // 8, (6 if seed % 3 == 0 else 3)  # WAVE_BATCH (src/rtl_airband.h:90); six batches: the slow CTCSS detector (0.4 s) completes windows
    streams = [_fuzz_streams_ct(rng, devices[d]["channels"], wave_rate, B, n_batches, dense=seed % 4 >= 2) for d in range(n_dev)]
    if seed % 2:  # every other seed with its squelch transitions aimed at the batch boundaries (test_host_demod._aim_at_boundaries)
        from test_host_demod import _aim_at_boundaries
        streams = [_aim_at_boundaries(rng, lambda d=d: pyoracle.Oracle([devices[d]], wave_rate=wave_rate, fm_demod=fm_demod), streams[d][0], streams[d][1], B, n_batches)
                   for d in range(n_dev)]
    return devices, wave_rate, fm_demod, B, n_batches, streams


@pytest.mark.parametrize("seed", range(int(os.environ.get("AIRBAND_FUZZ_SEEDS_WAVE64", "4"))))
def test_random_plans_with_wavefront_semantics(wave64, regroup, seed):
    """Random plans over EVERY kind -- CTCSS on FM and AM channels, lowpass + CTCSS, raw-I/Q outputs, notch, manual squelch -- on made-up stage-1 output
    with awkward values, several dongles (so that a wavefront's lanes sit in different squelch states and blocks are partly filled): the kernels with
    their wavefront semantics against the oracle, bit for bit, squelch trace (tone bit included), axcindicate, audio."""
    devices, wave_rate, fm_demod, B, n_batches, streams = random_scenario(seed)
    n_dev = len(devices)
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fm_demod=fm_demod)
    hd = HostDemod(wave64, devices, wave_rate, fm_demod)
    try:
        assert hd.B == B
        for b in range(n_batches):
            w = np.concatenate([s[0][:, b * B:(b + 1) * B] for s in streams])
            q = np.concatenate([s[1][:, 2 * b * B:2 * (b + 1) * B] for s in streams])
            want = [orc.run_bins(d, streams[d][0][:, b * B:(b + 1) * B], streams[d][1][:, 2 * b * B:2 * (b + 1) * B]) for d in range(n_dev)]
            hd.process_bins(w, q)
            got_w, got_a, got_t = hd.collect()
            wt = np.concatenate([x["trace"] for x in want])
            assert np.array_equal(got_t, wt), "seed %d batch %d: squelch trace (channels %s)" % (seed, b, np.nonzero((got_t != wt).any(axis=1))[0])
            assert np.array_equal(got_a, np.concatenate([x["axc"] for x in want])), "seed %d batch %d: axc" % (seed, b)
            ww = np.concatenate([x["waveout"] for x in want])
            same = (got_w.view(np.uint32) == ww.view(np.uint32)) | (np.isnan(got_w) & np.isnan(ww))
            assert same.all(), "seed %d batch %d: waveout (channels %s)" % (seed, b, np.nonzero((~same).any(axis=1))[0])
    finally:
        hd.close()
        orc.close()
