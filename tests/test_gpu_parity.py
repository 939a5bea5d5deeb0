"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): squelch open/close decisions bit-exact, float audio within 1e-4 RMS; stage 2 alone
(same stage-1 input) bit-identical; stage-1 bins within 1e-5 relative RMS of the oracle's float64 FFT.
"""
import os

import numpy as np
import pytest

import helpers
import pyoracle

pytestmark = pytest.mark.gpu


def _tweak(d, ch):
    if d % 2 == 1:
        ch[3]["has_iq_outputs"] = 1
        ch[0]["bandwidth_hz"] = 8000
        ch[2]["squelch_threshold_dbfs"] = -40
        ch[4]["squelch_snr_threshold_db"] = 6.0
        ch[6]["ampfactor"] = 2.5


def _bursty(carriers):
    """Transmitters that key in short, weak, irregular bursts: the squelch flaps, aborts on low signal, re-opens inside the
    closing delay, fades out again and again -- the rare-event branches of the state machine and the AM fade-out path."""
    sg = helpers.sg
    out = []
    for k, c in enumerate(carriers):
        period, on = [(0.11, 0.045), (0.31, 0.02), (0.26, 0.19), (0.07, 0.05)][k % 4]
        amp = [0.08, 0.03, 0.05, 0.012][(k // 2) % 4]
        out.append(sg.make_carrier(sg.PLAN_OFFSETS_HZ[k], sg.SAMPLE_RATE, amplitude=amp, kind=c.kind, ctcss_hz=100.0 if c.step_ctcss else 0.0, key_slot=k,
                                   key_period_s=period, key_on_s=on, key_slot_s=0.013))
    return out


@pytest.mark.parametrize("style", ["keyed", "bursty"])
@pytest.mark.parametrize("mixed,wave_rate", [(False, 8000), (True, 16000)])
def test_stage2_bit_exact_on_oracle_bins(pkg, built, mixed, wave_rate, style):
    """Feed the ORACLE's stage-1 output into GPU stage 2: everything must be bit-identical."""
    devices, carriers = helpers.plan_devices(3, mixed, _tweak if mixed else None)
    if style == "bursty":
        carriers = _bursty(carriers)
    n_batches = 10
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    src = pyoracle.Oracle(devices, wave_rate=wave_rate)
    raw = [src.run_device(d, pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers), n_batches) for d in range(3)]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=pkg.capi.FLAG_TRACE_SQUELCH) as hip:
        for b in range(n_batches):
            wavein = np.concatenate([r["raw_wavein"][b] for r in raw])
            iqin = np.concatenate([r["raw_iq"][b] for r in raw])
            want = [orc.run_bins(d, raw[d]["raw_wavein"][b], raw[d]["raw_iq"][b]) for d in range(3)]
            hip.process_bins(wavein, iqin)
            out = hip.collect(iq=True, stats=True)
            tr = hip.read_trace()
            assert np.array_equal(tr, np.concatenate([w["trace"] for w in want])), "batch %d: squelch trace" % b
            assert np.array_equal(out["axc"], np.concatenate([w["axc"] for w in want])), "batch %d: axc" % b
            ww = np.concatenate([w["waveout"] for w in want])
            assert np.array_equal(out["waveout"].view(np.uint32), ww.view(np.uint32)), "batch %d: waveout max diff %g" % (b, np.abs(out["waveout"] - ww).max())
            wi = np.concatenate([w["iq_out"] for w in want])
            assert np.array_equal(out["iq_out"].view(np.uint32), wi.view(np.uint32)), "batch %d: iq_out" % b
        k = 0
        opens = flappy = 0
        for d in range(3):
            for j in range(8):
                o, g = orc.stats(d, j), out["stats"][k]
                for f in ("noise_level", "signal_level", "squelch_level", "agcavgfast", "open_count", "flappy_count", "ctcss_count", "no_ctcss_count",
                          "active_counter", "bin", "squelch_state", "signal_outside_filter"):
                    assert o[f] == g[f], (d, j, f, o[f], g[f])
                opens += int(g["open_count"])
                flappy += int(g["flappy_count"])
                k += 1
        assert opens > 0
        if style == "bursty":
            assert opens > 50 and flappy > 10, (opens, flappy)  # the signal did exercise re-opening and flap detection


def test_opening_timer_expires_on_the_first_sample_of_a_batch(pkg, built):
    """The GPU twin of tests/test_host_demod.py::test_opening_timer_expires_on_the_first_sample_of_a_batch_with_the_post_filter_in_use: 64 NFM + lowpass
    channels whose OPENING delay runs out on sample 0 of batch 2 with the post-filter gate deciding against the delay-line entry the kernel has to carry
    across the batch boundary (src/squelch.cpp:381-398,467-475)."""
    wave_rate, B, n_dev = 16000, 2000, 8
    devices, level = helpers.boundary_devices(n_dev)
    wave, iq = helpers.boundary_streams(lambda: pyoracle.Oracle(devices, wave_rate=wave_rate), n_dev, B, level)
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    outcomes = set()
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=pkg.capi.FLAG_TRACE_SQUELCH) as hip:
        for b in range(3):
            w, q = wave[:, b * B:(b + 1) * B], iq[:, 2 * b * B:2 * (b + 1) * B]
            want = [orc.run_bins(d, w[8 * d:8 * d + 8], q[8 * d:8 * d + 8]) for d in range(n_dev)]
            hip.process_bins(np.ascontiguousarray(w), np.ascontiguousarray(q))
            out = hip.collect()
            tr = hip.read_trace()
            want_t = np.concatenate([r["trace"] for r in want])
            if b == 2:
                outcomes = {int(t[1]) & 7 for t in want_t}
            assert np.array_equal(tr, want_t), "batch %d: squelch trace (channels %s)" % (b, np.nonzero((tr != want_t).any(axis=1))[0])
            assert np.array_equal(out["axc"], np.concatenate([r["axc"] for r in want]))
            ww = np.concatenate([r["waveout"] for r in want])
            assert np.array_equal(out["waveout"].view(np.uint32), ww.view(np.uint32))
    assert orc.stats(0, 0)["squelch_level"] == level and 4 in outcomes
    orc.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("AIRBAND_FUZZ_SEEDS_GPU", "8"))))
def test_random_plans_on_the_gpu(pkg, built, seed):
    """The GPU twin of tests/test_host_wave64.py::test_random_plans_with_wavefront_semantics (same seeds, same scenarios): random plans over every kind -- CTCSS
    on FM and AM channels, lowpass + CTCSS, raw-I/Q outputs, notch, manual squelch, both discriminators -- on made-up stage-1 output with awkward values (exact
    zeros, squares that underflow, large values), dense keying, squelch transitions aimed at the batch boundaries.  Stage 2 on the real wavefront (its own
    v_sqrt / v_rcp sequences, DPP, cooperative stores) against the oracle: squelch trace, axcindicate, audio, raw I/Q bit for bit; NaN where the oracle has NaN.
    AIRBAND_FUZZ_SEEDS_GPU=N runs N seeds (profiles/r04_experiments.md I: the campaign that was run)."""
    from test_host_wave64 import random_scenario
    devices, wave_rate, fm_demod, B, n_batches, streams = random_scenario(seed, max_dev=40 if seed % 5 == 4 else 9)  # every fifth seed fills whole 64-slot blocks
    n_dev = len(devices)
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fm_demod=fm_demod)
    try:
        with pkg.AirbandHip(devices, wave_rate=wave_rate, fm_demod=fm_demod, flags=pkg.capi.FLAG_TRACE_SQUELCH) as hip:
            for b in range(n_batches):
                w = np.concatenate([s[0][:, b * B:(b + 1) * B] for s in streams])
                q = np.concatenate([s[1][:, 2 * b * B:2 * (b + 1) * B] for s in streams])
                want = [orc.run_bins(d, streams[d][0][:, b * B:(b + 1) * B], streams[d][1][:, 2 * b * B:2 * (b + 1) * B]) for d in range(n_dev)]
                hip.process_bins(np.ascontiguousarray(w), np.ascontiguousarray(q))
                out = hip.collect(iq=True)
                tr = hip.read_trace()
                wt = np.concatenate([x["trace"] for x in want])
                assert np.array_equal(tr, wt), "seed %d batch %d: squelch trace (channels %s)" % (seed, b, np.nonzero((tr != wt).any(axis=1))[0])
                assert np.array_equal(out["axc"], np.concatenate([x["axc"] for x in want])), "seed %d batch %d: axc" % (seed, b)
                for key in ("waveout", "iq_out"):
                    ww = np.concatenate([x[key] for x in want])
                    same = (out[key].view(np.uint32) == ww.view(np.uint32)) | (np.isnan(out[key]) & np.isnan(ww))
                    assert same.all(), "seed %d batch %d: %s (channels %s)" % (seed, b, key, np.nonzero((~same).any(axis=1))[0])
    finally:
        orc.close()


@pytest.mark.parametrize("force_fft", [False, True], ids=["dft_mfma", "fft_wave64"])
@pytest.mark.parametrize("mixed,wave_rate", [(False, 8000), (True, 16000)])
def test_end_to_end_stream(pkg, built, mixed, wave_rate, force_fft):
    """Raw u8 I/Q through submit/process/collect vs the oracle: decisions exact, audio <= 1e-4 RMS."""
    n_dev, n_batches = 4, 14
    devices, carriers = helpers.plan_devices(n_dev, mixed, _tweak if mixed else None)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]
    assert all(r["n_batches"] == n_batches for r in ref)
    opened = 0
    flags = pkg.capi.FLAG_TRACE_SQUELCH | (pkg.capi.FLAG_FORCE_FFT if force_fft else 0)
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=flags) as hip:
        assert hip.channelizer_name() == ("fft_wave64" if force_fft else "dft_mfma_i8")
        # ragged submits: the library must cope with arbitrary chunking of the stream
        pos = [0] * n_dev
        b = 0
        chunk = 300_001
        while b < n_batches:
            for d in range(n_dev):
                if pos[d] < nbytes:
                    pos[d] += hip.submit(d, iq[d][pos[d]:pos[d] + chunk + 2 * d])
            while hip.process():
                out = hip.collect(iq=True)
                tr = hip.read_trace()
                w, q = hip.read_bins()
                want_t = np.concatenate([r["trace"][b] for r in ref])
                assert np.array_equal(out["axc"], np.concatenate([r["axc"][b] for r in ref])), "batch %d axc" % b
                assert np.array_equal(tr, want_t), "batch %d: %d squelch-state mismatches" % (b, int((tr != want_t).sum()))
                ww = np.concatenate([r["waveout"][b] for r in ref])
                assert helpers.rms(out["waveout"] - ww) <= 1e-4, "batch %d audio rms %g" % (b, helpers.rms(out["waveout"] - ww))
                wi = np.concatenate([r["iq_out"][b] for r in ref])
                assert helpers.rms(out["iq_out"] - wi) <= 1e-4 * max(1.0, helpers.rms(wi))
                # stage-1 bins: magnitudes of AM channels that need raw I/Q are rewritten in place by stage 2 (as in the reference,
                # src/rtl_airband.cpp:524), so compare |bin| on every other channel -- NFM ones included, which read_bins
                # recomputes from the raw bin I/Q -- and re/im on all of them
                plain = np.array([bool(c["modulation"]) or not (c["bandwidth_hz"] or c["has_iq_outputs"]) for dev in devices for c in dev["channels"]])
                assert plain.sum() > len(plain) // 2
                assert helpers.rel_rms(w[plain], np.concatenate([r["raw_wavein"][b] for r in ref])[plain]) <= 1e-5
                assert helpers.rel_rms(q, np.concatenate([r["raw_iq"][b] for r in ref])) <= 1e-5
                opened += int((out["axc"] == ord("*")).sum())
                b += 1
    assert opened > 0, "test signal never opened a squelch: not a meaningful parity run"


def test_synthetic_dongles_identical_on_device_and_host(pkg, built):
    torch = pytest.importorskip("torch")
    devices, carriers = helpers.plan_devices(3, True)
    n = 70_000
    with pkg.AirbandHip(devices, wave_rate=16000) as hip:
        hip.set_signal_plan(carriers)
        buf = torch.zeros((3, 2 * n + 64), dtype=torch.uint8, device="cuda")
        start = 1_234_568
        hip.generate_iq(buf.data_ptr(), buf.stride(0), start, 2 * n, seed=0x5EED, device_index_offset=5)
        hip.synchronize()
        got = buf.cpu().numpy()
    for d in range(3):
        want = pkg.siggen.generate_u8(5 + d, start // 2, n, carriers)
        assert np.array_equal(got[d, :2 * n], want), "dongle %d" % d
        assert not got[d, 2 * n:].any()


def test_zero_copy_device_path_matches_host_ring_path(pkg, built):
    """airband_hip_process_device on an HBM-resident, tail-replicated span == submit/process on the same bytes."""
    torch = pytest.importorskip("torch")
    n_dev, n_batches, wave_rate = 3, 4, 16000
    devices, carriers = helpers.plan_devices(n_dev, True)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = np.stack([pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)])
    with pkg.AirbandHip(devices, wave_rate=wave_rate) as a, pkg.AirbandHip(devices, wave_rate=wave_rate) as b:
        g = a.geometry
        dbuf = torch.from_numpy(iq).cuda()
        off = 0
        for k in range(n_batches):
            for d in range(n_dev):
                a.submit(d, iq[d, off:off + (g.first_batch_bytes if k == 0 else g.batch_bytes) + (g.lookahead_bytes if k == 0 else 0)] if k == 0 else
                         iq[d, off + g.lookahead_bytes:off + g.lookahead_bytes + g.batch_bytes])
            assert a.process()
            ra = a.collect()
            b.process_device(dbuf.data_ptr() + off, dbuf.stride(0))
            rb = b.collect()
            assert np.array_equal(ra["waveout"].view(np.uint32), rb["waveout"].view(np.uint32))
            assert np.array_equal(ra["axc"], rb["axc"])
            off += g.first_batch_bytes if k == 0 else g.batch_bytes


def test_zero_copy_spans_that_do_not_start_on_16_bytes(pkg, built):
    """Hops of 250 bytes (2.0 MS/s at WAVE_RATE 16000): from the second batch on a zero-copy span starts 8 bytes off a 16-byte boundary (2 100 hops = 525 000
    bytes) and the dongles' rows are 2 bytes apart in alignment -- the matrix-core channelizer stages aligned pieces from the byte in front of the span.
    process_device on such spans == submit / process on the same bytes (whose staging buffer is aligned), bit for bit."""
    torch = pytest.importorskip("torch")
    capi = pkg.capi
    n_dev, n_batches, wave_rate, sr = 3, 4, 16000, 2_000_000
    devices, iq = helpers.format_case(pkg, capi.SFMT_U8, 9, sr, wave_rate, n_dev, n_batches)
    n = min(len(x) for x in iq)
    stride = (n + 2 + 255) // 256 * 256 + 2   # rows that differ in alignment by 2 bytes
    host = np.zeros((n_dev, stride), np.uint8)
    for d in range(n_dev):
        host[d, :n] = iq[d][:n]
    with pkg.AirbandHip(devices, wave_rate=wave_rate) as a, pkg.AirbandHip(devices, wave_rate=wave_rate) as b:
        assert a.channelizer_name() == b.channelizer_name() == "dft_mfma_i8"
        g = a.geometry
        assert g.batch_bytes == 250 * 2000 and g.first_batch_bytes % 16 == 8
        dbuf = torch.from_numpy(host).cuda()
        off = 0
        opened = 0
        for k in range(n_batches):
            take = (g.first_batch_bytes + g.lookahead_bytes) if k == 0 else g.batch_bytes
            lo = off if k == 0 else off + g.lookahead_bytes
            for d in range(n_dev):
                assert a.submit(d, host[d, lo:lo + take]) == take
            assert a.process()
            ra = a.collect()
            b.process_device(dbuf.data_ptr() + off, stride)
            rb = b.collect()
            assert np.array_equal(ra["waveout"].view(np.uint32), rb["waveout"].view(np.uint32)), k
            assert np.array_equal(ra["axc"], rb["axc"])
            opened += int((rb["axc"] == ord("*")).sum())
            off += g.first_batch_bytes if k == 0 else g.batch_bytes
        assert opened > 0


def test_zero_copy_spans_of_300_byte_hops(pkg, built):
    """Hops of 300 bytes (2.4 MS/s at WAVE_RATE 16000, config/noaa.conf's rate): the kernel variant reads its fragments 4 bytes at a time, so spans may sit at any
    multiple of 4 bytes off a 16-byte boundary (rows 4 bytes apart in alignment here: 0, 4, 8 off) and == submit / process bit for bit; a span only 2 bytes
    off would put those reads on a 2-byte boundary: refused (round-4 ADVICE)."""
    torch = pytest.importorskip("torch")
    capi = pkg.capi
    n_dev, n_batches, wave_rate, sr = 3, 3, 16000, 2_400_000
    devices, iq = helpers.format_case(pkg, capi.SFMT_U8, 9, sr, wave_rate, n_dev, n_batches)
    n = min(len(x) for x in iq)
    stride = (n + 4 + 255) // 256 * 256 + 4
    host = np.zeros((n_dev, stride), np.uint8)
    for d in range(n_dev):
        host[d, :n] = iq[d][:n]
    with pkg.AirbandHip(devices, wave_rate=wave_rate) as a, pkg.AirbandHip(devices, wave_rate=wave_rate) as b:
        assert a.channelizer_name() == b.channelizer_name() == "dft_mfma_i8"
        g = a.geometry
        assert g.batch_bytes == 300 * 2000
        dbuf = torch.from_numpy(host).cuda()
        with pytest.raises(pkg.AirbandError) as e:
            b.process_device(dbuf.data_ptr() + 2, stride)
        assert e.value.code == capi.EINVAL
        with pytest.raises(pkg.AirbandError):
            b.process_device(dbuf.data_ptr(), stride + 2)
        off = 0
        opened = 0
        for k in range(n_batches):
            take = (g.first_batch_bytes + g.lookahead_bytes) if k == 0 else g.batch_bytes
            lo = off if k == 0 else off + g.lookahead_bytes
            for d in range(n_dev):
                assert a.submit(d, host[d, lo:lo + take]) == take
            assert a.process()
            ra = a.collect()
            b.process_device(dbuf.data_ptr() + off, stride)
            rb = b.collect()
            assert np.array_equal(ra["waveout"].view(np.uint32), rb["waveout"].view(np.uint32)), k
            assert np.array_equal(ra["axc"], rb["axc"])
            opened += int((rb["axc"] == ord("*")).sum())
            off += g.first_batch_bytes if k == 0 else g.batch_bytes
        assert opened > 0


def test_collect_waits_for_a_batch_enqueued_on_the_callers_stream(pkg, built):
    """process_device(..., stream=S) runs the whole batch on S; collect / collect_channels / read_trace / collect_mixers issue
    their copies on the handle's own stream and must order themselves behind S on the GPU.  S is kept busy with a long
    independent kernel queue first, so a missing dependency shows up as stale results."""
    torch = pytest.importorskip("torch")
    n_dev, n_batches, wave_rate = 24, 4, 16000
    devices, carriers = helpers.plan_devices(n_dev, True)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = np.stack([pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)])
    mix = [(d, c, (d + c) % 2, 1.0, 0.0) for d in range(n_dev) for c in range(8)]
    T = pkg.capi.FLAG_TRACE_SQUELCH
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=T) as a, pkg.AirbandHip(devices, wave_rate=wave_rate, flags=T) as b:
        a.set_mixers(2, mix)
        b.set_mixers(2, mix)
        g = a.geometry
        dbuf = torch.from_numpy(iq).cuda()
        side = torch.cuda.Stream()
        junk = torch.zeros((4096, 4096), device="cuda")
        off = 0
        for k in range(n_batches):
            a.process_device(dbuf.data_ptr() + off, dbuf.stride(0))
            ra = a.collect(stats=True)
            ta, ma = a.read_trace(), a.collect_mixers()
            with torch.cuda.stream(side):
                for _ in range(40):  # ~tens of ms of queued work in front of the batch
                    junk = junk @ junk * 1e-4
            b.process_device(dbuf.data_ptr() + off, dbuf.stride(0), side.cuda_stream)
            part = b.collect(first_channel=8, n_channels=16, stats=True)   # ranged, repeatable
            rb = b.collect(stats=True)
            tb, mb = b.read_trace(), b.collect_mixers()
            assert np.array_equal(ra["waveout"].view(np.uint32), rb["waveout"].view(np.uint32)), "batch %d" % k
            assert np.array_equal(ra["axc"], rb["axc"]) and np.array_equal(ta, tb)
            assert np.array_equal(part["waveout"].view(np.uint32), ra["waveout"][8:24].view(np.uint32)) and np.array_equal(part["axc"], ra["axc"][8:24])
            assert part["stats"] == ra["stats"][8:24] and ra["stats"] == rb["stats"]
            for x, y in zip(ma, mb):
                assert np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8))
            off += g.first_batch_bytes if k == 0 else g.batch_bytes
        b.synchronize()  # waits for the caller's stream too


@pytest.mark.parametrize("mixed,wave_rate", [(False, 8000), (True, 16000)])
def test_pipelined_mode_is_the_sequential_mode_one_batch_late(pkg, built, mixed, wave_rate):
    """AIRBAND_HIP_FLAG_PIPELINE (stage 1 of batch k beside stage 2 of batch k-1, two-batch-deep rings, two streams):
    every output is bit-identical to the sequential handle's, one process call later; flush() drains the last batch.
    Both entry points are covered: the zero-copy device path and the host-ring path."""
    torch = pytest.importorskip("torch")
    n_dev, n_batches, n_more = 5, 9, 3
    devices, carriers = helpers.plan_devices(n_dev, mixed, _tweak if mixed else None)
    nbytes = helpers.stream_bytes(n_batches + n_more, wave_rate)
    iq = np.stack([pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)])
    mix = [(d, c, (d * 8 + c) % 3, 1.0 + 0.1 * c, 0.0) for d in range(n_dev) for c in range(8)]
    P = pkg.capi.FLAG_PIPELINE | pkg.capi.FLAG_TRACE_SQUELCH
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=pkg.capi.FLAG_TRACE_SQUELCH) as seq, pkg.AirbandHip(devices, wave_rate=wave_rate, flags=P) as pip, \
            pkg.AirbandHip(devices, wave_rate=wave_rate, flags=P) as pip_host:
        for h in (seq, pip, pip_host):
            h.set_mixers(3, mix)
        g = seq.geometry
        assert pip.geometry.first_batch_bytes == g.first_batch_bytes and pip.geometry.lookahead_bytes == g.lookahead_bytes
        dbuf = torch.from_numpy(iq).cuda()
        want = []
        off = 0
        opened = 0
        for k in range(n_batches + n_more):
            seq.process_device(dbuf.data_ptr() + off, dbuf.stride(0))
            r = seq.collect(iq=True, stats=True)
            r["trace"] = seq.read_trace()
            r["mix"] = seq.collect_mixers()
            want.append(r)
            opened += int((r["axc"] == ord("*")).sum())
            off += g.first_batch_bytes if k == 0 else g.batch_bytes
        assert opened > 0

        def check(got, k, what):
            w = want[k]
            assert np.array_equal(got["axc"], w["axc"]), "%s batch %d axc" % (what, k)
            assert np.array_equal(got["waveout"].view(np.uint32), w["waveout"].view(np.uint32)), "%s batch %d waveout" % (what, k)
            assert np.array_equal(got["iq_out"].view(np.uint32), w["iq_out"].view(np.uint32)), "%s batch %d iq_out" % (what, k)
            assert np.array_equal(got["trace"], w["trace"]), "%s batch %d trace" % (what, k)
            for a, b in zip(got["mix"], w["mix"]):
                assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)), "%s batch %d mixers" % (what, k)
            for x, y in zip(got["stats"], w["stats"]):
                for f in ("noise_level", "signal_level", "squelch_level", "agcavgfast", "open_count", "flappy_count", "ctcss_count", "no_ctcss_count",
                          "active_counter", "bin", "squelch_state", "signal_outside_filter"):
                    assert x[f] == y[f], (what, k, f)

        def grab(h):
            r = h.collect(iq=True, stats=True)
            r["trace"] = h.read_trace()
            r["mix"] = h.collect_mixers()
            return r

        # zero-copy device path
        off = 0
        for k in range(n_batches):
            pip.process_device(dbuf.data_ptr() + off, dbuf.stride(0))
            if k == 0:
                with pytest.raises(pkg.AirbandError) as e:  # nothing to collect yet
                    pip.collect()
                assert e.value.code == pkg.capi.EAGAIN
            else:
                check(grab(pip), k - 1, "device path")
            off += g.first_batch_bytes if k == 0 else g.batch_bytes
        pip.flush()
        check(grab(pip), n_batches - 1, "device path (flush)")
        pip.flush()  # idempotent
        # a drained pipeline starts again: the call after flush() only runs stage 1 (no results, nothing raced), then business as usual
        for k in range(n_batches, n_batches + n_more):
            pip.process_device(dbuf.data_ptr() + off, dbuf.stride(0))
            if k == n_batches:
                with pytest.raises(pkg.AirbandError) as e:
                    pip.collect()
                assert e.value.code == pkg.capi.EAGAIN
            else:
                check(grab(pip), k - 1, "device path after flush")
            off += g.batch_bytes
        pip.flush()
        check(grab(pip), n_batches + n_more - 1, "device path (second flush)")
        # host-ring path, fed in ragged pieces
        pos = [0] * n_dev
        done = started = 0
        while done < n_batches - 1:
            for d in range(n_dev):
                if pos[d] < nbytes:
                    pos[d] += pip_host.submit(d, iq[d][pos[d]:pos[d] + 250_003 + 16 * d])
            while pip_host.process():
                started += 1
                if started >= 2:
                    check(grab(pip_host), done, "host path")
                    done += 1
        pip_host.flush()
        check(grab(pip_host), done, "host path (flush)")
        t = pip.timing_totals()
        assert t["batches"] == n_batches + n_more and t["channelizer_ms"] > 0 and t["demod_ms"] > 0


def test_mixers_match_reference_order_sum(pkg, built):
    """GPU-side mixer sums (src/mixer.cpp:133-140,201-214) vs the oracle's restatement: small mixers bit-exact (same
    summation order), a many-input mixer within float tolerance; stereo via balance."""
    import ctypes as C
    n_dev, n_batches, wave_rate = 20, 6, 8000
    devices, carriers = helpers.plan_devices(n_dev, False)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    n_mixers = 4
    inputs = []
    for d in range(n_dev):
        for c in range(8):
            if c < 2:
                inputs.append((d, c, 3, 0.5, 0.0))                        # mixer 3: 40 inputs (one sequential run), mono
            elif c == 2 and d < 3:
                inputs.append((d, c, 0, 1.5, -0.5 if d % 2 else 0.25))     # mixer 0: stereo
            elif c >= 3:
                inputs.append((d, c, 1, 1.0, 0.0))                        # mixer 1: 100 inputs -> two runs
    capi = pkg.capi
    arr = (capi.MixerInput * len(inputs))(*[capi.MixerInput(a, b, c, d, e) for a, b, c, d, e in inputs])
    base = (np.arange(n_dev, dtype=np.int32) * 8)
    L = pyoracle.lib()
    with pkg.AirbandHip(devices, wave_rate=wave_rate) as hip:
        hip.set_mixers(n_mixers, inputs)
        pos = [0] * n_dev
        seen = False
        for b in range(n_batches):
            for d in range(n_dev):
                pos[d] += hip.submit(d, iq[d][pos[d]:])
            assert hip.process()
            out = hip.collect()
            left, right, sig = hip.collect_mixers()
            B = hip.B
            wl, wr, ws = np.zeros((n_mixers, B), np.float32), np.zeros((n_mixers, B), np.float32), np.zeros(n_mixers, np.uint8)
            w = np.ascontiguousarray(out["waveout"])
            a = np.ascontiguousarray(out["axc"])
            L.orc_mix(arr, len(inputs), base.ctypes.data, w.ctypes.data, a.ctypes.data, B, n_mixers, wl.ctypes.data, wr.ctypes.data, ws.ctypes.data)
            assert np.array_equal(sig, ws)
            for m in (0, 3):  # single-run mixers: identical order, identical bits
                assert np.array_equal(left[m].view(np.uint32), wl[m].view(np.uint32)), m
                assert np.array_equal(right[m].view(np.uint32), wr[m].view(np.uint32)), m
            assert helpers.rms(left[1] - wl[1]) <= 1e-5 * max(1.0, helpers.rms(wl[1]))
            assert not left[2].any() and not sig[2]
            seen |= bool(ws.any())
        assert seen
        # mixer_disable_input (src/mixer.cpp:96-110): masked connections add nothing and raise no signal flag; order of the rest is kept
        masked = [i for i, t in enumerate(inputs) if t[2] == 3 and t[0] % 3 == 0] + [i for i, t in enumerate(inputs) if t[2] == 0]
        for i in masked:
            hip.mixer_enable_input(i, False)
        rest = [t for i, t in enumerate(inputs) if i not in set(masked)]
        arr2 = (capi.MixerInput * len(rest))(*[capi.MixerInput(a, b, c, d, e) for a, b, c, d, e in rest])
        for d in range(n_dev):
            extra = pkg.siggen.generate_u8(d, nbytes // 2, hip.geometry.batch_bytes // 2, carriers)
            assert hip.submit(d, extra) == extra.nbytes
        assert hip.process()
        out = hip.collect()
        left, right, sig = hip.collect_mixers()
        wl, wr, ws = np.zeros((n_mixers, B), np.float32), np.zeros((n_mixers, B), np.float32), np.zeros(n_mixers, np.uint8)
        w = np.ascontiguousarray(out["waveout"])
        a = np.ascontiguousarray(out["axc"])
        L.orc_mix(arr2, len(rest), base.ctypes.data, w.ctypes.data, a.ctypes.data, B, n_mixers, wl.ctypes.data, wr.ctypes.data, ws.ctypes.data)
        assert np.array_equal(sig, ws) and not sig[0] and not left[0].any() and not right[0].any()
        assert np.array_equal(left[3].view(np.uint32), wl[3].view(np.uint32))
        hip.mixer_enable_input(masked[0], True)  # and back in
        with pytest.raises(pkg.AirbandError):
            hip.mixer_enable_input(len(inputs), False)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AIRBAND_FUZZ_SEEDS_MIXERS", "6"))))
def test_random_mixer_wirings(pkg, built, seed):
    """Random mixer wiring -- 1 ... 6 mixers, every channel into 0 ... 2 of them in random connection order, random ampfactor and balance (mono mixers, hard left /
    right), some connections disabled (mixer_disable_input), mixers with no input at all -- against the reference's summation (src/mixer.cpp:133-140,189-214)
    restated by the oracle on the GPU's own channel audio: signal flags equal, mixers of up to 64 inputs (one sequential run) bit for bit, larger ones within 1e-5."""
    capi = pkg.capi
    rng = np.random.default_rng(31_000 + seed)
    n_dev, n_batches, wave_rate = int(rng.integers(1, 14)), 3, 8000
    devices, carriers = helpers.plan_devices(n_dev, False)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    n_mixers = int(rng.integers(1, 7))
    conns = []
    for d in range(n_dev):
        for c in range(8):
            for _ in range(int(rng.choice([0, 1, 1, 2]))):
                bal = float(rng.choice([0.0, 0.0, -1.0, 1.0, -0.5, 0.25, 0.8]))
                conns.append((d, c, int(rng.integers(0, n_mixers)), float(rng.choice([1.0, 0.5, 2.0, 0.0, 3.7])), bal))
    order = rng.permutation(len(conns))
    conns = [conns[i] for i in order]  # connection order = input index = summation order inside a mixer
    if not conns:
        conns = [(0, 0, 0, 1.0, 0.0)]
    off = set(int(i) for i in np.nonzero(rng.random(len(conns)) < 0.15)[0])
    rest = [t for i, t in enumerate(conns) if i not in off]
    arr = (capi.MixerInput * max(1, len(rest)))(*[capi.MixerInput(a, b, c, d, e) for a, b, c, d, e in rest])
    per_mixer = [sum(1 for t in conns if t[2] == m) for m in range(n_mixers)]  # the library's runs of 64 count the disabled connections too
    # a mixer is stereo once ANY input was connected with a balance (mixer_connect_input, src/mixer.cpp:84-91), disabled or not; the oracle's helper only sees the
    # enabled ones: where those are all centred, the right channel is the left one (ampl = ampr = 1)
    right_is_left = [any(t[4] != 0.0 for t in conns if t[2] == m) and not any(t[4] != 0.0 for t in rest if t[2] == m) for m in range(n_mixers)]
    base = np.arange(n_dev, dtype=np.int32) * 8
    L = pyoracle.lib()
    with pkg.AirbandHip(devices, wave_rate=wave_rate) as hip:
        hip.set_mixers(n_mixers, conns)
        for i in off:
            hip.mixer_enable_input(i, False)
        pos = [0] * n_dev
        for b in range(n_batches):
            for d in range(n_dev):
                pos[d] += hip.submit(d, iq[d][pos[d]:])
            assert hip.process()
            out = hip.collect()
            left, right, sig = hip.collect_mixers()
            B = hip.B
            wl, wr, ws = np.zeros((n_mixers, B), np.float32), np.zeros((n_mixers, B), np.float32), np.zeros(n_mixers, np.uint8)
            w, a = np.ascontiguousarray(out["waveout"]), np.ascontiguousarray(out["axc"])
            L.orc_mix(arr, len(rest), base.ctypes.data, w.ctypes.data, a.ctypes.data, B, n_mixers, wl.ctypes.data, wr.ctypes.data, ws.ctypes.data)
            assert np.array_equal(sig, ws), "seed %d batch %d: signal flags %s vs %s" % (seed, b, sig, ws)
            for m in range(n_mixers):
                if right_is_left[m]:
                    wr[m] = wl[m]
                if per_mixer[m] <= 64:
                    assert np.array_equal(left[m].view(np.uint32), wl[m].view(np.uint32)), "seed %d batch %d mixer %d (%d inputs): left" % (seed, b, m, per_mixer[m])
                    assert np.array_equal(right[m].view(np.uint32), wr[m].view(np.uint32)), "seed %d batch %d mixer %d (%d inputs): right" % (seed, b, m, per_mixer[m])
                else:
                    assert helpers.rms(left[m] - wl[m]) <= 1e-5 * max(1.0, helpers.rms(wl[m])) and helpers.rms(right[m] - wr[m]) <= 1e-5 * max(1.0, helpers.rms(wr[m]))


def test_mixer_exchange_between_handles(pkg, built):
    """The mixer exchange of include/airband_hip.h at the library boundary, on one GPU: the dongles of a 12-dongle fleet on two handles (5 + 7, the way the
    shim or `bench.py --gpus 2` shards them), every handle summing its own inputs of the three mixers -- one handle has NO input of mixer 2, and none with a
    balance for the stereo mixer 1 (airband_hip_mixer_set_stereo) -- then (a) airband_hip_add_mixers, the same-GPU transport, against the one-handle sums,
    (b) the RCCL entry points (librccl loaded on first use, communicator of one rank from a unique id, airband_hip_allreduce_mixers on the handle's stream
    and on a caller's): with one rank the all-reduce must leave the sums as they are."""
    import torch
    n_dev, n_batches, wave_rate, split = 12, 5, 8000, 5
    devices, carriers = helpers.plan_devices(n_dev, False)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    conns = [(d, 0, 0, 1.0, 0.0) for d in range(n_dev)] + [(8, 2, 1, 1.5, -0.5), (2, 3, 1, 1.0, 0.0), (10, 2, 1, 0.5, 0.25)] + [(1, 5, 2, 2.0, 0.0), (3, 6, 2, 1.0, 0.0)]
    parts = [(0, split), (split, n_dev)]
    with pkg.AirbandHip(devices, wave_rate=wave_rate) as whole, pkg.AirbandHip(devices[:split], wave_rate=wave_rate) as a, pkg.AirbandHip(devices[split:], wave_rate=wave_rate) as b:
        whole.set_mixers(3, conns)
        for h, (lo, hi) in zip((a, b), parts):
            h.set_mixers(3, [(d - lo, c, m, amp, bal) for (d, c, m, amp, bal) in conns if lo <= d < hi])
            h.mixer_set_stereo(1, True)   # part a holds only the mono input of the stereo mixer
        uid = pkg.AirbandHip.comm_unique_id()
        whole.comm_init_rank(uid, 1, 0)
        side = torch.cuda.Stream()
        pos = [0] * n_dev
        opened = 0
        for k in range(n_batches):
            for d in range(n_dev):
                h, lo = (a, 0) if d < split else (b, split)
                n = whole.submit(d, iq[d][pos[d]:])
                assert h.submit(d - lo, iq[d][pos[d]:pos[d] + n]) == n
                pos[d] += n
            assert a.batch_ready() and b.batch_ready()
            assert whole.process() and a.process() and b.process()
            want_l, want_r, want_s = whole.collect_mixers()
            whole.allreduce_mixers(side.cuda_stream if k % 2 else 0)   # one rank: identity, on either stream
            got = whole.collect_mixers()
            assert np.array_equal(got[0].view(np.uint32), want_l.view(np.uint32)) and np.array_equal(got[1].view(np.uint32), want_r.view(np.uint32)) and np.array_equal(got[2], want_s)
            a.add_mixers(b)
            l, r, s = a.collect_mixers()
            assert np.array_equal(s, want_s)
            assert np.array_equal(l[2].view(np.uint32), want_l[2].view(np.uint32))          # both inputs in part a: the same additions
            assert np.array_equal(l[1].view(np.uint32), want_l[1].view(np.uint32)) or helpers.rms(l[1] - want_l[1]) <= 1e-6   # connection order (8, 2, 10) vs part order (2 | 8, 10)
            assert helpers.rms(l[0] - want_l[0]) <= 1e-6 * max(1.0, helpers.rms(want_l[0])) and helpers.rms(r[1] - want_r[1]) <= 1e-6
            assert np.abs(want_r[1]).max() > 0 or not want_s[1]
            opened += int(want_s.sum())
        assert opened > 0
        assert not a.batch_ready()


@pytest.mark.parametrize("mixed,wave_rate,n_dev", [(True, 16000, 36), (False, 8000, 20)], ids=["nfm_build", "am_build"])
def test_full_slot_blocks_of_every_kind(pkg, built, mixed, wave_rate, n_dev):
    """Enough dongles that every demod kind owns whole 64-slot blocks: those take the cooperative store paths (eight lanes per
    channel write whole 128-byte lines of the audio rows and of the CTCSS hand-off rows), which small configurations never reach.
    Stage 2 on the oracle's stage-1 output: bit-identical, blocks with padding lanes included (36 mixed dongles -> 144 AM, 72 + 72 NFM
    channels; 20 AM dongles -> 160 channels with WAVE_BATCH = 1000, whose last output run is a short one)."""
    n_batches = 4
    devices, carriers = helpers.plan_devices(n_dev, mixed, _tweak if mixed else None)
    carriers = _bursty(carriers)  # plenty of opens, fades and re-opens inside four batches
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    src = pyoracle.Oracle(devices, wave_rate=wave_rate)
    raw = [src.run_device(d, pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers), n_batches) for d in range(n_dev)]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    opened = 0
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=pkg.capi.FLAG_TRACE_SQUELCH) as hip:
        for b in range(n_batches):
            wavein = np.concatenate([r["raw_wavein"][b] for r in raw])
            iqin = np.concatenate([r["raw_iq"][b] for r in raw])
            want = [orc.run_bins(d, raw[d]["raw_wavein"][b], raw[d]["raw_iq"][b]) for d in range(n_dev)]
            hip.process_bins(wavein, iqin)
            out = hip.collect(iq=True)
            tr = hip.read_trace()
            assert np.array_equal(tr, np.concatenate([w["trace"] for w in want])), "batch %d: squelch trace" % b
            assert np.array_equal(out["axc"], np.concatenate([w["axc"] for w in want])), "batch %d: axc" % b
            ww = np.concatenate([w["waveout"] for w in want])
            assert np.array_equal(out["waveout"].view(np.uint32), ww.view(np.uint32)), "batch %d: waveout max diff %g" % (b, np.abs(out["waveout"] - ww).max())
            wi = np.concatenate([w["iq_out"] for w in want])
            assert np.array_equal(out["iq_out"].view(np.uint32), wi.view(np.uint32)), "batch %d: iq_out" % b
            opened += int((out["axc"] == ord("*")).sum())
    assert opened > 50


@pytest.mark.parametrize("fmt", ["u8", "SFMT_F32"])
def test_ragged_shapes_and_empty_inputs(pkg, built, fmt):
    """Edge shapes: a dongle with ONE channel, one with the maximum of 64 (every feature combination, all five demod kinds,
    mostly padding-free blocks), one with 8; zero-length submits, process() before enough data, collect() before any batch.
    Once with u8 dongles (int8 matrix-core channelizer) and once with CF32 ones (float32 matrix-core channelizer: eight groups of 8, unused columns)."""
    sg = pkg.siggen
    wave_rate, n_batches = 16000, 6
    base = dict(modulation=0, afc=0, squelch_threshold_dbfs=0, squelch_snr_threshold_db=-1.0, notch_freq=0.0, notch_q=0.0, ctcss_freq=0.0, bandwidth_hz=0,
                ampfactor=1.0, tau_us=-1, has_iq_outputs=0)
    spacing = 35_000
    many, carriers = [], []
    for k in range(64):
        off = (k - 32) * spacing + 5_000
        c = dict(base, frequency=sg.CENTERFREQ + off)
        c["modulation"] = 1 if k % 3 else 0
        if k % 3 == 1:
            c["ctcss_freq"] = 100.0 if k % 2 else 0.0
            c["notch_freq"] = 100.0 if k % 4 == 1 else 0.0
        if k % 5 == 0:
            c["bandwidth_hz"] = 9000
        if k % 7 == 0:
            c["has_iq_outputs"] = 1
        if k % 11 == 0:
            c["squelch_threshold_dbfs"] = -42
        if k % 13 == 0:
            c["ampfactor"] = 3.0
        many.append(c)
        if k % 4 == 0:  # 16 transmitters; the other channels only ever see noise
            carriers.append(sg.make_carrier(off, sg.SAMPLE_RATE, kind=c["modulation"], ctcss_hz=c["ctcss_freq"], key_slot=k, key_period_s=0.4, key_on_s=0.25, key_slot_s=0.03))
    one = [dict(many[8])]
    eight = [dict(many[k]) for k in range(0, 64, 8)]
    devices = [dict(channels=one), dict(channels=many), dict(channels=eight)]
    n_dev = len(devices)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = [sg.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    if fmt != "u8":
        for dev in devices:
            dev["sfmt"] = pkg.capi.SFMT_F32
        iq = [helpers.convert_format(x, pkg.capi.SFMT_F32, pkg.capi).view(np.uint8) for x in iq]
        nbytes *= 4
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=pkg.capi.FLAG_TRACE_SQUELCH) as hip:
        assert hip.channelizer_name() == ("dft_mfma_i8" if fmt == "u8" else "dft_mfma_f32")  # the 64-channel dongle runs as eight groups of 8 on the matrix-core path
        assert hip.total_channels == 73
        with pytest.raises(pkg.AirbandError) as e:
            hip.collect()
        assert e.value.code == pkg.capi.EAGAIN
        assert hip.process() is False                       # nothing queued at all
        assert hip.submit(0, iq[0][:0]) == 0                # empty submit
        assert hip.submit(1, iq[1][:1000]) == 1000          # far less than a batch
        assert hip.process() is False
        pos = [0, 1000, 0]
        b = opened = 0
        while b < n_batches:
            for d in range(n_dev):
                if pos[d] < nbytes:
                    pos[d] += hip.submit(d, iq[d][pos[d]:pos[d] + 700_001])
            while hip.process():
                out = hip.collect(iq=True)
                tr = hip.read_trace()
                assert np.array_equal(out["axc"], np.concatenate([r["axc"][b] for r in ref])), "batch %d axc" % b
                assert np.array_equal(tr, np.concatenate([r["trace"][b] for r in ref])), "batch %d squelch trace" % b
                ww = np.concatenate([r["waveout"][b] for r in ref])
                assert helpers.rms(out["waveout"] - ww) <= 1e-4
                wi = np.concatenate([r["iq_out"][b] for r in ref])
                assert helpers.rms(out["iq_out"] - wi) <= 1e-4 * max(1.0, helpers.rms(wi))
                opened += int((out["axc"] == ord("*")).sum())
                b += 1
        assert opened > 0


@pytest.mark.parametrize("sfmt_name,fft_log,sample_rate,wave_rate", [
    ("SFMT_S8", 9, 2_560_000, 8000), ("SFMT_S16", 9, 2_560_000, 16000), ("SFMT_F32", 9, 2_560_000, 16000),
    ("SFMT_U8", 8, 2_560_000, 16000), ("SFMT_U8", 10, 2_560_000, 8000), ("SFMT_U8", 11, 2_560_000, 16000), ("SFMT_S16", 13, 2_560_000, 8000),
    ("SFMT_U8", 9, 2_400_000, 16000), ("SFMT_U8", 9, 2_400_000, 8000), ("SFMT_U8", 9, 1_024_000, 8000), ("SFMT_S16", 9, 2_560_000, 8000), ("SFMT_S16", 8, 2_048_000, 16000),
    ("SFMT_S16", 9, 2_400_000, 16000), ("SFMT_U8", 9, 3_200_000, 8000),
    ("SFMT_U8", 10, 2_400_000, 16000), ("SFMT_U8", 11, 2_560_000, 8000), ("SFMT_U8", 10, 1_024_000, 16000),
    ("SFMT_S16", 10, 2_560_000, 16000), ("SFMT_S16", 11, 2_400_000, 8000),
    ("SFMT_U8", 12, 2_560_000, 16000), ("SFMT_U8", 13, 2_560_000, 8000), ("SFMT_S8", 10, 2_400_000, 16000), ("SFMT_S8", 12, 2_560_000, 8000), ("SFMT_S16", 12, 2_400_000, 16000),
    # hops of an odd number of samples (250 / 250 / 150 bytes at 2-byte alignment): u8 and s8 on the matrix-core path since round 4
    ("SFMT_U8", 9, 2_000_000, 16000), ("SFMT_S8", 10, 2_000_000, 16000), ("SFMT_U8", 8, 1_200_000, 16000), ("SFMT_U8", 11, 2_000_000, 16000),
    # CF32 (SoapySDR) on the float32 matrix pipe since round 4: fft 512 / 256, 2.56 / 2.4 MS/s, both WAVE_RATEs (hops of 160 / 320 / 150 / 300 samples: padded and unpadded rows)
    ("SFMT_F32", 9, 2_560_000, 8000), ("SFMT_F32", 8, 2_560_000, 16000), ("SFMT_F32", 9, 2_400_000, 16000), ("SFMT_F32", 9, 2_400_000, 8000), ("SFMT_F32", 10, 2_560_000, 16000),
    # ... and at fft 1024 / 2048 since round 5 (workgroups of eight waves; 2048: 128 resident B registers per wave; WAVE_RATE 8000: the large-tile variants)
    ("SFMT_F32", 11, 2_560_000, 16000), ("SFMT_F32", 10, 2_400_000, 8000), ("SFMT_F32", 11, 2_560_000, 8000), ("SFMT_F32", 11, 2_400_000, 16000),
    # ... and at fft 4096 / 8192 since round 6 (two / four window segments of 2 048 samples, one launch of the fft 2048 kernel each, partial sums parked between them)
    ("SFMT_F32", 12, 2_560_000, 16000), ("SFMT_F32", 13, 2_560_000, 16000), ("SFMT_F32", 12, 2_400_000, 8000), ("SFMT_F32", 13, 2_560_000, 8000),
    # ... and with hops of an odd number of samples (125 / 251 / 75: rows of the staged image at 8-byte alignment, fragments from two 8-byte LDS reads)
    ("SFMT_F32", 9, 2_000_000, 16000), ("SFMT_F32", 10, 2_008_000, 8000), ("SFMT_F32", 8, 1_200_000, 16000), ("SFMT_F32", 12, 2_000_000, 16000)])
def test_other_formats_fft_sizes_and_rates(pkg, built, sfmt_name, fft_log, sample_rate, wave_rate):
    """Sample formats s8/s16/f32, fft sizes 256..8192, sample rates whose hop is not a multiple of 16 bytes: the matrix-core path
    takes u8, s8 and CS16 at every fft size and every hop (window pieces of 512 samples on cooperating waves from 1024 up, two passes at 8192; hops of an
    odd number of samples through 2-byte-aligned fragment reads); f32 runs on the wavefront-FFT channelizer -- same parity bars either way."""
    capi = pkg.capi
    sfmt = getattr(capi, sfmt_name)
    n_dev, n_batches = 2, 7
    devices, iq = helpers.format_case(pkg, sfmt, fft_log, sample_rate, wave_rate, n_dev, n_batches)
    hop = round(sample_rate / wave_rate)
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fft_log=fft_log)
    ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]
    assert all(r["n_batches"] == n_batches for r in ref)
    opened = 0
    with pkg.AirbandHip(devices, wave_rate=wave_rate, fft_log=fft_log, flags=capi.FLAG_TRACE_SQUELCH) as hip:
        hop_bytes = 2 * hop * capi.BYTES_PER_SAMPLE[sfmt]
        expect_dft = (sfmt in (capi.SFMT_U8, capi.SFMT_S8) and 64 <= hop_bytes <= 1024) or (sfmt == capi.SFMT_S16 and hop_bytes % 4 == 0 and 128 <= hop_bytes <= 1280)
        expect_f32 = sfmt == capi.SFMT_F32   # CF32 on the float32 matrix pipe (channelizer_f32.hip; fft 1024 / 2048 since round 5; 4096 / 8192 as window segments and hops of an odd number of samples since round 6)
        assert hip.channelizer_name() == ("dft_mfma_i8" if expect_dft else "dft_mfma_f32" if expect_f32 else "fft_wave64")
        pos = [0] * n_dev
        for b in range(n_batches):
            for d in range(n_dev):
                raw = iq[d].view(np.uint8)
                pos[d] += hip.submit(d, raw[pos[d]:])
            assert hip.process(), "batch %d: not enough input queued" % b
            out = hip.collect()
            tr = hip.read_trace()
            want_t = np.concatenate([r["trace"][b] for r in ref])
            assert np.array_equal(out["axc"], np.concatenate([r["axc"][b] for r in ref])), "batch %d axc" % b
            assert np.array_equal(tr, want_t), "batch %d: %d squelch-state mismatches" % (b, int((tr != want_t).sum()))
            ww = np.concatenate([r["waveout"][b] for r in ref])
            assert helpers.rms(out["waveout"] - ww) <= 1e-4
            opened += int((out["axc"] == ord("*")).sum())
    assert opened > 0


STAGE1_RATES = {8000: [960_000, 1_024_000, 1_200_000, 1_440_000, 1_800_000, 2_000_000, 2_048_000, 2_400_000, 2_560_000, 2_880_000, 3_200_000],
                16000: [960_000, 1_024_000, 1_200_000, 1_440_000, 1_920_000, 2_000_000, 2_048_000, 2_400_000, 2_560_000, 2_880_000, 3_200_000]}


def random_stage1_case(pkg, seed, n_batches=2):
    """(devices, iq, fft_log, wave_rate, n_batches): a random channelizer configuration -- sample format, fft size, sample rate (hops of 60 ... 400 samples, even and
    odd, 16-byte aligned or not), 1 ... 5 dongles with 1 ... 8 (now and then up to 24) channels each at random frequencies (negative offsets = bins in the upper half, channels sharing a bin)
    -- and I/Q for it: noise plus a tone on every channel's bin at a random level, driven into the rails now and then."""
    capi = pkg.capi
    rng = np.random.default_rng(9000 + seed)
    sfmt = [capi.SFMT_U8, capi.SFMT_U8, capi.SFMT_S8, capi.SFMT_S16, capi.SFMT_F32][int(rng.integers(0, 5))]
    fft_log = int(rng.choice([8, 9, 9, 9, 10, 11, 12, 13]))
    wave_rate = int(rng.choice([8000, 16000]))
    rates = [r for r in STAGE1_RATES[wave_rate] if r // wave_rate < (1 << fft_log)]
    sample_rate = int(rng.choice(rates))
    hop, n_fft = sample_rate // wave_rate, 1 << fft_log
    n_dev = int(rng.integers(1, 6))
    gains = [float(rng.choice([8.0, 50.0, 200.0])) for _ in range(n_dev)] if sfmt == capi.SFMT_S16 else [1.0] * n_dev
    devices = []
    for d in range(n_dev):
        chans = []
        for k in range(int(rng.integers(9, 25)) if rng.random() < 0.15 else int(rng.integers(1, 9))):  # more than eight: several column sets of the DFT tables
            off = int(rng.uniform(-0.42, 0.42) * sample_rate / 1000) * 1000
            chans.append(dict(frequency=120_000_000 + off, modulation=int(rng.integers(0, 2)) if wave_rate == 16000 else 0, afc=0, squelch_threshold_dbfs=0,
                              squelch_snr_threshold_db=-1.0, notch_freq=0.0, notch_q=0.0, ctcss_freq=0.0, bandwidth_hz=0, ampfactor=1.0, tau_us=-1, has_iq_outputs=0))
        devices.append(dict(channels=chans, sample_rate=sample_rate, sfmt=sfmt, fullscale=0.0 if sfmt != capi.SFMT_S16 else 127.5 * gains[d]))
    n = (n_batches * (wave_rate // 8) + 100) * hop + n_fft + 8
    t = np.arange(n)
    iq = []
    for d in range(n_dev):
        z = rng.normal(0.0, float(rng.uniform(1.0, 25.0)), (n, 2)) @ np.array([1.0, 1j])
        for k in range(len(devices[d]["channels"])):
            b = int(pkg.derive_constants([devices[d]], k, wave_rate=wave_rate, fft_log=fft_log)[0])
            f = (b if b < n_fft // 2 else b - n_fft) / n_fft + float(rng.uniform(-0.3, 0.3)) / n_fft
            z += float(10.0 ** rng.uniform(0.0, 1.9)) * np.exp(2j * np.pi * (f * t + rng.random()))
        u8 = np.empty(2 * n, np.uint8)
        u8[0::2] = np.clip(np.round(z.real + 127.5), 0, 255)
        u8[1::2] = np.clip(np.round(z.imag + 127.5), 0, 255)
        iq.append(helpers.convert_format(u8, sfmt, capi, gains[d]))
    return devices, iq, fft_log, wave_rate, n_batches, (capi.FLAG_FORCE_FFT if rng.random() < 0.15 else 0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AIRBAND_FUZZ_SEEDS_STAGE1", "10"))))
def test_random_channelizer_configurations(pkg, built, seed):
    """Whatever channelizer the library picks for a random configuration (int8 matrix cores at any alignment, CF32 on the float32 matrix pipe, the wavefront FFT),
    its bins are the oracle's within 1e-5 relative RMS, magnitudes and raw I/Q, dongle by dongle."""
    devices, iq, fft_log, wave_rate, n_batches, flags = random_stage1_case(pkg, seed)
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fft_log=fft_log)
    try:
        ref = [orc.run_device(d, iq[d], n_batches) for d in range(len(devices))]
        assert all(r["n_batches"] == n_batches for r in ref)
        with pkg.AirbandHip(devices, wave_rate=wave_rate, fft_log=fft_log, flags=flags) as hip:
            what = "seed %d: %s, sfmt %d, fft %d, %d S/s, WAVE_RATE %d, channels %s" % (seed, hip.channelizer_name(), devices[0]["sfmt"], 1 << fft_log, devices[0]["sample_rate"],
                                                                                      wave_rate, [len(d["channels"]) for d in devices])
            pos = [0] * len(devices)
            for b in range(n_batches):
                for d in range(len(devices)):
                    raw = iq[d].view(np.uint8)
                    pos[d] += hip.submit(d, raw[pos[d]:])
                assert hip.process(), what
                hip.collect()
                w, q = hip.read_bins()
                k = 0
                for d, r in enumerate(ref):
                    nc = len(devices[d]["channels"])
                    assert helpers.rel_rms(w[k:k + nc], r["raw_wavein"][b]) <= 1e-5, "%s; batch %d dongle %d: |bin| %g" % (what, b, d, helpers.rel_rms(w[k:k + nc], r["raw_wavein"][b]))
                    assert helpers.rel_rms(q[k:k + nc], r["raw_iq"][b]) <= 1e-5, "%s; batch %d dongle %d: bin I/Q %g" % (what, b, d, helpers.rel_rms(q[k:k + nc], r["raw_iq"][b]))
                    k += nc
    finally:
        orc.close()


_CHUNK_SEEDS = ([int(x) for x in os.environ["AIRBAND_FUZZ_CHUNKS_LIST"].split(",")] if os.environ.get("AIRBAND_FUZZ_CHUNKS_LIST")
                else list(range(int(os.environ.get("AIRBAND_FUZZ_SEEDS_CHUNKS", "6")))))


@pytest.mark.parametrize("seed", _CHUNK_SEEDS)
def test_results_do_not_depend_on_how_the_bytes_arrive(pkg, built, seed):
    """The reference's input drivers append whatever the hardware hands them -- any number of bytes, in the middle of an I/Q pair if need be -- to the device's
    circular buffer (src/input-common.cpp circbuffer_append; src/input-file.cpp:113-147) and demodulate() only sees whole batches.  So: one random configuration
    (test_random_channelizer_configurations' generator), the same streams submitted once in large pieces and once in random ones (1 byte ... 1.3 batches, odd
    lengths, dongles in random order, process() whenever it says yes): every batch bit-identical -- bins, squelch trace, audio."""
    devices, iq, fft_log, wave_rate, _, flags = random_stage1_case(pkg, seed + 40_000, n_batches=4)
    n_dev, n_batches = len(devices), 4
    raw = [x.view(np.uint8) for x in iq]
    rng = np.random.default_rng(77_000 + seed)

    reread = []
    pipelined = seed % 3 == 2 and os.environ.get("AIRBAND_FUZZ_CHUNKS_PIPE", "1") != "0"  # every third seed: the randomly fed handle also runs AIRBAND_HIP_FLAG_PIPELINE (results one process() late, flush() for the last)

    def run(chunked):
        got = []
        started = 0
        with pkg.AirbandHip(devices, wave_rate=wave_rate, fft_log=fft_log, flags=flags | pkg.capi.FLAG_TRACE_SQUELCH | (pkg.capi.FLAG_PIPELINE if chunked and pipelined else 0)) as hip:
            batch_bytes = int(hip.geometry.batch_bytes)
            pos = [0] * n_dev
            stalled = 0
            while started < n_batches and stalled < 100_000:
                order = rng.permutation(n_dev) if chunked else range(n_dev)
                moved = 0
                for d in order:
                    left = len(raw[d]) - pos[d]
                    if left <= 0:
                        continue
                    want = min(left, int(rng.integers(1, int(1.3 * batch_bytes))) if chunked else left)
                    if chunked and rng.random() < 0.2:
                        want = min(left, int(rng.integers(1, 64)))
                    n = hip.submit(int(d), raw[d][pos[d]:pos[d] + want])
                    pos[d] += n
                    moved += n
                while started < n_batches and hip.process():
                    started += 1
                    moved += 1
                    if chunked and pipelined:
                        if started == 1:
                            continue  # stage 1 of the first batch only
                        if started == n_batches:  # the last call: batch n - 2 is out, flush() brings batch n - 1
                            out = hip.collect()
                            got.append((out["waveout"].copy(), out["axc"].copy(), hip.read_trace()))
                            hip.flush()
                    out = hip.collect()
                    # AIRBAND_FUZZ_CHUNKS_REREAD=1: the same device rows read a second time -- a transfer, not a computation
                    again = hip.collect(first_channel=0, n_channels=hip.total_channels) if os.environ.get("AIRBAND_FUZZ_CHUNKS_REREAD") == "1" else out
                    if not np.array_equal(out["waveout"].view(np.uint32), again["waveout"].view(np.uint32)):
                        x, y = out["waveout"].view(np.uint32), again["waveout"].view(np.uint32)
                        ch = np.nonzero((x != y).any(axis=1))[0]
                        reread.append("%s pieces, batch %d: two reads of the same result rows differ on channels %s, first index %s, count %s" % (
                            "random" if chunked else "large", len(got), list(ch[:8]), [int(np.nonzero(x[c] != y[c])[0][0]) for c in ch[:8]], [int((x[c] != y[c]).sum()) for c in ch[:8]]))
                    got.append((out["waveout"].copy(), out["axc"].copy(), hip.read_trace()) + (() if chunked and pipelined else hip.read_bins()))
                stalled = 0 if moved else stalled + 1
        assert len(got) == n_batches, "only %d batches came out" % len(got)
        return got

    a, b = run(False), run(True)
    problems = []
    against_oracle = ""
    for k in range(n_batches):
        for name, x, y in zip(("waveout", "axc", "trace", "|bin|", "bin I/Q"), a[k], b[k]):  # (the rings of a pipelined handle already hold the next batch's bins)
            xv, yv = (x.view(np.uint32), y.view(np.uint32)) if x.dtype == np.float32 else (x, y)
            if not np.array_equal(xv, yv):
                ch = np.nonzero((xv != yv).reshape(xv.shape[0], -1).any(axis=1))[0]
                problems.append("batch %d %s: channels %s, first differing index %s, count %s, max |diff| %s" % (
                    k, name, list(ch[:8]), [int(np.nonzero((xv[c] != yv[c]).ravel())[0][0]) for c in ch[:8]], [int((xv[c] != yv[c]).sum()) for c in ch[:8]],
                    ["%.3g" % float(np.abs(x[c].astype(np.float64) - y[c]).max()) for c in ch[:8]]))
    if problems and not pipelined:  # which of the two is it?  Stage-1 bins of both against the oracle, channel by channel
        orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fft_log=fft_log)
        ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]
        orc.close()
        for k in range(n_batches):
            rw = np.concatenate([r["raw_wavein"][k] for r in ref])
            for who, g in (("large pieces", a[k]), ("random pieces", b[k])):
                worst = max(range(rw.shape[0]), key=lambda c: helpers.rel_rms(g[3][c], rw[c]))
                against_oracle += " | batch %d %s: |bin| vs oracle %.2e (worst channel %d: %.2e)" % (k, who, helpers.rel_rms(g[3], rw), worst, helpers.rel_rms(g[3][worst], rw[worst]))
    assert not problems and not reread, "seed %d (sfmt %d, fft %d, %d S/s, pipelined %s, flags %d, channels %s): the two ways of submitting differ: %s%s  REREAD: %s" % (
        seed, devices[0]["sfmt"], 1 << fft_log, devices[0]["sample_rate"], pipelined, flags, [len(d["channels"]) for d in devices], "; ".join(problems), against_oracle, "; ".join(reread))


def test_a_later_handle_with_larger_workgroups(pkg, built):
    """Two CF32 handles in one process on the same kernel variant, the second with longer hops: 2.56 MS/s then 2.88 MS/s at WAVE_RATE 8000 need 90 and 99 KiB of LDS
    per workgroup.  The opt-in beyond 64 KiB (hipFuncSetAttribute) is per kernel variant and was made with the FIRST handle's size -- the second handle's launch was
    refused (found reading the launch code in round 4; the shim's one-handle-per-device-class is such a process).  Both against the oracle."""
    capi = pkg.capi
    wave_rate, n_batches = 8000, 2
    for sample_rate in (2_560_000, 2_880_000):
        devices, iq = helpers.format_case(pkg, capi.SFMT_F32, 9, sample_rate, wave_rate, 1, n_batches)
        orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
        ref = orc.run_device(0, iq[0], n_batches)
        orc.close()
        with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=capi.FLAG_TRACE_SQUELCH) as hip:
            assert hip.channelizer_name() == "dft_mfma_f32"
            raw, pos = iq[0].view(np.uint8), 0
            for b in range(n_batches):
                pos += hip.submit(0, raw[pos:])
                assert hip.process(), sample_rate
                out = hip.collect()
                w, _ = hip.read_bins()
                assert helpers.rel_rms(w, ref["raw_wavein"][b]) <= 1e-5
                assert np.array_equal(out["axc"], ref["axc"][b]) and np.array_equal(hip.read_trace(), ref["trace"][b])


@pytest.mark.parametrize("force_fft", [False, True], ids=["dft_mfma", "fft_wave64"])
def test_s8_negative_rail(pkg, built, force_fft):
    """The byte -128 of an s8 source: the reference never initialises its table entry (src/rtl_airband.cpp:322-324); library and oracle
    continue the table's rule (-128 / 128 = -1.0) on both channelizers.  A clipping signal: bursts driven into both rails."""
    capi = pkg.capi
    wave_rate, n_batches = 8000, 3
    devices, iq = helpers.format_case(pkg, capi.SFMT_S8, 9, 2_560_000, wave_rate, 1, n_batches)
    x = iq[0].astype(np.int32) * 6          # overdrive ...
    clipped = np.clip(x, -128, 127).astype(np.int8)   # ... into an ADC that clips at its rails, -128 included
    assert (clipped == -128).sum() > 1000
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    ref = orc.run_device(0, clipped, n_batches)
    flags = capi.FLAG_TRACE_SQUELCH | (capi.FLAG_FORCE_FFT if force_fft else 0)
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=flags) as hip:
        assert hip.channelizer_name() == ("fft_wave64" if force_fft else "dft_mfma_i8")
        raw = clipped.view(np.uint8)
        pos = 0
        for b in range(n_batches):
            pos += hip.submit(0, raw[pos:])
            assert hip.process()
            out = hip.collect()
            w, q = hip.read_bins()
            assert helpers.rel_rms(w, ref["raw_wavein"][b]) <= 1e-5
            assert np.array_equal(out["axc"], ref["axc"][b]) and np.array_equal(hip.read_trace(), ref["trace"][b])
            assert helpers.rms(out["waveout"] - ref["waveout"][b]) <= 1e-4


def test_device_enable_takes_a_failed_dongle_out(pkg, built):
    """airband_hip_device_enable(h, d, 0) = what demodulate() does with a failed input (src/rtl_airband.cpp:383-391): the dongle is
    passed by -- not waited for, not demodulated, out of its mixers -- and the others go on exactly as before."""
    capi = pkg.capi
    n_dev, n_batches, wave_rate, off_at = 9, 7, 16000, 3   # 9 dongles: slot blocks with enabled and disabled lanes side by side
    devices, carriers = helpers.plan_devices(n_dev, True, _tweak)
    nbytes = helpers.stream_bytes(n_batches, wave_rate)
    iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
    ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]
    gone = [1, 8]
    inputs = [(d, c, 0, 1.0, 0.0) for d in range(n_dev) for c in (0, 2)]
    rest = [t for t in inputs if t[0] not in gone]
    arr = (capi.MixerInput * len(rest))(*[capi.MixerInput(*t) for t in rest])
    base = (np.arange(n_dev, dtype=np.int32) * 8)
    L = pyoracle.lib()
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=capi.FLAG_TRACE_SQUELCH) as hip:
        hip.set_mixers(1, inputs)
        pos = [0] * n_dev
        frozen = out = None
        for b in range(n_batches):
            if b == off_at:
                frozen = out["stats"]  # of the last batch the dongles took part in
                for d in gone:
                    hip.device_enable(d, False)
                hip.device_enable(gone[0], False)  # idempotent
            for d in range(n_dev):
                if b < off_at or d not in gone:  # a failed input delivers nothing any more
                    pos[d] += hip.submit(d, iq[d][pos[d]:])
            assert hip.process(), "batch %d: the handle waited for a dongle that is switched off" % b
            out = hip.collect(stats=True)
            tr = hip.read_trace()
            for d in range(n_dev):
                sl = slice(8 * d, 8 * d + 8)
                if b >= off_at and d in gone:
                    assert (out["axc"][sl] == ord(" ")).all()
                    for j in range(8):  # state frozen at the moment it was taken out
                        for k in ("open_count", "active_counter", "noise_level", "signal_level", "squelch_state", "ctcss_count"):
                            assert out["stats"][8 * d + j][k] == frozen[8 * d + j][k], (b, d, j, k)
                    continue
                assert np.array_equal(out["axc"][sl], ref[d]["axc"][b]), (b, d)
                assert np.array_equal(tr[sl], ref[d]["trace"][b]), (b, d)
                assert helpers.rms(out["waveout"][sl] - ref[d]["waveout"][b]) <= 1e-4
            if b >= off_at:  # the mixer no longer hears the dongles that are gone (disable_device_outputs -> mixer_disable_input)
                left, right, sig = hip.collect_mixers()
                B = hip.B
                wl, wr, ws = np.zeros((1, B), np.float32), np.zeros((1, B), np.float32), np.zeros(1, np.uint8)
                w, a = np.ascontiguousarray(out["waveout"]), np.ascontiguousarray(out["axc"])
                L.orc_mix(arr, len(rest), base.ctypes.data, w.ctypes.data, a.ctypes.data, B, 1, wl.ctypes.data, wr.ctypes.data, ws.ctypes.data)
                assert np.array_equal(sig, ws) and np.array_equal(left[0].view(np.uint32), wl[0].view(np.uint32))
        assert hip.submit(gone[0], iq[0][:1000]) == 1000  # dropped, not queued
        for d in range(n_dev):
            hip.device_enable(d, False)
        assert hip.process() is False  # nothing left to demodulate: the caller's cue to stop (src/rtl_airband.cpp:377-381)
        with pytest.raises(pkg.AirbandError):
            hip.device_enable(n_dev, False)


@pytest.mark.parametrize("force_fft,sfmt_name", [(False, "SFMT_U8"), (True, "SFMT_U8"), (False, "SFMT_F32")], ids=["dft_mfma", "fft_wave64", "dft_mfma_f32"])
def test_afc(pkg, built, force_fft, sfmt_name):
    """AFC-enabled channels (src/rtl_airband.cpp:180-251).  On the matrix-core channelizers (int8; CF32 on the float32 pipe since round 6) a group with an AFC
    channel owns its coefficient table, one wavefront FFT per dongle gives AFC the spectrum of the batch's last hop, and the re-tune kernel rewrites the moved
    channel's columns; the wavefront-FFT channelizer reads the bin from the channel state."""
    n_dev, n_batches = 3, 14
    devices, carriers = helpers.afc_case(n_dev)
    sfmt = getattr(pkg.capi, sfmt_name)
    nbytes = helpers.stream_bytes(n_batches, 8000)
    iq = [helpers.convert_format(pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers), sfmt, pkg.capi) for d in range(n_dev)]
    for d in devices:
        d["sfmt"] = sfmt
    orc = pyoracle.Oracle(devices, wave_rate=8000)
    ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]
    moved = 0
    flags = pkg.capi.FLAG_TRACE_SQUELCH | (pkg.capi.FLAG_FORCE_FFT if force_fft else 0)
    with pkg.AirbandHip(devices, wave_rate=8000, flags=flags) as hip:
        assert hip.channelizer_name() == ("fft_wave64" if force_fft else "dft_mfma_i8" if sfmt == pkg.capi.SFMT_U8 else "dft_mfma_f32")
        pos = [0] * n_dev
        for b in range(n_batches):
            for d in range(n_dev):
                pos[d] += hip.submit(d, iq[d].view(np.uint8)[pos[d]:])
            assert hip.process()
            out = hip.collect(stats=True)
            want_a = np.concatenate([r["axc"][b] for r in ref])
            assert np.array_equal(out["axc"], want_a), "batch %d: %s vs %s" % (b, bytes(out["axc"]), bytes(want_a))
            assert np.array_equal(hip.read_trace(), np.concatenate([r["trace"][b] for r in ref]))
            assert helpers.rms(out["waveout"] - np.concatenate([r["waveout"][b] for r in ref])) <= 1e-4
            moved += int(((out["axc"] == ord("<")) | (out["axc"] == ord(">"))).sum())
        k = 0
        for d in range(n_dev):
            for j in range(8):
                assert out["stats"][k]["bin"] == orc.stats(d, j)["bin"], (d, j)
                k += 1
    assert moved > 0


def test_unmoved_channels_do_not_see_a_neighbours_afc(pkg, built):
    """Matrix-core channelizer: a group goes onto its private coefficient table while one of its channels is away from its base bin.  The columns of
    the channels that have NOT moved are the home table's, byte for byte (retune_kernel copies them): their stage-1 bins are bit-identical to those of a
    fleet in which nobody has AFC."""
    n_batches = 10
    devices, carriers = helpers.afc_case(1)
    plain = [dict(channels=[dict(c, afc=0) for c in devices[0]["channels"]])]
    iq = pkg.siggen.generate_u8(0, 0, helpers.stream_bytes(n_batches, 8000) // 2, carriers)
    still = [j for j, c in enumerate(devices[0]["channels"]) if c["afc"] == 0]
    assert len(still) == 2
    moved = 0
    with pkg.AirbandHip(devices, wave_rate=8000) as a, pkg.AirbandHip(plain, wave_rate=8000) as b:
        assert a.channelizer_name() == b.channelizer_name() == "dft_mfma_i8"
        pa = pb = 0
        for k in range(n_batches):
            pa += a.submit(0, iq[pa:])
            pb += b.submit(0, iq[pb:])
            assert a.process() and b.process()
            out = a.collect()
            moved += int(((out["axc"] == ord("<")) | (out["axc"] == ord(">"))).sum())
            wa, _ = a.read_bins()
            wb, _ = b.read_bins()
            for j in still:
                assert np.array_equal(wa[j].view(np.uint32), wb[j].view(np.uint32)), (k, j)
    assert moved > 0


def test_fft_channelizer_lds_budget(pkg, built):
    """The wavefront-FFT channelizer stages 16 hops of raw samples in LDS: a wide format at a high sample rate needs more than the default
    64 KiB (CS16 at 10 MS/s: 77 KiB -- the kernel opts in to the CU's full 160 KiB and runs), and one that cannot fit at all is refused by
    prepare() with the 'unsupported size' code instead of failing every batch at launch."""
    capi = pkg.capi
    chans, _ = pkg.siggen.baseline_plan(mixed=False)
    sr = 10_000_000
    for c in chans:
        c["frequency"] = 120_000_000 + int((c["frequency"] - 120_000_000) * 3.0)
    dev = [dict(channels=[dict(c) for c in chans], sample_rate=sr, sfmt=capi.SFMT_S16, fullscale=25500.0)]
    hop, B, n_batches = sr // 8000, 1000, 2
    rng = np.random.RandomState(7)
    n = (n_batches * B + 100) * hop + 512
    t = np.arange(n)
    sig = 3000.0 * np.exp(2j * np.pi * (chans[1]["frequency"] - 120_000_000) / sr * t) + rng.normal(0, 300, n) + 1j * rng.normal(0, 300, n)
    iq = np.empty(2 * n, np.int16)
    iq[0::2] = np.round(sig.real)
    iq[1::2] = np.round(sig.imag)
    orc = pyoracle.Oracle(dev, wave_rate=8000)
    ref = orc.run_device(0, iq, n_batches)
    with pkg.AirbandHip(dev, wave_rate=8000, flags=capi.FLAG_TRACE_SQUELCH) as hip:
        assert hip.channelizer_name() == "fft_wave64"  # hops of 5 000 bytes: beyond the matrix-core path's staging
        raw = iq.view(np.uint8)
        pos = 0
        for b in range(n_batches):
            pos += hip.submit(0, raw[pos:])
            assert hip.process()
            out = hip.collect()
            assert np.array_equal(out["axc"], ref["axc"][b]) and np.array_equal(hip.read_trace(), ref["trace"][b])
            assert helpers.rms(out["waveout"] - ref["waveout"][b]) <= 1e-4
    too_wide = [dict(channels=[dict(c) for c in chans], sample_rate=20_000_000, sfmt=capi.SFMT_F32)]
    with pytest.raises(pkg.AirbandError) as e:
        pkg.AirbandHip(too_wide, wave_rate=8000)
    assert e.value.code == capi.EBADSIZE
