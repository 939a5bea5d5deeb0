"""Parity at BASELINE scale: handles with 1 024 ... 65 536 dongles (BASELINE.json configs[1] and configs[2] at their stated
sizes) against the CPU oracle on a SAMPLE of dongles.

Every device_t of the reference is independent and any partition of them over demodulate() threads is correct
(src/rtl_airband.cpp:1052-1056,1070-1076), so a big handle is checked dongle by dongle on a sample: the indices around
the channelizer's 16 / 128-dongle XCD placement groups and the 64-slot demod blocks, first and last, plus pseudo-random
ones.  The big handles take branches small ones never reach: whole 128-dongle groups are XCD-permuted and the ragged
last group is not (200, 1 000 dongles), a dongle's tiles are split over several wavefronts or not (splits > 1 below
8 192 dongles, == 1 above), and at 65 536 dongles the stage-1 rings and the result rows are indexed beyond 2^31 elements.

Input is generated ON the GPU into an HBM-resident span per dongle and cycled through the way bench.py does it (first
batch with its AGC_EXTRA lead-in, then a ring of resident batches); the oracle twins get exactly those bytes, copied back.
"""
import numpy as np
import pytest

import helpers
import pyverify

pytestmark = pytest.mark.gpu

RING = 3
_IQ_POOL = {"buf": None}


def _resident_iq(torch, nbytes):
    """One device buffer for the resident I/Q of every case, grown when a bigger case comes along (the cases run smallest
    first).  Under pytest a tensor freed at the end of a case is not handed back to the driver before the next case starts
    (a stand-alone script with the same sequence gets it back at once), and two 160 GiB buffers do not fit one GPU."""
    buf = _IQ_POOL["buf"]
    biggest = 65536 * 2_624_512  # the 65 536-dongle AM case (hop 640 B): ask for that much as soon as a 65 536-dongle case shows up
    if nbytes > biggest // 2:
        nbytes = max(nbytes, biggest)
    if buf is None or buf.numel() < nbytes:
        _IQ_POOL["buf"] = None
        del buf
        torch.cuda.empty_cache()
        helpers.wait_for_gpu_memory(nbytes + (1 << 30), timeout_s=5.0)
        _IQ_POOL["buf"] = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    return _IQ_POOL["buf"]


def _tweak(d, ch):
    if d % 2 == 1:
        ch[0]["bandwidth_hz"] = 8000
        ch[2]["squelch_threshold_dbfs"] = -40
        ch[4]["squelch_snr_threshold_db"] = 6.0
        ch[6]["ampfactor"] = 2.5


CASES = [
    # n_dev, mixed, wave_rate, sampled dongles, pipelined, tweak, path ("" = u8 on the matrix-core channelizer, "force_fft" = u8 with
    # AIRBAND_HIP_FLAG_FORCE_FFT, "f32" = SoapySDR CF32 samples: both on the wavefront FFT, multi-workgroup placement and ragged last groups included)
    pytest.param(1024, False, 8000, 40, False, False, "", id="configs1_1024_am"),
    pytest.param(200, True, 16000, 24, False, True, "", id="200_mixed_partial_group"),
    pytest.param(1000, True, 16000, 32, False, False, "", id="1000_mixed_partial_group"),
    pytest.param(4096, True, 16000, 40, False, True, "", id="4096_mixed_splits2"),
    pytest.param(4096, True, 16000, 24, True, False, "", id="4096_mixed_pipelined"),
    pytest.param(65536, True, 16000, 48, False, False, "", id="configs2_65536_mixed"),
    pytest.param(65536, False, 8000, 32, False, False, "", id="65536_am"),
    pytest.param(4096, True, 16000, 32, False, True, "force_fft", id="4096_mixed_fft_wave64"),
    pytest.param(4099, True, 16000, 32, False, False, "f32", id="4099_mixed_SFMT_F32"),
    pytest.param(1030, True, 16000, 16, False, False, "f32_fft", id="1030_mixed_SFMT_F32_fft_wave64"),
    # round 6: a fleet in which no two dongles share a channel plan (every device_t derives its own bins, src/config.cpp:666-667) -- 5 000 coefficient tables, the ones
    # past the host builder's 4 096 built on the device -- stays on the matrix-core channelizer (it fell to the wavefront FFT past 4 096 plans)
    pytest.param(5000, True, 16000, 40, False, False, "plans", id="5000_mixed_5000_distinct_plans"),
    # round 6: stage 2 re-sorted at every batch boundary (AIRBAND_HIP_FLAG_REGROUP): same results, bit for bit
    pytest.param(4096, True, 16000, 40, False, True, "regroup", id="4096_mixed_regrouped"),
    pytest.param(1000, False, 8000, 24, False, False, "regroup", id="1000_am_regrouped_partial_block"),
    pytest.param(65536, True, 16000, 48, False, False, "regroup", id="configs2_65536_mixed_regrouped"),
    # ... and by the library's own choice where all of a handle's lane-per-channel wavefronts are resident at once (BASELINE configs[3]'s per-GPU shard)
    pytest.param(32768, True, 16000, 32, False, False, "auto_regroup", id="configs3_shard_32768_mixed_regrouped_by_residency"),
]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n_dev,mixed,wave_rate,k,pipelined,tweak,path", CASES)
def test_sampled_dongles_of_large_handles(pkg, built, n_dev, mixed, wave_rate, k, pipelined, tweak, path):
    torch = pytest.importorskip("torch")
    n_batches = 7
    chans, carriers = pkg.siggen.baseline_plan(mixed=mixed)

    n_plans = n_dev if path == "plans" else 1
    bin_hz = pkg.siggen.SAMPLE_RATE // 512

    def device(d):
        ch = [dict(c) for c in chans]
        if tweak:
            _tweak(d, ch)
        if n_plans > 1:  # channel c of dongle d sits ((d >> 2c) & 3) bins above the BASELINE plan's, and so does its carrier (set_signal_plan_shift below)
            for c, sh in zip(ch, pkg.siggen.plan_shift_bins(d, n_plans, len(ch))):
                c["frequency"] += sh * bin_hz
        return dict(channels=ch, sfmt=pkg.capi.SFMT_F32) if path.startswith("f32") else dict(channels=ch)

    devices = [device(d) for d in range(n_dev)]
    flags = pkg.capi.FLAG_TRACE_SQUELCH | (pkg.capi.FLAG_PIPELINE if pipelined else 0) | (pkg.capi.FLAG_FORCE_FFT if path in ("force_fft", "f32_fft") else 0)
    flags |= pkg.capi.FLAG_REGROUP if path == "regroup" else 0
    dongles = pyverify.sample_dongles(n_dev, k)
    hip = pkg.AirbandHip(devices, wave_rate=wave_rate, flags=flags)
    iq = spot = None
    try:
        assert hip.channelizer_name() == {"": "dft_mfma_i8", "force_fft": "fft_wave64", "f32": "dft_mfma_f32", "f32_fft": "fft_wave64", "plans": "dft_mfma_i8", "regroup": "dft_mfma_i8", "auto_regroup": "dft_mfma_i8"}[path]
        assert hip.stage2_regrouped() == (path in ("regroup", "auto_regroup")) or "AIRBAND_HIP_REGROUP" in __import__("os").environ
        g = hip.geometry
        lead = g.first_batch_bytes - g.batch_bytes
        span = lead + (RING + 1) * g.batch_bytes + g.lookahead_bytes
        stride = (span + 255) // 256 * 256
        iq = _resident_iq(torch, n_dev * stride)[:n_dev * stride].view(n_dev, stride)
        shift_step = pkg.siggen._turns(float(bin_hz), pkg.siggen.SAMPLE_RATE) if n_plans > 1 else 0
        if not path.startswith("f32"):
            hip.set_signal_plan(carriers)
            if n_plans > 1:
                hip.set_signal_plan_shift(n_plans, float(bin_hz), pkg.siggen.SAMPLE_RATE)
            hip.generate_iq(iq.data_ptr(), stride, 0, span, seed=0x5EED)
            hip.synchronize()
        else:  # the generator emits u8: a u8 handle generates slab by slab, (b - 127.5) / 127.5 makes CF32 of it (what bench.py --sample-format f32 does)
            slab = 512
            gen = pkg.AirbandHip([dict(channels=chans)] * slab, wave_rate=wave_rate)
            gen.set_signal_plan(carriers)
            tmp = torch.empty((slab, span // 4), dtype=torch.uint8, device="cuda")
            iqf = iq.view(torch.float32)
            for d0 in range(0, n_dev, slab):
                n = min(slab, n_dev - d0)
                gen.generate_iq(tmp.data_ptr(), span // 4, 0, span // 4, seed=0x5EED, device_index_offset=d0)
                gen.synchronize()
                iqf[d0:d0 + n, :span // 4] = (tmp[:n].to(torch.float32) - 127.5) / 127.5
            gen.close()
            del tmp
            torch.cuda.synchronize()
        host = {d: iq[d].cpu().numpy() for d in dongles}
        # the on-device generator is the host generator, also at the far end of the handle
        last = dongles[-1]
        assert last == n_dev - 1
        if not path.startswith("f32"):
            assert np.array_equal(host[last][:65536], pkg.siggen.generate_u8(last, 0, 32768, carriers, n_plans=n_plans, shift_step=shift_step))

        spot = pyverify.SpotCheck(device, dongles, wave_rate=wave_rate)

        def offset(i):
            return 0 if i == 0 else g.first_batch_bytes + ((i - 1) % RING) * g.batch_bytes

        opened = 0
        worst = 0.0
        for i in range(n_batches):
            hip.process_device(iq.data_ptr() + offset(i), stride)
            j = i - 1 if pipelined else i  # batch whose results the handle holds now
            if j < 0:
                continue
            spot.feed([host[d][offset(j):] for d in dongles])
            w = spot.compare(hip, trace=True, what="%d dongles" % n_dev)
            worst = max(worst, w["audio_rms"])
            opened += sum(int((r["axc"] == ord("*")).sum()) for r in spot.last)
        if pipelined:
            hip.flush()
            spot.feed([host[d][offset(n_batches - 1):] for d in dongles])
            spot.compare(hip, trace=True, what="%d dongles (flush)" % n_dev)
        assert opened > 0, "no sampled channel ever opened its squelch: not a meaningful parity run"
        # channels outside the sample: every one of them must at least have produced the right KIND of output
        # (finite audio, a legal axcindicate) -- catches a block of dongles that was never written at all
        for d in (n_dev // 3, (2 * n_dev) // 3):
            r = hip.collect(first_channel=8 * d, n_channels=8)
            assert np.isfinite(r["waveout"]).all() and set(np.unique(r["axc"])) <= {ord(" "), ord("*")}
    finally:
        if spot is not None:
            spot.close()
        hip.close()
        del iq
    print("%d dongles, %d sampled: worst audio RMS error %.3g" % (n_dev, len(dongles), worst))



def _tweak_all(d, ch):
    _tweak(1, ch)


REPLICA_CASES = [
    # n_dev, mixed, wave_rate, tweak (the same for every dongle), pipelined, path ("force_fft": the wavefront FFT)
    pytest.param(5000, True, 16000, True, False, "", id="5000_mixed_tweaked_partial_group"),
    pytest.param(4096, True, 16000, False, True, "", id="4096_mixed_pipelined"),
    pytest.param(65536, True, 16000, False, False, "", id="configs2_65536_mixed"),
    pytest.param(65536, False, 8000, False, False, "", id="65536_am"),
    # ~10^9 transforms through the wavefronts' LDS exchange in ONE process, every dongle bit-identical to dongle 0: the single-process twin of the rare event the
    # chunking fuzz saw with twelve processes on the GPU (profiles/r04_experiments.md I)
    pytest.param(65536, True, 16000, False, False, "force_fft", id="configs2_65536_mixed_fft_wave64"),
    # round 6: regrouped stage 2 -- whichever wavefront a channel lands in from batch to batch, its results are dongle 0's
    pytest.param(5000, True, 16000, True, False, "regroup", id="5000_mixed_tweaked_regrouped"),
]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n_dev,mixed,wave_rate,tweak,pipelined,path", REPLICA_CASES)
def test_every_dongle_of_a_replicated_handle(pkg, built, n_dev, mixed, wave_rate, tweak, pipelined, path):
    """The WHOLE handle, not a sample: every dongle gets dongle 0's channel plan and dongle 0's bytes, dongle 0 is checked against
    the oracle (trace, axcindicate, counters exact, audio <= 1e-4 RMS) and ALL other dongles' result rows, axcindicate, statistics
    and per-sample squelch traces must be bit-identical to dongle 0's (pyverify.replica_check)."""
    torch = pytest.importorskip("torch")
    n_batches = 7
    devices, carriers = helpers.plan_devices(1, mixed, _tweak_all if tweak else None)
    one = devices[0]
    flags = pkg.capi.FLAG_TRACE_SQUELCH | (pkg.capi.FLAG_PIPELINE if pipelined else 0) | (pkg.capi.FLAG_FORCE_FFT if path == "force_fft" else 0)
    flags |= pkg.capi.FLAG_REGROUP if path == "regroup" else 0
    hip = pkg.AirbandHip([one] * n_dev, wave_rate=wave_rate, flags=flags)
    iq = spot = None
    try:
        assert hip.channelizer_name() == ("fft_wave64" if path == "force_fft" else "dft_mfma_i8")
        g = hip.geometry
        lead = g.first_batch_bytes - g.batch_bytes
        span = lead + (RING + 1) * g.batch_bytes + g.lookahead_bytes
        stride = (span + 255) // 256 * 256
        iq = _resident_iq(torch, n_dev * stride)[:n_dev * stride].view(n_dev, stride)
        hip.set_signal_plan(carriers)
        hip.generate_iq(iq.data_ptr(), stride, 0, span, seed=0x5EED)
        hip.synchronize()
        host0 = iq[0].cpu().numpy()
        for d0 in range(1, n_dev, 4096):  # every dongle replays dongle 0's bytes
            iq[d0:d0 + 4096] = iq[0:1]
        torch.cuda.synchronize()
        assert torch.equal(iq[n_dev - 1], iq[0])
        spot = pyverify.SpotCheck([one], [0], wave_rate=wave_rate)

        def offset(i):
            return 0 if i == 0 else g.first_batch_bytes + ((i - 1) % RING) * g.batch_bytes

        opened = 0
        for i in range(n_batches):
            hip.process_device(iq.data_ptr() + offset(i), stride)
            j = i - 1 if pipelined else i
            if j < 0:
                continue
            spot.feed([host0[offset(j):]])
            spot.compare(hip, trace=True, what="replicated dongle 0 of %d" % n_dev)
            opened += int((spot.last[0]["axc"] == ord("*")).sum())
            if j in (0, 3, n_batches - 1):
                bad = pyverify.replica_check(hip, n_dev, 8)
                assert bad["waveout"] == bad["axc"] == bad["stats"] == bad["trace"] == 0, "batch %d: dongles that differ from dongle 0: %r" % (j, bad)
        assert opened > 0
    finally:
        if spot is not None:
            spot.close()
        hip.close()
        del iq
