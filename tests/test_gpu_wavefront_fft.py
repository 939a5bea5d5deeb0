"""The wavefront-FFT channelizer (csrc/channelizer_fft.hip) on the GPU beyond the cases tests/test_gpu_parity.py holds: its decimated variants
(fft_size 1024 / 2048 / 4096: 2 / 4 / 8 transforms of 512 points per hop), f32 at a hop that is not a power of two, and u8 at 2.0 MS/s -- hops of 250 bytes
(forced: since round 4 the matrix-core path takes them).  Same bars as everywhere: squelch trace and axcindicate equal to the oracle's, audio within 1e-4 RMS.
(tests/test_host_fft.py runs the same kernel source on the CPU against a float64 FFT; this file is what a GPU says.)"""
import numpy as np
import pytest

import helpers
import pyoracle

CASES = [
    # sfmt, fft_log, sample_rate, wave_rate, force the FFT path
    ("SFMT_U8", 9, 2_000_000, 16000, True),   # (round 4: the matrix-core path takes 250-byte hops too; forced here to keep the FFT path's odd-hop case)
    ("SFMT_F32", 10, 2_400_000, 8000, True),    # (round 5: CF32 at fft 1024 / 2048 runs on the float32 matrix pipe; forced here to keep the FFT path's decimated f32 cases)
    ("SFMT_F32", 11, 2_560_000, 16000, True),
    ("SFMT_F32", 12, 2_560_000, 16000, True),   # (round 6: CF32 at fft 4096 / 8192 runs on the float32 matrix pipe as window segments; forced here to keep the FFT path's 8-transform f32 case)
    ("SFMT_U8", 12, 2_560_000, 8000, True),
]


def _case(pkg, sfmt_name, fft_log, sample_rate, wave_rate, n_dev=2, n_batches=5):
    sfmt = getattr(pkg.capi, sfmt_name)
    devices, iq = helpers.format_case(pkg, sfmt, fft_log, sample_rate, wave_rate, n_dev, n_batches)
    orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fft_log=fft_log)
    ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]
    orc.close()
    assert all(r["n_batches"] == n_batches for r in ref)
    return devices, iq, ref


@pytest.mark.parametrize("sfmt_name,fft_log,sample_rate,wave_rate,force", CASES)
def test_the_cases_open_a_squelch_on_the_oracle(pkg, built, sfmt_name, fft_log, sample_rate, wave_rate, force):
    """CPU half: the synthetic streams of the GPU cases below do open squelches (so the GPU comparison is not vacuous)."""
    _, _, ref = _case(pkg, sfmt_name, fft_log, sample_rate, wave_rate, n_dev=1, n_batches=5)
    assert sum(int((a == ord("*")).sum()) for a in ref[0]["axc"]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("sfmt_name,fft_log,sample_rate,wave_rate,force", CASES)
def test_wavefront_fft_variants(pkg, built, sfmt_name, fft_log, sample_rate, wave_rate, force):
    capi = pkg.capi
    n_dev, n_batches = 2, 5
    devices, iq, ref = _case(pkg, sfmt_name, fft_log, sample_rate, wave_rate, n_dev, n_batches)
    opened = 0
    with pkg.AirbandHip(devices, wave_rate=wave_rate, fft_log=fft_log, flags=capi.FLAG_TRACE_SQUELCH | (capi.FLAG_FORCE_FFT if force else 0)) as hip:
        assert hip.channelizer_name() == "fft_wave64"
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
