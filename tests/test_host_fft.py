"""The wavefront-FFT channelizer (csrc/channelizer_fft.hip: f32 samples, hops that are not a multiple of four bytes, AIRBAND_HIP_FLAG_FORCE_FFT, AFC's
last-hop spectrum) with its wavefront semantics on the CPU: the kernel source compiled for the host through tests/hostshim_wave64/ (lanes as fibers;
shuffles, barriers and a wavefront's LDS exchanges are rendezvous points), launched by the file's own launch_channelizer_fft(), compared with a float64
FFT of the same converted, windowed samples (reference: src/rtl_airband.cpp:402-489).  Test infrastructure: it checks the LOGIC of the code the GPU runs
(index maps, twiddles, exchanges, ring layout); the GPU parity tests check the kernels.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "rtlsdr-airband_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

pkg = importlib.import_module("rtlsdr-airband_amd")
capi = pkg.capi
sg = pkg.siggen


@pytest.fixture(scope="module")
def hostfft(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no host clang++ in this image")
    out = str(tmp_path_factory.mktemp("hostfft") / "libhostfft.so")
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-DAB_WAVE64_EMU", "-I" + os.path.join(HERE, "hostshim_wave64"),
           "-I" + os.path.join(REPO, "include"), "-o", out, os.path.join(HERE, "host_fft_harness.cpp"), os.path.join(CSRC, "params.cpp")]
    subprocess.run(cmd, check=True)
    lib = C.CDLL(out)
    vp = C.c_void_p
    lib.hostfft_run.argtypes = [C.POINTER(capi.Config), vp, C.c_long, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.hostfft_hop_samples.argtypes = [C.POINTER(capi.Config)]
    return lib


def _samples(rng, sfmt, n):
    """n complex samples in the format's dtype + their float64 values as the reference converts them (src/rtl_airband.cpp:316-324,402-455)."""
    if sfmt == capi.SFMT_U8:
        raw = rng.integers(0, 256, 2 * n, dtype=np.uint8)
        val = (raw.astype(np.float64) - 127.5) / 127.5
    elif sfmt == capi.SFMT_S8:
        raw = rng.integers(-127, 128, 2 * n).astype(np.int8)
        val = raw.astype(np.float64) / 128.0
    elif sfmt == capi.SFMT_S16:
        raw = rng.integers(-30000, 30001, 2 * n).astype(np.int16)
        val = raw.astype(np.float64) / 32768.0
    else:
        raw = (rng.standard_normal(2 * n) * 0.3).astype(np.float32)
        val = raw.astype(np.float64)
    return raw, val[0::2] + 1j * val[1::2]


CASES = [
    # sfmt, fft_log, sample_rate, wave_rate, n_dev, n_hops
    (capi.SFMT_U8, 9, 2_560_000, 16000, 3, 37),
    (capi.SFMT_U8, 9, 2_560_000, 8000, 2, 16),
    (capi.SFMT_F32, 9, 2_560_000, 16000, 2, 21),
    (capi.SFMT_S16, 9, 2_400_000, 16000, 2, 18),   # hop 150 samples
    (capi.SFMT_S8, 8, 2_560_000, 16000, 2, 19),
    (capi.SFMT_U8, 8, 1_024_000, 8000, 2, 17),
    (capi.SFMT_F32, 10, 2_560_000, 8000, 2, 17),
    (capi.SFMT_U8, 10, 2_048_000, 16000, 1, 33),
    (capi.SFMT_U8, 11, 2_560_000, 8000, 1, 6),
    (capi.SFMT_S16, 12, 2_560_000, 8000, 1, 5),
    (capi.SFMT_U8, 13, 2_560_000, 8000, 1, 4),
    (capi.SFMT_F32, 11, 2_560_000, 16000, 2, 19),
    (capi.SFMT_S8, 12, 2_400_000, 16000, 1, 5),
    (capi.SFMT_U8, 9, 20_000_000, 8000, 1, 17),    # hops of 5 000 bytes: 94 KiB of dynamic LDS (the launch opts in above 64 KiB)
    (capi.SFMT_F32, 9, 9_600_000, 8000, 1, 17),   # 150 KiB of raw samples per tile: no room for the exchange buffers, the shuffle kernel at fft 512
]


@pytest.mark.parametrize("sfmt,fft_log,sample_rate,wave_rate,n_dev,n_hops", CASES)
def test_fft_kernel_source_on_the_host(hostfft, sfmt, fft_log, sample_rate, wave_rate, n_dev, n_hops):
    rng = np.random.default_rng(1000 * fft_log + sfmt + n_hops)
    N = 1 << fft_log
    chans, _ = sg.baseline_plan(mixed=wave_rate == 16000)
    scale = sample_rate / 2_560_000
    for c in chans:
        c["frequency"] = 120_000_000 + int((c["frequency"] - 120_000_000) * scale * 0.8)
    if n_hops % 2 == 0:  # a dongle with 40 channels: lanes 8 .. 39 pick up bins too (the bins themselves are drawn below)
        chans = [dict(chans[i % 8], frequency=chans[i % 8]["frequency"] + 1000 * (i // 8)) for i in range(40)]
    fullscale = 32768.0 if sfmt == capi.SFMT_S16 else 0.0
    devices = [dict(channels=[dict(c) for c in chans], sample_rate=sample_rate, sfmt=sfmt, fullscale=fullscale) for _ in range(n_dev)]
    cfg, keep = pkg.make_config(devices, wave_rate=wave_rate, fft_log=fft_log)
    hop = hostfft.hostfft_hop_samples(C.byref(cfg))
    assert hop == round(sample_rate / wave_rate)
    n_samp = (n_hops - 1) * hop + N
    n_ch = len(chans)
    bins = rng.integers(0, N, n_dev * n_ch).astype(np.int32)
    bins[0], bins[1], bins[2] = 0, N - 1, N // 2
    raws, vals = zip(*[_samples(rng, sfmt, n_samp) for _ in range(n_dev)])
    stride = (raws[0].nbytes + 15) // 16 * 16 + 16
    buf = np.zeros(n_dev * stride + 64, np.uint8)
    base = (-buf.ctypes.data) % 16  # 16-byte aligned spans, like the library's device buffers
    for d in range(n_dev):
        buf[base + d * stride: base + d * stride + raws[d].nbytes] = raws[d].view(np.uint8)
    mag = np.zeros((n_dev * n_ch, n_hops), np.float32)
    iqb = np.zeros((n_dev * n_ch, n_hops, 2), np.float32)
    spec = np.zeros((n_dev, 2 * N), np.float32)
    win = np.zeros(N, np.float32)
    rc = hostfft.hostfft_run(C.byref(cfg), buf.ctypes.data + base, stride, n_hops, 0, bins.ctypes.data, mag.ctypes.data, iqb.ctypes.data, spec.ctypes.data, win.ctypes.data)
    assert rc == 0, rc
    nfm = [c["modulation"] == 1 for c in chans]
    worst = 0.0
    for d in range(n_dev):
        frames = np.stack([vals[d][t * hop: t * hop + N] * win.astype(np.float64) for t in range(n_hops)])
        F = np.fft.fft(frames, axis=1)
        ref_rms = np.sqrt(np.mean(np.abs(F) ** 2))
        for j in range(n_ch):
            want = F[:, bins[d * n_ch + j]]
            if nfm[j]:  # stage 1 leaves the raw bin, stage 2 takes its magnitude
                got = iqb[d * n_ch + j, :, 0] + 1j * iqb[d * n_ch + j, :, 1]
                err = np.sqrt(np.mean(np.abs(got - want) ** 2)) / ref_rms
            else:
                err = np.sqrt(np.mean((mag[d * n_ch + j] - np.abs(want)) ** 2)) / ref_rms
            worst = max(worst, err)
        got = spec[d, 0::2] + 1j * spec[d, 1::2]  # no AFC channel: the kernel still leaves the last hop's spectrum when asked to
        worst = max(worst, np.sqrt(np.mean(np.abs(got - F[-1]) ** 2)) / ref_rms)
    assert worst < 2e-6, worst


@pytest.mark.parametrize("sfmt,fft_log", [(capi.SFMT_U8, 9), (capi.SFMT_S16, 10), (capi.SFMT_U8, 8), (capi.SFMT_U8, 12)])
def test_last_hop_spectrum_launch(hostfft, sfmt, fft_log):
    """The one-hop, one-wavefront launch matrix-core handles with AFC channels use (airband_hip.cpp, launch_last_hop_spectrum)."""
    rng = np.random.default_rng(77 + fft_log)
    N = 1 << fft_log
    chans, _ = sg.baseline_plan(mixed=False)
    devices = [dict(channels=[dict(c) for c in chans], sfmt=sfmt, fullscale=32768.0 if sfmt == capi.SFMT_S16 else 0.0) for _ in range(3)]
    cfg, keep = pkg.make_config(devices, wave_rate=8000, fft_log=fft_log)
    raws, vals = zip(*[_samples(rng, sfmt, N) for _ in range(3)])
    stride = (raws[0].nbytes + 15) // 16 * 16
    buf = np.zeros(3 * stride + 64, np.uint8)
    base = (-buf.ctypes.data) % 16
    for d in range(3):
        buf[base + d * stride: base + d * stride + raws[d].nbytes] = raws[d].view(np.uint8)
    mag = np.zeros((24, 1), np.float32)
    iqb = np.zeros((24, 1, 2), np.float32)
    spec = np.zeros((3, 2 * N), np.float32)
    win = np.zeros(N, np.float32)
    assert hostfft.hostfft_run(C.byref(cfg), buf.ctypes.data + base, stride, 1, 1, None, mag.ctypes.data, iqb.ctypes.data, spec.ctypes.data, win.ctypes.data) == 0
    assert not mag.any() and not iqb.any()  # a spectrum-only launch leaves the rings alone
    for d in range(3):
        F = np.fft.fft(vals[d] * win.astype(np.float64))
        got = spec[d, 0::2] + 1j * spec[d, 1::2]
        assert np.sqrt(np.mean(np.abs(got - F) ** 2)) / np.sqrt(np.mean(np.abs(F) ** 2)) < 2e-6


@pytest.mark.parametrize("seed", range(int(os.environ.get("AIRBAND_FUZZ_SEEDS_FFT", "8"))))
def test_random_geometries(hostfft, seed):
    """Random sample format, fft size, hop (odd ones too), number of hops, dongles, channels per dongle (1 .. 64) and alignment of the dongles' spans
    (any whole sample: the staging loop fetches the 16-byte pieces that straddle a span's ends byte by byte)."""
    rng = np.random.default_rng(9000 + seed)
    sfmt = int(rng.choice([capi.SFMT_U8, capi.SFMT_S8, capi.SFMT_S16, capi.SFMT_F32]))
    fft_log = int(rng.choice([8, 9, 9, 9, 10, 11, 12, 13]))
    wave_rate = int(rng.choice([8000, 16000]))
    hop = int(rng.integers(40, 700))
    sample_rate = hop * wave_rate
    N = 1 << fft_log
    n_dev = int(rng.integers(1, 4))
    n_ch = int(rng.integers(1, 65))
    n_hops = int(rng.integers(1, 40 if fft_log <= 10 else 8))
    base_ch, _ = sg.baseline_plan(mixed=wave_rate == 16000)
    chans = [dict(base_ch[i % 8], frequency=120_000_000 + 1000 * i) for i in range(n_ch)]
    fullscale = 32768.0 if sfmt == capi.SFMT_S16 else 0.0
    devices = [dict(channels=[dict(c) for c in chans], sample_rate=sample_rate, sfmt=sfmt, fullscale=fullscale) for _ in range(n_dev)]
    cfg, keep = pkg.make_config(devices, wave_rate=wave_rate, fft_log=fft_log)
    assert hostfft.hostfft_hop_samples(C.byref(cfg)) == hop
    n_samp = (n_hops - 1) * hop + N
    bins = rng.integers(0, N, n_dev * n_ch).astype(np.int32)
    raws, vals = zip(*[_samples(rng, sfmt, n_samp) for _ in range(n_dev)])
    bpc2 = 2 * capi.BYTES_PER_SAMPLE[sfmt]
    stride = ((raws[0].nbytes + 15) // 16 * 16 + 16 + bpc2 * int(rng.integers(0, 8)))
    buf = np.full(n_dev * stride + 128, 0xEE, np.uint8)  # whatever lies outside a span must not matter
    base = (-buf.ctypes.data) % 16 + bpc2 * int(rng.integers(0, 16 // bpc2 + 1))
    for d in range(n_dev):
        buf[base + d * stride: base + d * stride + raws[d].nbytes] = raws[d].view(np.uint8)
    mag = np.zeros((n_dev * n_ch, n_hops), np.float32)
    iqb = np.zeros((n_dev * n_ch, n_hops, 2), np.float32)
    spec = np.zeros((n_dev, 2 * N), np.float32)
    win = np.zeros(N, np.float32)
    rc = hostfft.hostfft_run(C.byref(cfg), buf.ctypes.data + base, stride, n_hops, 0, bins.ctypes.data, mag.ctypes.data, iqb.ctypes.data, spec.ctypes.data, win.ctypes.data)
    if rc == -200:
        pytest.skip("a tile of this geometry does not fit a CU's LDS (the library refuses it too)")
    assert rc == 0, rc
    nfm = [c["modulation"] == 1 for c in chans]
    worst = 0.0
    for d in range(n_dev):
        frames = np.stack([vals[d][t * hop: t * hop + N] * win.astype(np.float64) for t in range(n_hops)])
        F = np.fft.fft(frames, axis=1)
        ref_rms = np.sqrt(np.mean(np.abs(F) ** 2))
        for j in range(n_ch):
            want = F[:, bins[d * n_ch + j]]
            if nfm[j]:
                got = iqb[d * n_ch + j, :, 0] + 1j * iqb[d * n_ch + j, :, 1]
                err = np.sqrt(np.mean(np.abs(got - want) ** 2)) / ref_rms
            else:
                err = np.sqrt(np.mean((mag[d * n_ch + j] - np.abs(want)) ** 2)) / ref_rms
            worst = max(worst, err)
        got = spec[d, 0::2] + 1j * spec[d, 1::2]
        worst = max(worst, np.sqrt(np.mean(np.abs(got - F[-1]) ** 2)) / ref_rms)
    assert worst < 2e-6, (seed, sfmt, fft_log, hop, n_hops, n_dev, n_ch, worst)
