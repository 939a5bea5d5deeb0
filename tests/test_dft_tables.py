"""Host-only check of the matrix-core channelizer's coefficient tables (no GPU): the int8 digit tables, offset corrections and scales
the kernel uses reproduce the defining sum X[bin] = sum_n lev[b_n] w[n] exp(-2 pi i bin n / N) (src/rtl_airband.cpp:316-351,402-489)
on pseudo-random raw windows -- for every fft size the path takes (one table per 512-sample window piece above 512), for dongles with
more than 8 channels (one table per group of 8), for CS16 with per-dongle full scale, and for sample rates whose hop is not a
multiple of 16 bytes.  The GPU parity tests check the kernel; this pins the arithmetic it is fed."""
import pytest

import helpers


@pytest.mark.parametrize("fft_log", [8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("mixed,wave_rate", [(False, 8000), (True, 16000)])
def test_tables_reproduce_the_windowed_dft(pkg, built, fft_log, mixed, wave_rate):
    devices, _ = helpers.plan_devices(3, mixed)
    devices[1]["sample_rate"] = devices[0].get("sample_rate", 2_560_000)
    err = pkg.dft_selftest(devices, wave_rate=wave_rate, fft_log=fft_log, windows=3)
    assert err < 2e-6, err  # 24-bit coefficients: ~3e-7 of the RMS value


def test_more_than_eight_channels_and_odd_rates(pkg, built):
    sg = pkg.siggen
    chans = [dict(frequency=sg.CENTERFREQ + (k - 24) * 40_000 + 5_000, modulation=k % 2) for k in range(48)]  # config/big_mixer.conf has 48 on one device
    err = pkg.dft_selftest([dict(channels=chans), dict(channels=chans[:5])], wave_rate=16000, windows=2)
    assert err < 2e-6, err
    devices, _ = helpers.plan_devices(2, True)
    for d in devices:
        d["sample_rate"] = 2_400_000  # hops of 300 bytes
    assert pkg.dft_selftest(devices, wave_rate=16000) < 2e-6


def test_cs16_with_per_dongle_full_scale(pkg, built):
    devices, _ = helpers.plan_devices(2, True)
    for i, d in enumerate(devices):
        d["sfmt"] = pkg.capi.SFMT_S16
        d["fullscale"] = 25500.0 if i == 0 else 2047.5  # a 16-bit and a 12-bit source
    assert pkg.dft_selftest(devices, wave_rate=16000, windows=3) < 2e-6
    assert pkg.dft_selftest(devices, wave_rate=16000, windows=2, fft_log=10) < 2e-6  # window pieces of 512 samples, same coefficient tables


def test_private_tables_of_afc_groups(pkg, built):
    """Groups with an AFC channel own their coefficient table (it is re-tuned at run time); dongles without one keep sharing."""
    devices, _ = helpers.afc_case(3)
    plain, _ = helpers.plan_devices(2, False)
    assert pkg.dft_selftest(devices + plain, wave_rate=8000, windows=2) < 2e-6


def test_s8_tables(pkg, built):
    """s8 (mirisdr, SoapySDR CS8): the byte is the int8 operand as it is, i / 128, nothing to restore -- at one and at several window pieces."""
    devices, _ = helpers.plan_devices(2, False)
    for d in devices:
        d["sfmt"] = pkg.capi.SFMT_S8
    assert pkg.dft_selftest(devices, wave_rate=8000, windows=3) < 2e-6
    assert pkg.dft_selftest(devices, wave_rate=8000, windows=2, fft_log=12) < 2e-6


def test_odd_hops_have_tables_too(pkg, built):
    """2.408 MS/s at WAVE_RATE 8000 is a hop of 301 samples = 602 bytes, 2-byte aligned: on the matrix-core path since round 4 (the tables do not depend on the hop)."""
    devices, _ = helpers.plan_devices(1, False)
    devices[0]["sample_rate"] = 2_408_000
    assert pkg.dft_selftest(devices, wave_rate=8000, windows=2) < 2e-6


@pytest.mark.parametrize("kw", [dict(), dict(fft_log=8), dict(sample_rate=2_400_000), dict(wave_rate=16000), dict(fft_log=10), dict(fft_log=11), dict(fft_log=11, wave_rate=16000),
                                dict(fft_log=12), dict(fft_log=13), dict(fft_log=12, wave_rate=16000), dict(sample_rate=2_008_000), dict(sample_rate=2_000_000, wave_rate=16000)])
def test_f32_tables(pkg, built, kw):
    """CF32 dongles (channelizer_f32.hip): the float tables, contracted on the host in the kernel's order (four pieces -- eight at fft 1024 / 2048 --, K = 4 per
    instruction, float accumulation; round 6: fft 4096 / 8192 as two / four window segments of 2 048 samples, one launch each, their sums added in float),
    against the double-precision sum."""
    devices, _ = helpers.plan_devices(2, False)
    for d in devices:
        d["sfmt"] = pkg.capi.SFMT_F32
        d["sample_rate"] = kw.get("sample_rate", 2_560_000)
    assert pkg.dft_selftest(devices, wave_rate=kw.get("wave_rate", 8000), windows=3, fft_log=kw.get("fft_log", 9)) < 2e-6


@pytest.mark.parametrize("kw", [dict(sfmt="SFMT_F32", sample_rate=20_000_000), dict(sample_rate=8_200_000)])
def test_configurations_of_the_fft_channelizer_are_refused(pkg, built, kw):
    """f32 at 20 MS/s (a 16-hop tile of 2 500-sample hops does not fit the staging registers); 8.2 MS/s u8 at WAVE_RATE 8000 is a hop of 2 050
    bytes, beyond the staging buffers."""
    devices, _ = helpers.plan_devices(1, False)
    if "sfmt" in kw:
        devices[0]["sfmt"] = getattr(pkg.capi, kw["sfmt"])
    if "sample_rate" in kw:
        devices[0]["sample_rate"] = kw["sample_rate"]
    with pytest.raises(pkg.AirbandError) as e:
        pkg.dft_selftest(devices, wave_rate=8000, fft_log=kw.get("fft_log", 9))
    assert e.value.code == pkg.capi.EBADSIZE
