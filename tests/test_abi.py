"""CPU-side checks of the drop-in boundary: libairband_hip.so loads, exports every symbol include/airband_hip.h
declares, derives the same per-channel constants as the oracle / the reference, and REFUSES to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
import pyoracle
import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg, built):
    header = open(os.path.join(ROOT, "include", "airband_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(airband_hip_[a-z_]+)\s*\(", header)))
    assert len(declared) >= 20
    L = pkg.load_library()
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(pkg.EXPORTS) == declared


def test_struct_layouts_match_header(pkg):
    capi = pkg.capi
    assert C.sizeof(capi.ChannelCfg) == 48
    assert C.sizeof(capi.DeviceCfg) == 32
    assert C.sizeof(capi.Config) == 40
    assert C.sizeof(capi.MixerInput) == 20
    assert C.sizeof(capi.Geometry) == 56
    assert C.sizeof(capi.ChannelStats) == 72  # ABI 2: + signal_outside_filter, reserved


def _tweak(d, ch):
    ch[3]["has_iq_outputs"] = 1
    ch[0]["bandwidth_hz"] = 8000
    ch[2]["squelch_threshold_dbfs"] = -40 - d
    ch[4]["squelch_snr_threshold_db"] = 6.0 + d
    ch[5]["ctcss_freq"] = 67.0 + 11.3 * d
    ch[5]["notch_freq"] = 67.0 + 11.3 * d
    ch[5]["notch_q"] = 2.0 + d
    ch[7]["tau_us"] = 50 * d
    ch[7]["bandwidth_hz"] = 5000 + 1000 * d
    ch[6]["frequency"] += 1234 * d + 1


@pytest.mark.parametrize("fft_log", [8, 9, 11])
def test_derived_constants_match_oracle(pkg, built, fft_log):
    devices, _ = helpers.plan_devices(6, True, _tweak)
    for d in devices:
        d["tau_us"] = 100
    orc = pyoracle.Oracle(devices, wave_rate=16000, fft_log=fft_log)
    k = 0
    for d in range(6):
        for j in range(8):
            mine = pkg.derive_constants(devices, k, wave_rate=16000, fft_log=fft_log)
            want = orc.constants(d, j)
            assert [np.float64(x) for x in mine] == [np.float64(x) for x in want], (d, j, mine, want)
            k += 1


@pytest.mark.skipif(not pyref.have_ref(True), reason="oracle/_ref not built")
def test_derived_bins_and_derotation_match_reference(pkg, built):
    """bins (src/config.cpp:666-667) and dm_dphi (:679-712) against the reference build itself, incl. a sample rate that is
    not a multiple of WAVE_RATE (config/noaa.conf style 2.4 MS/s) and off-grid frequencies."""
    chans = [dict(frequency=120_000_000 + off, modulation=1) for off in (-1_000_000, -333_333, -5001, 0, 4999, 12_500, 777_777, 1_199_000)]
    for sr in (2_560_000, 2_400_000, 1_024_000):
        devices = [dict(channels=chans, sample_rate=sr)]
        nbytes = 2 * (16000 // 8 + 200) * round(sr / 16000) + 4096
        ref = pyref.run_reference(devices, [np.full(nbytes, 128, np.uint8)], 0, nfm=True)[0]
        for j in range(len(chans)):
            mine = pkg.derive_constants(devices, j, wave_rate=16000)
            assert mine[0] == ref["consts"][j][0] and mine[1] == ref["consts"][j][1], (sr, j, mine[:2], ref["consts"][j][:2])
            assert np.float32(mine[2]) == np.float32(ref["consts"][j][2])


def test_bad_configurations_are_rejected(pkg, built):
    capi = pkg.capi
    devices, _ = helpers.plan_devices(1, True)
    with pytest.raises(pkg.AirbandError) as e:
        pkg.derive_constants(devices, 0, wave_rate=16000, fft_log=7)
    assert e.value.code == capi.EBADSIZE  # same meaning as gpu_fft_prepare() -2 (src/rtl_airband.cpp:302-305)
    with pytest.raises(pkg.AirbandError) as e:
        pkg.derive_constants(devices, 0, wave_rate=12000)
    assert e.value.code == capi.EBADSIZE
    with pytest.raises(pkg.AirbandError) as e:
        pkg.derive_constants(devices, 0, wave_rate=8000)  # NFM channels need the NFM build's WAVE_RATE
    assert e.value.code == capi.EINVAL
    bad = [dict(channels=[dict(devices[0]["channels"][0], ampfactor=-1.0)])]
    with pytest.raises(pkg.AirbandError):
        pkg.derive_constants(bad, 0, wave_rate=16000)
    with pytest.raises(pkg.AirbandError):
        pkg.derive_constants(devices, 99, wave_rate=16000)


def test_no_cpu_fallback(pkg, built):
    """Without a HIP device prepare() must fail with -1 (ENODEV) -- the product never computes on the CPU."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    devices, _ = helpers.plan_devices(1, False)
    with pytest.raises(pkg.AirbandError) as e:
        pkg.AirbandHip(devices, wave_rate=8000)
    assert e.value.code == pkg.capi.ENODEV


def test_null_and_misuse_do_not_crash(pkg, built):
    """Every entry point checks its arguments (C callers get error codes, not segfaults)."""
    L = pkg.load_library()
    capi = pkg.capi
    assert L.airband_hip_prepare(None, None) == capi.EINVAL
    h = C.c_void_p()
    assert L.airband_hip_prepare(None, C.byref(h)) == capi.EINVAL and not h.value
    assert b"NULL" in L.airband_hip_last_error(None)
    L.airband_hip_release(None)
    for fn in ("airband_hip_process", "airband_hip_synchronize"):
        assert getattr(L, fn)(None) == capi.EINVAL
    assert L.airband_hip_get_geometry(None, None) == capi.EINVAL
    assert L.airband_hip_submit(None, 0, None, 0) == capi.EINVAL
    assert L.airband_hip_collect(None, None, None, None, None) == capi.EINVAL
    assert L.airband_hip_process_device(None, None, 0, None) == capi.EINVAL
    assert L.airband_hip_channelizer_name(None) == b"fft_wave64"
    bad = capi.Config(capi.ABI_VERSION + 7, 0, 9, 8000, 0, 0, 0, None)
    assert L.airband_hip_prepare(C.byref(bad), C.byref(h)) == capi.EINVAL
