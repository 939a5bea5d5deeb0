"""ctypes mirror of include/airband_hip.h (struct layouts and constants only; no logic)."""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 2
OK, ENODEV, EBADSIZE, ENOMEM, EINVAL, EAGAIN, ERUNTIME = 0, -1, -2, -3, -4, -5, -6
SFMT_U8, SFMT_S8, SFMT_S16, SFMT_F32 = 1, 2, 3, 4
MOD_AM, MOD_NFM = 0, 1
FM_FAST_ATAN2, FM_QUADRI_DEMOD = 0, 1
AGC_EXTRA = 100
FLAG_TRACE_SQUELCH, FLAG_RESERVED_2, FLAG_FORCE_FFT, FLAG_SERIAL_DEMOD, FLAG_PIPELINE, FLAG_REGROUP, FLAG_NO_REGROUP = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40

BYTES_PER_SAMPLE = {SFMT_U8: 1, SFMT_S8: 1, SFMT_S16: 2, SFMT_F32: 4}


class ChannelCfg(C.Structure):
    _fields_ = [("frequency", C.c_int32), ("modulation", C.c_int32), ("afc", C.c_int32), ("squelch_threshold_dbfs", C.c_int32),
                ("squelch_snr_threshold_db", C.c_float), ("notch_freq", C.c_float), ("notch_q", C.c_float), ("ctcss_freq", C.c_float),
                ("bandwidth_hz", C.c_int32), ("ampfactor", C.c_float), ("tau_us", C.c_int32), ("has_iq_outputs", C.c_int32)]


class DeviceCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("centerfreq", C.c_int32), ("sfmt", C.c_int32), ("fullscale", C.c_float), ("tau_us", C.c_int32),
                ("channel_count", C.c_int32), ("channels", C.POINTER(ChannelCfg))]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("flags", C.c_uint32), ("fft_size_log", C.c_int32), ("wave_rate", C.c_int32), ("fm_demod", C.c_int32),
                ("hip_device", C.c_int32), ("device_count", C.c_int32), ("devices", C.POINTER(DeviceCfg))]


class MixerInput(C.Structure):
    _fields_ = [("device", C.c_int32), ("channel", C.c_int32), ("mixer", C.c_int32), ("ampfactor", C.c_float), ("balance", C.c_float)]


class Geometry(C.Structure):
    _fields_ = [("fft_size", C.c_int32), ("wave_rate", C.c_int32), ("wave_batch", C.c_int32), ("device_count", C.c_int32), ("total_channels", C.c_int32),
                ("max_channels", C.c_int32), ("mixer_count", C.c_int32), ("wave_stride", C.c_int32), ("first_batch_bytes", C.c_int64),
                ("batch_bytes", C.c_int64), ("lookahead_bytes", C.c_int64)]


class ChannelStats(C.Structure):
    _fields_ = [("noise_level", C.c_float), ("signal_level", C.c_float), ("squelch_level", C.c_float), ("agcavgfast", C.c_float),
                ("open_count", C.c_uint64), ("flappy_count", C.c_uint64), ("ctcss_count", C.c_uint64), ("no_ctcss_count", C.c_uint64),
                ("active_counter", C.c_uint64), ("bin", C.c_int32), ("squelch_state", C.c_int32), ("signal_outside_filter", C.c_int32), ("reserved", C.c_int32)]


def channel_cfg(frequency, modulation=MOD_AM, afc=0, squelch_threshold_dbfs=0, squelch_snr_threshold_db=-1.0, notch_freq=0.0, notch_q=0.0, ctcss_freq=0.0,
                bandwidth_hz=0, ampfactor=1.0, tau_us=-1, has_iq_outputs=0) -> ChannelCfg:
    return ChannelCfg(int(frequency), int(modulation), int(afc), int(squelch_threshold_dbfs), float(squelch_snr_threshold_db), float(notch_freq),
                      float(notch_q), float(ctcss_freq), int(bandwidth_hz), float(ampfactor), int(tau_us), int(has_iq_outputs))


def device_cfg(channels, sample_rate=2_560_000, centerfreq=120_000_000, sfmt=SFMT_U8, fullscale=0.0, tau_us=-1):
    """Returns (DeviceCfg, keepalive array). ``channels`` = list of ChannelCfg or of kwargs dicts."""
    chs = [c if isinstance(c, ChannelCfg) else channel_cfg(**c) for c in channels]
    arr = (ChannelCfg * len(chs))(*chs)
    dev = DeviceCfg(int(sample_rate), int(centerfreq), int(sfmt), float(fullscale), int(tau_us), len(chs), C.cast(arr, C.POINTER(ChannelCfg)))
    return dev, arr
