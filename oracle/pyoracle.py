"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY: ctypes driver for oracle/liboracle_*.so (the C
restatement in airband_oracle.c).  Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""
from __future__ import annotations

import ctypes as C
import importlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
capi = importlib.import_module("rtlsdr-airband_amd.capi")
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle_am.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(capi.Config)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_wave_batch.argtypes = [C.c_void_p]
        L.orc_run_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 6
        L.orc_run_bins.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.orc_run_span.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.orc_channel_stats.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(capi.ChannelStats)]
        L.orc_channel_constants.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_window_coeff.restype = C.c_float
        L.orc_window_coeff.argtypes = [C.c_int, C.c_int]
        L.orc_sincos_lut.argtypes = [C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_dbfs_to_level.restype = C.c_float
        L.orc_dbfs_to_level.argtypes = [C.c_float, C.c_int]
        for f in ("orc_fast_atan2",):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_float, C.c_float]
        for f in ("orc_polar_disc_fast", "orc_fm_quadri_demod"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_float] * 4
        L.orc_tone_coeff.restype = C.c_float
        L.orc_tone_coeff.argtypes = [C.c_float, C.c_float, C.c_int]
        L.orc_notch_run.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.orc_lowpass_run.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_ctcss_run.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_squelch_new.restype = C.c_void_p
        L.orc_squelch_new.argtypes = [C.c_float, C.c_int, C.c_float, C.c_int, C.c_int]
        L.orc_squelch_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_squelch_raw_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_squelch_raw_audio.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_squelch_audio_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_squelch_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_squelch_free.argtypes = [C.c_void_p]
        L.orc_mix.argtypes = [C.POINTER(capi.MixerInput), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def build_config(devices, *, wave_rate: int, fft_log: int = 9, fm_demod: int = 0, flags: int = 0, hip_device: int = 0):
    """devices: list of dicts(channels=[kwargs...], sample_rate=..., centerfreq=..., sfmt=..., ...) -> (Config, keepalive)."""
    keep = []
    devs = []
    for dev in devices:
        dc, arr = capi.device_cfg(**dev)
        keep.append(arr)
        devs.append(dc)
    darr = (capi.DeviceCfg * len(devs))(*devs)
    keep.append(darr)
    cfg = capi.Config(capi.ABI_VERSION, flags, fft_log, wave_rate, fm_demod, hip_device, len(devs), C.cast(darr, C.POINTER(capi.DeviceCfg)))
    return cfg, keep


class Oracle:
    def __init__(self, devices, *, wave_rate: int, fft_log: int = 9, fm_demod: int = 0):
        self.L = lib()
        self.devices = devices
        cfg, self._keep = build_config(devices, wave_rate=wave_rate, fft_log=fft_log, fm_demod=fm_demod)
        self.h = self.L.orc_create(C.byref(cfg))
        if not self.h:
            raise ValueError("orc_create rejected the configuration")
        self.B = self.L.orc_wave_batch(self.h)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def run_device(self, d: int, iq: np.ndarray, max_batches: int):
        nch = len(self.devices[d]["channels"])
        B = self.B
        wave = np.zeros((max_batches, nch, B), np.float32)
        iqo = np.zeros((max_batches, nch, 2 * B), np.float32)
        axc = np.zeros((max_batches, nch), np.uint8)
        trace = np.zeros((max_batches, nch, B), np.uint8)
        rw = np.zeros((max_batches, nch, B), np.float32)
        ri = np.zeros((max_batches, nch, 2 * B), np.float32)
        iq = np.ascontiguousarray(iq)
        nb = self.L.orc_run_device(self.h, d, iq.ctypes.data, iq.nbytes, max_batches, wave.ctypes.data, iqo.ctypes.data, axc.ctypes.data, trace.ctypes.data,
                                   rw.ctypes.data, ri.ctypes.data)
        return dict(n_batches=nb, waveout=wave[:nb], iq_out=iqo[:nb], axc=axc[:nb], trace=trace[:nb], raw_wavein=rw[:nb], raw_iq=ri[:nb])

    def run_span(self, d: int, span: np.ndarray, *, trace: bool = True):
        """One batch from a span laid out like airband_hip_process_device's input (hop h at h * hop_bytes)."""
        nch = len(self.devices[d]["channels"])
        B = self.B
        wave = np.zeros((nch, B), np.float32)
        iqo = np.zeros((nch, 2 * B), np.float32)
        axc = np.zeros((nch,), np.uint8)
        tr = np.zeros((nch, B), np.uint8) if trace else None
        span = np.ascontiguousarray(span)
        self.L.orc_run_span(self.h, d, span.ctypes.data, wave.ctypes.data, iqo.ctypes.data, axc.ctypes.data, tr.ctypes.data if trace else None, None, None)
        return dict(waveout=wave, iq_out=iqo, axc=axc, trace=tr)

    def run_bins(self, d: int, wavein: np.ndarray, iq: np.ndarray):
        nch = len(self.devices[d]["channels"])
        B = self.B
        wave = np.zeros((nch, B), np.float32)
        iqo = np.zeros((nch, 2 * B), np.float32)
        axc = np.zeros((nch,), np.uint8)
        trace = np.zeros((nch, B), np.uint8)
        wavein = np.ascontiguousarray(wavein, np.float32)
        iq = np.ascontiguousarray(iq, np.float32)
        self.L.orc_run_bins(self.h, d, wavein.ctypes.data, iq.ctypes.data, wave.ctypes.data, iqo.ctypes.data, axc.ctypes.data, trace.ctypes.data)
        return dict(waveout=wave, iq_out=iqo, axc=axc, trace=trace)

    def stats(self, d: int, j: int) -> dict:
        st = capi.ChannelStats()
        self.L.orc_channel_stats(self.h, d, j, C.byref(st))
        return {f[0]: getattr(st, f[0]) for f in capi.ChannelStats._fields_}

    def constants(self, d: int, j: int):
        v = (C.c_double * 16)()
        self.L.orc_channel_constants(self.h, d, j, v)
        return list(v)
