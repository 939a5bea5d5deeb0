"""rtlsdr-airband_amd -- MI355X (gfx950) backend for RTLSDR-Airband's demodulate() hot path.

Thin Python host binding over the C ABI in include/airband_hip.h (libairband_hip.so, built in-tree by
``_build.build()``).  The product is the shared library; this module only marshals configuration and numpy /
torch buffers into it for tests, bench.py and multi-GPU launch scripts.  There is no Python or CPU compute
path here: without the HIP library and a GPU every data-path call raises.

Import with ``importlib.import_module("rtlsdr-airband_amd")`` (the directory name carries a hyphen).
"""
from __future__ import annotations

import ctypes as C
import importlib
import os
from typing import Optional, Sequence

import numpy as np

capi = importlib.import_module(__name__ + ".capi")
siggen = importlib.import_module(__name__ + ".siggen")
_build = importlib.import_module(__name__ + "._build")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AIRBAND_HIP_LIB") or os.path.join(HERE, "libairband_hip.so")  # env override: kernel experiments only

EXPORTS = [
    "airband_hip_prepare", "airband_hip_set_mixers", "airband_hip_release", "airband_hip_get_geometry", "airband_hip_last_error", "airband_hip_submit",
    "airband_hip_process", "airband_hip_process_device", "airband_hip_collect", "airband_hip_collect_mixers", "airband_hip_device_results",
    "airband_hip_synchronize", "airband_hip_process_bins", "airband_hip_read_bins", "airband_hip_read_trace", "airband_hip_channel_constants",
    "airband_hip_derive_constants", "airband_hip_last_timings", "airband_hip_channelizer_name", "airband_hip_set_signal_plan", "airband_hip_generate_iq",
    "airband_hip_flush", "airband_hip_timing_totals", "airband_hip_stream_wait_results", "airband_hip_mixer_enable_input",
    "airband_hip_device_enable", "airband_hip_gpu_count", "airband_hip_build_info", "airband_hip_dft_selftest", "airband_hip_collect_channels", "airband_hip_read_bins_channels", "airband_hip_read_trace_channels",
    "airband_hip_batch_ready", "airband_hip_mixer_set_stereo", "airband_hip_comm_unique_id", "airband_hip_comm_init_rank", "airband_hip_comm_init_all",
    "airband_hip_comm_group_begin", "airband_hip_comm_group_end", "airband_hip_allreduce_mixers", "airband_hip_add_mixers", "airband_hip_comm_destroy", "airband_hip_clear_mixers", "airband_hip_set_signal_plan_shift", "airband_hip_regrouped",
]

_lib = None


class DevicePtr:
    """A view of DEVICE memory the library owns (airband_hip_device_results) for consumers that stay on the GPU: `torch.as_tensor(DevicePtr(ptr, shape, "<f4"),
    device="cuda")` aliases the buffer through __cuda_array_interface__ -- no copy, no ownership.  bench.py reads the axcindicate row that way."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


class AirbandError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__("airband_hip error %d: %s" % (code, text))
        self.code = code


def load_library() -> C.CDLL:
    """dlopen libairband_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libairband_hip.so is missing: run __graft_entry__.build() / python rtlsdr-airband_amd/_build.py (no CPU fallback exists)")
    # PyTorch-ROCm wheels bundle their own HIP runtime (file name libamdhip64.so, SONAME libamdhip64.so.7).  Loading
    # it FIRST makes the dynamic loader resolve our DT_NEEDED libamdhip64.so.7 to that same copy; the other order
    # would put two HIP runtimes into one process (and the second one finds no GPU).  C/C++ hosts are unaffected.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
    L.airband_hip_prepare.argtypes = [C.POINTER(capi.Config), C.POINTER(vp)]
    L.airband_hip_set_mixers.argtypes = [vp, i32, C.POINTER(capi.MixerInput), i32]
    L.airband_hip_release.argtypes = [vp]
    L.airband_hip_release.restype = None
    L.airband_hip_get_geometry.argtypes = [vp, C.POINTER(capi.Geometry)]
    L.airband_hip_last_error.argtypes = [vp]
    L.airband_hip_last_error.restype = C.c_char_p
    L.airband_hip_submit.argtypes = [vp, i32, vp, sz]
    L.airband_hip_submit.restype = i64
    L.airband_hip_process.argtypes = [vp]
    L.airband_hip_batch_ready.argtypes = [vp]
    L.airband_hip_mixer_set_stereo.argtypes = [vp, i32, i32]
    L.airband_hip_comm_unique_id.argtypes = [vp]
    L.airband_hip_comm_init_rank.argtypes = [vp, vp, i32, i32]
    L.airband_hip_comm_init_all.argtypes = [C.POINTER(vp), i32]
    L.airband_hip_allreduce_mixers.argtypes = [vp, vp]
    L.airband_hip_add_mixers.argtypes = [vp, vp]
    L.airband_hip_comm_destroy.argtypes = [vp]
    L.airband_hip_clear_mixers.argtypes = [vp]
    L.airband_hip_comm_group_begin.argtypes = []
    L.airband_hip_comm_group_end.argtypes = []
    L.airband_hip_process_device.argtypes = [vp, vp, sz, vp]
    L.airband_hip_collect.argtypes = [vp, vp, vp, vp, vp]
    L.airband_hip_collect_channels.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    L.airband_hip_read_bins_channels.argtypes = [vp, i64, i64, vp, vp]
    L.airband_hip_read_trace_channels.argtypes = [vp, i64, i64, vp]
    L.airband_hip_collect_mixers.argtypes = [vp, vp, vp, vp]
    L.airband_hip_device_results.argtypes = [vp] + [C.POINTER(vp)] * 6
    L.airband_hip_synchronize.argtypes = [vp]
    L.airband_hip_process_bins.argtypes = [vp, vp, vp]
    L.airband_hip_read_bins.argtypes = [vp, vp, vp]
    L.airband_hip_read_trace.argtypes = [vp, vp]
    L.airband_hip_channel_constants.argtypes = [vp, i32, C.POINTER(C.c_double)]
    L.airband_hip_derive_constants.argtypes = [C.POINTER(capi.Config), i32, C.POINTER(C.c_double)]
    L.airband_hip_last_timings.argtypes = [vp, C.POINTER(C.c_float)]
    L.airband_hip_channelizer_name.argtypes = [vp]
    L.airband_hip_channelizer_name.restype = C.c_char_p
    L.airband_hip_set_signal_plan.argtypes = [vp, vp, i32, i32, vp]
    L.airband_hip_generate_iq.argtypes = [vp, vp, sz, u64, sz, u64, i32, vp]
    L.airband_hip_set_signal_plan_shift.argtypes = [vp, i32, C.c_uint32]
    L.airband_hip_regrouped.argtypes = [vp]
    L.airband_hip_dft_selftest.argtypes = [C.POINTER(capi.Config), i32, C.POINTER(C.c_double)]
    L.airband_hip_build_info.argtypes = []
    L.airband_hip_build_info.restype = C.c_char_p
    L.airband_hip_flush.argtypes = [vp]
    L.airband_hip_mixer_enable_input.argtypes = [vp, i32, i32]
    L.airband_hip_device_enable.argtypes = [vp, i32, i32]
    L.airband_hip_gpu_count.argtypes = []
    L.airband_hip_stream_wait_results.argtypes = [vp, vp]
    L.airband_hip_timing_totals.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
    _lib = L
    return L


def make_config(devices: Sequence[dict], *, wave_rate: int, fft_log: int = 9, fm_demod: int = 0, flags: int = 0, hip_device: int = 0):
    """devices: list of dict(channels=[channel kwargs | ChannelCfg], sample_rate=, centerfreq=, sfmt=, fullscale=, tau_us=)."""
    keep, devs = [], []
    for dev in devices:
        dc, arr = capi.device_cfg(**dev)
        keep.append(arr)
        devs.append(dc)
    darr = (capi.DeviceCfg * len(devs))(*devs)
    keep.append(darr)
    cfg = capi.Config(capi.ABI_VERSION, flags, fft_log, wave_rate, fm_demod, hip_device, len(devs), C.cast(darr, C.POINTER(capi.DeviceCfg)))
    return cfg, keep


def derive_constants(devices: Sequence[dict], channel_index: int, *, wave_rate: int, fft_log: int = 9, fm_demod: int = 0):
    """The 16 derived per-channel constants, computed by the library's host code (no GPU needed)."""
    L = load_library()
    cfg, keep = make_config(devices, wave_rate=wave_rate, fft_log=fft_log, fm_demod=fm_demod)
    v = (C.c_double * 16)()
    rc = L.airband_hip_derive_constants(C.byref(cfg), channel_index, v)
    if rc != 0:
        raise AirbandError(rc, (L.airband_hip_last_error(None) or b"").decode())
    return list(v)


def dft_selftest(devices: Sequence[dict], *, wave_rate: int, fft_log: int = 9, windows: int = 2) -> float:
    """Largest relative error of the matrix-core channelizer's coefficient tables for this configuration (host arithmetic, no GPU)."""
    L = load_library()
    cfg, keep = make_config(devices, wave_rate=wave_rate, fft_log=fft_log)
    err = C.c_double(0.0)
    rc = L.airband_hip_dft_selftest(C.byref(cfg), windows, C.byref(err))
    if rc != 0:
        raise AirbandError(rc, (L.airband_hip_last_error(None) or b"").decode())
    return float(err.value)


class AirbandHip:
    """One handle = the dongles demodulated on one GPU (the reference's demodulate() thread for a device shard,
    src/rtl_airband.cpp:1052-1086)."""

    def __init__(self, devices: Sequence[dict], *, wave_rate: int, fft_log: int = 9, fm_demod: int = 0, flags: int = 0, hip_device: int = 0):
        self.L = load_library()
        self.devices = list(devices)
        cfg, self._keep = make_config(devices, wave_rate=wave_rate, fft_log=fft_log, fm_demod=fm_demod, flags=flags, hip_device=hip_device)
        h = C.c_void_p()
        rc = self.L.airband_hip_prepare(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise AirbandError(rc, (self.L.airband_hip_last_error(None) or b"").decode())
        self.h = h
        g = capi.Geometry()
        self._check(self.L.airband_hip_get_geometry(self.h, C.byref(g)))
        self.geometry = g
        self.B = g.wave_batch
        self.total_channels = g.total_channels
        self.n_mixers = 0

    # ---- plumbing ------------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc < 0:
            raise AirbandError(rc, (self.L.airband_hip_last_error(self.h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.airband_hip_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- data path -----------------------------------------------------------------------------------
    def submit(self, dev: int, iq: np.ndarray) -> int:
        iq = np.ascontiguousarray(iq)
        return self._check(self.L.airband_hip_submit(self.h, dev, iq.ctypes.data, iq.nbytes))

    def process(self) -> bool:
        rc = self.L.airband_hip_process(self.h)
        if rc == capi.EAGAIN:
            return False
        self._check(rc)
        return True

    def process_device(self, d_iq_ptr: int, stride_bytes: int, stream: int = 0):
        self._check(self.L.airband_hip_process_device(self.h, C.c_void_p(d_iq_ptr), stride_bytes, C.c_void_p(stream)))

    def collect(self, *, iq: bool = False, stats: bool = False, first_channel: Optional[int] = None, n_channels: Optional[int] = None):
        """Results of the last batch; with first_channel / n_channels only that channel range (airband_hip_collect_channels:
        repeatable, does not mark the batch as collected)."""
        ranged = first_channel is not None
        n, B = (int(n_channels) if ranged else self.total_channels), self.B
        wave = np.empty((n, B), np.float32)
        axc = np.empty((n,), np.uint8)
        iqo = np.empty((n, 2 * B), np.float32) if iq else None
        st = (capi.ChannelStats * n)() if stats else None
        args = (wave.ctypes.data, iqo.ctypes.data if iq else None, axc.ctypes.data, C.cast(st, C.c_void_p) if stats else None)
        if ranged:
            self._check(self.L.airband_hip_collect_channels(self.h, int(first_channel), n, *args))
        else:
            self._check(self.L.airband_hip_collect(self.h, *args))
        out = dict(waveout=wave, axc=axc)
        if iq:
            out["iq_out"] = iqo
        if stats:
            out["stats"] = [{f[0]: getattr(s, f[0]) for f in capi.ChannelStats._fields_} for s in st]
        return out

    def synchronize(self):
        self._check(self.L.airband_hip_synchronize(self.h))

    def stream_wait_results(self, stream_ptr: int):
        """Make a consumer's hipStream_t wait (on the GPU) for the results of the last completed batch."""
        self._check(self.L.airband_hip_stream_wait_results(self.h, C.c_void_p(stream_ptr)))

    def flush(self):
        """Pipelined handles (FLAG_PIPELINE): run stage 2 of the batch the last process call started."""
        self._check(self.L.airband_hip_flush(self.h))

    def process_bins(self, wavein: np.ndarray, iq_in: np.ndarray):
        wavein = np.ascontiguousarray(wavein, np.float32)
        iq_in = np.ascontiguousarray(iq_in, np.float32)
        assert wavein.shape == (self.total_channels, self.B) and iq_in.shape == (self.total_channels, 2 * self.B)
        self._check(self.L.airband_hip_process_bins(self.h, wavein.ctypes.data, iq_in.ctypes.data))

    def read_bins(self, first_channel: int = 0, n_channels: Optional[int] = None):
        n, B = (self.total_channels - first_channel if n_channels is None else int(n_channels)), self.B
        w = np.empty((n, B), np.float32)
        q = np.empty((n, 2 * B), np.float32)
        self._check(self.L.airband_hip_read_bins_channels(self.h, int(first_channel), n, w.ctypes.data, q.ctypes.data))
        return w, q

    def read_trace(self, first_channel: int = 0, n_channels: Optional[int] = None) -> np.ndarray:
        n = self.total_channels - first_channel if n_channels is None else int(n_channels)
        t = np.empty((n, self.B), np.uint8)
        self._check(self.L.airband_hip_read_trace_channels(self.h, int(first_channel), n, t.ctypes.data))
        return t

    def constants(self, channel_index: int):
        v = (C.c_double * 16)()
        self._check(self.L.airband_hip_channel_constants(self.h, channel_index, v))
        return list(v)

    def last_timings(self):
        v = (C.c_float * 4)()
        self._check(self.L.airband_hip_last_timings(self.h, v))
        return dict(channelizer_ms=v[0], demod_ms=v[1], emit_ms=v[2], batch_ms=v[3])

    def timing_totals(self, reset: bool = False):
        """Sums of the per-stage GPU times over the batches finished since the last reset (waits for the enqueued ones)."""
        v = (C.c_double * 4)()
        n = C.c_int64(0)
        self._check(self.L.airband_hip_timing_totals(self.h, v, C.byref(n), 1 if reset else 0))
        return dict(channelizer_ms=v[0], demod_ms=v[1], emit_ms=v[2], batch_ms=v[3], batches=int(n.value))

    def channelizer_name(self) -> str:
        return self.L.airband_hip_channelizer_name(self.h).decode()

    def build_info(self) -> str:
        return self.L.airband_hip_build_info().decode()

    # ---- mixers ---------------------------------------------------------------------------------------
    def set_mixers(self, n_mixers: int, inputs: Sequence[tuple]):
        """inputs: (device, channel, mixer, ampfactor, balance) tuples in connection order."""
        arr = (capi.MixerInput * len(inputs))(*[capi.MixerInput(int(a), int(b), int(c), float(d), float(e)) for a, b, c, d, e in inputs])
        self._check(self.L.airband_hip_set_mixers(self.h, n_mixers, arr, len(inputs)))
        self.n_mixers = n_mixers

    def mixer_enable_input(self, input_index: int, enabled: bool):
        self._check(self.L.airband_hip_mixer_enable_input(self.h, input_index, 1 if enabled else 0))

    def mixer_set_stereo(self, mixer: int, stereo: bool):
        self._check(self.L.airband_hip_mixer_set_stereo(self.h, int(mixer), 1 if stereo else 0))

    # ---- the mixer exchange over RCCL (include/airband_hip.h): one handle per GPU, in one process or in one process each -----------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = load_library().airband_hip_comm_unique_id(buf)
        if rc < 0:
            raise AirbandError(rc, (load_library().airband_hip_last_error(None) or b"").decode())
        return bytes(buf)

    def comm_init_rank(self, unique_id: bytes, nranks: int, rank: int):
        assert len(unique_id) == 128
        self._check(self.L.airband_hip_comm_init_rank(self.h, (C.c_uint8 * 128).from_buffer_copy(unique_id), int(nranks), int(rank)))

    def allreduce_mixers(self, stream: int = 0):
        self._check(self.L.airband_hip_allreduce_mixers(self.h, C.c_void_p(stream)))

    def add_mixers(self, src: "AirbandHip"):
        self._check(self.L.airband_hip_add_mixers(self.h, src.h))

    def clear_mixers(self):
        """A handle that ran no batch this round adds nothing to the exchange (mixer_disable_input(), src/mixer.cpp:96-112)."""
        self._check(self.L.airband_hip_clear_mixers(self.h))

    @staticmethod
    def comm_init_all(handles: Sequence["AirbandHip"]):
        """One process, one handle per GPU (the reference-side shim's form): ncclCommInitAll over the handles' GPUs."""
        L = load_library()
        arr = (C.c_void_p * len(handles))(*[h.h for h in handles])
        rc = L.airband_hip_comm_init_all(arr, len(handles))
        if rc < 0:
            raise AirbandError(rc, (L.airband_hip_last_error(handles[0].h) or L.airband_hip_last_error(None) or b"").decode())

    @staticmethod
    def allreduce_mixers_group(handles: Sequence["AirbandHip"]):
        """The per-batch exchange of one thread that owns several handles: every handle's all-reduce inside one RCCL group."""
        L = load_library()
        rc = L.airband_hip_comm_group_begin()
        if rc < 0:
            raise AirbandError(rc, (L.airband_hip_last_error(None) or b"").decode())
        try:
            for h in handles:
                h.allreduce_mixers()
        finally:  # a failing rank must not leave the thread's RCCL group open: every later collective of the process would queue behind it
            rc = L.airband_hip_comm_group_end()
        if rc < 0:
            raise AirbandError(rc, (L.airband_hip_last_error(None) or b"").decode())

    def batch_ready(self) -> bool:
        rc = self.L.airband_hip_batch_ready(self.h)
        if rc == capi.EAGAIN:
            return False
        self._check(rc)
        return True

    def device_enable(self, dev: int, enabled: bool):
        """Switch a dongle off / on: the reference's handling of a failed input (src/rtl_airband.cpp:383-391)."""
        self._check(self.L.airband_hip_device_enable(self.h, dev, 1 if enabled else 0))

    def collect_mixers(self):
        left = np.empty((self.n_mixers, self.B), np.float32)
        right = np.empty((self.n_mixers, self.B), np.float32)
        sig = np.empty((self.n_mixers,), np.uint8)
        self._check(self.L.airband_hip_collect_mixers(self.h, left.ctypes.data, right.ctypes.data, sig.ctypes.data))
        return left, right, sig

    def device_results(self) -> dict:
        ptrs = [C.c_void_p() for _ in range(6)]
        self._check(self.L.airband_hip_device_results(self.h, *[C.byref(p) for p in ptrs]))
        names = ["waveout", "iq_out", "axc", "mix_left", "mix_right", "mix_signal"]
        return {k: (p.value or 0) for k, p in zip(names, ptrs)}

    # ---- synthetic dongles -----------------------------------------------------------------------------
    def set_signal_plan(self, carriers, noise_q8: Optional[int] = None):
        tab = siggen.carrier_table(carriers)
        sin = siggen.sin_table()
        if noise_q8 is None:
            noise_q8 = siggen.noise_mul_q8()
        self._check(self.L.airband_hip_set_signal_plan(self.h, tab.ctypes.data, len(carriers), int(noise_q8), sin.ctypes.data))

    def stage2_regrouped(self) -> bool:
        return bool(self.L.airband_hip_regrouped(self.h))

    def set_signal_plan_shift(self, n_plans: int, shift_hz: float, sample_rate: int):
        """Dongle d's carrier c is generated ((d mod n_plans) >> 2c & 3) * shift_hz above the plan's frequency (siggen.plan_shift_bins / generate_u8's twin)."""
        self._check(self.L.airband_hip_set_signal_plan_shift(self.h, int(n_plans), siggen._turns(shift_hz, sample_rate)))

    def generate_iq(self, d_iq_ptr: int, stride_bytes: int, start_byte: int, nbytes: int, *, seed: int = 0x5EED, device_index_offset: int = 0, stream: int = 0):
        self._check(self.L.airband_hip_generate_iq(self.h, C.c_void_p(d_iq_ptr), stride_bytes, start_byte, nbytes, seed, device_index_offset, C.c_void_p(stream)))
