"""oracle/pyref.py -- TEST INFRASTRUCTURE ONLY: ctypes driver for oracle/_ref/libairband_ref_*.so
(the real reference compiled in place, see oracle/ref_harness.cpp).  One process can host ONE reference
instance (the reference keeps its configuration in globals), so `run_reference` forks a worker."""
from __future__ import annotations

import ctypes as C
import importlib
import multiprocessing as mp
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
capi = importlib.import_module("rtlsdr-airband_amd.capi")


def ref_lib_path(nfm: bool, fast=False) -> str:
    """fast: False = parity build (-O2, IEEE, float64 FFT behind fftwf_*); True / "fast" = the reference's own flags
    (-O3 -march=native -ffast-math), same FFT; "fast32" = those flags with the float radix-4 FFT (oracle_fft32.c) -- timing only."""
    suffix = "" if not fast else ("_" + fast if isinstance(fast, str) else "_fast")
    return os.path.join(HERE, "_ref", "libairband_ref_%s%s.so" % ("nfm" if nfm else "am", suffix))


def have_ref(nfm: bool = True) -> bool:
    return os.path.exists(ref_lib_path(nfm))


def _load(nfm: bool, fast=False) -> C.CDLL:
    lib = C.CDLL(ref_lib_path(nfm, fast))
    lib.refh_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.refh_add_device.argtypes = [C.c_int, C.POINTER(capi.DeviceCfg)]
    lib.refh_start.argtypes = [C.c_int]
    lib.refh_run_device.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.refh_channel_stats.argtypes = [C.c_int, C.c_int, C.POINTER(capi.ChannelStats)]
    lib.refh_channel_constants.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.refh_set_trace_dir.argtypes = [C.c_char_p]
    lib.refh_throughput.argtypes = [C.POINTER(C.c_void_p), C.c_double, C.c_int, C.POINTER(C.c_double)]
    lib.refh_throughput.restype = C.c_long
    return lib


def load_units(nfm: bool) -> C.CDLL:
    """The reference's stand-alone pieces (Squelch, CTCSS, filters, LUTs) for unit-level pinning."""
    lib = _load(nfm)
    lib.refh_init(1, 9, 0, -1)  # fft_size 512, sincosf_lut_init()
    f32, vp, i = C.c_float, C.c_void_p, C.c_int
    lib.refh_squelch_new.restype = vp
    lib.refh_squelch_new.argtypes = [f32, i, f32]
    lib.refh_squelch_raw.argtypes = [vp, vp, i, vp, vp, vp]
    lib.refh_squelch_raw_filtered.argtypes = [vp, vp, vp, i, vp, vp, vp]
    lib.refh_squelch_raw_audio.argtypes = [vp, vp, vp, i, vp]
    lib.refh_squelch_audio_raw.argtypes = [vp, vp, vp, i, vp]
    lib.refh_squelch_counts.argtypes = [vp, vp]
    lib.refh_squelch_free.argtypes = [vp]
    lib.refh_ctcss_run.argtypes = [f32, f32, i, vp, i, vp, vp]
    lib.refh_tone_coeff.restype = f32
    lib.refh_tone_coeff.argtypes = [f32, f32, i]
    lib.refh_notch_run.argtypes = [f32, f32, f32, vp, i]
    lib.refh_lowpass_run.argtypes = [f32, f32, vp, vp, i]
    lib.refh_sincos_lut.argtypes = [C.c_uint32, C.POINTER(f32), C.POINTER(f32)]
    lib.refh_dbfs_to_level.restype = f32
    lib.refh_dbfs_to_level.argtypes = [f32]
    if nfm:
        lib.refh_fast_atan2.restype = f32
        lib.refh_fast_atan2.argtypes = [f32, f32]
        for n in ("refh_polar_disc_fast", "refh_fm_quadri_demod"):
            getattr(lib, n).restype = f32
            getattr(lib, n).argtypes = [f32] * 4
    return lib


TRACE_DT = np.dtype([("raw_input", np.float32), ("filtered_input", np.float32), ("audio_input", np.float32), ("noise_floor", np.float32),
                     ("pre_filter_capped", np.float32), ("post_filter_capped", np.float32), ("current_state", np.intc), ("delay", np.intc),
                     ("low_signalcount", np.intc), ("ctcss_fast_has_tone", np.intc), ("ctcss_slow_has_tone", np.intc)])


def _worker(q, nfm, fft_log, fm_demod, devices, iq_list, n_batches, trace_dir):
    try:
        lib = _load(nfm)
        wb = lib.refh_wave_batch()
        lib.refh_init(len(devices), fft_log, fm_demod, -1)
        if trace_dir:
            lib.refh_set_trace_dir(trace_dir.encode())
        keep = []
        for d, dev in enumerate(devices):
            dc, arr = capi.device_cfg(**dev)
            keep.append(arr)
            rc = lib.refh_add_device(d, C.byref(dc))
            assert rc == 0, rc
        lib.refh_start(1)
        res = []
        for d, dev in enumerate(devices):
            nch = len(dev["channels"])
            wave = np.zeros((n_batches, nch, wb), np.float32)
            iqo = np.zeros((n_batches, nch, 2 * wb), np.float32)
            axc = np.zeros((n_batches, nch), np.uint8)
            iq = np.ascontiguousarray(iq_list[d])
            nb = lib.refh_run_device(d, iq.ctypes.data, iq.nbytes, n_batches, wave.ctypes.data, iqo.ctypes.data, axc.ctypes.data)
            stats = []
            consts = []
            for j in range(nch):
                st = capi.ChannelStats()
                lib.refh_channel_stats(d, j, C.byref(st))
                stats.append({f[0]: getattr(st, f[0]) for f in capi.ChannelStats._fields_})
                cv = (C.c_double * 16)()
                lib.refh_channel_constants(d, j, cv)
                consts.append(list(cv))
            res.append(dict(n_batches=nb, waveout=wave[:nb], iq_out=iqo[:nb], axc=axc[:nb], stats=stats, consts=consts))
        lib.refh_stop()
        q.put(("ok", res))
    except BaseException as e:  # noqa: BLE001
        q.put(("err", repr(e)))


def run_reference(devices, iq_list, n_batches, *, nfm: bool, fft_log: int = 9, fm_demod: int = 0, trace_dir: str | None = None):
    """Runs the real reference demodulate() over raw I/Q (bytes per device) and returns, per device,
    dict(n_batches, waveout [nb][C][B], iq_out [nb][C][2B], axc [nb][C], stats, consts).
    `devices` = list of dicts(channels=[channel kwargs...], sample_rate=, centerfreq=, sfmt=, ...)."""
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(q, nfm, fft_log, fm_demod, devices, iq_list, n_batches, trace_dir))
    p.start()
    status, res = q.get()
    p.join()
    if status != "ok":
        raise RuntimeError(res)
    return res


def read_trace(trace_dir: str, d: int, j: int) -> np.ndarray:
    return np.fromfile(os.path.join(trace_dir, "squelch_debug-%d-%d.dat" % (d, j)), dtype=TRACE_DT)


def _tp_worker(q, nfm, fast, fft_log, devices, iq_list, seconds, threads):
    try:
        lib = _load(nfm, fast)
        lib.refh_init(len(devices), fft_log, 0, -1)
        keep = []
        for d, dev in enumerate(devices):
            dc, arr = capi.device_cfg(**dev)
            keep.append(arr)
            assert lib.refh_add_device(d, C.byref(dc)) == 0
        lib.refh_start(threads)
        bufs = [np.ascontiguousarray(x) for x in iq_list]
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        out = (C.c_double * 3)()
        nb = lib.refh_throughput(ptrs, float(seconds), 8, out)  # one consumer thread per 8 devices
        lib.refh_stop()
        q.put(("ok", (int(nb), float(out[2]), int(out[0]), int(out[1]))))
    except BaseException as e:  # noqa: BLE001
        q.put(("err", repr(e)))


def ring_bytes(fft_size: int = 512, bytes_per_sample: int = 1) -> int:
    """Bytes refh_throughput needs per device: the reference's ring (MIN_BUF_SIZE, src/rtl_airband.h:64) + tail."""
    return 2560000 + 2 * bytes_per_sample * fft_size


def reference_throughput(devices, iq_list, seconds: float, threads: int, *, nfm: bool, fast=True, fft_log: int = 9):
    """CPU baseline: the real reference demodulate() in `threads` pthreads over contiguous device shards, rings
    pre-filled and kept full by cursor rewind, one consumer thread per 8 devices.  Returns (batches completed over all devices, elapsed
    seconds, batches the consumers took, batches counted in output_overrun_count): completed = taken + overrun."""
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_tp_worker, args=(q, nfm, fast, fft_log, devices, iq_list, seconds, threads))
    p.start()
    status, res = q.get()
    p.join()
    if status != "ok":
        raise RuntimeError(res)
    return res


def _all_worker(q, nfm, fft_log, fm_demod, devices, iq_list, n_batches, hip_lib, fail_after, end_of_streams, extra=None):
    try:
        extra = extra or {}
        for k, v in (extra.get("env") or {}).items():
            os.environ[k] = v
        lib = _load(nfm, "patched" if hip_lib else False)
        lib.refh_run_all.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_int), C.c_double]
        lib.refh_fail_all_and_wait_exit.argtypes = [C.c_double]
        lib.refh_start_hip.argtypes = [C.c_char_p]
        lib.refh_hip_channel_stats.argtypes = [C.c_int, C.c_int, C.POINTER(capi.ChannelStats)]
        wb = lib.refh_wave_batch()
        lib.refh_init(len(devices), fft_log, fm_demod, -1)
        keep = []
        for d, dev in enumerate(devices):
            dc, arr = capi.device_cfg(**dev)
            keep.append(arr)
            assert lib.refh_add_device(d, C.byref(dc)) == 0
        nd, nch = len(devices), len(devices[0]["channels"])
        # mixers: (n_mixers, [(device, channel, mixer, ampfactor, balance), ...]) -- wired like parse_outputs() / mixer_connect_input() do it
        n_mix, conns = extra.get("mixers") or (0, [])
        if n_mix:
            lib.refh_connect.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
            lib.refh_set_mixer_outputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.refh_mixer_counters.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
            lib.refh_mixer_counters.restype = None
            assert lib.refh_add_mixers(n_mix) == 0
            for (d, j, m, amp, bal) in conns:
                assert lib.refh_connect(d, j, m, amp, bal) >= 0
        # file inputs: {device: (path, speedup_factor)} -- the reference's own input-file.cpp driver instead of the harness's rx role
        files = extra.get("file_inputs") or {}
        lib.refh_use_file_input.argtypes = [C.c_int, C.c_char_p, C.c_float]
        for d, (path, speedup) in files.items():
            assert lib.refh_use_file_input(int(d), path.encode(), float(speedup)) == 0
        if extra.get("tui_path"):
            lib.refh_tui_begin.argtypes = [C.c_char_p]
            assert lib.refh_tui_begin(extra["tui_path"].encode()) == 0
        mix_l = mix_r = mix_a = mix_got = None
        if n_mix:
            mix_l = np.zeros((n_mix, n_batches, wb), np.float32)
            mix_r = np.zeros((n_mix, n_batches, wb), np.float32)
            mix_a = np.zeros((n_mix, n_batches), np.uint8)
            mix_got = (C.c_int * n_mix)()
            lib.refh_set_mixer_outputs(mix_l.ctypes.data, mix_r.ctypes.data, mix_a.ctypes.data, C.cast(mix_got, C.c_void_p))
            if not hip_lib:
                assert lib.refh_start_mixer_thread() == 0
        rc = lib.refh_start_hip(hip_lib.encode()) if hip_lib else lib.refh_start(1)
        assert rc == 0, rc
        if n_mix and hip_lib:
            # mixer_thread() only once demodulate_hip() has claimed its mixers: started first (as main() does) it emits silence while the HIP runtime
            # comes up, and the batch-by-batch comparison with the demodulate() run would be off by that many batches
            lib.refh_wait_mixers_served.argtypes = [C.c_double]
            lib.refh_wait_mixers_served(30.0)
            assert lib.refh_start_mixer_thread() == 0
        if files:
            assert lib.refh_start_inputs() == 0
        wave = np.zeros((nd, n_batches, nch, wb), np.float32)
        iqo = np.zeros((nd, n_batches, nch, 2 * wb), np.float32)
        axc = np.zeros((nd, n_batches, nch), np.uint8)
        bufs = [np.ascontiguousarray(x) for x in iq_list]
        ptrs = (C.c_void_p * nd)(*[b.ctypes.data for b in bufs])
        sizes = (C.c_size_t * nd)(*[b.nbytes for b in bufs])
        fa = (C.c_int * nd)(*[(-1 if fail_after is None or fail_after[d] is None else int(fail_after[d])) for d in range(nd)])
        got = (C.c_int * nd)()
        short = lib.refh_run_all(ptrs, sizes, fa, n_batches, wave.ctypes.data, iqo.ctypes.data, axc.ctypes.data, got, float(extra.get("timeout_s", 120.0)))
        if extra.get("tui_path"):
            lib.refh_tui_end()
        stats = []
        for d in range(nd):
            row = []
            for j in range(nch):
                st = capi.ChannelStats()
                (lib.refh_hip_channel_stats if hip_lib else lib.refh_channel_stats)(d, j, C.byref(st))
                row.append({f[0]: getattr(st, f[0]) for f in capi.ChannelStats._fields_})
            stats.append(row)
        res = dict(n_batches=n_batches + short, batches=[int(g) for g in got], waveout=wave, iq_out=iqo, axc=axc, stats=stats,
                   outputs_disabled=[lib.refh_outputs_disabled(d) for d in range(nd)], devices_running=lib.refh_devices_running(),
                   input_state=[lib.refh_input_state(d) for d in range(nd)], output_overruns=[lib.refh_output_overruns(d) for d in range(nd)])
        if n_mix:
            res["mix_left"], res["mix_right"], res["mix_axc"], res["mix_batches"] = mix_l, mix_r, mix_a, [int(x) for x in mix_got]
            cnt = []
            for m in range(n_mix):
                c4 = (C.c_uint64 * 4)()
                lib.refh_mixer_counters(m, c4)
                cnt.append(dict(output_overruns=int(c4[0]), enabled=int(c4[1]), inputs=int(c4[2]), gpu_served=int(c4[3])))
            res["mixers"] = cnt
        if extra.get("wait_exit_s"):  # file inputs end on their own (INPUT_FAILED at end of file): the demodulator must then set do_exit and return
            lib.refh_wait_exit.argtypes = [C.c_double]
            res["exited_on_its_own"] = bool(lib.refh_wait_exit(float(extra["wait_exit_s"])))
            res["devices_running_at_exit"] = lib.refh_devices_running()
            res["input_state_at_exit"] = [lib.refh_input_state(d) for d in range(nd)]
        if end_of_streams:
            res["exited_on_its_own"] = bool(lib.refh_fail_all_and_wait_exit(20.0))
            res["devices_running_at_exit"] = lib.refh_devices_running()
            res["outputs_disabled_at_exit"] = [lib.refh_outputs_disabled(d) for d in range(nd)]
        lib.refh_stop()
        q.put(("ok", res))
    except BaseException as e:  # noqa: BLE001
        q.put(("err", repr(e)))


def run_reference_all(devices, iq_list, n_batches, *, nfm: bool, fft_log: int = 9, fm_demod: int = 0, hip_lib: str | None = None, fail_after=None,
                      end_of_streams: bool = False, mixers=None, file_inputs=None, tui_path=None, env=None, timeout_s: float = 120.0, wait_exit_s: float = 0.0):
    """All devices fed concurrently through the reference's own rings.  hip_lib=None: the reference's demodulate();
    hip_lib=path to libairband_hip.so: the harness built from the PATCHED reference (integration/airband_hip.patch applied to a
    scratch copy, integration/demod_hip.cpp compiled verbatim): demodulate_hip() instead of demodulate(), statistics read back
    through the reference's own Squelch getters.
    fail_after[d] = k: device d's input reports INPUT_FAILED (a file input at end of file) after it has delivered k batches.
    end_of_streams: afterwards every input fails; `exited_on_its_own` tells whether the demodulator then set do_exit and returned.
    mixers=(n, [(device, channel, mixer, ampfactor, balance), ...]): the reference's mixer_t objects and O_MIXER outputs, its mixer_thread() started
    (with the HIP backend the shim serves them on the GPU); file_inputs={device: (path, speedup_factor)}: the reference's own file input driver
    (src/input-file.cpp) feeds that device; tui_path: the waterfall (`tui` = 1) written to that file; env: environment for the child (AIRBAND_HIP_GPUS)."""
    ctx = mp.get_context("spawn" if hip_lib else "fork")
    q = ctx.Queue()
    extra = dict(mixers=mixers, file_inputs=file_inputs, tui_path=tui_path, env=env, timeout_s=timeout_s, wait_exit_s=wait_exit_s)
    p = ctx.Process(target=_all_worker, args=(q, nfm, fft_log, fm_demod, devices, iq_list, n_batches, hip_lib, fail_after, end_of_streams, extra))
    p.start()
    status, res = q.get()
    p.join()
    if status != "ok":
        raise RuntimeError(res)
    return res
