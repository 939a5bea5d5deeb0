#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s channelized+demodulated (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" = one batch (WAVE_BATCH output samples per channel = 1/8 s of signal = 320 000 complex samples per dongle)
of the whole hot path -- channelizer kernel, demod/squelch/filter kernel, emit (+ mixer sum) -- over every dongle of
the rank, with the raw u8 I/Q already resident in HBM (generated on the GPU before the timed region).
One process per GPU; dongles are independent, so ranks share nothing on the data path (weak scaling: per-GPU
work is fixed).  With N > 1 and mixers enabled, the per-rank mixer sums are all-reduced over RCCL each step
(config #5 of BASELINE.json) -- the only exchange step the reference's data flow has (src/mixer.cpp:133-140).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    # name: (dongles per GPU, mixed AM/NFM+CTCSS?, wave_rate)
    "cfg2": dict(dongles=1024, mixed=False, wave_rate=8000, desc="1 024 synthetic 2.56 MS/s dongles x 8 AM channels, FFT 512 (BASELINE configs[1])"),
    "cfg3": dict(dongles=65536, mixed=True, wave_rate=16000, desc="65 536 dongles x 8 channels mixed AM/NFM + CTCSS, FFT 512 (BASELINE configs[2])"),
    "cfg4": dict(dongles=32768, mixed=True, wave_rate=16000, desc="32 768 dongles per GPU x 8 channels mixed (BASELINE configs[3]/[4] shard)"),
    "tiny": dict(dongles=64, mixed=True, wave_rate=16000, desc="64 dongles x 8 mixed channels (plumbing check)"),
}
SAMPLES_PER_BATCH = 320_000  # complex samples per dongle per batch (2.56 MS/s / 8)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def cpu_baseline(pkg, hip, devices, wave_rate, mixed, seconds):
    """The reference's own demodulate() (oracle/_ref, compiled in place) timed on this box's host cores, on a
    bounded sample: min(nproc, 16) dongles of the same workload, one pthread per dongle shard."""
    import numpy as np
    import torch

    try:
        import pyref
    except Exception as e:  # noqa: BLE001
        return dict(value=None, unit="Msamples/s", cores=0, kind="reference", sample="unavailable: %r" % (e,))
    nfm = wave_rate == 16000
    threads = max(1, min(os.cpu_count() or 1, 16))
    n_dev = threads
    rb = pyref.ring_bytes()
    buf = torch.zeros((n_dev, rb), dtype=torch.uint8, device="cuda")
    # same generator, same plan; dongle indices 0..n_dev-1
    sub = pkg.AirbandHip(devices[:n_dev], wave_rate=wave_rate)
    _, carriers = pkg.siggen.baseline_plan(mixed=mixed)
    sub.set_signal_plan(carriers)
    sub.generate_iq(buf.data_ptr(), buf.stride(0), 0, rb)
    sub.synchronize()
    host = buf.cpu().numpy()
    sub.close()
    del buf
    kind = "reference"
    if pyref.have_ref(nfm):
        fast = os.path.exists(pyref.ref_lib_path(nfm, True))
        batches, el = pyref.reference_throughput(devices[:n_dev], [host[d] for d in range(n_dev)], seconds, threads, nfm=nfm, fast=fast)
        note = "oracle/_ref (%s build) " % ("-O3 -march=native -ffast-math" if fast else "-O2 strict")
        # SURVEY 8d asks for T = 1 next to T = nproc: one demodulate() thread over one dongle, a few seconds
        b1, e1 = pyref.reference_throughput(devices[:1], [host[0]], min(4.0, seconds), 1, nfm=nfm, fast=fast)
        one_thread = round(b1 * SAMPLES_PER_BATCH / e1 / 1e6, 3)
    else:
        import pyoracle

        kind, threads = "port", 1
        one_thread = None
        orc = pyoracle.Oracle(devices[:1], wave_rate=wave_rate)
        t0 = time.time()
        batches = 0
        while time.time() - t0 < seconds:
            batches += orc.run_device(0, host[0][:2 * 320000], 4)
        el = time.time() - t0
        note = "oracle C restatement "
    value = batches * SAMPLES_PER_BATCH / el / 1e6
    if one_thread is None:
        one_thread = round(value, 3)
    return dict(value=round(value, 3), unit="Msamples/s", cores=threads, kind=kind, value_1_thread=one_thread,
                sample=note + "%d dongles x 8 ch of the same workload for %.1f s wall (%d batches); FFT behind fftwf_* is oracle_fft.c, FFTW3 is not installed" %
                (n_dev, el, batches))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("AIRBAND_BENCH_WORKLOAD", "cfg3"), choices=sorted(WORKLOADS))
    ap.add_argument("--dongles", type=int, default=0, help="override dongles per GPU")
    ap.add_argument("--ring", type=int, default=3, help="distinct I/Q batches kept in HBM and cycled through")
    ap.add_argument("--mixers", type=int, default=0, help="number of mixers (BASELINE configs[4]: 64). Default 0 at every N, so that per-GPU work is the same "
                    "from 1 to 8 GPUs (configs[1]-[3] have no exchange step); with mixers and N > 1 the per-rank sums are all-reduced over RCCL every step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-path", action="store_true", help="additionally time the host-buffer path (submit over PCIe) on a small slice; reported separately, never as value")
    ap.add_argument("--pipelined", action="store_true", help="AIRBAND_HIP_FLAG_PIPELINE: stage 1 of batch k beside stage 2 of batch k-1 (results one batch late). "
                    "Measured gain at configs[2]: ~3 %% -- both halves lean on the same HBM / vector-issue capacity -- and the channelizer's own launch time, "
                    "which the roofline figure is built on, is then no longer that of the kernel alone; so the default is one batch at a time")
    ap.add_argument("--sequential", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--force-dist", action="store_true", help="initialise a process group even at world size 1 (plumbing check of the RCCL leg)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libairband_hip has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("rtlsdr-airband_amd")
    wl = WORKLOADS[args.workload]
    D = args.dongles or wl["dongles"]
    mixed, wave_rate = wl["mixed"], wl["wave_rate"]
    n_mixers = max(0, args.mixers)

    chans, carriers = pkg.siggen.baseline_plan(mixed=mixed)
    devices = [dict(channels=chans) for _ in range(D)]
    # --pipelined (AIRBAND_HIP_FLAG_PIPELINE): a step enqueues stage 1 of its batch beside stage 2 of the previous one -- every step
    # still does one full stage 1 and one full stage 2, of consecutive batches.  AIRBAND_BENCH_FLAGS adds AIRBAND_HIP_FLAG_* bits for
    # experiments (e.g. 8 = demod kinds one after the other, for per-kernel profiles).
    flags = int(os.environ.get("AIRBAND_BENCH_FLAGS", "0"), 0) | (pkg.capi.FLAG_PIPELINE if args.pipelined else 0)
    hip = pkg.AirbandHip(devices, wave_rate=wave_rate, hip_device=local_rank, flags=flags)
    g = hip.geometry
    if n_mixers:
        base = rank * D
        hip.set_mixers(n_mixers, [(d, c, ((base + d) * 8 + c) % n_mixers, 1.0, 0.0) for d in range(D) for c in range(8)])
    hip.set_signal_plan(carriers)

    # HBM-resident I/Q: lead-in + (ring + 1) batches + look-ahead per dongle, generated on the GPU
    lead = g.first_batch_bytes - g.batch_bytes
    span = lead + (args.ring + 1) * g.batch_bytes + g.lookahead_bytes
    stride = (span + 255) // 256 * 256
    iq = torch.empty((D, stride), dtype=torch.uint8, device="cuda")
    hip.generate_iq(iq.data_ptr(), stride, 0, span, seed=0x5EED, device_index_offset=rank * D)
    hip.synchronize()
    torch.cuda.synchronize()

    res = hip.device_results()
    mix_t = None
    if n_mixers and use_dist:
        # torch views over the library's device-side mixer sums, for the RCCL all-reduce
        class _Ptr:  # __cuda_array_interface__ shim
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = dict(shape=shape, typestr=typestr, data=(ptr, False), version=2)
        mix_t = torch.as_tensor(_Ptr(res["mix_left"], (n_mixers, hip.B), "<f4"), device="cuda")
        sig_t = torch.as_tensor(_Ptr(res["mix_signal"], (n_mixers,), "|u1"), device="cuda")

    # the RCCL all-reduce is issued on torch's stream: that stream waits (on the GPU) for the batch's mixer sums, and the next
    # process call orders its overwrite of them behind the all-reduce -- no host synchronisation inside a step
    cstream = torch.cuda.Stream() if mix_t is not None else None   # a real stream: torch's default one is the NULL handle
    consumer = cstream.cuda_stream if cstream is not None else 0

    def step(i):
        if i == 0:
            off = 0
        else:
            off = g.first_batch_bytes + ((i - 1) % args.ring) * g.batch_bytes
        hip.process_device(iq.data_ptr() + off, stride, consumer)
        if mix_t is not None:
            hip.stream_wait_results(consumer)
            with torch.cuda.stream(cstream):
                dist.all_reduce(mix_t, op=dist.ReduceOp.SUM)      # mixer sum over xGMI (src/mixer.cpp:133-140)
                dist.all_reduce(sig_t, op=dist.ReduceOp.MAX)      # axcindicate of the mixer (src/mixer.cpp:209)

    def sync():
        hip.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    sync()
    hip.timing_totals(reset=True)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    sync()
    elapsed = time.perf_counter() - t0
    # HIP events the library records around each kernel on the stream it runs on, read once after the timed region
    tt = hip.timing_totals()
    nb = max(1, tt["batches"])
    chan_ms, demod_ms, emit_ms = [tt["channelizer_ms"] / nb], [tt["demod_ms"] / nb], [tt["emit_ms"] / nb]
    hip.flush()
    sync()
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    total_samples = float(D) * world * SAMPLES_PER_BATCH * args.steps
    value = total_samples / elapsed / 1e6
    hop = g.batch_bytes // (2 * hip.B)
    alg_bytes_per_sample = 2.0 + 8 * 4.0 / hop          # u8 I/Q in, 8 channels of float audio out per hop
    ch_ms = float(np.mean(chan_ms)) if chan_ms else None
    roofline = None
    if ch_ms:
        achieved = alg_bytes_per_sample * D * SAMPLES_PER_BATCH / (ch_ms * 1e-3) / 1e9
        traffic = None
        cal = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(cal):
            try:
                traffic = json.load(open(cal)).get(args.workload, {}).get(hip.channelizer_name())
            except Exception:  # noqa: BLE001
                traffic = None
        roofline = dict(bound="hbm", kernel=hip.channelizer_name(), achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=traffic, avg_launch_ms=round(ch_ms, 4), algorithmic_bytes_per_launch=alg_bytes_per_sample * D * SAMPLES_PER_BATCH)
    out = dict(metric="IQ Msamples/sec channelized+demodulated per node; % HBM roofline", value=round(value, 2), unit="Msamples/s", n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
               data="synthetic", config=dict(workload=wl["desc"], dongles_per_gpu=D, channels_per_dongle=8, fft_size=g.fft_size, wave_rate=wave_rate,
                                               sample_format="u8", iq_resident="HBM", ring_batches=args.ring, mixers=n_mixers,
                                               schedule="pipelined: stage 1 of batch k beside stage 2 of batch k-1" if args.pipelined else "one batch at a time",
                                               parallelism="dongle-sharded x%d, %s" % (world, "RCCL all-reduce of mixer sums" if mix_t is not None else "no collective"),
                                               channelizer=hip.channelizer_name()),
               roofline=roofline,
               stage_ms=dict(channelizer=ch_ms, demod=float(np.mean(demod_ms)) if demod_ms else None, mixers_and_iq_out=float(np.mean(emit_ms)) if emit_ms else None),
               realtime_dongles=int(value / 2.56))
    if rank == 0 and world == 1 and args.host_path:
        # host-buffer path: the shim of INTEGRATION.md feeding pageable host memory through submit()/process()
        nd = min(D, 512)
        sub = pkg.AirbandHip(devices[:nd], wave_rate=wave_rate, hip_device=local_rank)
        gg = sub.geometry
        host = iq[:nd, :gg.first_batch_bytes + 3 * gg.batch_bytes + gg.lookahead_bytes].cpu().numpy()
        for d in range(nd):
            sub.submit(d, host[d, :gg.first_batch_bytes + gg.lookahead_bytes])
        sub.process(); sub.synchronize()
        t1 = time.perf_counter()
        off = gg.first_batch_bytes + gg.lookahead_bytes
        for k in range(3):
            for d in range(nd):
                sub.submit(d, host[d, off:off + gg.batch_bytes])
            sub.process()
            off += gg.batch_bytes
        sub.synchronize()
        el = time.perf_counter() - t1
        out["host_path"] = dict(value=round(nd * SAMPLES_PER_BATCH * 3 / el / 1e6, 1), unit="Msamples/s", dongles=nd, note="pageable host buffers -> pinned staging -> PCIe; includes the H2D copy")
        sub.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(pkg, hip, devices, wave_rate, mixed, args.cpu_seconds)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = dict(value=None, unit="Msamples/s", cores=0, kind="reference", sample="failed: %r" % (e,))
    elif rank == 0:
        out["cpu_baseline"] = None
    hip.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which a pipe buffers until exit: push it out first so that the JSON is the LAST line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
