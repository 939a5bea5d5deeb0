#!/usr/bin/env python3
"""Round 5: the experiment for round 4's rare event (profiles/r04_experiments.md I -- about one transform in 10^9 wrong on the wavefront FFT's LDS exchange while a
dozen PROCESSES shared the GPU, none in 4.6e9 with one process).  One worker of a P-process load.  A handle whose dongles ALL replay dongle 0's bytes runs on the
exchange kernel (u8 at fft 512 with FORCE_FFT, or CF32 at fft 4096 / 8192: 8 / 16 decimated transforms per hop); after every batch the stage-1 bins of every dongle
-- |bin| and raw I/Q of every hop -- are compared with dongle 0's, bit for bit (dongles are independent and identical).  Two shapes of load:
  big    hundreds of dongles per handle, results compared on the GPU (audio rows): ~10^9 hop transforms per process and minute, few launches;
  small  a few dongles per handle (the fuzz campaign's shape: 1 - 5), bins compared on the host: ~10^5 launches per process and minute, the GPU switching
         between the processes' queues all the time; every other small worker feeds the host path (submit / process) as the fuzz did.
The handle is torn down and rebuilt every few dozen batches (the fuzz created and destroyed handles all the time).
usage: r05_exchange_stress.py <tag> <seconds> <u8|f32|f32x> <dongles> <out.jsonl> [host]      (the library under test: AIRBAND_HIP_LIB)"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class DevPtr:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


def main():
    tag, seconds, fmt, D, out_path = sys.argv[1], float(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    host_path = len(sys.argv) > 6 and sys.argv[6] == "host"
    import numpy as np
    import torch

    pkg = importlib.import_module("rtlsdr-airband_amd")
    sg = pkg.siggen
    chans, carriers = sg.baseline_plan(mixed=True)
    wave_rate, ring = 16000, 2
    f32 = fmt.startswith("f32")
    fft_log = (13 if fmt == "f32x" else 12) if f32 else 9
    dev = dict(channels=chans, sfmt=pkg.capi.SFMT_F32) if f32 else dict(channels=chans)
    main_path = os.environ.get("R05_MAIN_PATH") == "1"  # u8 on the int8 matrix-core channelizer + stage 2: the path bench.py measures
    flags = 0 if (f32 or main_path) else pkg.capi.FLAG_FORCE_FFT
    diagnose = os.environ.get("R05_TRACE") == "1"  # per-sample squelch / tone trace: on an event, which of the chain's kernels went wrong first
    if diagnose:
        flags |= pkg.capi.FLAG_TRACE_SQUELCH
    small = D <= 16
    t_end = time.time() + seconds
    stats = dict(tag=tag, fmt=fmt, fft_log=fft_log, dongles=D, host_path=host_path, batches=0, hop_transforms=0, events=0, handles=0, lib=os.environ.get("AIRBAND_HIP_LIB", "product"))
    events = []
    gen = pkg.AirbandHip([dict(channels=chans)], wave_rate=wave_rate)  # the generator emits u8
    gen.set_signal_plan(carriers)
    iq = host_iq = None
    per_handle = 48 if small else 6
    while time.time() < t_end:
        hip = pkg.AirbandHip([dev] * D, wave_rate=wave_rate, flags=flags, fft_log=fft_log)
        stats["handles"] += 1
        assert hip.channelizer_name() == ("dft_mfma_i8" if (main_path and not f32) else "fft_wave64"), hip.channelizer_name()
        g = hip.geometry
        bpc = 4 if f32 else 1
        lead = g.first_batch_bytes - g.batch_bytes
        span = lead + (ring + 1) * g.batch_bytes + g.lookahead_bytes
        stride = (span + 255) // 256 * 256
        if iq is None:
            iq = torch.empty((D, stride), dtype=torch.uint8, device="cuda")
            if f32:
                tmp = torch.empty((1, span // bpc), dtype=torch.uint8, device="cuda")
                gen.generate_iq(tmp.data_ptr(), span // bpc, 0, span // bpc, seed=0x5EED)
                gen.synchronize()
                iq.view(torch.float32)[0, :span // 4] = (tmp[0].to(torch.float32) - 127.5) / 127.5
                del tmp
            else:
                gen.generate_iq(iq.data_ptr(), stride, 0, span, seed=0x5EED)
                gen.synchronize()
            iq[1:] = iq[0:1]
            torch.cuda.synchronize()
            host_iq = iq[0].cpu().numpy() if host_path else None
        res = hip.device_results()
        ws, B = g.wave_stride, hip.B
        wave = torch.as_tensor(DevPtr(res["waveout"], (D, 8, ws), "<i4"), device="cuda")[:, :, :B]  # bit patterns
        for i in range(per_handle):
            off = 0 if i == 0 else g.first_batch_bytes + ((i - 1) % ring) * g.batch_bytes
            if host_path:  # the same bytes through submit / process: pinned ring, DMA into the staging buffers
                n = (g.first_batch_bytes + g.lookahead_bytes) if i == 0 else g.batch_bytes
                lo = off if i == 0 else off + g.lookahead_bytes
                for d in range(D):
                    assert hip.submit(d, host_iq[lo:lo + n]) == n
                assert hip.process()
            else:
                hip.process_device(iq.data_ptr() + off, stride)
            stats["batches"] += 1
            stats["hop_transforms"] += D * (B + (100 if i == 0 else 0))
            if small:
                w, q = hip.read_bins()
                w = w.view(np.uint32).reshape(D, 8, B)
                q = q.view(np.uint32).reshape(D, 8, 2 * B)
                nw, nq = (w != w[0:1]), (q != q[0:1])
                bad = int(nw.any(axis=(1, 2)).sum() + nq.any(axis=(1, 2)).sum())
                if bad:
                    idx = np.argwhere(nq)[:6].tolist() or np.argwhere(nw)[:6].tolist()
                    ev = dict(tag=tag, fmt=fmt, handle=stats["handles"], batch=i, host_path=host_path, what="iq" if nq.any() else "mag", n_mag=int(nw.sum()), n_iq=int(nq.sum()), first=idx)
            else:
                hip.synchronize()
                ne = (wave != wave[0:1])
                bad = int(ne.any(dim=2).any(dim=1).sum().item())
                if bad:
                    idx = ne.nonzero()[:8].cpu().numpy().tolist()
                    chans = sorted(set(ne.any(dim=2).nonzero()[:, 1].cpu().numpy().tolist()))
                    ev = dict(tag=tag, fmt=fmt, handle=stats["handles"], batch=i, what="audio", dongles_differing=bad, channels=chans, first=idx)
                    if diagnose:  # trace byte: squelch state (bits 0-2) | open 8 | audio 16 | tone present 32
                        d, c, j = idx[0]
                        tr = hip.read_trace().reshape(D, 8, B)
                        dt = np.flatnonzero(tr[d, c] != tr[0, c])
                        jt = int(dt[0]) if len(dt) else -1
                        ev["trace_first"] = jt
                        ev["trace_n"] = int(len(dt))
                        if jt >= 0:
                            ev["trace_bits"] = [int(tr[0, c, jt]), int(tr[d, c, jt])]
                            ev["trace_xor_all"] = int(np.bitwise_or.reduce(tr[d, c] ^ tr[0, c]))
                        lo = max(0, j - 2)
                        ev["audio_good"] = wave[0, c, lo:lo + 8].view(torch.float32).cpu().numpy().tolist()
                        ev["audio_bad"] = wave[d, c, lo:lo + 8].view(torch.float32).cpu().numpy().tolist()
                        ev["lanes"] = sorted(set((ne[:, c].any(dim=1).nonzero()[:, 0] % 32).cpu().numpy().tolist()))
            if bad:
                stats["events"] += 1
                events.append(ev)
                print("EVENT", json.dumps(ev), flush=True)
            if time.time() >= t_end:
                break
        hip.close()
    gen.close()
    with open(out_path, "a") as f:
        f.write(json.dumps(dict(stats, event_list=events)) + "\n")
    print(json.dumps(stats), flush=True)


if __name__ == "__main__":
    main()
