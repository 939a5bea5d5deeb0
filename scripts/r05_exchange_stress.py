#!/usr/bin/env python3
"""Round 5: the decisive experiment for round 4's rare event (profiles/r04_experiments.md I -- one transform in ~10^9 wrong on the wavefront FFT's LDS exchange
when a dozen PROCESSES share the GPU, none in 4.6e9 with one process).  One worker of a P-process load: a handle whose dongles ALL replay dongle 0's bytes, on the
exchange kernel (u8 at fft 512 with FORCE_FFT, or CF32 at fft 4096: eight decimated transforms per hop), carriers keyed permanently so that every channel's audio
depends on every hop's bin; after every batch every dongle's result rows are compared with dongle 0's ON THE GPU (bit for bit: dongles are independent and
identical).  The handle is torn down and rebuilt every few batches (the fuzz campaign that saw the events created and destroyed handles all the time).
usage: r05_exchange_stress.py <tag> <seconds> <u8|f32> <dongles> <out.jsonl>      (the library under test: AIRBAND_HIP_LIB)"""
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
    import numpy as np
    import torch

    pkg = importlib.import_module("rtlsdr-airband_amd")
    sg = pkg.siggen
    chans, carriers = sg.baseline_plan(mixed=True)
    # every transmitter keyed all the time: a wrong bin anywhere shows in the audio
    import dataclasses

    carriers = [dataclasses.replace(c, key_period=0) for c in carriers]
    wave_rate, ring = 16000, 2
    f32 = fmt == "f32"
    fft_log = 12 if f32 else 9
    dev = dict(channels=chans, sfmt=pkg.capi.SFMT_F32) if f32 else dict(channels=chans)
    flags = 0 if f32 else pkg.capi.FLAG_FORCE_FFT
    t_end = time.time() + seconds
    stats = dict(tag=tag, fmt=fmt, dongles=D, batches=0, hop_transforms=0, events=0, handles=0, lib=os.environ.get("AIRBAND_HIP_LIB", "product"))
    events = []
    gen = pkg.AirbandHip([dict(channels=chans)], wave_rate=wave_rate)  # the generator emits u8
    gen.set_signal_plan(carriers)
    iq = None
    while time.time() < t_end:
        hip = pkg.AirbandHip([dev] * D, wave_rate=wave_rate, flags=flags, fft_log=fft_log)
        stats["handles"] += 1
        assert hip.channelizer_name() == "fft_wave64", hip.channelizer_name()
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
        res = hip.device_results()
        ws, B = g.wave_stride, hip.B
        wave = torch.as_tensor(DevPtr(res["waveout"], (D, 8, ws), "<i4"), device="cuda")[:, :, :B]  # bit patterns
        n_hops = B
        for i in range(2 * ring + 2):
            off = 0 if i == 0 else g.first_batch_bytes + ((i - 1) % ring) * g.batch_bytes
            hip.process_device(iq.data_ptr() + off, stride)
            hip.synchronize()
            ne = (wave != wave[0:1])
            bad = int(ne.any(dim=2).any(dim=1).sum().item())
            stats["batches"] += 1
            stats["hop_transforms"] += D * (n_hops + (100 if i == 0 else 0))
            if i > 0 and not bool((wave[0] != 0).any().item()):
                raise RuntimeError("no audio: the comparison would see nothing")
            if bad:
                stats["events"] += 1
                idx = ne.nonzero()[:8].cpu().numpy().tolist()
                ev = dict(tag=tag, fmt=fmt, handle=stats["handles"], batch=i, dongles_differing=bad, first=idx,
                          values=[(int(wave[d, c, s].item()), int(wave[0, c, s].item())) for d, c, s in idx[:4]])
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
