#!/usr/bin/env python3
"""Round 5, second hunt for the exchange kernel's rare event.  scripts/r05_exchange_stress.py compared the dongles of a handle with each other -- and is therefore BLIND to anything
the dongles of a handle SHARE (the twiddle and window tables, the code): a fault there makes every replica wrong in the same way.  The fuzz campaigns that do see the event compare
against the oracle or against a second run.  This worker does what they do, thousands of times faster: one handle, one batch of input, the SAME batch through stage 1 again and again;
the stage-1 bins (|bin| and raw I/Q of every hop) of every repetition must equal the first one's bit for bit.  Handles of random configurations (the fuzz's variety: format, fft size,
sample rate, channel count -- tables of different contents at recycled addresses) are created and destroyed all the time.  On a mismatch the batch is run three more times to see
whether the fault persists.
usage: r05_repeat_stress.py <tag> <seconds> <seed> <out.jsonl> [same]     (same: one configuration for the whole run -- no table churn)"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag, seconds, seed, out_path = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    same = len(sys.argv) > 5 and sys.argv[5] == "same"
    import numpy as np
    import torch

    pkg = importlib.import_module("rtlsdr-airband_amd")
    capi, sg = pkg.capi, pkg.siggen
    rng = np.random.default_rng(seed)
    chans, carriers = sg.baseline_plan(mixed=True)
    wave_rate = 16000
    t_end = time.time() + seconds
    stats = dict(tag=tag, seed=seed, same=same, handles=0, launches=0, hop_transforms=0, events=0, lib=os.environ.get("AIRBAND_HIP_LIB", "product"))
    events = []
    gen = pkg.AirbandHip([dict(channels=chans)], wave_rate=wave_rate)
    gen.set_signal_plan(carriers)
    fixed = None
    while time.time() < t_end:
        if fixed is None or not same:
            fmt = ["u8", "f32", "f32"][int(rng.integers(0, 3))]
            fft_log = int(rng.choice([9, 12, 13] if fmt == "u8" else [12, 13]))
            sr = int(rng.choice([2_560_000, 2_000_000, 1_920_000, 2_400_000]))
            D = int(rng.integers(1, 4))
            nch = [int(rng.integers(1, 9)) for _ in range(D)]
            fixed = (fmt, fft_log, sr, D, nch)
        fmt, fft_log, sr, D, nch = fixed
        f32 = fmt == "f32"
        devs = []
        for d in range(D):
            ch = [dict(c) for c in chans[:nch[d]]]
            devs.append(dict(channels=ch, sample_rate=sr, sfmt=capi.SFMT_F32) if f32 else dict(channels=ch, sample_rate=sr))
        flags = 0 if f32 else capi.FLAG_FORCE_FFT
        try:
            hip = pkg.AirbandHip(devs, wave_rate=wave_rate, flags=flags, fft_log=fft_log)
        except pkg.AirbandError:
            fixed = None
            continue
        stats["handles"] += 1
        if hip.channelizer_name() != "fft_wave64":
            hip.close()
            fixed = None
            continue
        g = hip.geometry
        bpc = 4 if f32 else 1
        span = g.first_batch_bytes + g.lookahead_bytes
        stride = (span + 255) // 256 * 256
        iq = torch.zeros((D, stride), dtype=torch.uint8, device="cuda")
        tmp = torch.empty((D, span // bpc), dtype=torch.uint8, device="cuda")
        for d in range(D):
            gen.generate_iq(tmp[d].data_ptr(), span // bpc, 0, span // bpc, seed=0x5EED, device_index_offset=int(rng.integers(0, 64)))
        gen.synchronize()
        if f32:
            iq.view(torch.float32)[:, :span // 4] = (tmp.to(torch.float32) - 127.5) / 127.5
        else:
            iq[:, :span] = tmp
        del tmp
        torch.cuda.synchronize()
        B = hip.B
        hip.close()

        def run_once():
            """a NEW handle (tables of the same contents at recycled addresses), the batch through stage 1 (+ stage 2), the stage-1 bins of its last WAVE_BATCH hops"""
            h = pkg.AirbandHip(devs, wave_rate=wave_rate, flags=flags, fft_log=fft_log)
            stats["handles"] += 1
            h.process_device(iq.data_ptr(), stride)
            w, q = h.read_bins()
            h.close()
            stats["launches"] += 1
            stats["hop_transforms"] += D * (B + 100)
            return w.view(np.uint32).copy(), q.view(np.uint32).copy()

        first = run_once()
        for rep in range(1, 12):
            cur = run_once()
            nw, nq = cur[0] != first[0], cur[1] != first[1]
            if nw.any() or nq.any():
                stats["events"] += 1
                again = []
                for k in range(3):
                    c2 = run_once()
                    again.append(dict(equals_first=bool((c2[0] == first[0]).all() and (c2[1] == first[1]).all()), equals_bad=bool((c2[0] == cur[0]).all() and (c2[1] == cur[1]).all())))
                ev = dict(tag=tag, cfg=dict(fmt=fmt, fft_log=fft_log, sr=sr, D=D, nch=nch), rep=rep, n_mag=int(nw.sum()), n_iq=int(nq.sum()),
                          mag_at=np.argwhere(nw)[:8].tolist(), iq_at=np.argwhere(nq)[:8].tolist(), again=again)
                events.append(ev)
                print("EVENT", json.dumps(ev), flush=True)
            if time.time() >= t_end:
                break
        del iq
    gen.close()
    with open(out_path, "a") as f:
        f.write(json.dumps(dict(stats, event_list=events)) + "\n")
    print(json.dumps(stats), flush=True)


if __name__ == "__main__":
    main()
