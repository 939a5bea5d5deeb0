#!/usr/bin/env python3
"""Round 5, third hunt: the fuzz tests' OWN code path as a loop.  Configurations from tests/test_gpu_parity.py::random_stage1_case that land on the wavefront FFT at fft >= 4096 (the class
with most of the events), each run R times exactly as test_results_do_not_depend_on_how_the_bytes_arrive runs its "large pieces" arm -- host path, everything submitted, process / collect /
read_trace / read_bins per batch, TRACE_SQUELCH -- every repetition compared with the first, bit for bit.  On a mismatch everything needed to say WHAT the wrong values are is written out.
usage: r05_fuzz_repro.py <tag> <seconds> <first seed> <out dir>"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    tag, seconds, seed0, out_dir = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import numpy as np

    pkg = importlib.import_module("rtlsdr-airband_amd")
    import test_gpu_parity as T

    capi = pkg.capi
    t_end = time.time() + seconds
    stats = dict(tag=tag, seed0=seed0, configs=0, runs=0, launches=0, events=0)
    seed = seed0
    n_batches = 4
    if os.environ.get("R05_INPROCESS_AGGRESSOR") == "1":  # a thread of THIS process runs int8 matrix-core handles on their own streams beside the victim
        import threading

        def aggressor():
            s2 = seed0 + 900_000
            while time.time() < t_end:
                s2 += 1
                dv, iq2, fl, wr, _, fg = T.random_stage1_case(pkg, s2 + 40_000, n_batches=4)
                with pkg.AirbandHip(dv, wave_rate=wr, fft_log=fl, flags=fg & ~capi.FLAG_FORCE_FFT) as h2:
                    if h2.channelizer_name() != "dft_mfma_i8":
                        continue
                    r2 = [x.view(np.uint8) for x in iq2]
                    p2 = [0] * len(dv)
                    for b in range(4):
                        for d in range(len(dv)):
                            p2[d] += h2.submit(d, r2[d][p2[d]:])
                        if not h2.process():
                            break
                        h2.collect()
                        stats["aggressor_batches"] = stats.get("aggressor_batches", 0) + 1

        threading.Thread(target=aggressor, daemon=True).start()
    while time.time() < t_end:
        seed += 1
        devices, iq, fft_log, wave_rate, _, flags = T.random_stage1_case(pkg, seed + 40_000, n_batches=n_batches)
        victim = os.environ.get("R05_VICTIM", "fft_wave64")  # which channelizer the victims run: the exchange kernel (default), or "dft_mfma_i8" / "dft_mfma_f32" as the library picks them
        if victim != "fft_wave64":
            flags &= ~capi.FLAG_FORCE_FFT
        elif fft_log < 12 or not (devices[0]["sfmt"] == capi.SFMT_F32 or (flags & capi.FLAG_FORCE_FFT)):
            if fft_log < 12:
                continue
            flags |= capi.FLAG_FORCE_FFT  # u8 / s8 / CS16 at fft >= 4096: forced onto the exchange kernel
        raw = [x.view(np.uint8) for x in iq]
        n_dev = len(devices)

        def run():
            got = []
            with pkg.AirbandHip(devices, wave_rate=wave_rate, fft_log=fft_log, flags=flags | capi.FLAG_TRACE_SQUELCH) as hip:
                if hip.channelizer_name() != victim:
                    return None
                pos = [0] * n_dev
                started = 0
                while started < n_batches:
                    for d in range(n_dev):
                        left = len(raw[d]) - pos[d]
                        if left > 0:
                            pos[d] += hip.submit(d, raw[d][pos[d]:pos[d] + left])
                    progressed = False
                    while started < n_batches and hip.process():
                        started += 1
                        progressed = True
                        out = hip.collect()
                        got.append((out["waveout"].copy(), out["axc"].copy(), hip.read_trace()) + hip.read_bins())
                        stats["launches"] += 1
                    if not progressed:
                        break
            return got

        first = run()
        if first is None or len(first) != n_batches:
            continue
        stats["configs"] += 1
        stats["runs"] += 1
        for rep in range(1, 10):
            cur = run()
            stats["runs"] += 1
            bad = []
            for k in range(n_batches):
                for name, x, y in zip(("waveout", "axc", "trace", "mag", "iq"), first[k], cur[k]):
                    xv, yv = (x.view(np.uint32), y.view(np.uint32)) if x.dtype == np.float32 else (x, y)
                    if not np.array_equal(xv, yv):
                        bad.append((k, name))
            if bad:
                stats["events"] += 1
                k = bad[0][0]
                path = os.path.join(out_dir, "event_%s_%d_%d.npz" % (tag, seed, rep))
                np.savez_compressed(path, mag_first=first[k][3], iq_first=first[k][4], mag_cur=cur[k][3], iq_cur=cur[k][4], wave_first=first[k][0], wave_cur=cur[k][0])
                nm = np.argwhere(first[k][3].view(np.uint32) != cur[k][3].view(np.uint32))
                nq = np.argwhere(first[k][4].view(np.uint32) != cur[k][4].view(np.uint32))
                ev = dict(tag=tag, seed=seed, rep=rep, bad=bad[:10], sfmt=int(devices[0]["sfmt"]), fft=1 << fft_log, sr=int(devices[0]["sample_rate"]), wave_rate=wave_rate, flags=int(flags),
                          channels=[len(d["channels"]) for d in devices], mag_at=nm[:12].tolist(), iq_at=nq[:12].tolist(),
                          mag_vals=[(float(first[k][3][c, j]), float(cur[k][3][c, j])) for c, j in nm[:12]], iq_vals=[(float(first[k][4][c, j]), float(cur[k][4][c, j])) for c, j in nq[:12]], file=path)
                print("EVENT", json.dumps(ev), flush=True)
                # which of the two is wrong?  a third run
                third = run()
                stats["runs"] += 1
                same_first = all(np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b) for a, b in zip(first[k], third[k]))
                print("EVENT-THIRD-RUN equals first:", same_first, flush=True)
            if time.time() >= t_end:
                break
    with open(os.path.join(out_dir, "%s.jsonl" % tag.split(".")[0]), "a") as f:
        f.write(json.dumps(stats) + "\n")
    print(json.dumps(stats), flush=True)


if __name__ == "__main__":
    main()
