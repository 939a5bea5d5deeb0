#!/usr/bin/env python3
"""Aggressor for scripts/r05_fuzz_repro.py: random configurations of the fuzz generator that land on the MATRIX-CORE channelizers (int8: LDS DMA staging; f32: register staging), small
handles, four batches each through the host path, for <seconds>.  usage: r05_aggressor.py <seconds> <first seed> [i8|f32|any]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    seconds, seed, want = float(sys.argv[1]), int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "any")
    import numpy as np

    pkg = importlib.import_module("rtlsdr-airband_amd")
    import test_gpu_parity as T

    t_end = time.time() + seconds
    n = 0
    while time.time() < t_end:
        seed += 1
        devices, iq, fft_log, wave_rate, _, flags = T.random_stage1_case(pkg, seed + 40_000, n_batches=4)
        flags &= ~pkg.capi.FLAG_FORCE_FFT
        with pkg.AirbandHip(devices, wave_rate=wave_rate, fft_log=fft_log, flags=flags) as hip:
            name = hip.channelizer_name()
            if name == "fft_wave64" or (want == "i8" and name != "dft_mfma_i8") or (want == "f32" and name != "dft_mfma_f32"):
                continue
            raw = [x.view(np.uint8) for x in iq]
            pos = [0] * len(devices)
            for b in range(4):
                for d in range(len(devices)):
                    pos[d] += hip.submit(d, raw[d][pos[d]:])
                if not hip.process():
                    break
                hip.collect()
                n += 1
    print("aggressor batches", n, flush=True)


if __name__ == "__main__":
    main()
