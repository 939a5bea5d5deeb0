#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref, compiled in place from /root/reference).
Run in the build container only (it needs /root/reference); the fixtures travel, the reference does not.

Each fixture: one dongle of the BASELINE channel plan (with a few per-channel options switched on), a seeded
synthetic u8 I/Q stream (regenerated from the recorded seed by rtlsdr-airband_amd/siggen.py, its SHA-256 is stored)
and the reference's outputs: waveout / iq_out / axcindicate per batch plus the squelch statistics.
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import helpers  # noqa: E402
import pyref  # noqa: E402

sg = importlib.import_module("rtlsdr-airband_amd.siggen")

CASES = {
    "am8": dict(mixed=False, wave_rate=8000, fm_demod=0, dongle=0, n_batches=10, tweak=False),
    "mixed_nfm": dict(mixed=True, wave_rate=16000, fm_demod=0, dongle=3, n_batches=10, tweak=True),
    "mixed_quadri": dict(mixed=True, wave_rate=16000, fm_demod=1, dongle=6, n_batches=8, tweak=False),
    # what the matrix-core channelizer claims beyond u8 / fft 512 / 2.56 MS/s (helpers.format_case): CS16 as SoapySDR delivers it, a two-piece
    # window, a hop that is not a multiple of 16 bytes, s8 as mirisdr delivers it
    "cs16_fft512": dict(mixed=True, wave_rate=16000, fm_demod=0, dongle=2, n_batches=6, tweak=False, format=("SFMT_S16", 9, 2_560_000)),
    "u8_fft1024": dict(mixed=False, wave_rate=8000, fm_demod=0, dongle=4, n_batches=6, tweak=False, format=("SFMT_U8", 10, 2_560_000)),
    "u8_2400k": dict(mixed=True, wave_rate=16000, fm_demod=0, dongle=1, n_batches=6, tweak=False, format=("SFMT_U8", 9, 2_400_000)),
    "s8_fft512": dict(mixed=False, wave_rate=8000, fm_demod=0, dongle=7, n_batches=6, tweak=False, format=("SFMT_S8", 9, 2_560_000)),
}


def tweak(d, ch):
    ch[3]["has_iq_outputs"] = 1
    ch[0]["bandwidth_hz"] = 8000
    ch[2]["squelch_threshold_dbfs"] = -40
    ch[4]["squelch_snr_threshold_db"] = 6.0
    ch[6]["ampfactor"] = 2.5


def build_case(name):
    c = CASES[name]
    if "format" in c:
        pkg = importlib.import_module("rtlsdr-airband_amd")
        sfmt_name, fft_log, sample_rate = c["format"]
        devices, iq = helpers.format_case(pkg, getattr(pkg.capi, sfmt_name), fft_log, sample_rate, c["wave_rate"], 1, c["n_batches"], first_dongle=c["dongle"])
        return c, devices, None, iq[0]
    devices, carriers = helpers.plan_devices(1, c["mixed"], tweak if c["tweak"] else None)
    iq = sg.generate_u8(c["dongle"], 0, helpers.stream_bytes(c["n_batches"], c["wave_rate"]) // 2, carriers)
    return c, devices, carriers, iq


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        c, devices, carriers, iq = build_case(name)
        ref = pyref.run_reference(devices, [iq], c["n_batches"], nfm=c["wave_rate"] == 16000, fm_demod=c["fm_demod"], fft_log=c["format"][1] if "format" in c else 9)[0]
        assert ref["n_batches"] == c["n_batches"]
        keep_iq = [j for j, ch in enumerate(devices[0]["channels"]) if ch["has_iq_outputs"]]
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), waveout=ref["waveout"].astype(np.float32), axc=ref["axc"],
                            iq_out=ref["iq_out"][:, keep_iq].astype(np.float32), iq_channels=np.array(keep_iq, np.int32),
                            iq_sha256=np.frombuffer(hashlib.sha256(iq.tobytes()).digest(), np.uint8),
                            stats=json.dumps(ref["stats"]), case=json.dumps(c), channels=json.dumps(devices[0]["channels"]))
        print(name, os.path.getsize(os.path.join(out_dir, name + ".npz")) // 1024, "KiB; open batches per channel:", (ref["axc"] == ord("*")).sum(axis=0))


if __name__ == "__main__":
    main()
