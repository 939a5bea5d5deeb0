"""oracle/pyverify.py -- TEST INFRASTRUCTURE ONLY: spot checks of LARGE handles against the CPU oracle.

A handle with tens of thousands of dongles cannot be compared with the oracle dongle by dongle in reasonable time, and
it takes code paths small handles never reach (XCD permutation of whole 128-dongle groups, splits == 1, rings and result
rows beyond 2^31 elements).  Every dongle is independent (the reference gives each device_t its own state and any
partition of them across demodulate() threads is correct, src/rtl_airband.cpp:1052-1076), so the check is a SAMPLE:
fixed indices around the group / block boundaries plus pseudo-random ones, each run through the oracle on exactly the
bytes the GPU saw (copied back from HBM) and compared per batch.

Used by tests/test_gpu_scale.py and by `bench.py --verify K` (after the timed region; never inside it).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

import pyoracle

STAT_EXACT = ("open_count", "flappy_count", "ctcss_count", "no_ctcss_count", "active_counter", "bin", "squelch_state")
STAT_CLOSE = ("noise_level", "signal_level", "squelch_level", "agcavgfast")


def sample_dongles(n_dev: int, k: int, seed: int = 0x5EED) -> List[int]:
    """Indices to check: the edges of the channelizer's 16/128-dongle placement groups and of the 64-slot demod blocks,
    the first and last dongle, then pseudo-random ones up to k in total."""
    fixed = [0, 1, 7, 8, 15, 16, 127, 128, 129, 1023, 1024, n_dev // 2, n_dev - 129, n_dev - 128, n_dev - 2, n_dev - 1]
    out: List[int] = []
    for d in fixed:
        if 0 <= d < n_dev and d not in out:
            out.append(d)
    out = out[:max(1, k)] if k < len(out) else out
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    while len(out) < min(k, n_dev):
        d = int(rng.randint(0, n_dev))
        if d not in out:
            out.append(d)
    return sorted(out)


class SpotCheck:
    """Oracle twins of a few dongles of a big handle.  `devices` is the full device list the handle was prepared with
    (or a callable index -> device dict); `dongles` the indices to follow."""

    def __init__(self, devices, dongles: Sequence[int], *, wave_rate: int, fft_log: int = 9, fm_demod: int = 0, chan_base: Optional[Sequence[int]] = None):
        self.dongles = list(dongles)
        get = devices if callable(devices) else (lambda i: devices[i])
        self.cfg = [get(d) for d in self.dongles]
        # one oracle instance per followed dongle: they can then advance on separate host threads
        self.orc = [pyoracle.Oracle([c], wave_rate=wave_rate, fft_log=fft_log, fm_demod=fm_demod) for c in self.cfg]
        self.n_ch = [len(c["channels"]) for c in self.cfg]
        if chan_base is None:  # uniform channel count is the only case where this can be guessed
            assert len(set(self.n_ch)) == 1, "pass chan_base for ragged handles"
            chan_base = [d * self.n_ch[0] for d in self.dongles]
        self.chan_base = list(chan_base)
        self.last: List[dict] = [None] * len(self.dongles)
        self.batches = 0
        self.pool = ThreadPoolExecutor(max_workers=max(1, min(len(self.dongles), os.cpu_count() or 1, 16)))

    def close(self):
        self.pool.shutdown(wait=True)
        for o in self.orc:
            o.close()

    def feed(self, spans: Sequence[np.ndarray], *, trace: bool = True):
        """Advance every followed dongle by ONE batch; spans[i] = the bytes dongle i's process_device span started at."""
        def one(i):
            return self.orc[i].run_span(0, spans[i], trace=trace)
        self.last = list(self.pool.map(one, range(len(self.dongles))))
        self.batches += 1

    def compare(self, hip, *, trace: bool = True, rms_tol: float = 1e-4, what: str = "") -> Dict[str, float]:
        """Last batch of the handle vs last batch of the oracle twins: squelch decisions exact (per-sample trace when the
        handle records it, otherwise the open/closed pattern of the audio), axcindicate and the cumulative counters exact,
        audio within rms_tol.  Raises AssertionError; returns the worst figures seen."""
        worst = dict(audio_rms=0.0, level_rel=0.0)
        for i, d in enumerate(self.dongles):
            want = self.last[i]
            nch, cb = self.n_ch[i], self.chan_base[i]
            got = hip.collect(first_channel=cb, n_channels=nch, stats=True)
            tag = "%s dongle %d batch %d" % (what, d, self.batches - 1)
            assert np.array_equal(got["axc"], want["axc"]), "%s: axcindicate %r vs oracle %r" % (tag, bytes(got["axc"]), bytes(want["axc"]))
            gw, ow = got["waveout"], want["waveout"]
            if trace:
                tr = hip.read_trace(cb, nch)
                bad = int((tr != want["trace"]).sum())
                assert bad == 0, "%s: %d per-sample squelch-state mismatches" % (tag, bad)
            else:
                flip = ((gw != 0) != (ow != 0)) & ((np.abs(gw) > 1e-5) | (np.abs(ow) > 1e-5))
                assert not flip.any(), "%s: %d samples open on one side and closed on the other" % (tag, int(flip.sum()))
            err = float(np.sqrt(np.mean((gw.astype(np.float64) - ow) ** 2)))
            assert err <= rms_tol, "%s: audio RMS error %g > %g" % (tag, err, rms_tol)
            worst["audio_rms"] = max(worst["audio_rms"], err)
            for j in range(nch):
                o, g = self.orc[i].stats(0, j), got["stats"][j]
                for f in STAT_EXACT:
                    assert o[f] == g[f], "%s channel %d: %s %r vs oracle %r" % (tag, j, f, g[f], o[f])
                for f in STAT_CLOSE:
                    # relative, with a floor: an NFM channel's agcavgfast is the DC estimate of the discriminator output, i.e. noise around 0
                    rel = abs(o[f] - g[f]) / max(abs(o[f]), 1e-2)
                    assert rel <= 1e-4, "%s channel %d: %s %r vs oracle %r" % (tag, j, f, g[f], o[f])
                    worst["level_rel"] = max(worst["level_rel"], rel)
        return worst


# airband_hip_channel_stats as a numpy record (include/airband_hip.h): 4 floats, 5 uint64, 4 int32 = 72 bytes (ABI 2)
STATS_DT = np.dtype([("noise_level", "<f4"), ("signal_level", "<f4"), ("squelch_level", "<f4"), ("agcavgfast", "<f4"), ("open_count", "<u8"),
                     ("flappy_count", "<u8"), ("ctcss_count", "<u8"), ("no_ctcss_count", "<u8"), ("active_counter", "<u8"), ("bin", "<i4"), ("squelch_state", "<i4"), ("signal_outside_filter", "<i4"), ("reserved", "<i4")])
assert STATS_DT.itemsize == 72


def replica_check(hip, n_dev: int, n_ch: int, *, trace: bool = True, chunk: int = 2048) -> Dict[str, int]:
    """WHOLE-handle check for a handle whose dongles were all given dongle 0's channel plan and dongle 0's bytes: dongles are
    independent (src/rtl_airband.cpp:1052-1076), so every dongle's results of the last batch -- audio rows, axcindicate, every
    statistic and (if recorded) the per-sample squelch trace -- must be BIT-identical to dongle 0's, wherever the dongle sits in
    the handle (XCD placement group, slot block, ring offset beyond 2^31 elements).  Dongle 0 itself is tied to the oracle by a
    SpotCheck.  Returns the number of dongles that differ per output (all zero = pass) and the first offender."""
    import ctypes as C

    L, B = hip.L, hip.B
    bad = dict(waveout=0, axc=0, stats=0, trace=0, first_bad=-1, dongles=n_dev)
    ref = None
    wave = np.empty((chunk * n_ch, B), np.float32)
    axc = np.empty((chunk * n_ch,), np.uint8)
    st = np.empty((chunk * n_ch,), STATS_DT)
    tr = np.empty((chunk * n_ch, B), np.uint8) if trace else None
    for d0 in range(0, n_dev, chunk):
        n = min(chunk, n_dev - d0)
        rc = L.airband_hip_collect_channels(hip.h, d0 * n_ch, n * n_ch, wave.ctypes.data, None, axc.ctypes.data, C.c_void_p(st.ctypes.data))
        assert rc == 0, rc
        if trace:
            rc = L.airband_hip_read_trace_channels(hip.h, d0 * n_ch, n * n_ch, tr.ctypes.data)
            assert rc == 0, rc
        if ref is None:
            ref = dict(wave=wave[:n_ch].view(np.uint32).copy(), axc=axc[:n_ch].copy(), st=st[:n_ch].copy().view(np.uint8).reshape(n_ch, STATS_DT.itemsize),
                       tr=tr[:n_ch].copy() if trace else None)
        diff = {
            "waveout": (wave[:n * n_ch].view(np.uint32).reshape(n, n_ch, B) != ref["wave"]).any(axis=(1, 2)),
            "axc": (axc[:n * n_ch].reshape(n, n_ch) != ref["axc"]).any(axis=1),
            "stats": (st[:n * n_ch].view(np.uint8).reshape(n, n_ch, STATS_DT.itemsize) != ref["st"]).any(axis=(1, 2)),
        }
        if trace:
            diff["trace"] = (tr[:n * n_ch].reshape(n, n_ch, B) != ref["tr"]).any(axis=(1, 2))
        for k, m in diff.items():
            c = int(m.sum())
            bad[k] += c
            if c and bad["first_bad"] < 0:
                bad["first_bad"] = d0 + int(np.argmax(m))
    return bad
