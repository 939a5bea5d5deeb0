"""Stress repro for a non-deterministic difference test_results_do_not_depend_on_how_the_bytes_arrive saw once under 12 concurrent workers (seed 346: CF32, fft 8192,
3.2 MS/s, wavefront FFT).  Runs the seed's two submissions ITERS times, prints every difference between them and which of the two disagrees with the oracle."""
import importlib
import os
import sys

R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (R, R + "/tests", R + "/oracle"):
    sys.path.insert(0, p)
import numpy as np

pkg = importlib.import_module("rtlsdr-airband_amd")
import helpers
import pyoracle
import test_gpu_parity as T

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 346
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tag = sys.argv[3] if len(sys.argv) > 3 else ""
devices, iq, fft_log, wave_rate, _, flags = T.random_stage1_case(pkg, seed + 40000, n_batches=4)
n_dev, n_batches = len(devices), 4
raw = [x.view(np.uint8) for x in iq]
orc = pyoracle.Oracle(devices, wave_rate=wave_rate, fft_log=fft_log)
ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]


def run(chunked, rng):
    got = []
    started = 0
    with pkg.AirbandHip(devices, wave_rate=wave_rate, fft_log=fft_log, flags=flags | pkg.capi.FLAG_TRACE_SQUELCH) as hip:
        name = hip.channelizer_name()
        bb = int(hip.geometry.batch_bytes)
        pos = [0] * n_dev
        stalled = 0
        while started < n_batches and stalled < 100000:
            order = rng.permutation(n_dev) if chunked else range(n_dev)
            moved = 0
            for d in order:
                left = len(raw[d]) - pos[d]
                if left <= 0:
                    continue
                want = min(left, int(rng.integers(1, int(1.3 * bb))) if chunked else left)
                if chunked and rng.random() < 0.2:
                    want = min(left, int(rng.integers(1, 64)))
                n = hip.submit(int(d), raw[d][pos[d]:pos[d] + want])
                pos[d] += n
                moved += n
            while started < n_batches and hip.process():
                started += 1
                moved += 1
                out = hip.collect()
                w, q = hip.read_bins()
                got.append((out["waveout"].copy(), out["axc"].copy(), hip.read_trace(), w, q, list(pos)))
            stalled = 0 if moved else stalled + 1
    return got, name


bad = 0
for it in range(iters):
    rng = np.random.default_rng(77000 + seed)
    (a, name), (b, _) = run(False, rng), run(True, rng)
    for k in range(n_batches):
        refw = np.concatenate([r["raw_wavein"][k] for r in ref])
        refq = np.concatenate([r["raw_iq"][k] for r in ref])
        refo = np.concatenate([r["waveout"][k] for r in ref])
        for i, nm in enumerate(("waveout", "axc", "trace", "|bin|", "binIQ")):
            x, y = a[k][i], b[k][i]
            if x.dtype == np.float32:
                x, y = x.view(np.uint32), y.view(np.uint32)
            if not np.array_equal(x, y):
                bad += 1
                ch = np.nonzero((x != y).reshape(x.shape[0], -1).any(axis=1))[0]
                print("%s it %d %s batch %d %s differs on channels %s first idx %s count %s" % (tag, it, name, k, nm, ch, [int(np.nonzero((x[c] != y[c]).ravel())[0][0]) for c in ch][:8],
                                                                                               [int((x[c] != y[c]).sum()) for c in ch][:8]), flush=True)
                for who, g in (("whole", a[k]), ("chunked", b[k])):
                    print("    %s pos %s: |bin| vs oracle %.2e, I/Q %.2e, waveout rms %.2e, per channel |bin| %s" % (
                        who, g[5], helpers.rel_rms(g[3], refw), helpers.rel_rms(g[4], refq), helpers.rms(g[0] - refo),
                        ["%.1e" % helpers.rel_rms(g[3][c], refw[c]) for c in ch][:8]), flush=True)
print("%s seed %d: %d iterations, %d differing arrays" % (tag, seed, iters, bad), flush=True)
