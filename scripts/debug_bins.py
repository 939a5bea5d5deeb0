import sys, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import helpers, pyoracle
pkg = importlib.import_module("rtlsdr-airband_amd")
mixed, wave_rate = False, 8000
n_dev, n_batches = 4, 6
devices, carriers = helpers.plan_devices(n_dev, mixed)
nbytes = helpers.stream_bytes(n_batches, wave_rate)
iq = [pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers) for d in range(n_dev)]
orc = pyoracle.Oracle(devices, wave_rate=wave_rate)
ref = [orc.run_device(d, iq[d], n_batches) for d in range(n_dev)]
for trial in range(2):
    with pkg.AirbandHip(devices, wave_rate=wave_rate, flags=pkg.capi.FLAG_TRACE_SQUELCH) as hip:
        for d in range(n_dev):
            hip.submit(d, iq[d])
        for b in range(n_batches):
            assert hip.process()
            out = hip.collect()
            w, q = hip.read_bins()
            want = np.concatenate([r["raw_wavein"][b] for r in ref])
            err = np.abs(w - want) / (np.abs(want) + 1e-3)
            bad = np.argwhere(err > 1e-4)
            print("trial", trial, "batch", b, "bad count", len(bad), "rows(time) of bad:", sorted(set(bad[:, 1].tolist()))[:20], "chans:", sorted(set(bad[:, 0].tolist()))[:40])
