"""Experiment driver for the AB_DEMOD_TIMING build: shader-clock cycles per phase of the lane-per-channel demod wavefronts."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("rtlsdr-airband_amd")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
chans, carriers = pkg.siggen.baseline_plan(mixed=True)
hip = pkg.AirbandHip([dict(channels=chans) for _ in range(D)], wave_rate=16000, flags=8 if serial else 0)
g = hip.geometry
span = g.first_batch_bytes + 2 * g.batch_bytes + g.lookahead_bytes
stride = (span + 255) // 256 * 256
iq = torch.empty((D, stride), dtype=torch.uint8, device="cuda")
hip.set_signal_plan(carriers); hip.generate_iq(iq.data_ptr(), stride, 0, span); hip.synchronize()
hip.process_device(iq.data_ptr(), stride); hip.process_device(iq.data_ptr() + g.first_batch_bytes, stride); hip.synchronize()
buf = (C.c_ulonglong * 40)()
hip.L.airband_hip_debug_demod_cycles.argtypes = [C.c_void_p, C.c_int]
hip.L.airband_hip_debug_demod_cycles(buf, 1)
hip.process_device(iq.data_ptr() + g.first_batch_bytes + g.batch_bytes, stride); hip.synchronize()
hip.L.airband_hip_debug_demod_cycles(buf, 0)
names = ["prologue", "fetch issue", "touch (wait loads)", "samples", "flush", "epilogue"]
for k, kind in enumerate(["AM", "NFM", "NFM+LP", "CTCSS front", "generic"]):
    n = buf[k * 8 + 7]
    if not n:
        continue
    tot = sum(buf[k * 8 + i] for i in range(6))
    print("%-12s waves %6d  cycles/wave %9.0f  | " % (kind, n, tot / n) + "  ".join("%s %.1f%%" % (names[i], 100.0 * buf[k * 8 + i] / tot) for i in range(6)))
