import sys, time, importlib
sys.path[:0] = ['.', 'oracle', 'tests']
import numpy as np, torch
pkg = importlib.import_module('rtlsdr-airband_amd')
import pyverify
def free(tag):
    torch.cuda.synchronize(); print(tag, round(torch.cuda.mem_get_info()[0] / 2**30, 1), flush=True)
free('start')
chans, car = pkg.siggen.baseline_plan(mixed=True)
n_dev = 65536
devs = [dict(channels=[dict(c) for c in chans]) for _ in range(n_dev)]
mode = sys.argv[1] if len(sys.argv) > 1 else 'full'
hip = pkg.AirbandHip(devs, wave_rate=16000, flags=1)
free('handle')
g = hip.geometry
stride = 2624000
iq = torch.empty((n_dev, stride), dtype=torch.uint8, device='cuda')
hip.set_signal_plan(car); hip.generate_iq(iq.data_ptr(), stride, 0, 2600000); hip.synchronize()
free('iq')
dongles = pyverify.sample_dongles(n_dev, 8)
if mode in ('full', 'host'):
    host = {d: iq[d].cpu().numpy() for d in dongles}
    free('host copies')
if mode == 'full':
    spot = pyverify.SpotCheck(lambda d: devs[d], dongles, wave_rate=16000)
    for i in range(3):
        hip.process_device(iq.data_ptr() + (0 if i == 0 else g.first_batch_bytes), stride)
        spot.feed([host[d][(0 if i == 0 else g.first_batch_bytes):] for d in dongles])
        spot.compare(hip, trace=True)
    spot.close()
    free('after compare')
hip.close(); free('after close')
del iq; torch.cuda.synchronize(); torch.cuda.empty_cache(); free('after del iq')
time.sleep(2); free('2s later')
