"""N > 1 path on CPU: two gloo ranks, dongles sharded, per-rank mixer partials all-reduced == the single-process
mixer sum (BASELINE config #5 shape at toy size).  The per-dongle audio comes from the CPU oracle here; on GPUs the
same host logic (rtlsdr-airband_amd/multigpu.py) wraps the HIP handle (bench.py --gpus N)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DONGLES, N_MIXERS, N_BATCHES, WAVE_RATE = 4, 5, 5, 16000


def _oracle_audio(d_start, d_end):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import helpers
    import pyoracle

    pkg = importlib.import_module("rtlsdr-airband_amd")
    devices, carriers = helpers.plan_devices(d_end - d_start, True)
    nbytes = helpers.stream_bytes(N_BATCHES, WAVE_RATE)
    orc = pyoracle.Oracle(devices, wave_rate=WAVE_RATE)
    outs = [orc.run_device(i, pkg.siggen.generate_u8(d_start + i, 0, nbytes // 2, carriers), N_BATCHES) for i in range(d_end - d_start)]
    wave = np.concatenate([o["waveout"] for o in outs], axis=1)  # [batch][channels][B]
    axc = np.concatenate([o["axc"] for o in outs], axis=1)
    return wave, axc


def _mix_oracle(inputs, n_dev, wave, axc):
    import pyoracle
    capi = importlib.import_module("rtlsdr-airband_amd.capi")
    L = pyoracle.lib()
    arr = (capi.MixerInput * len(inputs))(*[capi.MixerInput(*map(lambda v: v, (int(a), int(b), int(c), float(d), float(e)))) for a, b, c, d, e in inputs])
    base = np.arange(n_dev, dtype=np.int32) * 8
    B = wave.shape[1]
    left, right, sig = np.zeros((N_MIXERS, B), np.float32), np.zeros((N_MIXERS, B), np.float32), np.zeros(N_MIXERS, np.uint8)
    w = np.ascontiguousarray(wave)
    a = np.ascontiguousarray(axc)
    L.orc_mix(arr, len(inputs), base.ctypes.data, w.ctypes.data, a.ctypes.data, B, N_MIXERS, left.ctypes.data, right.ctypes.data, sig.ctypes.data)
    return left, right, sig


def _rank_main(rank, world, port, q):
    import torch
    import torch.distributed as dist

    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    d0, d1 = mg.shard_range(N_DONGLES, rank, world)
    wave, axc = _oracle_audio(d0, d1)
    inputs = mg.baseline_mixer_inputs(d0, d1, 8, N_MIXERS)
    res = []
    for b in range(N_BATCHES):
        left, right, sig = _mix_oracle(inputs, d1 - d0, wave[b], axc[b])
        tl, tr, ts = torch.from_numpy(left), torch.from_numpy(right), torch.from_numpy(sig)
        mg.allreduce_mixers(tl, tr, ts)
        res.append((tl.numpy().copy(), ts.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(res)


def test_two_rank_mixer_allreduce_matches_single_process(built):
    import torch.multiprocessing as mp

    mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
    assert mg.shard_range(10, 0, 4) == (0, 2) and mg.shard_range(10, 3, 4) == (7, 10)
    assert sum(b - a for a, b in (mg.shard_range(262144, r, 8) for r in range(8))) == 262144
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    wave, axc = _oracle_audio(0, N_DONGLES)
    inputs = mg.baseline_mixer_inputs(0, N_DONGLES, 8, N_MIXERS)
    any_signal = False
    for b in range(N_BATCHES):
        left, right, sig = _mix_oracle(inputs, N_DONGLES, wave[b], axc[b])
        gl, gs = got[b]
        assert np.array_equal(gs, sig)
        # float summation order differs between one process and two partial sums: tolerance parity
        assert np.sqrt(np.mean((gl - left) ** 2)) <= 1e-4
        hl, hr, hs = mg.mix_on_host(inputs, [8 * i for i in range(N_DONGLES)], wave[b], axc[b], N_MIXERS)
        assert np.array_equal(hs, sig) and np.array_equal(hl.view(np.uint32), left.view(np.uint32))
        any_signal |= bool(sig.any())
    assert any_signal
