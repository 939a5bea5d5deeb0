"""N > 1 on real GPUs (skipped below two visible devices -- the driver's lease has one; runs the day a bigger node appears): two ranks,
one HIP handle each on its own GPU, BASELINE configs[4] wiring (channel (d, c) -> mixer (d * 8 + c) mod M), per-rank mixer partials
all-reduced with RCCL through the SAME entry bench.py and the reference-side shim use (airband_hip_allreduce_mixers, include/airband_hip.h; the
communicator id travels over torch.distributed) == the sums of ONE handle holding
all dongles, within 1e-4 RMS (float summation order differs); signal flags equal.  At a size the oracle finishes in
seconds the single-handle sums are tied to the oracle's ordered sum as well (src/mixer.cpp:133-140,201-214)."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_MIXERS, N_BATCHES, WAVE_RATE = 64, 4, 16000


def _paths():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run_handle(pkg, torch, gpu, d_start, d_end, comm=None):
    """Handle with the global dongles [d_start, d_end) on GPU `gpu`; returns per batch (left, has_signal), all-reduced if `comm` (a callable that gives the handle its communicator) is given."""
    mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
    chans, carriers = pkg.siggen.baseline_plan(mixed=True)
    n = d_end - d_start
    torch.cuda.set_device(gpu)
    out = []
    with pkg.AirbandHip([dict(channels=chans)] * n, wave_rate=WAVE_RATE, hip_device=gpu) as hip:
        hip.set_mixers(N_MIXERS, mg.baseline_mixer_inputs(d_start, d_end, 8, N_MIXERS))
        hip.set_signal_plan(carriers)
        g = hip.geometry
        span = g.first_batch_bytes + (N_BATCHES - 1) * g.batch_bytes + g.lookahead_bytes
        stride = (span + 255) // 256 * 256
        iq = torch.empty((n, stride), dtype=torch.uint8, device="cuda:%d" % gpu)
        hip.generate_iq(iq.data_ptr(), stride, 0, span, seed=0x5EED, device_index_offset=d_start)
        hip.synchronize()
        if comm is not None:
            comm(hip)
        for b in range(N_BATCHES):
            off = 0 if b == 0 else g.first_batch_bytes + (b - 1) * g.batch_bytes
            hip.process_device(iq.data_ptr() + off, stride)
            if comm is not None:
                hip.allreduce_mixers()  # enqueued behind the batch on the handle's stream; collect_mixers() orders itself behind it
            left, _, sig = hip.collect_mixers()
            out.append((left.copy(), sig.copy()))
        del iq
    return out


def _rank_main(rank, world, port, per_rank, q):
    _paths()
    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("rtlsdr-airband_amd")
    mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    d0, d1 = mg.shard_range(per_rank * world, rank, world)
    res = _run_handle(pkg, torch, rank, d0, d1, comm=lambda hip: mg.init_mixer_exchange(hip, rank, world, dist))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(res)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("per_rank", [6, 2048], ids=["2x6_dongles_vs_oracle", "2x2048_dongles"])
def test_two_gpus_rccl_mixer_sum(pkg, built, per_rank):
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU)")
    import torch.multiprocessing as mp

    _paths()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, per_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = _run_handle(pkg, torch, 0, 0, 2 * per_rank)  # every dongle on one GPU: the summation the RCCL result is measured against
    any_signal = False
    for b in range(N_BATCHES):
        assert np.array_equal(got[b][1], one[b][1]), "batch %d: mixer signal flags" % b
        err = float(np.sqrt(np.mean((got[b][0].astype(np.float64) - one[b][0]) ** 2)))
        assert err <= 1e-4, "batch %d: RCCL sum vs single-handle sum, RMS %g" % (b, err)
        any_signal |= bool(one[b][1].any())
    assert any_signal
    if per_rank <= 8:  # the oracle's ordered sum over every input (src/mixer.cpp:133-140)
        import helpers
        import pyoracle

        mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
        n = 2 * per_rank
        devices, carriers = helpers.plan_devices(n, True)
        nbytes = helpers.stream_bytes(N_BATCHES, WAVE_RATE)
        orc = pyoracle.Oracle(devices, wave_rate=WAVE_RATE)
        outs = [orc.run_device(d, pkg.siggen.generate_u8(d, 0, nbytes // 2, carriers), N_BATCHES) for d in range(n)]
        inputs = mg.baseline_mixer_inputs(0, n, 8, N_MIXERS)
        for b in range(N_BATCHES):
            wave = np.concatenate([o["waveout"][b] for o in outs])
            axc = np.concatenate([o["axc"][b] for o in outs])
            left, right, sig = mg.mix_on_host(inputs, [8 * i for i in range(n)], wave, axc, N_MIXERS)
            assert np.array_equal(sig, got[b][1])
            assert float(np.sqrt(np.mean((left.astype(np.float64) - got[b][0]) ** 2))) <= 1e-4
