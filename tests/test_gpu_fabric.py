"""The shipping mixer exchange with MORE THAN ONE RANK on a one-GPU box.  RCCL proper refuses two ranks on one GPU, and the driver's lease has one, so the
library is pointed (AIRBAND_HIP_RCCL_LIB) at tests/fake_rccl/libfake_rccl.so -- test infrastructure: the eight librccl entry points libairband_hip.so binds, as
a rank-ordered all-reduce through shared memory.  Everything on the library's side of those eight calls is the product: airband_hip_comm_unique_id /
_comm_init_rank (bench.py's form: one process per rank, the id carried by torch.distributed), airband_hip_comm_init_all / _comm_group_begin / _allreduce_mixers /
_comm_group_end (the reference-side shim's form: one thread, one handle per rank), airband_hip_clear_mixers.  Each case runs in a child process: librccl is bound
once per process, and other tests of the suite bind the real one.  (tests/test_gpu_multi.py is the same exchange over the real librccl; it needs two GPUs.)"""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
N_MIXERS, N_BATCHES, WAVE_RATE, PER_RANK = 64, 4, 16000, 6

need_double = pytest.mark.skipif(not os.path.exists(FAKE_RCCL), reason="tests/fake_rccl/libfake_rccl.so not built")


def _paths():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _stream(pkg, torch, hip, n, d_start):
    g = hip.geometry
    span = g.first_batch_bytes + (N_BATCHES - 1) * g.batch_bytes + g.lookahead_bytes
    stride = (span + 255) // 256 * 256
    iq = torch.empty((n, stride), dtype=torch.uint8, device="cuda:0")
    hip.generate_iq(iq.data_ptr(), stride, 0, span, seed=0x5EED, device_index_offset=d_start)
    hip.synchronize()
    return iq, stride


def _offset(g, b):
    return 0 if b == 0 else g.first_batch_bytes + (b - 1) * g.batch_bytes


def _handle(pkg, d_start, d_end):
    mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
    chans, carriers = pkg.siggen.baseline_plan(mixed=True)
    hip = pkg.AirbandHip([dict(channels=chans)] * (d_end - d_start), wave_rate=WAVE_RATE, hip_device=0)
    hip.set_mixers(N_MIXERS, mg.baseline_mixer_inputs(d_start, d_end, 8, N_MIXERS))  # BASELINE configs[4] wiring
    hip.set_signal_plan(carriers)
    return hip


def _rank_main(rank, world, port, log, q):
    """bench.py --gpus N's form: one process per rank (both on GPU 0 here), the communicator id from rank 0 over torch.distributed (gloo)."""
    try:
        os.environ["AIRBAND_HIP_RCCL_LIB"] = FAKE_RCCL
        os.environ["FAKE_RCCL_LOG"] = log
        _paths()
        import torch
        import torch.distributed as dist

        pkg = importlib.import_module("rtlsdr-airband_amd")
        mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        d0, d1 = mg.shard_range(PER_RANK * world, rank, world)
        hip = _handle(pkg, d0, d1)
        iq, stride = _stream(pkg, torch, hip, d1 - d0, d0)
        mg.init_mixer_exchange(hip, rank, world, dist)
        out = []
        for b in range(N_BATCHES):
            hip.process_device(iq.data_ptr() + _offset(hip.geometry, b), stride)
            part = hip.collect_mixers()  # this rank's partial sums
            hip.allreduce_mixers()      # in place, on the handle's stream behind the batch
            tot = hip.collect_mixers()
            out.append(([x.copy() for x in part], [x.copy() for x in tot]))
        dist.barrier()
        hip.close()
        dist.destroy_process_group()
        q.put((rank, "ok", out))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, "error", traceback.format_exc()))


def _one_handle(pkg, torch, n):
    hip = _handle(pkg, 0, n)
    iq, stride = _stream(pkg, torch, hip, n, 0)
    out = []
    for b in range(N_BATCHES):
        hip.process_device(iq.data_ptr() + _offset(hip.geometry, b), stride)
        out.append([x.copy() for x in hip.collect_mixers()])
    hip.close()
    return out


@need_double
@pytest.mark.timeout(600)
def test_two_ranks_two_processes_one_gpu(pkg, built, tmp_path):
    """Two processes, a handle with six dongles each, 64 mixers: after airband_hip_allreduce_mixers every rank holds rank 0's partial + rank 1's (bit for bit:
    the stand-in adds in rank order), flags the maximum; and that is the one-handle sum of all twelve dongles within 1e-4 RMS."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    log = str(tmp_path / "fake_rccl.log")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, log, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, status, res = q.get(timeout=500)
        assert status == "ok", res
        got[rank] = res
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = _one_handle(pkg, torch, 2 * PER_RANK)
    any_signal = False
    for b in range(N_BATCHES):
        (l0, r0, s0), t0 = got[0][b]
        (l1, r1, s1), t1 = got[1][b]
        for a, c in zip(t0, t1):
            assert np.array_equal(a, c), "batch %d: the ranks disagree about the node's sums" % b
        assert np.array_equal(t0[0], l0 + l1) and np.array_equal(t0[1], r0 + r1) and np.array_equal(t0[2], np.maximum(s0, s1)), b
        assert np.array_equal(t0[2], one[b][2]), "batch %d: mixer signal flags" % b
        err = float(np.sqrt(np.mean((t0[0].astype(np.float64) - one[b][0]) ** 2)))
        assert err <= 1e-4, "batch %d: exchanged sum vs single-handle sum, RMS %g" % (b, err)
        any_signal |= bool(one[b][2].any()) and bool(np.abs(l0).max() > 0) and bool(np.abs(l1).max() > 0)
    assert any_signal, "both ranks must contribute"
    lines = [l for l in open(log).read().splitlines() if l.startswith("allreduce ")]
    assert len(lines) == 2 * 3 * N_BATCHES and all("nranks=2" in l for l in lines)
    assert len({l.split("pid=")[1] for l in lines}) == 2


def _clique_main(log, q):
    """The shim's form: one thread, comm_init_all over the handles, per batch ONE group with every handle's allreduce_mixers; from the third batch on the
    second handle runs no batch (every dongle of it switched off) and is cleared instead."""
    try:
        os.environ["AIRBAND_HIP_RCCL_LIB"] = FAKE_RCCL
        os.environ["FAKE_RCCL_LOG"] = log
        _paths()
        import torch

        pkg = importlib.import_module("rtlsdr-airband_amd")
        hs = [_handle(pkg, r * PER_RANK, (r + 1) * PER_RANK) for r in range(3)]
        streams = [_stream(pkg, torch, h, PER_RANK, r * PER_RANK) for r, h in enumerate(hs)]
        pkg.AirbandHip.comm_init_all(hs)
        out = []
        for b in range(N_BATCHES):
            parts = []
            for r, h in enumerate(hs):
                if r == 1 and b >= 2:
                    if b == 2:
                        for d in range(PER_RANK):
                            h.device_enable(d, False)
                    h.clear_mixers()
                    parts.append(None)
                    continue
                iq, stride = streams[r]
                h.process_device(iq.data_ptr() + _offset(h.geometry, b), stride)
                parts.append([x.copy() for x in h.collect_mixers()])
            pkg.AirbandHip.allreduce_mixers_group(hs)
            out.append((parts, [[x.copy() for x in h.collect_mixers()] for h in hs]))
        for h in hs:
            h.close()
        q.put(("ok", out))
    except Exception:  # noqa: BLE001
        import traceback

        q.put(("error", traceback.format_exc()))


@need_double
@pytest.mark.timeout(600)
def test_three_ranks_one_thread_and_a_rank_without_a_batch(pkg, built, tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    log = str(tmp_path / "fake_rccl.log")
    p = ctx.Process(target=_clique_main, args=(log, q))
    p.start()
    status, out = q.get(timeout=500)
    p.join(timeout=120)
    assert status == "ok", out
    assert p.exitcode == 0
    seen = False
    for b, (parts, totals) in enumerate(out):
        for t in totals[1:]:
            for a, c in zip(totals[0], t):
                assert np.array_equal(a, c), b
        left = np.zeros_like(totals[0][0])
        right = np.zeros_like(left)
        sig = np.zeros_like(totals[0][2])
        for pr in parts:  # rank order; a rank without a batch adds nothing -- not its last batch's sums, not the last exchange's totals
            if pr is None:
                continue
            left, right, sig = left + pr[0], right + pr[1], np.maximum(sig, pr[2])
        assert np.array_equal(totals[0][0], left) and np.array_equal(totals[0][1], right) and np.array_equal(totals[0][2], sig), b
        seen |= bool(sig.any())
    assert seen
    lines = [l for l in open(log).read().splitlines() if l.startswith("allreduce ")]
    assert len(lines) == 3 * 3 * N_BATCHES and all("nranks=3" in l for l in lines)


@need_double
@pytest.mark.timeout(600)
def test_bench_with_two_ranks_on_one_gpu(tmp_path):
    """The REAL `bench.py --gpus 2 --mixers 64` body, end to end, before the first multi-GPU node sees it (round-5 review, item 7): its own launcher
    (torch.distributed.run, two ranks), the process group, `device_index_offset` = rank x dongles, the per-step exchange through airband_hip_allreduce_mixers,
    the barrier + MAX-reduced elapsed time, ONE JSON line from rank 0.  Both ranks sit on GPU 0 (AIRBAND_BENCH_LOCAL_DEVICE), torch.distributed runs over gloo and the
    library's exchange through the stand-in (RCCL proper refuses two ranks on one GPU) -- everything else is what the driver will run with --gpus 8."""
    import json
    import subprocess

    env = dict(os.environ)
    env.update(AIRBAND_HIP_RCCL_LIB=FAKE_RCCL, FAKE_RCCL_LOG=str(tmp_path / "fake_rccl.log"), AIRBAND_BENCH_LOCAL_DEVICE="0", AIRBAND_BENCH_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("AIRBAND_BENCH_SPAWNED", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--mixers", "64", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
           "--verify", "2"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=540, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert "x2" in out["config"]["parallelism"] and "RCCL" in out["config"]["parallelism"] and out["config"]["mixers"] == 64
    # whole-job aggregate: both ranks' dongles (64 each) over the max-over-ranks time
    per_step = 2 * 64 * 320000
    assert out["value"] is not None and abs(out["value"] - per_step / (out["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * out["value"]
    assert out["verified_dongles"] == 2, out.get("verify")
    log = open(str(tmp_path / "fake_rccl.log")).read()
    assert "nranks=2" in log  # the exchange did run between two ranks
