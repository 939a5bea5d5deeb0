"""N > 1 on CPU: two gloo ranks, dongles sharded (multigpu.shard_range), BASELINE configs[4] wiring at toy size, per-rank mixer partials all-reduced == the
single-process mixer sum.  What runs here of the code that ships: the partition, the wiring, and multigpu.init_mixer_exchange() -- how rank 0's communicator id
reaches the other ranks (bench.py --gpus N and tests/test_gpu_multi.py call the same function).  What cannot run here is the library's own exchange entry
(airband_hip_allreduce_mixers needs a handle, a handle needs a GPU): its place is taken by a ctypes object that makes the SAME eight librccl calls in the same
order (group start, SUM left, SUM right, MAX flags, group end) on host buffers, against tests/fake_rccl/ in host mode -- the stand-in the GPU suite points the
real library at (tests/test_gpu_fabric.py, tests/test_dropin_shim.py), so this file also checks that checker.  The per-dongle audio comes from the CPU oracle."""
import ctypes as C
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DONGLES, N_MIXERS, N_BATCHES, WAVE_RATE = 4, 5, 5, 16000
FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
NCCL_UINT8, NCCL_FLOAT, NCCL_SUM, NCCL_MAX = 1, 7, 0, 2  # ncclDataType_t / ncclRedOp_t (rccl.h)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _paths():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _fake():
    os.environ["FAKE_RCCL_HOST"] = "1"
    L = C.CDLL(FAKE_RCCL)
    L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    L.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    L.ncclGetErrorString.restype = C.c_char_p
    return L


class HostHandle:
    """Stands where an AirbandHip handle stands for init_mixer_exchange() and the per-batch exchange; its "device" buffers are numpy arrays."""

    L = None

    def __init__(self, n_mixers, wave_batch):
        self.left = np.zeros((n_mixers, wave_batch), np.float32)
        self.right = np.zeros((n_mixers, wave_batch), np.float32)
        self.sig = np.zeros((n_mixers,), np.uint8)
        self.comm = C.c_void_p()

    @classmethod
    def comm_unique_id(cls) -> bytes:
        u = UniqueId()
        assert cls.L.ncclGetUniqueId(C.byref(u)) == 0
        return bytes(bytearray(u)[:128])

    def comm_init_rank(self, unique_id: bytes, nranks: int, rank: int):
        u = UniqueId.from_buffer_copy(unique_id)
        rc = self.L.ncclCommInitRank(C.byref(self.comm), nranks, u, rank)
        assert rc == 0, self.L.ncclGetErrorString(rc)

    def allreduce_mixers(self):  # the calls of airband_hip_allreduce_mixers (csrc/airband_hip.cpp), in its order
        L = self.L
        assert L.ncclGroupStart() == 0
        assert L.ncclAllReduce(self.left.ctypes.data, self.left.ctypes.data, self.left.size, NCCL_FLOAT, NCCL_SUM, self.comm, None) == 0
        assert L.ncclAllReduce(self.right.ctypes.data, self.right.ctypes.data, self.right.size, NCCL_FLOAT, NCCL_SUM, self.comm, None) == 0
        assert L.ncclAllReduce(self.sig.ctypes.data, self.sig.ctypes.data, self.sig.size, NCCL_UINT8, NCCL_MAX, self.comm, None) == 0
        rc = L.ncclGroupEnd()
        assert rc == 0, L.ncclGetErrorString(rc)

    def close(self):
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


def _oracle_audio(d_start, d_end):
    _paths()
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
    arr = (capi.MixerInput * len(inputs))(*[capi.MixerInput(int(a), int(b), int(c), float(d), float(e)) for a, b, c, d, e in inputs])
    base = np.arange(n_dev, dtype=np.int32) * 8
    B = wave.shape[1]
    left, right, sig = np.zeros((N_MIXERS, B), np.float32), np.zeros((N_MIXERS, B), np.float32), np.zeros(N_MIXERS, np.uint8)
    w = np.ascontiguousarray(wave)
    a = np.ascontiguousarray(axc)
    L.orc_mix(arr, len(inputs), base.ctypes.data, w.ctypes.data, a.ctypes.data, B, N_MIXERS, left.ctypes.data, right.ctypes.data, sig.ctypes.data)
    return left, right, sig


def _rank_main(rank, world, port, q):
    try:
        import torch.distributed as dist

        _paths()
        mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
        HostHandle.L = _fake()
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        d0, d1 = mg.shard_range(N_DONGLES, rank, world)
        wave, axc = _oracle_audio(d0, d1)
        inputs = mg.baseline_mixer_inputs(d0, d1, 8, N_MIXERS)
        h = HostHandle(N_MIXERS, wave.shape[2])
        uid = mg.init_mixer_exchange(h, rank, world, dist)
        res = []
        for b in range(N_BATCHES):
            left, right, sig = _mix_oracle(inputs, d1 - d0, wave[b], axc[b])
            h.left[:], h.right[:], h.sig[:] = left, right, sig
            h.allreduce_mixers()
            res.append((left, sig, h.left.copy(), h.sig.copy()))
        dist.barrier()
        h.close()
        dist.destroy_process_group()
        q.put((rank, "ok", (uid, res)))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, "error", traceback.format_exc()))


def test_two_rank_mixer_allreduce_matches_single_process(built):
    import torch.multiprocessing as mp

    mg = importlib.import_module("rtlsdr-airband_amd.multigpu")
    assert os.path.exists(FAKE_RCCL), "build() compiles tests/fake_rccl/"
    assert mg.shard_range(10, 0, 4) == (0, 2) and mg.shard_range(10, 3, 4) == (7, 10)
    assert sum(b - a for a, b in (mg.shard_range(262144, r, 8) for r in range(8))) == 262144
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, status, res = q.get(timeout=600)
        assert status == "ok", res
        got[rank] = res
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] and len(got[0][0]) == 128, "both ranks must hold rank 0's communicator id"
    wave, axc = _oracle_audio(0, N_DONGLES)
    inputs = mg.baseline_mixer_inputs(0, N_DONGLES, 8, N_MIXERS)
    any_signal = False
    for b in range(N_BATCHES):
        left, right, sig = _mix_oracle(inputs, N_DONGLES, wave[b], axc[b])
        p0, s0, t0, ts0 = got[0][1][b]
        p1, s1, t1, ts1 = got[1][1][b]
        assert np.array_equal(t0, t1) and np.array_equal(ts0, ts1), "the ranks disagree"
        assert np.array_equal(t0, p0 + p1) and np.array_equal(ts0, np.maximum(s0, s1)), "rank-ordered sum of the partials"
        assert np.array_equal(ts0, sig)
        # float summation order differs between one process and two partial sums: tolerance parity
        assert np.sqrt(np.mean((t0 - left) ** 2)) <= 1e-4
        hl, hr, hs = mg.mix_on_host(inputs, [8 * i for i in range(N_DONGLES)], wave[b], axc[b], N_MIXERS)
        assert np.array_equal(hs, sig) and np.array_equal(hl.view(np.uint32), left.view(np.uint32))
        any_signal |= bool(sig.any()) and bool(np.abs(p0).max() > 0) and bool(np.abs(p1).max() > 0)
    assert any_signal


def test_one_thread_drives_a_clique_in_a_group(built):
    """ncclCommInitAll + one group with every rank's collectives (the reference-side shim's form), on host buffers: three ranks, three collectives each,
    matched by order; SUM in rank order, MAX of the flags; then a second batch on the same communicators."""
    L = _fake()
    n = 3
    comms = (C.c_void_p * n)()
    assert L.ncclCommInitAll(comms, n, (C.c_int * n)(0, 0, 0)) == 0
    rng = np.random.default_rng(5)
    for batch in range(2):
        a = [rng.standard_normal(1000).astype(np.float32) for _ in range(n)]
        b = [rng.standard_normal(7).astype(np.float32) for _ in range(n)]
        f = [rng.integers(0, 2, 9).astype(np.uint8) for _ in range(n)]
        want_a = (a[0] + a[1]) + a[2]
        want_b = (b[0] + b[1]) + b[2]
        want_f = np.maximum(np.maximum(f[0], f[1]), f[2])
        assert L.ncclGroupStart() == 0
        for r in range(n):
            assert L.ncclGroupStart() == 0  # nested, as airband_hip_allreduce_mixers nests its own group inside the shim's
            assert L.ncclAllReduce(a[r].ctypes.data, a[r].ctypes.data, a[r].size, NCCL_FLOAT, NCCL_SUM, comms[r], None) == 0
            assert L.ncclAllReduce(b[r].ctypes.data, b[r].ctypes.data, b[r].size, NCCL_FLOAT, NCCL_SUM, comms[r], None) == 0
            assert L.ncclAllReduce(f[r].ctypes.data, f[r].ctypes.data, f[r].size, NCCL_UINT8, NCCL_MAX, comms[r], None) == 0
            assert L.ncclGroupEnd() == 0
        assert L.ncclGroupEnd() == 0
        for r in range(n):
            assert np.array_equal(a[r], want_a) and np.array_equal(b[r], want_b) and np.array_equal(f[r], want_f), (batch, r)
    for r in range(n):
        L.ncclCommDestroy(comms[r])


def test_init_mixer_exchange_refuses_a_missing_id(built):
    mg = importlib.import_module("rtlsdr-airband_amd.multigpu")

    class Dist:
        @staticmethod
        def broadcast_object_list(box, src=0):
            box[0] = None  # the transport lost it

    class H:
        @staticmethod
        def comm_unique_id():
            return b"\0" * 128

        def comm_init_rank(self, *a):
            raise AssertionError("must not be reached")

    with pytest.raises(RuntimeError):
        mg.init_mixer_exchange(H(), 1, 2, Dist)
