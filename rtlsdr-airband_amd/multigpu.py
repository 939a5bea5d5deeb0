"""Multi-GPU host logic: one process per GPU, dongles sharded contiguously across ranks (the reference's own
multiple_demod_threads partitioning, src/rtl_airband.cpp:1052-1086), no data-path collective -- except the mixer
sum of BASELINE config #5, the one real exchange step in the reference's data flow (src/mixer.cpp:133-140,201-214):
every rank reduces its local inputs into per-mixer partial sums and the partials are all-reduced (RCCL over xGMI on
GPUs, gloo in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous dongle range [start, end) of `rank`: device_start/device_end of demod_params_t."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def baseline_mixer_inputs(d_start: int, d_end: int, channels_per_dongle: int, n_mixers: int) -> List[tuple]:
    """SURVEY.md 8d config #5 wiring: channel (d, c) -> mixer (d*C + c) mod n_mixers, ampfactor 1, balance 0.
    Device indices are LOCAL to the rank (0 .. d_end-d_start), mixer indices are global."""
    return [(d - d_start, c, (d * channels_per_dongle + c) % n_mixers, 1.0, 0.0) for d in range(d_start, d_end) for c in range(channels_per_dongle)]


class _DevicePtr:
    """__cuda_array_interface__ shim: a torch view over memory the library owns (no copy, no ownership)."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


def device_mixer_views(hip, n_mixers: int, stereo: bool = False):
    """torch tensors over a handle's device-side mixer sums (airband_hip_device_results): (left [M][B] f32, right or None, has_signal [M] u8).
    They alias the library's buffers: what an RCCL all-reduce writes is what airband_hip_collect_mixers() then reads."""
    import torch

    res = hip.device_results()
    left = torch.as_tensor(_DevicePtr(res["mix_left"], (n_mixers, hip.B), "<f4"), device="cuda")
    right = torch.as_tensor(_DevicePtr(res["mix_right"], (n_mixers, hip.B), "<f4"), device="cuda") if stereo else None
    sig = torch.as_tensor(_DevicePtr(res["mix_signal"], (n_mixers,), "|u1"), device="cuda")
    return left, right, sig


def allreduce_mixers(left, right, has_signal, force: bool = False):
    """In-place all-reduce of per-rank mixer partials: SUM for the waveforms (the right channel too when any mixer is stereo), MAX
    for the signal flags (mixer channel axcindicate, src/mixer.cpp:209).  Tensors may live on CPU (gloo) or GPU (nccl = RCCL).
    force: also at world size 1 (plumbing check of the collective leg)."""
    import torch.distributed as dist

    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return
    dist.all_reduce(left, op=dist.ReduceOp.SUM)
    if right is not None:
        dist.all_reduce(right, op=dist.ReduceOp.SUM)
    dist.all_reduce(has_signal, op=dist.ReduceOp.MAX)


def mix_on_host(inputs: Sequence[tuple], chan_base: Sequence[int], waveout: np.ndarray, axc: np.ndarray, n_mixers: int):
    """Reference-order mixer sum on the host (numpy float32, connection order), for small consumers and tests."""
    B = waveout.shape[1]
    left = np.zeros((n_mixers, B), np.float32)
    right = np.zeros((n_mixers, B), np.float32)
    sig = np.zeros((n_mixers,), np.uint8)
    stereo = np.zeros((n_mixers,), bool)
    for d, c, m, amp, bal in inputs:
        if bal != 0.0:
            stereo[m] = True
    for d, c, m, amp, bal in inputs:
        ch = chan_base[d] + c
        if axc[ch] == ord(" "):
            continue
        ml = np.float32(amp) * np.float32(min(1.0, 1.0 - bal))
        mr = np.float32(amp) * np.float32(min(1.0, 1.0 + bal))
        if ml != 0:
            left[m] += waveout[ch] * ml
        if stereo[m] and mr != 0:
            right[m] += waveout[ch] * mr
        sig[m] = 1
    return left, right, sig
