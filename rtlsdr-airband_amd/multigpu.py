"""Multi-GPU host logic: one process per GPU, dongles sharded contiguously across ranks (the reference's own
multiple_demod_threads partitioning, src/rtl_airband.cpp:1052-1086), no data-path collective -- except the mixer
sum of BASELINE config #5, the one real exchange step in the reference's data flow (src/mixer.cpp:133-140,201-214):
every rank reduces its local inputs into per-mixer partial sums on its GPU and the partials are all-reduced in place by the
library itself (airband_hip_allreduce_mixers: librccl over xGMI, on the handle's stream).  What lives here is the host logic around
it: the partition, BASELINE configs[4]'s wiring, how the communicator id reaches the ranks, and a reference-order host sum."""
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


def init_mixer_exchange(hip, rank: int, world: int, dist=None, unique_id=None):
    """One process per GPU: gives the handle its rank in the RCCL communicator the mixer sums are all-reduced over
    (airband_hip_comm_unique_id on rank 0, airband_hip_comm_init_rank on every rank; include/airband_hip.h).  torch.distributed (`dist`,
    any backend: nccl under bench.py --gpus N, gloo in the CPU tests) only carries rank 0's 128-byte id to the other ranks; the data
    path never goes through it -- the per-batch exchange is hip.allreduce_mixers(), librccl called by the library on the handle's stream.
    `unique_id`: rank 0's id source (default: the library's).  Returns the id every rank ended up with."""
    if dist is None:
        import torch.distributed as dist
    box = [(unique_id() if unique_id is not None else type(hip).comm_unique_id()) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != 128:
        raise RuntimeError("rank %d: no communicator id arrived from rank 0" % rank)
    hip.comm_init_rank(bytes(box[0]), world, rank)
    return bytes(box[0])


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
