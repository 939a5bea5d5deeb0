"""Deterministic synthetic "dongles"  -- integer-only so that this numpy generator and the
HIP generator kernel (csrc/siggen.hip) emit IDENTICAL bytes.

The reference has no IQ generator: its ``generate_signal`` helper makes real-valued audio-rate test tones
(reference: src/generate_signal.cpp:32-86) and the ``file`` input replays raw u8 interleaved I/Q
(reference: src/input-file.cpp:82-147).  This module produces such u8 I/Q files' content: per dongle a
sum of keyed AM / NFM(+CTCSS sub-tone) carriers plus approximately Gaussian noise.

All arithmetic is fixed point:
  * phases are u32 turns (2**32 = one turn), advanced by ``step * n`` (mod 2**32);
  * sin/cos come from a 4096-entry int16 Q15 table (index = phase >> 20);
  * amplitudes are Q8 ADC counts; noise is an Irwin-Hall(4) sum of hash bytes scaled by a Q8 multiplier;
  * sample value = clamp(128 + floor(x_q8 / 256), 0, 255)   (u8, I then Q).
"""
from __future__ import annotations

import dataclasses
from typing import List, Sequence

import numpy as np

SIN_BITS = 12
SIN_LEN = 1 << SIN_BITS
MASK32 = np.uint64(0xFFFFFFFF)

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_DEVMUL = np.uint64(0xD1B54A32D192ED03)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def sin_table() -> np.ndarray:
    """int16 Q15 sine table shared by host and device generators."""
    i = np.arange(SIN_LEN, dtype=np.float64)
    return np.round(32767.0 * np.sin(2.0 * np.pi * i / SIN_LEN)).astype(np.int16)


def mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 output function on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def hash_u64(seed: int, dev: int, n: np.ndarray) -> np.ndarray:
    """Counter-based 64-bit hash keyed (seed, dongle, sample index)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (np.uint64(dev) * _DEVMUL)
        return mix64(base + n.astype(np.uint64) * _GOLDEN)


@dataclasses.dataclass
class Carrier:
    """One transmitter inside a dongle's passband (all fields integer, see module docstring)."""

    step: int            # carrier phase step per input sample, u32 turns
    amp_q8: int          # carrier amplitude, Q8 counts
    kind: int            # 0 = AM, 1 = NFM
    step_mod: int        # audio tone phase step, u32 turns
    am_depth_q15: int    # AM modulation depth, Q15
    k_beta: int          # NFM: phase deviation multiplier (u32 turns per Q15 unit of sin)
    step_ctcss: int      # NFM sub-tone phase step (0 = none)
    k_beta_ctcss: int    # NFM sub-tone phase deviation multiplier
    key_period: int      # keying period in samples (0 = always on)
    key_on: int          # on-time in samples
    key_slot: int        # t0 = key_slot_len * ((dongle + key_slot) % 8)
    key_slot_len: int    # slot length in samples

    def as_row(self) -> List[int]:
        return [self.step, self.amp_q8, self.kind, self.step_mod, self.am_depth_q15, self.k_beta, self.step_ctcss, self.k_beta_ctcss,
                self.key_period, self.key_on, self.key_slot, self.key_slot_len]


CARRIER_FIELDS = 12


def _turns(freq_hz: float, rate_hz: float) -> int:
    return int(round(freq_hz / rate_hz * 2.0**32)) & 0xFFFFFFFF


def make_carrier(offset_hz: float, sample_rate: int, *, amplitude: float = 0.08, kind: int = 0, tone_hz: float = 1000.0, am_depth: float = 0.5,
                 deviation_hz: float = 2500.0, ctcss_hz: float = 0.0, ctcss_dev_hz: float = 500.0, key_period_s: float = 1.5, key_on_s: float = 0.75,
                 key_slot: int = 0, key_slot_s: float = 0.125) -> Carrier:
    """Carrier ``offset_hz`` away from the dongle's centre frequency (SURVEY.md 8d defaults)."""
    beta = deviation_hz / tone_hz if kind == 1 else 0.0            # FM modulation index (radians)
    k_beta = int(round(beta / (2 * np.pi) * 2.0**32 / 32768.0))
    beta_c = ctcss_dev_hz / ctcss_hz if (kind == 1 and ctcss_hz > 0) else 0.0
    k_beta_c = int(round(beta_c / (2 * np.pi) * 2.0**32 / 32768.0))
    return Carrier(step=_turns(offset_hz, sample_rate), amp_q8=int(round(amplitude * 127.5 * 256)), kind=kind, step_mod=_turns(tone_hz, sample_rate),
                   am_depth_q15=int(round(am_depth * 32768)), k_beta=k_beta, step_ctcss=_turns(ctcss_hz, sample_rate) if ctcss_hz > 0 else 0,
                   k_beta_ctcss=k_beta_c, key_period=int(round(key_period_s * sample_rate)), key_on=int(round(key_on_s * sample_rate)), key_slot=key_slot,
                   key_slot_len=int(round(key_slot_s * sample_rate)))


def carrier_table(carriers: Sequence[Carrier]) -> np.ndarray:
    """int64 [n_carriers, CARRIER_FIELDS] table handed verbatim to the device generator."""
    return np.array([c.as_row() for c in carriers], dtype=np.int64).reshape(len(carriers), CARRIER_FIELDS)


def noise_mul_q8(sigma: float = 0.02) -> int:
    """Q8 multiplier turning the Irwin-Hall(4) byte sum (sigma 147.8) into Q8 counts of std ``sigma``*127.5."""
    ih_sigma = np.sqrt(4.0 * (256.0**2 - 1.0) / 12.0)
    return int(round(sigma * 127.5 * 256.0 / ih_sigma * 256.0))


def plan_shift_bins(dev: int, n_plans: int, n_channels: int = 8) -> List[int]:
    """Fleets whose dongles do not share a channel plan (bench.py --distinct-plans, airband_hip_set_signal_plan_shift): dongle ``dev`` belongs to plan
    p = dev mod n_plans and its channel c sits ((p >> 2c) & 3) shift units above the common plan's frequency -- 4**8 = 65 536 distinct plans of eight channels."""
    p = dev % n_plans if n_plans > 1 else 0
    return [(p >> (2 * (c & 15))) & 3 for c in range(n_channels)]


def generate_u8(dev: int, start_sample: int, n_samples: int, carriers: Sequence[Carrier], *, seed: int = 0x5EED, noise_q8: int | None = None,
                chunk: int = 1 << 20, n_plans: int = 1, shift_step: int = 0) -> np.ndarray:
    """u8 interleaved I/Q bytes [2*n_samples] for dongle ``dev``, stream samples [start, start+n).  n_plans / shift_step (u32 turns per sample): see plan_shift_bins."""
    if noise_q8 is None:
        noise_q8 = noise_mul_q8()
    tab = sin_table().astype(np.int64)
    out = np.empty(2 * n_samples, dtype=np.uint8)
    for c0 in range(0, n_samples, chunk):
        m = min(chunk, n_samples - c0)
        n = np.arange(start_sample + c0, start_sample + c0 + m, dtype=np.uint64)
        acc_i = np.zeros(m, dtype=np.int64)
        acc_q = np.zeros(m, dtype=np.int64)
        for ci, c in enumerate(carriers):
            with np.errstate(over="ignore"):
                ph0 = mix64(np.array([(seed ^ 0xC0FFEE) + ((dev << 8) | ci)], dtype=np.uint64))[0] & MASK32
                step = (c.step + shift_step * plan_shift_bins(dev, n_plans, len(carriers))[ci]) & 0xFFFFFFFF
                ph = (np.uint64(step) * n + ph0) & MASK32
                pa = (np.uint64(c.step_mod) * n) & MASK32
            s_mod = tab[(pa >> np.uint64(32 - SIN_BITS)).astype(np.int64)]
            if c.kind == 0:
                amp = (c.amp_q8 * (32768 + ((c.am_depth_q15 * s_mod) >> 15))) >> 15
            else:
                amp = np.full(m, c.amp_q8, dtype=np.int64)
                dphi = c.k_beta * s_mod
                if c.step_ctcss:
                    with np.errstate(over="ignore"):
                        pc = (np.uint64(c.step_ctcss) * n) & MASK32
                    dphi = dphi + c.k_beta_ctcss * tab[(pc >> np.uint64(32 - SIN_BITS)).astype(np.int64)]
                ph = (ph.astype(np.int64) + dphi).astype(np.uint64) & MASK32
            if c.key_period:
                t0 = c.key_slot_len * ((dev + c.key_slot) % 8)
                on = ((n + np.uint64(c.key_period - (t0 % c.key_period))) % np.uint64(c.key_period)) < np.uint64(c.key_on)
                amp = np.where(on, amp, 0)
            idx = (ph >> np.uint64(32 - SIN_BITS)).astype(np.int64)
            acc_i += (amp * tab[(idx + SIN_LEN // 4) & (SIN_LEN - 1)]) >> 15
            acc_q += (amp * tab[idx]) >> 15
        h = hash_u64(seed, dev, n)
        b = [((h >> np.uint64(8 * k)) & np.uint64(0xFF)).astype(np.int64) for k in range(8)]
        n_i = b[0] + b[1] + b[2] + b[3] - 510
        n_q = b[4] + b[5] + b[6] + b[7] - 510
        acc_i += (n_i * noise_q8) >> 8
        acc_q += (n_q * noise_q8) >> 8
        out[2 * c0:2 * (c0 + m):2] = np.clip(128 + (acc_i >> 8), 0, 255).astype(np.uint8)
        out[2 * c0 + 1:2 * (c0 + m):2] = np.clip(128 + (acc_q >> 8), 0, 255).astype(np.uint8)
    return out


# ---------------------------------------------------------------------------------------------------
# BASELINE.json channel plans
# ---------------------------------------------------------------------------------------------------
CENTERFREQ = 120_000_000
SAMPLE_RATE = 2_560_000
PLAN_OFFSETS_HZ = [-1_000_000, -750_000, -500_000, -250_000, 250_000, 500_000, 750_000, 1_000_000]


def baseline_plan(mixed: bool, key_on_s: float = 0.75):
    """(channel config dicts, carriers) for one dongle of the BASELINE configs.

    mixed=False: 8 AM channels (configs #1/#2).  mixed=True: odd channels NFM; c%4==1 carries a 100 Hz CTCSS
    sub-tone with ``ctcss=100`` and ``notch=100``; c%4==3 has ``bandwidth=12500`` (configs #3-#5).
    key_on_s: seconds of every 1.5 s a transmitter is keyed (0.75; bench.py --key-on-s measures quieter bands).
    """
    chans, carriers = [], []
    for c, off in enumerate(PLAN_OFFSETS_HZ):
        cfg = dict(frequency=CENTERFREQ + off, modulation=0, afc=0, squelch_threshold_dbfs=0, squelch_snr_threshold_db=-1.0, notch_freq=0.0, notch_q=0.0,
                   ctcss_freq=0.0, bandwidth_hz=0, ampfactor=1.0, tau_us=-1, has_iq_outputs=0)
        if mixed and (c % 2 == 1):
            cfg["modulation"] = 1
            if c % 4 == 1:
                cfg["ctcss_freq"] = 100.0
                cfg["notch_freq"] = 100.0
                carriers.append(make_carrier(off, SAMPLE_RATE, kind=1, ctcss_hz=100.0, key_slot=c, key_on_s=key_on_s))
            else:
                cfg["bandwidth_hz"] = 12500
                carriers.append(make_carrier(off, SAMPLE_RATE, kind=1, key_slot=c, key_on_s=key_on_s))
        else:
            carriers.append(make_carrier(off, SAMPLE_RATE, kind=0, key_slot=c, key_on_s=key_on_s))
        chans.append(cfg)
    return chans, carriers
