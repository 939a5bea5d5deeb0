"""csrc/exact_math.h -- the demod kernels' short sequences for a correctly rounded sqrtf() and for x / (a per-channel constant) --
compiled as plain C++ and compared with the host's IEEE operations (tests/host_exact_math.cpp).  CPU only: this proves the
ARITHMETIC of the sequences (every significand, the seams of their ranges, special values); that the kernels use them
unchanged is what the bit-exact GPU parity tests (test_gpu_parity.py) then show.
"""
import ctypes as C
import math
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "rtlsdr-airband_amd", "csrc")


@pytest.fixture(scope="module")
def em(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("exactmath") / "libexactmath.so")
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-I" + os.path.join(REPO, "include"), "-o", out,
           os.path.join(HERE, "host_exact_math.cpp"), os.path.join(CSRC, "params.cpp")]
    subprocess.run(cmd, check=True)
    lib = C.CDLL(out)
    lib.em_sqrt_binades.argtypes = [C.c_int, C.c_int]
    lib.em_sqrt_binades.restype = C.c_int64
    lib.em_sqrt_specials.argtypes = [C.c_int]
    lib.em_sqrt_specials.restype = C.c_int64
    lib.em_div_reciprocal.argtypes = [C.c_float]
    lib.em_div_reciprocal.restype = C.c_float
    lib.em_div_random.argtypes = [C.c_float, C.c_int]
    lib.em_div_random.restype = C.c_int64
    lib.em_plan_gain.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.em_plan_gain.restype = C.c_int
    return lib


def test_sqrt_every_significand_with_the_hardware_root_off_by_an_ulp(em):
    # an even and an odd exponent side by side cover every (significand, exponent parity) a square root can see
    assert em.em_sqrt_binades(0, 1) == 0
    # the same at the edges of the range the short sequence is used in: just above 2^-96, and the two largest binades
    assert em.em_sqrt_binades(-96, 7) == 0
    assert em.em_sqrt_binades(126, 7) == 0
    assert em.em_sqrt_binades(-41, 13) == 0


def test_sqrt_special_values_small_inputs_and_four_at_once(em):
    assert em.em_sqrt_specials(400_000) == 0


def plan_gain(em, bandwidth, wave_rate):
    g, r = C.c_float(), C.c_float()
    assert em.em_plan_gain(bandwidth, wave_rate, C.byref(g), C.byref(r)) == 0
    return g.value, r.value


@pytest.mark.parametrize("bandwidth,wave_rate", [(12500, 16000), (6250, 16000), (5000, 16000), (25000, 16000)])
def test_the_plans_lowpass_gains_divide_in_three_instructions(em, bandwidth, wave_rate):
    """build_plan() tries every significand (div_const_reciprocal); BASELINE configs[2]-[4] use bandwidth 12500 at 16 kHz."""
    g, r = plan_gain(em, bandwidth, wave_rate)
    assert g > 0 and r != 0.0
    assert r == C.c_float(1.0 / g).value
    assert em.em_div_random(g, 300_000) == 0


@pytest.mark.parametrize("g", [1.0, 3.0, 0.1, 1234.567, 7.0e5, 1.9999999, 2.0 ** -30, 2.0 ** 30, math.pi])
def test_division_by_other_constants(em, g):
    gf = C.c_float(g).value
    r = em.em_div_reciprocal(gf)
    assert r != 0.0  # (no divisor is known for which the corrected product fails; the check is there because no proof is offered)
    assert em.em_div_random(gf, 200_000) == 0


@pytest.mark.parametrize("g", [0.0, -3.0, 2.0 ** -50, 2.0 ** 50, float("inf"), float("nan")])
def test_divisors_outside_the_checked_range_fall_back_to_the_general_division(em, g):
    gf = C.c_float(g).value
    assert em.em_div_reciprocal(gf) == 0.0
    if g == g and g != 0.0:
        assert em.em_div_random(gf, 50_000) == 0  # lo = +inf: every lane takes the IEEE division
