/* csrc/exact_math.h -- correctly rounded sqrtf(x) and x / (a per-channel constant) in fewer vector instructions than the compiler's general
 * sequences, for the demod kernels (demod.hip), whose results have to equal the reference's IEEE float operations bit for bit.
 *
 * Why: the lane-per-channel kernels are vector-issue-bound (profiles/r03_summary.md: instructions x 4 cycles / SIMDs accounts for the time),
 * and with -fhip-fp32-correctly-rounded-divide-sqrt
 *   sqrtf(x) is 16 instructions: scale x up if it is below 2^-96 (3), v_sqrt_f32 (<= 1 ulp, quarter rate), pick the correctly rounded one of
 *            {s - 1 ulp, s, s + 1 ulp} by the sign of two exact FMA residuals (8), scale back (2), pass 0 / inf through (2);
 *   x / g    is 11: two v_div_scale, v_rcp_f32 (quarter rate), two Newton steps on the reciprocal, quotient + two residual corrections,
 *            v_div_fmas, v_div_fixup.
 * The NFM + lowpass kind does three such square roots and two such divisions (by LowpassFilter's gain, src/filters.cpp:146-163) per sample.
 *
 *  - ab_sqrt_rn(): the SAME sequence without the scaling and the pass-through -- i.e. exactly what the compiler's sequence computes for an
 *    x it does not scale.  The residual test is exact as long as s * s does not underflow, which is what the compiler's 2^-96 threshold is
 *    for; 0, -0, inf, NaN and negative x come out right without the pass-through (the neighbours of 0 / inf are NaN or a denormal whose
 *    residual compares false).  Lanes with x < 2^-96 (zeros among them: that keeps the test one compare) take the compiler's sequence
 *    behind a branch that a wavefront without such a lane skips: 10 instructions instead of 16, four at once 39.
 *  - ab_div_const(): q = RN(x * r), e = x - g * q (one FMA), RN(q + e * r) with r = RN(1 / g) -- three instructions.  For a GIVEN g this is
 *    the correctly rounded x / g for every x or it is not (Brisebarre, Muller, Raina: "Accelerating correctly rounded floating-point
 *    division when the divisor is known in advance", IEEE TC 2004); rather than port their criterion, params.cpp TRIES all 2^23
 *    significands of x against the host's IEEE division when the plan is built (div_const_reciprocal(), ~70 ms per distinct g) and hands
 *    the kernels r only if every one matched (else 0: those channels divide the slow way).  The operations scale exactly with x's exponent
 *    while nothing under- or overflows, so one binade of x decides all of 2^-60 <= |x| <= 2^60; lanes outside that range (zeros included:
 *    the sign of a zero quotient is not preserved) take the IEEE division behind a branch.
 *
 * The header also compiles as plain C++ (tests/host_exact_math.cpp): there AB_HW_SQRT is a correctly rounded square root nudged by the
 * test to either neighbour, which covers everything a <= 1 ulp v_sqrt_f32 can return.
 */
#ifndef AIRBAND_CSRC_EXACT_MATH_H
#define AIRBAND_CSRC_EXACT_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>

namespace airband {

#if defined(__HIPCC__)
#define AB_EM_FN __device__ __forceinline__
#define AB_EM_UNLIKELY(x) __builtin_expect(!!(x), 0)
#ifndef AB_HW_SQRT
#define AB_HW_SQRT(x) __builtin_amdgcn_sqrtf(x) /* v_sqrt_f32: 1 ulp */
#endif
AB_EM_FN unsigned ab_bits(float f) { return __float_as_uint(f); }
AB_EM_FN float ab_float(unsigned u) { return __uint_as_float(u); }
#else
#define AB_EM_FN static inline
#define AB_EM_UNLIKELY(x) (x)
#ifndef AB_HW_SQRT
#define AB_HW_SQRT(x) sqrtf(x) /* (the test harness defines its own: a square root that is off by an ulp either way) */
#endif
AB_EM_FN unsigned ab_bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
AB_EM_FN float ab_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
#endif

constexpr unsigned AB_SQRT_SMALL_BITS = 0x0f800000u; /* 2^-96: below it the compiler's sequence scales x (and so must we: the residuals underflow) */
constexpr float AB_DIV_CONST_LO = 0x1p-60f, AB_DIV_CONST_HI = 0x1p60f;

/* x < 2^-96.  ab_sqrt_core() is not good for 0 < x < 2^-96 only; zeros and negative x go the general way with them because that makes the test one
 * compare, and the minimum of four sums of squares one compare for the four (a channel whose bin is exactly 0 is a dongle that delivers no noise) */
AB_EM_FN bool ab_sqrt_is_small(float x) { return !(x >= ab_float(AB_SQRT_SMALL_BITS)); } /* (a NaN goes the general way too: whatever fminf() makes of one) */

AB_EM_FN float ab_sqrt_core(float x) {
    const float s = AB_HW_SQRT(x);
    const float below = ab_float(ab_bits(s) - 1u), above = ab_float(ab_bits(s) + 1u);
    const float r_below = __builtin_fmaf(-below, s, x); /* x - (s - ulp) * s, exact in sign */
    const float r_above = __builtin_fmaf(-above, s, x);
    float r = (0.0f >= r_below) ? below : s; /* s was too large */
    r = (0.0f < r_above) ? above : r;        /* s was too small */
    return r;
}
/* sqrtf(x), correctly rounded, for any x */
AB_EM_FN float ab_sqrt_rn(float x) {
    float r = ab_sqrt_core(x);
    if (AB_EM_UNLIKELY(ab_sqrt_is_small(x))) r = sqrtf(x);
    return r;
}
/* four at once: one test, one seldom-taken branch.  (x[] are results of arithmetic -- sums of squares: a NaN among them is a quiet one, which fminf()
 * skips and the core returns unchanged.) */
AB_EM_FN void ab_sqrt_rn4(const float* x, float* out) {
    for (int i = 0; i < 4; i++) out[i] = ab_sqrt_core(x[i]);
    if (AB_EM_UNLIKELY(ab_sqrt_is_small(__builtin_fminf(__builtin_fminf(x[0], x[1]), __builtin_fminf(x[2], x[3])))))
        for (int i = 0; i < 4; i++) out[i] = sqrtf(x[i]);
}

/* x / g for 2^-60 <= |x| <= 2^60, given r = div_const_reciprocal(g) != 0 (params.cpp) */
AB_EM_FN float ab_div_const_core(float x, float g, float r) {
    const float q = x * r;
    const float e = __builtin_fmaf(-g, q, x);
    return __builtin_fmaf(e, r, q);
}
/* `lo` is AB_DIV_CONST_LO for a channel whose r is usable and +inf otherwise: then no x is in range and the division is the general one */
AB_EM_FN bool ab_div_const_in_range(float x, float lo) {
    const float a = __builtin_fabsf(x);
    return (unsigned)(a >= lo) & (unsigned)(a <= AB_DIV_CONST_HI);
}
/* (xr / g, xi / g): LowpassFilter::apply divides both parts of the sample by the same gain */
AB_EM_FN void ab_div_const2(float xr, float xi, float g, float r, float lo, float& qr, float& qi) {
    qr = ab_div_const_core(xr, g, r);
    qi = ab_div_const_core(xi, g, r);
    const float ar = __builtin_fabsf(xr), ai = __builtin_fabsf(xi);
    /* both in [lo, 2^60]: the smaller against lo, the larger against the top (fminf / fmaxf skip a NaN operand, and a NaN needs no general path:
     * either way its quotient is a NaN) */
    const bool ok = (__builtin_fminf(ar, ai) >= lo) & (__builtin_fmaxf(ar, ai) <= AB_DIV_CONST_HI);
    if (AB_EM_UNLIKELY(!ok)) {
        qr = xr / g;
        qi = xi / g;
    }
}

}  // namespace airband
#endif
