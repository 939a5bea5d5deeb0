// tests/host_exact_math.cpp -- TEST HARNESS ONLY (built by tests/test_exact_math.py with g++, never part of the library).
// Compiles csrc/exact_math.h as plain C++ and checks its two short sequences against the host's IEEE sqrtf / division:
//  - the square root with v_sqrt_f32 replaced by a correctly rounded root moved by -1 / 0 / +1 ulp (whenever that is still within
//    1 ulp of the true root, the instruction's documented accuracy), over every significand of two neighbouring binades;
//  - the division by a constant over every significand (that is params.cpp's own check, div_const_reciprocal) and over random
//    dividends of every exponent, zeros, infinities and values outside the sequence's range (ab_div_const2 has to branch for those).
#include <cmath>
#include <cstdint>
#include <cstring>

static int g_nudge = 0; /* ulps added to the correctly rounded root */
static float nudged_sqrt(float x);
#define AB_HW_SQRT(x) nudged_sqrt(x)
#include "../rtlsdr-airband_amd/csrc/exact_math.h"
#include "../rtlsdr-airband_amd/csrc/params.h"

using namespace airband;

static float nudged_sqrt(float x) {
    const float s = sqrtf(x);
    if (g_nudge == 0 || !(s > 0.0f) || std::isinf(s)) return s;
    const float t = ab_float(ab_bits(s) + (unsigned)g_nudge);
    /* only roots a 1-ulp instruction may return */
    const double exact = std::sqrt((double)x), ulp = (double)ab_float(ab_bits(s) + 1u) - (double)s;
    return std::fabs((double)t - exact) <= ulp ? t : s;
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

/* the kernels hand these functions results of arithmetic, and arithmetic never returns a SIGNALING NaN (fminf(small, sNaN) is a quiet NaN that the
 * next fminf() skips, and the small value with it) */
static float quiet(float x) { return x != x ? ab_float(ab_bits(x) | 0x00400000u) : x; }

extern "C" {

// every significand of [2^e, 2^(e+2)) with the root nudged by -1, 0, +1 ulp; returns the number of wrong results
int64_t em_sqrt_binades(int e, int step) {
    int64_t bad = 0;
    for (g_nudge = -1; g_nudge <= 1; g_nudge++)
        for (int k = 0; k < 2; k++)
            for (uint32_t m = 0; m < (1u << 23); m += (uint32_t)step) {
                const float x = ab_float(((uint32_t)(127 + e + k) << 23) | m);
                if (ab_bits(ab_sqrt_rn(x)) != ab_bits(sqrtf(x))) bad++;
            }
    g_nudge = 0;
    return bad;
}

// special values, the 2^-96 seam, random floats of every exponent (also negative, NaN), and the four-at-once form
int64_t em_sqrt_specials(int n_random) {
    int64_t bad = 0;
    auto same = [](float a, float b) { return (a != a && b != b) || ab_bits(a) == ab_bits(b); };
    const uint32_t fixed[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x007fffffu, 0x00800000u, 0x0f7fffffu, 0x0f800000u, 0x0f800001u, 0x3f800000u,
                              0x7f7fffffu, 0x7f800000u, 0x7fc00000u, 0xbf800000u, 0xff800000u, 0x0f000000u, 0x10000000u};
    for (g_nudge = -1; g_nudge <= 1; g_nudge++) {
        for (uint32_t u : fixed)
            if (!same(ab_sqrt_rn(ab_float(u)), sqrtf(ab_float(u)))) bad++;
        for (int i = 0; i < n_random; i++) {
            float x[4], got[4];
            for (int k = 0; k < 4; k++) x[k] = quiet(ab_float(rnd()));
            if ((i & 7) == 0) x[i & 3] = ab_float(rnd() & 0x0fffffffu); /* something small in the group now and then */
            ab_sqrt_rn4(x, got);
            for (int k = 0; k < 4; k++) {
                if (!same(got[k], sqrtf(x[k]))) bad++;
                if (!same(ab_sqrt_rn(x[k]), sqrtf(x[k]))) bad++;
            }
        }
    }
    g_nudge = 0;
    return bad;
}

// params.cpp's verdict on a divisor (0 = not usable)
float em_div_reciprocal(float g) { return div_const_reciprocal(g); }

// ab_div_const2 against the IEEE quotient: random dividends of every exponent and sign, zeros, infinities, NaN; lo as the kernels derive it
int64_t em_div_random(float g, int n_random) {
    const float r = div_const_reciprocal(g);
    const float lo = r != 0.0f ? AB_DIV_CONST_LO : INFINITY;
    int64_t bad = 0;
    auto same = [](float a, float b) { return (a != a && b != b) || ab_bits(a) == ab_bits(b); };
    const uint32_t fixed[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x00800000u, 0x7f7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x21800000u /* 2^-60 */,
                              0x217fffffu, 0x5d800000u /* 2^60 */, 0x5d800001u, 0x3f800000u, 0xbf800000u};
    for (uint32_t a : fixed)
        for (uint32_t b : fixed) {
            float qr, qi;
            ab_div_const2(ab_float(a), ab_float(b), g, r, lo, qr, qi);
            if (!same(qr, ab_float(a) / g) || !same(qi, ab_float(b) / g)) bad++;
        }
    for (int i = 0; i < n_random; i++) {
        const float xr = ab_float(rnd()), xi = ab_float(rnd());
        float qr, qi;
        ab_div_const2(xr, xi, g, r, lo, qr, qi);
        if (!same(qr, xr / g) || !same(qi, xi / g)) bad++;
        /* the three-instruction core alone, wherever it claims to be valid */
        if (r != 0.0f && ab_div_const_in_range(xr, lo) && ab_bits(ab_div_const_core(xr, g, r)) != ab_bits(xr / g)) bad++;
    }
    return bad;
}

// the lowpass gain and its reciprocal as a plan holds them for one NFM channel of `bandwidth_hz`
int em_plan_gain(int bandwidth_hz, int wave_rate, float* gain, float* rgain) {
    airband_hip_channel_cfg ch;
    std::memset(&ch, 0, sizeof(ch));
    ch.frequency = 120100000;
    ch.modulation = AIRBAND_MOD_NFM;
    ch.squelch_snr_threshold_db = -1.0f;
    ch.ampfactor = 1.0f;
    ch.tau_us = -1;
    ch.bandwidth_hz = bandwidth_hz;
    airband_hip_device_cfg dv;
    std::memset(&dv, 0, sizeof(dv));
    dv.sample_rate = 2560000;
    dv.centerfreq = 120000000;
    dv.sfmt = AIRBAND_SFMT_U8;
    dv.tau_us = -1;
    dv.channel_count = 1;
    dv.channels = &ch;
    airband_hip_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = AIRBAND_HIP_ABI_VERSION;
    cfg.fft_size_log = 9;
    cfg.wave_rate = wave_rate;
    cfg.device_count = 1;
    cfg.devices = &dv;
    Plan plan;
    if (build_plan(&cfg, plan) != 0) return -1;
    *gain = plan.cc[0].lp_gain;
    *rgain = plan.cc[0].lp_rgain;
    return 0;
}
}
