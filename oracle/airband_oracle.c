/* oracle/airband_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Plain-C restatement of the reference's demodulate() hot path, written from the behaviour described in
 * SURVEY.md section 8 and checked bit-for-bit against the real reference compiled in place
 * (oracle/_ref, tests/test_oracle_vs_reference.py) and against committed golden vectors generated
 * from it (tests/golden/).  Every function cites the reference lines whose behaviour it restates.
 *
 * Build flags matter: -O2 -ffp-contract=off -fno-fast-math (IEEE float, no FMA contraction), the same
 * flags oracle/_ref is built with.  The FFT behind the reference's fftwf_* calls is oracle_fft.c in both.
 *
 * Comparisons are restated operand for operand (std::min(a, b) is (b < a) ? b : a): round 4 found the squelch's
 * two minima written the other way round -- equal for finite values, different for the NaN an unstable
 * lowpass produces (test_oracle_is_the_reference_on_random_plans, profiles/r04_experiments.md I).
 */
#define _GNU_SOURCE 1
#include "airband_oracle.h"

#include <complex.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "shim/fftw3.h"

#define AGC_EXTRA AIRBAND_AGC_EXTRA
#define MAX_TONES 52
#define SQ_BUF 102

enum { ST_CLOSED = 0, ST_OPENING = 1, ST_CLOSING = 2, ST_ABORT = 3, ST_OPEN = 4 };

/* ------------------------------------------------------------------------------------------------
 * CTCSS: Goertzel tone bank (reference: src/ctcss.cpp, src/ctcss.h)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int enabled;
    float ctcss_freq;
    int window;
    uint64_t found, not_found;
    int n_tones;
    float tone_freq[MAX_TONES], coeff[MAX_TONES], mag[MAX_TONES], q1[MAX_TONES], q2[MAX_TONES];
    int enough, count, has_tone;
} ctcss_t;

static const float k_standard_tones[51] = {67.0f,  69.3f,  71.9f,  74.4f,  77.0f,  79.7f,  82.5f,  85.4f,  88.5f,  91.5f,  94.8f,  97.4f,  100.0f,
                                           103.5f, 107.2f, 110.9f, 114.8f, 118.8f, 123.0f, 127.3f, 131.8f, 136.5f, 141.3f, 146.2f, 150.0f, 151.4f,
                                           156.7f, 159.8f, 162.2f, 165.5f, 167.9f, 171.3f, 173.8f, 177.3f, 179.9f, 183.5f, 186.2f, 189.9f, 192.8f,
                                           196.6f, 199.5f, 203.5f, 206.5f, 210.7f, 218.1f, 225.7f, 229.1f, 233.6f, 241.8f, 250.3f, 254.1f};

/* Goertzel coefficient of one detector (reference: src/ctcss.cpp:31-42): the bin index k is the tone
 * frequency rounded to the nearest multiple of rate/window; coeff = 2 cos(2 pi k / window). */
float orc_tone_coeff(float tone_freq, float sample_rate, int window) {
    int k = (int)(0.5 + (double)((float)window * tone_freq / sample_rate));
    float omega = (float)((2.0 * M_PI * k) / window);
    return (float)(2.0 * cos((double)omega));
}

static void ctcss_reset(ctcss_t* c) { /* src/ctcss.cpp:165-172 (+ToneDetector::reset :56-59: magnitude is NOT cleared) */
    if (!c->enabled) return;
    for (int i = 0; i < c->n_tones; i++) c->q1[i] = c->q2[i] = 0.0f;
    c->enough = 0;
    c->count = 0;
    c->has_tone = 0;
}

static void ctcss_add(ctcss_t* c, float f, float rate) { /* src/ctcss.cpp:61-73: drop detectors with a duplicate coefficient */
    float co = orc_tone_coeff(f, rate, c->window);
    for (int i = 0; i < c->n_tones; i++)
        if (c->coeff[i] == co) return;
    c->tone_freq[c->n_tones] = f;
    c->coeff[c->n_tones] = co;
    c->mag[c->n_tones] = 0.0f;
    c->n_tones++;
}

static void ctcss_init(ctcss_t* c, float freq, float rate, int window) { /* src/ctcss.cpp:105-122 */
    memset(c, 0, sizeof(*c));
    c->enabled = 1;
    c->ctcss_freq = freq;
    c->window = window;
    ctcss_add(c, freq, rate); /* target first, then every standard tone at least 5 Hz away */
    for (int i = 0; i < 51; i++) {
        if (fabsf(freq - k_standard_tones[i]) < 5) continue;
        ctcss_add(c, k_standard_tones[i], rate);
    }
    ctcss_reset(c);
}

static void ctcss_sample(ctcss_t* c, float s) { /* src/ctcss.cpp:124-163 with ToneDetector::process_sample :44-54 */
    if (!c->enabled) return;
    for (int i = 0; i < c->n_tones; i++) {
        float q0 = c->coeff[i] * c->q1[i] - c->q2[i] + s;
        c->q2[i] = c->q1[i];
        c->q1[i] = q0;
    }
    c->count++;
    if (c->count < c->window) return;
    c->enough = 1;
    /* window complete: tone present iff the target's power equals the maximum AND exceeds the mean */
    float total = 0.0f, best = 0.0f;
    for (int i = 0; i < c->n_tones; i++) {
        float m = c->q1[i] * c->q1[i] + c->q2[i] * c->q2[i] - c->q1[i] * c->q2[i] * c->coeff[i];
        c->mag[i] = m;
        total += m;
        if (i == 0 || m > best) best = m;
    }
    float avg = total / (float)c->n_tones;
    float target = c->mag[0]; /* the target is always detector 0 */
    if (target == best && target > avg) {
        c->has_tone = 1;
        c->found++;
    } else {
        c->has_tone = 0;
        c->not_found++;
    }
    for (int i = 0; i < c->n_tones; i++) c->q1[i] = c->q2[i] = 0.0f;
    c->count = 0;
}

static int ctcss_has_tone(const ctcss_t* c) { return !c->enabled || c->has_tone; } /* src/ctcss.h:82 */

/* ------------------------------------------------------------------------------------------------
 * Squelch (reference: src/squelch.cpp, src/squelch.h)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float noise_floor;
    int manual;
    float manual_level, normal_ratio, flappy_ratio, cap;
    float pre_full, pre_capped, post_full, post_capped;
    float level_cache;
    int using_post;
    float pre_vs_post;
    int open_delay, close_delay, low_abort;
    int next, cur;
    int delay;
    uint64_t open_count, sample_count, flappy_count;
    int low_count;
    uint64_t recent_size, flap_threshold, recent_open, closed_count;
    int head, tail;
    float buf[SQ_BUF];
    ctcss_t fast, slow;
} squelch_t;

static void sq_recalc_cap(squelch_t* s) { /* src/squelch.cpp:492-499 */
    if (s->manual)
        s->cap = 1.5f * s->manual_level;
    else
        s->cap = 1.5f * s->normal_ratio * s->noise_floor;
}

static void sq_set_snr(squelch_t* s, float db) { /* src/squelch.cpp:93-103 */
    s->manual = 0;
    s->normal_ratio = (float)pow(10.0, db / 20.0);
    s->flappy_ratio = s->normal_ratio * 0.9f;
    sq_recalc_cap(s);
}

static void sq_set_level(squelch_t* s, float level) { /* src/squelch.cpp:79-91 */
    if (level > 0) {
        s->manual = 1;
        s->manual_level = level;
    } else {
        s->manual = 0;
    }
    sq_recalc_cap(s);
}

static void sq_init(squelch_t* s) { /* src/squelch.cpp:36-77 */
    memset(s, 0, sizeof(*s));
    s->noise_floor = 5.0f;
    sq_set_snr(s, 9.54f);
    s->manual_level = -1.0f;
    s->pre_full = s->pre_capped = 0.001f;
    s->post_full = s->post_capped = 0.001f;
    s->level_cache = 0.0f;
    s->using_post = 0;
    s->pre_vs_post = 0.9f;
    s->open_delay = 197;
    s->close_delay = 197;
    s->low_abort = 88;
    s->next = s->cur = ST_CLOSED;
    s->sample_count = UINT64_MAX; /* size_t -1: the first sample wraps it to 0 */
    s->recent_size = 1000;
    s->flap_threshold = 3;
    s->head = 0;
    s->tail = 1;
}

static int sq_flapping(const squelch_t* s) { return s->recent_open >= s->flap_threshold; }

static float sq_level(squelch_t* s) { /* src/squelch.cpp:164-177: lazily cached, 0 means "recompute" */
    if (s->manual) return s->manual_level;
    if (s->level_cache == 0.0f) {
        if (sq_flapping(s) && s->flappy_ratio < s->normal_ratio)
            s->level_cache = s->flappy_ratio * s->noise_floor;
        else
            s->level_cache = s->normal_ratio * s->noise_floor;
    }
    return s->level_cache;
}

static int sq_has_pre(squelch_t* s) { return s->pre_capped >= sq_level(s); }                       /* :462-464 */
static int sq_has_post(squelch_t* s) { return s->using_post && s->post_capped >= s->buf[s->tail]; } /* :466-468 */
static int sq_has_signal(squelch_t* s) {                                                            /* :470-475 */
    if (s->using_post) return sq_has_pre(s) && sq_has_post(s);
    return sq_has_pre(s);
}

static void sq_request(squelch_t* s, int want) { /* src/squelch.cpp:297-361: clamp illegal transitions */
    switch (s->cur) {
        case ST_CLOSED:
            if (want == ST_CLOSING || want == ST_ABORT) want = ST_CLOSED;
            else if (want == ST_OPEN) want = ST_OPENING;
            break;
        case ST_OPENING:
            if (want == ST_ABORT) want = ST_CLOSED;
            break;
        case ST_ABORT:
            if (want != ST_ABORT && want != ST_CLOSED) want = ST_CLOSED;
            break;
        case ST_OPEN:
            if (want == ST_CLOSED) want = ST_CLOSING;
            else if (want == ST_OPENING) want = ST_OPEN;
            break;
        default:
            break;
    }
    s->next = want;
}

static void sq_advance(squelch_t* s) { /* src/squelch.cpp:363-460: apply the decision taken during the previous sample */
    switch (s->next) {
        case ST_OPENING:
            if (s->cur != ST_OPENING) {
                s->delay = 0;
                s->low_count = 0;
                s->using_post = 0;
                s->cur = ST_OPENING;
            } else if (++s->delay >= s->open_delay) {
                if (s->closed_count < s->recent_size) { /* count as a recent open for flap detection */
                    s->recent_open++;
                    if (sq_flapping(s)) s->flappy_count++;
                    s->level_cache = 0.0f;
                }
                s->next = sq_has_signal(s) ? ST_OPEN : ST_CLOSED;
            }
            break;
        case ST_CLOSING:
            if (s->cur != ST_CLOSING) {
                s->delay = 0;
                s->cur = ST_CLOSING;
            } else if (++s->delay >= s->close_delay) {
                if (!sq_has_signal(s)) {
                    s->next = ST_CLOSED;
                } else { /* signal came back: straight to OPEN without counting an open */
                    s->cur = ST_OPEN;
                    s->next = ST_OPEN;
                }
            }
            break;
        case ST_ABORT:
            if (s->cur != ST_ABORT) {
                if (s->cur != ST_CLOSING) s->delay = 0; /* keep CLOSING's running delay */
                s->cur = ST_ABORT;
            } else if (++s->delay >= s->close_delay) {
                s->next = ST_CLOSED;
            }
            break;
        case ST_OPEN:
            if (s->cur != ST_OPEN) {
                s->open_count++;
                s->cur = ST_OPEN;
            }
            break;
        case ST_CLOSED:
        default:
            if (s->cur != ST_CLOSED) {
                s->using_post = 0;
                s->closed_count = 0;
                s->cur = ST_CLOSED;
                ctcss_reset(&s->fast);
                ctcss_reset(&s->slow);
            } else if (s->closed_count < s->recent_size) {
                s->closed_count++;
            } else if (s->closed_count == s->recent_size) {
                s->recent_open = 0;
                s->level_cache = 0.0f;
            }
            break;
    }
    s->tail = (s->tail + 1) % SQ_BUF;
    s->head = (s->head + 1) % SQ_BUF;
}

static void sq_avg(squelch_t* s, float* full, float* capped, float x) { /* src/squelch.cpp:501-514 */
    const float decay = 0.99f;
    const float fresh = (float)(1.0 - (double)decay);
    *full = *full * decay + x * fresh;
    if (*capped >= s->cap && x >= s->cap) {
        *capped = s->cap;
    } else {
        float v = *capped * decay + x * fresh;
        *capped = v < s->cap ? v : s->cap; /* std::min(moving_avg_cap_, v) = (v < cap) ? v : cap: a NaN (an unstable lowpass: bandwidth above WAVE_RATE) yields the cap, not the NaN */
    }
}

static void sq_raw(squelch_t* s, float x) { /* src/squelch.cpp:195-246 */
    sq_advance(s);
    s->sample_count++;
    if (s->sample_count % 16 == 0) { /* :477-490 noise floor follows min(level, floor) */
        const float decay = 0.97f;
        const float fresh = (float)(1.0 - (double)decay);
        float lo = s->noise_floor < s->pre_capped ? s->noise_floor : s->pre_capped; /* std::min(pre_filter_.capped_, noise_floor_) */
        s->noise_floor = s->noise_floor * decay + lo * fresh + 1e-6f;
        sq_recalc_cap(s);
        s->level_cache = 0.0f;
    }
    sq_avg(s, &s->pre_full, &s->pre_capped, x);
    s->buf[s->head] = s->pre_capped * s->pre_vs_post;
    if (s->cur == ST_OPEN && !sq_has_signal(s)) sq_request(s, ST_CLOSING);
    if (s->cur == ST_CLOSED && sq_has_signal(s)) sq_request(s, ST_OPENING);
    if (s->cur != ST_CLOSED && s->cur != ST_ABORT) {
        if (x >= sq_level(s)) {
            s->low_count = 0;
        } else if (++s->low_count >= s->low_abort) {
            sq_request(s, ST_ABORT);
        }
    }
}

static int sq_should_filter(squelch_t* s) { return (sq_has_pre(s) || s->cur != ST_CLOSED) && s->cur != ST_ABORT; } /* :136-138 */
static int sq_should_audio(const squelch_t* s) { return s->cur == ST_OPEN || s->cur == ST_CLOSING; }               /* :140-142 */
static int sq_first_open(const squelch_t* s) { return s->cur != ST_OPEN && s->next == ST_OPEN; }                   /* :144-146 */
static int sq_last_open(const squelch_t* s) {                                                                      /* :148-150 */
    return (s->cur == ST_CLOSING && s->next == ST_CLOSED) || (s->cur != ST_ABORT && s->next == ST_ABORT);
}
static int sq_tone(const squelch_t* s) {
    if (!s->slow.enabled) return 1;
    return s->slow.enough ? ctcss_has_tone(&s->slow) : ctcss_has_tone(&s->fast);
}
static int sq_is_open(const squelch_t* s) { /* :118-134 */
    if (s->cur == ST_OPEN || s->cur == ST_CLOSING) return sq_tone(s);
    return 0;
}

static void sq_filtered(squelch_t* s, float x) { /* src/squelch.cpp:248-276 */
    if (!sq_should_filter(s)) return;
    if (s->cur == ST_OPENING) {
        if (s->delay < SQ_BUF) return;
        if (s->delay == SQ_BUF) s->post_full = s->post_capped = s->buf[s->tail];
    }
    s->using_post = 1;
    sq_avg(s, &s->post_full, &s->post_capped, x);
    if (s->post_capped < s->buf[s->tail]) sq_request(s, ST_CLOSED);
}

static void sq_audio(squelch_t* s, float x) { /* src/squelch.cpp:278-295 */
    if (!s->slow.enabled) return;
    if (s->cur != ST_CLOSED) {
        ctcss_sample(&s->slow, x);
        if (!s->slow.enough) ctcss_sample(&s->fast, x);
    }
}

static void sq_set_ctcss(squelch_t* s, float freq, float rate) { /* src/squelch.cpp:105-116 */
    ctcss_init(&s->fast, freq, rate, (int)(rate * 0.05));
    ctcss_init(&s->slow, freq, rate, (int)(rate * 0.4));
}

/* ------------------------------------------------------------------------------------------------
 * Filters (reference: src/filters.cpp)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int enabled;
    float d0, d1, d2, x[3], y[3];
} notch_t;

static void notch_init(notch_t* f, float freq, float rate, float q) { /* src/filters.cpp:30-48 */
    memset(f, 0, sizeof(*f));
    if (freq <= 0.0f) return;
    f->enabled = 1;
    float wo = (float)(2 * M_PI * (double)(freq / rate));
    float e = 1 / (1 + tanf(wo / (q * 2)));
    float p = cosf(wo);
    f->d0 = e;
    f->d1 = 2 * e * p;
    f->d2 = (2 * e - 1);
}

static void notch_apply(notch_t* f, float* v) { /* src/filters.cpp:50-64 */
    if (!f->enabled) return;
    f->x[0] = f->x[1];
    f->x[1] = f->x[2];
    f->x[2] = *v;
    f->y[0] = f->y[1];
    f->y[1] = f->y[2];
    f->y[2] = f->d0 * f->x[2] - f->d1 * f->x[1] + f->d0 * f->x[0] + f->d1 * f->y[1] - f->d2 * f->y[0];
    *v = f->y[2];
}

typedef struct {
    int enabled;
    float yc0, yc1, gain;
    float xr[3], xi[3], yr[3], yi[3];
} lowpass_t;

static double complex bilinear(double complex p) { return (2.0 + p) / (2.0 - p); } /* src/filters.cpp:101-103 */

/* coefficients of (z - r0)(z - r1), lowest power first (src/filters.cpp:119-144) */
static void poly_from_roots(const double complex r[2], double complex c[3]) {
    c[0] = 1.0;
    c[1] = 0.0;
    c[2] = 0.0;
    for (int i = 0; i < 2; i++) {
        double complex nw = -r[i];
        for (int k = 2; k >= 1; k--) c[k] = (nw * c[k]) + c[k - 1];
        c[0] = nw * c[0];
    }
}

static double complex poly_eval(const double complex c[3], double complex z) { /* src/filters.cpp:110-117 */
    double complex sum = 0.0;
    for (int i = 2; i >= 0; i--) sum = (sum * z) + c[i];
    return sum;
}

static void lowpass_init(lowpass_t* f, float freq, float rate) { /* src/filters.cpp:70-99: 2nd-order Bessel, bilinear transform */
    memset(f, 0, sizeof(*f));
    if (freq <= 0.0f) return;
    f->enabled = 1;
    double raw_alpha = (double)freq / rate;
    double warped = tan(M_PI * raw_alpha) / M_PI;
    const double complex bessel = -1.10160133059e+00 + 6.36009824757e-01 * I;
    double complex zeros[2] = {-1.0, -1.0};
    double complex poles[2];
    poles[0] = bilinear(M_PI * 2 * warped * bessel);
    poles[1] = bilinear(M_PI * 2 * warped * conj(bessel));
    double complex top[3], bot[3];
    poly_from_roots(zeros, top);
    poly_from_roots(poles, bot);
    double complex g = poly_eval(top, 1.0) / poly_eval(bot, 1.0);
    f->gain = (float)hypot(cimag(g), creal(g));
    f->yc0 = (float)(-(creal(bot[0]) / creal(bot[2])));
    f->yc1 = (float)(-(creal(bot[1]) / creal(bot[2])));
}

static void lowpass_apply(lowpass_t* f, float* r, float* j) { /* src/filters.cpp:146-163 */
    if (!f->enabled) return;
    f->xr[0] = f->xr[1];
    f->xi[0] = f->xi[1];
    f->xr[1] = f->xr[2];
    f->xi[1] = f->xi[2];
    f->xr[2] = *r / f->gain;
    f->xi[2] = *j / f->gain;
    f->yr[0] = f->yr[1];
    f->yi[0] = f->yi[1];
    f->yr[1] = f->yr[2];
    f->yi[1] = f->yi[2];
    f->yr[2] = (f->xr[0] + f->xr[2]) + (2.0f * f->xr[1]) + (f->yc0 * f->yr[0]) + (f->yc1 * f->yr[1]);
    f->yi[2] = (f->xi[0] + f->xi[2]) + (2.0f * f->xi[1]) + (f->yc0 * f->yi[0]) + (f->yc1 * f->yi[1]);
    *r = f->yr[2];
    *j = f->yi[2];
}

/* ------------------------------------------------------------------------------------------------
 * small math helpers
 * ---------------------------------------------------------------------------------------------- */
static float g_sin_lut[257], g_cos_lut[257];
static int g_lut_ready = 0;

static void lut_init(void) { /* src/util.cpp:105-110 */
    if (g_lut_ready) return;
    for (uint32_t i = 0; i < 256; i++) sincosf((float)(2.0F * M_PI * (float)i / 256.0f), g_sin_lut + i, g_cos_lut + i);
    g_sin_lut[256] = g_sin_lut[0];
    g_cos_lut[256] = g_cos_lut[0];
    g_lut_ready = 1;
}

void orc_sincos_lut(uint32_t phi, float* s, float* c) { /* src/util.cpp:113-127: 24-bit phase, linear interpolation */
    lut_init();
    uint32_t idx = phi >> 16;
    float fract = (float)(phi & 0xffff) / 65536.0f;
    *s = g_sin_lut[idx] + (g_sin_lut[idx + 1] - g_sin_lut[idx]) * fract;
    *c = g_cos_lut[idx] + (g_cos_lut[idx + 1] - g_cos_lut[idx]) * fract;
}

float orc_dbfs_to_level(float dbfs, int fft_size) { /* src/util.cpp:169-176 */
    float offset = 7.54f + 10.0f * log10f((float)(fft_size / 2)) - 2.38f;
    return (float)(pow(10.0, (dbfs - offset) / 20.0f) * fft_size);
}

float orc_fast_atan2(float y, float x) { /* src/rtl_airband.cpp:147-166: first-order rational approximation per quadrant pair */
    const float pi4 = (float)M_PI_4, pi34 = (float)(3 * M_PI_4);
    if (x == 0.0f && y == 0.0f) return 0;
    float ay = y < 0.0f ? -y : y;
    float a;
    if (x >= 0.0f)
        a = pi4 - pi4 * (x - ay) / (x + ay);
    else
        a = pi34 - pi4 * (x + ay) / (ay - x);
    return y < 0.0f ? -a : a;
}

float orc_polar_disc_fast(float ar, float aj, float br, float bj) { /* src/rtl_airband.cpp:141-144,168-172: angle of a * conj(b), in turns/2 */
    float nbj = -bj;
    float cr = ar * br - aj * nbj;
    float cj = aj * br + ar * nbj;
    return (float)(orc_fast_atan2(cj, cr) * M_1_PI);
}

float orc_fm_quadri_demod(float ar, float aj, float br, float bj) { /* src/rtl_airband.cpp:174-176 */
    return (float)((br * aj - ar * bj) / (ar * ar + aj * aj + 1.0f) * M_1_PI);
}

float orc_window_coeff(int fft_size, int i) { /* src/rtl_airband.cpp:335-351: 7-term cosine sum, float literals widened to double */
    const double a0 = 0.27105140069342f, a1 = 0.43329793923448f, a2 = 0.21812299954311f, a3 = 0.06592544638803f, a4 = 0.01081174209837f,
                 a5 = 0.00077658482522f, a6 = 0.00001388721735f;
    const double d = (double)(fft_size - 1);
    double x = a0 - (a1 * cos((2.0 * M_PI * i) / d)) + (a2 * cos((4.0 * M_PI * i) / d)) - (a3 * cos((6.0 * M_PI * i) / d)) + (a4 * cos((8.0 * M_PI * i) / d)) -
               (a5 * cos((10.0 * M_PI * i) / d)) + (a6 * cos((12.0 * M_PI * i) / d));
    return (float)x;
}

/* ------------------------------------------------------------------------------------------------
 * data model
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int modulation, afc, needs_raw_iq, has_iq_outputs;
    int bin, base_bin;
    uint32_t dm_dphi, dm_phi;
    float alpha, pr, pj, prev_waveout;
    float agcavgfast, ampfactor;
    uint64_t active_counter;
    char axc;
    squelch_t sq;
    notch_t notch;
    lowpass_t lowpass;
    float *wavein, *waveout, *iq_in, *iq_out; /* WAVE_LEN, WAVE_LEN, 2*WAVE_LEN, 2*WAVE_LEN */
} chan_t;

typedef struct {
    int sample_rate, centerfreq, sfmt, bps_sample;
    float fullscale;
    int n_ch;
    chan_t* ch;
    int waveend;
    size_t hop_bytes;
    unsigned char* pend; /* unconsumed stream bytes */
    size_t pend_len, pend_cap;
    float* last_fft; /* full spectrum of the most recent hop (for AFC) */
} odev_t;

struct orc {
    int fft_size, fft_log, wave_rate, wave_batch, wave_len, fm_demod;
    int n_dev, total_ch;
    odev_t* dev;
    float* window;
    float lev_u8[256], lev_s8[256];
    fftwf_complex *fin, *fout;
    fftwf_plan plan;
};

int orc_wave_batch(const orc_t* o) { return o->wave_batch; }
int orc_total_channels(const orc_t* o) { return o->total_ch; }

static float tau_to_alpha(int wave_rate, int tau_us) { /* src/config.cpp:648,775 */
    return tau_us == 0 ? 0.0f : (float)exp(-1.0f / (wave_rate * 1e-6 * tau_us));
}

orc_t* orc_create(const airband_hip_config* cfg) {
    if (cfg->fft_size_log < 8 || cfg->fft_size_log > 13) return NULL;
    if (cfg->wave_rate != 8000 && cfg->wave_rate != 16000) return NULL;
    orc_t* o = (orc_t*)calloc(1, sizeof(*o));
    o->fft_log = cfg->fft_size_log;
    o->fft_size = 1 << cfg->fft_size_log;
    o->wave_rate = cfg->wave_rate;
    o->wave_batch = cfg->wave_rate / 8;             /* WAVE_BATCH, src/rtl_airband.h:73 */
    o->wave_len = 2 * o->wave_batch + AGC_EXTRA;    /* WAVE_LEN,  src/rtl_airband.h:75 */
    o->fm_demod = cfg->fm_demod;
    o->n_dev = cfg->device_count;
    o->dev = (odev_t*)calloc((size_t)o->n_dev, sizeof(odev_t));
    lut_init();
    for (int i = 0; i < 256; i++) o->lev_u8[i] = (i - 127.5f) / 127.5f; /* src/rtl_airband.cpp:319-321 */
    /* :322-324.  The reference's loop starts at -127: entry 128 (the byte -128) is never written there and holds whatever the stack did.
     * Any value refines that; this restatement -- and the library -- continue the table's own rule, -128 / 128 = -1.0 (the negative rail of
     * an 8-bit ADC, symmetric to +127 / 128).  Every comparison with the reference itself uses streams without that byte. */
    for (int i = -128; i < 128; i++) o->lev_s8[(uint8_t)i] = i / 128.0f;
    o->window = (float*)malloc(sizeof(float) * (size_t)o->fft_size);
    for (int i = 0; i < o->fft_size; i++) o->window[i] = orc_window_coeff(o->fft_size, i);
    o->fin = fftwf_alloc_complex((size_t)o->fft_size);
    o->fout = fftwf_alloc_complex((size_t)o->fft_size);
    o->plan = fftwf_plan_dft_1d(o->fft_size, o->fin, o->fout, FFTW_FORWARD, FFTW_MEASURE);
    const float global_alpha = (float)exp(-1.0f / (o->wave_rate * 2e-4)); /* src/rtl_airband.cpp:87 */
    const float rate = (float)o->wave_rate;
    for (int d = 0; d < o->n_dev; d++) {
        const airband_hip_device_cfg* dc = cfg->devices + d;
        odev_t* dev = o->dev + d;
        dev->sample_rate = dc->sample_rate;
        dev->centerfreq = dc->centerfreq;
        dev->sfmt = dc->sfmt;
        switch (dc->sfmt) { /* src/input-file.cpp:171-173, src/input-soapysdr.cpp:45-64 */
            case AIRBAND_SFMT_U8:
            case AIRBAND_SFMT_S8: dev->bps_sample = 1; dev->fullscale = (float)SCHAR_MAX - 0.5f; break;
            case AIRBAND_SFMT_S16: dev->bps_sample = 2; dev->fullscale = (float)SHRT_MAX - 0.5f; break;
            default: dev->bps_sample = 4; dev->fullscale = 1.0f; break;
        }
        if (dc->fullscale > 0) dev->fullscale = dc->fullscale;
        dev->hop_bytes = 2 * (size_t)dev->bps_sample * (size_t)round((double)dev->sample_rate / (double)o->wave_rate); /* src/rtl_airband.cpp:394 */
        dev->n_ch = dc->channel_count;
        dev->ch = (chan_t*)calloc((size_t)dev->n_ch, sizeof(chan_t));
        dev->last_fft = (float*)calloc(2 * (size_t)o->fft_size, sizeof(float));
        float dev_alpha = dc->tau_us >= 0 ? tau_to_alpha(o->wave_rate, dc->tau_us) : global_alpha;
        for (int j = 0; j < dev->n_ch; j++) {
            const airband_hip_channel_cfg* cc = dc->channels + j;
            chan_t* c = dev->ch + j;
            c->wavein = (float*)calloc((size_t)o->wave_len, sizeof(float));
            c->waveout = (float*)calloc((size_t)o->wave_len, sizeof(float));
            c->iq_in = (float*)calloc(2 * (size_t)o->wave_len, sizeof(float));
            c->iq_out = (float*)calloc(2 * (size_t)o->wave_len, sizeof(float));
            for (int k = 0; k < AGC_EXTRA; k++) { /* src/config.cpp:313-316 */
                c->wavein[k] = 20;
                c->waveout[k] = 0.5;
            }
            c->axc = ' ';
            c->pr = c->pj = 0;
            c->prev_waveout = 0.5;
            c->alpha = dev_alpha;
            c->afc = cc->afc & 0xff;
            c->modulation = cc->modulation;
            c->agcavgfast = 0.5f;
            sq_init(&c->sq);
            if (cc->squelch_threshold_dbfs < 0) /* src/config.cpp:452-471 */
                sq_set_level(&c->sq, orc_dbfs_to_level((float)cc->squelch_threshold_dbfs, o->fft_size));
            else
                sq_set_level(&c->sq, 0);
            if (cc->squelch_snr_threshold_db >= 0.0f) sq_set_snr(&c->sq, cc->squelch_snr_threshold_db); /* :473-515 */
            if (cc->notch_freq > 0) notch_init(&c->notch, cc->notch_freq, rate, cc->notch_q > 0 ? cc->notch_q : 10.0f); /* :516-564 */
            if (cc->ctcss_freq > 0) sq_set_ctcss(&c->sq, cc->ctcss_freq, rate);                                        /* :565-591 */
            if (cc->bandwidth_hz != 0) {                                                                               /* :592-619 */
                c->needs_raw_iq = 1;
                if (cc->bandwidth_hz > 0) lowpass_init(&c->lowpass, (float)cc->bandwidth_hz / 2, rate);
            }
            c->ampfactor = cc->ampfactor;
            if (cc->tau_us >= 0) c->alpha = tau_to_alpha(o->wave_rate, cc->tau_us);
            if (cc->has_iq_outputs) c->has_iq_outputs = c->needs_raw_iq = 1;
            /* bin index (src/config.cpp:666-667): note the INTEGER sample_rate / fft_size */
            c->bin = c->base_bin =
                (int)((size_t)ceil((cc->frequency + dev->sample_rate - dev->centerfreq) / (double)(dev->sample_rate / o->fft_size) - 1.0) % (size_t)o->fft_size);
            if (c->modulation == AIRBAND_MOD_NFM) c->needs_raw_iq = 1;
            if (c->needs_raw_iq) { /* derotation step (src/config.cpp:679-712) */
                double f = (double)(cc->frequency - dev->centerfreq);
                double dec = (double)dev->sample_rate / (double)o->wave_rate;
                double corr = (double)o->wave_rate / 2.0;
                corr *= (dec - round(dec));
                corr *= (double)(cc->frequency - dev->centerfreq) / ((double)dev->sample_rate / 2.0);
                f -= corr;
                f /= (double)o->wave_rate;
                f -= trunc(f);
                f *= 256.0 * 65536.0;
                c->dm_dphi = (uint32_t)((int)f);
                c->dm_phi = 0;
            }
            o->total_ch++;
        }
    }
    return o;
}

void orc_destroy(orc_t* o) {
    if (!o) return;
    for (int d = 0; d < o->n_dev; d++) {
        for (int j = 0; j < o->dev[d].n_ch; j++) {
            free(o->dev[d].ch[j].wavein);
            free(o->dev[d].ch[j].waveout);
            free(o->dev[d].ch[j].iq_in);
            free(o->dev[d].ch[j].iq_out);
        }
        free(o->dev[d].ch);
        free(o->dev[d].pend);
        free(o->dev[d].last_fft);
    }
    free(o->dev);
    free(o->window);
    fftwf_destroy_plan(o->plan);
    fftwf_free(o->fin);
    fftwf_free(o->fout);
    free(o);
}

/* ------------------------------------------------------------------------------------------------
 * stage 1: one hop = convert x window -> FFT -> per-channel bin (src/rtl_airband.cpp:402-492)
 * ---------------------------------------------------------------------------------------------- */
static void stage1_hop(orc_t* o, odev_t* dev, const unsigned char* p) {
    const int N = o->fft_size;
    if (dev->sfmt == AIRBAND_SFMT_S16) {
        const float scale = 1.0f / dev->fullscale;
        const short* s = (const short*)p;
        for (int i = 0; i < N; i++) {
            o->fin[i][0] = scale * (float)s[2 * i] * o->window[i];
            o->fin[i][1] = scale * (float)s[2 * i + 1] * o->window[i];
        }
    } else if (dev->sfmt == AIRBAND_SFMT_F32) {
        const float scale = 1.0f / dev->fullscale;
        const float* s = (const float*)p;
        for (int i = 0; i < N; i++) {
            o->fin[i][0] = scale * s[2 * i] * o->window[i];
            o->fin[i][1] = scale * s[2 * i + 1] * o->window[i];
        }
    } else {
        const float* lev = dev->sfmt == AIRBAND_SFMT_U8 ? o->lev_u8 : o->lev_s8;
        for (int i = 0; i < N; i++) {
            o->fin[i][0] = lev[p[2 * i]] * o->window[i];
            o->fin[i][1] = lev[p[2 * i + 1]] * o->window[i];
        }
    }
    fftwf_execute(o->plan);
    memcpy(dev->last_fft, o->fout, sizeof(float) * 2 * (size_t)N);
    for (int j = 0; j < dev->n_ch; j++) {
        chan_t* c = dev->ch + j;
        const float re = o->fout[c->bin][0], im = o->fout[c->bin][1];
        c->wavein[dev->waveend] = sqrtf(re * re + im * im);
        if (c->needs_raw_iq) {
            c->iq_in[2 * dev->waveend] = re;
            c->iq_in[2 * dev->waveend + 1] = im;
        }
    }
    dev->waveend += 1;
}

/* AFC (src/rtl_airband.cpp:180-251): on a squelch-open edge walk to a stronger neighbouring bin */
static float afc_power(const float* fft, int bin) { return fft[2 * bin] * fft[2 * bin] + fft[2 * bin + 1] * fft[2 * bin + 1]; }

static int afc_walk(const orc_t* o, const float* fft, int base, float base_value, int afc, int step) {
    float threshold = 0;
    int bin;
    for (bin = base;; bin += step) {
        if (step < 0) {
            if (bin < -step) break;
        } else if (bin + step >= o->fft_size) {
            break;
        }
        const float value = afc_power(fft, bin + step);
        if (value <= base_value) break;
        if (base == bin) {
            threshold = (value - base_value) / (float)afc;
        } else {
            if ((value - base_value) < threshold) break;
            threshold = (float)(threshold + threshold / 10.0);
        }
    }
    return bin;
}

static void afc_finalize(const orc_t* o, odev_t* dev, chan_t* c, char prev_axc) {
    if (c->afc == 0) return;
    if (c->axc != ' ' && prev_axc == ' ') {
        const int base = c->base_bin;
        const float base_value = afc_power(dev->last_fft, base);
        int bin = afc_walk(o, dev->last_fft, base, base_value, c->afc, -1);
        if (bin == base) bin = afc_walk(o, dev->last_fft, base, base_value, c->afc, 1);
        if (c->bin != bin) {
            c->bin = bin;
            if (bin > base)
                c->axc = '<'; /* AFC_UP, src/rtl_airband.h:99 */
            else if (bin < base)
                c->axc = '>'; /* AFC_DOWN */
        }
    } else if (c->axc == ' ' && prev_axc != ' ') {
        c->bin = c->base_bin;
    }
}

/* ------------------------------------------------------------------------------------------------
 * stage 2: the per-channel sequential loop over one batch (src/rtl_airband.cpp:494-648)
 * ---------------------------------------------------------------------------------------------- */
static void stage2_channel(orc_t* o, odev_t* dev, chan_t* c, uint8_t* trace) {
    const int B = o->wave_batch;
    squelch_t* sq = &c->sq;
    const char prev_axc = c->axc;
    c->axc = ' ';
    for (int j = AGC_EXTRA; j < B + AGC_EXTRA; j++) {
        float* re = &c->iq_in[2 * (j - AGC_EXTRA)]; /* I/Q lags the magnitude by AGC_EXTRA hops */
        float* im = re + 1;
        sq_raw(sq, c->wavein[j]);
        if (sq_should_filter(sq) && c->needs_raw_iq) {
            float s, co;
            orc_sincos_lut(c->dm_phi, &s, &co);
            /* derotate: (re + j im) * (cos - j sin) */
            float ns = -s;
            float tr = *re * co - *im * ns;
            float ti = *im * co + *re * ns;
            c->dm_phi = (c->dm_phi + c->dm_dphi) & 0xffffff;
            lowpass_apply(&c->lowpass, &tr, &ti);
            *re = tr;
            *im = ti;
            c->wavein[j] = (float)sqrt((double)(tr * tr + ti * ti));
            if (c->lowpass.enabled) sq_filtered(sq, c->wavein[j]);
        }
        if (c->modulation == AIRBAND_MOD_AM) {
            if (sq_first_open(sq)) { /* bootstrap the AGC from the preceding AGC_EXTRA magnitudes */
                for (int k = j - AGC_EXTRA; k < j; k++)
                    if (c->wavein[k] >= sq_level(sq)) c->agcavgfast = c->agcavgfast * 0.9f + c->wavein[k] * 0.1f;
            } else if (sq_last_open(sq)) { /* fade out what was already written */
                for (int k = j - AGC_EXTRA + 1; k < j; k++) c->waveout[k] = c->waveout[k - 1] * 0.94f;
            }
        }
        float out = c->waveout[j];
        if (sq_should_audio(sq)) {
            if (c->modulation == AIRBAND_MOD_AM) {
                if (c->wavein[j] > sq_level(sq)) c->agcavgfast = c->agcavgfast * 0.995f + c->wavein[j] * 0.005f;
                out = (c->wavein[j - AGC_EXTRA] - c->agcavgfast) / (c->agcavgfast * 1.5f);
                if (fabsf(out) > 0.8f) {
                    out *= 0.85f;
                    c->agcavgfast *= 1.15f;
                }
            } else if (c->modulation == AIRBAND_MOD_NFM) {
                if (o->fm_demod == AIRBAND_FM_FAST_ATAN2)
                    out = orc_polar_disc_fast(*re, *im, c->pr, c->pj);
                else
                    out = orc_fm_quadri_demod(*re, *im, c->pr, c->pj);
                c->pr = *re;
                c->pj = *im;
                c->agcavgfast = c->agcavgfast * 0.995f + out * 0.005f; /* DC block */
                out -= c->agcavgfast;
                out = out * (1.0f - c->alpha) + c->prev_waveout * c->alpha; /* de-emphasis */
                c->prev_waveout = out;
            }
            sq_audio(sq, out);
        }
        const int open = sq_is_open(sq);
        if (open) {
            notch_apply(&c->notch, &out);
            out *= c->ampfactor;
            if (isnan(out))
                out = 0.0f;
            else if (out > 1.0)
                out = 1.0f;
            else if (out < -1.0)
                out = -1.0f;
            c->axc = '*';
            if (c->has_iq_outputs) {
                c->iq_out[2 * (j - AGC_EXTRA)] = *re;
                c->iq_out[2 * (j - AGC_EXTRA) + 1] = *im;
            }
        } else {
            out = 0;
            if (c->has_iq_outputs) {
                c->iq_out[2 * (j - AGC_EXTRA)] = 0;
                c->iq_out[2 * (j - AGC_EXTRA) + 1] = 0;
            }
        }
        c->waveout[j] = out;
        if (trace) trace[j - AGC_EXTRA] = (uint8_t)((sq->cur & 7) | (open ? 8 : 0) | (sq_should_audio(sq) ? 16 : 0) | ((sq->slow.enabled && sq_tone(sq)) ? 32 : 0));
    }
    /* carry the last AGC_EXTRA magnitudes / I/Q into the next batch (src/rtl_airband.cpp:621-624) */
    memmove(c->wavein, c->wavein + B, (size_t)(dev->waveend - B) * sizeof(float));
    if (c->needs_raw_iq) memmove(c->iq_in, c->iq_in + 2 * B, (size_t)(dev->waveend - B) * sizeof(float) * 2);
    afc_finalize(o, dev, c, prev_axc);
    if (c->axc != ' ') c->active_counter++;
}

/* consumer side (src/output.cpp:917-922): hand out waveout[0..B), then move the tail to the front */
static void emit_batch(orc_t* o, odev_t* dev, float* waveout, float* iq_out, char* axc) {
    const int B = o->wave_batch;
    for (int j = 0; j < dev->n_ch; j++) {
        chan_t* c = dev->ch + j;
        if (waveout) memcpy(waveout + (size_t)j * B, c->waveout, sizeof(float) * (size_t)B);
        if (iq_out) memcpy(iq_out + (size_t)j * 2 * B, c->iq_out, sizeof(float) * 2 * (size_t)B);
        if (axc) axc[j] = c->axc;
        memcpy(c->waveout, c->waveout + B, AGC_EXTRA * sizeof(float));
    }
}

static void grab_raw(orc_t* o, odev_t* dev, float* raw_wavein, float* raw_iq) {
    const int B = o->wave_batch;
    for (int j = 0; j < dev->n_ch; j++) {
        chan_t* c = dev->ch + j;
        if (raw_wavein) memcpy(raw_wavein + (size_t)j * B, c->wavein + AGC_EXTRA, sizeof(float) * (size_t)B);
        if (raw_iq) {
            if (c->needs_raw_iq)
                memcpy(raw_iq + (size_t)j * 2 * B, c->iq_in + 2 * AGC_EXTRA, sizeof(float) * 2 * (size_t)B);
            else
                memset(raw_iq + (size_t)j * 2 * B, 0, sizeof(float) * 2 * (size_t)B);
        }
    }
}

int orc_run_device(orc_t* o, int d, const void* iq, size_t nbytes, int max_batches, float* waveout, float* iq_out, char* axc, uint8_t* trace, float* raw_wavein,
                   float* raw_iq) {
    odev_t* dev = o->dev + d;
    const int B = o->wave_batch, C = dev->n_ch;
    if (dev->pend_len + nbytes > dev->pend_cap) {
        dev->pend_cap = dev->pend_len + nbytes;
        dev->pend = (unsigned char*)realloc(dev->pend, dev->pend_cap);
    }
    memcpy(dev->pend + dev->pend_len, iq, nbytes);
    dev->pend_len += nbytes;
    const size_t need = dev->hop_bytes + (size_t)o->fft_size * (size_t)dev->bps_sample * 2; /* src/rtl_airband.cpp:395 */
    size_t pos = 0;
    int nb = 0;
    while (dev->pend_len - pos >= need && nb < max_batches) {
        stage1_hop(o, dev, dev->pend + pos);
        pos += dev->hop_bytes;
        if (dev->waveend >= B + AGC_EXTRA) {
            grab_raw(o, dev, raw_wavein ? raw_wavein + (size_t)nb * C * B : NULL, raw_iq ? raw_iq + (size_t)nb * C * 2 * B : NULL);
            for (int j = 0; j < C; j++) stage2_channel(o, dev, dev->ch + j, trace ? trace + ((size_t)nb * C + j) * B : NULL);
            dev->waveend -= B;
            emit_batch(o, dev, waveout ? waveout + (size_t)nb * C * B : NULL, iq_out ? iq_out + (size_t)nb * C * 2 * B : NULL, axc ? axc + (size_t)nb * C : NULL);
            nb++;
        }
    }
    memmove(dev->pend, dev->pend + pos, dev->pend_len - pos);
    dev->pend_len -= pos;
    return nb;
}

/* One batch straight from a caller-held span, the way airband_hip_process_device() is fed: hop h reads the fft_size samples at
 * span + h * hop_bytes; the first call of a stream consumes WAVE_BATCH + AGC_EXTRA hops (waveend starts at 0,
 * src/config.cpp:805), every later call WAVE_BATCH.  The span may come from anywhere (bench.py cycles through a few resident
 * batches), so unlike orc_run_device nothing is buffered.  Returns 1 (one batch produced). */
int orc_run_span(orc_t* o, int d, const void* span, float* waveout, float* iq_out, char* axc, uint8_t* trace, float* raw_wavein, float* raw_iq) {
    odev_t* dev = o->dev + d;
    const int B = o->wave_batch, C = dev->n_ch;
    const unsigned char* p = (const unsigned char*)span;
    while (dev->waveend < B + AGC_EXTRA) {
        stage1_hop(o, dev, p);
        p += dev->hop_bytes;
    }
    grab_raw(o, dev, raw_wavein, raw_iq);
    for (int j = 0; j < C; j++) stage2_channel(o, dev, dev->ch + j, trace ? trace + (size_t)j * B : NULL);
    dev->waveend -= B;
    emit_batch(o, dev, waveout, iq_out, axc);
    return 1;
}

int orc_run_bins(orc_t* o, int d, const float* wavein, const float* iq, float* waveout, float* iq_out, char* axc, uint8_t* trace) {
    odev_t* dev = o->dev + d;
    const int B = o->wave_batch;
    /* the very first batch of a stream also owns the AGC_EXTRA lead-in hops; callers of this entry point
     * provide B hops per call, the lead-in keeps its config-time prefill (src/config.cpp:313-316) */
    if (dev->waveend == 0) dev->waveend = AGC_EXTRA;
    for (int j = 0; j < dev->n_ch; j++) {
        chan_t* c = dev->ch + j;
        memcpy(c->wavein + AGC_EXTRA, wavein + (size_t)j * B, sizeof(float) * (size_t)B);
        if (c->needs_raw_iq) memcpy(c->iq_in + 2 * AGC_EXTRA, iq + (size_t)j * 2 * B, sizeof(float) * 2 * (size_t)B);
    }
    dev->waveend += B;
    for (int j = 0; j < dev->n_ch; j++) stage2_channel(o, dev, dev->ch + j, trace ? trace + (size_t)j * B : NULL);
    dev->waveend -= B;
    emit_batch(o, dev, waveout, iq_out, axc);
    return 1;
}

int orc_channel_stats(orc_t* o, int d, int j, airband_hip_channel_stats* out) {
    chan_t* c = o->dev[d].ch + j;
    out->noise_level = c->sq.noise_floor;
    out->signal_level = c->sq.pre_full;
    out->squelch_level = sq_level(&c->sq);
    out->agcavgfast = c->agcavgfast;
    out->open_count = c->sq.open_count;
    out->flappy_count = c->sq.flappy_count;
    out->ctcss_count = c->sq.slow.found;
    out->no_ctcss_count = c->sq.slow.not_found;
    out->active_counter = c->active_counter;
    out->bin = c->bin;
    out->squelch_state = c->sq.cur;
    /* Squelch::signal_outside_filter (src/squelch.cpp:152-154) */
    out->signal_outside_filter = (c->sq.using_post && c->sq.pre_capped >= sq_level(&c->sq) && !(c->sq.post_capped >= c->sq.buf[c->sq.tail])) ? 1 : 0;
    out->reserved = 0;
    return 0;
}

int orc_channel_constants(orc_t* o, int d, int j, double* v) {
    chan_t* c = o->dev[d].ch + j;
    v[0] = c->bin;
    v[1] = c->dm_dphi;
    v[2] = c->alpha;
    v[3] = c->notch.d0;
    v[4] = c->notch.d1;
    v[5] = c->notch.d2;
    v[6] = c->lowpass.gain;
    v[7] = c->lowpass.yc0;
    v[8] = c->lowpass.yc1;
    v[9] = c->sq.normal_ratio;
    v[10] = c->sq.manual ? c->sq.manual_level : -1.0;
    v[11] = c->sq.fast.enabled ? c->sq.fast.n_tones : 0;
    v[12] = c->sq.slow.enabled ? c->sq.slow.n_tones : 0;
    v[13] = c->needs_raw_iq;
    v[14] = c->sq.fast.enabled ? c->sq.fast.window : 0;
    v[15] = c->sq.slow.enabled ? c->sq.slow.window : 0;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * stand-alone drivers (unit-level pinning against oracle/_ref and the reference's own unit tests)
 * ---------------------------------------------------------------------------------------------- */
void orc_notch_run(float freq, float rate, float q, float* x, int n) {
    notch_t f;
    notch_init(&f, freq, rate, q);
    for (int i = 0; i < n; i++) notch_apply(&f, x + i);
}

void orc_lowpass_run(float freq, float rate, float* re, float* im, int n) {
    lowpass_t f;
    lowpass_init(&f, freq, rate);
    for (int i = 0; i < n; i++) lowpass_apply(&f, re + i, im + i);
}

void orc_ctcss_run(float ctcss_freq, float sample_rate, int window, const float* x, int n, unsigned char* has_tone, uint64_t* counts2) {
    ctcss_t c;
    ctcss_init(&c, ctcss_freq, sample_rate, window);
    for (int i = 0; i < n; i++) {
        ctcss_sample(&c, x[i]);
        if (has_tone) has_tone[i] = (unsigned char)((ctcss_has_tone(&c) ? 1 : 0) | (c.enough ? 2 : 0));
    }
    counts2[0] = c.found;
    counts2[1] = c.not_found;
}

void* orc_squelch_new(float snr_db, int manual_dbfs, float ctcss_freq, int wave_rate, int fft_size) {
    squelch_t* s = (squelch_t*)malloc(sizeof(*s));
    sq_init(s);
    if (manual_dbfs < 0) sq_set_level(s, orc_dbfs_to_level((float)manual_dbfs, fft_size));
    if (snr_db >= 0) sq_set_snr(s, snr_db);
    if (ctcss_freq > 0) sq_set_ctcss(s, ctcss_freq, (float)wave_rate);
    return s;
}

static unsigned char sq_flags(squelch_t* s) {
    return (unsigned char)((sq_is_open(s) ? 1 : 0) | (sq_should_audio(s) ? 2 : 0) | (sq_should_filter(s) ? 4 : 0) | (sq_first_open(s) ? 8 : 0) | (sq_last_open(s) ? 16 : 0));
}

void orc_squelch_raw(void* p, const float* x, int n, unsigned char* flags, float* noise, float* level) {
    squelch_t* s = (squelch_t*)p;
    for (int i = 0; i < n; i++) {
        sq_raw(s, x[i]);
        if (flags) flags[i] = sq_flags(s);
        if (noise) noise[i] = s->noise_floor;
        if (level) level[i] = sq_level(s);
    }
}

/* raw + filtered path in the order of the demod loop (src/rtl_airband.cpp:507,510,526): process_raw_sample, then, when
 * should_filter_sample(), process_filtered_sample of the lowpass-filtered magnitude */
void orc_squelch_raw_filtered(void* p, const float* raw, const float* filtered, int n, unsigned char* flags, float* noise, float* level) {
    squelch_t* s = (squelch_t*)p;
    for (int i = 0; i < n; i++) {
        sq_raw(s, raw[i]);
        if (sq_should_filter(s)) sq_filtered(s, filtered[i]);
        if (flags) flags[i] = sq_flags(s);
        if (noise) noise[i] = s->noise_floor;
        if (level) level[i] = sq_level(s);
    }
}

void orc_squelch_raw_audio(void* p, const float* raw, const float* audio, int n, unsigned char* flags) {
    squelch_t* s = (squelch_t*)p;
    for (int i = 0; i < n; i++) {
        sq_raw(s, raw[i]);
        if (sq_should_audio(s)) sq_audio(s, audio[i]);
        if (flags) flags[i] = (unsigned char)((sq_is_open(s) ? 1 : 0) | (sq_should_audio(s) ? 2 : 0));
    }
}

void orc_squelch_audio_raw(void* p, const float* raw, const float* audio, int n, unsigned char* flags) { /* order of src/test_squelch.cpp:186-189 */
    squelch_t* s = (squelch_t*)p;
    for (int i = 0; i < n; i++) {
        sq_audio(s, audio[i]);
        sq_raw(s, raw[i]);
        if (flags) flags[i] = (unsigned char)((sq_is_open(s) ? 1 : 0) | (sq_should_audio(s) ? 2 : 0));
    }
}

void orc_squelch_counts(void* p, uint64_t* out4) {
    squelch_t* s = (squelch_t*)p;
    out4[0] = s->open_count;
    out4[1] = s->flappy_count;
    out4[2] = s->slow.found;
    out4[3] = s->slow.not_found;
}

void orc_squelch_free(void* p) { free(p); }

/* mixer sum (src/mixer.cpp:133-140 mix_waveforms, :201-214 per-input accumulation, :82-83 ampl/ampr).
 * Deterministic restatement: every connected input is "ready" every batch, inputs are added in
 * connection order. */
void orc_mix(const airband_hip_mixer_input* in, int n_in, const int* chan_base, const float* waveout, const char* axc, int B, int n_mixers, float* out_l, float* out_r,
             uint8_t* sig) {
    memset(out_l, 0, sizeof(float) * (size_t)n_mixers * (size_t)B);
    memset(out_r, 0, sizeof(float) * (size_t)n_mixers * (size_t)B);
    memset(sig, 0, (size_t)n_mixers);
    uint8_t* stereo = (uint8_t*)calloc((size_t)n_mixers, 1);
    for (int i = 0; i < n_in; i++)
        if (in[i].balance != 0.0f) stereo[in[i].mixer] = 1;
    for (int i = 0; i < n_in; i++) {
        const int ch = chan_base[in[i].device] + in[i].channel;
        if (axc[ch] == ' ') continue; /* has_signal false: input contributes nothing (src/mixer.cpp:119-122,203) */
        const float ampl = fminf(1.0f, 1.0f - in[i].balance), ampr = fminf(1.0f, 1.0f + in[i].balance);
        const float ml = in[i].ampfactor * ampl, mr = in[i].ampfactor * ampr;
        const float* w = waveout + (size_t)ch * B;
        float* l = out_l + (size_t)in[i].mixer * B;
        float* r = out_r + (size_t)in[i].mixer * B;
        if (ml != 0.0f)
            for (int s = 0; s < B; s++) l[s] += w[s] * ml;
        if (stereo[in[i].mixer] && mr != 0.0f)
            for (int s = 0; s < B; s++) r[s] += w[s] * mr;
        sig[in[i].mixer] = 1;
    }
    free(stereo);
}
