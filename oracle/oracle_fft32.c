/* oracle/oracle_fft32.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Single-precision stand-in for the three FFTW3 entry points of the reference's hot path (src/rtl_airband.cpp:262-264
 * fftwf_plan_dft_1d, :460 fftwf_execute), used ONLY by the throughput builds of oracle/_ref (the "_fast32" libraries that
 * bench.py's cpu_baseline times).  Parity work keeps oracle_fft.c (float64 radix-2, rounded once): that one defines the
 * numbers, this one defines a fairer clock -- FFTW computes in float with SIMD codelets, so timing the reference behind a
 * float64 scalar transform understates "the FFTW path".
 *
 * Algorithm: Stockham autosort, radix 4 (one trailing radix-2 pass when log2 n is odd), split re/im work arrays so that the
 * inner loops are unit-stride and auto-vectorise under -O3 -march=native -ffast-math; twiddles precomputed per pass.
 * Same transform as FFTW_FORWARD: X[k] = sum_n x[n] exp(-2 pi i k n / N), unnormalised.
 */
#define _GNU_SOURCE 1
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "shim/fftw3.h"

#define MAX_PASSES 8

struct oracle_fft_plan {
    int n;
    fftwf_complex* in;
    fftwf_complex* out;
    int n_pass;
    int radix[MAX_PASSES];
    float* tw[MAX_PASSES]; /* per pass: [3][m/4] (radix 4) or [1][m/2] (radix 2) pairs (re, im) as two arrays */
    float *ar, *ai, *br, *bi;
};

fftwf_complex* fftwf_alloc_complex(size_t n) {
    void* p = NULL;
    if (posix_memalign(&p, 64, n * sizeof(fftwf_complex)) != 0) return NULL;
    memset(p, 0, n * sizeof(fftwf_complex));
    return (fftwf_complex*)p;
}

void fftwf_free(void* p) { free(p); }

static float* falloc(size_t n) {
    void* p = NULL;
    if (posix_memalign(&p, 64, n * sizeof(float)) != 0) return NULL;
    return (float*)p;
}

fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex* in, fftwf_complex* out, int sign, unsigned flags) {
    (void)flags;
    if (sign != FFTW_FORWARD || n < 4 || (n & (n - 1)) != 0) return NULL;
    struct oracle_fft_plan* p = (struct oracle_fft_plan*)calloc(1, sizeof(*p));
    p->n = n;
    p->in = in;
    p->out = out;
    p->ar = falloc((size_t)n);
    p->ai = falloc((size_t)n);
    p->br = falloc((size_t)n);
    p->bi = falloc((size_t)n);
    int m = n;
    while (m > 1) {
        const int r = (m % 4 == 0) ? 4 : 2;
        const int k = p->n_pass++;
        p->radix[k] = r;
        const int q = m / r;
        p->tw[k] = falloc((size_t)2 * (size_t)(r - 1) * (size_t)q);
        for (int t = 1; t < r; t++)
            for (int j = 0; j < q; j++) {
                const double a = -2.0 * M_PI * (double)t * (double)j / (double)m;
                p->tw[k][((size_t)(t - 1) * 2 + 0) * (size_t)q + (size_t)j] = (float)cos(a);
                p->tw[k][((size_t)(t - 1) * 2 + 1) * (size_t)q + (size_t)j] = (float)sin(a);
            }
        m = q;
    }
    return p;
}

void fftwf_destroy_plan(fftwf_plan p) {
    if (!p) return;
    for (int k = 0; k < p->n_pass; k++) free(p->tw[k]);
    free(p->ar);
    free(p->ai);
    free(p->br);
    free(p->bi);
    free(p);
}

/* One Stockham pass of radix 4: sub-transform length m, stride s (n = m * s).
 * x[q + s*(j + t*m/4)] -> y[q + s*(4j + t)] */
static void pass4(int m, int s, const float* restrict tw, const float* restrict xr, const float* restrict xi, float* restrict yr, float* restrict yi) {
    const int q4 = m / 4;
    const float *w1r = tw, *w1i = tw + q4, *w2r = tw + 2 * q4, *w2i = tw + 3 * q4, *w3r = tw + 4 * q4, *w3i = tw + 5 * q4;
    if (s == 1) { /* first pass: vectorise over the twiddle index */
        for (int j = 0; j < q4; j++) {
            const float a_r = xr[j], a_i = xi[j], b_r = xr[j + q4], b_i = xi[j + q4];
            const float c_r = xr[j + 2 * q4], c_i = xi[j + 2 * q4], d_r = xr[j + 3 * q4], d_i = xi[j + 3 * q4];
            const float apc_r = a_r + c_r, apc_i = a_i + c_i, amc_r = a_r - c_r, amc_i = a_i - c_i;
            const float bpd_r = b_r + d_r, bpd_i = b_i + d_i, bmd_r = b_r - d_r, bmd_i = b_i - d_i;
            /* -j * (b - d) = (bmd_i, -bmd_r) */
            const float t1r = amc_r + bmd_i, t1i = amc_i - bmd_r;
            const float t2r = apc_r - bpd_r, t2i = apc_i - bpd_i;
            const float t3r = amc_r - bmd_i, t3i = amc_i + bmd_r;
            yr[4 * j] = apc_r + bpd_r;
            yi[4 * j] = apc_i + bpd_i;
            yr[4 * j + 1] = t1r * w1r[j] - t1i * w1i[j];
            yi[4 * j + 1] = t1r * w1i[j] + t1i * w1r[j];
            yr[4 * j + 2] = t2r * w2r[j] - t2i * w2i[j];
            yi[4 * j + 2] = t2r * w2i[j] + t2i * w2r[j];
            yr[4 * j + 3] = t3r * w3r[j] - t3i * w3i[j];
            yi[4 * j + 3] = t3r * w3i[j] + t3i * w3r[j];
        }
        return;
    }
    for (int j = 0; j < q4; j++) {
        const float u1r = w1r[j], u1i = w1i[j], u2r = w2r[j], u2i = w2i[j], u3r = w3r[j], u3i = w3i[j];
        const float* pr = xr + (size_t)s * j;
        const float* pi = xi + (size_t)s * j;
        float* o_r = yr + (size_t)s * 4 * j;
        float* o_i = yi + (size_t)s * 4 * j;
        const size_t h = (size_t)s * q4;
        for (int q = 0; q < s; q++) { /* unit stride on both sides */
            const float a_r = pr[q], a_i = pi[q], b_r = pr[q + h], b_i = pi[q + h];
            const float c_r = pr[q + 2 * h], c_i = pi[q + 2 * h], d_r = pr[q + 3 * h], d_i = pi[q + 3 * h];
            const float apc_r = a_r + c_r, apc_i = a_i + c_i, amc_r = a_r - c_r, amc_i = a_i - c_i;
            const float bpd_r = b_r + d_r, bpd_i = b_i + d_i, bmd_r = b_r - d_r, bmd_i = b_i - d_i;
            const float t1r = amc_r + bmd_i, t1i = amc_i - bmd_r;
            const float t2r = apc_r - bpd_r, t2i = apc_i - bpd_i;
            const float t3r = amc_r - bmd_i, t3i = amc_i + bmd_r;
            o_r[q] = apc_r + bpd_r;
            o_i[q] = apc_i + bpd_i;
            o_r[q + s] = t1r * u1r - t1i * u1i;
            o_i[q + s] = t1r * u1i + t1i * u1r;
            o_r[q + 2 * s] = t2r * u2r - t2i * u2i;
            o_i[q + 2 * s] = t2r * u2i + t2i * u2r;
            o_r[q + 3 * s] = t3r * u3r - t3i * u3i;
            o_i[q + 3 * s] = t3r * u3i + t3i * u3r;
        }
    }
}

static void pass2(int m, int s, const float* restrict tw, const float* restrict xr, const float* restrict xi, float* restrict yr, float* restrict yi) {
    const int q2 = m / 2;
    const float *wr = tw, *wi = tw + q2;
    for (int j = 0; j < q2; j++) {
        const float ur = wr[j], ui = wi[j];
        const float* pr = xr + (size_t)s * j;
        const float* pi = xi + (size_t)s * j;
        float* o_r = yr + (size_t)s * 2 * j;
        float* o_i = yi + (size_t)s * 2 * j;
        const size_t h = (size_t)s * q2;
        for (int q = 0; q < s; q++) {
            const float a_r = pr[q], a_i = pi[q], b_r = pr[q + h], b_i = pi[q + h];
            const float dr = a_r - b_r, di = a_i - b_i;
            o_r[q] = a_r + b_r;
            o_i[q] = a_i + b_i;
            o_r[q + s] = dr * ur - di * ui;
            o_i[q + s] = dr * ui + di * ur;
        }
    }
}

void fftwf_execute(const fftwf_plan p) {
    const int n = p->n;
    float *xr = p->ar, *xi = p->ai, *yr = p->br, *yi = p->bi;
    for (int i = 0; i < n; i++) {
        xr[i] = p->in[i][0];
        xi[i] = p->in[i][1];
    }
    int m = n, s = 1;
    for (int k = 0; k < p->n_pass; k++) {
        if (p->radix[k] == 4)
            pass4(m, s, p->tw[k], xr, xi, yr, yi);
        else
            pass2(m, s, p->tw[k], xr, xi, yr, yi);
        s *= p->radix[k];
        m /= p->radix[k];
        float* t = xr; xr = yr; yr = t;
        t = xi; xi = yi; yi = t;
    }
    for (int i = 0; i < n; i++) {
        p->out[i][0] = xr[i];
        p->out[i][1] = xi[i];
    }
}
