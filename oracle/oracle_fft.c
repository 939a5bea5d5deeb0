/* oracle/oracle_fft.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU stand-in for the three FFTW3 entry points the reference's hot path calls
 * (reference: src/rtl_airband.cpp:262-264 plan creation, :460 fftwf_execute).  FFTW3 itself is an
 * un-vendored, un-pinned system dependency of the reference (src/CMakeLists.txt:231-238) and is
 * absent from this image, so its published algorithm -- the forward, unnormalised DFT
 *      X[k] = sum_n x[n] * exp(-2*pi*i*k*n/N)
 * -- is restated here as an iterative radix-2 decimation-in-time transform evaluated in float64
 * and rounded to float once on output.  Both the in-place-compiled reference (oracle/_ref) and the
 * C restatement (oracle/airband_oracle.c) call this same routine, so the two agree bit-for-bit.
 */
#define _GNU_SOURCE 1
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "shim/fftw3.h"

struct oracle_fft_plan {
    int n, log2n;
    fftwf_complex* in;
    fftwf_complex* out;
    double* tw_re; /* n/2 twiddles */
    double* tw_im;
    int* rev;      /* bit-reversal permutation */
    double* wr;    /* work arrays */
    double* wi;
};

fftwf_complex* fftwf_alloc_complex(size_t n) {
    void* p = NULL;
    if (posix_memalign(&p, 64, n * sizeof(fftwf_complex)) != 0) return NULL;
    memset(p, 0, n * sizeof(fftwf_complex));
    return (fftwf_complex*)p;
}

void fftwf_free(void* p) { free(p); }

fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex* in, fftwf_complex* out, int sign, unsigned flags) {
    (void)flags;
    if (sign != FFTW_FORWARD || n < 2 || (n & (n - 1)) != 0) return NULL;
    struct oracle_fft_plan* p = (struct oracle_fft_plan*)calloc(1, sizeof(*p));
    p->n = n;
    p->in = in;
    p->out = out;
    while ((1 << p->log2n) < n) p->log2n++;
    p->tw_re = (double*)malloc(sizeof(double) * (size_t)(n / 2));
    p->tw_im = (double*)malloc(sizeof(double) * (size_t)(n / 2));
    p->rev = (int*)malloc(sizeof(int) * (size_t)n);
    p->wr = (double*)malloc(sizeof(double) * (size_t)n);
    p->wi = (double*)malloc(sizeof(double) * (size_t)n);
    for (int k = 0; k < n / 2; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        p->tw_re[k] = cos(a);
        p->tw_im[k] = sin(a);
    }
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int b = 0; b < p->log2n; b++)
            if (i & (1 << b)) r |= 1 << (p->log2n - 1 - b);
        p->rev[i] = r;
    }
    return p;
}

void fftwf_destroy_plan(fftwf_plan p) {
    if (!p) return;
    free(p->tw_re);
    free(p->tw_im);
    free(p->rev);
    free(p->wr);
    free(p->wi);
    free(p);
}

void fftwf_execute(const fftwf_plan p) {
    const int n = p->n;
    double* wr = p->wr;
    double* wi = p->wi;
    for (int i = 0; i < n; i++) {
        wr[p->rev[i]] = (double)p->in[i][0];
        wi[p->rev[i]] = (double)p->in[i][1];
    }
    for (int half = 1; half < n; half <<= 1) {
        const int step = n / (2 * half);
        for (int base = 0; base < n; base += 2 * half) {
            for (int j = 0; j < half; j++) {
                const double tr = p->tw_re[j * step], ti = p->tw_im[j * step];
                const int a = base + j, b = a + half;
                const double xr = wr[b] * tr - wi[b] * ti;
                const double xi = wr[b] * ti + wi[b] * tr;
                wr[b] = wr[a] - xr;
                wi[b] = wi[a] - xi;
                wr[a] += xr;
                wi[a] += xi;
            }
        }
    }
    for (int i = 0; i < n; i++) {
        p->out[i][0] = (float)wr[i];
        p->out[i][1] = (float)wi[i];
    }
}
