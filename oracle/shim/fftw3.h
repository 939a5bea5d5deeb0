/* oracle/shim/fftw3.h -- TEST INFRASTRUCTURE ONLY.
 * FFTW3 (libfftw3f, un-pinned system package of the reference: src/CMakeLists.txt:231-238) is not
 * installed in this image and cannot be fetched.  The three entry points the reference's hot path
 * uses (src/rtl_airband.cpp:262-264 and :460) are declared here and implemented by
 * oracle/oracle_fft.c: a float64 radix-2 transform rounded to float on output -- the same
 * mathematical transform (forward, unnormalised, sign -1).  Every "vs FFTW" statement in this
 * repository carries that caveat. */
#ifndef ORACLE_SHIM_FFTW3_H
#define ORACLE_SHIM_FFTW3_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef float fftwf_complex[2];
struct oracle_fft_plan;
typedef struct oracle_fft_plan* fftwf_plan;
#define FFTW_FORWARD (-1)
#define FFTW_MEASURE (0U)
fftwf_complex* fftwf_alloc_complex(size_t n);
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex* in, fftwf_complex* out, int sign, unsigned flags);
void fftwf_execute(const fftwf_plan p);
void fftwf_destroy_plan(fftwf_plan p);
void fftwf_free(void* p);
#ifdef __cplusplus
}
#endif
#endif
