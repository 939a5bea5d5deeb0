/* oracle/airband_oracle.h -- TEST INFRASTRUCTURE ONLY.
 * CPU restatement of the reference's demodulate() hot path (see airband_oracle.c).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker. */
#ifndef AIRBAND_ORACLE_H
#define AIRBAND_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#include "airband_hip.h" /* configuration structs are shared with the product ABI */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc orc_t;

orc_t* orc_create(const airband_hip_config* cfg);
void orc_destroy(orc_t* o);
int orc_wave_batch(const orc_t* o);
int orc_total_channels(const orc_t* o);

/* Streams nbytes of raw I/Q into device d (hop by hop, availability rule of src/rtl_airband.cpp:394-400)
 * and stores every batch that completes, up to max_batches:
 *   waveout [max_batches][C][B], iq_out [max_batches][C][2B], axc [max_batches][C],
 *   trace [max_batches][C][B] (per-sample squelch byte, same encoding as airband_hip_read_trace),
 *   raw_wavein [max_batches][C][B], raw_iq [max_batches][C][2B] (stage-1 output of the batch's NEW hops,
 *   before stage 2 touches it).  Any pointer may be NULL.  Returns batches produced by this call. */
int orc_run_device(orc_t* o, int d, const void* iq, size_t nbytes, int max_batches, float* waveout, float* iq_out, char* axc, uint8_t* trace, float* raw_wavein,
                   float* raw_iq);

/* One batch of device d from a caller-held span laid out as airband_hip_process_device() expects it (hop h at
 * span + h * hop_bytes; WAVE_BATCH + AGC_EXTRA hops on the first call of a stream, WAVE_BATCH afterwards).  Outputs as one
 * batch of orc_run_device ([C][B] ...), any pointer may be NULL. */
int orc_run_span(orc_t* o, int d, const void* span, float* waveout, float* iq_out, char* axc, uint8_t* trace, float* raw_wavein, float* raw_iq);

/* Stage-2 only for device d: consume B new hops of caller-provided stage-1 output (wavein [C][B], iq [C][2B])
 * and produce one batch. */
int orc_run_bins(orc_t* o, int d, const float* wavein, const float* iq, float* waveout, float* iq_out, char* axc, uint8_t* trace);

int orc_channel_stats(orc_t* o, int d, int j, airband_hip_channel_stats* out);
/* same 16 slots as airband_hip_channel_constants() */
int orc_channel_constants(orc_t* o, int d, int j, double* out16);

/* stand-alone pieces for unit-level pinning against the reference */
float orc_window_coeff(int fft_size, int i);                       /* src/rtl_airband.cpp:335-351 */
void orc_sincos_lut(uint32_t phi, float* s, float* c);             /* src/util.cpp:113-127 */
float orc_dbfs_to_level(float dbfs, int fft_size);                 /* src/util.cpp:169-176 */
float orc_fast_atan2(float y, float x);                            /* src/rtl_airband.cpp:147-166 */
float orc_polar_disc_fast(float ar, float aj, float br, float bj); /* src/rtl_airband.cpp:168-172 */
float orc_fm_quadri_demod(float ar, float aj, float br, float bj); /* src/rtl_airband.cpp:174-176 */
float orc_tone_coeff(float tone_freq, float sample_rate, int window); /* src/ctcss.cpp:31-42 */
void orc_notch_run(float freq, float rate, float q, float* x, int n); /* src/filters.cpp:30-64 */
void orc_lowpass_run(float freq, float rate, float* re, float* im, int n); /* src/filters.cpp:70-163 */
void orc_ctcss_run(float ctcss_freq, float sample_rate, int window, const float* x, int n, unsigned char* has_tone, uint64_t* counts2);
void* orc_squelch_new(float snr_db, int manual_dbfs, float ctcss_freq, int wave_rate, int fft_size);
void orc_squelch_raw(void* s, const float* x, int n, unsigned char* flags, float* noise, float* level);
void orc_squelch_raw_filtered(void* s, const float* raw, const float* filtered, int n, unsigned char* flags, float* noise, float* level);
void orc_squelch_raw_audio(void* s, const float* raw, const float* audio, int n, unsigned char* flags);
void orc_squelch_audio_raw(void* s, const float* raw, const float* audio, int n, unsigned char* flags);
void orc_squelch_counts(void* s, uint64_t* out4);
void orc_squelch_free(void* s);

/* mixer sum restated (src/mixer.cpp:133-140,201-214): out_l/out_r [n_mixers][B], sig [n_mixers];
 * waveout [total_channels][B], axc [total_channels]; inputs as in the product ABI. */
void orc_mix(const airband_hip_mixer_input* in, int n_in, const int* chan_base, const float* waveout, const char* axc, int B, int n_mixers, float* out_l, float* out_r,
             uint8_t* sig);

#ifdef __cplusplus
}
#endif
#endif
