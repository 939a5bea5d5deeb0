/* include/airband_hip.h -- C ABI of libairband_hip.so
 *
 * MI355X (gfx950) backend for the one data-parallel hot path of RTLSDR-Airband: the demodulate()
 * loop (reference: src/rtl_airband.cpp:286-672) -- sliding windowed-FFT channelizer, per-bin AM/NFM
 * demodulation, squelch (+CTCSS), IIR notch / Bessel lowpass -- for many independent "dongles"
 * (device_t) at once.  Plain C, plain pointers and sizes; no C++/torch types cross this boundary.
 *
 * Shape of the API follows the only GPU-offload precedent in the reference, the VideoCore FFT
 * backend (reference: src/hello_fft/gpu_fft.h:66-74 gpu_fft_prepare/execute/release, used at
 * src/rtl_airband.cpp:296,458,363): prepare -> {submit, process, collect}* -> release, negative int
 * error codes with the same meaning (-1 no device, -2 unsupported size, -3 out of memory).
 *
 * What each entry point replaces in the reference is cited next to it.
 *
 * Deployment: one process per GPU (any number of handles, streams and threads inside it) is how the library is meant to run.  Until round 5 several processes running its
 * launches on ONE GPU at the same time produced wrong results now and then; what failed were packed-f32 vector instructions, and the library is built without them since
 * (profiles/r05_event_hunt.md).  No wrong value is known of the present build in either arrangement.
 */
#ifndef AIRBAND_HIP_H
#define AIRBAND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIRBAND_HIP_ABI_VERSION 2u /* 2: airband_hip_channel_stats carries signal_outside_filter (the TUI's '~') */

/* error codes (negative ints; 0 = success).  -1/-2/-3 keep gpu_fft_prepare()'s meaning
 * (reference: src/rtl_airband.cpp:297-310). */
#define AIRBAND_HIP_OK 0
#define AIRBAND_HIP_ENODEV (-1)   /* no usable HIP device / HIP runtime error at prepare            */
#define AIRBAND_HIP_EBADSIZE (-2) /* fft_size_log outside [8,13], bad wave_rate, bad counts         */
#define AIRBAND_HIP_ENOMEM (-3)   /* device or host allocation failed                                */
#define AIRBAND_HIP_EINVAL (-4)   /* NULL / inconsistent argument                                    */
#define AIRBAND_HIP_EAGAIN (-5)   /* not enough input queued for a batch / no batch to collect       */
#define AIRBAND_HIP_ERUNTIME (-6) /* HIP launch / copy failure after prepare                         */

/* sample formats: numeric values equal sample_format_t (reference: src/input-common.h:31) */
#define AIRBAND_SFMT_U8 1
#define AIRBAND_SFMT_S8 2
#define AIRBAND_SFMT_S16 3
#define AIRBAND_SFMT_F32 4

/* modulations: numeric values equal enum modulations (reference: src/rtl_airband.h:195-200) */
#define AIRBAND_MOD_AM 0
#define AIRBAND_MOD_NFM 1

/* FM discriminator choice: enum fm_demod_algo (reference: src/rtl_airband.cpp:88-89, -Q option) */
#define AIRBAND_FM_FAST_ATAN2 0
#define AIRBAND_FM_QUADRI_DEMOD 1

/* constants of the reference build (reference: src/rtl_airband.h:67-75) */
#define AIRBAND_AGC_EXTRA 100

/* prepare() flags */
#define AIRBAND_HIP_FLAG_TRACE_SQUELCH 0x1u /* record per-sample squelch state (parity debugging;       \
                                               mirrors the reference's DEBUG_SQUELCH dump,               \
                                               src/squelch.cpp:593-633)                                 */
#define AIRBAND_HIP_FLAG_RESERVED_2 0x2u    /* (was KEEP_BINS: the stage-1 rings always hold the last batch's bins, airband_hip_read_bins) */
#define AIRBAND_HIP_FLAG_FORCE_FFT 0x4u     /* always use the full wavefront-FFT channelizer             */
#define AIRBAND_HIP_FLAG_SERIAL_DEMOD 0x8u  /* run the per-kind demod kernels one after another instead   \
                                               of side by side on forked streams (profiling aid)          */
#define AIRBAND_HIP_FLAG_PIPELINE 0x10u     /* throughput mode: a process call enqueues stage 1 of ITS batch \
                                               beside stage 2 of the PREVIOUS batch (the channelizer is       \
                                               HBM-bound, the demod kernels are not).  Results lag one batch: \
                                               the first call produces none, airband_hip_flush() drains the   \
                                               last.  Same values, bit for bit, as the sequential mode.       \
                                               Ignored when a channel has AFC (stage 1 of the next batch needs \
                                               stage 2's verdict, src/rtl_airband.cpp:222-251).               */
#define AIRBAND_HIP_FLAG_REGROUP 0x20u      /* stage 2 re-sorts its channels at every batch boundary so that   \
                                               channels whose squelch is closed share wavefronts (csrc/demod.hip,\
                                               "regrouping"): same values, bit for bit; what changes is which   \
                                               64 channels a wavefront works on.  Pays on a band whose channels \
                                               are mostly quiet; costs ring-line fetches where neighbouring     \
                                               channels differ in state.  AIRBAND_HIP_REGROUP=0|1 in the         \
                                               environment overrides the flags either way (A/B measurements;    \
                                               2 and 3 select the two other forms that were measured: line      \
                                               groups sorted with free-running wavefronts, and a per-batch      \
                                               permutation in front of one-wavefront workgroups --              \
                                               profiles/r06_experiments.md L, M).                               \
                                               With neither this flag nor NO_REGROUP the library decides by      \
                                               residency: on while all of the handle's lane-per-channel          \
                                               wavefronts are resident at once on a full chip (about 24 000 to   \
                                               50 000 dongles of eight channels on an MI355X), off otherwise.   */
#define AIRBAND_HIP_FLAG_NO_REGROUP 0x40u   /* never regroup (slot order), whatever the handle's size          */

/* Per-channel configuration: the values a multichannel-mode `channels` entry carries after
 * parse_channels() (reference: src/config.cpp:306-726).  The library derives bin index, derotation
 * step, filter coefficients, squelch constants and CTCSS tone banks from these exactly the way the
 * reference does (bins :666-667, dm_dphi :679-712, notch :516-564, ctcss :565-591,
 * bandwidth :592-619, ampfactor :620-645, tau :646-650, squelch thresholds :436-515). */
typedef struct airband_hip_channel_cfg {
    int32_t frequency;                /* Hz; freq_t.frequency                                          */
    int32_t modulation;               /* AIRBAND_MOD_*                                                 */
    int32_t afc;                      /* channel_t.afc, 0 = off                                        */
    int32_t squelch_threshold_dbfs;   /* `squelch_threshold`: 0 = auto squelch, <0 = manual level      */
    float squelch_snr_threshold_db;   /* `squelch_snr_threshold`: -1 = keep default 9.54 dB            */
    float notch_freq;                 /* `notch` in Hz, 0 = off                                        */
    float notch_q;                    /* `notch_q`, 0 = default 10                                     */
    float ctcss_freq;                 /* `ctcss` in Hz, 0 = off                                        */
    int32_t bandwidth_hz;             /* `bandwidth`, 0 = off; lowpass at bandwidth/2, needs raw I/Q   */
    float ampfactor;                  /* `ampfactor` (default 1.0)                                     */
    int32_t tau_us;                   /* `tau` in microseconds, -1 = inherit device/global             */
    int32_t has_iq_outputs;           /* channel has a raw-I/Q output (channel_t.has_iq_outputs)       */
} airband_hip_channel_cfg;

/* Per-device ("dongle") configuration: the input_t fields demodulate() reads
 * (reference: src/input-common.h:39-57) plus the channel list. */
typedef struct airband_hip_device_cfg {
    int32_t sample_rate; /* Hz, input_t.sample_rate                                                   */
    int32_t centerfreq;  /* Hz, input_t.centerfreq                                                    */
    int32_t sfmt;        /* AIRBAND_SFMT_*                                                            */
    float fullscale;     /* input_t.fullscale; 0 = the driver default for sfmt                        */
    int32_t tau_us;      /* device-level `tau`, -1 = global default (200 us)                          */
    int32_t channel_count;
    const airband_hip_channel_cfg* channels;
} airband_hip_device_cfg;

typedef struct airband_hip_config {
    uint32_t abi_version; /* AIRBAND_HIP_ABI_VERSION                                                   */
    uint32_t flags;       /* AIRBAND_HIP_FLAG_*                                                        */
    int32_t fft_size_log; /* global fft_size_log, 8..13 (reference: src/rtl_airband.cpp:786-800)       */
    int32_t wave_rate;    /* 8000 = AM-only build, 16000 = NFM build (reference: src/rtl_airband.h:67) */
    int32_t fm_demod;     /* AIRBAND_FM_*                                                              */
    int32_t hip_device;   /* HIP device ordinal this handle lives on                                   */
    int32_t device_count;
    const airband_hip_device_cfg* devices;
} airband_hip_config;

/* Mixer wiring (reference: src/mixer.cpp:57-94 mixer_connect_input, :133-140 mix_waveforms).
 * One entry per (channel -> mixer) connection. */
typedef struct airband_hip_mixer_input {
    int32_t device;   /* dongle index                                                                 */
    int32_t channel;  /* channel index within the dongle                                              */
    int32_t mixer;    /* destination mixer index                                                      */
    float ampfactor;  /* mixinput_t.ampfactor                                                         */
    float balance;    /* -1..1; ampl=min(1,1-bal), ampr=min(1,1+bal)                                  */
} airband_hip_mixer_input;

/* Geometry the library derived; lets the caller size its buffers. */
typedef struct airband_hip_geometry {
    int32_t fft_size;          /* 1 << fft_size_log                                                   */
    int32_t wave_rate;         /* WAVE_RATE                                                           */
    int32_t wave_batch;        /* WAVE_BATCH = WAVE_RATE/8 output samples per batch per channel        */
    int32_t device_count;
    int32_t total_channels;    /* sum of channel_count                                                */
    int32_t max_channels;      /* max channel_count                                                   */
    int32_t mixer_count;
    int32_t wave_stride;       /* floats between two channels' rows in the DEVICE waveout buffer (airband_hip_device_results) */
    int64_t first_batch_bytes; /* per device: IQ bytes the first batch consumes (incl. AGC_EXTRA hops) */
    int64_t batch_bytes;       /* per device: IQ bytes every later batch consumes                     */
    int64_t lookahead_bytes;   /* per device: bytes past the batch the last FFT window still reads    */
} airband_hip_geometry;

/* Per-channel squelch/AGC statistics mirrored back to the host after every batch: the getters the
 * stats file and TUI read (reference: src/output.cpp:617-761, src/rtl_airband.cpp:633-640). */
typedef struct airband_hip_channel_stats {
    float noise_level;    /* Squelch::noise_level()                                                   */
    float signal_level;   /* Squelch::signal_level()                                                  */
    float squelch_level;  /* Squelch::squelch_level()                                                 */
    float agcavgfast;     /* freq_t.agcavgfast                                                        */
    uint64_t open_count;  /* Squelch::open_count()                                                    */
    uint64_t flappy_count;
    uint64_t ctcss_count;
    uint64_t no_ctcss_count;
    uint64_t active_counter; /* freq_t.active_counter                                                 */
    int32_t bin;             /* dev->bins[i] (moves only with AFC)                                    */
    int32_t squelch_state;   /* Squelch::State after the batch                                        */
    int32_t signal_outside_filter; /* Squelch::signal_outside_filter() after the batch (src/squelch.cpp:152-154): the TUI prints '~' for it (src/rtl_airband.cpp:633) */
    int32_t reserved;
} airband_hip_channel_stats;

typedef struct airband_hip_handle airband_hip_handle;

/* ---- lifecycle ------------------------------------------------------------------------------ */

/* Replaces: fftwf plan creation in init_demod() (reference: src/rtl_airband.cpp:253-266), the
 * LUT/window set-up at the top of demodulate() (:316-351), sincosf_lut_init() (src/util.cpp:105)
 * and the per-channel derivations of parse_channels() listed above.
 * Returns 0 or a negative AIRBAND_HIP_E* code; *out is NULL on failure. */
int airband_hip_prepare(const airband_hip_config* cfg, airband_hip_handle** out);

/* Optional: wire channels into mixers before the first batch (reference: src/config.cpp mixer outputs,
 * src/mixer.cpp:57-94).  mixer_count mixers, n_inputs connections. */
int airband_hip_set_mixers(airband_hip_handle* h, int32_t mixer_count, const airband_hip_mixer_input* inputs, int32_t n_inputs);

/* Forces the right channel of one mixer on (or back to what its local inputs say): a mixer is stereo as soon as ANY of its inputs has a
 * balance (mixer->channel.mode = MM_STEREO, src/mixer.cpp:84-85), and when its inputs are spread over several handles (GPUs) the handles
 * without such an input must produce a right-channel partial sum all the same.  Call after airband_hip_set_mixers. */
int airband_hip_mixer_set_stereo(airband_hip_handle* h, int32_t mixer, int32_t stereo);

/* ---- the mixer exchange: the one step of the path where dongles on different GPUs meet (src/mixer.cpp:133-140,201-214) ----------------
 * Every handle holds PARTIAL sums of the mixers over its own dongles (airband_hip_set_mixers with the same mixer_count on each).  The
 * exchange is an in-place all-reduce of those buffers -- SUM of left and right, MAX of the signal flags -- over RCCL (xGMI between the
 * GPUs of a node), enqueued on the GPU behind the batch's results: no host synchronisation.  librccl.so is loaded on first use
 * (AIRBAND_HIP_RCCL_LIB=<path> names another library with the same eight entry points).
 *   one process per GPU (bench.py --gpus N):  rank 0 calls airband_hip_comm_unique_id and hands the 128 bytes to the others (any
 *       transport), every rank calls airband_hip_comm_init_rank(h, id, nranks, rank);
 *   one process, one handle per GPU (the reference-side shim, integration/demod_hip.cpp):  airband_hip_comm_init_all(handles, n) --
 *       the handles must sit on n DIFFERENT GPUs;
 * then, per batch and per handle (from n threads or one after the other inside airband_hip_comm_group_begin / _end):
 *       airband_hip_allreduce_mixers(h, stream)   -- stream NULL = the handle's own; airband_hip_collect_mixers() then returns the node's sums.
 * Handles of one process that share a GPU need no fabric: airband_hip_add_mixers(dst, src) adds src's partial sums to dst's with a kernel
 * on dst's stream, ordered behind src's batch (summation order: dst, then src -- deterministic).
 * Float summation order differs from a single handle's connection order, so mixer parity across handles is tolerance parity (1e-4 RMS). */
#define AIRBAND_HIP_COMM_ID_BYTES 128
int airband_hip_comm_unique_id(uint8_t id[AIRBAND_HIP_COMM_ID_BYTES]);
int airband_hip_comm_init_rank(airband_hip_handle* h, const uint8_t id[AIRBAND_HIP_COMM_ID_BYTES], int32_t nranks, int32_t rank);
int airband_hip_comm_init_all(airband_hip_handle** handles, int32_t n);
int airband_hip_comm_group_begin(void);
int airband_hip_comm_group_end(void);
int airband_hip_allreduce_mixers(airband_hip_handle* h, void* stream);
int airband_hip_add_mixers(airband_hip_handle* dst, airband_hip_handle* src);
int airband_hip_comm_destroy(airband_hip_handle* h);
/* Zeroes the handle's partial sums and signal flags, ordered on the GPU behind whatever touched them last.  For a handle that takes part in
 * an exchange (add_mixers / allreduce_mixers) but ran NO batch this round because every dongle of it is switched off: the reference keeps
 * mixing the inputs that are left (mixer_disable_input(), src/mixer.cpp:96-112, called for a failed device's outputs, src/rtl_airband.cpp:383-391),
 * so such a handle must add nothing -- while its buffers still hold its last batch's sums or, after an in-place all-reduce, the node's. */
int airband_hip_clear_mixers(airband_hip_handle* h);

/* Masks one mixer connection out (enabled = 0) or back in, by its index in the `inputs` array handed to airband_hip_set_mixers:
 * a masked input adds nothing and does not raise the mixer's signal flag -- mixer_disable_input(), which the reference
 * calls when a device fails (src/mixer.cpp:96-110, src/rtl_airband.cpp:377-391).  Takes effect from the next batch. */
int airband_hip_mixer_enable_input(airband_hip_handle* h, int32_t input_index, int32_t enabled);

/* Switches one dongle of the handle off (enabled = 0) or back on.  Replaces: what demodulate() does when input->state is no longer
 * INPUT_RUNNING (reference: src/rtl_airband.cpp:383-391 -- the device is passed by from then on and its outputs are disabled;
 * src/input-file.cpp:101-111 sets INPUT_FAILED at end of file, the drivers on a dead dongle).  A disabled dongle
 *   - is not waited for by airband_hip_process()'s availability rule, and bytes submitted for it are dropped;
 *   - is skipped by both stages: its channels' squelch / filter / AGC state stays frozen, its result rows are not written again,
 *     its channels report axcindicate ' ' (NO_SIGNAL);
 *   - adds nothing to any mixer it is wired into (mixer_disable_input() for each of its outputs, src/mixer.cpp:96-110).
 * Takes effect with the next process call (a pipelined handle still runs stage 2 of the batch already under way).  Re-enabling is an
 * extension (the reference's INPUT_DISABLED is final): the dongle rejoins at the handle's common stream position with the state it
 * was frozen with; the AGC_EXTRA samples of lead-in the first batch back sees are unspecified.
 * With every dongle disabled airband_hip_process() returns AIRBAND_HIP_EAGAIN -- the caller's cue to stop (the reference exits,
 * src/rtl_airband.cpp:377-381). */
int airband_hip_device_enable(airband_hip_handle* h, int32_t dev, int32_t enabled);

/* Number of HIP devices visible to the process (0 when there is none / no runtime): lets a host without HIP headers spread its
 * demodulate() shards over the GPUs of a node (reference: one demodulate thread per shard, src/rtl_airband.cpp:1052-1086,1110-1112). */
int airband_hip_gpu_count(void);

/* Replaces: gpu_fft_release() on do_exit (reference: src/rtl_airband.cpp:360-365). */
void airband_hip_release(airband_hip_handle* h);

int airband_hip_get_geometry(const airband_hip_handle* h, airband_hip_geometry* out);

/* Human-readable text of the last error on this handle (or of the last failed prepare when h is NULL). */
const char* airband_hip_last_error(const airband_hip_handle* h);

/* ---- data path ------------------------------------------------------------------------------ */

/* Host-ring path.  Appends `nbytes` of raw interleaved I/Q of device `dev` to the library's pinned staging ring of that device
 * (one CPU copy; the batch later leaves the ring by DMA, asynchronously, on a copy stream).
 * Replaces: the consumer side of input->buffer (reference: src/rtl_airband.cpp:370-375,:669); the reference-side shim calls it with
 * the span [bufs, bufe) the rx thread produced (circbuffer_append, src/input-helpers.cpp:37-63) and then advances bufs.
 * Returns the number of bytes accepted -- fewer than nbytes when the ring is full (it holds the first batch with its lead-in plus
 * three more batches, about the reference's own 2 560 000-byte ring) -- or <0.
 * Thread-compatibility: calls for DIFFERENT devices may run concurrently (one rx / feeder thread per dongle, like the reference's
 * input threads); calls for one device, and airband_hip_process(), must not overlap each other. */
int64_t airband_hip_submit(airband_hip_handle* h, int32_t dev, const void* iq, size_t nbytes);

/* Runs ONE batch (WAVE_BATCH output samples for every channel of every device) if every device has
 * enough queued input (the availability rule of src/rtl_airband.cpp:394-400 applied per batch);
 * AIRBAND_HIP_EAGAIN otherwise.  Asynchronous: returns after enqueueing the kernels.
 * Replaces: one WAVE_BATCH worth of demodulate() iterations for all devices
 * (reference: src/rtl_airband.cpp:402-492 stage 1 and :494-655 stage 2). */
int airband_hip_process(airband_hip_handle* h);

/* Would airband_hip_process() run a batch now (OK) or not yet (EAGAIN)?  The availability rule alone; nothing is enqueued.  A caller that
 * drives several handles in lockstep -- the shards of one device class on the GPUs of a node, whose mixer sums belong to the same batch --
 * asks every one of them first (integration/demod_hip.cpp). */
int airband_hip_batch_ready(airband_hip_handle* h);

/* Zero-copy path for HBM-resident I/Q.  `d_iq` is a DEVICE pointer; device `d`'s span for this batch
 * starts at d_iq + d*stride_bytes and holds at least (first_)batch_bytes + lookahead_bytes bytes:
 * the stream bytes of this batch followed by the bytes the last window overlaps into the next batch
 * (exactly what a tail-replicated ring, src/input-helpers.cpp:43-51, holds at that offset).
 * Alignment: where a hop is a whole number of 16-byte pieces (2.56 MS/s: 320 / 640 bytes) d_iq and stride_bytes are multiples of 16, and so is every
 * batch's offset into a stream; for any other hop they are multiples of the largest power of two that divides the hop's bytes (2.4 MS/s: 300 bytes -> 4,
 * 600 -> 8; 2.0 MS/s: 250 bytes -> 2, i.e. whole I/Q samples) -- batch offsets, being multiples of the hop, keep that alignment -- and the
 * channelizer then stages aligned pieces from the aligned byte at or in front of the span, i.e. it may READ up to 15 bytes in front of d_iq + d*stride_bytes
 * (bytes of the same allocation: the tail of the previous batch, or of the previous dongle's row).  AIRBAND_HIP_EINVAL otherwise.
 * `stream` is a hipStream_t (NULL = the handle's own stream). */
int airband_hip_process_device(airband_hip_handle* h, const void* d_iq, size_t stride_bytes, void* stream);

/* Waits for the oldest un-collected batch and copies its results to HOST memory.  Any pointer may be
 * NULL to skip that output.
 *   waveout  [total_channels][wave_batch]   = channel->waveout[0..WAVE_BATCH) as process_outputs()
 *            sees it (reference: src/output.cpp:460,535) -- the library performs the output thread's
 *            tail copy (src/output.cpp:920) itself;
 *   iq_out   [total_channels][2*wave_batch] = channel->iq_out (zeros for channels without I/Q outputs);
 *   axcindicate [total_channels]            = channel->axcindicate (' ', '*', '<', '>');
 *   stats    [total_channels].
 * Channels are numbered device-major: index = (sum of channel_count of earlier devices) + channel.
 * Replaces: the hand-off at src/rtl_airband.cpp:649-662 (waveavail / Signal::send). */
int airband_hip_collect(airband_hip_handle* h, float* waveout, float* iq_out, char* axcindicate, airband_hip_channel_stats* stats);

/* The same four outputs for the channel range [first_channel, first_channel + n_channels) only (device-major numbering as
 * above; arrays are sized for n_channels).  Does NOT mark the batch as collected, so it can be called any number of times
 * between two process calls -- the reference's per-device consumer (process_outputs() walks one device_t at a time,
 * src/output.cpp:905-935), and the spot checks of handles too large to copy whole (bench.py --verify). */
int airband_hip_collect_channels(airband_hip_handle* h, int64_t first_channel, int64_t n_channels, float* waveout, float* iq_out, char* axcindicate,
                                 airband_hip_channel_stats* stats);

/* Mixer outputs of the batch last collected: left [mixer_count][wave_batch], right likewise (zeros for
 * mono mixers), has_signal [mixer_count] (reference: src/mixer.cpp:201-214). HOST pointers.
 * Multi-GPU callers all-reduce (sum / max) these across ranks. */
int airband_hip_collect_mixers(airband_hip_handle* h, float* left, float* right, uint8_t* has_signal);

/* Device-side views of the current result buffers (valid until the next process call), for consumers
 * that stay on the GPU (e.g. an RCCL all-reduce of the mixer sums).  Any out-pointer may be NULL.
 * d_waveout is laid out like the reference's channel->waveout arrays: channel c's WAVE_BATCH samples start at
 * d_waveout + c * geometry.wave_stride (the AGC_EXTRA floats behind them are the carry into the next batch). */
int airband_hip_device_results(airband_hip_handle* h, float** d_waveout, float** d_iq_out, uint8_t** d_axc, float** d_mix_left, float** d_mix_right,
                               uint8_t** d_mix_signal);

/* Makes `stream` (a hipStream_t of a GPU-side consumer, e.g. the stream an RCCL all-reduce of the mixer sums is issued on)
 * wait for the results of the batch the last process call completed -- no host synchronisation.  The consumer hands its
 * stream to the next airband_hip_process_device() call, which then orders the overwrite of the result buffers behind it. */
int airband_hip_stream_wait_results(airband_hip_handle* h, void* stream);

/* AIRBAND_HIP_FLAG_PIPELINE handles: enqueues stage 2 of the batch whose stage 1 the last process call started, so that its
 * results can be collected without feeding another batch (end of stream).  No-op otherwise. */
int airband_hip_flush(airband_hip_handle* h);

/* Blocks until everything enqueued on the handle has finished. */
int airband_hip_synchronize(airband_hip_handle* h);

/* ---- introspection used by parity tests and the bench ---------------------------------------- */

/* Stage-2 only: run the per-channel demod/squelch/filter batch on caller-provided stage-1 output
 * (HOST pointers): wavein [total_channels][wave_batch] magnitudes and iq_in [total_channels][2*wave_batch]
 * raw bin I/Q for the batch's WAVE_BATCH new hops.  Lets tests feed the oracle's exact stage-1 values
 * and demand bit-identical stage-2 results.  NFM channels: `wavein` is ignored -- stage 2 recomputes |bin| = sqrtf(re^2 + im^2) from
 * iq_in, as the reference computes it from the same two floats (src/rtl_airband.cpp:484-487).  AFC is not run on these batches (there is
 * no spectrum to look at).  Not available on pipelined handles. */
int airband_hip_process_bins(airband_hip_handle* h, const float* wavein, const float* iq_in);

/* Stage-1 output of the last batch, i.e. what the reference holds in channel->wavein[AGC_EXTRA..] / iq_in[2*AGC_EXTRA..] after
 * its FFT loop (src/rtl_airband.cpp:483-489): wavein [total_channels][wave_batch], iq_in [total_channels][2*wave_batch] (zeros for
 * channels that do not need raw I/Q).  HOST pointers.  NFM channels: stage 1 stores only the raw bin I/Q, |bin| is recomputed
 * here as sqrtf(re^2 + im^2) exactly as stage 2 does.  AM channels with a lowpass filter / raw-I/Q output: stage 2 overwrites
 * wavein[j] with the filtered magnitude in place, as the reference does (src/rtl_airband.cpp:524) -- that is what is returned. */
int airband_hip_read_bins(airband_hip_handle* h, float* wavein, float* iq_in);
int airband_hip_read_bins_channels(airband_hip_handle* h, int64_t first_channel, int64_t n_channels, float* wavein, float* iq_in);

/* Per-sample squelch trace of the last batch (needs AIRBAND_HIP_FLAG_TRACE_SQUELCH):
 * state [total_channels][wave_batch] bytes: bits 0-2 Squelch::State after process_raw_sample,
 * bit 3 is_open(), bit 4 should_process_audio(), bit 5 CTCSS has_tone (slow if enough samples else fast). */
int airband_hip_read_trace(airband_hip_handle* h, uint8_t* state);
int airband_hip_read_trace_channels(airband_hip_handle* h, int64_t first_channel, int64_t n_channels, uint8_t* state);

/* Derived per-channel constants (bin, dm_dphi, filter taps ...) as the library computed them; lets
 * tests compare against the reference's own derivation.  out_vals must hold 16 doubles:
 *  [0] bin  [1] dm_dphi  [2] alpha  [3] notch d0 [4] d1 [5] d2  [6] lowpass gain [7] ycoeff0 [8] ycoeff1
 *  [9] normal_signal_ratio [10] manual_level(-1 if auto) [11] n_tones_fast [12] n_tones_slow
 *  [13] needs_raw_iq [14] window_fast [15] window_slow */
int airband_hip_channel_constants(const airband_hip_handle* h, int32_t channel_index, double* out_vals);

/* Same slots, computed without any GPU (pure host arithmetic): lets CPU-only tests pin the derivations. */
int airband_hip_derive_constants(const airband_hip_config* cfg, int32_t channel_index, double* out_vals);

/* Host-only check of the matrix-core channelizer's coefficient tables for `cfg` (needs no GPU): the value the kernel's integer digit
 * sums recombine to, on `windows` pseudo-random raw windows per (dongle, group of 8 channels), against the defining sum
 * X[bin] = sum_n lev[b_n] w[n] exp(-2 pi i bin n / N) (reference: src/rtl_airband.cpp:316-351,402-489) evaluated in double.
 * *max_rel_err = largest error / RMS of the exact values.  AIRBAND_HIP_EBADSIZE when `cfg` would run on the wavefront-FFT channelizer. */
int airband_hip_dft_selftest(const airband_hip_config* cfg, int32_t windows, double* max_rel_err);

/* Milliseconds the GPU spent on the last finished batch (HIP events on the streams the kernels run on):
 * [0] channelizer kernel, [1] demod kernels (+ per-kind emit), [2] joint emit / mixers, [3] their sum.  Waits for the
 * enqueued batches. */
int airband_hip_last_timings(airband_hip_handle* h, float* ms4);

/* Sums of the same four figures over every batch that has FINISHED since the last reset, and how many batches that is.
 * Waits for the enqueued batches; callers that must not stall inside a run call it once, after airband_hip_synchronize(). */
int airband_hip_timing_totals(airband_hip_handle* h, double* ms4_sum, int64_t* n_batches, int32_t reset);

/* Extra preprocessor defines this library was compiled with ("" for the product build; kernel experiments are built under a
 * different file name and say here what they changed, so that a measurement can always be traced to the code that produced it). */
const char* airband_hip_build_info(void);

/* 1 if stage 2 of this handle re-sorts its channels at batch boundaries (AIRBAND_HIP_FLAG_REGROUP, AIRBAND_HIP_REGROUP=1 in the environment, or the library's own choice by residency), else 0. */
int airband_hip_regrouped(const airband_hip_handle* h);

/* Name of the channelizer variant the handle selected ("fft_wave64" / "dft_mfma_i8"). */
const char* airband_hip_channelizer_name(const airband_hip_handle* h);

/* Uploads the transmitter table of the synthetic dongles: carriers [n_carriers][12] int64 rows
 * (rtlsdr-airband_amd/siggen.py::Carrier.as_row), the Q8 noise multiplier and the 4096-entry int16 sine table. */
int airband_hip_set_signal_plan(airband_hip_handle* h, const int64_t* carriers, int32_t n_carriers, int32_t noise_q8, const int16_t* sin_table4096);

/* Synthetic fleets whose dongles do NOT share a channel plan (every device_t derives its own bins, src/config.cpp:666-667): dongle d (global index, see
 * device_index_offset below) belongs to plan p = d mod n_plans, and its carrier c is generated ((p >> 2c) & 3) * shift_step (u32 turns per sample) above the
 * table's frequency -- up to 4^8 = 65 536 distinct plans of eight carriers.  The caller configures the channels' frequencies to match.  n_plans = 1 (default): off. */
int airband_hip_set_signal_plan_shift(airband_hip_handle* h, int32_t n_plans, uint32_t shift_step);

/* Deterministic synthetic dongles, generated on the GPU straight into HBM (integer-only arithmetic so
 * the numpy generator in the tests produces identical bytes).  Fills, for every device d in
 * [0, device_count), nbytes of u8 I/Q starting at stream byte offset `start_byte` into
 * d_iq + d*stride_bytes.  See DESIGN.md "synthetic dongles".  d_iq is a DEVICE pointer. */
int airband_hip_generate_iq(airband_hip_handle* h, void* d_iq, size_t stride_bytes, uint64_t start_byte, size_t nbytes, uint64_t seed,
                            int32_t device_index_offset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AIRBAND_HIP_H */
