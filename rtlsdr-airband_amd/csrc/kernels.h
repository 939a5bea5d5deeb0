/* csrc/kernels.h -- launch interfaces of the HIP kernels (host side: airband_hip.cpp). */
#ifndef AIRBAND_CSRC_KERNELS_H
#define AIRBAND_CSRC_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/airband_hip.h"
#include "common.h"

namespace airband {

/* Device buffers are blocked time-major rings (see ab_ring_base in common.h): logical row r of the current batch
 * lives at physical row (row0 + r) mod ring_rows, ring_rows = WAVE_BATCH + AGC_EXTRA.  Logical rows [0, AGC_EXTRA) are the carry from
 * the previous batch (what the reference keeps with its memmove / tail copy, src/rtl_airband.cpp:621-624 and
 * src/output.cpp:920), rows [AGC_EXTRA, AGC_EXTRA + WAVE_BATCH) are this batch's new hops.  Advancing row0 by
 * WAVE_BATCH per batch replaces both copies. */

struct ChannelizerArgs {
    const uint8_t* iq;      /* device: dongle d's span starts at iq + d * iq_stride */
    long iq_stride;         /* bytes */
    const DevConst* dev;
    const ChanState* cs;    /* for the (AFC-movable) bin of every slot */
    const ChanConst* cc;
    const int* ext_to_slot; /* channel (device-major external index) -> demod slot */
    const float* window;    /* fft_size */
    const float* window_dec;/* fft_size >= 1024: [fft_size / 512][512] the window at samples n2 + (fft_size / 512) n1, row n2 (decimated wavefront FFT); else unused */
    const float2* twiddle;  /* fft_size: exp(-2 pi i k / fft_size) */
    float* mag;             /* |bin| ring, blocked by 64 slots and transposed in tiles of AB_TILE_ROWS rows (common.h: ab_tile_base / ab_tile_off) */
    float2* iq_bins;        /* raw bin I/Q ring, same layout */
    float* last_spectrum;   /* [n_dev][2*fft_size] full FFT of the batch's last hop (AFC), or null */
    int n_dev, fft_log;
    int hop_samples, bytes_per_sample, sfmt;
    float scale;
    int row0, ring_rows;
    int first_row;          /* first logical row to produce (0 on the first batch, AGC_EXTRA later) */
    int n_hops;             /* hops to produce */
    int max_ch;
    int spectrum_only;      /* 1: no ring output, only last_spectrum of dongles with an AFC channel (handles whose batches run on the matrix-core channelizer) */
};

/* matrix-core channelizer (channelizer_dft.hip) */
struct DftArgs {
    const uint8_t* iq;
    long iq_stride;
    const DevConst* dev;
    const ChanConst* cc;
    const int* ext_to_slot;
    const int* item_dev;    /* [n_items] work items: (dongle, group of 8 channels, coefficient-table index) */
    const int* item_group;
    const int* item_bset;
    const int8_t* bfrag;    /* [n_bsets][window pieces][3 digits][k-steps][64 lanes][16 bytes] MFMA B fragments (k-steps = min(fft_size, 512) / 32) */
    const double* corr;     /* [n_bsets * window pieces][16] offset restoring (b - 127.5) from (b - 128), in table units */
    double unscale;         /* u8: 1 / (table scale * 127.5); CS16: 1 / table scale (the kernel multiplies by the dongle's 1 / fullscale) */
    float* mag;
    float2* iq_bins;
    int n_dev, n_items, splits, fft_size, sfmt;
    int piece0, np_total;            /* fft_size > 512: first window piece of this launch / number of pieces of the window (set by launch_channelizer_dft) */
    float4* partial;                 /* fft_size 8192 only: [n_items][tiles][64] partial sums between the two passes (dft_partial_tiles() tiles per item) */
    int edge_hi_zero;                /* most significant digit is zero in k-steps 0,1,14,15 for every coefficient table */
    int hop_bytes, lds_per_buf, sub, nbuf; /* sub = 16-hop MFMA tiles per staging step; nbuf staging buffers */
    int row0, ring_rows, first_row, n_hops;
    int extra_lds;          /* bytes of LDS the launch asks for beyond what it uses: AIRBAND_HIP_FLAG_PIPELINE handles hold the channelizer to five wavefronts per CU, so that
                             * every CU has register room for the stage-2 wavefronts of the batch before (airband_hip.cpp; profiles/r06_experiments.md I) */
};

/* CF32 dongles on the float32 matrix pipe (channelizer_f32.hip) */
struct F32Args {
    const uint8_t* iq;
    long iq_stride;
    const DevConst* dev;
    const ChanConst* cc;
    const int* ext_to_slot;
    const int* item_dev;    /* work items as for DftArgs: (dongle, group of 8 channels, coefficient-table index) */
    const int* item_group;
    const int* item_bset;
    const float* btab;      /* [n_bsets][f32_nw pieces][MFMAs per piece][64 lanes] window x twiddle, ordered as the kernel contracts (params.cpp, build_f32_tables) */
    float* mag;
    float2* iq_bins;
    int n_items, splits, fft_size;
    int hop_bytes, pad, lds_per_buf; /* bytes per hop (8 x hop_samples), padding per hop in the staged image, bytes per staging buffer */
    int row0, ring_rows, first_row, n_hops;
    int seg, n_seg;         /* fft_size 4096 / 8192: window segment of this launch / segments per window (set by launch_channelizer_f32) */
    float4* partial;        /* those sizes only: [n_items][f32_partial_tiles() tiles][64] partial sums between the segments' launches */
};

struct DemodArgs {
    const ChanConst* cc;
    ChanState* cs;
    float* mag;
    const float2* iq;
    float* out_wave;        /* [total_channels][wave_stride]: channel->waveout rows (tail + WAVE_BATCH), written directly */
    uint8_t* out_axc;       /* [total_channels] */
    const int* slot_to_ext;
    int wave_stride;
    int tail_copy;          /* 0 for the very first batch: nothing has been consumed yet */
    float2* iq_out;         /* [slot blocks][wave_batch][64] raw I/Q of open samples (channels with has_iq_outputs), emit_iq_kernel turns it channel-major */
    float* sqbuf;           /* [slot blocks][AB_SQ_BUF][64] the squelch's pre-filter delay line (ab_ring_base); only the generic kind still stores it -- the NFM + lowpass kind recomputes its entries (SqShadow) */
    const float* ct_coeff;  /* [n_ctcss][2 detectors][AB_MAX_TONES] */
    float* ct_q;            /* [n_ctcss][2 detectors][q1|q2][AB_MAX_TONES] */
    uint8_t* trace;         /* [slot blocks][wave_batch][64] per-sample squelch trace, or null */
    /* split (CTCSS-capable) kinds: front -> tone -> back hand-off */
    /* hand-off rows, channel-major (the tone kernel reads a channel's samples across its lanes):
     *   generic kind  [slots of the generic blocks][wave_batch] (pre-notch audio, flag word) pairs
     *   NFM + CTCSS   [slots of its blocks][ct_pk_pitch] one word per sample (demod.hip, HAND_WORD), rows of whole 128-byte lines */
    float2* ct_af;
    unsigned* ct_ap;
    int ct_pk_pitch;
    int ct_pk_first_block, ct_pk_n_blocks, ct_gen_first_block, ct_gen_n_blocks;
    unsigned long long* ct_mask;   /* [ct blocks][wave_batch / 50][64] tone-present bits, blocks counted from ct_first_block (both split kinds) */
    int ct_first_block, ct_n_blocks;
    const float* sin_lut;   /* 257 */
    const float* cos_lut;   /* 257 */
    int ct_stride;
    int n_slots, wave_batch, row0, ring_rows;
    /* REGROUPED handles (AIRBAND_HIP_FLAG_REGROUP, demod.hip "regrouping"): the lane-per-channel kernels run as workgroups of AB_REGROUP_WAVES wavefronts that
     * share AB_REGROUP_WAVES x 64 consecutive slots and deal them out among themselves by squelch state -- the channels that are not at rest in CLOSED to the first
     * wavefronts, the closed ones to the last, so that whole wavefronts of closed channels skip the open channels' instructions -- and walk the batch in step
     * (a workgroup barrier every eight samples), so that the ring lines neighbouring slots share are fetched from memory once.  Everything a lane touches is
     * addressed by its SLOT: which lane works on which slot does not change a result.  0: lane l of block b works on slot 64 b + l. */
    int regroup;
    /* regroup == 3 (round 6, third form): a per-batch permutation of every kind's slots, built by regroup_perm_kernel in front of the stage -- inside segments of sixteen
     * blocks the LINE GROUPS (slots that share a 128-byte ring line) with a channel that is not at rest come first, the groups at rest behind them -- read by ordinary
     * one-wavefront workgroups: lane l of block b works on slot perm[64 b + l].  Nothing is shared between wavefronts, nobody waits, no workgroup needs four slots at once. */
    int* perm;
    uint8_t* sq_key;        /* [n_slots] split kinds: the front kernel leaves 1 where the channel had audio or went CLOSED in this batch; the tone kernel skips the others, regrouped handles' back kernel deals its slots out by it */
};

struct EmitArgs { /* raw I/Q outputs only: audio goes straight to its channel row */
    const float2* iq_out;
    const int* slot_to_ext;
    float* out_iq;          /* [total_channels][2*wave_batch] or null */
    int n_slots, wave_batch;
};

struct MixArgs {
    const float* out_wave;  /* [total_channels][wave_stride] */
    int wave_stride;
    const uint8_t* out_axc;
    const int* in_chan;     /* [n_inputs] external channel index, grouped by mixer */
    const float* in_ml;     /* ampfactor * ampl */
    const float* in_mr;     /* ampfactor * ampr */
    const int* run_first;       /* [n_runs + 1] offsets into the input arrays: runs of <= AB_MIX_RUN inputs of one mixer */
    const int* run_mixer;       /* [n_runs] */
    const int* mixer_first_run; /* [n_mixers + 1] */
    const uint8_t* mixer_stereo;
    float* run_left;            /* [n_runs][wave_batch] stage-A sums */
    float* run_right;
    uint8_t* run_signal;        /* [n_runs] */
    float* left;            /* [n_mixers][wave_batch] */
    float* right;
    uint8_t* has_signal;    /* [n_mixers] */
    int n_mixers, n_runs, wave_batch;
};

struct SiggenArgs {
    uint8_t* iq;
    long stride;            /* bytes per dongle */
    const int16_t* sin_tab; /* 4096 */
    const long long* carriers; /* [n_carriers][12] */
    int n_carriers;
    int n_dev;
    int dev_offset;
    unsigned long long start_sample;
    long n_samples;
    unsigned long long seed;
    int noise_q8;
    /* fleets whose dongles do not share a channel plan (bench.py --distinct-plans): dongle `dev` belongs to plan p = dev mod n_plans, and its carrier c sits
     * ((p >> 2c) & 3) * plan_shift_step further up than the table says -- 4^8 = 65 536 distinct plans of eight carriers; n_plans <= 1: every dongle as the table says */
    int n_plans;
    unsigned plan_shift_step; /* u32 turns per sample */
};

void launch_channelizer_fft(const ChannelizerArgs& a, hipStream_t stream);
size_t fft_lds_bytes(int fft_log, int hop_samples, int bytes_per_sample); /* dynamic LDS per workgroup: prepare() refuses configurations above the CU's 160 KiB, the launch opts in above 64 KiB */
bool dft_supported(int fft_size, int hop_bytes, int sfmt, int max_ch);
int dft_lds_per_buf(int hop_bytes, int win_bytes, int np);
int dft_sub(int hop_bytes, int win_bytes, int np);
int dft_nbuf(int hop_bytes, int win_bytes, int np);
int dft_partial_tiles(int n_hops_max); /* 16-hop tiles a work item may touch in one launch (sizes DftArgs::partial) */
void launch_channelizer_dft(const DftArgs& a, hipStream_t stream);
/* side: 3 extra streams, ev: 4 events (fork + 3 joins); both may be null -> everything on `stream`, one kind after the other */
void launch_demod(const DemodArgs& a, const int* kind_first_block, const int* kind_n_blocks, hipStream_t stream, hipStream_t* side, hipEvent_t* ev);
/* waves per workgroup = pieces the contraction index (2 fft_size values) is cut into: four up to fft_size 512 (64 / 32 resident B registers per wave), eight for 1024 and 2048
 * (64 / 128): a workgroup of eight waves is two per SIMD, which is what 128 B registers beside everything else allow anyway */
inline int f32_nw(int fft_size) { return fft_size <= 512 ? 4 : 8; }
/* fft_size 4096 / 8192: the window in segments of 2 048 samples, each contracted by the fft 2048 kernel with the segment's own table (one launch per segment) */
inline int f32_seg_size(int fft_size) { return fft_size < 2048 ? fft_size : 2048; }
inline int f32_n_seg(int fft_size) { return fft_size / f32_seg_size(fft_size); }
int f32_partial_tiles(int n_hops_max); /* 16-hop tiles a work item may touch in one launch (sizes F32Args::partial) */
/* layout of the staged image (channelizer_f32.hip): 1 = hops of an odd number of samples (no padding, fragments from two 8-byte reads), 3 = hops of an odd number of 16-byte units (no padding), 2 = 16 bytes of padding per 256 stream
 * bytes (hops that are multiples of 256 bytes; offsets become immediates), 0 = 16 bytes per hop where the hop is an even number of 16-byte units.  fft 512 with the small tiles
 * of WAVE_RATE 16000 stays on 0: layout 2's 6 % more LDS would cost it its third workgroup per CU (measured: 22.1 -> 24.1 ms; every other variant gains, fft 2048 27 %) */
inline int f32_layout(int fft_size, int hop_bytes) {
    if (hop_bytes & 8) return 1;
    if ((hop_bytes / 16) & 1) return 3; /* an odd number of 16-byte units per hop (2.4 MS/s: 1 200 / 2 400 bytes): the rows fall in different bank groups as they are -- no padding, offsets are immediates */
    if (hop_bytes % 256) return 0;
    const bool small_tile = 15 * hop_bytes + 8 * f32_seg_size(fft_size) <= 6 * 64 * f32_nw(fft_size) * 16;
    return (fft_size == 512 && small_tile) ? 0 : 2;
}
bool f32_supported(int fft_size, int hop_samples, int sfmt);
int f32_pad_bytes(int hop_samples);
int f32_lds_per_buf(int fft_size, int hop_samples);
void launch_channelizer_f32(const F32Args& a, hipStream_t stream);
void launch_emit_iq(const EmitArgs& a, hipStream_t stream);
void launch_axc(const ChanConst* cc, const ChanState* cs, const int* slot_to_ext, uint8_t* out_axc, int n_slots, hipStream_t stream);
void launch_mix(const MixArgs& a, hipStream_t stream);
/* dst += src for the mixer sums of two handles on one GPU (airband_hip_add_mixers): left, right [n_mixers][wave_batch], signal flags OR-ed */
void launch_mix_add(float* dl, float* dr, uint8_t* ds, const float* sl, const float* sr, const uint8_t* ss, int n_mixers, int wave_batch, hipStream_t stream);
void launch_stats(const ChanConst* cc, const ChanState* cs, const int* slot_to_ext, int n_slots, airband_hip_channel_stats* out, hipStream_t stream);
void launch_siggen(const SiggenArgs& a, hipStream_t stream);
void launch_afc(const ChanConst* cc, ChanState* cs, const float* spectrum, int fft_size, int n_slots, int* moved_epoch, int epoch, hipStream_t stream);
/* matrix-core channelizer + AFC: (re)builds, in place, the coefficient columns of every channel whose table is not built for the bin the channel
 * is tuned to now -- all columns of a private table at start-up (bset_bin = -1), the two columns of a channel AFC has just moved afterwards */
struct RetuneArgs {
    const ChanConst* cc;
    const ChanState* cs;
    const DevConst* dev;
    const int* ext_to_slot;
    const int* item_dev;
    const int* item_group;
    int* item_bset;         /* what the channelizer reads: item_home while every channel of the group sits on its base bin, item_private otherwise */
    const int* item_private;/* the group's own table (>= n_shared), or its shared one for groups without an AFC channel */
    const int* item_home;
    int* bset_bin;          /* [n_bsets][8] */
    int8_t* bfrag;
    double* corr;
    float* ftab;            /* CF32 handles (channelizer_f32.hip): the float tables instead of bfrag / corr (both null then); else null */
    const float* window;    /* fft_size */
    int n_items, fft_size, n_shared;
    const int* moved_epoch; /* the batch number afc_kernel stamped when it last moved a channel; the kernel works only if it equals `epoch` (start-up build: both 0) */
    int epoch;
};
void launch_retune(const RetuneArgs& a, hipStream_t stream);
/* shared coefficient tables [first, first + n) built on the device, every column from bset_bin[table][8] (the host builds the first few thousand of a fleet's distinct
 * channel plans, params.cpp; the rest here, at prepare() time) */
void launch_build_tables(int8_t* bfrag, double* corr, const float* window, const int* bset_bin, int first, int n, int fft_size, hipStream_t stream);
/* scatter channel-major host-provided bins into the time-major rings (airband_hip_process_bins) */
void launch_scatter_bins(const float* wavein, const float* iqin, const int* slot_to_ext, const ChanConst* cc, float* mag, float2* iq, int n_slots,
                         int wave_batch, int row0, int ring_rows, hipStream_t stream);
/* the batch's new stage-1 rows (and squelch trace) of channels [first, first + n), channel-major (airband_hip_read_bins / read_trace);
 * |bin| of NFM channels, which stage 1 does not store, is recomputed from the raw bin I/Q the way stage 2 does */
void launch_gather_channels(const float* mag, const float2* iq, const uint8_t* trace, const int* ext_to_slot, const ChanConst* cc, int first, int n, float* wavein,
                            float* iqin, uint8_t* trace_out, int wave_batch, int row0, int ring_rows, hipStream_t stream);

}  // namespace airband
#endif
