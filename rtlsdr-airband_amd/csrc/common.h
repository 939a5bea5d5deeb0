/* csrc/common.h -- data model shared by the host side (params.cpp, airband_hip.cpp) and the HIP kernels.
 *
 * Vocabulary follows the reference: a "dongle" is a device_t (reference: src/rtl_airband.h:249-272), a
 * "channel" is a channel_t with its single freq_t in multichannel mode (:223-247,:201-221), a "hop" is one
 * slide of the FFT window = one output audio sample (src/rtl_airband.cpp:394,669), a "batch" is WAVE_BATCH
 * hops (src/rtl_airband.h:73).
 */
#ifndef AIRBAND_CSRC_COMMON_H
#define AIRBAND_CSRC_COMMON_H

#include <stdint.h>

#define AB_AGC_EXTRA 100  /* AGC_EXTRA, src/rtl_airband.h:74 */
#define AB_SQ_BUF 102     /* Squelch::buffer_size_, src/squelch.cpp:66 */
#define AB_MAX_TONES 52   /* target + 51 standard CTCSS tones, src/ctcss.cpp:101-122 */
#define AB_MAX_CH_PER_DEV 64
#define AB_MIX_RUN 64       /* mixer inputs summed sequentially by one stage-A run */
/* Result rows (channel->waveout): AB_OUT_PAD floats of padding, the AGC_EXTRA carry, then the batch -- so that the batch's
 * samples start on a 128-byte boundary and leave the demod kernels as whole cache lines (AB_OUT_RUN floats per lane) */
#define AB_OUT_RUN 32
#define AB_OUT_PAD (128 - AB_AGC_EXTRA)

/* Squelch::State numeric values (src/squelch.h:102-108) */
enum { AB_ST_CLOSED = 0, AB_ST_OPENING = 1, AB_ST_CLOSING = 2, AB_ST_ABORT = 3, AB_ST_OPEN = 4 };

/* ChanConst.flags */
#define AB_F_NOTCH 0x1u
#define AB_F_LOWPASS 0x2u
#define AB_F_CTCSS 0x4u
#define AB_F_MANUAL 0x8u
#define AB_F_RAW_IQ 0x10u
#define AB_F_IQ_OUT 0x20u
#define AB_F_NFM 0x40u
#define AB_F_QUADRI 0x80u
#define AB_F_VALID 0x100u

/* Per-channel constants (derived once by params.cpp the way src/config.cpp does). */
struct ChanConst {
    uint32_t flags;
    int32_t dev;        /* dongle index */
    int32_t chan;       /* channel index inside the dongle */
    int32_t ext_index;  /* device-major external channel number */
    int32_t base_bin;   /* dev->base_bins[i], src/config.cpp:666-667 */
    int32_t afc;        /* channel_t.afc */
    uint32_t dm_dphi;   /* src/config.cpp:679-712 */
    float alpha;        /* NFM de-emphasis, src/rtl_airband.cpp:87, src/config.cpp:648,775 */
    float ampfactor;
    float notch_d0, notch_d1, notch_d2;        /* src/filters.cpp:41-47 */
    float lp_gain, lp_yc0, lp_yc1;             /* src/filters.cpp:90-96 */
    float sq_manual_level;                     /* Squelch::manual_signal_level_ */
    float sq_normal_ratio, sq_flappy_ratio;    /* src/squelch.cpp:93-103 */
    int32_t ct_slot;                           /* index into the CTCSS tone tables, -1 = none */
    int32_t ct_ntones[2];                      /* [0] fast detector, [1] slow detector */
    int32_t ct_window[2];
    float lp_rgain;     /* RN(1 / lp_gain) when x * r corrected once is the correctly rounded x / lp_gain for every x (params.cpp, div_const_reciprocal), else 0 */
    int32_t pad[1];
};

/* Per-channel mutable state that survives from batch to batch. */
struct ChanState {
    /* freq_t / channel_t */
    float agcavgfast, pr, pj, prev_waveout;
    uint32_t dm_phi;
    int32_t bin; /* dev->bins[i]; moves only with AFC */
    int32_t axc; /* channel->axcindicate of the last batch */
    uint32_t active_counter;
    /* Squelch (src/squelch.h:117-158) */
    float noise_floor, cap, pre_full, pre_capped, post_full, post_capped, level_cache;
    int32_t using_post, next, cur, delay, low_count, head, tail;
    uint32_t sample_count; /* only (count % 16) is observable; starts at 0xffffffff like size_t(-1) */
    uint32_t open_count, flappy_count, recent_open, closed_count;
    /* NotchFilter / LowpassFilter delay lines */
    float nx[3], ny[3];
    float lxr[3], lxi[3], lyr[3], lyi[3]; /* ([0] of each is not kept: LowpassFilter::apply overwrites it before reading it) */
    /* CTCSS detectors: [0] fast, [1] slow (src/ctcss.h:84-95) */
    int32_t ct_enough[2], ct_count[2], ct_has_tone[2];
    uint32_t ct_found[2], ct_not_found[2];
    int32_t axc_prev; /* axcindicate before the last batch (what `AFC afc(dev, i)` captures, src/rtl_airband.cpp:222,496) */
    /* the pre-filter average and noise floor as they stood 101 samples ago (squelch_fsm.h, SqShadow): the NFM + lowpass kind recomputes the
     * squelch's delay line from them instead of storing and re-reading it */
    float sh_nf, sh_cap, sh_capped;
    /* what the channel's result row holds: bit 0 = its AGC_EXTRA carry is all zeros, bit 1 = its batch area is -- a channel that stays closed
     * leaves both alone instead of rewriting 8 KiB of zeros per batch (demod.hip, RowZero) */
    int32_t row_zero;
    /* the delay-line entry the batch's LAST sample saw (SqRegs::dly): update_current_state() tests the post-filter gate against buffer_[buffer_tail_]
     * before the tail moves (src/squelch.cpp:390,407,467), so the first sample of the next batch needs it once more -- the shadow itself has stepped on */
    float sh_dly;
    int32_t pad[1];
};

/* Per-dongle constants for the channelizer. */
struct DevConst {
    int32_t sfmt;          /* AIRBAND_SFMT_* */
    int32_t bytes_per_sample;
    int32_t hop_samples;   /* round(sample_rate / WAVE_RATE), src/rtl_airband.cpp:394 */
    int32_t n_ch;
    int32_t chan_base;     /* first internal slot of this dongle's channels */
    float scale;           /* 1/fullscale for S16/F32 (src/rtl_airband.cpp:403,421) */
    int32_t any_raw_iq;
    int32_t disabled;      /* airband_hip_device_enable(h, dev, 0): both stages skip the dongle (a failed input, src/rtl_airband.cpp:377-391) */
    int32_t any_afc;       /* some channel of the dongle has afc != 0: the spectrum of each batch's last hop is needed (src/rtl_airband.cpp:626-630) */
    int32_t pad;
};

/* scale of the matrix-core channelizer's 24-bit coefficient tables (params.cpp builds them, misc_kernels.hip re-tunes a column when AFC moves a bin):
 * max |coefficient| < 1  ->  |value| <= 127 * 65536 + 127 * 256 + 127 */
#define AB_DFT_COEF_SCALE 8355000.0

/* Demod kinds: slots are sorted so that the 64 lanes of a demod wavefront run the same code path. */
enum { AB_KIND_AM = 0, AB_KIND_NFM = 1, AB_KIND_NFM_LOWPASS = 2, AB_KIND_NFM_CTCSS = 3, AB_KIND_GENERIC = 4, AB_KIND_COUNT = 5 };

/* Stage-1/stage-2 exchange buffers are blocked time-major rings: element (row, slot) of a buffer with `rows` rows
 * lives at ((slot / 64) * rows + row) * 64 + slot % 64.  One demod wavefront (64 slots) therefore streams ONE
 * contiguous region row by row (256 bytes per row), and a channelizer wavefront writes its dongle's few slots at a
 * 256-byte row stride inside that same region instead of at a multi-megabyte stride. */
#define AB_SLOT_BLOCK 64
#if defined(__HIPCC__)
#define AB_HD __host__ __device__
#else
#define AB_HD
#endif
static inline AB_HD long ab_ring_base(int slot, int rows) { return ((long)(slot >> 6) * rows) * AB_SLOT_BLOCK + (slot & 63); }

/* The two stage-1 -> stage-2 rings (|bin| and raw bin I/Q) are additionally transposed inside tiles of AB_TILE_ROWS rows:
 * element (row, slot) lives at ((slot/64 * tiles + row/T) * 64 + slot%64) * T + row%T, tiles = ring_rows / T.
 * The matrix-core channelizer produces 16 hops x 16 columns per MFMA tile with 4 consecutive hops per lane, so a lane
 * stores 16 contiguous bytes and the lanes of a column complete a contiguous run per slot; the demod kernels fetch the same
 * tiles back 16 bytes (4 rows) per lane.  T = 8: a slot's tile is 32 bytes of |bin| / 64 bytes of raw I/Q, the four AM (two
 * NFM) channels of a dongle are adjacent slots, so the channelizer still completes whole 128-byte lines per dongle, and a demod
 * wavefront consumes every line it touches within one 8-sample group (T = 16 left half of each line for a later group: by then
 * it had left the caches, and stage 2 fetched 33 GB per batch instead of 17). */
#ifndef AB_TILE_ROWS
#define AB_TILE_ROWS 8 /* 4, 8 or 16 */
#endif
static inline AB_HD long ab_tile_base(int slot, int tiles) { return (((long)(slot >> 6) * tiles) * AB_SLOT_BLOCK + (slot & 63)) * AB_TILE_ROWS; }
/* row >= 0 and the offset stays far below 2^31 elements (ring_rows * 64): unsigned shifts and masks, no signed-division fix-ups and no
 * 64-bit scalar arithmetic -- the demod kernels compute four of these per group of four samples */
static inline AB_HD int ab_tile_off(int row) {
    const unsigned r = (unsigned)row;
    return (int)((r / (unsigned)AB_TILE_ROWS) * (unsigned)(AB_SLOT_BLOCK * AB_TILE_ROWS) + (r % (unsigned)AB_TILE_ROWS));
}

#endif
