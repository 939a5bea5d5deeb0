/* csrc/demod.hip -- stage 2 of the hot path on gfx950: the per-channel sequential loop of demodulate()
 * (reference: src/rtl_airband.cpp:494-648) with everything it calls: Squelch (src/squelch.cpp), CTCSS Goertzel
 * banks (src/ctcss.cpp), NotchFilter / LowpassFilter (src/filters.cpp), sincosf_lut (src/util.cpp:113-127) and
 * the FM helpers (src/rtl_airband.cpp:141-176).
 *
 * Mapping: one wavefront owns 64 consecutive channel slots of ONE demod kind, one lane per (dongle, channel).  Time is
 * strictly serial per channel (IIR / EMA / FSM state), so a lane walks the batch's WAVE_BATCH samples in order with all of
 * its state in registers.  The batch is walked four samples (a "quad") at a time:
 *   input    a quad's stage-1 values are one or two 16-byte loads per stream out of the lane's own 16-row ring tile; two
 *            register sets alternate, the loads of quad k+1 are in flight while the per-sample code works through quad k;
 *   step     squelch state machine (squelch_fsm.h: lane masks in scalar registers) + derotation / lowpass + AM AGC or FM
 *            discriminator, then -- fused kinds -- output gating, notch, ampfactor, clamp, AM fade-out.  Audio is parked in LDS
 *            and leaves every 32 samples as whole 128-byte lines of the channel's channel->waveout-shaped result row.
 * Kinds that can carry a CTCSS tone are split in three kernels instead: this file's demod_kernel<.., true> is the FRONT
 * (squelch + discriminator -> pre-notch audio + flag word per sample, channel-major hand-off buffer), tone_kernel turns the
 * problem 90 degrees (one wavefront per channel, lane t = tone t of the fast and the slow Goertzel bank: the reference's
 * ~104 multiply-adds per audio sample are one 3-op recurrence per lane), back_kernel gates (squelch open AND tone present),
 * notches and writes the result rows.
 *
 * This file MUST be compiled with -ffp-contract=off: squelch decisions have to be bit-identical to the
 * reference's scalar float code given the same stage-1 input, so no FMA contraction, IEEE divide and sqrt
 * (-fhip-fp32-correctly-rounded-divide-sqrt) and the reference's operation order everywhere (including the
 * index-order float sum of tone powers in the CTCSS decision).
 */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "squelch_fsm.h"
#include "exact_math.h"

namespace airband {

namespace {

/* "These vector registers are needed now": an empty asm statement that names them, so that the compiler's wait for the loads that fill them lands
 * HERE (see demod_wave: before the next group's loads are issued).  The file also compiles as plain C++ for tests/host_demod_harness.cpp, where there
 * is nothing to wait for. */
#if !defined(AB_NEEDED_NOW)
#if defined(__HIPCC__)
#define AB_NEEDED_NOW(...) asm volatile("" ::__VA_ARGS__)
#define AB_V(x) "v"(x)
#else
#define AB_NEEDED_NOW(...) ((void)0)
#define AB_V(x) 0
#endif
#endif
/* a point the compiler's scheduler does not move instructions across (nothing in the ISA) */
#if !defined(AB_SCHED_FENCE)
#if defined(__has_builtin) && __has_builtin(__builtin_amdgcn_sched_barrier)
#define AB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define AB_SCHED_FENCE() ((void)0)
#endif
#endif
/* The 64 lanes of a wavefront execute in lockstep, and LDS operations of one wavefront complete in order: where lanes exchange data through LDS
 * WITHOUT a barrier (the cooperative row stores, the tone kernel's power sum) the code relies on that.  AB_LOCKSTEP() marks those places; it is
 * nothing on the GPU.  tests/hostshim_wave64 runs the lanes as fibers and makes them meet there. */
/* AB_LOCKSTEP_FENCED(): the same rendezvous with the COMPILER's order pinned as well -- a wavefront-scope fence and a wave barrier, no instruction in the ISA -- where a
 * value parked in LDS by one lane is read back by OTHER lanes (the tone kernel's broadcasts): without it the order of the park and the reads rests on the compiler's
 * may-alias analysis alone (round-5 ADVICE; channelizer_fft.hip's AB_WAVE_SYNC pins its exchanges the same way). */
#if !defined(AB_LOCKSTEP)
#define AB_LOCKSTEP() ((void)0)
#if defined(__HIPCC__)
#define AB_LOCKSTEP_FENCED()                                     \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#else
#define AB_LOCKSTEP_FENCED() ((void)0)
#endif
#elif !defined(AB_LOCKSTEP_FENCED)
#define AB_LOCKSTEP_FENCED() AB_LOCKSTEP()
#endif
/* the kernel's dynamic LDS array */
#if !defined(AB_DYNAMIC_LDS)
#define AB_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

/* per-sample flag word parked in LDS between the phases */
constexpr unsigned FL_AUDIO = 1u;   /* Squelch::should_process_audio()                        */
constexpr unsigned FL_FADE = 2u;    /* AM last_open_sample(): fade out the previous AGC_EXTRA */
constexpr unsigned FL_RESET = 4u;   /* squelch went CLOSED on this sample: CTCSS::reset()      */
constexpr int FL_STATE_SHIFT = 3;   /* bits 3..5: Squelch::State, for the trace                */
/* one-word hand-off of the NFM + CTCSS kind (see demod_wave): */
constexpr unsigned HAND_NAN = 0x7FC00000u;  /* an audio sample that is a NaN                                          */
constexpr unsigned HAND_IDLE = 0xFFC00000u; /* no audio on this sample; | 1: the squelch went CLOSED on it            */
__device__ __forceinline__ bool hand_is_audio(unsigned w) { return (w >> 1) != (HAND_IDLE >> 1); }

/* fast_atan2 / polar_disc_fast / fm_quadri_demod (src/rtl_airband.cpp:141-176) */
__device__ __forceinline__ float fast_atan2_dev(float y, float x) {
    /* The reference picks one of  pi/4 - pi/4 * (x - |y|) / (x + |y|)   (x >= 0)
     *                             3pi/4 - pi/4 * (x + |y|) / (|y| - x)   (x <  0).
     * The sign of x differs from lane to lane, so both arms would run under masks -- two IEEE divisions.  Selecting the
     * operands instead leaves one: same operations on the same values in either case (|y| - x is the exact negation of
     * x - |y|), hence the same float. */
    const float pi4 = (float)0.78539816339744830962, pi34 = (float)(3 * 0.78539816339744830962);
    const float ay = y < 0.0f ? -y : y;
    const float diff = x - ay, sum = x + ay;
    const bool right = x >= 0.0f;
    const float num = right ? diff : sum;
    const float den = right ? sum : -diff;
    const float base = right ? pi4 : pi34;
    const float a = base - pi4 * num / den;
    const float r = y < 0.0f ? -a : a;
    return (x == 0.0f && y == 0.0f) ? 0.0f : r;
}

__device__ __forceinline__ int ring_row(int r, int R) { return r >= R ? r - R : r; }

/* Output path of one sample (src/rtl_airband.cpp:532-547 fade-out, :589-620 gating): shared by the fused loop and the back kernel */
struct OutRegs {
    float nx0, nx1, nx2, ny0, ny1, ny2; /* NotchFilter delay line */
    int axc;
};

/* Where a lane's audio goes: its channel's row of the result buffer, laid out like the reference's channel->waveout
 * (src/rtl_airband.h:230): row[k], k in [0, WAVE_BATCH + AGC_EXTRA); sample j of the batch lands at k = j + AGC_EXTRA, the
 * consumer reads [0, WAVE_BATCH), and the last AGC_EXTRA entries are copied to the front when the next batch starts -- the
 * output thread's tail copy (src/output.cpp:920).  Samples leave in runs of RUN through the lane's LDS column (stride 64
 * floats): a lane stores whole 128-byte lines of its own row (rows are padded so that k = AGC_EXTRA is line-aligned). */
constexpr int RUN = AB_OUT_RUN; /* 32 floats = one 128-byte cache line of the lane's row per flush (64-byte runs: 10.6 ms of stage 2, 32-byte runs: 11.5 ms, whole lines: 9.9 ms) */
constexpr int OSTRIDE = 65; /* floats between two samples of the LDS staging area [RUN][OSTRIDE]: odd, so that both the per-lane writes and the
                               transposed reads of the cooperative flush spread over the banks */
struct WaveRow {
    float* row;
    float* staged; /* LDS, element i of the current run at staged[i * stride] */
    int stride;
    int j0;        /* first sample of the current run */
    /* cooperative flush (blocks whose 64 lanes all own a channel): eight lanes store one channel's whole 128-byte line, so a
     * store instruction is eight full lines instead of 64 sixteen-byte pieces of 64 different lines */
    const float* stage_base; /* LDS [RUN][OSTRIDE] */
    const int* ext_of;       /* LDS [64]: external channel index of the block's lanes */
    float* out_wave;
    int wave_stride;
    bool coop;
    int* skip_of;            /* LDS [64]: this run of the lane's channel is zeros over zeros (cooperative flush: the storing lanes look it up) */
};

/* Rows that stay zero are left alone.  A channel whose squelch is closed writes 0.0f for every sample -- 8 KiB of zeros per batch over the zeros of the
 * batch before, plus the tail copy of zeros onto zeros: 40 % of the audio traffic at the BASELINE duty cycle.  ChanState::row_zero remembers what the
 * row holds (bit 0: the AGC_EXTRA carry is all zeros, bit 1: the batch area is); a run of samples during which the channel was never open is skipped when
 * the batch area was all zeros already, the tail copy when both are.  "Never open" means every sample was the literal 0.0f of the closed branch
 * (emit_sample); an open sample that happens to be 0 counts as content. */
struct RowZero {
    int held;        /* ChanState::row_zero as loaded (after the tail copy: bit 0 = bit 1) */
    bool run_open;   /* the channel was open somewhere in the run being staged */
    bool batch_open; /* ... somewhere in the batch */
};
__device__ __forceinline__ bool row_skip_run(const RowZero& z) { return (z.held & 2) && !z.run_open; }
__device__ __forceinline__ void row_run_done(RowZero& z) {
    z.batch_open = z.batch_open || z.run_open;
    z.run_open = false;
}

/* AGC_EXTRA = 100 floats in rounds of five 16-byte pieces: all 25 pieces at once were 100 live registers in front of the sample loop --
 * they set the kernels' register counts (back kernel 126 -> 96, AM kind 121 -> 108, NFM + lowpass 166 -> 157) and the back kernel staged
 * them through scratch memory */
__device__ __forceinline__ void wave_tail_copy(float* row, int B) {
    typedef float v4f __attribute__((ext_vector_type(4))); /* (plain vector values: an array of float4 here ends up as dead stores to scratch memory) */
    const v4f* src = reinterpret_cast<const v4f*>(row + B);
    v4f* dst = reinterpret_cast<v4f*>(row);
    static_assert((AB_AGC_EXTRA / 4) % 5 == 0, "whole rounds");
#pragma unroll 1
    for (int c = 0; c < AB_AGC_EXTRA / 4; c += 5) {
        const v4f t0 = src[c], t1 = src[c + 1], t2 = src[c + 2], t3 = src[c + 3], t4 = src[c + 4];
        dst[c] = t0; dst[c + 1] = t1; dst[c + 2] = t2; dst[c + 3] = t3; dst[c + 4] = t4;
    }
}

__device__ __forceinline__ void row_tail_copy(RowZero& z, float* row, int B) { /* src/output.cpp:920, unless it would copy zeros onto zeros */
    if ((z.held & 3) != 3) wave_tail_copy(row, B);
    z.held = (z.held & 2) ? 3 : 0; /* the carry now holds the previous batch's tail: zeros if that batch was all zeros, else unknown */
}

__device__ __forceinline__ void wave_flush(const WaveRow& w, RowZero& z, int n = RUN) { /* the finished run (n samples) -> row[j0 + AGC_EXTRA ...): 16-byte aligned by construction */
    const bool skip = row_skip_run(z);
    row_run_done(z);
    if (w.coop) { /* wave-uniform */
        const int lane = threadIdx.x & 63, q = lane & 7;
        AB_LOCKSTEP(); /* every lane has staged its run */
        w.skip_of[lane] = skip ? 1 : 0; /* (LDS operations of one wave complete in order: the lanes that store this channel's line read it below) */
        AB_LOCKSTEP();
        if (4 * q < n) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int c = i * 8 + (lane >> 3);
                if (w.skip_of[c]) continue;
                const float* src = w.stage_base + (4 * q) * OSTRIDE + c;
                float4* dst = reinterpret_cast<float4*>(w.out_wave + (long)w.ext_of[c] * w.wave_stride + AB_OUT_PAD + AB_AGC_EXTRA + w.j0 + 4 * q);
                *dst = make_float4(src[0], src[OSTRIDE], src[2 * OSTRIDE], src[3 * OSTRIDE]);
            }
        }
        AB_LOCKSTEP(); /* ... before any lane stages its next run over it */
        return;
    }
    if (skip) return;
    float4* dst = reinterpret_cast<float4*>(w.row + AB_AGC_EXTRA + w.j0);
#pragma unroll
    for (int q = 0; q < RUN / 4; q++)
        if (4 * q < n)
        dst[q] = make_float4(w.staged[(4 * q) * w.stride], w.staged[(4 * q + 1) * w.stride], w.staged[(4 * q + 2) * w.stride], w.staged[(4 * q + 3) * w.stride]);
}

__device__ __forceinline__ void emit_sample(const DemodArgs& a, const ChanConst& cc, OutRegs& o, const WaveRow& w, RowZero& z, float2* iqout, uint8_t* trace, int j, bool audio, bool fade,
                                            bool tone, int state, float out, float re, float im, bool write_iq_always) {
    constexpr long S = AB_SLOT_BLOCK;
    if (AB_UNLIKELY(fade)) { /* AM, squelch just closing: waveout[k] = waveout[k-1] * 0.94 over the previous AGC_EXTRA-1 samples */
        z.run_open = true; /* rewrites samples of the run being staged (and of runs that are in the row already) */
        float prev = w.row[j]; /* = output of sample j - AGC_EXTRA: left its run long ago */
#pragma nounroll
        for (int k = j + 1; k < j + AB_AGC_EXTRA; k++) {
            prev = prev * 0.94f;
            const int i = k - AB_AGC_EXTRA - w.j0; /* position in the run still parked in LDS, if it is that recent */
            if (i >= 0)
                w.staged[i * w.stride] = prev;
            else
                w.row[k] = prev;
        }
    }
    const bool open = audio && tone; /* Squelch::is_open (src/squelch.cpp:118-134) */
    if (open) {
        if (cc.flags & AB_F_NOTCH) { /* NotchFilter::apply (src/filters.cpp:50-64) */
            o.nx0 = o.nx1; o.nx1 = o.nx2; o.nx2 = out;
            o.ny0 = o.ny1; o.ny1 = o.ny2;
            o.ny2 = cc.notch_d0 * o.nx2 - cc.notch_d1 * o.nx1 + cc.notch_d0 * o.nx0 + cc.notch_d1 * o.ny1 - cc.notch_d2 * o.ny0;
            out = o.ny2;
        }
        out *= cc.ampfactor;
        /* NaN -> 0, else clamp to [-1, 1] (src/rtl_airband.cpp:597-603): the median of (out, -1, 1) IS the clamp -- one of its
         * inputs, unchanged -- in a single instruction */
        out = (out != out) ? 0.0f : __builtin_amdgcn_fmed3f(out, -1.0f, 1.0f);
        o.axc = '*';
        z.run_open = true;
    } else {
        out = 0.0f;
    }
    w.staged[(j - w.j0) * w.stride] = out;
    if (cc.flags & AB_F_IQ_OUT) {
        if (open) {
            if (write_iq_always) iqout[(long)j * S] = make_float2(re, im);
        } else {
            iqout[(long)j * S] = make_float2(0.0f, 0.0f);
        }
    }
    if (trace) trace[(long)j * S] = (uint8_t)((state & 7) | (open ? 8 : 0) | (audio ? 16 : 0) | (((cc.flags & AB_F_CTCSS) && tone) ? 32 : 0));
}

/* Feature bits that are compile-time constants inside a specialised kind (everything else stays a run-time flag) */
constexpr unsigned KIND_MASK = AB_F_NFM | AB_F_RAW_IQ | AB_F_LOWPASS | AB_F_CTCSS | AB_F_IQ_OUT;
template <int KIND>
struct KindBits {
    static constexpr unsigned value = KIND == AB_KIND_NFM           ? (AB_F_NFM | AB_F_RAW_IQ)
                                      : KIND == AB_KIND_NFM_LOWPASS ? (AB_F_NFM | AB_F_RAW_IQ | AB_F_LOWPASS)
                                      : KIND == AB_KIND_NFM_CTCSS   ? (AB_F_NFM | AB_F_RAW_IQ | AB_F_CTCSS)
                                                                    : 0u;
};

template <int KIND, bool WAVE_HAS_CTCSS, int W>
__device__ __forceinline__ void demod_wave(const DemodArgs& a, ChanConst cc, ChanState* sp, int slot, const float2* lut, float* ostage, const int* ext_of, int* skip_of, const int* slot_of,
                                           bool full_block) {
    const int lane = threadIdx.x & 63;
    constexpr long S = AB_SLOT_BLOCK;
    const int R = a.ring_rows, B = a.wave_batch;
    const bool valid = (cc.flags & AB_F_VALID) != 0;
    /* inside a specialised kind the modulation / filter feature bits are constants: the compiler drops the other paths */
    if (KIND != AB_KIND_GENERIC) cc.flags = (cc.flags & ~KIND_MASK) | KindBits<KIND>::value;

    if (!valid) return; /* padding slots of the last block: lanes share nothing (no barriers, per-lane LDS columns) */

    const bool nfm = cc.flags & AB_F_NFM, raw_iq = cc.flags & AB_F_RAW_IQ, lowpass = cc.flags & AB_F_LOWPASS;
    /* feature bits as lane masks (compile-time all-or-nothing inside a specialised kind) */
    const lmask m_am = ab_ballot(!nfm), m_raw_iq = ab_ballot(raw_iq);

    Lane L;
    L.m_lowpass = ab_ballot(lowpass);
    L.m_manual = ab_ballot((cc.flags & AB_F_MANUAL) != 0);
    L.manual_level = cc.sq_manual_level;
    L.normal_ratio = cc.sq_normal_ratio;
    L.flappy_ratio = cc.sq_flappy_ratio;
    L.m_flappy_lower = ab_ballot(cc.sq_flappy_ratio < cc.sq_normal_ratio);
    L.sqbuf = a.sqbuf + ab_ring_base(slot, AB_SQ_BUF);
    L.S = S;
    /* NFM+lowpass kind: the 102-deep delay line is read 101 samples after it is written, so a chunk's reads can be
     * fetched up front with the other inputs (LDS slot .y, unused by NFM) instead of one dependent L2 round trip per sample */
    L.prefetched_delay = (KIND == AB_KIND_NFM_LOWPASS);
    /* only channels with a lowpass filter ever touch the delay line: the other kinds move head/tail once per batch */
    L.track_delay_line = (KIND == AB_KIND_NFM_LOWPASS || KIND == AB_KIND_GENERIC);
    L.may_post_filter = (KIND == AB_KIND_NFM_LOWPASS || KIND == AB_KIND_GENERIC);
    L.all_lowpass = (KIND == AB_KIND_NFM_LOWPASS);
    /* ... and that kind no longer stores the delay line at all: a shadow of the pre-filter average, fed the AGC_EXTRA-delayed input, holds its entries
     * (squelch_fsm.h, SqShadow): 8 bytes of traffic per sample and channel less, 2.1 GB per batch at BASELINE configs[2] */
    L.shadow_delay = (KIND == AB_KIND_NFM_LOWPASS);

    SqRegs s;
    sq_load(s, L, sp, true);
    /* sample_count_ starts at -1 (src/squelch.cpp:58) and every batch is a multiple of four samples long, so the noise floor's every-16th
     * sample is always the first one of a group of four: the other three do not look (squelch_fsm.h, sq_raw_quiet).  Should a handle ever
     * hold a count that is not aligned like that (wave-uniform test), every sample looks. */
    const bool aligned4 = ((s.sample_count + 1u) & 3u) == 0u;
    SqShadow sh = {sp->sh_nf, sp->sh_cap, sp->sh_capped};
    const bool first_batch = a.tail_copy == 0; /* the stream's first 101 samples read the zeros the reference's delay line starts with (src/squelch.cpp:69) */
    /* buffer_[buffer_tail_] as the previous batch's last sample left it (0 in a fresh buffer): NOT sq_shadow_value(sh), which is already the entry under the
     * tail's next position -- the OPENING / CLOSING timers that run out on a batch's first sample gate on this one (tests/test_host_demod.py, ..._first_sample_of_a_batch...) */
    s.dly = KIND == AB_KIND_NFM_LOWPASS ? sp->sh_dly : 0.0f;
    float agc = sp->agcavgfast, pr = sp->pr, pj = sp->pj, prev_out = sp->prev_waveout;
    unsigned dm_phi = sp->dm_phi;
    OutRegs o;
    o.nx0 = sp->nx[0]; o.nx1 = sp->nx[1]; o.nx2 = sp->nx[2]; o.ny0 = sp->ny[0]; o.ny1 = sp->ny[1]; o.ny2 = sp->ny[2];
    o.axc = ' ';
    /* LowpassFilter's delay lines: only [1] and [2] carry over -- apply() shifts [1] into [0] before it reads [0] (src/filters.cpp:146-163), so [0] is a
     * temporary of the step (four exec-masked moves per sample less than keeping it as state) */
    float lxr1 = sp->lxr[1], lxr2 = sp->lxr[2], lxi1 = sp->lxi[1], lxi2 = sp->lxi[2];
    float lyr1 = sp->lyr[1], lyr2 = sp->lyr[2], lyi1 = sp->lyi[1], lyi2 = sp->lyi[2];

    const float one_minus_alpha = 1.0f - cc.alpha;
    const float div_lo = cc.lp_rgain != 0.0f ? AB_DIV_CONST_LO : __builtin_inff(); /* exact_math.h: the range in which x / lp_gain is three instructions */

    float* mag = a.mag + ab_tile_base(slot, R / AB_TILE_ROWS);        /* tile-transposed rings: row r at ab_tile_off(r) */
    const float2* iqin = a.iq + ab_tile_base(slot, R / AB_TILE_ROWS);
    const int ext = a.slot_to_ext[slot];
    WaveRow wrow;
    wrow.row = a.out_wave + (long)ext * a.wave_stride + AB_OUT_PAD;
    wrow.stride = OSTRIDE;
    wrow.j0 = 0;
    wrow.stage_base = ostage;
    wrow.ext_of = ext_of;
    wrow.out_wave = a.out_wave;
    wrow.wave_stride = a.wave_stride;
    wrow.coop = full_block;
    wrow.skip_of = skip_of;
    lmask audio_seen = 0; /* split kinds: lanes whose squelch let audio through, or went CLOSED, somewhere in this batch (a scalar register pair: no vector work) -- a channel
                           * without either has nothing for the tone kernel to do, and regrouped handles deal the back kernel's slots out by it */
    RowZero rz = {sp->row_zero, false, false};
    float2* iqout = a.iq_out + ab_ring_base(slot, B);
    uint8_t* trace = a.trace ? a.trace + ab_ring_base(slot, B) : nullptr;
    wrow.staged = ostage + lane; /* [RUN][OSTRIDE] floats behind the sincos table */
    if (!WAVE_HAS_CTCSS && a.tail_copy) row_tail_copy(rz, wrow.row, B); /* src/output.cpp:920; the back kernel does it for the split kinds */
    /* split kinds: what the tone / back kernels need of every sample, channel-major [ct slot][sample].
     * NFM + CTCSS (the kind that matters: every CTCSS channel of a plain NFM plan): ONE 32-bit word per sample, HAND_WORD below.
     * Generic kind (AM + CTCSS, raw-I/Q outputs, lowpass + CTCSS): (audio, flags) pairs -- it also has an AM fade-out flag to carry. */
    constexpr bool PACKED = WAVE_HAS_CTCSS && KIND == AB_KIND_NFM_CTCSS;
    float2* ct_af = (WAVE_HAS_CTCSS && !PACKED) ? a.ct_af + (long)(slot - a.ct_gen_first_block * 64) * B : nullptr;
    unsigned* ct_ap = PACKED ? a.ct_ap + (long)(slot - a.ct_pk_first_block * 64) * a.ct_pk_pitch : nullptr; /* rows of whole 128-byte lines */
    /* the front kernel has no audio rows of its own, so the row staging area holds hand-off samples instead: 16 pairs
     * [HAND_RUN][OSTRIDE] float2 or 32 words [RUN][OSTRIDE], flushed like the audio rows (eight lanes store one channel's 128-byte line) */
    constexpr int HAND_RUN = PACKED ? RUN : 16;
    float2* hand_base = reinterpret_cast<float2*>(ostage);
    float2* hand = hand_base + lane;
    unsigned* handw_base = reinterpret_cast<unsigned*>(ostage);
    unsigned* handw = handw_base + lane;
    /* the cooperative flush stores OTHER lanes' rows: row of the wavefront's channel c = the row of slot_of[c] (LDS; the lanes of a regrouped wavefront do not
     * work on consecutive slots) */
    auto ct_row = [&](int c) { return a.ct_af + (long)(slot_of[c] - a.ct_gen_first_block * 64) * B; };
    auto ct_roww = [&](int c) { return a.ct_ap + (long)(slot_of[c] - a.ct_pk_first_block * 64) * a.ct_pk_pitch; };
    auto hand_flush = [&](int n, int jstart) { /* n samples starting at batch sample jstart */
        if (PACKED) {
            if (full_block) { /* wave-uniform */
                const int q = lane & 7; /* samples 4q .. 4q + 3 */
                AB_LOCKSTEP(); /* every lane has parked its words */
                if (4 * q < n) {
#pragma unroll
                    for (int i0 = 0; i0 < 8; i0++) {
#if defined(AB_FLUSH_REVERSED) /* experiment builds (profiles/r05_event_hunt.md): the eight cooperative stores of a flush in the opposite order */
                        const int i = 7 - i0;
#else
                        const int i = i0;
#endif
                        const int c = i * 8 + (lane >> 3);
                        const unsigned* src = handw_base + (4 * q) * OSTRIDE + c;
                        *reinterpret_cast<uint4*>(ct_roww(c) + jstart + 4 * q) = make_uint4(src[0], src[OSTRIDE], src[2 * OSTRIDE], src[3 * OSTRIDE]);
                    }
                }
                AB_LOCKSTEP();
            } else {
                for (int i = 0; i < n; i += 4)
                    *reinterpret_cast<uint4*>(ct_ap + jstart + i) = make_uint4(handw[i * OSTRIDE], handw[(i + 1) * OSTRIDE], handw[(i + 2) * OSTRIDE], handw[(i + 3) * OSTRIDE]);
            }
            return;
        }
        if (full_block) { /* wave-uniform */
            const int q = lane & 7; /* samples 2q, 2q + 1 */
            AB_LOCKSTEP();
            if (2 * q < n) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int c = i * 8 + (lane >> 3);
                    const float2 p0 = hand_base[(2 * q) * OSTRIDE + c], p1 = hand_base[(2 * q + 1) * OSTRIDE + c];
                    *reinterpret_cast<float4*>(ct_row(c) + jstart + 2 * q) = make_float4(p0.x, p0.y, p1.x, p1.y);
                }
            }
            AB_LOCKSTEP();
        } else {
            for (int i = 0; i < n; i += 2) {
                const float2 p0 = hand[i * OSTRIDE], p1 = hand[(i + 1) * OSTRIDE];
                *reinterpret_cast<float4*>(ct_af + jstart + i) = make_float4(p0.x, p0.y, p1.x, p1.y);
            }
        }
    };

    /* ---- input: register ping-pong, a GROUP of 4 samples at a time ----------------------------------------------------------
     * Four samples of a channel are 16 contiguous bytes of its 16-row tile (|bin|) or 32 (raw bin I/Q): one or two 16-byte loads
     * per stream.  Two register sets alternate: the loads of group k+1 are issued as soon as group k's registers have ARRIVED
     * (touch(): the compiler's own wait lands there), and fly while the per-sample code works through group k -- a wave no longer
     * sits out a memory round trip per group -- and nothing is staged through LDS on the way in (round 1 parked 8-sample chunks
     * in LDS, 4-8 KiB per wave, and consumed the loads right where it issued them).
     * row0, AGC_EXTRA and WAVE_BATCH are multiples of 4 and the ring length is a multiple of 16, so four samples are always 16-byte
     * aligned inside one tile and never straddle the ring wrap.  Stage 2 may rewrite a magnitude (AM lanes that need raw I/Q, :524)
     * that is read back AGC_EXTRA = 100 samples later: far outside the 16 samples a prefetch runs ahead. */
    constexpr int GS = 4; /* samples per group; 2 * GS divides WAVE_BATCH = 1000 / 2000 */
    constexpr int GQ = GS / 4;
    /* Experiment builds only (-DAB_MASKED_DELAY, DESIGN 7.1 d; the product's source is token for token what it was): the AGC_EXTRA-delayed values are only USED by lanes
     * whose squelch lets samples through -- the AM kind's delayed magnitude by OPEN / CLOSING lanes (src/rtl_airband.cpp:553-557), the plain NFM and CTCSS-front kinds'
     * delayed raw I/Q by lanes that filter (:510-530).  The fetch stays unconditional (wait pairing, below) but a lane that is CLOSED when the fetch is issued points it at
     * the line it reads anyway (the CURRENT hops), so its piece of the delayed line is not requested from memory.  Group::real says which lanes fetched the real thing; a
     * lane that turns out to need it re-fetches (once per squelch opening and lane).  The lowpass kind feeds its delay-line shadow from the delayed values and the generic
     * kind is rare: both fetch as ever. */
#if defined(AB_MASKED_DELAY)
    constexpr bool MASKD = KIND == AB_KIND_AM || KIND == AB_KIND_NFM || KIND == AB_KIND_NFM_CTCSS;
#define AB_MD_PARAM , const int r
#define AB_MD_ARG(r) , r
#else
#define AB_MD_PARAM
#define AB_MD_ARG(r)
#endif
    struct Group {
        float4 mc[GQ], md[GQ], c01[GQ], c23[GQ], q01[GQ], q23[GQ];
#if defined(AB_MASKED_DELAY)
        lmask real; /* lanes whose delayed values are the real ones */
#endif
    };
    auto fetch = [&](Group& q, int j0, int tail0) { /* tail0 = squelch delay-line tail at the start of the group (lowpass kind) */
#if defined(AB_MASKED_DELAY)
        const lmask need = MASKD ? (~s.cC & s.active) : ~(lmask)0;
        const bool real_lane = !MASKD || ab_lane(need);
        q.real = need;
#endif
#pragma unroll
        for (int g = 0; g < GQ; g++) {
            const int rc = ring_row(a.row0 + AB_AGC_EXTRA + j0 + 4 * g, R); /* current hops */
#if defined(AB_MASKED_DELAY)
            const int rd = real_lane ? ring_row(a.row0 + j0 + 4 * g, R) : rc;
#else
            const int rd = ring_row(a.row0 + j0 + 4 * g, R);                /* hops AGC_EXTRA earlier */
#endif
            if (!nfm) {
                q.mc[g] = *reinterpret_cast<const float4*>(mag + ab_tile_off(rc));
                q.md[g] = *reinterpret_cast<const float4*>(mag + ab_tile_off(rd));
            } else { /* NFM: wavein[j] = sqrtf(re^2 + im^2) of the current hop's raw bin (src/rtl_airband.cpp:484-487), recomputed when used */
                const float4* cp = reinterpret_cast<const float4*>(iqin + ab_tile_off(rc));
                q.c01[g] = cp[0];
                q.c23[g] = cp[1];
            }
            if (raw_iq) {
                const float4* qp = reinterpret_cast<const float4*>(iqin + ab_tile_off(rd));
                q.q01[g] = qp[0];
                q.q23[g] = qp[1];
            }
        }
        (void)tail0; /* (the NFM + lowpass kind used to fetch its four delay-line entries here; it recomputes them now: SqShadow) */
    };
#if defined(AB_MASKED_DELAY)
    /* the current group's delayed values (the per-sample code takes them from here, so that a re-fetch reaches the group's later samples too) */
    float g_md[4] = {0.0f, 0.0f, 0.0f, 0.0f}, g_qr[4] = {0.0f, 0.0f, 0.0f, 0.0f}, g_qi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    lmask g_real = 0;
    int g_jq = 0;
    auto refetch = [&](const lmask miss) { /* wave-uniform call; the lanes of `miss` load their real delayed values of the current group */
        if (ab_lane(miss)) {
            const int rd = ring_row(a.row0 + g_jq, R);
            if (!nfm) {
                const float4 v = *reinterpret_cast<const float4*>(mag + ab_tile_off(rd));
                g_md[0] = v.x; g_md[1] = v.y; g_md[2] = v.z; g_md[3] = v.w;
            } else {
                const float4* qp = reinterpret_cast<const float4*>(iqin + ab_tile_off(rd));
                const float4 u = qp[0], v = qp[1];
                g_qr[0] = u.x; g_qi[0] = u.y; g_qr[1] = u.z; g_qi[1] = u.w;
                g_qr[2] = v.x; g_qi[2] = v.y; g_qr[3] = v.z; g_qi[3] = v.w;
            }
        }
        g_real |= miss;
    };
#endif
    /* "these registers are needed now": the compiler puts its wait for the group's loads here, BEFORE the next group's loads are issued */
    auto touch = [&](const Group& q) {
#pragma unroll
        for (int g = 0; g < GQ; g++) {
            if (!nfm) AB_NEEDED_NOW(AB_V(q.mc[g].x), AB_V(q.mc[g].y), AB_V(q.mc[g].z), AB_V(q.mc[g].w), AB_V(q.md[g].x), AB_V(q.md[g].y), AB_V(q.md[g].z), AB_V(q.md[g].w));
            else AB_NEEDED_NOW(AB_V(q.c01[g].x), AB_V(q.c01[g].y), AB_V(q.c01[g].z), AB_V(q.c01[g].w), AB_V(q.c23[g].x), AB_V(q.c23[g].y), AB_V(q.c23[g].z), AB_V(q.c23[g].w));
            if (raw_iq) AB_NEEDED_NOW(AB_V(q.q01[g].x), AB_V(q.q01[g].y), AB_V(q.q01[g].z), AB_V(q.q01[g].w), AB_V(q.q23[g].x), AB_V(q.q23[g].y), AB_V(q.q23[g].z), AB_V(q.q23[g].w));
        }
    };
    auto tail_in = [&](int n) { /* the delay-line tail advances once per sample (sq_advance) */
        int t = s.tail + n;
        return t >= AB_SQ_BUF ? t - AB_SQ_BUF : t;
    };

    /* ---- the sequential per-sample step ------------------------------------------------------------------------------------
     * Two versions of everything behind process_raw_sample(): the general one, and the one a QUIET wavefront takes (squelch_fsm.h:
     * every lane CLOSED or OPEN, nothing pending) -- there no sample is a first or last open one, audio is wanted by exactly the
     * OPEN lanes, and the code is one straight run with a few seldom-taken exits. */
    /* the AM kind and the CTCSS front get the second version; it doubles the per-sample code, and the register-hungry kinds lose more to spills than they gain */
    constexpr bool SPLIT_REST = KIND == AB_KIND_AM || KIND == AB_KIND_NFM_CTCSS;
#if defined(AB_MASKED_DELAY)
    auto rest = [&](auto quiet_tag, const int j, float cur_mag, float delayed_mag, float re, float im, const lmask went_closed AB_MD_PARAM) {
#else
    auto rest = [&](auto quiet_tag, const int j, float cur_mag, const float delayed_mag, float re, float im, const lmask went_closed) {
#endif
        constexpr bool Q = decltype(quiet_tag)::value;
        /* (in a specialised NFM kind every lane works on raw I/Q: m_raw_iq is the set of live lanes, which the compiler cannot know to be non-empty) */
        constexpr bool ALL_RAW_IQ = (KindBits<KIND>::value & AB_F_RAW_IQ) != 0;
        if (ALL_RAW_IQ || ab_any(m_raw_iq)) { /* src/rtl_airband.cpp:510-530 */
            const lmask filt = ALL_RAW_IQ ? sq_should_filter(s) : sq_should_filter(s) & m_raw_iq;
#if defined(AB_MASKED_DELAY)
            if (MASKD && ALL_RAW_IQ) { /* a lane that filters from this very sample on (pre-filter signal on a CLOSED lane) and fetched the stand-in */
                const lmask miss = filt & ~g_real;
                if (AB_UNLIKELY(ab_any(miss))) {
                    refetch(miss);
                    re = g_qr[r];
                    im = g_qi[r];
                }
            }
#endif
            if (ab_lane(filt)) { /* per-lane float work only: lane masks are not touched inside divergent code */
                const unsigned idx = dm_phi >> 16; /* sincosf_lut (src/util.cpp:113-127) */
                const float fract = (float)(dm_phi & 0xffffu) / 65536.0f;
                const float2 e0 = lut[idx], e1 = lut[idx + 1];
                const float s0 = e0.x, s1 = e1.x, c0 = e0.y, c1 = e1.y;
                const float swf = s0 + (s1 - s0) * fract;
                const float cwf = c0 + (c1 - c0) * fract;
                const float nswf = -swf;
                float tr = re * cwf - im * nswf; /* multiply(real, imag, cwf, -swf) */
                float ti = im * cwf + re * nswf;
                dm_phi = (dm_phi + cc.dm_dphi) & 0xffffffu;
                if (lowpass) { /* LowpassFilter::apply (src/filters.cpp:146-163) */
                    const float lxr0 = lxr1, lxi0 = lxi1;
                    lxr1 = lxr2; lxi1 = lxi2;
                    ab_div_const2(tr, ti, cc.lp_gain, cc.lp_rgain, div_lo, lxr2, lxi2); /* tr / gain, ti / gain */
                    const float lyr0 = lyr1, lyi0 = lyi1;
                    lyr1 = lyr2; lyi1 = lyi2;
                    lyr2 = (lxr0 + lxr2) + (2.0f * lxr1) + (cc.lp_yc0 * lyr0) + (cc.lp_yc1 * lyr1);
                    lyi2 = (lxi0 + lxi2) + (2.0f * lxi1) + (cc.lp_yc0 * lyi0) + (cc.lp_yc1 * lyi1);
                    tr = lyr2;
                    ti = lyi2;
                }
                re = tr;
                im = ti;
                cur_mag = ab_sqrt_rn(re * re + im * im); /* double sqrt rounded to float == correctly rounded sqrtf (exact_math.h) */
                /* the reference overwrites wavein[j] here (src/rtl_airband.cpp:524); only AM reads it back later */
                if (!nfm) mag[ab_tile_off(ring_row(a.row0 + AB_AGC_EXTRA + j, R))] = cur_mag;
            }
            sq_filtered(s, L, filt, cur_mag); /* process_filtered_sample for the lanes with a lowpass filter */
        }

        lmask fade_m = 0;
        if (!Q && ab_any(m_am)) { /* src/rtl_airband.cpp:532-547; first / last open samples only exist while a transition is pending */
            if (AB_UNLIKELY(ab_lane(sq_first_open(s) & m_am))) {
                const float lvl = sq_level(s);
#pragma nounroll
                for (int k = j; k < j + AB_AGC_EXTRA; k++) { /* the AGC_EXTRA magnitudes before the current one */
                    const float w = mag[ab_tile_off(ring_row(a.row0 + k, R))];
                    if (w >= lvl) agc = agc * 0.9f + w * 0.1f;
                }
            }
            fade_m = sq_last_open(s) & m_am;
        }
        /* (the compiler does not fold the lane test of a constant-zero mask: without the kind spelled out the NFM kinds keep the
         * fade-out loop of emit_sample and test for it on every sample) */
        constexpr bool KIND_IS_NFM = (KindBits<KIND>::value & AB_F_NFM) != 0;
        const bool fade = (Q || KIND_IS_NFM) ? false : ab_lane(fade_m);

        float out = 0.0f;
#if defined(AB_MASKED_DELAY)
        if (MASKD && !ALL_RAW_IQ) { /* AM kind: OPENING precedes OPEN by the opening delay, so this does not happen; kept as the general rule */
            const lmask miss = (Q ? s.cO : sq_should_audio(s)) & ~g_real;
            if (AB_UNLIKELY(ab_any(miss))) {
                refetch(miss);
                delayed_mag = g_md[r];
            }
        }
#endif
        if (WAVE_HAS_CTCSS) audio_seen |= (Q ? s.cO : sq_should_audio(s)) | went_closed;
        const bool audio = ab_lane(Q ? s.cO : sq_should_audio(s));
        if (audio) {
            if (!nfm) { /* AM: src/rtl_airband.cpp:553-563 */
                if (cur_mag > sq_level(s)) agc = agc * 0.995f + cur_mag * 0.005f;
                out = (delayed_mag - agc) / (agc * 1.5f);
                if (fabsf(out) > 0.8f) {
                    out *= 0.85f;
                    agc *= 1.15f;
                }
            } else { /* NFM: src/rtl_airband.cpp:565-582 */
                if (!(cc.flags & AB_F_QUADRI)) {
                    const float nbj = -pj;
                    const float cr = re * pr - im * nbj;
                    const float cj = im * pr + re * nbj;
                    out = (float)((double)fast_atan2_dev(cj, cr) * 0.31830988618379067154);
                } else {
                    out = (float)((double)((pr * im - re * pj) / (re * re + im * im + 1.0f)) * 0.31830988618379067154);
                }
                pr = re;
                pj = im;
                agc = agc * 0.995f + out * 0.005f;
                out -= agc;
                out = out * one_minus_alpha + prev_out * cc.alpha;
                prev_out = out;
            }
        }
        const int state = !a.trace ? 0 : Q ? (ab_lane(s.cO) ? AB_ST_OPEN : AB_ST_CLOSED) : sq_cur(s);
        if (PACKED) {
            /* HAND_WORD: the audio sample's own bits while the squelch wants audio (a NaN, which no finite input produces, as THE
             * canonical quiet NaN: every NaN ends the same way downstream -- tone powers fail their comparisons, the notch state
             * stays NaN, the output is forced to 0, src/rtl_airband.cpp:597); otherwise a negative quiet NaN no audio sample can
             * equal, whose lowest bit says "the squelch went CLOSED on this sample" (CTCSS::reset).  Half the bytes of a pair, and the
             * squelch state the trace records travels in the trace buffer itself (debug handles only). */
            const unsigned w = audio ? ((out != out) ? HAND_NAN : __float_as_uint(out)) : (HAND_IDLE | (ab_lane(went_closed) ? 1u : 0u));
            handw[(j & (HAND_RUN - 1)) * OSTRIDE] = w;
            if (trace) trace[(long)j * S] = (uint8_t)((state & 7) | (audio ? 16 : 0));
        } else if (WAVE_HAS_CTCSS) {
            /* front half of a CTCSS-capable kind: hand (pre-notch audio, flags) to the tone and back kernels.  Raw I/Q of
             * an open sample is written now; the back kernel zeroes it again if the tone gate turns out closed. */
            const unsigned f = (audio ? FL_AUDIO : 0u) | (fade ? FL_FADE : 0u) | (ab_lane(went_closed) ? FL_RESET : 0u) | ((unsigned)state << FL_STATE_SHIFT);
            /* parked in LDS; 16 samples leave together as whole 128-byte lines of the channel-major hand-off rows */
            hand[((j & (HAND_RUN - 1)) * OSTRIDE)] = make_float2(out, __uint_as_float(f));
            if ((cc.flags & AB_F_IQ_OUT) && audio) iqout[(long)j * S] = make_float2(re, im);
        } else {
            emit_sample(a, cc, o, wrow, rz, iqout, trace, j, audio, fade, true, state, out, re, im, true);
        }
    };
    auto sample = [&](const int j, const float cur_mag, const float delayed_mag /* lowpass kind: the prefetched delay-line entry */, const float re, const float im, const bool first_of_group AB_MD_PARAM) {
        lmask went_closed = 0;
        if (AB_LIKELY(s.quiet)) sq_raw_quiet(s, L, cur_mag, delayed_mag, first_of_group || !aligned4);
        else went_closed = sq_raw_full(s, L, cur_mag, delayed_mag);
        /* a request raised by this very sample ends the quiet spell at once: its last-open handling is in the general version */
        if (SPLIT_REST && AB_LIKELY(s.quiet)) rest(std::true_type{}, j, cur_mag, delayed_mag, re, im, went_closed AB_MD_ARG(r)); /* the sample that settles the last lane still reports who just closed */
        else rest(std::false_type{}, j, cur_mag, delayed_mag, re, im, went_closed AB_MD_ARG(r));
    };
    /* ---- four samples of a STABLE wavefront as ONE basic block (AM kind, NFM + CTCSS front) ------------------------------------------
     * sq_raw_stable4() (squelch_fsm.h) runs the four squelch steps on a copy of the state and commits it only if no lane asked for a
     * transition: then no lane changed state, audio is wanted by the same lanes (OPEN and CLOSING) for all four samples, and what is left
     * is those lanes' per-sample float chain -- one exec-masked region for the four samples instead of three per sample, no scalar
     * mask algebra between them.  Round 3 measured that nothing but the NUMBER of instructions moves stage 2 (profiles/r03_experiments.md):
     * the per-sample version issues ~60 vector + ~60 scalar + ~15 branch instructions per AM sample, most of the scalar ones and all
     * of the branches for events that do not happen in a quiet group. */
    /* (the plain NFM kind was given the same block and lost it again: with the fused output path behind the discriminator it needs 143 spilled registers
     * at three waves per SIMD) */
    constexpr bool SPEC4 = KIND == AB_KIND_AM || KIND == AB_KIND_NFM_CTCSS;
    const bool wave_has_notch = ab_any(ab_ballot((cc.flags & AB_F_NOTCH) != 0));
    const bool wave_has_iq_out = ab_any(ab_ballot((cc.flags & AB_F_IQ_OUT) != 0));
    auto stable_tail4 = [&](const int jq, const float* mcs, const float* mds, const float* qr, const float* qi) {
        if (WAVE_HAS_CTCSS) audio_seen |= sq_should_audio(s);
        const bool open = ab_lane(sq_should_audio(s)); /* Squelch::should_process_audio() == is_open() (no tone gate in these kinds), the same lanes for the four samples */
        /* Squelch::should_filter_sample() for the four samples: a CLOSED lane with signal would have ended the stable spell, so it is every lane that is not CLOSED or aborting */
        const bool filt = ab_lane(~s.cC & ~s.cA & s.active);
        const int st_lane = trace ? sq_cur(s) : 0;
        if (KIND == AB_KIND_AM) {
            float out4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (open) {
                const float lvl = sq_level(s);
#pragma unroll
                for (int r = 0; r < 4; r++) { /* src/rtl_airband.cpp:553-563, then :589-603 */
                    if (mcs[r] > lvl) agc = agc * 0.995f + mcs[r] * 0.005f;
                    float out = (mds[r] - agc) / (agc * 1.5f);
                    if (fabsf(out) > 0.8f) {
                        out *= 0.85f;
                        agc *= 1.15f;
                    }
                    if (wave_has_notch) { /* wave-uniform; the lanes without a notch filter keep their value */
                        if (cc.flags & AB_F_NOTCH) {
                            o.nx0 = o.nx1; o.nx1 = o.nx2; o.nx2 = out;
                            o.ny0 = o.ny1; o.ny1 = o.ny2;
                            o.ny2 = cc.notch_d0 * o.nx2 - cc.notch_d1 * o.nx1 + cc.notch_d0 * o.nx0 + cc.notch_d1 * o.ny1 - cc.notch_d2 * o.ny0;
                            out = o.ny2;
                        }
                    }
                    out *= cc.ampfactor;
                    out4[r] = (out != out) ? 0.0f : __builtin_amdgcn_fmed3f(out, -1.0f, 1.0f);
                }
                o.axc = '*';
                rz.run_open = true;
            }
#pragma unroll
            for (int r = 0; r < 4; r++) wrow.staged[(jq + r - wrow.j0) * wrow.stride] = out4[r];
            if (AB_UNLIKELY(wave_has_iq_out)) { /* an AM-kind channel has no raw I/Q: its iq_out rows are zeros, open or not (emit_sample) */
                if (cc.flags & AB_F_IQ_OUT)
#pragma unroll
                    for (int r = 0; r < 4; r++) iqout[(long)(jq + r) * S] = make_float2(0.0f, 0.0f);
            }
            if (AB_UNLIKELY(trace != nullptr)) {
                const uint8_t tb = (uint8_t)((st_lane & 7) | (open ? (8 | 16) : 0));
#pragma unroll
                for (int r = 0; r < 4; r++) trace[(long)(jq + r) * S] = tb;
            }
        } else { /* NFM + CTCSS front: derotation, discriminator, de-emphasis -> one hand-off word per sample (see rest()) */
            unsigned w4[4] = {HAND_IDLE, HAND_IDLE, HAND_IDLE, HAND_IDLE};
            if (filt) { /* OPENING lanes derotate (their phase accumulator runs) without producing audio; OPEN and CLOSING lanes do both */
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float re0 = qr[r], im0 = qi[r];
                    const unsigned idx = dm_phi >> 16; /* sincosf_lut (src/util.cpp:113-127) */
                    const float fract = (float)(dm_phi & 0xffffu) / 65536.0f;
                    const float2 e0 = lut[idx], e1 = lut[idx + 1];
                    const float s0 = e0.x, s1 = e1.x, c0 = e0.y, c1 = e1.y;
                    const float swf = s0 + (s1 - s0) * fract;
                    const float cwf = c0 + (c1 - c0) * fract;
                    const float nswf = -swf;
                    const float re = re0 * cwf - im0 * nswf; /* multiply(real, imag, cwf, -swf) */
                    const float im = im0 * cwf + re0 * nswf;
                    dm_phi = (dm_phi + cc.dm_dphi) & 0xffffffu;
                    if (open) {
                        float out;
                        if (!(cc.flags & AB_F_QUADRI)) {
                            const float nbj = -pj;
                            const float cr = re * pr - im * nbj;
                            const float cj = im * pr + re * nbj;
                            out = (float)((double)fast_atan2_dev(cj, cr) * 0.31830988618379067154);
                        } else {
                            out = (float)((double)((pr * im - re * pj) / (re * re + im * im + 1.0f)) * 0.31830988618379067154);
                        }
                        pr = re;
                        pj = im;
                        agc = agc * 0.995f + out * 0.005f;
                        out -= agc;
                        out = out * one_minus_alpha + prev_out * cc.alpha;
                        prev_out = out;
                        w4[r] = (out != out) ? HAND_NAN : __float_as_uint(out);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) handw[((jq + r) & (HAND_RUN - 1)) * OSTRIDE] = w4[r];
            if (AB_UNLIKELY(trace != nullptr)) {
                const uint8_t tb = (uint8_t)((st_lane & 7) | (open ? 16 : 0));
#pragma unroll
                for (int r = 0; r < 4; r++) trace[(long)(jq + r) * S] = tb;
            }
        }
    };
    /* ---- four samples of a STABLE wavefront of the NFM + lowpass kind as one block (round 6; the review's item 1 c) ----------------------------------------------
     * The same bargain as sq_raw_stable4() / stable_tail4() with the post-filter path inside it.  While no lane is entering a state, no timer runs out and no CLOSED lane is
     * due to forget its recent opens (sq_stable4), four process_raw_sample() / filter / process_filtered_sample() rounds change no lane mask -- unless a lane asks for a
     * transition: OPEN without signal (the pre-filter average OR the post-filter gate), CLOSED with it, a low-signal abort, the post-filter average falling under the delay
     * line's entry (src/squelch.cpp:248-276).  So the rounds run on COPIES -- the squelch's averages and counters, the derotation phase, the biquad's delay lines -- with the
     * requests only collected; if none came the copies are the state after four reference rounds, bit for bit, and are committed, and the audio chain of the OPEN / CLOSING lanes
     * runs for the four filtered samples behind it.  Otherwise nothing is committed and the four samples go through sample() as before.  What it saves is the per-sample scalar
     * work (request algebra, the any() branches around every region, emit_sample's flag tests): an open wavefront's own instruction stream is what the stage's time follows
     * (profiles/r06_experiments.md M), and this kind's is the longest.
     * MEASURED AND NOT ADOPTED (profiles/r06_experiments.md O): bit-exact (emulated wavefront, 113 GPU cases), 13 % fewer branches and 8 % fewer scalar instructions -- and the
     * kind needs 168 registers instead of 123 (three wavefronts per SIMD instead of four) with 30 of them and 149 scalar ones spilled: alone 2.09 -> 2.81 ms, the stage 5.63 -> 6.65.
     * The copies a roll-back needs do not fit beside the per-sample path they fall back to.  Experiment builds: -DAB_LP_STABLE4. */
    constexpr bool SPEC4_LP = KIND == AB_KIND_NFM_LOWPASS
#if !defined(AB_LP_STABLE4)
                              && false
#endif
        ;
    auto lp_stable4 = [&](const int jq, const float* mcs, const float* mds, const float* qr, const float* qi) -> bool {
        SqRegs t = s;
        unsigned phi = dm_phi;
        float xr1 = lxr1, xr2 = lxr2, xi1 = lxi1, xi2 = lxi2, yr1 = lyr1, yr2 = lyr2, yi1 = lyi1, yi2 = lyi2;
        float fre[4], fim[4];
        const lmask timed = t.cOg | t.cCg | t.cA;
        const lmask counting = t.cOg | t.cCg | t.cO; /* the lanes whose low-signal run length is kept (:233-245) */
        const lmask care = t.cO | t.cC, want = t.cO;  /* OPEN lanes ask for CLOSING without the signal, CLOSED lanes for OPENING with it */
        lmask bad = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            /* update_current_state() where nothing is entered and nothing expires (sq_advance): the closed-sample count, the timers, the delay line's cursors */
            const lmask below = ab_ballot(t.closed_count < 1000u);
            t.closed_count += ab_lane(t.cC & below) ? 1u : 0u;
            t.delay += ab_lane(timed) ? 1 : 0;
            t.tail = t.tail + 1 == AB_SQ_BUF ? 0 : t.tail + 1;
            t.head = t.head + 1 == AB_SQ_BUF ? 0 : t.head + 1;
            t.dly = mds[r];
            t.sample_count++;
            if (r == 0 && AB_UNLIKELY((t.sample_count & 15u) == 0u)) sq_noise_floor(t, L); /* (aligned4: only the first of the four can be a 16th sample) */
            sq_avg(t.cap, t.pre_full, t.pre_capped, mcs[r]);
            const lmask pre = ab_ballot(t.pre_capped >= t.lvl);
            lmask sig = pre; /* has_signal(), src/squelch.cpp:462-475 */
            if (ab_any(t.using_post)) sig &= ~t.using_post | ab_ballot(t.post_capped >= t.dly);
            const lmask low = counting & ab_ballot(!(mcs[r] >= t.lvl));
            const int run_len = t.low_count + 1;
            const int idle_len = ab_lane(counting) ? 0 : t.low_count;
            t.low_count = ab_lane(low) ? run_len : idle_len;
            bad |= ((sig ^ want) & care) | (low & ab_ballot(t.low_count >= 88)); /* low_signal_abort_ */
            /* should_filter_sample() (:136-138), then the lanes it names: derotation, Bessel lowpass, the filtered magnitude (src/rtl_airband.cpp:510-530) */
            const lmask filt = (pre | ~t.cC) & ~t.cA & t.active;
            float re = qr[r], im = qi[r], fmag = 0.0f;
            if (ab_lane(filt)) {
                const unsigned idx = phi >> 16; /* sincosf_lut (src/util.cpp:113-127) */
                const float fract = (float)(phi & 0xffffu) / 65536.0f;
                const float2 e0 = lut[idx], e1 = lut[idx + 1];
                const float s0 = e0.x, s1 = e1.x, c0 = e0.y, c1 = e1.y;
                const float swf = s0 + (s1 - s0) * fract;
                const float cwf = c0 + (c1 - c0) * fract;
                const float nswf = -swf;
                const float tr = re * cwf - im * nswf; /* multiply(real, imag, cwf, -swf) */
                const float ti = im * cwf + re * nswf;
                phi = (phi + cc.dm_dphi) & 0xffffffu;
                /* LowpassFilter::apply (src/filters.cpp:146-163) */
                const float xr0 = xr1, xi0 = xi1;
                xr1 = xr2; xi1 = xi2;
                ab_div_const2(tr, ti, cc.lp_gain, cc.lp_rgain, div_lo, xr2, xi2);
                const float yr0 = yr1, yi0 = yi1;
                yr1 = yr2; yi1 = yi2;
                yr2 = (xr0 + xr2) + (2.0f * xr1) + (cc.lp_yc0 * yr0) + (cc.lp_yc1 * yr1);
                yi2 = (xi0 + xi2) + (2.0f * xi1) + (cc.lp_yc0 * yi0) + (cc.lp_yc1 * yi1);
                re = yr2;
                im = yi2;
                fmag = ab_sqrt_rn(re * re + im * im);
            }
            fre[r] = re;
            fim[r] = im;
            /* process_filtered_sample() (:248-276) for those lanes; a lane whose post-filter average falls under the delay line's entry asks for CLOSED: the spell ends */
            if (ab_any(filt)) {
                const float delayed = t.dly;
                const lmask run = filt & ~(t.cOg & ab_ballot(t.delay < AB_SQ_BUF));
                const lmask seed = run & t.cOg & ab_ballot(t.delay == AB_SQ_BUF);
                float full = ab_lane(seed) ? delayed : t.post_full, capped = ab_lane(seed) ? delayed : t.post_capped;
                t.using_post |= run;
                sq_avg(t.cap, full, capped, fmag);
                t.post_full = ab_lane(run) ? full : t.post_full;
                t.post_capped = ab_lane(run) ? capped : t.post_capped;
                bad |= run & ab_ballot(capped < delayed);
            }
            AB_SCHED_FENCE(); /* one sample after the other: interleaved, the four rounds' temporaries are all live at once (278 spilled registers) */
        }
        bad |= t.cC & ~ab_ballot(t.closed_count < 1000u) & t.recent_nz; /* sq_saturated(): the count is monotonic, so the end of the group tells */
        if (AB_UNLIKELY(ab_any(bad & t.active))) return false;
        s = t;
        dm_phi = phi;
        lxr1 = xr1; lxr2 = xr2; lxi1 = xi1; lxi2 = xi2; lyr1 = yr1; lyr2 = yr2; lyi1 = yi1; lyi2 = yi2;
        /* the audio chain of the lanes whose squelch lets samples through -- the same lanes for the four samples (src/rtl_airband.cpp:565-620) */
        const bool audio = ab_lane(sq_should_audio(s));
        const int state = !a.trace ? 0 : sq_cur(s);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float out = 0.0f;
            const float re = fre[r], im = fim[r];
            if (audio) {
                if (!(cc.flags & AB_F_QUADRI)) {
                    const float nbj = -pj;
                    const float cr = re * pr - im * nbj;
                    const float cj = im * pr + re * nbj;
                    out = (float)((double)fast_atan2_dev(cj, cr) * 0.31830988618379067154);
                } else {
                    out = (float)((double)((pr * im - re * pj) / (re * re + im * im + 1.0f)) * 0.31830988618379067154);
                }
                pr = re;
                pj = im;
                agc = agc * 0.995f + out * 0.005f;
                out -= agc;
                out = out * one_minus_alpha + prev_out * cc.alpha;
                prev_out = out;
            }
            emit_sample(a, cc, o, wrow, rz, iqout, trace, jq + r, audio, false, true, state, out, re, im, true);
            AB_SCHED_FENCE();
        }
        return true;
    };
    auto group = [&](const Group& q, int j0) {
#pragma unroll
        for (int g = 0; g < GQ; g++) {
            float mcs[4], mds[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qr[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (!nfm) {
                mcs[0] = q.mc[g].x; mcs[1] = q.mc[g].y; mcs[2] = q.mc[g].z; mcs[3] = q.mc[g].w;
                mds[0] = q.md[g].x; mds[1] = q.md[g].y; mds[2] = q.md[g].z; mds[3] = q.md[g].w;
            } else {
                const float pw[4] = {q.c01[g].x * q.c01[g].x + q.c01[g].y * q.c01[g].y, q.c01[g].z * q.c01[g].z + q.c01[g].w * q.c01[g].w,
                                     q.c23[g].x * q.c23[g].x + q.c23[g].y * q.c23[g].y, q.c23[g].z * q.c23[g].z + q.c23[g].w * q.c23[g].w};
                ab_sqrt_rn4(pw, mcs); /* sqrtf of each, correctly rounded (exact_math.h) */
            }
            if (raw_iq) {
                qr[0] = q.q01[g].x; qi[0] = q.q01[g].y; qr[1] = q.q01[g].z; qi[1] = q.q01[g].w;
                qr[2] = q.q23[g].x; qi[2] = q.q23[g].y; qr[3] = q.q23[g].z; qi[3] = q.q23[g].w;
            }
            const int jq = j0 + 4 * g;
            if (KIND == AB_KIND_NFM_LOWPASS) {
                /* the delay-line entry each sample sees -- what sample n - 101 pushed -- from the shadow average (squelch_fsm.h, SqShadow), which then
                 * takes in this iteration's AGC_EXTRA-delayed sample, the squelch's input of 100 samples ago: the magnitude of the very raw bin the
                 * squelch computed it from.  (The first 101 samples of a stream see the zeros of a fresh buffer; the shadow starts with squelch sample 0.) */
                const unsigned c0 = s.sample_count; /* before the group's first increment */
                const float dpw[4] = {qr[0] * qr[0] + qi[0] * qi[0], qr[1] * qr[1] + qi[1] * qi[1], qr[2] * qr[2] + qi[2] * qi[2], qr[3] * qr[3] + qi[3] * qi[3]};
                float dmag[4];
                ab_sqrt_rn4(dpw, dmag);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int j = jq + r;
                    mds[r] = (first_batch && j < 101) ? 0.0f : sq_shadow_value(sh);
                    if (!(first_batch && j < AB_AGC_EXTRA)) sq_shadow_step(sh, L, dmag[r], c0 + (unsigned)r + 2u);
                }
            }
            if ((jq % RUN) == 0) wrow.j0 = jq;
#if defined(AB_MASKED_DELAY)
            if (MASKD) {
#pragma unroll
                for (int r = 0; r < 4; r++) { g_md[r] = mds[r]; g_qr[r] = qr[r]; g_qi[r] = qi[r]; }
                g_real = q.real;
                g_jq = jq;
                /* lanes that already let samples through (or are about to) but were CLOSED when this group's fetch was issued, one or two groups ago */
                const lmask want0 = nfm ? ((~s.cC | ~s.nC) & ~s.cA & s.active) : (s.cO | s.cCg | s.nO | s.nCg);
                const lmask miss0 = want0 & ~g_real;
                if (AB_UNLIKELY(ab_any(miss0))) refetch(miss0);
                if (SPEC4 && AB_LIKELY(aligned4 && sq_stable4(s))) { /* wave-uniform */
                    if (AB_LIKELY(sq_raw_stable4(s, L, mcs))) {
                        stable_tail4(jq, mcs, g_md, g_qr, g_qi);
                        continue;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) sample(jq + r, mcs[r], g_md[r], g_qr[r], g_qi[r], r == 0, r);
                continue;
            }
#endif
            if (SPEC4 && AB_LIKELY(aligned4 && sq_stable4(s))) { /* wave-uniform */
                if (AB_LIKELY(sq_raw_stable4(s, L, mcs))) {
                    stable_tail4(jq, mcs, mds, qr, qi);
                    continue;
                }
            }
            if (SPEC4_LP && AB_LIKELY(aligned4 && sq_stable4(s))) { /* wave-uniform */
                if (AB_LIKELY(lp_stable4(jq, mcs, mds, qr, qi))) continue;
            }
#pragma unroll
            for (int r = 0; r < 4; r++) sample(jq + r, mcs[r], mds[r], qr[r], qi[r], r == 0 AB_MD_ARG(r));
        }
    };
    /* finished output runs leave AFTER the next group's loads have been waited for: the compiler's wait is `s_waitcnt vmcnt(0)`, which
     * also waits for every store still in flight -- issued the other way round, each run's 128-byte-line stores were waited out in
     * full (write acknowledgements take microseconds) before the next group could start.  GS is 4 and the runs are 32 (audio)
     * or 16 (hand-off) samples long, so a run can only end with a group. */
    auto flush = [&](int j0) {
        if (!WAVE_HAS_CTCSS && ((j0 + GS) % RUN) == 0) wave_flush(wrow, rz);
        if (WAVE_HAS_CTCSS && ((j0 + GS) % HAND_RUN) == 0) hand_flush(HAND_RUN, j0 + GS - HAND_RUN);
    };

    Group qa, qb;
    fetch(qa, 0, s.tail);
    touch(qa);
    /* WAVE_BATCH is 1000 or 2000 (params.cpp): a whole number of group PAIRS.  Every fetch and every touch below is unconditional --
     * behind an `if` the compiler can no longer pair a wait with its loads and falls back to waiting for everything in flight at the
     * first use of a group, which is right after the NEXT group's loads were issued.  The pair after the last one re-reads the
     * batch's last group instead (never used). */
    for (int j0 = 0; j0 < B; j0 += 2 * GS) {
        /* regrouped handles: the workgroup's wavefronts walk the batch in step (every wavefront runs the same number of iterations; one whose lanes have all left is
         * not waited for), so that a ring line two of them read is in L2 for the second -- a wavefront of closed channels runs ~3x faster than one of open ones.
         * A waiting wavefront costs no issue slot, which is what stage 2 is short of. */
#if !defined(AB_REGROUP_FREE) /* experiment builds: the regrouped workgroup's wavefronts NOT in step (closed ones run ahead, finish and free their registers; shared ring lines are fetched once per reader) */
        if (W > 1 && a.regroup != 2) __syncthreads(); /* (line-group regrouping: nothing is shared, nobody waits) */
#endif
        fetch(qb, j0 + GS, tail_in(GS)); /* flies under this group's samples */
        group(qa, j0);
        touch(qb);
        flush(j0);
        fetch(qa, j0 + 2 * GS < B ? j0 + 2 * GS : B - GS, tail_in(GS));
        group(qb, j0 + GS);
        touch(qa);
        flush(j0 + GS);
    }

    if (!WAVE_HAS_CTCSS && (B % RUN) != 0) wave_flush(wrow, rz, B % RUN); /* WAVE_BATCH = 1000: the last run is a short one */
    if (WAVE_HAS_CTCSS && (B % HAND_RUN) != 0) hand_flush(B % HAND_RUN, B - B % HAND_RUN);
    if (!WAVE_HAS_CTCSS) { /* the back kernel owns these in the split kinds */
        if (o.axc != ' ') sp->active_counter++;
        sp->axc_prev = sp->axc;
        sp->axc = o.axc;
        a.out_axc[ext] = (uint8_t)o.axc;
        sp->nx[0] = o.nx0; sp->nx[1] = o.nx1; sp->nx[2] = o.nx2; sp->ny[0] = o.ny0; sp->ny[1] = o.ny1; sp->ny[2] = o.ny2;
        sp->row_zero = rz.batch_open ? 0 : ((rz.held & 1) | 2); /* an open batch may also have faded into the carry */
    }
    sp->agcavgfast = agc; sp->pr = pr; sp->pj = pj; sp->prev_waveout = prev_out; sp->dm_phi = dm_phi;
    if (KIND == AB_KIND_NFM_LOWPASS) { sp->sh_nf = sh.nf; sp->sh_cap = sh.cap; sp->sh_capped = sh.capped; sp->sh_dly = s.dly; }
    if (KIND == AB_KIND_GENERIC) sp->sh_dly = sq_delayed(s, L); /* buffer_[buffer_tail_] for the stats mirror (signal_outside_filter); this kind keeps the delay line in memory */
    sq_store(s, L, sp, B);
    if (WAVE_HAS_CTCSS) a.sq_key[slot] = ab_lane(audio_seen) ? 1 : 0; /* the tone kernel skips the channels without (round 6); regrouped handles: the back kernel deals its slots out by it */
    sp->lxr[1] = lxr1; sp->lxr[2] = lxr2; sp->lxi[1] = lxi1; sp->lxi[2] = lxi2;
    sp->lyr[1] = lyr1; sp->lyr[2] = lyr2; sp->lyi[1] = lyi1; sp->lyi[2] = lyi2;
}

constexpr int TONE_GROUP = 50; /* samples per tone-kernel step; divides WAVE_BATCH = 1000 and 2000, fits one wavefront's lanes */

}  // namespace

/* One kernel per demod kind: register allocation (hence occupancy) is then set by that kind's code path alone --
 * the AM kernel does not pay for the lowpass registers.  Slot blocks of one kind are contiguous.
 * CTCSS-capable kinds are split in three: this "front" (squelch + discriminator -> audio, flags), the tone kernel
 * (Goertzel banks, one wavefront per channel) and the back kernel (gate, notch, output). */
#ifndef AB_AM_WAVES_N
#define AB_AM_WAVES_N 4 /* experiment builds: 5 = the AM kind held to 96 registers (14 of them spilled) so that a fifth wavefront fits a SIMD -- profiles/r06_experiments.md N */
#endif
constexpr int AB_DEMOD_WAVES = 3, AB_AM_WAVES = AB_AM_WAVES_N, AB_FRONT_WAVES = 3; /* (the front at four waves: 128 VGPRs with 18 of them spilled once it carries the quiet-group path -- 6.30 ms of stage 2 against 6.20 at three) */

/* ---- regrouping (AIRBAND_HIP_FLAG_REGROUP): closed channels share wavefronts -----------------------------------------------------------------------
 * Slots are assigned by demod KIND when a handle is prepared, and a wavefront works on 64 consecutive slots: at any time about half of its lanes (on the
 * BASELINE signal; nine in ten on a real band) are channels whose squelch is closed, and they ride through the open lanes' instructions under an exec mask --
 * ~180 of the NFM + lowpass kind's ~240 vector instructions per sample, ~35 of the AM kind's ~60.  Stage 2 is bound by vector issue, so those are
 * paid in full; a wavefront whose lanes are ALL closed skips them.  Everything a lane touches is addressed by its slot, so which 64 slots a wavefront works on is
 * free.  Round 6 first re-sorted every kind's slots GLOBALLY at batch boundaries: 30 % fewer vector instructions as predicted -- and a slower stage, because the
 * slots that share a 128-byte line of the stage-1 rings (four AM / two NFM neighbours) then sat in different wavefronts that read the line milliseconds apart, once
 * each from memory (+40 ... +90 % ring fetches, profiles/r06_regroup_global/).  Hence this form: a WORKGROUP of AB_REGROUP_WAVES wavefronts owns that many x 64
 * consecutive slots, deals them out among its wavefronts (stable partition: channels not at rest in CLOSED, then the closed ones, then slots without a channel),
 * and the wavefronts walk the batch in step (demod_wave: a barrier every eight samples): whatever line two of them share, the second reader finds it in L2.  The
 * closed wavefronts wait at the barriers -- without using an issue slot.  Results are the same bit for bit: same operations on the same values, lane for lane. */
constexpr int AB_REGROUP_WAVES = 4;
constexpr int WAVE_LDS_FLOATS = RUN * OSTRIDE + 3 * 64; /* a wavefront's own LDS: the output-line staging area, ext_of / skip_of / slot_of */

/* key: 0 = channel with work, 1 = channel at rest, 2 = slot without a channel, 3 = wavefront beyond the kind's last block (no slot at all).  Returns the slot this
 * lane works on (-1: none).  slot_at [W * 64], wcnt [W][2]: LDS of the workgroup. */
template <int W>
__device__ __forceinline__ int wg_regroup(int key, int wave, int lane, int home, int nb, int* slot_at, int* wcnt) {
    const lmask m0 = __ballot(key == 0), m1 = __ballot(key == 1), m2 = __ballot(key == 2);
    if (lane == 0) {
        wcnt[2 * wave] = __builtin_popcountll(m0);
        wcnt[2 * wave + 1] = __builtin_popcountll(m1);
    }
    __syncthreads();
    int tot0 = 0, tot1 = 0, off0 = 0, off1 = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        const int c0 = wcnt[2 * w], c1 = wcnt[2 * w + 1];
        tot0 += c0;
        tot1 += c1;
        off0 += w < wave ? c0 : 0;
        off1 += w < wave ? c1 : 0;
    }
    const lmask below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int at = key == 0   ? off0 + __builtin_popcountll(m0 & below)
                   : key == 1 ? tot0 + off1 + __builtin_popcountll(m1 & below)
                   : key == 2 ? tot0 + tot1 + (wave * 64 - off0 - off1) + __builtin_popcountll(m2 & below) /* (the earlier wavefronts' slots without a channel) */
                              : -1;
    if (at >= 0) slot_at[at] = home;
    __syncthreads();
    const int p = wave * 64 + lane;
    return p < nb * 64 ? slot_at[p] : -1;
}

template <int KIND, bool WAVE_HAS_CTCSS, int W>
__device__ __forceinline__ void demod_block(const DemodArgs& a, int first_block, int n_blocks, float* lds_demod) {
    const int lane = threadIdx.x & 63;
    const int wave = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0; /* (the same number on every lane: scalar) */
    /* the (sin, cos) table of sincosf_lut (src/util.cpp:105-127), 257 float2 -- a sample's
     * derotation then costs one LDS read instead of four dependent trips to L2 on the serial path */
    float2* lut = reinterpret_cast<float2*>(lds_demod);
    if (KIND != AB_KIND_AM) {
        for (int i = threadIdx.x; i < 257; i += 64 * W) lut[i] = make_float2(a.sin_lut[i], a.cos_lut[i]);
        if (W == 1) __syncthreads(); /* one wavefront per block: orders the table writes before any lane's reads (regrouped: the barriers of wg_regroup do) */
    }
    float* areas = reinterpret_cast<float*>(lut + (KIND == AB_KIND_AM ? 0 : 258));
    float* ostage = areas + wave * WAVE_LDS_FLOATS;
    int* ext_of = reinterpret_cast<int*>(ostage + RUN * OSTRIDE);
    int* skip_of = ext_of + 64;
    int* slot_of = skip_of + 64;
    int slot; /* padding slots carry flags == 0 */
    if (W == 1) {
        slot = (first_block + blockIdx.x) * 64 + lane;
        if (a.perm) slot = a.perm[slot]; /* regroup == 3: the batch's permutation (regroup_perm_kernel) */
    } else {
        int* slot_at = reinterpret_cast<int*>(areas + W * WAVE_LDS_FLOATS);
        int* wcnt = slot_at + W * 64;
        const int block0 = first_block + blockIdx.x * W;
        const int left = first_block + n_blocks - block0, nb = left < W ? left : W; /* the kind's last workgroup may own fewer blocks */
        const int home = (block0 + wave) * 64 + lane;
        int key = 3;
        if (wave < nb) {
            const ChanState* hp = a.cs + home;
            const bool valid = (a.cc[home].flags & AB_F_VALID) != 0;
            bool busy = valid && (hp->cur != AB_ST_CLOSED || hp->next != AB_ST_CLOSED);
            if (a.regroup == 2) {
                /* LINE GROUPS move together (round 6, second form): the slots that share a 128-byte line of the stage-1 rings -- four of the |bin| ring (32 bytes per slot and
                 * 8-row tile), two of the raw-I/Q ring (64) -- are busy if any of them is.  Then no line is read by two wavefronts, the workgroup's wavefronts need not walk
                 * in step, and a wavefront of closed groups finishes in a third of the time and gives its registers back.  The price: a closed channel next to an open one
                 * stays in the open wavefronts (on the BASELINE signal 28 % of the NFM pairs and 8 % of the AM quads are closed outright; on a quiet band most are). */
                constexpr int G = (KIND == AB_KIND_AM || KIND == AB_KIND_GENERIC) ? 4 : 2;
                const lmask any_busy = __ballot(busy);
                const lmask gmask = ((1ull << G) - 1ull) << (lane & ~(G - 1));
                busy = (any_busy & gmask) != 0ull;
            }
            key = !valid ? 2 : busy ? 0 : 1;
        }
        slot = wg_regroup<W>(key, wave, lane, home, nb, slot_at, wcnt);
        if (slot < 0) return; /* whole wavefronts only (wave-uniform), behind the workgroup's set-up barriers */
    }
    const ChanConst cc = a.cc[slot];
    ext_of[lane] = a.slot_to_ext[slot];
    slot_of[lane] = slot;
    const bool full_block = __ballot((cc.flags & AB_F_VALID) != 0) == ~0ull; /* padding lanes leave early and cannot take part in a cooperative store */
    if (W == 1) __syncthreads();
    else AB_LOCKSTEP_FENCED(); /* the tables are the wavefront's own */
    demod_wave<KIND, WAVE_HAS_CTCSS, W>(a, cc, a.cs + slot, slot, lut, ostage, ext_of, skip_of, slot_of, full_block);
}

/* wavefronts per SIMD the register allocation is held to.  A lane-per-channel wavefront advances at about one instruction per
 * eight cycles whatever shares its SIMD (VALU -> SGPR -> SALU -> VALU hops of the lane-mask state machine), so throughput
 * rises with residency until the vector pipe saturates: the AM kind, light on registers, is built for four (2.12 -> 1.63 ms alone). */
template <int KIND, bool WAVE_HAS_CTCSS, int W>
__global__ __launch_bounds__(64 * W, KIND == AB_KIND_AM ? AB_AM_WAVES : KIND == AB_KIND_NFM_CTCSS ? AB_FRONT_WAVES : AB_DEMOD_WAVES) void demod_kernel(DemodArgs a, int first_block, int n_blocks) {
    AB_DYNAMIC_LDS(float, lds_demod);
    demod_block<KIND, WAVE_HAS_CTCSS, W>(a, first_block, n_blocks, lds_demod);
}

/* regroup == 3: the batch's permutation of one kind's slots.  One workgroup of sixteen wavefronts per segment of sixteen blocks (1 024 slots): wg_regroup's stable partition with
 * the line group as the unit -- a group is busy if any of its G slots has a channel that is not at rest in CLOSED (G = 4 where the |bin| ring is read, 2 for the raw-I/Q ring) --
 * written to perm[] where the one-wavefront kernels of the stage pick their slots up.  ~10 us per batch at 65 536 dongles. */
constexpr int AB_PERM_WAVES = 16;
template <int G>
__global__ __launch_bounds__(64 * AB_PERM_WAVES) void regroup_perm_kernel(DemodArgs a, int first_block, int n_blocks) {
    __shared__ int slot_at[AB_PERM_WAVES * 64];
    __shared__ int wcnt[AB_PERM_WAVES * 2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int block0 = first_block + blockIdx.x * AB_PERM_WAVES;
    const int left = first_block + n_blocks - block0, nb = left < AB_PERM_WAVES ? left : AB_PERM_WAVES;
    const int home = (block0 + wave) * 64 + lane;
    int key = 3;
    if (wave < nb) {
        const ChanState* hp = a.cs + home;
        const bool valid = (a.cc[home].flags & AB_F_VALID) != 0;
        const bool busy1 = valid && (hp->cur != AB_ST_CLOSED || hp->next != AB_ST_CLOSED);
        const lmask any_busy = __ballot(busy1);
        const lmask gmask = ((1ull << G) - 1ull) << (lane & ~(G - 1));
        key = !valid ? 2 : (any_busy & gmask) != 0ull ? 0 : 1;
    }
    const int slot = wg_regroup<AB_PERM_WAVES>(key, wave, lane, home, nb, slot_at, wcnt);
    if (slot >= 0) a.perm[home] = slot;
}

/* CTCSS tone detection (reference: src/ctcss.cpp, driven by Squelch::process_audio_sample src/squelch.cpp:278-295).
 * One wavefront per channel; lane t is tone t of the fast (0.05 s) and of the slow (0.4 s) Goertzel bank, so the
 * reference's ~104 multiply-adds per audio sample are one 3-op recurrence per lane.  The channel's whole batch is walked
 * with the state in registers; audio and flags come from the front kernel 50 samples at a time (lane u fetches sample u,
 * v_readlane feeds the serial loop), the verdict leaves as a 50-bit mask per step.  Tiny register footprint -> 8 waves per
 * SIMD hide the dependent-issue latency of the recurrence. */
/* (experiment builds: -DAB_TONE_WAVES8 holds the register allocation to eight waves per SIMD, at the price of a few spilled registers) */
#if defined(AB_TONE_WAVES8)
#define AB_TONE_RESIDENCY __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define AB_TONE_RESIDENCY
#endif
template <bool PACKED>
__global__ __launch_bounds__(256) AB_TONE_RESIDENCY void tone_kernel(DemodArgs a, int first_block, int n_blocks) { /* blocks [first_block, first_block + n_blocks) of the kind: any sub-range */
    __shared__ float power[4][64];
    /* a step's 50 audio samples, parked by the lanes that fetched them and read back as wave-wide BROADCASTS (every lane the same address, four samples per
     * ds_read_b128): the steady-state recurrences then take their input from a vector register -- a v_readlane per sample was a quarter of their vector
     * instructions (round 5: stage 2 is bound by vector issue, its critical path is front -> tone -> back) */
    __shared__ __attribute__((aligned(16))) float xs_all[4][64];
    /* the wave index is the same number on every lane: told so, the compiler keeps the channel's constants and counters in scalar
     * registers and fetches them with scalar loads */
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int lane = threadIdx.x & 63;
    if (wave >= n_blocks * 64) return;
    const int slot = first_block * 64 + wave;
    const unsigned flags = a.cc[slot].flags;
    if (!(flags & AB_F_VALID) || !(flags & AB_F_CTCSS)) return;
    const ChanConst cc = a.cc[slot];
    ChanState* sp = a.cs + slot;
    float* scratch = power[threadIdx.x >> 6];
    float* xs = xs_all[threadIdx.x >> 6];
    const int B = a.wave_batch, NG = B / TONE_GROUP;
    int enough0 = sp->ct_enough[0], enough1 = sp->ct_enough[1], count0 = sp->ct_count[0], count1 = sp->ct_count[1];
    int has0 = sp->ct_has_tone[0], has1 = sp->ct_has_tone[1];
    unsigned found0 = sp->ct_found[0], found1 = sp->ct_found[1], nf0 = sp->ct_not_found[0], nf1 = sp->ct_not_found[1];
    const int n0 = cc.ct_ntones[0], n1 = cc.ct_ntones[1], win0 = cc.ct_window[0], win1 = cc.ct_window[1];
    /* tone tables: [ct_slot][detector][tone] coefficients, [ct_slot][detector][q1|q2][tone] state */
    const float* ctab = a.ct_coeff + (long)cc.ct_slot * 2 * AB_MAX_TONES;
    float* qtab = a.ct_q + (long)cc.ct_slot * 4 * AB_MAX_TONES;
    /* MERGED (round 6): the fast detector's window is an eighth of the slow one's, its tones fall on an eighth as many distinct Goertzel bins (11 against 49 for the
     * standard tone set at 16 kHz), and 11 + 49 <= 64: where both banks fit the wavefront side by side -- slow tone i on lane i, fast tone i on lane n1 + i -- the
     * steady state runs ONE three-operation recurrence per sample for both detectors instead of two.  Both detectors run for the first 0.4 s of a transmission: a
     * third of the kernel's vector instructions on the BASELINE signal.  Otherwise (wave-uniform) lane i holds tone i of both banks as before. */
    const bool merged = n0 + n1 <= 64;
    const int fi = merged ? lane - n1 : lane; /* the fast tone this lane holds, if any */
    const bool t0 = fi >= 0 && fi < n0, t1 = lane < n1;
    const int fbase = merged ? n1 : 0;       /* lane of fast tone 0 */
    const float c0 = t0 ? ctab[fi] : 0.0f, c1 = t1 ? ctab[AB_MAX_TONES + lane] : 0.0f;
    float q1f = t0 ? qtab[fi] : 0.0f, q2f = t0 ? qtab[AB_MAX_TONES + fi] : 0.0f;
    float q1s = t1 ? qtab[2 * AB_MAX_TONES + lane] : 0.0f, q2s = t1 ? qtab[3 * AB_MAX_TONES + lane] : 0.0f;

    const long blk = (wave >> 6) + (first_block - a.ct_first_block); /* verdict masks: one table over both split kinds */
    /* Round 6: a channel whose squelch neither let a sample through nor went CLOSED in this batch (the front kernel's note) has forty idle steps in front of it: the
     * detectors' state cannot change, every step's verdict is the standing one.  More than half of the channels of the BASELINE signal at any time: their wavefronts used
     * to fetch and decode the whole hand-off row to find that out (0.6 GB and ~1e8 vector instructions per batch at configs[2]). */
    if (!a.sq_key[slot]) {
        const unsigned long long standing = ((sp->ct_enough[1] ? sp->ct_has_tone[1] : sp->ct_has_tone[0]) != 0) ? ~0ull : 0ull;
        unsigned long long* mp = a.ct_mask + (blk * (a.wave_batch / TONE_GROUP)) * AB_SLOT_BLOCK + (wave & 63);
        if (lane < a.wave_batch / TONE_GROUP) mp[(long)lane * AB_SLOT_BLOCK] = standing;
        return;
    }
    /* this channel's batch, contiguous: 50 lanes fetch 400 (one-word hand-off: 200) consecutive bytes; rows are counted from the kind's first block */
    const long hrow = wave + (long)(first_block - (PACKED ? a.ct_pk_first_block : a.ct_gen_first_block)) * 64;
    const float2* af = PACKED ? nullptr : a.ct_af + hrow * B;
    const unsigned* ap = PACKED ? a.ct_ap + hrow * a.ct_pk_pitch : nullptr;
    unsigned long long* maskp = a.ct_mask + (blk * NG) * AB_SLOT_BLOCK + (wave & 63);
    /* The batch is walked DEPTH steps at a time: the (audio, flags) pairs of the next DEPTH steps are in flight while the current
     * DEPTH are worked through.  A channel whose squelch is closed does next to nothing per step, so with one step ahead its
     * wavefront walked the batch at one memory round trip per step.  The loads are unconditional (lanes past the 50th re-read
     * sample 49, steps past the last re-read the last): a load behind an `if` makes the compiler wait for everything in flight at
     * the first use of the data.  The verdicts of a group of steps are stored after the next group's loads have been issued, so
     * the wait at the top of a group never sits out stores that have only just left. */
    constexpr int DEPTH = 10; /* divides WAVE_BATCH / TONE_GROUP = 20 / 40 */
    const int ldlane = lane < TONE_GROUP ? lane : TONE_GROUP - 1;
    /* either hand-off format ends up as (audio, flag word) in registers: the one-word format is decoded where it is consumed */
    auto fetch = [&](int g) {
        const int i = (g < NG ? g : NG - 1) * TONE_GROUP + ldlane;
        if (PACKED) return make_float2(__uint_as_float(ap[i]), 0.0f);
        return af[i];
    };
    float2 ahead[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; k++) ahead[k] = fetch(k);
    unsigned long long verdict[DEPTH]; /* wave-uniform: scalar registers */
    auto step = [&](const int g, const float2 got, unsigned long long& mask_out) {
        float2 cur = lane < TONE_GROUP ? got : make_float2(0.0f, 0.0f);
        if (PACKED) { /* word -> (audio, FL_AUDIO | FL_RESET) */
            const unsigned w = lane < TONE_GROUP ? __float_as_uint(got.x) : HAND_IDLE;
            const bool au = hand_is_audio(w);
            cur = make_float2(au ? __uint_as_float(w) : 0.0f, __uint_as_float(au ? FL_AUDIO : ((w & 1u) ? FL_RESET : 0u)));
        }
        const float ax = cur.x;
        const unsigned fl = __float_as_uint(cur.y);
        unsigned long long mask = 0;
        const bool idle = __ballot((fl & (FL_AUDIO | FL_RESET)) != 0) == 0ull;
        const bool all_audio = __ballot(lane < TONE_GROUP && (fl & (FL_AUDIO | FL_RESET)) != FL_AUDIO) == 0ull;
        if (idle) {
            /* squelch closed for the whole step: detector state cannot change */
            mask = (enough1 ? has1 : has0) ? ~0ull : 0ull;
        } else if (all_audio && count1 + TONE_GROUP < win1 && (enough1 || count0 + TONE_GROUP < win0)) {
            /* steady state: squelch open throughout and no detector window ends inside the step -> only the recurrences
             * (ToneDetector::process_sample, src/ctcss.cpp:44-54) */
#if defined(AB_TONE_READLANE) /* experiment builds: round 4's loops (a v_readlane per sample), for A/B timing */
            if (enough1) {
                for (int u = 0; u < TONE_GROUP; u++) {
                    const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ax), u));
                    const float q0 = c1 * q1s - q2s + x;
                    q2s = q1s;
                    q1s = q0;
                }
            } else {
                for (int u = 0; u < TONE_GROUP; u++) {
                    const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ax), u));
                    const float q0 = c1 * q1s - q2s + x;
                    q2s = q1s;
                    q1s = q0;
                    const float p0 = c0 * q1f - q2f + x;
                    q2f = q1f;
                    q1f = p0;
                }
                count0 += TONE_GROUP;
            }
#else
            AB_LOCKSTEP_FENCED(); /* (the previous step's broadcasts have been read: a wavefront's LDS operations complete in order) */
            xs[lane] = ax;  /* lanes past the 50th park zeros that nobody reads */
            AB_LOCKSTEP_FENCED(); /* the park is ordered before every lane's reads, for the compiler too */
            static_assert(TONE_GROUP % 4 == 2 && TONE_GROUP + 2 <= 64, "twelve quads and one pair; the last quad read covers two parked zeros");
            /* one quad ahead (the next broadcast flies under this quad's twelve dependent operations); unrolled by two only (#pragma unroll 2 below): all thirteen reads at once
             * cost 45 more registers and half the kernel's residency, which is what hides the recurrence's dependent-issue latency */
            const float4* xs4 = reinterpret_cast<const float4*>(xs);
            float4 v = xs4[0];
            if (enough1) {
#pragma unroll 2
                for (int u4 = 0; u4 < TONE_GROUP / 4; u4++) {
                    const float4 nxt = xs4[u4 + 1];
                    const float x4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float q0 = c1 * q1s - q2s + x4[r];
                        q2s = q1s;
                        q1s = q0;
                    }
                    v = nxt;
                }
                const float x2[2] = {v.x, v.y};
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const float q0 = c1 * q1s - q2s + x2[r];
                    q2s = q1s;
                    q1s = q0;
                }
            } else if (merged) { /* both detectors, side by side in the lanes: one recurrence (a lane without a tone computes on zeros and is never stored) */
                const float cm = t1 ? c1 : c0;
                float q1m = t1 ? q1s : q1f, q2m = t1 ? q2s : q2f;
#pragma unroll 2
                for (int u4 = 0; u4 < TONE_GROUP / 4; u4++) {
                    const float4 nxt = xs4[u4 + 1];
                    const float x4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float q0 = cm * q1m - q2m + x4[r];
                        q2m = q1m;
                        q1m = q0;
                    }
                    v = nxt;
                }
                const float x2[2] = {v.x, v.y};
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const float q0 = cm * q1m - q2m + x2[r];
                    q2m = q1m;
                    q1m = q0;
                }
                q1s = t1 ? q1m : q1s; q2s = t1 ? q2m : q2s;
                q1f = t1 ? q1f : q1m; q2f = t1 ? q2f : q2m;
                count0 += TONE_GROUP;
            } else { /* the fast detector runs until the slow one has a full window (src/squelch.cpp:288-293) */
#pragma unroll 2
                for (int u4 = 0; u4 < TONE_GROUP / 4; u4++) {
                    const float4 nxt = xs4[u4 + 1];
                    const float x4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float q0 = c1 * q1s - q2s + x4[r];
                        q2s = q1s;
                        q1s = q0;
                        const float p0 = c0 * q1f - q2f + x4[r];
                        q2f = q1f;
                        q1f = p0;
                    }
                    v = nxt;
                }
                const float x2[2] = {v.x, v.y};
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const float q0 = c1 * q1s - q2s + x2[r];
                    q2s = q1s;
                    q1s = q0;
                    const float p0 = c0 * q1f - q2f + x2[r];
                    q2f = q1f;
                    q1f = p0;
                }
                count0 += TONE_GROUP;
            }
#endif
            count1 += TONE_GROUP;
            mask = (enough1 ? has1 : has0) ? ~0ull : 0ull;
        } else {
            for (int u = 0; u < TONE_GROUP; u++) {
                const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ax), u));
                const unsigned f = __builtin_amdgcn_readlane(fl, u);
                if (f & FL_RESET) { /* CTCSS::reset (src/ctcss.cpp:165-172) on both detectors, at the squelch's transition to CLOSED */
                    q1f = q2f = q1s = q2s = 0.0f;
                    enough0 = enough1 = count0 = count1 = has0 = has1 = 0;
                }
                if (f & FL_AUDIO) { /* Squelch::process_audio_sample: slow always, fast until slow has a window */
#pragma unroll
                    for (int k = 1; k >= 0; k--) {
                        if (k == 0 && enough1) break;
                        float& q1 = k ? q1s : q1f;
                        float& q2 = k ? q2s : q2f;
                        const float co = k ? c1 : c0;
                        int& count = k ? count1 : count0;
                        const int win = k ? win1 : win0, n = k ? n1 : n0;
                        const float q0 = co * q1 - q2 + x;
                        q2 = q1;
                        q1 = q0;
                        if (++count >= win) { /* CTCSS::process_audio_sample window end (src/ctcss.cpp:141-162) */
                            scratch[lane] = q1 * q1 + q2 * q2 - q1 * q2 * co;
                            AB_LOCKSTEP(); /* every tone's power is in LDS (the readfirstlane below keeps the next window's writes behind these reads) */
                            float total = 0.0f, best = 0.0f;
                            const int base = k ? 0 : fbase; /* lane of this bank's tone 0 */
                            for (int i = 0; i < n; i++) { /* index-order float sum, as ToneDetectorSet::sorted_powers does */
                                const float m = scratch[base + i];
                                total += m;
                                if (i == 0 || m > best) best = m;
                            }
                            const float target = scratch[base];
                            const float avg = total / (float)n;
                            const bool present = __builtin_amdgcn_readfirstlane((int)(target == best && target > avg)) != 0;
                            if (k) { enough1 = 1; has1 = present; if (present) found1++; else nf1++; }
                            else { enough0 = 1; has0 = present; if (present) found0++; else nf0++; }
                            q1 = 0.0f;
                            q2 = 0.0f;
                            count = 0;
                        }
                    }
                }
                const bool tone = enough1 ? (has1 != 0) : (has0 != 0); /* Squelch::is_open's detector choice (src/squelch.cpp:122-130) */
                if (tone) mask |= 1ull << u;
            }
        }
        mask_out = mask;
    };
    for (int g = 0; g < NG; g += DEPTH) {
        float2 now[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) AB_NEEDED_NOW(AB_V(ahead[k].x), AB_V(ahead[k].y)); /* the data is needed now: the wait lands here, before the next fetches go out */
#pragma unroll
        for (int k = 0; k < DEPTH; k++) now[k] = ahead[k];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) ahead[k] = fetch(g + DEPTH + k);
        if (g > 0 && lane == 0) {
#pragma unroll
            for (int k = 0; k < DEPTH; k++) maskp[(long)(g - DEPTH + k) * AB_SLOT_BLOCK] = verdict[k];
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) step(g + k, now[k], verdict[k]);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < DEPTH; k++) maskp[(long)(NG - DEPTH + k) * AB_SLOT_BLOCK] = verdict[k];
    }
    if (t0) { qtab[fi] = q1f; qtab[AB_MAX_TONES + fi] = q2f; }
    if (t1) { qtab[2 * AB_MAX_TONES + lane] = q1s; qtab[3 * AB_MAX_TONES + lane] = q2s; }
    if (lane == 0) {
        sp->ct_enough[0] = enough0; sp->ct_enough[1] = enough1; sp->ct_count[0] = count0; sp->ct_count[1] = count1;
        sp->ct_has_tone[0] = has0; sp->ct_has_tone[1] = has1;
        sp->ct_found[0] = found0; sp->ct_found[1] = found1; sp->ct_not_found[0] = nf0; sp->ct_not_found[1] = nf1;
    }
}

/* Back half of the split kinds: output gating (squelch open AND tone present), notch, ampfactor, clamp, AM fade-out
 * (reference: src/rtl_airband.cpp:532-547,589-620), one lane per channel, finished output runs of 32 samples staged through LDS and stored as whole lines. */
template <bool PACKED, int W>
__global__ __launch_bounds__(64 * W) void back_kernel(DemodArgs a, int first_block, int n_blocks) {
    __shared__ float staged_all[W][RUN][OSTRIDE];
    __shared__ int ext_of_all[W][64];
    __shared__ int skip_of_all[W][64];
    __shared__ int slot_at[W * 64];
    __shared__ int wcnt[W * 2];
    const int lane = threadIdx.x & 63;
    const int wave = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    float (*staged)[OSTRIDE] = staged_all[wave];
    int* ext_of = ext_of_all[wave];
    int* skip_of = skip_of_all[wave];
    int slot;
    if (W == 1) {
        slot = (first_block + blockIdx.x) * 64 + lane;
    } else { /* regrouped handles (demod_block): the channels that had audio in this batch -- the front kernel's note -- to the first wavefronts */
        const int block0 = first_block + blockIdx.x * W;
        const int left = first_block + n_blocks - block0, nb = left < W ? left : W;
        const int home = (block0 + wave) * 64 + lane;
        int key = 3;
        if (wave < nb) key = !(a.cc[home].flags & AB_F_VALID) ? 2 : a.sq_key[home] ? 0 : 1;
        slot = wg_regroup<W>(key, wave, lane, home, nb, slot_at, wcnt);
        if (slot < 0) return;
    }
    const ChanConst cc = a.cc[slot];
    ext_of[lane] = a.slot_to_ext[slot];
    const bool full_block = __ballot((cc.flags & AB_F_VALID) != 0) == ~0ull;
    if (W == 1) __syncthreads();
    else AB_LOCKSTEP_FENCED();
    if (!(cc.flags & AB_F_VALID)) return;
    ChanState* sp = a.cs + slot;
    const int B = a.wave_batch, NG = B / TONE_GROUP;
    OutRegs o;
    o.nx0 = sp->nx[0]; o.nx1 = sp->nx[1]; o.nx2 = sp->nx[2]; o.ny0 = sp->ny[0]; o.ny1 = sp->ny[1]; o.ny2 = sp->ny[2];
    o.axc = ' ';
    const int ext = a.slot_to_ext[slot];
    WaveRow w;
    w.row = a.out_wave + (long)ext * a.wave_stride + AB_OUT_PAD;
    w.staged = &staged[0][lane];
    w.stride = OSTRIDE;
    w.j0 = 0;
    w.stage_base = &staged[0][0];
    w.ext_of = ext_of;
    w.out_wave = a.out_wave;
    w.wave_stride = a.wave_stride;
    w.coop = full_block;
    w.skip_of = skip_of;
    RowZero rz = {sp->row_zero, false, false};
    if (a.tail_copy) row_tail_copy(rz, w.row, B); /* src/output.cpp:920 */
    float2* iqout = a.iq_out + ab_ring_base(slot, B);
    uint8_t* trace = a.trace ? a.trace + ab_ring_base(slot, B) : nullptr;
    /* every lane walks its own contiguous (channel-major) hand-off row, 8 samples = 4 x 16 bytes ahead: all bytes of the
     * lines it touches are its own, the re-use is served by L2 */
    const float4* af = PACKED ? reinterpret_cast<const float4*>(a.ct_ap + (long)(slot - a.ct_pk_first_block * 64) * a.ct_pk_pitch)
                              : reinterpret_cast<const float4*>(a.ct_af + (long)(slot - a.ct_gen_first_block * 64) * B);
    /* the tone kernel (one wavefront per SLOT) leaves a channel's verdicts at (slot's block, slot's lane) */
    const unsigned long long* maskp = a.ct_mask + ((long)((slot >> 6) - a.ct_first_block) * NG) * AB_SLOT_BLOCK + (slot & 63);
    const bool is_ct = (cc.flags & AB_F_CTCSS) != 0;
    constexpr int PIECE = 8; /* samples per fetch; WAVE_BATCH = 1000 / 2000 is a whole number of them */
    /* A piece = 8 (audio, flags) pairs + the tone kernel's verdict masks of the one or two 50-sample steps it lies in.  Piece k + 1
     * flies while piece k is worked through; every load is unconditional (the piece after the last re-reads the last one, a lane
     * without CTCSS reads a mask nobody wrote and ignores it), the wait sits right at the top, and a finished output run leaves
     * only after the next fetch is out -- see demod_wave for why each of these matters. */
    constexpr int NQ = PACKED ? PIECE / 4 : PIECE / 2; /* 16-byte loads per piece */
    float4 nxt[NQ];
    unsigned long long nm_lo, nm_hi;
    auto fetch = [&](int j) {
#pragma unroll
        for (int q = 0; q < NQ; q++) nxt[q] = af[j / (PIECE / NQ) + q];
        const int gl = j / TONE_GROUP;
        nm_lo = maskp[(long)gl * AB_SLOT_BLOCK];
        nm_hi = maskp[(long)(gl + 1 < NG ? gl + 1 : NG - 1) * AB_SLOT_BLOCK];
    };
    fetch(0);
    int jg = 0; /* position of the current sample in its tone-kernel step */
    for (int j0 = 0; j0 < B; j0 += PIECE) {
#if !defined(AB_REGROUP_FREE)
        if (W > 1 && a.regroup != 2) __syncthreads(); /* regrouped handles: the workgroup's wavefronts walk the batch in step (demod_wave) */
#endif
#pragma unroll
        for (int q = 0; q < NQ; q++) AB_NEEDED_NOW(AB_V(nxt[q].x), AB_V(nxt[q].y), AB_V(nxt[q].z), AB_V(nxt[q].w));
        AB_NEEDED_NOW(AB_V(nm_lo), AB_V(nm_hi));
        float4 cur[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) cur[q] = nxt[q];
        unsigned long long mask = is_ct ? nm_lo : ~0ull;
        const unsigned long long mask_hi = is_ct ? nm_hi : ~0ull;
        fetch(j0 + PIECE < B ? j0 + PIECE : B - PIECE);
        if (j0 > 0 && (j0 % RUN) == 0) wave_flush(w, rz);
        if ((j0 % RUN) == 0) w.j0 = j0;
#pragma unroll
        for (int u = 0; u < PIECE; u++) {
            const bool tone = ((mask >> jg) & 1ull) != 0;
            if (PACKED) {
                const float4 p = cur[u >> 2];
                const unsigned wd = __float_as_uint((u & 3) == 0 ? p.x : (u & 3) == 1 ? p.y : (u & 3) == 2 ? p.z : p.w);
                const bool au = hand_is_audio(wd);
                /* the squelch state the trace records was left in the trace buffer by the front kernel (debug handles only) */
                const int st = trace ? (int)(trace[(long)(j0 + u) * AB_SLOT_BLOCK] & 7u) : 0;
                emit_sample(a, cc, o, w, rz, iqout, trace, j0 + u, au, false, tone, st, au ? __uint_as_float(wd) : 0.0f, 0.0f, 0.0f, false);
            } else {
                const float4 p = cur[u >> 1];
                const float x = (u & 1) ? p.z : p.x;
                const unsigned f = __float_as_uint((u & 1) ? p.w : p.y);
                emit_sample(a, cc, o, w, rz, iqout, trace, j0 + u, (f & FL_AUDIO) != 0, (f & FL_FADE) != 0, tone, (int)((f >> FL_STATE_SHIFT) & 7u), x, 0.0f, 0.0f, false);
            }
            if (++jg == TONE_GROUP) { /* wave-uniform: every lane is on the same sample */
                jg = 0;
                mask = mask_hi;
            }
        }
    }
    if ((B % RUN) == 0) wave_flush(w, rz); /* the last whole run (a short one is handled below) */
    if ((B % RUN) != 0) wave_flush(w, rz, B % RUN);
    sp->row_zero = rz.batch_open ? 0 : ((rz.held & 1) | 2);
    if (o.axc != ' ') sp->active_counter++;
    sp->axc_prev = sp->axc;
    sp->axc = o.axc;
    a.out_axc[ext] = (uint8_t)o.axc;
    sp->nx[0] = o.nx0; sp->nx[1] = o.nx1; sp->nx[2] = o.nx2; sp->ny[0] = o.ny0; sp->ny[1] = o.ny1; sp->ny[2] = o.ny2;
}

/* The kinds are independent of each other (different channels), and none of their kernels fills the chip on its own: a
 * lane-per-channel kernel holds at most 4 waves per SIMD and the NFM kinds have only half that many wavefronts at BASELINE
 * config #3.  So the split chain (front -> tone -> back, the longest) goes on the caller's stream and the fused kinds run
 * beside it on side streams, forked and joined with events (works the same under graph capture). */

/* threads per workgroup of the tone kernel (one wavefront per channel either way): 256, or 64 with AIRBAND_HIP_TONE_THREADS=64 (round 5 experiment, profiles/r05_event_hunt.md) */
static int tone_threads() {
    static const int n = [] {
        const char* e = getenv("AIRBAND_HIP_TONE_THREADS");
        return (e && atoi(e) == 64) ? 64 : 256;
    }();
    return n;
}

void launch_demod(const DemodArgs& a, const int* kind_first_block, const int* kind_n_blocks, hipStream_t stream, hipStream_t* side, hipEvent_t* ev) {
    constexpr int RW = AB_REGROUP_WAVES;
    const bool rg = a.regroup == 1 || a.regroup == 2; /* the workgroup forms; 3: one-wavefront workgroups behind a permutation */
    if (a.regroup == 3 && a.perm) {
        for (int k = 0; k < AB_KIND_COUNT; k++) {
            const int n = kind_n_blocks[k], f = kind_first_block[k];
            if (n <= 0) continue;
            const dim3 grid((n + AB_PERM_WAVES - 1) / AB_PERM_WAVES), block(64 * AB_PERM_WAVES);
            if (k == AB_KIND_AM || k == AB_KIND_GENERIC) hipLaunchKernelGGL((regroup_perm_kernel<4>), grid, block, 0, stream, a, f, n);
            else hipLaunchKernelGGL((regroup_perm_kernel<2>), grid, block, 0, stream, a, f, n);
        }
    }
    auto lds_of = [&](int k) { /* sincos table; per wavefront: output-line staging, ext_of, skip_of, slot_of; regrouped: the workgroup's slot table and counts */
        return (size_t)(k == AB_KIND_AM ? 0 : 258 * sizeof(float2)) + (size_t)(rg ? RW : 1) * WAVE_LDS_FLOATS * sizeof(float) + (rg ? (size_t)(RW * 64 + 2 * RW) * sizeof(int) : 0);
    };
    auto launch_kind = [&](int k, hipStream_t s) {
        const size_t lds = lds_of(k);
        const int n = kind_n_blocks[k], f = kind_first_block[k];
        if (n <= 0) return;
        const dim3 grid(rg ? (n + RW - 1) / RW : n), block(rg ? 64 * RW : 64);
#define AB_LAUNCH_KIND(KIND, CT)                                                                     \
    if (rg) hipLaunchKernelGGL((demod_kernel<KIND, CT, RW>), grid, block, lds, s, a, f, n);          \
    else hipLaunchKernelGGL((demod_kernel<KIND, CT, 1>), grid, block, lds, s, a, f, n)
        switch (k) {
            case AB_KIND_AM: AB_LAUNCH_KIND(AB_KIND_AM, false); break;
            case AB_KIND_NFM: AB_LAUNCH_KIND(AB_KIND_NFM, false); break;
            case AB_KIND_NFM_LOWPASS: AB_LAUNCH_KIND(AB_KIND_NFM_LOWPASS, false); break;
            case AB_KIND_NFM_CTCSS: AB_LAUNCH_KIND(AB_KIND_NFM_CTCSS, true); break;
            default: AB_LAUNCH_KIND(AB_KIND_GENERIC, true); break;
        }
#undef AB_LAUNCH_KIND
    };
    const bool fork = side != nullptr && ev != nullptr;
    const int fused[3] = {AB_KIND_NFM_LOWPASS, AB_KIND_NFM, AB_KIND_AM};
    if (fork) (void)hipEventRecord(ev[0], stream); /* stage 1 is done at this point of the caller's stream */
    /* the split chain (front -> tone -> back) is the longest dependent sequence of stage 2: it is enqueued first (and its stream has
     * the higher priority), the fused kinds fill in beside it.  (Round 3 also ran the chain as 2 and 4 independent chains over block
     * ranges on streams of their own, so that one range's tone and back kernels would overlap the next range's front: 6.81 / 6.92 and
     * 6.50 / 6.69 ms against 6.70 / 6.62 -- nothing; and the AM kind BEHIND the chain on this stream instead of beside it: 6.10 against 5.85 - 5.97,
     * worse -- profiles/r03_experiments.md B, H; the stage is bound by the sum of its work.) */
    launch_kind(AB_KIND_NFM_CTCSS, stream);
    launch_kind(AB_KIND_GENERIC, stream);
    const int tt = tone_threads();
    if (a.ct_pk_n_blocks > 0) hipLaunchKernelGGL(tone_kernel<true>, dim3((a.ct_pk_n_blocks * 64 * 64 + tt - 1) / tt), dim3(tt), 0, stream, a, a.ct_pk_first_block, a.ct_pk_n_blocks);
    if (a.ct_gen_n_blocks > 0) hipLaunchKernelGGL(tone_kernel<false>, dim3((a.ct_gen_n_blocks * 64 * 64 + tt - 1) / tt), dim3(tt), 0, stream, a, a.ct_gen_first_block, a.ct_gen_n_blocks);
    if (a.ct_pk_n_blocks > 0) {
        if (rg) hipLaunchKernelGGL((back_kernel<true, RW>), dim3((a.ct_pk_n_blocks + RW - 1) / RW), dim3(64 * RW), 0, stream, a, a.ct_pk_first_block, a.ct_pk_n_blocks);
        else hipLaunchKernelGGL((back_kernel<true, 1>), dim3(a.ct_pk_n_blocks), dim3(64), 0, stream, a, a.ct_pk_first_block, a.ct_pk_n_blocks);
    }
    if (a.ct_gen_n_blocks > 0) {
        if (rg) hipLaunchKernelGGL((back_kernel<false, RW>), dim3((a.ct_gen_n_blocks + RW - 1) / RW), dim3(64 * RW), 0, stream, a, a.ct_gen_first_block, a.ct_gen_n_blocks);
        else hipLaunchKernelGGL((back_kernel<false, 1>), dim3(a.ct_gen_n_blocks), dim3(64), 0, stream, a, a.ct_gen_first_block, a.ct_gen_n_blocks);
    }
    /* the fused kinds as forked launches (one kernel per kind keeps each kind's own register budget: the AM kind runs four waves per
     * SIMD, the heavier ones three; a single launch with host-interleaved blocks was measured equal at best, 7.8 vs 7.7 ms);
     * without side streams (AIRBAND_HIP_FLAG_SERIAL_DEMOD, profiling) one after the other */
    /* a handle without CTCSS-capable channels has nothing on the caller's stream: its first fused kind runs there, not on a side stream behind
     * two cross-queue dependencies (BASELINE configs[1], AM only: 0.05 ms of a 0.32 ms stage) */
    bool caller_stream_free = a.ct_pk_n_blocks <= 0 && a.ct_gen_n_blocks <= 0 && kind_n_blocks[AB_KIND_NFM_CTCSS] <= 0 && kind_n_blocks[AB_KIND_GENERIC] <= 0;
    for (int i = 0; i < 3; i++) {
        if (kind_n_blocks[fused[i]] <= 0) continue;
        const bool on_side = fork && !caller_stream_free;
        caller_stream_free = false;
        hipStream_t s = on_side ? side[i] : stream;
        if (on_side) (void)hipStreamWaitEvent(s, ev[0], 0);
        launch_kind(fused[i], s);
        if (on_side) {
            (void)hipEventRecord(ev[1 + i], s);
            (void)hipStreamWaitEvent(stream, ev[1 + i], 0);
        }
    }
}

/* ---- raw I/Q outputs: time-major device rows -> the channel-major layout the output thread consumes (reference:
 * src/output.cpp:521,535 read channel->iq_out).  Only handles with has_iq_outputs channels run this; audio needs no
 * such pass, the demod kernels write channel->waveout rows directly.  64 slots x 64 samples per block, transposed
 * through LDS so both sides are coalesced. */
__global__ __launch_bounds__(256) void emit_iq_kernel(EmitArgs a) {
    __shared__ float tile[64][65];
    const int slot0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int B = a.wave_batch;
    const int n_tiles = (B + 63) / 64;
    const int per = (n_tiles + gridDim.y - 1) / gridDim.y;
    const int tile_begin = blockIdx.y * per, tile_end = min(n_tiles, tile_begin + per);
    for (int tl = tile_begin; tl < tile_end; tl++) {
        for (int comp = 0; comp < 2; comp++) {
            __syncthreads();
            for (int r = ty; r < 64; r += 4) {
                const int t = tl * 64 + r;
                float x = 0.0f;
                if (t < B && slot0 + tx < a.n_slots) {
                    const float2 q = a.iq_out[((long)blockIdx.x * B + t) * AB_SLOT_BLOCK + tx];
                    x = comp ? q.y : q.x;
                }
                tile[r][tx] = x;
            }
            __syncthreads();
            for (int r = ty; r < 64; r += 4) {
                const int slot = slot0 + r, t = tl * 64 + tx;
                if (slot < a.n_slots && t < B) {
                    const int ext = a.slot_to_ext[slot];
                    if (ext >= 0) a.out_iq[((long)ext * B + t) * 2 + comp] = tile[tx][r];
                }
            }
        }
    }
}

/* axcindicate after AFC has had its say (afc.finalize() may turn '*' into '<' / '>', src/rtl_airband.cpp:626-630) */
__global__ void axc_kernel(const ChanConst* cc, const ChanState* cs, const int* slot_to_ext, uint8_t* out_axc, int n_slots) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    const int ext = slot_to_ext[slot];
    if (ext >= 0 && (cc[slot].flags & AB_F_VALID)) out_axc[ext] = (uint8_t)cs[slot].axc; /* channels of a disabled dongle keep their NO_SIGNAL */
}

void launch_emit_iq(const EmitArgs& a, hipStream_t stream) {
    const int blocks = (a.n_slots + 63) / 64;
    if (blocks <= 0 || !a.out_iq) return;
    const int n_tiles = (a.wave_batch + 63) / 64;
    int ysplit = blocks >= 4096 ? 2 : (16384 / blocks);
    if (ysplit < 1) ysplit = 1;
    if (ysplit > n_tiles) ysplit = n_tiles;
    hipLaunchKernelGGL(emit_iq_kernel, dim3(blocks, ysplit), dim3(256), 0, stream, a);
}

void launch_axc(const ChanConst* cc, const ChanState* cs, const int* slot_to_ext, uint8_t* out_axc, int n_slots, hipStream_t stream) {
    hipLaunchKernelGGL(axc_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, stream, cc, cs, slot_to_ext, out_axc, n_slots);
}

/* ---- stats mirror (reference getters: src/output.cpp:617-761) ---------------------------------------------- */
__global__ void stats_kernel(const ChanConst* cc, const ChanState* cs, const int* slot_to_ext, int n_slots, airband_hip_channel_stats* out) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    const int ext = slot_to_ext[slot];
    if (ext < 0) return;
    const ChanConst c = cc[slot];
    const ChanState s = cs[slot];
    airband_hip_channel_stats o;
    o.noise_level = s.noise_floor;
    o.signal_level = s.pre_full;
    float lvl;
    if (c.flags & AB_F_MANUAL) lvl = c.sq_manual_level;
    else lvl = ((s.recent_open >= 3u && c.sq_flappy_ratio < c.sq_normal_ratio) ? c.sq_flappy_ratio : c.sq_normal_ratio) * s.noise_floor;
    o.squelch_level = lvl;
    o.agcavgfast = s.agcavgfast;
    o.open_count = s.open_count;
    o.flappy_count = s.flappy_count;
    o.ctcss_count = s.ct_found[1];
    o.no_ctcss_count = s.ct_not_found[1];
    o.active_counter = s.active_counter;
    o.bin = s.bin;
    o.squelch_state = s.cur;
    /* Squelch::signal_outside_filter() (src/squelch.cpp:152-154): using_post_filter_ && has_pre_filter_signal() && !has_post_filter_signal(), the latter
     * against buffer_[buffer_tail_] as the batch's last sample left it (ChanState::sh_dly) */
    o.signal_outside_filter = (s.using_post && s.pre_capped >= lvl && !(s.post_capped >= s.sh_dly)) ? 1 : 0;
    o.reserved = 0;
    out[ext] = o;
}

void launch_stats(const ChanConst* cc, const ChanState* cs, const int* slot_to_ext, int n_slots, airband_hip_channel_stats* out, hipStream_t stream) {
    hipLaunchKernelGGL(stats_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, stream, cc, cs, slot_to_ext, n_slots, out);
}

}  // namespace airband
