/* csrc/squelch_fsm.h -- the reference's Squelch (src/squelch.cpp, state list src/squelch.h:117-158) as the demod kernels run it:
 * one lane per channel, no divergent control flow around the state machine.
 *
 * Shape of the code, and why:
 *  - Every per-channel BOOLEAN (which Squelch::State, using_post_filter_, every predicate) is kept as a wave-wide LANE MASK
 *    (`lmask`, bit l = lane l's value) instead of a per-lane variable.  Lane masks live in scalar register pairs, so the
 *    whole five-way state machine -- which transition is legal, which predicate the caller sees -- is scalar-unit bit
 *    arithmetic and costs no vector-ALU slot.  The lane-per-channel kernels are vector-ALU bound (a wave64 VALU instruction
 *    occupies the SIMD for four cycles), so this is where the time goes.  Only counters, float compares (one v_cmp each,
 *    `ab_ballot`) and the moving averages remain vector work; a mask steers a per-lane select through `ab_lane`
 *    (v_cndmask with the scalar pair as its condition).
 *  - Lane masks are wave-uniform values: they are only ever assigned in wave-uniform control flow.  Per-lane conditions
 *    are folded in with mask arithmetic, never with an `if` around a mask update.
 *  - Rare events (a state change, a timer running out, the flap counter being cleared) sit behind wave-uniform
 *    `ab_any()` branches: the body is written with per-lane selects and is correct for every lane whenever it runs, so it
 *    is skipped exactly when no lane of the wavefront needs it.
 *  - squelch_level() is kept as an always-valid cache `lvl`: the reference invalidates its cache exactly when
 *    noise_floor_ or recent_open_count_ change (:389,:451,:489) and recomputes the same product, so refreshing the value at
 *    those three places yields the float every call of the reference would see.
 *
 * The header also compiles as plain C++, where a "wavefront" is one lane and a lane mask is one bit: tests/ builds it with
 * g++ and checks it sample by sample against the oracle (tests/test_host_fsm.py) -- that is a test of this logic, not a CPU
 * code path of the product; nothing in the library calls it on the host.
 */
#ifndef AIRBAND_CSRC_SQUELCH_FSM_H
#define AIRBAND_CSRC_SQUELCH_FSM_H

#include "common.h"

namespace airband {

typedef unsigned long long lmask;

#if defined(__HIPCC__)
#define AB_FSM_FN __device__ __forceinline__
AB_FSM_FN lmask ab_ballot(bool b) { return __ballot(b); }                                   /* per-lane bool -> lane mask (one v_cmp) */
AB_FSM_FN bool ab_lane(lmask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }          /* lane mask -> this lane's bool        */
AB_FSM_FN bool ab_any(lmask m) { return m != 0ull; }
AB_FSM_FN unsigned ab_uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); } /* a value every lane holds alike: keep it in a scalar register */
#define AB_LIKELY(x) __builtin_expect(!!(x), 1)
#define AB_UNLIKELY(x) __builtin_expect(!!(x), 0)
AB_FSM_FN bool ab_rare(lmask m) { return AB_UNLIKELY(m != 0ull); } /* ab_any() of an event that is seldom there: its code goes out of line */
#else
#define AB_FSM_FN static inline
AB_FSM_FN lmask ab_ballot(bool b) { return b ? 1ull : 0ull; }
AB_FSM_FN bool ab_lane(lmask m) { return (m & 1ull) != 0ull; }
AB_FSM_FN bool ab_any(lmask m) { return (m & 1ull) != 0ull; }
AB_FSM_FN unsigned ab_uniform(unsigned v) { return v; }
#define AB_LIKELY(x) (x)
#define AB_UNLIKELY(x) (x)
AB_FSM_FN bool ab_rare(lmask m) { return (m & 1ull) != 0ull; }
#endif

struct SqRegs { /* Squelch members that change per sample */
    float noise_floor, cap, pre_full, pre_capped, post_full, post_capped;
    float lvl;                    /* squelch_level() of the current noise_floor_ / recent_open_count_ */
    lmask active;                 /* lanes that own a channel */
    lmask using_post;             /* using_post_filter_ */
    lmask recent_nz;              /* recent_open_count_ != 0 (kept beside the counter: it changes on three seldom-run paths only) */
    lmask cC, cOg, cCg, cA, cO;   /* current_state_ == CLOSED / OPENING / CLOSING / LOW_SIGNAL_ABORT / OPEN */
    lmask nC, nOg, nCg, nA, nO;   /* next_state_ */
    int delay, low_count, head, tail;
    unsigned sample_count, open_count, flappy_count, recent_open, closed_count;
    float dly; /* buffer_[buffer_tail_] for the current tail, when the kind prefetches the delay line (else unused) */
    /* QUIET (wave-uniform): every lane's next_state_ equals its current_state_, and that state is CLOSED or OPEN -- no timer runs,
     * nothing is being entered, nothing can expire.  Keyed transmissions spend >= 85 % of their samples like this, and then the
     * transition algebra of update_current_state() is a no-op: sq_raw() skips it and only watches for the next request. */
    bool quiet;
};

struct Lane { /* per-lane constants */
    bool prefetched_delay; /* SqRegs::dly is maintained by the caller instead of reading sqbuf from memory (compile-time per kind) */
    bool track_delay_line; /* head/tail advance per sample (only kinds that touch the delay line need them inside a batch) */
    bool may_post_filter;  /* some lane of this kind may have a lowpass filter, i.e. using_post_filter_ can ever be set (compile-time per kind) */
    bool all_lowpass;      /* every lane of this kind has one (compile-time per kind) */
    bool shadow_delay;     /* the caller recomputes the delay line's entries (SqShadow) instead of storing them: nothing is pushed (compile-time per kind) */
    lmask m_lowpass, m_manual, m_flappy_lower;
    float manual_level, normal_ratio, flappy_ratio;
    float* sqbuf; /* this lane's column of the 102-deep pre-filter delay line, stride S */
    long S;
};

AB_FSM_FN void sq_set_cur(SqRegs& s, int st) {
    s.cC = ab_ballot(st == AB_ST_CLOSED); s.cOg = ab_ballot(st == AB_ST_OPENING); s.cCg = ab_ballot(st == AB_ST_CLOSING);
    s.cA = ab_ballot(st == AB_ST_ABORT); s.cO = ab_ballot(st == AB_ST_OPEN);
}
AB_FSM_FN void sq_set_next(SqRegs& s, int st) {
    s.nC = ab_ballot(st == AB_ST_CLOSED); s.nOg = ab_ballot(st == AB_ST_OPENING); s.nCg = ab_ballot(st == AB_ST_CLOSING);
    s.nA = ab_ballot(st == AB_ST_ABORT); s.nO = ab_ballot(st == AB_ST_OPEN);
}
AB_FSM_FN int sq_cur(const SqRegs& s) {
    return ab_lane(s.cO) ? AB_ST_OPEN : ab_lane(s.cA) ? AB_ST_ABORT : ab_lane(s.cCg) ? AB_ST_CLOSING : ab_lane(s.cOg) ? AB_ST_OPENING : AB_ST_CLOSED;
}
AB_FSM_FN int sq_next(const SqRegs& s) {
    return ab_lane(s.nO) ? AB_ST_OPEN : ab_lane(s.nA) ? AB_ST_ABORT : ab_lane(s.nCg) ? AB_ST_CLOSING : ab_lane(s.nOg) ? AB_ST_OPENING : AB_ST_CLOSED;
}

/* Squelch::squelch_level() (src/squelch.cpp:164-177) */
AB_FSM_FN float sq_level_compute(const SqRegs& s, const Lane& L) {
    const lmask flappy = ab_ballot(s.recent_open >= 3u /* flap_opens_threshold_ */) & L.m_flappy_lower;
    const float ratio = ab_lane(flappy) ? L.flappy_ratio : L.normal_ratio;
    const float lvl = ratio * s.noise_floor;
    return ab_lane(L.m_manual) ? L.manual_level : lvl;
}
AB_FSM_FN float sq_level(const SqRegs& s) { return s.lvl; }

AB_FSM_FN float sq_delayed(const SqRegs& s, const Lane& L) { return L.prefetched_delay ? s.dly : L.sqbuf[(long)s.tail * L.S]; }

AB_FSM_FN lmask sq_has_pre(const SqRegs& s) { return ab_ballot(s.pre_capped >= s.lvl); }

AB_FSM_FN lmask sq_has_signal(const SqRegs& s, const Lane& L) { /* src/squelch.cpp:462-475 */
    lmask sig = sq_has_pre(s);
    /* using_post_filter_ can only ever be set on channels with a lowpass filter */
    if (L.may_post_filter && ab_any(s.using_post)) sig &= ~s.using_post | ab_ballot(s.post_capped >= sq_delayed(s, L));
    return sig;
}

/* CLOSED lanes that have counted their 1000 closed samples and still have recent opens on record: the next update_current_state()
 * forgets those (src/squelch.cpp:444-452) and refreshes the squelch level -- work of the general version */
AB_FSM_FN lmask sq_saturated(const SqRegs& s) { return s.cC & ~ab_ballot(s.closed_count < 1000u) & s.recent_nz; }

AB_FSM_FN bool sq_is_quiet(const SqRegs& s) {
    const lmask busy = (s.nC ^ s.cC) | (s.nO ^ s.cO) | s.cOg | s.cCg | s.cA | s.nOg | s.nCg | s.nA | sq_saturated(s);
    return !ab_any(busy & s.active);
}

/* Squelch::update_current_state (src/squelch.cpp:363-460).  Returns the lanes whose squelch just went CLOSED (the reference
 * resets both CTCSS detectors at that point, :440-441). */
AB_FSM_FN lmask sq_advance(SqRegs& s, const Lane& L) {
    const lmask same = (s.nC & s.cC) | (s.nOg & s.cOg) | (s.nCg & s.cCg) | (s.nA & s.cA) | (s.nO & s.cO);
    const lmask entering = s.active & ~same;
    const lmask timed = s.nOg | s.nCg | s.nA;
    const lmask staying = timed & same;
    const lmask idle_closed = s.nC & same;
    lmask went_closed = 0;
    s.delay += ab_lane(staying) ? 1 : 0; /* the timer of OPENING / CLOSING / LOW_SIGNAL_ABORT */
    if (ab_rare(entering)) {
        /* delay_ is zeroed on entry to a timed state, except that ABORT entered from CLOSING keeps CLOSING's running delay */
        const lmask zero_delay = entering & timed & ~(s.nA & s.cCg);
        s.delay = ab_lane(zero_delay) ? 0 : s.delay;
        s.low_count = ab_lane(entering & s.nOg) ? 0 : s.low_count;
        s.using_post &= ~(entering & (s.nOg | s.nC));
        s.open_count += ab_lane(entering & s.nO) ? 1u : 0u;
        went_closed = entering & s.nC;
        s.closed_count = ab_lane(went_closed) ? 0u : s.closed_count;
    }
    /* current_state_ = next_state_ (when nothing is entered the two are equal already) */
    s.cC = s.nC; s.cOg = s.nOg; s.cCg = s.nCg; s.cA = s.nA; s.cO = s.nO;
    const lmask expired = staying & ab_ballot(s.delay >= 197); /* open_delay_ == close_delay_ == 197 */
    if (ab_rare(expired)) {
        /* OPENING delay over: count a recent open for flap detection before looking at the signal (:381-392) */
        const lmask bump = expired & s.nOg & ab_ballot(s.closed_count < 1000u);
        s.recent_open += ab_lane(bump) ? 1u : 0u;
        s.recent_nz |= bump;
        s.flappy_count += ab_lane(bump & ab_ballot(s.recent_open >= 3u)) ? 1u : 0u;
        s.lvl = sq_level_compute(s, L);
        const lmask sig = sq_has_signal(s, L);
        const lmask to_open = expired & (s.nOg | s.nCg) & sig; /* OPENING -> OPEN; CLOSING with the signal back -> OPEN */
        const lmask reopen = expired & s.nCg & sig;            /* ... the latter straight away, without counting an open */
        s.cCg &= ~reopen;
        s.cO |= reopen;
        s.nO |= to_open;
        s.nC |= expired & ~to_open;
        s.nOg &= ~expired;
        s.nCg &= ~expired;
        s.nA &= ~expired;
    }
    /* CLOSED and staying there: count closed samples up to recent_sample_size_ = 1000, then forget the recent opens */
    const lmask below = ab_ballot(s.closed_count < 1000u);
    const lmask forget = idle_closed & ~below & s.recent_nz;
    s.closed_count += ab_lane(idle_closed & below) ? 1u : 0u;
    if (ab_rare(forget)) {
        s.recent_open = ab_lane(forget) ? 0u : s.recent_open;
        s.recent_nz &= ~forget;
        s.lvl = sq_level_compute(s, L);
    }
    if (L.track_delay_line) {
        s.tail = s.tail + 1 == AB_SQ_BUF ? 0 : s.tail + 1;
        s.head = s.head + 1 == AB_SQ_BUF ? 0 : s.head + 1;
    }
    return went_closed;
}

/* Squelch::update_moving_avg (src/squelch.cpp:501-514) */
AB_FSM_FN void sq_avg(float cap, float& full, float& capped, float x) {
    const float decay = 0.99f;
    const float fresh = (float)(1.0 - (double)0.99f);
    const float xf = x * fresh;
    full = full * decay + xf;
    const float v = capped * decay + xf;
    const float vm = v < cap ? v : cap; /* std::min(moving_avg_cap_, v) as libstdc++ spells it, (v < cap) ? v : cap: a NaN -- an unstable lowpass, bandwidth above WAVE_RATE -- yields the cap */
    capped = (capped >= cap && x >= cap) ? cap : vm; /* the reference short-circuits this case; the value is `cap` either way it is written */
}

/* buffer_[buffer_head_] = pre_filter_.capped_ * pre_vs_post_factor_ (src/squelch.cpp:218-219): only ever read on the post-filter path,
 * i.e. by channels with a lowpass filter -- the other lanes skip the store.  (all_lowpass: the lane test of an all-ones mask is not
 * folded by the compiler, so the kind that has the filter on every lane says so.) */
AB_FSM_FN void sq_delay_line_push(const SqRegs& s, const Lane& L) {
    if (!L.may_post_filter || L.shadow_delay) return;
    if (L.all_lowpass) {
        L.sqbuf[(long)s.head * L.S] = s.pre_capped * 0.9f;
    } else if (ab_any(L.m_lowpass)) {
        if (ab_lane(L.m_lowpass)) L.sqbuf[(long)s.head * L.S] = s.pre_capped * 0.9f;
    }
}

/* calculate_noise_floor(), every 16th sample (src/squelch.cpp:477-490).  sample_count_ starts at SIZE_MAX on every channel and counts
 * every sample of every channel, so it is the same number on all lanes: the test is scalar and `sweep` is all lanes or none. */
AB_FSM_FN void sq_noise_floor(SqRegs& s, const Lane& L) {
    const float decay = 0.97f;
    const float fresh = (float)(1.0 - (double)0.97f);
    const float lo = s.noise_floor < s.pre_capped ? s.noise_floor : s.pre_capped; /* std::min(pre_filter_.capped_, noise_floor_), src/squelch.cpp:481 */
    s.noise_floor = s.noise_floor * decay + lo * fresh + 1e-6f;
    s.cap = ab_lane(L.m_manual) ? 1.5f * L.manual_level : 1.5f * L.normal_ratio * s.noise_floor;
    s.lvl = sq_level_compute(s, L);
}

/* The squelch's delay line WITHOUT the delay line.  buffer_[buffer_head_] = pre_filter_.capped_ * 0.9 is written once per sample and read 101
 * samples later (buffer_size_ 102, tail one ahead of head: src/squelch.cpp:66-69,218-219,453-456): 8 bytes of memory traffic per sample and channel
 * for a value that is a pure function of the input stream -- the capped moving average depends on the samples, on its cap, and the cap on the
 * noise floor, which depends on the capped average (calculate_noise_floor, every 16th sample).  The demod kernels have the input of 100 samples
 * ago in hand anyway (the AGC_EXTRA-delayed stream that feeds the audio path), so a second copy of that little machine, fed one sample per
 * sample and running 101 samples behind the squelch, holds the very floats the delay line would return: same operations, same order, same inputs.
 * `phase` = the squelch's sample_count_ AFTER its increment; the shadow's own count is phase - 101, and it sweeps its noise floor when that is a
 * multiple of 16.  The first 101 samples of a stream read the calloc'ed zeros of the reference's buffer: the caller returns 0 for them and starts
 * feeding the shadow with squelch sample 0. */
struct SqShadow {
    float nf, cap, capped;
};
AB_FSM_FN float sq_shadow_value(const SqShadow& h) { return h.capped * 0.9f; }
AB_FSM_FN void sq_shadow_step(SqShadow& h, const Lane& L, float x, unsigned phase) {
    if (((phase - 101u) & 15u) == 0u) { /* calculate_noise_floor + calculate_moving_avg_cap: sq_noise_floor() without the level cache */
        const float decay = 0.97f;
        const float fresh = (float)(1.0 - (double)0.97f);
        const float lo = h.nf < h.capped ? h.nf : h.capped;
        h.nf = h.nf * decay + lo * fresh + 1e-6f;
        h.cap = ab_lane(L.m_manual) ? 1.5f * L.manual_level : 1.5f * L.normal_ratio * h.nf;
    }
    const float decay = 0.99f;
    const float fresh = (float)(1.0 - (double)0.99f);
    const float xf = x * fresh;
    const float v = h.capped * decay + xf;
    const float vm = v < h.cap ? v : h.cap;
    h.capped = (h.capped >= h.cap && x >= h.cap) ? h.cap : vm;
}

/* Squelch::process_raw_sample (src/squelch.cpp:195-246) of a QUIET wavefront: every lane is CLOSED or OPEN and asks for nothing.
 * update_current_state() (:363-460) then only counts the CLOSED lanes' closed samples and moves the delay line; afterwards the
 * only requests a lane can raise are OPEN -> CLOSING (signal gone), OPEN -> LOW_SIGNAL_ABORT (:233-245) and CLOSED -> OPENING.
 * Straight-line code but for three seldom-taken exits. */
/* may_sweep: false where the caller knows that this sample cannot be a 16th one (the demod kernels: while sample_count_ + 1 is a multiple
 * of 4 at the start of a batch -- it always is, the count starts at -1 and batches are multiples of four samples long; the kernel checks it and
 * looks on every sample otherwise -- only the first sample of a group of four can be).
 * No branch but the noise-floor one: what the seldom events change is written with mask algebra that is a no-op when nothing happens
 * (a quiet wavefront has no lane in next-state OPENING / CLOSING / ABORT, so assigning the freshly computed -- usually empty -- request
 * masks is exact), and a CLOSED lane that is due to forget its recent opens ends the quiet spell instead of being handled here. */
AB_FSM_FN void sq_raw_quiet(SqRegs& s, const Lane& L, float x, float dly_new, bool may_sweep = true) {
    const lmask below = ab_ballot(s.closed_count < 1000u);
    s.closed_count += ab_lane(s.cC & below) ? 1u : 0u;
    if (L.track_delay_line) {
        s.tail = s.tail + 1 == AB_SQ_BUF ? 0 : s.tail + 1;
        s.head = s.head + 1 == AB_SQ_BUF ? 0 : s.head + 1;
    }
    s.dly = dly_new;
    s.sample_count++;
    if (may_sweep && AB_UNLIKELY((s.sample_count & 15u) == 0u)) sq_noise_floor(s, L);
    sq_avg(s.cap, s.pre_full, s.pre_capped, x);
    sq_delay_line_push(s, L);
    const lmask sig = sq_has_signal(s, L);
    const lmask low = s.cO & ab_ballot(!(x >= s.lvl));
    const int run = s.low_count + 1;
    const int idle = ab_lane(s.cO) ? 0 : s.low_count;
    s.low_count = ab_lane(low) ? run : idle;
    const lmask abort_now = low & ab_ballot(s.low_count >= 88); /* low_signal_abort_ */
    const lmask to_closing = s.cO & ~sig, to_opening = s.cC & sig;
    const lmask any_req = to_closing | to_opening | abort_now;
    s.nA = abort_now;
    s.nCg = to_closing & ~abort_now;
    s.nOg = to_opening;
    s.nO &= ~any_req;
    s.nC &= ~any_req;
    s.quiet = !ab_any((any_req | sq_saturated(s)) & s.active);
}

/* FOUR samples of a STABLE wavefront at once (kinds without a post-filter path whose delay-line cursors move once per batch: AM, NFM,
 * the NFM + CTCSS front).  Stable = no lane is entering a state (next_state_ == current_state_ everywhere), no timer of an OPENING /
 * CLOSING / LOW_SIGNAL_ABORT lane runs out within the four samples, no CLOSED lane is due to forget its recent opens.  That is the quiet
 * case (everything CLOSED or OPEN) plus the ~200-sample waits of the timed states -- at WAVE_RATE 8000 a third of a batch.  While the
 * wavefront stays stable its lane masks do not change at all, and what the four process_raw_sample() calls do is a straight run of
 * vector arithmetic: the two moving averages, the low-signal run length, the timers (+4), the CLOSED lanes' closed-sample count
 * (saturating at 1000, so four samples are one add and one min), and the noise floor on a 16th sample, which the caller's alignment puts
 * on the first of the four.  The only thing asked of the scalar unit per sample is to collect the lanes that END the stable spell: a
 * request (OPEN without signal, CLOSED with signal, the low-signal abort of OPENING / CLOSING / OPEN).  The samples are worked on a COPY of
 * the state; if no lane ended the spell the copy is the state after four reference calls, bit for bit, and is committed (returns true).
 * Otherwise nothing is committed (returns false) and the caller runs the four samples one at a time through sq_raw(), as before.  One
 * scalar branch per four samples instead of several per sample, and a basic block long enough to be scheduled.
 * Precondition: sq_stable4(s), (s.sample_count + 1) % 4 == 0, !L.may_post_filter, !L.track_delay_line. */
AB_FSM_FN bool sq_stable4(const SqRegs& s) {
    const lmask moving = (s.nC ^ s.cC) | (s.nOg ^ s.cOg) | (s.nCg ^ s.cCg) | (s.nA ^ s.cA) | (s.nO ^ s.cO);
    const lmask timed = s.cOg | s.cCg | s.cA;
    const lmask expiring = timed & ab_ballot(s.delay > 197 - 1 - 4); /* the timer is tested after its increment (:381, :400, :415): delay + 4 must stay below 197 */
    return !ab_any((moving | expiring | sq_saturated(s)) & s.active);
}
AB_FSM_FN bool sq_raw_stable4(SqRegs& s, const Lane& L, const float* x) {
    SqRegs t = s;
    const unsigned cc4 = t.closed_count + (ab_lane(t.cC) ? 4u : 0u);
    t.closed_count = cc4 < 1000u ? cc4 : 1000u;
    t.delay += ab_lane(t.cOg | t.cCg | t.cA) ? 4 : 0;
    t.sample_count += 1u;
    if (AB_UNLIKELY((t.sample_count & 15u) == 0u)) sq_noise_floor(t, L);
    t.sample_count += 3u;
    const lmask counting = t.cOg | t.cCg | t.cO; /* the lanes whose low-signal run length is kept (:233-245) */
    const lmask want = t.cO;                     /* OPEN lanes ask for CLOSING without the signal, CLOSED lanes for OPENING with it; the timed states ask for nothing */
    const lmask care = t.cO | t.cC;
    lmask bad = 0;
    for (int r = 0; r < 4; r++) {
        sq_avg(t.cap, t.pre_full, t.pre_capped, x[r]);
        const lmask sig = ab_ballot(t.pre_capped >= t.lvl);
        const lmask low = counting & ab_ballot(!(x[r] >= t.lvl));
        const int run = t.low_count + 1;
        const int idle = ab_lane(counting) ? 0 : t.low_count;
        t.low_count = ab_lane(low) ? run : idle;
        bad |= ((sig ^ want) & care) | (low & ab_ballot(t.low_count >= 88)); /* low_signal_abort_ */
    }
    bad |= t.cC & ~ab_ballot(t.closed_count < 1000u) & t.recent_nz; /* sq_saturated(): the count is monotonic, so the end of the group tells */
    if (AB_UNLIKELY(ab_any(bad & t.active))) return false;
    s = t;
    return true;
}

/* Squelch::process_raw_sample (src/squelch.cpp:195-246), general case.  Returns the lanes whose squelch just went CLOSED. */
AB_FSM_FN lmask sq_raw_full(SqRegs& s, const Lane& L, float x, float dly_new) {
    const lmask went_closed = sq_advance(s, L); /* evaluates the post-filter gate against buffer_[tail] BEFORE the tail moves */
    s.dly = dly_new;                            /* ... everything after it sees the entry under the advanced tail */
    s.sample_count++;
    if ((s.sample_count & 15u) == 0u) sq_noise_floor(s, L);
    sq_avg(s.cap, s.pre_full, s.pre_capped, x);
    sq_delay_line_push(s, L);
    const lmask sig = sq_has_signal(s, L);
    /* set_state() requests (:297-361), already clamped: OPEN -> CLOSING, CLOSED -> OPENING are legal as asked */
    const lmask to_closing = s.cO & ~sig;
    const lmask to_opening = s.cC & sig;
    /* low-signal abort (:233-245): LOW_SIGNAL_ABORT asked from OPENING is clamped to CLOSED, from CLOSING / OPEN it stands */
    const lmask counting = s.cOg | s.cCg | s.cO;
    const lmask low = counting & ab_ballot(!(x >= s.lvl));
    const int run = s.low_count + 1;
    const int idle = ab_lane(counting) ? 0 : s.low_count;
    s.low_count = ab_lane(low) ? run : idle;
    const lmask abort_now = low & ab_ballot(s.low_count >= 88); /* low_signal_abort_ */
    const lmask any_req = to_closing | to_opening | abort_now;
    if (ab_any(any_req)) {
        s.nC = (s.nC & ~any_req) | (abort_now & s.cOg);
        s.nA = (s.nA & ~any_req) | (abort_now & ~s.cOg);
        s.nCg = (s.nCg & ~any_req) | (to_closing & ~abort_now);
        s.nOg = (s.nOg & ~any_req) | (to_opening & ~abort_now);
        s.nO &= ~any_req;
    }
    s.quiet = sq_is_quiet(s);
    return went_closed;
}

AB_FSM_FN lmask sq_raw(SqRegs& s, const Lane& L, float x, float dly_new) {
    if (AB_LIKELY(s.quiet)) {
        sq_raw_quiet(s, L, x, dly_new);
        return 0;
    }
    return sq_raw_full(s, L, x, dly_new);
}

AB_FSM_FN lmask sq_should_filter(const SqRegs& s) { return (sq_has_pre(s) | ~s.cC) & ~s.cA & s.active; } /* :136-138 */
AB_FSM_FN lmask sq_should_audio(const SqRegs& s) { return s.cO | s.cCg; }                                /* :140-142 */
AB_FSM_FN lmask sq_first_open(const SqRegs& s) { return ~s.cO & s.nO; }                                  /* :144-146 */
AB_FSM_FN lmask sq_last_open(const SqRegs& s) { return (s.cCg & s.nC) | (~s.cA & s.nA); }                /* :148-150 */

/* Squelch::process_filtered_sample (src/squelch.cpp:248-276) for the lanes in `filt` (= should_filter_sample() lanes that
 * have a lowpass filter; the caller ran the filter for them and hands over the filtered magnitude) */
AB_FSM_FN void sq_filtered(SqRegs& s, const Lane& L, lmask filt, float x) {
    filt &= L.m_lowpass;
    if (!ab_any(filt)) return;
    const float delayed = sq_delayed(s, L);
    /* OPENING: nothing until the delay line holds post-opening samples, then the averages start from its oldest entry */
    const lmask run = filt & ~(s.cOg & ab_ballot(s.delay < AB_SQ_BUF));
    const lmask seed = run & s.cOg & ab_ballot(s.delay == AB_SQ_BUF);
    float full = ab_lane(seed) ? delayed : s.post_full, capped = ab_lane(seed) ? delayed : s.post_capped;
    s.using_post |= run;
    sq_avg(s.cap, full, capped, x);
    s.post_full = ab_lane(run) ? full : s.post_full;
    s.post_capped = ab_lane(run) ? capped : s.post_capped;
    const lmask close = run & ab_ballot(capped < delayed);
    if (ab_rare(close)) { /* set_state(CLOSED): from OPEN that is clamped to CLOSING, from anywhere else it stands */
        s.nC = (s.nC & ~close) | (close & ~s.cO);
        s.nCg = (s.nCg & ~close) | (close & s.cO);
        s.nOg &= ~close;
        s.nA &= ~close;
        s.nO &= ~close;
        s.quiet = false; /* re-derived by the next sq_raw() */
    }
}

/* ChanState <-> registers; `valid` = this lane owns a channel */
AB_FSM_FN void sq_load(SqRegs& s, const Lane& L, const ChanState* sp, bool valid) {
    s.active = ab_ballot(valid);
    s.noise_floor = sp->noise_floor; s.cap = sp->cap; s.pre_full = sp->pre_full; s.pre_capped = sp->pre_capped;
    s.post_full = sp->post_full; s.post_capped = sp->post_capped;
    s.using_post = ab_ballot(valid && sp->using_post != 0);
    sq_set_next(s, valid ? sp->next : -1);
    sq_set_cur(s, valid ? sp->cur : -1);
    s.delay = sp->delay; s.low_count = sp->low_count;
    s.head = (int)ab_uniform((unsigned)sp->head); s.tail = (int)ab_uniform((unsigned)sp->tail); /* like sample_count_: the same on every channel */
    s.sample_count = ab_uniform(sp->sample_count); s.open_count = sp->open_count;
    s.flappy_count = sp->flappy_count; s.recent_open = sp->recent_open; s.closed_count = sp->closed_count;
    s.recent_nz = ab_ballot(valid && sp->recent_open != 0u);
    s.lvl = sq_level_compute(s, L);
    s.dly = 0.0f;
    s.quiet = sq_is_quiet(s);
}

/* `samples` = how many process_raw_sample calls ran since sq_load (head/tail advance once per call, :453-456) */
AB_FSM_FN void sq_store(const SqRegs& s, const Lane& L, ChanState* sp, int samples) {
    sp->noise_floor = s.noise_floor; sp->cap = s.cap; sp->pre_full = s.pre_full; sp->pre_capped = s.pre_capped;
    sp->post_full = s.post_full; sp->post_capped = s.post_capped;
    sp->using_post = ab_lane(s.using_post) ? 1 : 0; sp->next = sq_next(s); sp->cur = sq_cur(s); sp->delay = s.delay; sp->low_count = s.low_count;
    int head = s.head, tail = s.tail;
    if (!L.track_delay_line) {
        head = (head + samples) % AB_SQ_BUF;
        tail = (tail + samples) % AB_SQ_BUF;
    }
    sp->head = head; sp->tail = tail; sp->sample_count = s.sample_count; sp->open_count = s.open_count;
    sp->flappy_count = s.flappy_count; sp->recent_open = s.recent_open; sp->closed_count = s.closed_count;
}

}  // namespace airband

#endif
