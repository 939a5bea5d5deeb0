/* csrc/demod.hip -- stage 2 of the hot path on gfx950: the per-channel sequential loop of demodulate()
 * (reference: src/rtl_airband.cpp:494-648) with everything it calls: Squelch (src/squelch.cpp), CTCSS Goertzel
 * banks (src/ctcss.cpp), NotchFilter / LowpassFilter (src/filters.cpp), sincosf_lut (src/util.cpp:113-127) and
 * the FM helpers (src/rtl_airband.cpp:141-176).
 *
 * Mapping: one lane per (dongle, channel); 64 consecutive internal channel slots per wavefront; the batch's
 * WAVE_BATCH samples are walked sequentially by every lane (IIR / EMA / FSM state makes time strictly serial),
 * all per-channel state lives in registers for the duration of the batch and in ChanState between batches.
 * Stage-1 results arrive time-major ([hop][slot]) so that the 64 lanes of a wave read one contiguous 256-byte
 * row per step.
 *
 * This file MUST be compiled with -ffp-contract=off: squelch decisions have to be bit-identical to the
 * reference's scalar float code given the same stage-1 input, so no FMA contraction, IEEE divide and sqrt
 * (-fhip-fp32-correctly-rounded-divide-sqrt) and the reference's operation order everywhere.
 */
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace airband {

struct SqRegs { /* Squelch members that change per sample (src/squelch.h:117-158) */
    float noise_floor, cap, pre_full, pre_capped, post_full, post_capped, level_cache;
    int using_post, next, cur, delay, low_count, head, tail;
    unsigned sample_count, open_count, flappy_count, recent_open, closed_count;
};

struct CtRegs { /* both CTCSS detectors of one channel (src/ctcss.h:84-95) */
    int enough[2], count[2], has_tone[2];
    unsigned found[2], not_found[2];
};

struct Lane {
    unsigned flags;
    float manual_level, normal_ratio, flappy_ratio;
    /* tables */
    float* sqbuf;       /* this lane's column of the 102-deep pre-filter delay line, stride S */
    const float* ctc;   /* this lane's CTCSS coefficient column */
    float* ctq;         /* this lane's CTCSS q1/q2 column */
    int ct_stride;
    int ct_n[2], ct_win[2];
    long S;
};

__device__ __forceinline__ bool sq_flapping(const SqRegs& s) { return s.recent_open >= 3u; } /* flap_opens_threshold_ */

/* Squelch::squelch_level() (src/squelch.cpp:164-177): cached, 0 means recompute */
__device__ __forceinline__ float sq_level(SqRegs& s, const Lane& L) {
    if (L.flags & AB_F_MANUAL) return L.manual_level;
    if (s.level_cache == 0.0f) {
        if (sq_flapping(s) && L.flappy_ratio < L.normal_ratio)
            s.level_cache = L.flappy_ratio * s.noise_floor;
        else
            s.level_cache = L.normal_ratio * s.noise_floor;
    }
    return s.level_cache;
}

__device__ __forceinline__ bool sq_has_pre(SqRegs& s, const Lane& L) { return s.pre_capped >= sq_level(s, L); }

__device__ __forceinline__ bool sq_has_signal(SqRegs& s, const Lane& L) { /* src/squelch.cpp:462-475 */
    if (s.using_post) return sq_has_pre(s, L) && (s.post_capped >= L.sqbuf[(long)s.tail * L.S]);
    return sq_has_pre(s, L);
}

/* Squelch::set_state (src/squelch.cpp:297-361): clamp transitions that are not allowed from the current state */
__device__ __forceinline__ void sq_request(SqRegs& s, int want) {
    if (s.cur == AB_ST_CLOSED) {
        if (want == AB_ST_CLOSING || want == AB_ST_ABORT) want = AB_ST_CLOSED;
        else if (want == AB_ST_OPEN) want = AB_ST_OPENING;
    } else if (s.cur == AB_ST_OPENING) {
        if (want == AB_ST_ABORT) want = AB_ST_CLOSED;
    } else if (s.cur == AB_ST_ABORT) {
        if (want != AB_ST_ABORT && want != AB_ST_CLOSED) want = AB_ST_CLOSED;
    } else if (s.cur == AB_ST_OPEN) {
        if (want == AB_ST_CLOSED) want = AB_ST_CLOSING;
        else if (want == AB_ST_OPENING) want = AB_ST_OPEN;
    }
    s.next = want;
}

/* CTCSS::reset for both detectors (src/ctcss.cpp:165-172): Goertzel state cleared, magnitude irrelevant */
__device__ void ct_reset(CtRegs& c, const Lane& L) {
    if (!(L.flags & AB_F_CTCSS)) return;
    for (int k = 0; k < 2; k++) {
        for (int t = 0; t < L.ct_n[k]; t++) {
            float* q = L.ctq + (long)((k * AB_MAX_TONES + t) * 2) * L.ct_stride;
            q[0] = 0.0f;
            q[L.ct_stride] = 0.0f;
        }
        c.enough[k] = 0;
        c.count[k] = 0;
        c.has_tone[k] = 0;
    }
}

/* Squelch::update_current_state (src/squelch.cpp:363-460) */
__device__ __forceinline__ void sq_advance(SqRegs& s, CtRegs& c, const Lane& L) {
    if (s.next == AB_ST_OPENING) {
        if (s.cur != AB_ST_OPENING) {
            s.delay = 0;
            s.low_count = 0;
            s.using_post = 0;
            s.cur = AB_ST_OPENING;
        } else if (++s.delay >= 197) {                 /* open_delay_ */
            if (s.closed_count < 1000u) {              /* recent_sample_size_ */
                s.recent_open++;
                if (sq_flapping(s)) s.flappy_count++;
                s.level_cache = 0.0f;
            }
            s.next = sq_has_signal(s, L) ? AB_ST_OPEN : AB_ST_CLOSED;
        }
    } else if (s.next == AB_ST_CLOSING) {
        if (s.cur != AB_ST_CLOSING) {
            s.delay = 0;
            s.cur = AB_ST_CLOSING;
        } else if (++s.delay >= 197) {                 /* close_delay_ */
            if (!sq_has_signal(s, L)) {
                s.next = AB_ST_CLOSED;
            } else {
                s.cur = AB_ST_OPEN;
                s.next = AB_ST_OPEN;
            }
        }
    } else if (s.next == AB_ST_ABORT) {
        if (s.cur != AB_ST_ABORT) {
            if (s.cur != AB_ST_CLOSING) s.delay = 0;
            s.cur = AB_ST_ABORT;
        } else if (++s.delay >= 197) {
            s.next = AB_ST_CLOSED;
        }
    } else if (s.next == AB_ST_OPEN) {
        if (s.cur != AB_ST_OPEN) {
            s.open_count++;
            s.cur = AB_ST_OPEN;
        }
    } else { /* CLOSED */
        if (s.cur != AB_ST_CLOSED) {
            s.using_post = 0;
            s.closed_count = 0;
            s.cur = AB_ST_CLOSED;
            ct_reset(c, L);
        } else if (s.closed_count < 1000u) {
            s.closed_count++;
        } else if (s.closed_count == 1000u) {
            s.recent_open = 0;
            s.level_cache = 0.0f;
        }
    }
    s.tail = s.tail + 1 == AB_SQ_BUF ? 0 : s.tail + 1;
    s.head = s.head + 1 == AB_SQ_BUF ? 0 : s.head + 1;
}

/* Squelch::update_moving_avg (src/squelch.cpp:501-514) */
__device__ __forceinline__ void sq_avg(float cap, float& full, float& capped, float x) {
    const float decay = 0.99f;
    const float fresh = (float)(1.0 - (double)0.99f);
    full = full * decay + x * fresh;
    if (capped >= cap && x >= cap) {
        capped = cap;
    } else {
        const float v = capped * decay + x * fresh;
        capped = cap < v ? cap : v;
    }
}

/* Squelch::process_raw_sample (src/squelch.cpp:195-246) */
__device__ __forceinline__ void sq_raw(SqRegs& s, CtRegs& c, const Lane& L, float x) {
    sq_advance(s, c, L);
    s.sample_count++;
    if ((s.sample_count & 15u) == 0u) { /* calculate_noise_floor, :477-490 */
        const float decay = 0.97f;
        const float fresh = (float)(1.0 - (double)0.97f);
        const float lo = s.pre_capped < s.noise_floor ? s.pre_capped : s.noise_floor;
        s.noise_floor = s.noise_floor * decay + lo * fresh + 1e-6f;
        s.cap = (L.flags & AB_F_MANUAL) ? 1.5f * L.manual_level : 1.5f * L.normal_ratio * s.noise_floor;
        s.level_cache = 0.0f;
    }
    sq_avg(s.cap, s.pre_full, s.pre_capped, x);
    if (L.flags & AB_F_LOWPASS) L.sqbuf[(long)s.head * L.S] = s.pre_capped * 0.9f; /* only ever read on the post-filter path */
    if (s.cur == AB_ST_OPEN && !sq_has_signal(s, L)) sq_request(s, AB_ST_CLOSING);
    if (s.cur == AB_ST_CLOSED && sq_has_signal(s, L)) sq_request(s, AB_ST_OPENING);
    if (s.cur != AB_ST_CLOSED && s.cur != AB_ST_ABORT) {
        if (x >= sq_level(s, L)) {
            s.low_count = 0;
        } else if (++s.low_count >= 88) { /* low_signal_abort_ */
            sq_request(s, AB_ST_ABORT);
        }
    }
}

__device__ __forceinline__ bool sq_should_filter(SqRegs& s, const Lane& L) { return (sq_has_pre(s, L) || s.cur != AB_ST_CLOSED) && s.cur != AB_ST_ABORT; }
__device__ __forceinline__ bool sq_should_audio(const SqRegs& s) { return s.cur == AB_ST_OPEN || s.cur == AB_ST_CLOSING; }
__device__ __forceinline__ bool sq_first_open(const SqRegs& s) { return s.cur != AB_ST_OPEN && s.next == AB_ST_OPEN; }
__device__ __forceinline__ bool sq_last_open(const SqRegs& s) {
    return (s.cur == AB_ST_CLOSING && s.next == AB_ST_CLOSED) || (s.cur != AB_ST_ABORT && s.next == AB_ST_ABORT);
}

/* Squelch::process_filtered_sample (src/squelch.cpp:248-276) */
__device__ __forceinline__ void sq_filtered(SqRegs& s, const Lane& L, float x) {
    if (!sq_should_filter(s, L)) return;
    const float delayed = L.sqbuf[(long)s.tail * L.S];
    if (s.cur == AB_ST_OPENING) {
        if (s.delay < AB_SQ_BUF) return;
        if (s.delay == AB_SQ_BUF) s.post_full = s.post_capped = delayed;
    }
    s.using_post = 1;
    sq_avg(s.cap, s.post_full, s.post_capped, x);
    if (s.post_capped < delayed) sq_request(s, AB_ST_CLOSED);
}

/* one CTCSS detector, one sample (src/ctcss.cpp:44-54,124-163) */
__device__ void ct_sample(CtRegs& c, const Lane& L, int k, float x) {
    const int n = L.ct_n[k];
    const bool last = (c.count[k] + 1 >= L.ct_win[k]);
    float total = 0.0f, best = 0.0f, target = 0.0f;
    for (int t = 0; t < n; t++) {
        const long off = (long)(k * AB_MAX_TONES + t);
        const float co = L.ctc[off * L.ct_stride];
        float* q = L.ctq + off * 2 * L.ct_stride;
        const float q1 = q[0], q2 = q[L.ct_stride];
        const float q0 = co * q1 - q2 + x;
        if (!last) {
            q[L.ct_stride] = q1;
            q[0] = q0;
        } else { /* window complete: power of this tone, then clear for the next window */
            const float m = q0 * q0 + q1 * q1 - q0 * q1 * co;
            total += m;
            if (t == 0) {
                target = m;
                best = m;
            } else if (m > best) {
                best = m;
            }
            q[0] = 0.0f;
            q[L.ct_stride] = 0.0f;
        }
    }
    c.count[k]++;
    if (!last) return;
    c.enough[k] = 1;
    const float avg = total / (float)n;
    if (target == best && target > avg) {
        c.has_tone[k] = 1;
        c.found[k]++;
    } else {
        c.has_tone[k] = 0;
        c.not_found[k]++;
    }
    c.count[k] = 0;
}

/* Squelch::process_audio_sample (src/squelch.cpp:278-295) */
__device__ __forceinline__ void sq_audio(const SqRegs& s, CtRegs& c, const Lane& L, float x) {
    if (!(L.flags & AB_F_CTCSS)) return;
    if (s.cur != AB_ST_CLOSED) {
        ct_sample(c, L, 1, x);
        if (!c.enough[1]) ct_sample(c, L, 0, x);
    }
}

__device__ __forceinline__ bool sq_tone(const CtRegs& c, const Lane& L) {
    if (!(L.flags & AB_F_CTCSS)) return true;
    return c.enough[1] ? (c.has_tone[1] != 0) : (c.has_tone[0] != 0);
}

/* fast_atan2 / polar_disc_fast / fm_quadri_demod (src/rtl_airband.cpp:141-176) */
__device__ __forceinline__ float fast_atan2_dev(float y, float x) {
    const float pi4 = (float)0.78539816339744830962, pi34 = (float)(3 * 0.78539816339744830962);
    if (x == 0.0f && y == 0.0f) return 0.0f;
    const float ay = y < 0.0f ? -y : y;
    float a;
    if (x >= 0.0f)
        a = pi4 - pi4 * (x - ay) / (x + ay);
    else
        a = pi34 - pi4 * (x + ay) / (ay - x);
    return y < 0.0f ? -a : a;
}

__global__ __launch_bounds__(64) void demod_kernel(DemodArgs a) {
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= a.n_slots) return;
    const ChanConst cc = a.cc[slot];
    if (!(cc.flags & AB_F_VALID)) return;
    const long S = a.stride;
    const int R = a.ring_rows, B = a.wave_batch;

    Lane L;
    L.flags = cc.flags;
    L.manual_level = cc.sq_manual_level;
    L.normal_ratio = cc.sq_normal_ratio;
    L.flappy_ratio = cc.sq_flappy_ratio;
    L.sqbuf = a.sqbuf + slot;
    L.S = S;
    L.ct_stride = a.ct_stride;
    L.ctc = a.ct_coeff + (cc.ct_slot >= 0 ? cc.ct_slot : 0);
    L.ctq = a.ct_q + (cc.ct_slot >= 0 ? cc.ct_slot : 0);
    L.ct_n[0] = cc.ct_ntones[0];
    L.ct_n[1] = cc.ct_ntones[1];
    L.ct_win[0] = cc.ct_window[0];
    L.ct_win[1] = cc.ct_window[1];

    ChanState* sp = a.cs + slot;
    SqRegs s;
    CtRegs c;
    s.noise_floor = sp->noise_floor; s.cap = sp->cap; s.pre_full = sp->pre_full; s.pre_capped = sp->pre_capped;
    s.post_full = sp->post_full; s.post_capped = sp->post_capped; s.level_cache = sp->level_cache;
    s.using_post = sp->using_post; s.next = sp->next; s.cur = sp->cur; s.delay = sp->delay; s.low_count = sp->low_count;
    s.head = sp->head; s.tail = sp->tail; s.sample_count = sp->sample_count; s.open_count = sp->open_count;
    s.flappy_count = sp->flappy_count; s.recent_open = sp->recent_open; s.closed_count = sp->closed_count;
    for (int k = 0; k < 2; k++) {
        c.enough[k] = sp->ct_enough[k]; c.count[k] = sp->ct_count[k]; c.has_tone[k] = sp->ct_has_tone[k];
        c.found[k] = sp->ct_found[k]; c.not_found[k] = sp->ct_not_found[k];
    }
    float agc = sp->agcavgfast, pr = sp->pr, pj = sp->pj, prev_out = sp->prev_waveout;
    unsigned dm_phi = sp->dm_phi;
    float nx0 = sp->nx[0], nx1 = sp->nx[1], nx2 = sp->nx[2], ny0 = sp->ny[0], ny1 = sp->ny[1], ny2 = sp->ny[2];
    float lxr0 = sp->lxr[0], lxr1 = sp->lxr[1], lxr2 = sp->lxr[2], lxi0 = sp->lxi[0], lxi1 = sp->lxi[1], lxi2 = sp->lxi[2];
    float lyr0 = sp->lyr[0], lyr1 = sp->lyr[1], lyr2 = sp->lyr[2], lyi0 = sp->lyi[0], lyi1 = sp->lyi[1], lyi2 = sp->lyi[2];

    const bool nfm = cc.flags & AB_F_NFM, raw_iq = cc.flags & AB_F_RAW_IQ, lowpass = cc.flags & AB_F_LOWPASS;
    const bool notch = cc.flags & AB_F_NOTCH, iq_outputs = cc.flags & AB_F_IQ_OUT;
    const float one_minus_alpha = 1.0f - cc.alpha;
    int axc = ' ';

    float* mag = a.mag + slot;
    const float2* iqin = a.iq + slot;
    float* wave = a.wave + slot;
    float2* iqout = a.iq_out + slot;
    uint8_t* trace = a.trace ? a.trace + slot : nullptr;

    constexpr int CH = 8; /* samples fetched ahead per round: hides HBM latency behind the serial recurrences */
    for (int j0 = 0; j0 < B; j0 += CH) {
        float xs[CH], xd[CH];
        float2 qd[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
            int rc = a.row0 + AB_AGC_EXTRA + j0 + u;   /* current hop (logical row j+AGC_EXTRA) */
            if (rc >= R) rc -= R;
            int rd = a.row0 + j0 + u;                  /* hop AGC_EXTRA earlier */
            if (rd >= R) rd -= R;
            xs[u] = mag[(long)rc * S];
            xd[u] = nfm ? 0.0f : mag[(long)rd * S];
            qd[u] = raw_iq ? iqin[(long)rd * S] : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
            const int j = j0 + u;
            int rc = a.row0 + AB_AGC_EXTRA + j;
            if (rc >= R) rc -= R;
            float re = qd[u].x, im = qd[u].y;
            float cur_mag = xs[u];

            sq_raw(s, c, L, cur_mag);

            if (raw_iq && sq_should_filter(s, L)) { /* src/rtl_airband.cpp:510-530 */
                /* sincosf_lut (src/util.cpp:113-127) */
                const unsigned idx = dm_phi >> 16;
                const float fract = (float)(dm_phi & 0xffffu) / 65536.0f;
                const float s0 = a.sin_lut[idx], s1 = a.sin_lut[idx + 1], c0 = a.cos_lut[idx], c1 = a.cos_lut[idx + 1];
                const float swf = s0 + (s1 - s0) * fract;
                const float cwf = c0 + (c1 - c0) * fract;
                const float nswf = -swf;
                float tr = re * cwf - im * nswf;  /* multiply(real, imag, cwf, -swf) */
                float ti = im * cwf + re * nswf;
                dm_phi = (dm_phi + cc.dm_dphi) & 0xffffffu;
                if (lowpass) { /* LowpassFilter::apply (src/filters.cpp:146-163) */
                    lxr0 = lxr1; lxi0 = lxi1;
                    lxr1 = lxr2; lxi1 = lxi2;
                    lxr2 = tr / cc.lp_gain; lxi2 = ti / cc.lp_gain;
                    lyr0 = lyr1; lyi0 = lyi1;
                    lyr1 = lyr2; lyi1 = lyi2;
                    lyr2 = (lxr0 + lxr2) + (2.0f * lxr1) + (cc.lp_yc0 * lyr0) + (cc.lp_yc1 * lyr1);
                    lyi2 = (lxi0 + lxi2) + (2.0f * lxi1) + (cc.lp_yc0 * lyi0) + (cc.lp_yc1 * lyi1);
                    tr = lyr2;
                    ti = lyi2;
                }
                re = tr;
                im = ti;
                cur_mag = sqrtf(re * re + im * im); /* double sqrt rounded to float == correctly rounded sqrtf */
                mag[(long)rc * S] = cur_mag;
                if (lowpass) sq_filtered(s, L, cur_mag);
            }

            if (!nfm) { /* src/rtl_airband.cpp:532-547 */
                if (sq_first_open(s)) {
                    const float lvl = sq_level(s, L);
                    for (int k = j; k < j + AB_AGC_EXTRA; k++) { /* logical rows j .. j+AGC_EXTRA-1 = the AGC_EXTRA hops before the current one */
                        int rk = a.row0 + k;
                        if (rk >= R) rk -= R;
                        const float w = mag[(long)rk * S];
                        if (w >= lvl) agc = agc * 0.9f + w * 0.1f;
                    }
                } else if (sq_last_open(s)) {
                    int rp = a.row0 + j;           /* logical row (j+AGC_EXTRA) - AGC_EXTRA */
                    if (rp >= R) rp -= R;
                    float prev = wave[(long)rp * S];
                    for (int k = j + 1; k < j + AB_AGC_EXTRA; k++) {
                        int rk = a.row0 + k;
                        if (rk >= R) rk -= R;
                        prev = prev * 0.94f;
                        wave[(long)rk * S] = prev;
                    }
                }
            }

            float out = 0.0f;
            const bool audio = sq_should_audio(s);
            if (audio) {
                if (!nfm) { /* AM: src/rtl_airband.cpp:553-563 */
                    if (cur_mag > sq_level(s, L)) agc = agc * 0.995f + cur_mag * 0.005f;
                    out = (xd[u] - agc) / (agc * 1.5f);
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
                sq_audio(s, c, L, out);
            }

            const bool open = audio && sq_tone(c, L); /* Squelch::is_open (src/squelch.cpp:118-134) */
            float2 qo = make_float2(0.0f, 0.0f);
            if (open) { /* src/rtl_airband.cpp:590-611 */
                if (notch) { /* NotchFilter::apply (src/filters.cpp:50-64) */
                    nx0 = nx1; nx1 = nx2; nx2 = out;
                    ny0 = ny1; ny1 = ny2;
                    ny2 = cc.notch_d0 * nx2 - cc.notch_d1 * nx1 + cc.notch_d0 * nx0 + cc.notch_d1 * ny1 - cc.notch_d2 * ny0;
                    out = ny2;
                }
                out *= cc.ampfactor;
                if (out != out) out = 0.0f;
                else if (out > 1.0f) out = 1.0f;
                else if (out < -1.0f) out = -1.0f;
                axc = '*';
                qo = make_float2(re, im);
            } else {
                out = 0.0f;
            }
            /* The AM fade-out above may already have written logical rows up to j+AGC_EXTRA-1; this is row j+AGC_EXTRA */
            wave[(long)rc * S] = out;
            if (iq_outputs) iqout[(long)j * S] = qo;
            if (trace) trace[(long)j * S] = (uint8_t)((s.cur & 7) | (open ? 8 : 0) | (audio ? 16 : 0) | (((cc.flags & AB_F_CTCSS) && sq_tone(c, L)) ? 32 : 0));
        }
    }

    if (axc != ' ') sp->active_counter++;
    sp->axc = axc;
    sp->agcavgfast = agc; sp->pr = pr; sp->pj = pj; sp->prev_waveout = prev_out; sp->dm_phi = dm_phi;
    sp->noise_floor = s.noise_floor; sp->cap = s.cap; sp->pre_full = s.pre_full; sp->pre_capped = s.pre_capped;
    sp->post_full = s.post_full; sp->post_capped = s.post_capped; sp->level_cache = s.level_cache;
    sp->using_post = s.using_post; sp->next = s.next; sp->cur = s.cur; sp->delay = s.delay; sp->low_count = s.low_count;
    sp->head = s.head; sp->tail = s.tail; sp->sample_count = s.sample_count; sp->open_count = s.open_count;
    sp->flappy_count = s.flappy_count; sp->recent_open = s.recent_open; sp->closed_count = s.closed_count;
    sp->nx[0] = nx0; sp->nx[1] = nx1; sp->nx[2] = nx2; sp->ny[0] = ny0; sp->ny[1] = ny1; sp->ny[2] = ny2;
    sp->lxr[0] = lxr0; sp->lxr[1] = lxr1; sp->lxr[2] = lxr2; sp->lxi[0] = lxi0; sp->lxi[1] = lxi1; sp->lxi[2] = lxi2;
    sp->lyr[0] = lyr0; sp->lyr[1] = lyr1; sp->lyr[2] = lyr2; sp->lyi[0] = lyi0; sp->lyi[1] = lyi1; sp->lyi[2] = lyi2;
    for (int k = 0; k < 2; k++) {
        sp->ct_enough[k] = c.enough[k]; sp->ct_count[k] = c.count[k]; sp->ct_has_tone[k] = c.has_tone[k];
        sp->ct_found[k] = c.found[k]; sp->ct_not_found[k] = c.not_found[k];
    }
}

void launch_demod(const DemodArgs& a, hipStream_t stream) {
    const int blocks = (a.n_slots + 63) / 64;
    hipLaunchKernelGGL(demod_kernel, dim3(blocks), dim3(64), 0, stream, a);
}

/* ---- emit: time-major device results -> the channel-major layout the output thread consumes -------------
 * (reference: src/output.cpp:460,521,535 read channel->waveout[0..WAVE_BATCH) / iq_out; :920 tail copy is implicit
 * in the ring rotation).  64 slots x 64 samples per block, transposed through LDS so both sides are coalesced. */
__global__ __launch_bounds__(256) void emit_kernel(EmitArgs a) {
    __shared__ float tile[64][65];
    const int slot0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int R = a.ring_rows;
    for (int r = ty; r < 64; r += 4) {
        const int t = t0 + r;
        float v = 0.0f;
        if (t < a.wave_batch && slot0 + tx < a.n_slots) {
            int pr = a.row0 + t;
            if (pr >= R) pr -= R;
            v = a.wave[(long)pr * a.stride + slot0 + tx];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int slot = slot0 + r, t = t0 + tx;
        if (slot < a.n_slots && t < a.wave_batch) {
            const int ext = a.slot_to_ext[slot];
            if (ext >= 0) a.out_wave[(long)ext * a.wave_batch + t] = tile[tx][r];
        }
    }
    if (a.out_iq) {
        for (int comp = 0; comp < 2; comp++) {
            __syncthreads();
            for (int r = ty; r < 64; r += 4) {
                const int t = t0 + r;
                float v = 0.0f;
                if (t < a.wave_batch && slot0 + tx < a.n_slots) {
                    const float2 q = a.iq_out[(long)t * a.stride + slot0 + tx];
                    v = comp ? q.y : q.x;
                }
                tile[r][tx] = v;
            }
            __syncthreads();
            for (int r = ty; r < 64; r += 4) {
                const int slot = slot0 + r, t = t0 + tx;
                if (slot < a.n_slots && t < a.wave_batch) {
                    const int ext = a.slot_to_ext[slot];
                    if (ext >= 0) a.out_iq[((long)ext * a.wave_batch + t) * 2 + comp] = tile[tx][r];
                }
            }
        }
    }
    if (blockIdx.y == 0 && threadIdx.x < 64) {
        const int slot = slot0 + threadIdx.x;
        if (slot < a.n_slots) {
            const int ext = a.slot_to_ext[slot];
            if (ext >= 0) a.out_axc[ext] = (uint8_t)a.cs[slot].axc;
        }
    }
}

void launch_emit(const EmitArgs& a, hipStream_t stream) {
    dim3 grid((a.n_slots + 63) / 64, (a.wave_batch + 63) / 64);
    hipLaunchKernelGGL(emit_kernel, grid, dim3(256), 0, stream, a);
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
    else if (s.level_cache != 0.0f) lvl = s.level_cache;
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
    out[ext] = o;
}

void launch_stats(const ChanConst* cc, const ChanState* cs, const int* slot_to_ext, int n_slots, airband_hip_channel_stats* out, hipStream_t stream) {
    hipLaunchKernelGGL(stats_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, stream, cc, cs, slot_to_ext, n_slots, out);
}

}  // namespace airband
