// tests/host_fsm_harness.cpp -- TEST HARNESS ONLY (built by tests/test_host_fsm.py with g++, never part of the library).
// Compiles the demod kernels' squelch state machine (csrc/squelch_fsm.h) as plain C++, one "lane", and drives it with the
// constants and initial state the library itself derives (csrc/params.cpp), so that the logic can be compared sample by
// sample with the oracle / the reference's Squelch without a GPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../rtlsdr-airband_amd/csrc/params.h"
#include "../rtlsdr-airband_amd/csrc/squelch_fsm.h"

using namespace airband;

namespace {
unsigned char flags_of(const SqRegs& s) { /* same bit layout as the oracle's sq_flags() (no CTCSS: is_open == should_audio) */
    const bool audio = ab_lane(sq_should_audio(s));
    return (unsigned char)((audio ? 1 : 0) | (audio ? 2 : 0) | (ab_lane(sq_should_filter(s)) ? 4 : 0) | (ab_lane(sq_first_open(s)) ? 8 : 0) |
                           (ab_lane(sq_last_open(s)) ? 16 : 0));
}
}  // namespace

extern "C" {

// mode: 5 = as 2, but the delay-line entry comes from the shadow moving average (squelch_fsm.h, SqShadow) instead of the stored line;
// 3 = as 0, but aligned groups of four samples of a stable "wavefront" go through sq_raw_stable4() (what the AM kind and the CTCSS front do;
// counts[4] = groups committed that way);
// 0 = no lowpass (head/tail moved once at the end, as the AM/NFM kinds do), 1 = lowpass with the delay line read from
// memory (generic kind), 2 = lowpass with the delay-line entry prefetched before the call (NFM+lowpass kind).
// `chunk` splits the run into pieces with a store/load of ChanState in between (what happens between batches).
// out_state: cur, next, delay, low_count, head, tail, using_post, sample_count; counts: open, flappy, recent_open, closed_count
int hostfsm_run(float snr_db, int manual_dbfs, int mode, int chunk, const float* raw, const float* filtered, int n, unsigned char* flags, float* noise, float* level,
                int64_t* out_state, uint64_t* counts) {
    airband_hip_channel_cfg ch;
    std::memset(&ch, 0, sizeof(ch));
    ch.frequency = 120100000;
    ch.modulation = AIRBAND_MOD_AM;
    ch.squelch_threshold_dbfs = manual_dbfs;
    ch.squelch_snr_threshold_db = snr_db;
    ch.ampfactor = 1.0f;
    ch.tau_us = -1;
    airband_hip_device_cfg dv;
    std::memset(&dv, 0, sizeof(dv));
    dv.sample_rate = 2560000;
    dv.centerfreq = 120000000;
    dv.sfmt = AIRBAND_SFMT_U8;
    dv.tau_us = -1;
    dv.channel_count = 1;
    dv.channels = &ch;
    airband_hip_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = AIRBAND_HIP_ABI_VERSION;
    cfg.fft_size_log = 9;
    cfg.wave_rate = 8000;
    cfg.device_count = 1;
    cfg.devices = &dv;
    Plan plan;
    if (build_plan(&cfg, plan) != 0) return -1;
    const ChanConst cc = plan.cc[0];
    ChanState st = plan.cs0[0];

    std::vector<float> sqbuf(AB_SQ_BUF, 0.0f);
    Lane L;
    const bool lowpass = mode == 1 || mode == 2 || mode == 5;
    L.m_lowpass = ab_ballot(lowpass);
    L.m_manual = ab_ballot((cc.flags & AB_F_MANUAL) != 0);
    L.manual_level = cc.sq_manual_level;
    L.normal_ratio = cc.sq_normal_ratio;
    L.flappy_ratio = cc.sq_flappy_ratio;
    L.m_flappy_lower = ab_ballot(cc.sq_flappy_ratio < cc.sq_normal_ratio);
    L.sqbuf = sqbuf.data();
    L.S = 1;
    L.prefetched_delay = mode == 2 || mode == 5;
    L.track_delay_line = lowpass;
    L.may_post_filter = lowpass;
    L.all_lowpass = false;
    L.shadow_delay = mode == 5;

    uint64_t groups_committed = 0, *groups = &groups_committed;
    if (chunk <= 0) chunk = n;
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = n - i0 < chunk ? n - i0 : chunk;
        SqRegs s;
        sq_load(s, L, &st, true);
        if (mode == 2) s.dly = sqbuf[s.tail];
        SqShadow sh = {st.sh_nf, st.sh_cap, st.sh_capped};
        if (mode == 5) s.dly = i0 >= 102 ? sq_shadow_value(sh) : 0.0f; /* what buffer_[tail] holds in front of sample i0: written by sample i0 - 102 */
        for (int i = i0; i < i0 + m; i++) {
            if (mode == 3 && ((s.sample_count + 1u) & 3u) == 0u && i + 4 <= i0 + m && sq_stable4(s) && sq_raw_stable4(s, L, raw + i)) {
                /* a committed stable group: lane masks, noise floor and level are those of all four samples */
                ++*groups;
                for (int r = 0; r < 4; r++) {
                    if (flags) flags[i + r] = flags_of(s);
                    if (noise) noise[i + r] = s.noise_floor;
                    if (level) level[i + r] = sq_level(s);
                }
                i += 3;
                continue;
            }
            float dly_new = 0.0f;
            if (mode == 2) dly_new = sqbuf[(s.tail + 1) % AB_SQ_BUF]; /* the entry the sample sees after its tail increment */
            if (mode == 5) { /* sample i sees what sample i - 101 wrote: feed the shadow that sample, then read it */
                if (i >= 101) sq_shadow_step(sh, L, raw[i - 101], (unsigned)i);
                dly_new = i >= 101 ? sq_shadow_value(sh) : 0.0f;
            }
            sq_raw(s, L, raw[i], dly_new);
            if (lowpass) sq_filtered(s, L, sq_should_filter(s), filtered[i]);
            if (flags) flags[i] = flags_of(s);
            if (noise) noise[i] = s.noise_floor;
            if (level) level[i] = sq_level(s);
        }
        sq_store(s, L, &st, m);
        st.sh_nf = sh.nf; st.sh_cap = sh.cap; st.sh_capped = sh.capped;
    }
    out_state[0] = st.cur; out_state[1] = st.next; out_state[2] = st.delay; out_state[3] = st.low_count; out_state[4] = st.head; out_state[5] = st.tail;
    out_state[6] = st.using_post; out_state[7] = st.sample_count;
    counts[0] = st.open_count; counts[1] = st.flappy_count; counts[2] = st.recent_open; counts[3] = st.closed_count; counts[4] = groups_committed;
    return 0;
}
}
