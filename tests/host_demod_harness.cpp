// tests/host_demod_harness.cpp -- TEST HARNESS ONLY (built by tests/test_host_demod.py with the host clang++ into a temporary directory; never part
// of, linked into or loaded by the library).
// Compiles the stage-2 kernel source itself -- csrc/demod.hip, through tests/hostshim/hip/hip_runtime.h -- as plain C++ and runs the lane-per-channel
// demod kinds that have no cross-lane step (AM, NFM, NFM + lowpass) one lane at a time on host memory laid out like the library's device buffers,
// so that the per-sample arithmetic of the kernels (squelch, derotation, lowpass, discriminator, AGC, notch, clamp, fade-out, tail copy, zero-row skipping,
// the short sqrt / division sequences) can be compared bit for bit with the oracle WITHOUT a GPU.  The CTCSS kinds (wavefront-per-channel tone kernel)
// and the channelizer (matrix cores, cross-lane exchange) are not reachable this way; the GPU parity tests cover those.
//
// Second mode (-DAB_WAVE64_EMU, through tests/hostshim_wave64/ instead): the 64 lanes of a wavefront run as fibers that meet at every cross-lane
// operation, kernels are "launched" by the library's own launch_demod() -- so ALL stage-2 kinds run, the CTCSS chain (front, tone, back kernels) and the
// cooperative stores of full 64-channel blocks included, with real lane masks.
#include <cstdlib>
#include <new>
#include <vector>

#include "../rtlsdr-airband_amd/csrc/demod.hip"
#include "../rtlsdr-airband_amd/csrc/params.h"

using namespace airband;

namespace {

template <class T>
T* aligned_array(size_t n, T fill) {
    void* p = nullptr;
    if (posix_memalign(&p, 128, (n ? n : 1) * sizeof(T)) != 0) throw std::bad_alloc();
    T* a = static_cast<T*>(p);
    for (size_t i = 0; i < n; i++) a[i] = fill;
    return a;
}

int kind_of(const ChanConst& c) { /* airband_hip_prepare()'s rule (csrc/airband_hip.cpp) */
    if (c.flags & AB_F_IQ_OUT) return AB_KIND_GENERIC;
    const bool nfm = c.flags & AB_F_NFM, raw = c.flags & AB_F_RAW_IQ, lp = c.flags & AB_F_LOWPASS, ct = c.flags & AB_F_CTCSS;
    if (!nfm) return (!raw && !ct) ? AB_KIND_AM : AB_KIND_GENERIC;
    if (ct && lp) return AB_KIND_GENERIC;
    return ct ? AB_KIND_NFM_CTCSS : lp ? AB_KIND_NFM_LOWPASS : AB_KIND_NFM;
}

struct HostDemod {
    Plan plan;
    int B = 0, R = 0, n_slots = 0, wave_stride = 0, row0 = 0;
    uint64_t batches = 0;
    std::vector<int> slot_to_ext, ext_to_slot;
    int kind_first[AB_KIND_COUNT] = {0}, kind_blocks[AB_KIND_COUNT] = {0};
    std::vector<ChanConst> cc;
    std::vector<ChanState> cs;
    float* mag = nullptr;
    float2* iq = nullptr;
    float2* iq_out = nullptr;
    float* sqbuf = nullptr;
    float* out_wave = nullptr;
    uint8_t* out_axc = nullptr;
    uint8_t* trace = nullptr;
    float* lds = nullptr;
    /* split kinds (wave64 mode) */
    float* ct_coeff = nullptr;
    float* ct_q = nullptr;
    float2* ct_af = nullptr;
    unsigned long long* ct_mask = nullptr;
    int ct_first_block = 0, ct_n_blocks = 0, ct_pk_pitch = 0, ct_stride = 0;
    /* regrouped stage 2 (AB_HOST_REGROUP=1 in the environment when the handle is created; wave64 mode): workgroups of AB_REGROUP_WAVES wavefronts deal their slots out by squelch state */
    bool regroup = false;
    int regroup_mode = 1;
    std::vector<uint8_t> sq_key;
    std::vector<int> perm;
    ~HostDemod() {
        free(mag); free(iq); free(iq_out); free(sqbuf); free(out_wave); free(out_axc); free(trace); free(lds);
        free(ct_coeff); free(ct_q); free(ct_af); free(ct_mask);
    }
};

#ifndef AB_WAVE64_EMU
template <int KIND>
void run_kind(HostDemod* h, const DemodArgs& a) {
    for (int b = 0; b < h->kind_blocks[KIND]; b++) {
        float2* lut = reinterpret_cast<float2*>(h->lds);
        for (int i = 0; i < 257; i++) lut[i] = make_float2(a.sin_lut[i], a.cos_lut[i]); /* a lane fills every 64th entry; one lane at a time needs them all */
        for (unsigned lane = 0; lane < 64; lane++) {
            threadIdx.x = lane;
            blockIdx.x = (unsigned)b;
            demod_block<KIND, false, 1>(a, h->kind_first[KIND], h->kind_blocks[KIND], h->lds);
        }
    }
    threadIdx.x = 0;
    blockIdx.x = 0;
}
#endif

}  // namespace

#ifdef AB_WAVE64_EMU
namespace airband {
alignas(16) float lds_demod[258 * 2 + AB_REGROUP_WAVES * WAVE_LDS_FLOATS + AB_REGROUP_WAVES * 64 + 2 * AB_REGROUP_WAVES]; /* demod_kernel's dynamic LDS at its largest (regrouped workgroups) */ /* demod_kernel's dynamic LDS (one block runs at a time) */
}
#endif

extern "C" {

// 0 or a negative AIRBAND_HIP_E* code; -100: the plan has a channel of a kind this harness cannot run (CTCSS, raw-I/Q outputs)
int hostdemod_create(const airband_hip_config* cfg, int trace, void** out) {
    HostDemod* h = new HostDemod();
    const int rc = build_plan(cfg, h->plan);
    if (rc != 0) {
        delete h;
        return rc;
    }
    const Plan& p = h->plan;
    h->B = p.wave_batch;
    h->R = (p.wave_batch + AB_AGC_EXTRA + 15) / 16 * 16;
    ChanConst pad_c;
    ChanState pad_s;
    std::memset(&pad_c, 0, sizeof(pad_c));
    std::memset(&pad_s, 0, sizeof(pad_s));
    pad_c.ct_slot = -1;
    pad_s.axc = ' ';
    h->ext_to_slot.assign(p.total_ch, -1);
    int blocks = 0;
    for (int k = 0; k < AB_KIND_COUNT; k++) {
        bool any = false;
        for (int e = 0; e < p.total_ch; e++) {
            if (kind_of(p.cc[e]) != k) continue;
#ifndef AB_WAVE64_EMU
            if (k == AB_KIND_NFM_CTCSS || k == AB_KIND_GENERIC) {
                delete h;
                return -100;
            }
#endif
            any = true;
            h->ext_to_slot[e] = (int)h->cc.size();
            h->slot_to_ext.push_back(e);
            h->cc.push_back(p.cc[e]);
            h->cs.push_back(p.cs0[e]);
        }
        if (!any) continue;
        while (h->cc.size() % AB_SLOT_BLOCK) {
            h->slot_to_ext.push_back(-1);
            h->cc.push_back(pad_c);
            h->cs.push_back(pad_s);
        }
        h->kind_first[k] = blocks;
        h->kind_blocks[k] = (int)h->cc.size() / AB_SLOT_BLOCK - blocks;
        blocks = (int)h->cc.size() / AB_SLOT_BLOCK;
    }
    h->n_slots = (int)h->cc.size();
    h->wave_stride = (AB_OUT_PAD + AB_AGC_EXTRA + h->B + AB_OUT_RUN - 1) / AB_OUT_RUN * AB_OUT_RUN;
    const size_t ring = (size_t)h->R * h->n_slots;
    h->mag = aligned_array<float>(ring, 20.0f); /* src/config.cpp:313-316 */
    h->iq = aligned_array<float2>(ring, make_float2(0.0f, 0.0f));
    h->iq_out = aligned_array<float2>((size_t)h->B * h->n_slots, make_float2(0.0f, 0.0f));
    h->sqbuf = aligned_array<float>((size_t)AB_SQ_BUF * h->n_slots, 0.0f);
    h->out_wave = aligned_array<float>((size_t)p.total_ch * h->wave_stride, 0.0f);
    for (int c = 0; c < p.total_ch; c++)
        for (int k = 0; k < AB_AGC_EXTRA; k++) h->out_wave[(size_t)c * h->wave_stride + AB_OUT_PAD + k] = 0.5f;
    h->out_axc = aligned_array<uint8_t>((size_t)p.total_ch, (uint8_t)' ');
    if (trace) h->trace = aligned_array<uint8_t>((size_t)h->B * h->n_slots, 0);
    h->lds = aligned_array<float>(258 * 2 + (size_t)RUN * OSTRIDE + 3 * 64, 0.0f);
    { /* CTCSS tables and the hand-off buffers of the split kinds, as airband_hip_prepare() lays them out */
        const int n_ct = (int)p.tones.size();
        h->ct_stride = n_ct;
        h->ct_coeff = aligned_array<float>((size_t)(n_ct > 0 ? n_ct : 1) * 2 * AB_MAX_TONES, 0.0f);
        for (int s = 0; s < n_ct; s++)
            for (int k = 0; k < 2; k++)
                for (int t = 0; t < p.tones[s].n[k]; t++) h->ct_coeff[((size_t)s * 2 + k) * AB_MAX_TONES + t] = p.tones[s].coeff[k][t];
        h->ct_q = aligned_array<float>((size_t)(n_ct > 0 ? n_ct : 1) * 4 * AB_MAX_TONES, 0.0f);
        h->ct_n_blocks = h->kind_blocks[AB_KIND_NFM_CTCSS] + h->kind_blocks[AB_KIND_GENERIC];
        h->ct_first_block = h->kind_blocks[AB_KIND_NFM_CTCSS] ? h->kind_first[AB_KIND_NFM_CTCSS] : h->kind_first[AB_KIND_GENERIC];
        h->ct_pk_pitch = (h->B + 31) / 32 * 32;
        if (h->ct_n_blocks > 0) {
            h->ct_af = aligned_array<float2>((size_t)h->kind_blocks[AB_KIND_NFM_CTCSS] * AB_SLOT_BLOCK * h->ct_pk_pitch / 2 + (size_t)h->kind_blocks[AB_KIND_GENERIC] * AB_SLOT_BLOCK * h->B,
                                             make_float2(0.0f, 0.0f));
            h->ct_mask = aligned_array<unsigned long long>((size_t)h->ct_n_blocks * (h->B / 50) * AB_SLOT_BLOCK, 0ull);
        }
    }
#ifdef AB_WAVE64_EMU
    {
        const char* e = getenv("AB_HOST_REGROUP");
        h->regroup = e && *e && *e != '0';
        h->regroup_mode = (e && *e == '2') ? 2 : (e && *e == '3') ? 3 : 1; /* 2: line groups sorted, the workgroup's wavefronts free-running; 3: a permutation in front of one-wavefront workgroups */
        if (h->regroup_mode == 3) h->perm.assign((size_t)h->n_slots, 0);
    }
#endif
    h->sq_key.assign((size_t)h->n_slots, 0); /* the front kernel's note per channel (tone kernel: channels without audio in the batch are skipped) */
    *out = h;
    return 0;
}

void hostdemod_destroy(void* hv) { delete static_cast<HostDemod*>(hv); }

int hostdemod_wave_batch(void* hv) { return static_cast<HostDemod*>(hv)->B; }

// one batch of stage 2 on the given stage-1 output: wavein [total_channels][wave_batch], iq_in [total_channels][2 * wave_batch] -- what
// airband_hip_process_bins() takes (scatter_bins_kernel, then launch_demod)
int hostdemod_process_bins(void* hv, const float* wavein, const float* iq_in) {
    HostDemod* h = static_cast<HostDemod*>(hv);
    const int B = h->B, R = h->R;
    for (int slot = 0; slot < h->n_slots; slot++) {
        const int ext = h->slot_to_ext[slot];
        if (ext < 0) continue;
        for (int t = 0; t < B; t++) {
            int row = h->row0 + AB_AGC_EXTRA + t;
            if (row >= R) row -= R;
            const long off = ab_tile_base(slot, R / AB_TILE_ROWS) + ab_tile_off(row);
            h->mag[off] = wavein[(long)ext * B + t];
            if (h->cc[slot].flags & AB_F_RAW_IQ) h->iq[off] = make_float2(iq_in[((long)ext * B + t) * 2], iq_in[((long)ext * B + t) * 2 + 1]);
        }
    }
    DemodArgs a;
    std::memset(&a, 0, sizeof(a));
    a.cc = h->cc.data();
    a.cs = h->cs.data();
    a.mag = h->mag;
    a.iq = h->iq;
    a.out_wave = h->out_wave;
    a.out_axc = h->out_axc;
    a.slot_to_ext = h->slot_to_ext.data();
    a.wave_stride = h->wave_stride;
    a.tail_copy = h->batches > 0 ? 1 : 0;
    a.iq_out = h->iq_out;
    a.sqbuf = h->sqbuf;
    a.trace = h->trace;
    a.sin_lut = h->plan.sin_lut.data();
    a.cos_lut = h->plan.cos_lut.data();
    a.n_slots = h->n_slots;
    a.wave_batch = B;
    a.row0 = h->row0;
    a.ring_rows = R;
    a.sq_key = h->sq_key.data();
#ifdef AB_WAVE64_EMU
    a.ct_coeff = h->ct_coeff;
    a.ct_q = h->ct_q;
    a.ct_pk_first_block = h->kind_first[AB_KIND_NFM_CTCSS];
    a.ct_pk_n_blocks = h->kind_blocks[AB_KIND_NFM_CTCSS];
    a.ct_gen_first_block = h->kind_first[AB_KIND_GENERIC];
    a.ct_gen_n_blocks = h->kind_blocks[AB_KIND_GENERIC];
    a.ct_pk_pitch = h->ct_pk_pitch;
    a.ct_ap = reinterpret_cast<unsigned*>(h->ct_af);
    a.ct_af = h->ct_af + (size_t)a.ct_pk_n_blocks * AB_SLOT_BLOCK * h->ct_pk_pitch / 2;
    a.ct_mask = h->ct_mask;
    a.ct_first_block = h->ct_first_block;
    a.ct_n_blocks = h->ct_n_blocks;
    a.ct_stride = h->ct_stride;
    a.sq_key = h->sq_key.data();
    if (h->regroup) a.regroup = h->regroup_mode;
    if (h->regroup && h->regroup_mode == 3) a.perm = h->perm.data();
    launch_demod(a, h->kind_first, h->kind_blocks, nullptr, nullptr, nullptr); /* the library's own launch sequence; every launch runs to completion */
#else
    run_kind<AB_KIND_NFM_LOWPASS>(h, a);
    run_kind<AB_KIND_NFM>(h, a);
    run_kind<AB_KIND_AM>(h, a);
#endif
    h->row0 = (h->row0 + B) % R;
    h->batches++;
    return 0;
}

// results of the last batch: waveout [total_channels][wave_batch], axc [total_channels], trace [total_channels][wave_batch] (if created with trace)
void hostdemod_collect(void* hv, float* waveout, uint8_t* axc, uint8_t* trace) {
    HostDemod* h = static_cast<HostDemod*>(hv);
    const int B = h->B, n = h->plan.total_ch;
    for (int c = 0; c < n; c++) {
        if (waveout) std::memcpy(waveout + (size_t)c * B, h->out_wave + (size_t)c * h->wave_stride + AB_OUT_PAD, sizeof(float) * B);
        if (axc) axc[c] = h->out_axc[c];
        if (trace && h->trace) {
            const int slot = h->ext_to_slot[c];
            for (int t = 0; t < B; t++) trace[(size_t)c * B + t] = h->trace[ab_ring_base(slot, B) + (long)t * AB_SLOT_BLOCK];
        }
    }
}

// per-channel statistics after the last batch, by the library's own stats_kernel (what airband_hip_collect() hands out): out [total_channels]
void hostdemod_stats(void* hv, airband_hip_channel_stats* out) {
    HostDemod* h = static_cast<HostDemod*>(hv);
    blockDim.x = 64;
    for (int slot = 0; slot < h->n_slots; slot++) {
        blockIdx.x = (unsigned)(slot / 64);
        threadIdx.x = (unsigned)(slot % 64);
        stats_kernel(h->cc.data(), h->cs.data(), h->slot_to_ext.data(), h->n_slots, out);
    }
    blockIdx.x = 0;
    threadIdx.x = 0;
}
}
