/* csrc/airband_hip.cpp -- the C ABI of libairband_hip.so (see include/airband_hip.h) and the host-side batch
 * driver: what demodulate() does around its two inner loops (reference: src/rtl_airband.cpp:359-400 input
 * accounting, :494/:649-669 batch hand-off), restated for "all dongles, one batch at a time" on one HIP stream.
 *
 * There is NO CPU fallback in this library: every data-path entry point needs the HIP device the handle was
 * prepared on and fails with AIRBAND_HIP_ENODEV / AIRBAND_HIP_ERUNTIME otherwise.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h> /* types and prototypes only: librccl.so is loaded on first use (airband_hip_comm_*) */

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/airband_hip.h"
#include "common.h"
#include "kernels.h"
#include "params.h"

/* upper bound for the int8 coefficient tables of a handle -- one per distinct group of eight bins plus one per AFC group; past it prepare() picks the wavefront-FFT
 * channelizer (override for tests) */
#ifndef AB_PRIVATE_TABLE_BUDGET
#define AB_PRIVATE_TABLE_BUDGET ((size_t)8 << 30)
#endif

/* upper bound for the float coefficient tables (CF32 dongles on the float32 matrix pipe) of a handle, host-built: 512 MiB = 8 192 distinct channel plans at fft 512 */
#ifndef AB_F32_TABLE_BUDGET
#define AB_F32_TABLE_BUDGET ((size_t)512 << 20)
#endif

using namespace airband;

namespace {

thread_local std::string g_prepare_error;

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace

struct airband_hip_handle {
    Plan plan;
    uint32_t flags = 0;
    int hip_device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side[3] = {nullptr, nullptr, nullptr}; /* fused demod kinds run beside the CTCSS chain */
    hipEvent_t fork_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    /* AIRBAND_HIP_FLAG_PIPELINE: stage 1 of batch k runs on `front` while stage 2 of batch k-1 runs on `stream` */
    bool pipeline = false;
    hipStream_t front = nullptr;
    hipEvent_t ev_in = nullptr, ev_back = nullptr, ev_wait = nullptr, front_done[2] = {nullptr, nullptr};
    hipStream_t last_stream = nullptr; /* stream the last sequential batch ran on (the caller's or ours) */
    hipEvent_t ev_spec[2] = {nullptr, nullptr}; /* AFC on the matrix-core channelizer: the last hop's spectrum runs on a side stream beside stage 1 (fork, done) */
    hipEvent_t ev_last = nullptr;      /* recorded behind every batch that ran on a CALLER's stream: collect / read_* / synchronize / release
                                          order themselves behind it (the caller's stream itself may be gone by then, our event is not) */
    bool ev_last_pending = false;
    int row0_front = 0;            /* ring row of the batch stage 1 writes next (== row0 when not pipelined) */
    uint64_t front_batches = 0;    /* batches whose stage 1 has been enqueued */
    /* per-stage GPU time: a pool of event sets (one per process call) harvested lazily, so that nobody has to
     * synchronise inside a run to read timings */
    static constexpr int EV_POOL = 16;
    hipEvent_t evp[EV_POOL][5] = {}; /* stage 1 begin / end, stage 2 begin, demod end, batch end */
    uint8_t evp_state[EV_POOL] = {0}; /* bit 0: stage-1 pair recorded, bit 1: stage-2 pair recorded */
    double t_sum[4] = {0, 0, 0, 0};
    int64_t t_n[2] = {0, 0};          /* harvested stage-1 / stage-2 pairs */
    float t_last[4] = {0, 0, 0, 0};
    bool timings_valid = false;
    std::string error;

    /* geometry */
    int B = 0, R = 0, N = 0;
    int n_slots = 0;      /* demod slots: channels sorted by kind, every kind padded to whole 64-slot blocks */
    std::vector<int> slot_to_ext, ext_to_slot;
    int kind_first_block[AB_KIND_COUNT] = {0}, kind_n_blocks[AB_KIND_COUNT] = {0};
    int64_t hop_bytes = 0, first_batch_bytes = 0, batch_bytes = 0, lookahead_bytes = 0;
    int row0 = 0;
    int wave_stride = 0;   /* floats between two channels' rows of d_out_wave */
    uint64_t batches_done = 0;
    bool results_ready = false;
    uint64_t overruns = 0;

    /* device memory */
    DevBuf<DevConst> d_dev;
    DevBuf<ChanConst> d_cc;
    DevBuf<ChanState> d_cs;
    DevBuf<int> d_slot_to_ext, d_ext_to_slot;
    DevBuf<uint8_t> d_block_kind;
    DevBuf<float> d_window, d_sin, d_cos, d_twiddle;
    DevBuf<float> d_window_dec; /* fft_size >= 1024: the window de-interleaved by sample index mod (fft_size / 512), for the decimated wavefront FFT (channelizer_fft.hip) */
    DevBuf<float> d_mag, d_sqbuf, d_ct_coeff, d_ct_q;
    DevBuf<float2> d_iq, d_iq_out, d_ct_af;
    DevBuf<unsigned long long> d_ct_mask;
    int ct_first_block = 0, ct_n_blocks = 0, ct_pk_pitch = 0;
    /* AIRBAND_HIP_FLAG_REGROUP: the batch's slot order (demod.hip, "regrouping") */
    bool regroup = false;
    int regroup_mode = 1;          /* 1: channels sorted inside lockstep workgroups; 2: line groups sorted, wavefronts free-running (demod.hip) */
    DevBuf<int> d_perm;       /* regroup mode 3: the batch's slot permutation (demod.hip, regroup_perm_kernel) */
    DevBuf<uint8_t> d_sq_key; /* split kinds: the front kernel's note for the back kernel (had audio in this batch) */
    DevBuf<uint8_t> d_trace;
    DevBuf<float> d_out_wave, d_out_iq;
    DevBuf<uint8_t> d_out_axc;
    DevBuf<airband_hip_channel_stats> d_stats;
    DevBuf<float> d_tmp_wavein, d_tmp_iqin, d_spectrum;
    bool any_afc = false, afc_spectrum_valid = false; /* process_bins() has no spectrum: AFC is skipped there */
    DevBuf<uint8_t> d_tmp_trace;
    int ct_stride = 0;
    /* matrix-core channelizer */
    bool use_f32 = false;          /* CF32 dongles on the float32 matrix pipe (channelizer_f32.hip) */
    DevBuf<float> d_ftab;
    bool use_dft = false;
    DevBuf<int> d_item_dev, d_item_group, d_item_bset, d_item_private, d_item_home; /* d_item_bset: what stage 1 reads (the re-tune kernel switches AFC groups between their home and private tables) */
    DevBuf<int8_t> d_bfrag;
    DevBuf<double> d_bcorr;
    DevBuf<float> d_dft_partial; /* fft_size 8192: partial sums between the two passes of eight window pieces */
    DevBuf<int> d_bset_bin;      /* [n_bsets][8] bin each coefficient column pair is built for (AFC re-tunes private tables on the device) */
    const void* last_iq = nullptr; /* input of the batch stage 1 ran last (AFC looks at its last hop once stage 2 has decided) */
    size_t last_iq_stride = 0;
    int last_n_hops = 0;

    /* host-ring path: one PINNED circular buffer per dongle (row d of h_ring, ring_cap bytes).  submit() copies the caller's bytes
     * straight into it -- the only CPU copy on the way -- and may be called for DIFFERENT dongles from several threads at once;
     * process() ships a batch with (at most two, where the span wraps) strided DMA transfers on a copy stream into one of two
     * device staging buffers while the kernels of the previous batch still read the other. */
    std::atomic<uint8_t*> h_ring{nullptr};                 /* published (release) by host_path_init() once ring_cap / stage_stride / the staging buffers exist; read (acquire) by submit() and process() */
    int64_t ring_cap = 0;                                  /* bytes per dongle */
    std::unique_ptr<std::atomic<uint64_t>[]> ring_wr;      /* per dongle: stream bytes accepted so far */
    uint64_t ring_rd = 0;                                  /* stream position of the next batch (common to all dongles: they advance in lockstep) */
    std::atomic<uint64_t> ring_free{0};                    /* stream position up to which the ring may be overwritten (lags ring_rd by the batch in flight) */
    DevBuf<uint8_t> d_stage2[2];
    int64_t stage_stride = 0;
    hipStream_t h2d = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_stage_read[2] = {nullptr, nullptr};
    uint64_t host_batches = 0;
    std::mutex host_init_lock;

    /* dongles switched off with airband_hip_device_enable(): skipped by the availability rule and by both stages */
    std::unique_ptr<std::atomic<uint8_t>[]> dev_enabled; /* (atomic: a feeder thread's submit() reads its dongle's flag while the demod thread switches it) */
    int n_enabled = 0;
    std::vector<ChanConst> cc_slots; /* host copy of d_cc (slot order): the VALID bit of a dongle's slots follows its enable state */

    /* mixers */
    std::vector<int> mix_pos;       /* connection index (order of airband_hip_set_mixers) -> position in the per-mixer grouped arrays */
    std::vector<int> mix_chan_host; /* grouped external channel indices, for re-enabling an input */
    std::vector<uint8_t> mix_user_on; /* by position: airband_hip_mixer_enable_input()'s say; an input counts while this AND its dongle are on */
    int n_mixers = 0;
    int n_mix_runs = 0;
    DevBuf<int> d_mix_chan, d_mix_first, d_mix_run_first, d_mix_run_mixer, d_mix_first_run;
    DevBuf<float> d_mix_run_left, d_mix_run_right;
    DevBuf<uint8_t> d_mix_run_signal;
    DevBuf<float> d_mix_ml, d_mix_mr, d_mix_left, d_mix_right;
    DevBuf<uint8_t> d_mix_stereo, d_mix_signal;

    /* the mixer exchange (airband_hip_comm_*): this handle's rank in an RCCL communicator over the GPUs that hold the other dongles */
    ncclComm_t comm = nullptr;
    hipEvent_t ev_peer = nullptr; /* airband_hip_add_mixers: "src's batch is done" for dst's stream */

    /* synthetic dongles */
    DevBuf<int16_t> d_sin_tab;
    DevBuf<long long> d_carriers;
    int n_carriers = 0, noise_q8 = 0;
    int sig_n_plans = 1;            /* airband_hip_set_signal_plan_shift */
    unsigned sig_shift_step = 0;
};

namespace {

int fail(airband_hip_handle* h, int code, const std::string& msg) {
    if (h) h->error = msg;
    else g_prepare_error = msg;
    return code;
}

#define HIP_TRY(h, expr, code)                                                                                   \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return fail(h, code, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

template <class T>
hipError_t upload(DevBuf<T>& b, const std::vector<T>& v) {
    hipError_t e = b.alloc(v.size());
    if (e != hipSuccess) return e;
    if (v.empty()) return hipSuccess;
    return hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

void destroy(airband_hip_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->hip_device);
    (void)airband_hip_comm_destroy(h);
    if (h->ev_peer) (void)hipEventDestroy(h->ev_peer);
    /* everything in flight must be done before its memory goes away: the front stream of a pipelined handle, the forked demod streams */
    if (h->ev_last && h->ev_last_pending) (void)hipEventSynchronize(h->ev_last); /* a batch enqueued on a caller's stream */
    if (h->front) (void)hipStreamSynchronize(h->front);
    for (auto& st : h->side)
        if (st) (void)hipStreamSynchronize(st);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->d_dev.release(); h->d_cc.release(); h->d_cs.release(); h->d_slot_to_ext.release(); h->d_ext_to_slot.release(); h->d_block_kind.release();
    h->d_window.release(); h->d_sin.release(); h->d_cos.release(); h->d_twiddle.release(); h->d_window_dec.release();
    h->d_mag.release(); h->d_sqbuf.release(); h->d_ct_coeff.release(); h->d_ct_q.release();
    h->d_iq.release(); h->d_iq_out.release(); h->d_trace.release(); h->d_ct_af.release(); h->d_ct_mask.release();
    h->d_out_wave.release(); h->d_out_iq.release(); h->d_out_axc.release(); h->d_stats.release();
    h->d_tmp_wavein.release(); h->d_tmp_iqin.release(); h->d_tmp_trace.release(); h->d_spectrum.release();
    h->d_sq_key.release(); h->d_perm.release();
    h->d_ftab.release(); h->d_item_dev.release(); h->d_item_group.release(); h->d_item_bset.release(); h->d_item_private.release(); h->d_item_home.release(); h->d_bfrag.release(); h->d_bcorr.release(); h->d_dft_partial.release(); h->d_bset_bin.release();
    if (h->h2d) (void)hipStreamSynchronize(h->h2d);
    h->d_stage2[0].release(); h->d_stage2[1].release();
    if (h->h_ring.load()) (void)hipHostFree(h->h_ring.load());
    for (auto& e : h->ev_h2d)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : h->ev_stage_read)
        if (e) (void)hipEventDestroy(e);
    if (h->h2d) (void)hipStreamDestroy(h->h2d);
    h->d_mix_chan.release(); h->d_mix_first.release(); h->d_mix_ml.release(); h->d_mix_mr.release();
    h->d_mix_left.release(); h->d_mix_right.release(); h->d_mix_stereo.release(); h->d_mix_signal.release();
    h->d_mix_run_first.release(); h->d_mix_run_mixer.release(); h->d_mix_first_run.release();
    h->d_mix_run_left.release(); h->d_mix_run_right.release(); h->d_mix_run_signal.release();
    h->d_sin_tab.release(); h->d_carriers.release();
    for (auto& set : h->evp)
        for (auto& e : set)
            if (e) (void)hipEventDestroy(e);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_back) (void)hipEventDestroy(h->ev_back);
    if (h->ev_wait) (void)hipEventDestroy(h->ev_wait);
    if (h->ev_last) (void)hipEventDestroy(h->ev_last);
    for (auto& e : h->ev_spec)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : h->front_done)
        if (e) (void)hipEventDestroy(e);
    if (h->front) (void)hipStreamDestroy(h->front);
    for (auto& e : h->fork_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& st : h->side)
        if (st) (void)hipStreamDestroy(st);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

/* Results of a batch that ran on a caller's stream: make the handle's own stream (on which collect / read_* / the stats kernel
 * are issued) wait for it on the GPU. */
void order_behind_last_batch(airband_hip_handle* h) {
    if (h->ev_last && h->ev_last_pending) (void)hipStreamWaitEvent(h->stream, h->ev_last, 0);
}

/* Per-stage GPU times: every batch records into its own event set; finished sets are folded into running sums here. */
void harvest_timings(airband_hip_handle* h, bool wait) {
    for (int i = 0; i < airband_hip_handle::EV_POOL; i++) {
        if (h->evp_state[i] != 3) continue;
        hipEvent_t* e = h->evp[i];
        if (wait) {
            if (hipEventSynchronize(e[4]) != hipSuccess || hipEventSynchronize(e[1]) != hipSuccess) continue;
        } else if (hipEventQuery(e[4]) != hipSuccess || hipEventQuery(e[1]) != hipSuccess) {
            continue;
        }
        float a = 0, b = 0, c = 0;
        if (hipEventElapsedTime(&a, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&b, e[2], e[3]) == hipSuccess && hipEventElapsedTime(&c, e[3], e[4]) == hipSuccess) {
            h->t_last[0] = a; h->t_last[1] = b; h->t_last[2] = c; h->t_last[3] = a + b + c;
            for (int k = 0; k < 4; k++) h->t_sum[k] += h->t_last[k];
            h->t_n[0]++;
            h->timings_valid = true;
        }
        h->evp_state[i] = 0;
    }
}

hipEvent_t* event_set(airband_hip_handle* h, uint64_t batch, int half) {
    const int i = (int)(batch % airband_hip_handle::EV_POOL);
    if (half == 0 && h->evp_state[i] != 0) { /* the pool wrapped around a set nobody has read yet */
        harvest_timings(h, true);
        h->evp_state[i] = 0;
    }
    h->evp_state[i] |= (uint8_t)(1u << half);
    return h->evp[i];
}

void launch_retune_tables(airband_hip_handle* h, hipStream_t s, int epoch) {
    RetuneArgs ra;
    ra.cc = h->d_cc.p;
    ra.cs = h->d_cs.p;
    ra.dev = h->d_dev.p;
    ra.ext_to_slot = h->d_ext_to_slot.p;
    ra.item_dev = h->d_item_dev.p;
    ra.item_group = h->d_item_group.p;
    ra.item_bset = h->d_item_bset.p;
    ra.item_private = h->d_item_private.p;
    ra.item_home = h->d_item_home.p;
    ra.bset_bin = h->d_bset_bin.p;
    ra.bfrag = h->d_bfrag.p;
    ra.corr = h->d_bcorr.p;
    ra.ftab = h->use_f32 ? h->d_ftab.p : nullptr;
    ra.window = h->d_window.p;
    ra.n_items = (int)h->plan.item_dev.size();
    ra.fft_size = h->plan.fft_size;
    ra.n_shared = h->plan.n_shared_bsets;
    ra.moved_epoch = h->d_bset_bin.p + (size_t)h->plan.n_bsets * 8; /* one more int behind the table */
    ra.epoch = epoch;
    launch_retune(ra, s);
}

/* matrix-core handles with AFC channels: the full spectrum of the batch's LAST hop, for the dongles that have one (one wavefront FFT per
 * such dongle per batch -- afc.finalize(dev, i, fftout) runs once per batch on the output of its last FFT, src/rtl_airband.cpp:626-630) */
void launch_last_hop_spectrum(airband_hip_handle* h, hipStream_t s) {
    const Plan& p = h->plan;
    ChannelizerArgs ca;
    ca.iq = (const uint8_t*)h->last_iq + (size_t)(h->last_n_hops - 1) * (size_t)h->hop_bytes;
    ca.iq_stride = (long)h->last_iq_stride;
    ca.dev = h->d_dev.p;
    ca.cs = h->d_cs.p;
    ca.cc = h->d_cc.p;
    ca.ext_to_slot = h->d_ext_to_slot.p;
    ca.window = h->d_window.p;
    ca.window_dec = h->d_window_dec.p;
    ca.twiddle = reinterpret_cast<const float2*>(h->d_twiddle.p);
    ca.mag = h->d_mag.p;
    ca.iq_bins = h->d_iq.p;
    ca.last_spectrum = h->d_spectrum.p;
    ca.n_dev = p.n_dev;
    ca.fft_log = p.fft_log;
    ca.hop_samples = p.dev[0].hop_samples;
    ca.bytes_per_sample = p.dev[0].bytes_per_sample;
    ca.sfmt = p.dev[0].sfmt;
    ca.scale = p.dev[0].scale;
    ca.row0 = 0;
    ca.ring_rows = h->R;
    ca.first_row = 0;
    ca.n_hops = 1;
    ca.max_ch = p.max_ch;
    ca.spectrum_only = 1;
    launch_channelizer_fft(ca, s);
}

/* stage 2 + emit (+ mixers) of the batch whose stage-1 rows are already in the rings */
int run_back_half(airband_hip_handle* h, hipStream_t s) {
    hipEvent_t* ev = event_set(h, h->batches_done, 1);
    (void)hipEventRecord(ev[2], s);
    DemodArgs da;
    da.cc = h->d_cc.p;
    da.cs = h->d_cs.p;
    da.mag = h->d_mag.p;
    da.iq = h->d_iq.p;
    da.out_wave = h->d_out_wave.p; /* row starts; the kernels skip AB_OUT_PAD themselves */
    da.out_axc = h->d_out_axc.p;
    da.slot_to_ext = h->d_slot_to_ext.p;
    da.wave_stride = h->wave_stride;
    da.tail_copy = h->batches_done > 0 ? 1 : 0; /* the consumer's tail copy (src/output.cpp:920) happens after it has read a batch */
    da.iq_out = h->d_iq_out.p;
    da.sqbuf = h->d_sqbuf.p;
    da.ct_coeff = h->d_ct_coeff.p;
    da.ct_q = h->d_ct_q.p;
    da.trace = (h->flags & AIRBAND_HIP_FLAG_TRACE_SQUELCH) ? h->d_trace.p : nullptr;
    da.ct_pk_first_block = h->kind_first_block[AB_KIND_NFM_CTCSS];
    da.ct_pk_n_blocks = h->kind_n_blocks[AB_KIND_NFM_CTCSS];
    da.ct_gen_first_block = h->kind_first_block[AB_KIND_GENERIC];
    da.ct_gen_n_blocks = h->kind_n_blocks[AB_KIND_GENERIC];
    da.ct_pk_pitch = h->ct_pk_pitch;
    da.ct_ap = reinterpret_cast<unsigned*>(h->d_ct_af.p);                                                       /* one-word rows first ... */
    da.ct_af = h->d_ct_af.p + (size_t)da.ct_pk_n_blocks * AB_SLOT_BLOCK * h->ct_pk_pitch / 2;                   /* ... then the pairs of the generic kind */
    da.ct_mask = h->d_ct_mask.p;
    da.ct_first_block = h->ct_first_block;
    da.ct_n_blocks = h->ct_n_blocks;
    da.sin_lut = h->d_sin.p;
    da.cos_lut = h->d_cos.p;
    da.ct_stride = h->ct_stride;
    da.n_slots = h->n_slots;
    da.wave_batch = h->B;
    da.row0 = h->row0;
    da.ring_rows = h->R;
    da.regroup = h->regroup ? h->regroup_mode : 0;
    da.sq_key = h->d_sq_key.p;
    da.perm = (h->regroup && h->regroup_mode == 3) ? h->d_perm.p : nullptr;
    launch_demod(da, h->kind_first_block, h->kind_n_blocks, s, (h->flags & AIRBAND_HIP_FLAG_SERIAL_DEMOD) ? nullptr : h->side, h->fork_ev);
    if (h->any_afc && h->afc_spectrum_valid) { /* afc.finalize(), src/rtl_airband.cpp:626-630: may turn '*' into '<' / '>' */
        const bool tables = h->use_dft || h->use_f32; /* the matrix-core channelizers: a channel's bin is baked into its coefficient columns */
        if (tables) (void)hipStreamWaitEvent(s, h->ev_spec[1], 0); /* the last hop's spectrum, computed beside stage 1 */
        const int epoch = (int)(h->batches_done % 0x7fffffff) + 1; /* never 0: that is the start-up build's */
        launch_afc(h->d_cc.p, h->d_cs.p, h->d_spectrum.p, h->N, h->n_slots, tables ? h->d_bset_bin.p + (size_t)h->plan.n_bsets * 8 : nullptr, epoch, s);
        if (tables) launch_retune_tables(h, s, epoch); /* the next batch's stage 1 reads the moved channels' new columns */
        launch_axc(h->d_cc.p, h->d_cs.p, h->d_slot_to_ext.p, h->d_out_axc.p, h->n_slots, s);
    }
    (void)hipEventRecord(ev[3], s);
    if (h->d_out_iq.p) { /* handles with has_iq_outputs channels: raw I/Q rows -> channel-major */
        EmitArgs ea;
        ea.iq_out = h->d_iq_out.p;
        ea.slot_to_ext = h->d_slot_to_ext.p;
        ea.out_iq = h->d_out_iq.p;
        ea.n_slots = h->n_slots;
        ea.wave_batch = h->B;
        launch_emit_iq(ea, s);
    }
    if (h->n_mixers > 0) {
        MixArgs ma;
        ma.out_wave = h->d_out_wave.p + AB_OUT_PAD;
        ma.wave_stride = h->wave_stride;
        ma.out_axc = h->d_out_axc.p;
        ma.in_chan = h->d_mix_chan.p;
        ma.in_ml = h->d_mix_ml.p;
        ma.in_mr = h->d_mix_mr.p;
        ma.run_first = h->d_mix_run_first.p;
        ma.run_mixer = h->d_mix_run_mixer.p;
        ma.mixer_first_run = h->d_mix_first_run.p;
        ma.mixer_stereo = h->d_mix_stereo.p;
        ma.run_left = h->d_mix_run_left.p;
        ma.run_right = h->d_mix_run_right.p;
        ma.run_signal = h->d_mix_run_signal.p;
        ma.n_runs = h->n_mix_runs;
        ma.left = h->d_mix_left.p;
        ma.right = h->d_mix_right.p;
        ma.has_signal = h->d_mix_signal.p;
        ma.n_mixers = h->n_mixers;
        ma.wave_batch = h->B;
        launch_mix(ma, s);
    }
    (void)hipEventRecord(ev[4], s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, AIRBAND_HIP_ERUNTIME, std::string("kernel launch: ") + hipGetErrorString(e));
    /* rotate the rings: this batch's last AGC_EXTRA rows become the next batch's carry */
    h->row0 = (h->row0 + h->B) % h->R;
    h->batches_done++;
    if (h->results_ready) h->overruns++; /* like dev->output_overrun_count (src/rtl_airband.cpp:649-654) */
    h->results_ready = true;
    harvest_timings(h, false);
    return AIRBAND_HIP_OK;
}

/* position k of the grouped mixer-input arrays: the channel it reads, or -1 while the connection (airband_hip_mixer_enable_input) or its dongle
 * (airband_hip_device_enable) is switched off -- a masked input is skipped like mixer->input_mask[i] == false (src/mixer.cpp:96-110,192) */
hipError_t write_mix_input(airband_hip_handle* h, int k) {
    const int ch = h->mix_chan_host[k];
    const int v = (h->mix_user_on[k] && h->dev_enabled[h->plan.cc[ch].dev]) ? ch : -1;
    hipError_t e = hipMemcpyAsync(h->d_mix_chan.p + k, &v, sizeof(int), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(h->stream); /* `v` is a stack variable */
}

}  // namespace

extern "C" {

const char* airband_hip_last_error(const airband_hip_handle* h) { return h ? h->error.c_str() : g_prepare_error.c_str(); }

int airband_hip_derive_constants(const airband_hip_config* cfg, int32_t channel_index, double* out_vals) {
    Plan plan;
    int rc = build_plan(cfg, plan);
    if (rc != AIRBAND_HIP_OK) return fail(nullptr, rc, plan.error);
    if (channel_index < 0 || channel_index >= plan.total_ch || !out_vals) return fail(nullptr, AIRBAND_HIP_EINVAL, "bad channel index");
    channel_constants(plan, channel_index, out_vals);
    return AIRBAND_HIP_OK;
}

int airband_hip_dft_selftest(const airband_hip_config* cfg, int32_t windows, double* max_rel_err) {
    if (!cfg || !max_rel_err) return fail(nullptr, AIRBAND_HIP_EINVAL, "NULL argument");
    Plan plan;
    const int rc = build_plan(cfg, plan);
    if (rc != AIRBAND_HIP_OK) return fail(nullptr, rc, plan.error);
    const int hop_bytes = 2 * plan.dev[0].bytes_per_sample * plan.dev[0].hop_samples;
    if (plan.uniform_hop && f32_supported(plan.fft_size, plan.dev[0].hop_samples, plan.dev[0].sfmt)) { /* CF32: the float tables of channelizer_f32.hip */
        build_dft_tables(plan);
        build_f32_tables(plan);
        *max_rel_err = f32_table_selftest(plan, windows < 1 ? 1 : windows);
        return AIRBAND_HIP_OK;
    }
    if (!plan.uniform_hop || !dft_supported(plan.fft_size, hop_bytes, plan.dev[0].sfmt, plan.max_ch))
        return fail(nullptr, AIRBAND_HIP_EBADSIZE, "configuration does not take the matrix-core channelizer");
    build_dft_tables(plan);
    *max_rel_err = dft_table_selftest(plan, windows < 1 ? 1 : windows);
    return AIRBAND_HIP_OK;
}

int airband_hip_prepare(const airband_hip_config* cfg, airband_hip_handle** out) {
    if (!out) return fail(nullptr, AIRBAND_HIP_EINVAL, "out is NULL");
    *out = nullptr;
    airband_hip_handle* h = new (std::nothrow) airband_hip_handle();
    if (!h) return fail(nullptr, AIRBAND_HIP_ENOMEM, "host allocation failed");
    int rc = build_plan(cfg, h->plan);
    if (rc != AIRBAND_HIP_OK) {
        g_prepare_error = h->plan.error;
        delete h;
        return rc;
    }
    const Plan& p = h->plan;
    if (!p.uniform_hop) {
        delete h;
        return fail(nullptr, AIRBAND_HIP_EINVAL, "all dongles of one handle must share sample format and sample_rate/WAVE_RATE hop; use one handle per class");
    }
    bool any_afc = false;
    for (const ChanConst& c : p.cc) any_afc |= c.afc != 0;
    h->flags = cfg->flags;
    h->hip_device = cfg->hip_device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->hip_device < 0 || cfg->hip_device >= ndev) {
        delete h;
        return fail(nullptr, AIRBAND_HIP_ENODEV, "no usable HIP device (libairband_hip has no CPU fallback)");
    }
#define PREP_TRY(expr, code)                                                                 \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            g_prepare_error = std::string(#expr) + ": " + hipGetErrorString(e_);             \
            destroy(h);                                                                      \
            return code;                                                                     \
        }                                                                                    \
    } while (0)
    PREP_TRY(hipSetDevice(cfg->hip_device), AIRBAND_HIP_ENODEV);
    int prio_lo = 0, prio_hi = 0; /* numerically lower = more urgent */
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    PREP_TRY(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi), AIRBAND_HIP_ENODEV);
    for (auto& set : h->evp)
        for (auto& e : set) PREP_TRY(hipEventCreate(&e), AIRBAND_HIP_ENODEV);
    /* AFC needs stage 2's verdict on batch k before stage 1 of batch k+1 picks its bins (src/rtl_airband.cpp:222-251):
     * such handles stay sequential */
    h->pipeline = (cfg->flags & AIRBAND_HIP_FLAG_PIPELINE) && !any_afc;
    if (h->pipeline) {
        PREP_TRY(hipStreamCreateWithFlags(&h->front, hipStreamNonBlocking), AIRBAND_HIP_ENODEV);
        PREP_TRY(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
        PREP_TRY(hipEventCreateWithFlags(&h->ev_back, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
        for (auto& e : h->front_done) PREP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
    }
    for (auto& e : h->fork_ev) PREP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
    for (auto& st : h->side) PREP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_lo), AIRBAND_HIP_ENODEV);

    h->B = p.wave_batch;
    /* ring rows, whole 16-row tiles: one batch plus its AGC_EXTRA carry -- or two batches deep when stage 1 of the next batch
     * is written while stage 2 still reads this one */
    h->R = ((h->pipeline ? 2 : 1) * p.wave_batch + AB_AGC_EXTRA + 15) / 16 * 16; /* whole 16-hop MFMA tiles (a multiple of AB_TILE_ROWS too) */
    h->N = p.fft_size;
    /* demod slots: sort the channels by demod kind so that a 64-lane wavefront runs ONE code path (AM, NFM,
     * NFM+lowpass, NFM+CTCSS, everything else); kinds start on 64-slot block boundaries */
    std::vector<ChanConst> cc_slots;
    std::vector<ChanState> cs_slots;
    std::vector<uint8_t> block_kind;
    {
        auto kind_of = [](const ChanConst& c) {
            if (c.flags & AB_F_IQ_OUT) return (int)AB_KIND_GENERIC;
            const bool nfm = c.flags & AB_F_NFM, raw = c.flags & AB_F_RAW_IQ, lp = c.flags & AB_F_LOWPASS, ct = c.flags & AB_F_CTCSS;
            if (!nfm) return (!raw && !ct) ? (int)AB_KIND_AM : (int)AB_KIND_GENERIC;
            if (ct && lp) return (int)AB_KIND_GENERIC;
            return ct ? (int)AB_KIND_NFM_CTCSS : lp ? (int)AB_KIND_NFM_LOWPASS : (int)AB_KIND_NFM;
        };
        h->ext_to_slot.assign(p.total_ch, -1);
        ChanConst pad_c;
        ChanState pad_s;
        std::memset(&pad_c, 0, sizeof(pad_c));
        std::memset(&pad_s, 0, sizeof(pad_s));
        pad_c.ct_slot = -1;
        pad_s.axc = ' ';
        for (int k = 0; k < AB_KIND_COUNT; k++) {
            bool any = false;
            for (int e = 0; e < p.total_ch; e++) {
                if (kind_of(p.cc[e]) != k) continue;
                any = true;
                h->ext_to_slot[e] = (int)cc_slots.size();
                h->slot_to_ext.push_back(e);
                cc_slots.push_back(p.cc[e]);
                cs_slots.push_back(p.cs0[e]);
            }
            if (!any) continue;
            while (cc_slots.size() % AB_SLOT_BLOCK) {
                h->slot_to_ext.push_back(-1);
                cc_slots.push_back(pad_c);
                cs_slots.push_back(pad_s);
            }
            h->kind_first_block[k] = (int)block_kind.size();
            while (block_kind.size() < cc_slots.size() / AB_SLOT_BLOCK) block_kind.push_back((uint8_t)k);
            h->kind_n_blocks[k] = (int)block_kind.size() - h->kind_first_block[k];
        }
        h->n_slots = (int)cc_slots.size();
    }
    h->hop_bytes = 2LL * p.dev[0].bytes_per_sample * p.dev[0].hop_samples;
    h->first_batch_bytes = h->hop_bytes * (h->B + AB_AGC_EXTRA);
    h->batch_bytes = h->hop_bytes * h->B;
    h->lookahead_bytes = 2LL * p.dev[0].bytes_per_sample * p.fft_size - h->hop_bytes;
    if (h->lookahead_bytes < 0) h->lookahead_bytes = 0;
    /* hops that are not multiples of 16 bytes (2.4 MS/s): the channelizer stages whole 16-byte pieces, the piece holding the span's last
     * byte included -- make batch + look-ahead a whole number of pieces so that callers size (and fill) their spans accordingly */
    h->lookahead_bytes += (16 - (h->batch_bytes + h->lookahead_bytes) % 16) % 16;

    /* constants */
    PREP_TRY(upload(h->d_dev, p.dev), AIRBAND_HIP_ENOMEM);
    PREP_TRY(upload(h->d_cc, cc_slots), AIRBAND_HIP_ENOMEM);
    PREP_TRY(upload(h->d_cs, cs_slots), AIRBAND_HIP_ENOMEM);
    PREP_TRY(upload(h->d_slot_to_ext, h->slot_to_ext), AIRBAND_HIP_ENOMEM);
    PREP_TRY(upload(h->d_ext_to_slot, h->ext_to_slot), AIRBAND_HIP_ENOMEM);
    PREP_TRY(upload(h->d_block_kind, block_kind), AIRBAND_HIP_ENOMEM);
    PREP_TRY(upload(h->d_window, p.window), AIRBAND_HIP_ENOMEM);
    if (p.fft_size >= 1024) { /* row n2 = the window at samples n2, n2 + M, n2 + 2 M, ... (M = fft_size / 512): what transform n2 of a decimated FFT multiplies by, contiguous */
        const int M = p.fft_size / 512;
        std::vector<float> dec((size_t)p.fft_size);
        for (int n = 0; n < p.fft_size; n++) dec[(size_t)(n % M) * 512 + n / M] = p.window[n];
        PREP_TRY(upload(h->d_window_dec, dec), AIRBAND_HIP_ENOMEM);
    }
    PREP_TRY(upload(h->d_sin, p.sin_lut), AIRBAND_HIP_ENOMEM);
    PREP_TRY(upload(h->d_twiddle, p.twiddle), AIRBAND_HIP_ENOMEM);
    PREP_TRY(upload(h->d_cos, p.cos_lut), AIRBAND_HIP_ENOMEM);
    /* CTCSS tables: [ct_slot][detector][tone] coefficients and [ct_slot][detector][q1|q2][tone] Goertzel state, so
     * that the 52 tone lanes of demod phase 2 read one contiguous run */
    {
        const int n_ct = (int)p.tones.size();
        h->ct_stride = n_ct;
        std::vector<float> coeff((size_t)(n_ct > 0 ? n_ct : 1) * 2 * AB_MAX_TONES, 0.0f);
        for (int s = 0; s < n_ct; s++)
            for (int k = 0; k < 2; k++)
                for (int t = 0; t < p.tones[s].n[k]; t++) coeff[((size_t)s * 2 + k) * AB_MAX_TONES + t] = p.tones[s].coeff[k][t];
        PREP_TRY(upload(h->d_ct_coeff, coeff), AIRBAND_HIP_ENOMEM);
        PREP_TRY(h->d_ct_q.alloc((size_t)(n_ct > 0 ? n_ct : 1) * 4 * AB_MAX_TONES), AIRBAND_HIP_ENOMEM);
        PREP_TRY(hipMemset(h->d_ct_q.p, 0, h->d_ct_q.n * sizeof(float)), AIRBAND_HIP_ENOMEM);
    }
    /* rings, with the reference's config-time prefill of the lead-in (src/config.cpp:313-316) */
    const size_t ring = (size_t)h->R * h->n_slots; /* blocked: [n_slots/64][R][64] */
    PREP_TRY(h->d_mag.alloc(ring), AIRBAND_HIP_ENOMEM);
    PREP_TRY(h->d_iq.alloc(ring), AIRBAND_HIP_ENOMEM);
    PREP_TRY(h->d_iq_out.alloc((size_t)h->B * h->n_slots), AIRBAND_HIP_ENOMEM);
    PREP_TRY(h->d_sqbuf.alloc((size_t)AB_SQ_BUF * h->n_slots), AIRBAND_HIP_ENOMEM);
    {
        const float lead_in = 20.0f; /* wavein[0 .. AGC_EXTRA) = 20.0f, src/config.cpp:313-316 (as a 32-bit pattern: no host copy of the rings, 4 GB at 65 536 dongles) */
        int bits;
        std::memcpy(&bits, &lead_in, sizeof(bits));
        PREP_TRY(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(h->d_mag.p), bits, ring), AIRBAND_HIP_ENOMEM);
    }
    PREP_TRY(hipMemset(h->d_iq.p, 0, ring * sizeof(float2)), AIRBAND_HIP_ENOMEM);
    PREP_TRY(hipMemset(h->d_iq_out.p, 0, (size_t)h->B * h->n_slots * sizeof(float2)), AIRBAND_HIP_ENOMEM);
    PREP_TRY(hipMemset(h->d_sqbuf.p, 0, (size_t)AB_SQ_BUF * h->n_slots * sizeof(float)), AIRBAND_HIP_ENOMEM);
    /* hand-off buffers of the split kinds (NFM+CTCSS and generic are adjacent in slot order) */
    h->ct_n_blocks = h->kind_n_blocks[AB_KIND_NFM_CTCSS] + h->kind_n_blocks[AB_KIND_GENERIC];
    h->ct_first_block = h->kind_n_blocks[AB_KIND_NFM_CTCSS] ? h->kind_first_block[AB_KIND_NFM_CTCSS] : h->kind_first_block[AB_KIND_GENERIC];
    if (h->ct_n_blocks > 0) {
        /* one 32-bit word per sample for the NFM + CTCSS kind (rows padded to whole 128-byte lines), (audio, flags) pairs for the generic kind */
        h->ct_pk_pitch = (h->B + 31) / 32 * 32;
        PREP_TRY(h->d_ct_af.alloc((size_t)h->kind_n_blocks[AB_KIND_NFM_CTCSS] * AB_SLOT_BLOCK * h->ct_pk_pitch / 2 + (size_t)h->kind_n_blocks[AB_KIND_GENERIC] * AB_SLOT_BLOCK * h->B),
                 AIRBAND_HIP_ENOMEM);
        PREP_TRY(h->d_ct_mask.alloc((size_t)h->ct_n_blocks * (h->B / 50) * AB_SLOT_BLOCK), AIRBAND_HIP_ENOMEM);
    }
    {   /* regrouped stage 2 (demod.hip, "regrouping") */
        /* Default (neither flag, no environment override): by residency.  A regrouped workgroup's wavefronts of closed channels spend most of the batch waiting at the
         * lockstep barriers -- without using an issue slot, but holding their registers.  While ALL of a handle's lane-per-channel wavefronts are resident at once (about
         * four to six per SIMD) that costs nothing and the 22 % of vector instructions regrouping removes are time (32 768 dongles x 8 mixed channels: stage 2 3.42 ->
         * 3.02 ms; 49 152: 4.53 -> 4.33); past that the waiting wavefronts keep the next round's out (65 536: 5.8 -> 6.05), and a chip that is not full is bound by ONE
         * wavefront's dependent chain, which regrouping does not shorten (<= 16 384: 2.48 -> 2.54) -- profiles/r06_experiments.md D. */
        int n_lane_blocks = 0;
        for (int k = 0; k < AB_KIND_COUNT; k++) n_lane_blocks += h->kind_n_blocks[k];
        int cus = 0, cur_dev = 0;
        (void)hipGetDevice(&cur_dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cur_dev) != hipSuccess || cus <= 0) cus = 256;
        const double waves_per_simd = (double)n_lane_blocks / (4.0 * cus);
        const bool by_residency = waves_per_simd >= 2.75 && waves_per_simd <= 6.25;
        const char* e = getenv("AIRBAND_HIP_REGROUP");
        h->regroup = e && *e ? (*e != '0') : (h->flags & AIRBAND_HIP_FLAG_REGROUP) ? true : (h->flags & AIRBAND_HIP_FLAG_NO_REGROUP) ? false : by_residency;
        h->regroup_mode = (e && *e == '2') ? 2 : (e && *e == '3') ? 3 : 1;
        if (h->regroup && h->regroup_mode == 3) PREP_TRY(h->d_perm.alloc((size_t)h->n_slots), AIRBAND_HIP_ENOMEM);
        if (h->regroup || h->ct_n_blocks > 0) { /* the front kernel's note per channel: had audio / went CLOSED in this batch (tone kernel; regrouped back kernel) */
            PREP_TRY(h->d_sq_key.alloc((size_t)h->n_slots), AIRBAND_HIP_ENOMEM);
            PREP_TRY(hipMemset(h->d_sq_key.p, 0, (size_t)h->n_slots), AIRBAND_HIP_ENOMEM);
        }
    }
    if (h->flags & AIRBAND_HIP_FLAG_TRACE_SQUELCH) {
        PREP_TRY(h->d_trace.alloc((size_t)h->B * h->n_slots), AIRBAND_HIP_ENOMEM);
        PREP_TRY(hipMemset(h->d_trace.p, 0, (size_t)h->B * h->n_slots), AIRBAND_HIP_ENOMEM);
    }
    /* results */
    /* channel->waveout rows (src/rtl_airband.h:230): [AGC_EXTRA tail of the previous batch][WAVE_BATCH]; the consumer reads the first
     * WAVE_BATCH entries.  Config-time prefill of the lead-in as in src/config.cpp:313-316 (waveout[0..AGC_EXTRA) = 0.5). */
    h->wave_stride = (AB_OUT_PAD + AB_AGC_EXTRA + h->B + AB_OUT_RUN - 1) / AB_OUT_RUN * AB_OUT_RUN; /* whole 128-byte lines per row */
    {
        PREP_TRY(h->d_out_wave.alloc((size_t)p.total_ch * h->wave_stride), AIRBAND_HIP_ENOMEM);
        PREP_TRY(hipMemset(h->d_out_wave.p, 0, h->d_out_wave.n * sizeof(float)), AIRBAND_HIP_ENOMEM);
        /* the lead-in columns, a few thousand rows per strided copy (not a host image of every row: 4.5 GB at 65 536 dongles) */
        const int chunk = p.total_ch < 4096 ? p.total_ch : 4096;
        const std::vector<float> lead((size_t)chunk * AB_AGC_EXTRA, 0.5f);
        for (int c = 0; c < p.total_ch; c += chunk) {
            const int rows = p.total_ch - c < chunk ? p.total_ch - c : chunk;
            PREP_TRY(hipMemcpy2D(h->d_out_wave.p + (size_t)c * h->wave_stride + AB_OUT_PAD, (size_t)h->wave_stride * sizeof(float), lead.data(), AB_AGC_EXTRA * sizeof(float),
                                 AB_AGC_EXTRA * sizeof(float), (size_t)rows, hipMemcpyHostToDevice),
                     AIRBAND_HIP_ENOMEM);
        }
    }
    PREP_TRY(h->d_out_axc.alloc((size_t)p.total_ch), AIRBAND_HIP_ENOMEM);
    bool any_iq_out = false;
    for (const ChanConst& c : p.cc) any_iq_out |= (c.flags & AB_F_IQ_OUT) != 0;
    if (any_iq_out) {
        PREP_TRY(h->d_out_iq.alloc((size_t)p.total_ch * h->B * 2), AIRBAND_HIP_ENOMEM);
        PREP_TRY(hipMemset(h->d_out_iq.p, 0, h->d_out_iq.n * sizeof(float)), AIRBAND_HIP_ENOMEM);
    }
    /* channelizer variant: matrix-core pruned DFT when the configuration qualifies, wavefront FFT otherwise */
    /* AFC moves bins at run time and needs the full spectrum of each batch's last hop: that is the FFT kernel's job */
    h->any_afc = any_afc;
    if (any_afc) PREP_TRY(h->d_spectrum.alloc((size_t)p.n_dev * p.fft_size * 2), AIRBAND_HIP_ENOMEM);
    h->use_dft = !(h->flags & AIRBAND_HIP_FLAG_FORCE_FFT) && dft_supported(p.fft_size, (int)h->hop_bytes, p.dev[0].sfmt, p.max_ch);
    if (h->use_dft) {
        build_dft_tables(h->plan, false);
        /* One table per DISTINCT group of eight bins (shared between the work items that have it: a fleet of identical dongles has one, a fleet in which every
         * device_t derives its own bins -- src/config.cpp:666-667 -- as many as it has groups) plus one private table per group with an AFC channel.  Bounded by
         * their bytes alone (65 536 tables are 3.2 GB at fft 512, ~51 GB at fft 8192); past the budget the handle runs on the wavefront FFT.  Round 6: a COUNT used to
         * stand here (more than 4 096 distinct plans -> wavefront FFT, 7x slower), which contradicted the private tables of the AFC path right beside it. */
        const size_t tab_bytes_each = (size_t)3 * (p.fft_size > 512 ? 16 : p.fft_size / 32) * 64 * 16 * (p.fft_size > 512 ? p.fft_size / 512 : 1);
        const size_t table_bytes = (size_t)p.n_bsets * tab_bytes_each;
        if (table_bytes > AB_PRIVATE_TABLE_BUDGET) {
            h->use_dft = false;
        } else {
            PREP_TRY(upload(h->d_item_dev, p.item_dev), AIRBAND_HIP_ENOMEM);
            PREP_TRY(upload(h->d_item_group, p.item_group), AIRBAND_HIP_ENOMEM);
            PREP_TRY(upload(h->d_item_bset, p.item_home), AIRBAND_HIP_ENOMEM); /* every channel starts on its base bin */
            PREP_TRY(upload(h->d_item_private, p.item_bset), AIRBAND_HIP_ENOMEM);
            PREP_TRY(upload(h->d_item_home, p.item_home), AIRBAND_HIP_ENOMEM);
            /* the host has built the shared tables; the private ones (groups with an AFC channel) follow them, zeroed, and are built by the
             * re-tune kernel right here: every column of theirs still stands at bin -1 */
            const int np_t = p.fft_size > 512 ? p.fft_size / 512 : 1;
            const size_t tab_bytes = (size_t)3 * (p.fft_size > 512 ? 16 : p.fft_size / 32) * 64 * 16 * np_t;
            PREP_TRY(h->d_bfrag.alloc((size_t)p.n_bsets * tab_bytes), AIRBAND_HIP_ENOMEM);
            PREP_TRY(h->d_bcorr.alloc((size_t)p.n_bsets * np_t * 16), AIRBAND_HIP_ENOMEM);
            PREP_TRY(hipMemset(h->d_bfrag.p, 0, h->d_bfrag.n), AIRBAND_HIP_ENOMEM);
            PREP_TRY(hipMemset(h->d_bcorr.p, 0, h->d_bcorr.n * sizeof(double)), AIRBAND_HIP_ENOMEM);
            if (!p.bfrag.empty()) PREP_TRY(hipMemcpy(h->d_bfrag.p, p.bfrag.data(), p.bfrag.size(), hipMemcpyHostToDevice), AIRBAND_HIP_ENOMEM);
            if (!p.bcorr.empty()) PREP_TRY(hipMemcpy(h->d_bcorr.p, p.bcorr.data(), p.bcorr.size() * sizeof(double), hipMemcpyHostToDevice), AIRBAND_HIP_ENOMEM);
            {
                std::vector<int> with_epoch(p.bset_bins);
                with_epoch.push_back(0); /* the "last moved in batch" stamp: 0 = the start-up build below */
                PREP_TRY(upload(h->d_bset_bin, with_epoch), AIRBAND_HIP_ENOMEM);
            }
            /* shared tables the host did not build (fleets with more than a few thousand distinct channel plans), then the private ones: a private table's columns
             * are copied from its home table, so the home tables come first */
            launch_build_tables(h->d_bfrag.p, h->d_bcorr.p, h->d_window.p, h->d_bset_bin.p, p.n_host_bsets, p.n_shared_bsets - p.n_host_bsets, p.fft_size, h->stream);
            if (p.n_bsets > p.n_shared_bsets) launch_retune_tables(h, h->stream, 0);
            if (p.n_bsets > p.n_host_bsets) {
                PREP_TRY(hipGetLastError(), AIRBAND_HIP_ENODEV);
                PREP_TRY(hipStreamSynchronize(h->stream), AIRBAND_HIP_ENODEV);
            }
            if (p.fft_size > 4096) /* [work items][tiles][64 lanes] float4 */
                PREP_TRY(h->d_dft_partial.alloc((size_t)p.item_dev.size() * dft_partial_tiles(h->B + AB_AGC_EXTRA) * 64 * 4), AIRBAND_HIP_ENOMEM);
        }
    }
    /* CF32 (SoapySDR): the float32 matrix pipe.  Round 6: also with AFC channels -- a group with one owns a private float table whose column pairs the re-tune kernel
     * moves (misc_kernels.hip, build_column_pair_f32), exactly as the int8 path does; such handles stayed on the wavefront FFT before. */
    h->use_f32 = !h->use_dft && !(h->flags & AIRBAND_HIP_FLAG_FORCE_FFT) && f32_supported(p.fft_size, p.dev[0].hop_samples, p.dev[0].sfmt);
    if (h->use_f32) {
        build_dft_tables(h->plan, false); /* the work items and the shared bin sets (its int8 tables are not used) */
        /* bytes, not a count: a float table is f32_nw x (2 N / 4 / f32_nw) x 64 lanes x 4 bytes = 128 N bytes -- 64 KiB at fft 512, 256 KiB at 2048.  The shared tables are
         * built on the host (params.cpp, build_f32_tables) and read once per work item per launch: past a budget the handle runs on the wavefront FFT rather than on a
         * gigabyte-sized host build; the private ones (device-built) count against the budget the int8 path's private tables have */
        const size_t ftab_each = 128u * (size_t)p.fft_size;
        const size_t ftab_bytes = (size_t)p.n_shared_bsets * ftab_each;
        if (ftab_bytes > AB_F32_TABLE_BUDGET || (size_t)p.n_bsets * ftab_each > AB_PRIVATE_TABLE_BUDGET) {
            h->use_f32 = false;
        } else {
            build_f32_tables(h->plan);
            PREP_TRY(upload(h->d_item_dev, p.item_dev), AIRBAND_HIP_ENOMEM);
            PREP_TRY(upload(h->d_item_group, p.item_group), AIRBAND_HIP_ENOMEM);
            PREP_TRY(upload(h->d_item_bset, p.item_home), AIRBAND_HIP_ENOMEM);
            if (p.n_bsets > p.n_shared_bsets) { /* groups with an AFC channel: their private tables follow the shared ones, built by the re-tune kernel right here */
                PREP_TRY(upload(h->d_item_private, p.item_bset), AIRBAND_HIP_ENOMEM);
                PREP_TRY(upload(h->d_item_home, p.item_home), AIRBAND_HIP_ENOMEM);
                std::vector<int> with_epoch(p.bset_bins);
                with_epoch.push_back(0);
                PREP_TRY(upload(h->d_bset_bin, with_epoch), AIRBAND_HIP_ENOMEM);
                PREP_TRY(h->d_ftab.alloc((size_t)p.n_bsets * ftab_each / sizeof(float)), AIRBAND_HIP_ENOMEM);
                PREP_TRY(hipMemset(h->d_ftab.p, 0, (size_t)p.n_bsets * ftab_each), AIRBAND_HIP_ENOMEM);
                PREP_TRY(hipMemcpy(h->d_ftab.p, p.ftab.data(), p.ftab.size() * sizeof(float), hipMemcpyHostToDevice), AIRBAND_HIP_ENOMEM);
                launch_retune_tables(h, h->stream, 0);
                PREP_TRY(hipGetLastError(), AIRBAND_HIP_ENODEV);
                PREP_TRY(hipStreamSynchronize(h->stream), AIRBAND_HIP_ENODEV);
            } else
            PREP_TRY(upload(h->d_ftab, p.ftab), AIRBAND_HIP_ENOMEM);
            /* fft_size 4096 / 8192: partial sums between the launches of the window's segments (channelizer_f32.hip) */
            if (f32_n_seg(p.fft_size) > 1) PREP_TRY(h->d_dft_partial.alloc((size_t)p.item_dev.size() * f32_partial_tiles(h->B + AB_AGC_EXTRA) * 64 * 4), AIRBAND_HIP_ENOMEM);
        }
        h->plan.bfrag.clear(); h->plan.bfrag.shrink_to_fit();
        h->plan.ftab.clear(); h->plan.ftab.shrink_to_fit();
    }
    if (!h->use_dft && !h->use_f32) {
        const size_t lds = fft_lds_bytes(p.fft_log, p.dev[0].hop_samples, p.dev[0].bytes_per_sample);
        if (lds > 160 * 1024) { /* e.g. F32 at 20 MS/s: a 16-hop tile of raw samples does not fit a CU's LDS */
            g_prepare_error = "sample_rate x bytes_per_sample too large for the FFT channelizer's LDS tile (" + std::to_string(lds) + " > 163840 bytes)";
            destroy(h);
            return AIRBAND_HIP_EBADSIZE;
        }
    }
    h->ring_wr.reset(new std::atomic<uint64_t>[p.n_dev]);
    for (int d = 0; d < p.n_dev; d++) h->ring_wr[d].store(0);
    h->dev_enabled.reset(new std::atomic<uint8_t>[p.n_dev]);
    for (int d = 0; d < p.n_dev; d++) h->dev_enabled[d].store(1);
    h->n_enabled = p.n_dev;
    h->cc_slots = cc_slots;
#undef PREP_TRY
    *out = h;
    return AIRBAND_HIP_OK;
}

void airband_hip_release(airband_hip_handle* h) { destroy(h); }

int airband_hip_get_geometry(const airband_hip_handle* h, airband_hip_geometry* g) {
    if (!h || !g) return AIRBAND_HIP_EINVAL;
    g->fft_size = h->N;
    g->wave_rate = h->plan.wave_rate;
    g->wave_batch = h->B;
    g->device_count = h->plan.n_dev;
    g->total_channels = h->plan.total_ch;
    g->max_channels = h->plan.max_ch;
    g->mixer_count = h->n_mixers;
    g->wave_stride = h->wave_stride;
    g->first_batch_bytes = h->first_batch_bytes;
    g->batch_bytes = h->batch_bytes;
    g->lookahead_bytes = h->lookahead_bytes;
    return AIRBAND_HIP_OK;
}

int airband_hip_set_mixers(airband_hip_handle* h, int32_t mixer_count, const airband_hip_mixer_input* in, int32_t n_in) {
    if (!h || mixer_count < 1 || n_in < 0 || (!in && n_in > 0)) return fail(h, AIRBAND_HIP_EINVAL, "bad mixer arguments");
    /* n_in == 0: a handle whose dongles feed no mixer still takes part in the exchange with all-zero partial sums */
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    const Plan& p = h->plan;
    std::vector<int> first(mixer_count + 1, 0), chan(n_in);
    std::vector<float> ml(n_in), mr(n_in);
    std::vector<uint8_t> stereo(mixer_count, 0);
    for (int i = 0; i < n_in; i++) {
        if (in[i].mixer < 0 || in[i].mixer >= mixer_count || in[i].device < 0 || in[i].device >= p.n_dev || in[i].channel < 0 ||
            in[i].channel >= p.dev[in[i].device].n_ch)
            return fail(h, AIRBAND_HIP_EINVAL, "mixer input out of range");
        first[in[i].mixer + 1]++;
        if (in[i].balance != 0.0f) stereo[in[i].mixer] = 1; /* src/mixer.cpp:84-85 */
    }
    for (int m = 0; m < mixer_count; m++) first[m + 1] += first[m];
    std::vector<int> cur(first.begin(), first.end() - 1);
    std::vector<int> pos(n_in, 0);
    for (int i = 0; i < n_in; i++) { /* stable: connection order inside a mixer is kept (summation order) */
        const int k = cur[in[i].mixer]++;
        pos[i] = k;
        chan[k] = p.chan_base[in[i].device] + in[i].channel;
        ml[k] = in[i].ampfactor * fminf(1.0f, 1.0f - in[i].balance); /* src/mixer.cpp:82-83,203-208 */
        mr[k] = in[i].ampfactor * fminf(1.0f, 1.0f + in[i].balance);
    }
    std::vector<int> run_first, run_mixer, first_run(mixer_count + 1, 0);
    for (int m = 0; m < mixer_count; m++) {
        first_run[m] = (int)run_mixer.size();
        for (int i = first[m]; i < first[m + 1]; i += AB_MIX_RUN) {
            run_first.push_back(i);
            run_mixer.push_back(m);
        }
    }
    first_run[mixer_count] = (int)run_mixer.size();
    run_first.push_back(n_in);
    const int n_runs = (int)run_mixer.size();
    /* the mixer kernels index runs / mixers with blockIdx.y: say so here rather than fail at the first launch */
    if (n_runs > 65535 || mixer_count > 65535) return fail(h, AIRBAND_HIP_EBADSIZE, "more than 65 535 mixers (or runs of 64 mixer inputs) on one handle");
    order_behind_last_batch(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream), AIRBAND_HIP_ERUNTIME); /* a batch under way still sums the old wiring */
    h->n_mixers = 0; /* until the new wiring is complete: a failure below leaves a handle without mixers, not one with freed tables */
    h->n_mix_runs = 0;
    h->d_mix_chan.release(); h->d_mix_first.release(); h->d_mix_ml.release(); h->d_mix_mr.release();
    h->d_mix_left.release(); h->d_mix_right.release(); h->d_mix_stereo.release(); h->d_mix_signal.release();
    h->d_mix_run_first.release(); h->d_mix_run_mixer.release(); h->d_mix_first_run.release();
    h->d_mix_run_left.release(); h->d_mix_run_right.release(); h->d_mix_run_signal.release();
    HIP_TRY(h, upload(h->d_mix_run_first, run_first), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, upload(h->d_mix_run_mixer, run_mixer), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, upload(h->d_mix_first_run, first_run), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, h->d_mix_run_left.alloc((size_t)n_runs * h->B), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, h->d_mix_run_right.alloc((size_t)n_runs * h->B), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, h->d_mix_run_signal.alloc((size_t)n_runs), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, upload(h->d_mix_chan, chan), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, upload(h->d_mix_first, first), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, upload(h->d_mix_ml, ml), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, upload(h->d_mix_mr, mr), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, upload(h->d_mix_stereo, stereo), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, h->d_mix_left.alloc((size_t)mixer_count * h->B), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, h->d_mix_right.alloc((size_t)mixer_count * h->B), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, h->d_mix_signal.alloc((size_t)mixer_count), AIRBAND_HIP_ENOMEM);
    h->mix_pos = pos;
    h->mix_chan_host = chan;
    h->mix_user_on.assign(n_in, 1);
    for (int k = 0; k < n_in; k++) /* inputs of dongles that are already switched off stay out */
        if (!h->dev_enabled[p.cc[chan[k]].dev]) HIP_TRY(h, write_mix_input(h, k), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamSynchronize(h->stream), AIRBAND_HIP_ERUNTIME);
    h->n_mix_runs = n_runs;
    h->n_mixers = mixer_count;
    return AIRBAND_HIP_OK;
}

int airband_hip_mixer_enable_input(airband_hip_handle* h, int32_t input_index, int32_t enabled) {
    if (!h || h->n_mixers <= 0) return fail(h, AIRBAND_HIP_EINVAL, "no mixers configured");
    if (input_index < 0 || input_index >= (int32_t)h->mix_pos.size()) return fail(h, AIRBAND_HIP_EINVAL, "mixer input index out of range");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    const int k = h->mix_pos[input_index];
    h->mix_user_on[k] = enabled ? 1 : 0;
    HIP_TRY(h, write_mix_input(h, k), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamSynchronize(h->stream), AIRBAND_HIP_ERUNTIME);
    return AIRBAND_HIP_OK;
}

int airband_hip_device_enable(airband_hip_handle* h, int32_t dev, int32_t enabled) {
    if (!h) return AIRBAND_HIP_EINVAL;
    const Plan& p = h->plan;
    if (dev < 0 || dev >= p.n_dev) return fail(h, AIRBAND_HIP_EINVAL, "device index out of range");
    const uint8_t on = enabled ? 1 : 0;
    if (h->dev_enabled[dev].load() == on) return AIRBAND_HIP_OK;
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    hipStream_t s = h->stream;
    order_behind_last_batch(h);
    /* everything below is ordered on the handle's stream: the batches already enqueued still see the old state, the next one the new */
    h->plan.dev[dev].disabled = on ? 0 : 1;
    HIP_TRY(h, hipMemcpyAsync(&h->d_dev.p[dev], &h->plan.dev[dev], sizeof(DevConst), hipMemcpyHostToDevice, s), AIRBAND_HIP_ERUNTIME);
    const int c0 = p.chan_base[dev], nc = p.dev[dev].n_ch;
    std::vector<uint8_t> blank((size_t)nc, (uint8_t)' ');
    for (int c = c0; c < c0 + nc; c++) { /* the demod kernels treat a slot without AB_F_VALID like padding: the lane leaves at once, its state stays as it is */
        const int slot = h->ext_to_slot[c];
        ChanConst& cc = h->cc_slots[slot];
        cc.flags = on ? (cc.flags | AB_F_VALID) : (cc.flags & ~AB_F_VALID);
        HIP_TRY(h, hipMemcpyAsync(&h->d_cc.p[slot].flags, &cc.flags, sizeof(cc.flags), hipMemcpyHostToDevice, s), AIRBAND_HIP_ERUNTIME);
    }
    /* channel->axcindicate of a device that is not demodulated any more: NO_SIGNAL */
    if (!on) HIP_TRY(h, hipMemcpyAsync(h->d_out_axc.p + c0, blank.data(), (size_t)nc, hipMemcpyHostToDevice, s), AIRBAND_HIP_ERUNTIME);
    /* host-ring path: a dongle that comes back joins the others at the common stream position with an empty queue -- its write cursor is put there
     * BEFORE the dongle is published as enabled (release / acquire with submit()'s load): a feeder thread that sees it enabled never sees the stale cursor */
    if (on && h->ring_wr) h->ring_wr[dev].store(h->ring_rd, std::memory_order_release);
    h->dev_enabled[dev].store(on, std::memory_order_release);
    h->n_enabled += on ? 1 : -1;
    /* its mixer connections: mixer_disable_input() for every output of the device, as disable_device_outputs() does (src/output.cpp, src/mixer.cpp:96-110) */
    for (size_t k = 0; k < h->mix_chan_host.size(); k++)
        if (p.cc[h->mix_chan_host[k]].dev == dev) HIP_TRY(h, write_mix_input(h, (int)k), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamSynchronize(s), AIRBAND_HIP_ERUNTIME); /* the host buffers above go out of scope */
    return AIRBAND_HIP_OK;
}

int airband_hip_gpu_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* stage 1 of the next batch (index front_batches, ring rows from row0_front) on stream s */
static int launch_front(airband_hip_handle* h, const void* d_iq, size_t stride_bytes, hipStream_t s) {
    const Plan& p = h->plan;
    const bool first = h->front_batches == 0;
    hipEvent_t* ev = event_set(h, h->front_batches, 0);
    /* hipGetLastError() is sticky: whatever an earlier, unchecked call of this thread left behind (ours or the host application's) is not this launch's error */
    (void)hipGetLastError();
    hipError_t launch_err = hipSuccess;
    if (h->use_f32) {
        F32Args a;
        a.iq = (const uint8_t*)d_iq;
        a.iq_stride = (long)stride_bytes;
        a.dev = h->d_dev.p;
        a.cc = h->d_cc.p;
        a.ext_to_slot = h->d_ext_to_slot.p;
        a.item_dev = h->d_item_dev.p;
        a.item_group = h->d_item_group.p;
        a.item_bset = h->d_item_bset.p;
        a.btab = h->d_ftab.p;
        a.mag = h->d_mag.p;
        a.iq_bins = h->d_iq.p;
        a.n_items = (int)p.item_dev.size();
        a.fft_size = p.fft_size;
        a.hop_bytes = (int)h->hop_bytes;
        a.pad = f32_pad_bytes(p.dev[0].hop_samples);
        a.lds_per_buf = f32_lds_per_buf(p.fft_size, p.dev[0].hop_samples);
        a.seg = 0;
        a.n_seg = 1;
        a.partial = reinterpret_cast<float4*>(h->d_dft_partial.p);
        a.row0 = h->row0_front;
        a.ring_rows = h->R;
        a.first_row = first ? 0 : AB_AGC_EXTRA;
        a.n_hops = first ? h->B + AB_AGC_EXTRA : h->B;
        /* enough workgroups to fill 256 CUs x 2 even with few dongles: split each dongle's tiles */
        const int tiles = (a.n_hops + 15) / 16 + 1;
        int splits = (2048 + a.n_items - 1) / a.n_items;
        if (splits > tiles / 4) splits = tiles / 4;
        if (splits < 1) splits = 1;
        a.splits = splits;
        h->last_iq = d_iq;
        h->last_iq_stride = stride_bytes;
        h->last_n_hops = a.n_hops;
        h->afc_spectrum_valid = h->any_afc;
        if (h->any_afc) { /* the spectrum of the batch's last hop, on a side stream beside stage 1 (as on the int8 path below) */
            for (auto& e : h->ev_spec)
                if (!e) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
            (void)hipEventRecord(h->ev_spec[0], s);
            (void)hipStreamWaitEvent(h->side[0], h->ev_spec[0], 0);
            launch_last_hop_spectrum(h, h->side[0]);
            (void)hipEventRecord(h->ev_spec[1], h->side[0]);
        }
        (void)hipEventRecord(ev[0], s);
        launch_channelizer_f32(a, s);
        launch_err = hipGetLastError(); /* right behind the launch: the event record below would mask it (or be blamed for it) */
        (void)hipEventRecord(ev[1], s);
    } else if (h->use_dft) {
        DftArgs a;
        a.iq = (const uint8_t*)d_iq;
        a.iq_stride = (long)stride_bytes;
        a.dev = h->d_dev.p;
        a.cc = h->d_cc.p;
        a.ext_to_slot = h->d_ext_to_slot.p;
        a.item_dev = h->d_item_dev.p;
        a.item_group = h->d_item_group.p;
        a.item_bset = h->d_item_bset.p;
        a.bfrag = h->d_bfrag.p;
        a.corr = h->d_bcorr.p;
        /* table units -> sample units: u8 (b - 127.5) / 127.5; s8 i / 128; CS16 x / fullscale (the kernel multiplies by the dongle's 1 / fullscale) */
        a.unscale = p.dev[0].sfmt == AIRBAND_SFMT_S16 ? p.b_unscale * 127.5 : p.dev[0].sfmt == AIRBAND_SFMT_S8 ? p.b_unscale * 127.5 / 128.0 : p.b_unscale;
        a.sfmt = p.dev[0].sfmt;
        a.edge_hi_zero = p.b_edge_hi_zero ? 1 : 0;
        a.mag = h->d_mag.p;
        a.iq_bins = h->d_iq.p;
        a.n_dev = p.n_dev;
        a.n_items = (int)p.item_dev.size();
        a.fft_size = p.fft_size;
        a.hop_bytes = (int)h->hop_bytes;
        /* the whole window is staged, also when it is worked on in pieces of 512 samples -- up to eight of them per launch (fft_size 8192: two passes) */
        const int np_total = p.fft_size > 512 ? p.fft_size / 512 : 1, np = np_total > 8 ? 8 : np_total;
        const int win_bytes = 2 * p.fft_size * p.dev[0].bytes_per_sample / np_total * np;
        a.partial = reinterpret_cast<float4*>(h->d_dft_partial.p);
        a.lds_per_buf = dft_lds_per_buf((int)h->hop_bytes, win_bytes, np);
        a.nbuf = dft_nbuf((int)h->hop_bytes, win_bytes, np);
        a.sub = dft_sub((int)h->hop_bytes, win_bytes, np);
        a.row0 = h->row0_front;
        a.ring_rows = h->R;
        a.first_row = first ? 0 : AB_AGC_EXTRA;
        a.n_hops = first ? h->B + AB_AGC_EXTRA : h->B;
        /* Pipelined handles (stage 1 of this batch runs beside stage 2 of the batch before): eight channelizer wavefronts of ~250 registers ARE a CU's register file, and
         * stage-2 wavefronts then only get onto a CU when one of them retires.  Held to FIVE per CU (it loses ~5 % alone: 7 and 6 per CU cost nothing, 4 cost 12 %,
         * profiles/r06_occupancy/) the channelizer leaves three SIMDs a wavefront's worth of registers each: configs[2] 14.05 ms sequential, 13.77 pipelined as before,
         * 13.05 like this (13.5 / 14.0 at 4 / 6 per CU; profiles/r06_pipelined/).  The LDS it asks for and never touches is what holds it there. */
        a.extra_lds = 0;
        if (h->pipeline && np_total == 1) {
            const int used = a.nbuf * a.lds_per_buf, want = 28 * 1024; /* 160 KiB / 28 KiB = 5 */
            if (used < want) a.extra_lds = want - used;
        }
        /* enough waves to fill 256 CUs x 8 waves even with few dongles: split each dongle's tiles */
        const int steps = ((a.n_hops + 15) / 16 + 1 + a.sub - 1) / a.sub;
        int splits = (8192 + a.n_items - 1) / a.n_items;
        if (splits > steps / 4) splits = steps / 4;
        if (splits < 1) splits = 1;
        a.splits = splits;
        h->last_iq = d_iq;
        h->last_iq_stride = stride_bytes;
        h->last_n_hops = a.n_hops;
        h->afc_spectrum_valid = h->any_afc;
        if (h->any_afc) {
            /* AFC::finalize looks at the spectrum of the batch's last hop (src/rtl_airband.cpp:626-630): it depends on the input only, so it is
             * computed on a side stream BESIDE stage 1 (behind the previous batch's AFC, which read the buffer it overwrites) and joined in
             * front of afc_kernel (run_back_half) */
            for (auto& e : h->ev_spec)
                if (!e) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
            (void)hipEventRecord(h->ev_spec[0], s);
            (void)hipStreamWaitEvent(h->side[0], h->ev_spec[0], 0);
            launch_last_hop_spectrum(h, h->side[0]);
            (void)hipEventRecord(h->ev_spec[1], h->side[0]);
        }
        (void)hipEventRecord(ev[0], s);
        launch_channelizer_dft(a, s);
        launch_err = hipGetLastError();
        (void)hipEventRecord(ev[1], s);
    } else {
        ChannelizerArgs ca;
        ca.iq = (const uint8_t*)d_iq;
        ca.iq_stride = (long)stride_bytes;
        ca.dev = h->d_dev.p;
        ca.cs = h->d_cs.p;
        ca.cc = h->d_cc.p;
        ca.ext_to_slot = h->d_ext_to_slot.p;
        ca.window = h->d_window.p;
        ca.window_dec = h->d_window_dec.p;
        ca.twiddle = reinterpret_cast<const float2*>(h->d_twiddle.p);
        ca.mag = h->d_mag.p;
        ca.iq_bins = h->d_iq.p;
        ca.last_spectrum = h->any_afc ? h->d_spectrum.p : nullptr;
        ca.n_dev = p.n_dev;
        ca.fft_log = p.fft_log;
        ca.hop_samples = p.dev[0].hop_samples;
        ca.bytes_per_sample = p.dev[0].bytes_per_sample;
        ca.sfmt = p.dev[0].sfmt;
        ca.scale = p.dev[0].scale;
        ca.row0 = h->row0_front;
        ca.ring_rows = h->R;
        ca.first_row = first ? 0 : AB_AGC_EXTRA; /* the first batch also produces the AGC_EXTRA lead-in hops (waveend starts at 0, src/config.cpp:805) */
        ca.n_hops = first ? h->B + AB_AGC_EXTRA : h->B;
        ca.max_ch = p.max_ch;
        ca.spectrum_only = 0;
        (void)hipEventRecord(ev[0], s);
        h->afc_spectrum_valid = h->any_afc;
        launch_channelizer_fft(ca, s);
        launch_err = hipGetLastError();
        (void)hipEventRecord(ev[1], s);
    }
    /* a refused launch (an LDS opt-in that failed, a bad grid) is this call's error, not a puzzle for whoever synchronises next */
    if (launch_err != hipSuccess) return fail(h, AIRBAND_HIP_ERUNTIME, std::string("channelizer launch: ") + hipGetErrorString(launch_err));
    h->row0_front = (h->row0_front + h->B) % h->R;
    h->front_batches++;
    return AIRBAND_HIP_OK;
}

int airband_hip_process_device(airband_hip_handle* h, const void* d_iq, size_t stride_bytes, void* stream) {
    if (!h || !d_iq) return fail(h, AIRBAND_HIP_EINVAL, "NULL argument");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    /* hops of whole 16-byte pieces: the channelizer's transfers address the span directly, so it must start on one.  Any other hop (300, 250 bytes ...):
     * a batch of such hops cannot start on 16 bytes every time anyway (2 100 hops of 250 bytes = 525 000), the kernel stages from the aligned byte in
     * front of the span and only whole samples are asked for */
    /* ... at the alignment of the fragment reads the kernel variant for this hop uses (channelizer_dft.hip, launch_generic: the largest power of two up to 16
     * that divides the hop -- 300 bytes: 4, 600: 8, 250: 2): the staged bytes keep the span's offset from 16 bytes, and a 4- or 8-byte LDS read must not land
     * on a 2-byte boundary.  Batch offsets are multiples of the hop, so a stream that starts aligned stays aligned. */
    uintptr_t al = 16;
    while (al > 2 && (h->hop_bytes % (int64_t)al) != 0) al >>= 1;
    if (al < (uintptr_t)(2 * h->plan.dev[0].bytes_per_sample)) al = (uintptr_t)(2 * h->plan.dev[0].bytes_per_sample);
    const uintptr_t need = al - 1;
    if (h->use_f32 && ((((uintptr_t)d_iq) | (uintptr_t)stride_bytes) & 15))
        return fail(h, AIRBAND_HIP_EINVAL, "d_iq and stride_bytes must be multiples of 16 (the channelizer fetches 16 bytes per lane)");
    if (h->use_dft && ((((uintptr_t)d_iq) | (uintptr_t)stride_bytes) & need))
        return fail(h, AIRBAND_HIP_EINVAL, (h->hop_bytes % 16) == 0 ? "d_iq and stride_bytes must be multiples of 16 (the channelizer fetches 16 bytes per lane)"
                                                                    : "d_iq and stride_bytes must be multiples of the largest power of two (up to 16) that divides the hop's bytes");
    if (!h->pipeline) {
        hipStream_t s = stream ? (hipStream_t)stream : h->stream;
        h->last_stream = s;
        const int rc_front = launch_front(h, d_iq, stride_bytes, s);
        if (rc_front != AIRBAND_HIP_OK) return rc_front;
        const int rc = run_back_half(h, s);
        if (s != h->stream) { /* collect() and friends run on h->stream: give them something to wait for */
            if (!h->ev_last) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
            HIP_TRY(h, hipEventRecord(h->ev_last, s), AIRBAND_HIP_ERUNTIME);
            h->ev_last_pending = true;
        } else {
            h->ev_last_pending = false;
        }
        return rc;
    }
    /* Pipelined: stage 1 of this batch (k) goes on the front stream and runs beside stage 2 of batch k-1, which is enqueued
     * right after it on the handle's stream.  Stage 1 (k) overwrites the ring rows stage 2 (k-2) read, and may use the
     * caller's input as soon as the caller's stream (or everything enqueued on the handle so far) has produced it. */
    hipStream_t in = stream ? (hipStream_t)stream : h->stream;
    HIP_TRY(h, hipEventRecord(h->ev_in, in), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamWaitEvent(h->front, h->ev_in, 0), AIRBAND_HIP_ERUNTIME);
    if (in != h->stream) {
        /* a caller stream orders BOTH halves: its earlier work (producing this input, consuming the previous results) is done
         * before stage 1 reads and before stage 2 overwrites the result buffers */
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_in, 0), AIRBAND_HIP_ERUNTIME);
        HIP_TRY(h, hipEventRecord(h->ev_back, h->stream), AIRBAND_HIP_ERUNTIME);
        HIP_TRY(h, hipStreamWaitEvent(h->front, h->ev_back, 0), AIRBAND_HIP_ERUNTIME);
    }
    const uint64_t k = h->front_batches;
    const int rc_front = launch_front(h, d_iq, stride_bytes, h->front);
    if (rc_front != AIRBAND_HIP_OK) return rc_front;
    HIP_TRY(h, hipEventRecord(h->front_done[k & 1], h->front), AIRBAND_HIP_ERUNTIME);
    /* nothing to demodulate yet -- the very first call, or the first call after airband_hip_flush() drained the pipeline:
     * the results of this batch appear with the next call (or flush) */
    if (h->batches_done == k) return AIRBAND_HIP_OK;
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->front_done[(k - 1) & 1], 0), AIRBAND_HIP_ERUNTIME);
    return run_back_half(h, h->stream);
}

int airband_hip_stream_wait_results(airband_hip_handle* h, void* stream) {
    if (!h || !stream) return fail(h, AIRBAND_HIP_EINVAL, "NULL argument");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    if (!h->ev_wait) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_wait, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
    hipStream_t res = h->pipeline ? h->stream : (h->last_stream ? h->last_stream : h->stream);
    if (res == (hipStream_t)stream) return AIRBAND_HIP_OK;
    HIP_TRY(h, hipEventRecord(h->ev_wait, res), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamWaitEvent((hipStream_t)stream, h->ev_wait, 0), AIRBAND_HIP_ERUNTIME);
    return AIRBAND_HIP_OK;
}

int airband_hip_flush(airband_hip_handle* h) {
    if (!h) return AIRBAND_HIP_EINVAL;
    if (!h->pipeline || h->front_batches == h->batches_done) return AIRBAND_HIP_OK;
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->front_done[(h->front_batches - 1) & 1], 0), AIRBAND_HIP_ERUNTIME);
    return run_back_half(h, h->stream);
}

/* first use of the host-ring path: pinned rings, device staging, copy stream */
static int host_path_init(airband_hip_handle* h) {
    std::lock_guard<std::mutex> guard(h->host_init_lock);
    if (h->h_ring.load(std::memory_order_acquire)) return AIRBAND_HIP_OK;
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    /* bounded like the reference's ring (MIN_BUF_SIZE = 2 560 000 bytes ~ 4 batches, src/rtl_airband.h:64): the first batch with its
     * lead-in, the batch in flight, one more being filled, and the look-ahead */
    h->ring_cap = (h->first_batch_bytes + 3 * h->batch_bytes + h->lookahead_bytes + 4095) / 4096 * 4096;
    h->stage_stride = (h->first_batch_bytes + h->lookahead_bytes + 255) / 256 * 256;
    for (auto& b : h->d_stage2)
        if (!b.p) HIP_TRY(h, b.alloc((size_t)h->stage_stride * h->plan.n_dev), AIRBAND_HIP_ENOMEM);
    if (!h->h2d) HIP_TRY(h, hipStreamCreateWithFlags(&h->h2d, hipStreamNonBlocking), AIRBAND_HIP_ENODEV);
    for (auto& e : h->ev_h2d)
        if (!e) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
    for (auto& e : h->ev_stage_read)
        if (!e) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
    uint8_t* ring = nullptr;
    HIP_TRY(h, hipHostMalloc((void**)&ring, (size_t)h->ring_cap * h->plan.n_dev, hipHostMallocDefault), AIRBAND_HIP_ENOMEM);
    h->h_ring.store(ring, std::memory_order_release); /* published last: submit() / process() test this pointer without the lock */
    return AIRBAND_HIP_OK;
}

int64_t airband_hip_submit(airband_hip_handle* h, int32_t dev, const void* iq, size_t nbytes) {
    if (!h || !iq) return fail(h, AIRBAND_HIP_EINVAL, "NULL argument");
    if (dev < 0 || dev >= h->plan.n_dev) return fail(h, AIRBAND_HIP_EINVAL, "device index out of range");
    if (!h->dev_enabled[dev]) return (int64_t)nbytes; /* a disabled dongle's bytes are dropped, as nobody reads a failed input's ring any more */
    uint8_t* ring = h->h_ring.load(std::memory_order_acquire);
    if (!ring) { /* the first submit of a handle sets the path up; concurrent first submits for different dongles serialise on the lock inside */
        const int rc = host_path_init(h);
        if (rc != AIRBAND_HIP_OK) return rc;
        ring = h->h_ring.load(std::memory_order_acquire);
    }
    const uint64_t wr = h->ring_wr[dev].load(std::memory_order_relaxed);
    const uint64_t used = wr - h->ring_free.load(std::memory_order_acquire);
    size_t take = nbytes;
    if (used + take > (uint64_t)h->ring_cap) take = (uint64_t)h->ring_cap > used ? (size_t)((uint64_t)h->ring_cap - used) : 0;
    uint8_t* row = ring + (size_t)dev * h->ring_cap;
    const size_t pos = (size_t)(wr % (uint64_t)h->ring_cap);
    const size_t first = take < (size_t)h->ring_cap - pos ? take : (size_t)h->ring_cap - pos;
    std::memcpy(row + pos, iq, first);
    if (take > first) std::memcpy(row, (const uint8_t*)iq + first, take - first);
    h->ring_wr[dev].store(wr + take, std::memory_order_release);
    return (int64_t)take;
}

/* would airband_hip_process() run a batch now?  The availability rule alone, nothing is enqueued.  -2 from airband_hip_process's own codes is not
 * used: OK = yes, EAGAIN = not yet (or every dongle is switched off). */
int airband_hip_batch_ready(airband_hip_handle* h) {
    if (!h) return AIRBAND_HIP_EINVAL;
    if (h->n_enabled == 0) return AIRBAND_HIP_EAGAIN;
    if (!h->h_ring.load(std::memory_order_acquire)) return AIRBAND_HIP_EAGAIN; /* nothing has been submitted yet */
    const int64_t need = (h->front_batches == 0 ? h->first_batch_bytes : h->batch_bytes) + h->lookahead_bytes;
    for (int d = 0; d < h->plan.n_dev; d++)
        if (h->dev_enabled[d] && (int64_t)(h->ring_wr[d].load(std::memory_order_acquire) - h->ring_rd) < need) return AIRBAND_HIP_EAGAIN;
    return AIRBAND_HIP_OK;
}

int airband_hip_process(airband_hip_handle* h) {
    if (!h) return AIRBAND_HIP_EINVAL;
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    if (!h->h_ring.load(std::memory_order_acquire)) {
        const int rc = host_path_init(h);
        if (rc != AIRBAND_HIP_OK) return rc;
    }
    uint8_t* const ring = h->h_ring.load(std::memory_order_acquire);
    const bool first = h->front_batches == 0;
    const int64_t consume = first ? h->first_batch_bytes : h->batch_bytes;
    const int64_t need = consume + h->lookahead_bytes; /* availability rule (src/rtl_airband.cpp:394-400) applied to a whole batch */
    if (h->n_enabled == 0) return AIRBAND_HIP_EAGAIN; /* every dongle switched off: nothing to demodulate (the reference exits, src/rtl_airband.cpp:377-381) */
    for (int d = 0; d < h->plan.n_dev; d++) /* dongles switched off (failed inputs) are not waited for: next_device() passes them by, src/rtl_airband.cpp:383-391 */
        if (h->dev_enabled[d] && (int64_t)(h->ring_wr[d].load(std::memory_order_acquire) - h->ring_rd) < need) return AIRBAND_HIP_EAGAIN;
    const int b = (int)(h->host_batches & 1);
    /* the DMA of the previous batch has left the ring: its bytes (up to the look-ahead the next batch re-reads) may be overwritten */
    if (h->host_batches > 0) {
        HIP_TRY(h, hipEventSynchronize(h->ev_h2d[b ^ 1]), AIRBAND_HIP_ERUNTIME);
        h->ring_free.store(h->ring_rd, std::memory_order_release);
    }
    /* staging buffer b was last read by the kernels of batch (k - 2) */
    if (h->host_batches >= 2) HIP_TRY(h, hipStreamWaitEvent(h->h2d, h->ev_stage_read[b], 0), AIRBAND_HIP_ERUNTIME);
    const size_t pos = (size_t)(h->ring_rd % (uint64_t)h->ring_cap);
    const size_t run = (size_t)need < (size_t)h->ring_cap - pos ? (size_t)need : (size_t)h->ring_cap - pos;
    HIP_TRY(h, hipMemcpy2DAsync(h->d_stage2[b].p, (size_t)h->stage_stride, ring + pos, (size_t)h->ring_cap, run, (size_t)h->plan.n_dev, hipMemcpyHostToDevice, h->h2d),
            AIRBAND_HIP_ERUNTIME);
    if (run < (size_t)need) /* the span wraps around the end of the rings */
        HIP_TRY(h, hipMemcpy2DAsync(h->d_stage2[b].p + run, (size_t)h->stage_stride, ring, (size_t)h->ring_cap, (size_t)need - run, (size_t)h->plan.n_dev,
                                    hipMemcpyHostToDevice, h->h2d),
                AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipEventRecord(h->ev_h2d[b], h->h2d), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_h2d[b], 0), AIRBAND_HIP_ERUNTIME);
    if (h->pipeline) HIP_TRY(h, hipStreamWaitEvent(h->front, h->ev_h2d[b], 0), AIRBAND_HIP_ERUNTIME);
    h->ring_rd += (uint64_t)consume;
    h->host_batches++;
    const int rc = airband_hip_process_device(h, h->d_stage2[b].p, (size_t)h->stage_stride, nullptr);
    /* stage 1 of this batch (the only reader of the staging buffer) is enqueued: mark the point after which the buffer is free again */
    (void)hipEventRecord(h->ev_stage_read[b], h->pipeline ? h->front : h->stream);
    return rc;
}

int airband_hip_process_bins(airband_hip_handle* h, const float* wavein, const float* iq_in) {
    if (!h || !wavein || !iq_in) return fail(h, AIRBAND_HIP_EINVAL, "NULL argument");
    if (h->pipeline) return fail(h, AIRBAND_HIP_EINVAL, "process_bins (stage 2 only) is not available on a pipelined handle");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    const size_t n = (size_t)h->plan.total_ch * h->B;
    if (h->d_tmp_wavein.n < n) {
        h->d_tmp_wavein.release();
        HIP_TRY(h, h->d_tmp_wavein.alloc(n), AIRBAND_HIP_ENOMEM);
    }
    if (h->d_tmp_iqin.n < 2 * n) {
        h->d_tmp_iqin.release();
        HIP_TRY(h, h->d_tmp_iqin.alloc(2 * n), AIRBAND_HIP_ENOMEM);
    }
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_tmp_wavein.p, wavein, n * sizeof(float), hipMemcpyHostToDevice, s), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipMemcpyAsync(h->d_tmp_iqin.p, iq_in, 2 * n * sizeof(float), hipMemcpyHostToDevice, s), AIRBAND_HIP_ERUNTIME);
    hipEvent_t* ev = event_set(h, h->front_batches, 0);
    (void)hipEventRecord(ev[0], s);
    h->afc_spectrum_valid = false;
    launch_scatter_bins(h->d_tmp_wavein.p, h->d_tmp_iqin.p, h->d_slot_to_ext.p, h->d_cc.p, h->d_mag.p, h->d_iq.p, h->n_slots, h->B, h->row0, h->R, s);
    (void)hipEventRecord(ev[1], s);
    h->row0_front = (h->row0_front + h->B) % h->R;
    h->front_batches++;
    return run_back_half(h, s);
}

int airband_hip_synchronize(airband_hip_handle* h) {
    if (!h) return AIRBAND_HIP_EINVAL;
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    if (h->front) HIP_TRY(h, hipStreamSynchronize(h->front), AIRBAND_HIP_ERUNTIME);
    if (h->ev_last && h->ev_last_pending) HIP_TRY(h, hipEventSynchronize(h->ev_last), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamSynchronize(h->stream), AIRBAND_HIP_ERUNTIME);
    return AIRBAND_HIP_OK;
}

/* results of channels [first, first + n) of the batch last completed; `consume` marks the batch as collected */
static int collect_range(airband_hip_handle* h, int64_t first, int64_t n, float* waveout, float* iq_out, char* axc, airband_hip_channel_stats* stats, bool consume) {
    if (!h) return AIRBAND_HIP_EINVAL;
    if (first < 0 || n < 0 || first + n > h->plan.total_ch) return fail(h, AIRBAND_HIP_EINVAL, "channel range out of bounds");
    if (!h->results_ready && consume) return AIRBAND_HIP_EAGAIN;
    if (h->batches_done == 0) return AIRBAND_HIP_EAGAIN;
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    hipStream_t s = h->stream;
    order_behind_last_batch(h);
    const size_t nch = (size_t)n;
    if (stats && nch) {
        if (!h->d_stats.p) HIP_TRY(h, h->d_stats.alloc((size_t)h->plan.total_ch), AIRBAND_HIP_ENOMEM);
        launch_stats(h->d_cc.p, h->d_cs.p, h->d_slot_to_ext.p, h->n_slots, h->d_stats.p, s);
        HIP_TRY(h, hipMemcpyAsync(stats, h->d_stats.p + first, nch * sizeof(airband_hip_channel_stats), hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
    }
    if (waveout && nch)
        HIP_TRY(h, hipMemcpy2DAsync(waveout, h->B * sizeof(float), h->d_out_wave.p + (size_t)first * h->wave_stride + AB_OUT_PAD, (size_t)h->wave_stride * sizeof(float),
                                    h->B * sizeof(float), nch, hipMemcpyDeviceToHost, s),
                AIRBAND_HIP_ERUNTIME);
    if (iq_out && nch) {
        if (h->d_out_iq.p)
            HIP_TRY(h, hipMemcpyAsync(iq_out, h->d_out_iq.p + (size_t)first * h->B * 2, nch * h->B * 2 * sizeof(float), hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
        else
            std::memset(iq_out, 0, nch * h->B * 2 * sizeof(float));
    }
    if (axc && nch) HIP_TRY(h, hipMemcpyAsync(axc, h->d_out_axc.p + first, nch, hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamSynchronize(s), AIRBAND_HIP_ERUNTIME);
    if (consume) h->results_ready = false;
    return AIRBAND_HIP_OK;
}

int airband_hip_collect(airband_hip_handle* h, float* waveout, float* iq_out, char* axc, airband_hip_channel_stats* stats) {
    if (!h) return AIRBAND_HIP_EINVAL;
    return collect_range(h, 0, h->plan.total_ch, waveout, iq_out, axc, stats, true);
}

int airband_hip_collect_channels(airband_hip_handle* h, int64_t first_channel, int64_t n_channels, float* waveout, float* iq_out, char* axc,
                                 airband_hip_channel_stats* stats) {
    return collect_range(h, first_channel, n_channels, waveout, iq_out, axc, stats, false);
}

int airband_hip_collect_mixers(airband_hip_handle* h, float* left, float* right, uint8_t* has_signal) {
    if (!h) return AIRBAND_HIP_EINVAL;
    if (h->n_mixers <= 0) return fail(h, AIRBAND_HIP_EINVAL, "no mixers configured");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    hipStream_t s = h->stream;
    order_behind_last_batch(h);
    const size_t n = (size_t)h->n_mixers * h->B;
    if (left) HIP_TRY(h, hipMemcpyAsync(left, h->d_mix_left.p, n * sizeof(float), hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
    if (right) HIP_TRY(h, hipMemcpyAsync(right, h->d_mix_right.p, n * sizeof(float), hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
    if (has_signal) HIP_TRY(h, hipMemcpyAsync(has_signal, h->d_mix_signal.p, (size_t)h->n_mixers, hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamSynchronize(s), AIRBAND_HIP_ERUNTIME);
    return AIRBAND_HIP_OK;
}

int airband_hip_device_results(airband_hip_handle* h, float** d_waveout, float** d_iq_out, uint8_t** d_axc, float** d_mix_left, float** d_mix_right,
                               uint8_t** d_mix_signal) {
    if (!h) return AIRBAND_HIP_EINVAL;
    if (d_waveout) *d_waveout = h->d_out_wave.p + AB_OUT_PAD;
    if (d_iq_out) *d_iq_out = h->d_out_iq.p;
    if (d_axc) *d_axc = h->d_out_axc.p;
    if (d_mix_left) *d_mix_left = h->d_mix_left.p;
    if (d_mix_right) *d_mix_right = h->d_mix_right.p;
    if (d_mix_signal) *d_mix_signal = h->d_mix_signal.p;
    return AIRBAND_HIP_OK;
}

static int gather_last(airband_hip_handle* h, int64_t first, int64_t nch, float* wavein, float* iq_in, uint8_t* trace) {
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    if (first < 0 || nch < 0 || first + nch > h->plan.total_ch) return fail(h, AIRBAND_HIP_EINVAL, "channel range out of bounds");
    if (h->batches_done == 0) return fail(h, AIRBAND_HIP_EAGAIN, "no batch processed yet");
    if (nch == 0) return AIRBAND_HIP_OK;
    const size_t n = (size_t)nch * h->B;
    if (wavein && h->d_tmp_wavein.n < n) {
        h->d_tmp_wavein.release();
        HIP_TRY(h, h->d_tmp_wavein.alloc(n), AIRBAND_HIP_ENOMEM);
    }
    if (iq_in && h->d_tmp_iqin.n < 2 * n) {
        h->d_tmp_iqin.release();
        HIP_TRY(h, h->d_tmp_iqin.alloc(2 * n), AIRBAND_HIP_ENOMEM);
    }
    if (trace && h->d_tmp_trace.n < n) {
        h->d_tmp_trace.release();
        HIP_TRY(h, h->d_tmp_trace.alloc(n), AIRBAND_HIP_ENOMEM);
    }
    hipStream_t s = h->stream;
    order_behind_last_batch(h);
    const int prev_row0 = (h->row0 + h->R - h->B) % h->R; /* row0 of the batch just finished */
    launch_gather_channels(h->d_mag.p, h->d_iq.p, h->d_trace.p, h->d_ext_to_slot.p, h->d_cc.p, (int)first, (int)nch, wavein ? h->d_tmp_wavein.p : nullptr,
                           iq_in ? h->d_tmp_iqin.p : nullptr, trace ? h->d_tmp_trace.p : nullptr, h->B, prev_row0, h->R, s);
    if (wavein) HIP_TRY(h, hipMemcpyAsync(wavein, h->d_tmp_wavein.p, n * sizeof(float), hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
    if (iq_in) HIP_TRY(h, hipMemcpyAsync(iq_in, h->d_tmp_iqin.p, 2 * n * sizeof(float), hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
    if (trace) HIP_TRY(h, hipMemcpyAsync(trace, h->d_tmp_trace.p, n, hipMemcpyDeviceToHost, s), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamSynchronize(s), AIRBAND_HIP_ERUNTIME);
    return AIRBAND_HIP_OK;
}

int airband_hip_read_bins(airband_hip_handle* h, float* wavein, float* iq_in) {
    if (!h) return AIRBAND_HIP_EINVAL;
    return gather_last(h, 0, h->plan.total_ch, wavein, iq_in, nullptr);
}

int airband_hip_read_bins_channels(airband_hip_handle* h, int64_t first_channel, int64_t n_channels, float* wavein, float* iq_in) {
    if (!h) return AIRBAND_HIP_EINVAL;
    return gather_last(h, first_channel, n_channels, wavein, iq_in, nullptr);
}

int airband_hip_read_trace(airband_hip_handle* h, uint8_t* state) {
    if (!h || !state) return AIRBAND_HIP_EINVAL;
    if (!(h->flags & AIRBAND_HIP_FLAG_TRACE_SQUELCH)) return fail(h, AIRBAND_HIP_EINVAL, "handle was prepared without AIRBAND_HIP_FLAG_TRACE_SQUELCH");
    return gather_last(h, 0, h->plan.total_ch, nullptr, nullptr, state);
}

int airband_hip_read_trace_channels(airband_hip_handle* h, int64_t first_channel, int64_t n_channels, uint8_t* state) {
    if (!h || !state) return AIRBAND_HIP_EINVAL;
    if (!(h->flags & AIRBAND_HIP_FLAG_TRACE_SQUELCH)) return fail(h, AIRBAND_HIP_EINVAL, "handle was prepared without AIRBAND_HIP_FLAG_TRACE_SQUELCH");
    return gather_last(h, first_channel, n_channels, nullptr, nullptr, state);
}

int airband_hip_channel_constants(const airband_hip_handle* h, int32_t channel_index, double* out_vals) {
    if (!h || !out_vals || channel_index < 0 || channel_index >= h->plan.total_ch) return AIRBAND_HIP_EINVAL;
    channel_constants(h->plan, channel_index, out_vals);
    return AIRBAND_HIP_OK;
}

int airband_hip_last_timings(airband_hip_handle* h, float* ms4) {
    if (!h || !ms4) return AIRBAND_HIP_EINVAL;
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    harvest_timings(h, true);
    if (!h->timings_valid) return AIRBAND_HIP_EAGAIN;
    for (int k = 0; k < 4; k++) ms4[k] = h->t_last[k];
    return AIRBAND_HIP_OK;
}

int airband_hip_timing_totals(airband_hip_handle* h, double* ms4_sum, int64_t* n_batches, int32_t reset) {
    if (!h) return AIRBAND_HIP_EINVAL;
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    harvest_timings(h, true);
    if (ms4_sum)
        for (int k = 0; k < 4; k++) ms4_sum[k] = h->t_sum[k];
    if (n_batches) *n_batches = h->t_n[0];
    if (reset) {
        for (double& v : h->t_sum) v = 0.0;
        h->t_n[0] = 0;
    }
    return AIRBAND_HIP_OK;
}

#ifndef AB_BUILD_DEFINES
#define AB_BUILD_DEFINES ""
#endif
const char* airband_hip_build_info(void) { return AB_BUILD_DEFINES; }

int airband_hip_regrouped(const airband_hip_handle* h) { return (h && h->regroup) ? 1 : 0; }

const char* airband_hip_channelizer_name(const airband_hip_handle* h) {
    return (h && h->use_dft) ? "dft_mfma_i8" : (h && h->use_f32) ? "dft_mfma_f32" : "fft_wave64";
}

int airband_hip_set_signal_plan(airband_hip_handle* h, const int64_t* carriers, int32_t n_carriers, int32_t noise_q8, const int16_t* sin_table4096) {
    if (!h || !carriers || !sin_table4096 || n_carriers < 1 || n_carriers > 16) return fail(h, AIRBAND_HIP_EINVAL, "bad signal plan (1..16 carriers)");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    h->d_carriers.release();
    h->d_sin_tab.release();
    std::vector<long long> c(carriers, carriers + (size_t)n_carriers * 12);
    std::vector<int16_t> t(sin_table4096, sin_table4096 + 4096);
    HIP_TRY(h, upload(h->d_carriers, c), AIRBAND_HIP_ENOMEM);
    HIP_TRY(h, upload(h->d_sin_tab, t), AIRBAND_HIP_ENOMEM);
    h->n_carriers = n_carriers;
    h->noise_q8 = noise_q8;
    return AIRBAND_HIP_OK;
}

int airband_hip_set_signal_plan_shift(airband_hip_handle* h, int32_t n_plans, uint32_t shift_step) {
    if (!h || n_plans < 1 || n_plans > 65536) return fail(h, AIRBAND_HIP_EINVAL, "bad plan count (1..65536)");
    h->sig_n_plans = n_plans;
    h->sig_shift_step = shift_step;
    return AIRBAND_HIP_OK;
}

int airband_hip_generate_iq(airband_hip_handle* h, void* d_iq, size_t stride_bytes, uint64_t start_byte, size_t nbytes, uint64_t seed, int32_t device_index_offset,
                            void* stream) {
    if (!h || !d_iq) return fail(h, AIRBAND_HIP_EINVAL, "NULL argument");
    if (h->n_carriers == 0) return fail(h, AIRBAND_HIP_EINVAL, "call airband_hip_set_signal_plan first");
    if (h->plan.dev[0].sfmt != AIRBAND_SFMT_U8) return fail(h, AIRBAND_HIP_EINVAL, "the synthetic generator emits u8 I/Q");
    if ((start_byte & 1) || (nbytes & 1)) return fail(h, AIRBAND_HIP_EINVAL, "byte ranges must cover whole I/Q pairs");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    SiggenArgs a;
    a.iq = (uint8_t*)d_iq;
    a.stride = (long)stride_bytes;
    a.sin_tab = h->d_sin_tab.p;
    a.carriers = h->d_carriers.p;
    a.n_carriers = h->n_carriers;
    a.n_dev = h->plan.n_dev;
    a.dev_offset = device_index_offset;
    a.start_sample = start_byte / 2;
    a.n_samples = (long)(nbytes / 2);
    a.seed = seed;
    a.noise_q8 = h->noise_q8;
    a.n_plans = h->sig_n_plans;
    a.plan_shift_step = h->sig_shift_step;
    launch_siggen(a, stream ? (hipStream_t)stream : h->stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, AIRBAND_HIP_ERUNTIME, std::string("siggen launch: ") + hipGetErrorString(e));
    return AIRBAND_HIP_OK;
}

/* ---- the mixer exchange (include/airband_hip.h) ------------------------------------------------------------------------------------ */
namespace {
struct Rccl {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool named = false;         /* loaded from AIRBAND_HIP_RCCL_LIB */
    bool shared_gpu_ok = false; /* the library says (exported symbol airband_rccl_allows_shared_gpu) that two ranks of a communicator may sit on ONE GPU: the test stand-in does, RCCL does not */
    std::string why;
};
Rccl g_rccl;
std::mutex g_rccl_lock;

/* librccl.so on first use: a process that never exchanges mixer sums never loads it */
Rccl* rccl() {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    if (g_rccl.dl) return &g_rccl;
    if (!g_rccl.why.empty()) return nullptr;
    void* dl = nullptr;
    std::string tried;
    /* AIRBAND_HIP_RCCL_LIB names the library instead (a site's own RCCL build; the in-process stand-in tests/fake_rccl/ the GPU suite uses to run
     * the exchange with two ranks on a one-GPU box).  Whether two ranks of a communicator may share a GPU is a capability the library exports
     * (airband_rccl_allows_shared_gpu, below): the stand-in has it, RCCL does not. */
    const char* named = getenv("AIRBAND_HIP_RCCL_LIB");
    if (named && *named) {
        dl = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
        if (!dl) {
            const char* e = dlerror(); /* once: dlerror() clears the message it returns */
            tried = std::string(named) + ": " + (e ? e : "not found");
        }
        g_rccl.named = dl != nullptr;
    } else {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            dl = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (dl) break;
            const char* e = dlerror();
            if (tried.empty()) tried = std::string("librccl.so: ") + (e ? e : "not found");
        }
    }
    if (!dl) {
        g_rccl.why = tried.empty() ? std::string("librccl.so: not found") : tried;
        return nullptr;
    }
#define AB_RCCL_SYM(field, name)                              \
    *(void**)(&g_rccl.field) = dlsym(dl, name);               \
    if (!g_rccl.field) {                                      \
        g_rccl.why = std::string("librccl.so lacks ") + name; \
        dlclose(dl);                                          \
        return nullptr;                                       \
    }
    AB_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") AB_RCCL_SYM(CommInitRank, "ncclCommInitRank") AB_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    AB_RCCL_SYM(AllReduce, "ncclAllReduce") AB_RCCL_SYM(CommDestroy, "ncclCommDestroy") AB_RCCL_SYM(GroupStart, "ncclGroupStart")
    AB_RCCL_SYM(GroupEnd, "ncclGroupEnd") AB_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef AB_RCCL_SYM
    {   /* a capability the library declares itself, not something inferred from how it was named: a site's own RCCL build named through AIRBAND_HIP_RCCL_LIB
         * keeps the duplicate-GPU check of comm_init_all */
        typedef int (*cap_fn)(void);
        cap_fn cap = reinterpret_cast<cap_fn>(dlsym(dl, "airband_rccl_allows_shared_gpu"));
        g_rccl.shared_gpu_ok = cap && cap() != 0;
    }
    g_rccl.dl = dl;
    return &g_rccl;
}

#define RCCL_TRY(h, R, expr)                                                                                        \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess) return fail(h, AIRBAND_HIP_ERUNTIME, std::string(#expr ": ") + (R)->GetErrorString(r_)); \
    } while (0)

/* the stream on which the last batch's mixer sums become final */
hipStream_t results_stream(airband_hip_handle* h) { return h->pipeline ? h->stream : (h->last_stream ? h->last_stream : h->stream); }
}  // namespace

int airband_hip_mixer_set_stereo(airband_hip_handle* h, int32_t mixer, int32_t stereo) {
    if (!h || h->n_mixers <= 0 || mixer < 0 || mixer >= h->n_mixers) return fail(h, AIRBAND_HIP_EINVAL, "mixer index out of range");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    order_behind_last_batch(h);
    const uint8_t v = stereo ? 1 : 0;
    HIP_TRY(h, hipMemcpyAsync(h->d_mix_stereo.p + mixer, &v, 1, hipMemcpyHostToDevice, h->stream), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipStreamSynchronize(h->stream), AIRBAND_HIP_ERUNTIME);
    return AIRBAND_HIP_OK;
}

int airband_hip_comm_unique_id(uint8_t id[AIRBAND_HIP_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == AIRBAND_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    Rccl* R = rccl();
    if (!R) return fail(nullptr, AIRBAND_HIP_ENODEV, g_rccl.why);
    ncclUniqueId u;
    RCCL_TRY(nullptr, R, R->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return AIRBAND_HIP_OK;
}

int airband_hip_comm_init_rank(airband_hip_handle* h, const uint8_t id[AIRBAND_HIP_COMM_ID_BYTES], int32_t nranks, int32_t rank) {
    if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, AIRBAND_HIP_EINVAL, "bad communicator arguments");
    if (h->comm) return fail(h, AIRBAND_HIP_EINVAL, "the handle already has a communicator");
    Rccl* R = rccl();
    if (!R) return fail(h, AIRBAND_HIP_ENODEV, g_rccl.why);
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    RCCL_TRY(h, R, R->CommInitRank(&h->comm, nranks, u, rank));
    return AIRBAND_HIP_OK;
}

int airband_hip_comm_init_all(airband_hip_handle** hs, int32_t n) {
    if (!hs || n < 1) return fail(nullptr, AIRBAND_HIP_EINVAL, "bad communicator arguments");
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) {
        if (!hs[i] || hs[i]->comm) return fail(hs[i], AIRBAND_HIP_EINVAL, "NULL handle, or a handle that already has a communicator");
        devs[i] = hs[i]->hip_device;
        if (hs[i]->n_mixers != hs[0]->n_mixers || hs[i]->B != hs[0]->B) return fail(hs[i], AIRBAND_HIP_EINVAL, "the handles of a clique need the same mixer_count and WAVE_BATCH");
    }
    Rccl* R = rccl();
    if (!R) return fail(hs[0], AIRBAND_HIP_ENODEV, g_rccl.why);
    if (!R->shared_gpu_ok) /* RCCL proper refuses a communicator with one GPU twice, late and with a generic message */
        for (int i = 0; i < n; i++)
            for (int k = 0; k < i; k++)
                if (devs[k] == devs[i]) return fail(hs[i], AIRBAND_HIP_EINVAL, "two handles of the clique share a GPU: use airband_hip_add_mixers between them");
    std::vector<ncclComm_t> comms(n, nullptr);
    RCCL_TRY(hs[0], R, R->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; i++) hs[i]->comm = comms[i];
    return AIRBAND_HIP_OK;
}

int airband_hip_comm_group_begin(void) {
    Rccl* R = rccl();
    if (!R) return fail(nullptr, AIRBAND_HIP_ENODEV, g_rccl.why);
    RCCL_TRY(nullptr, R, R->GroupStart());
    return AIRBAND_HIP_OK;
}

int airband_hip_comm_group_end(void) {
    Rccl* R = rccl();
    if (!R) return fail(nullptr, AIRBAND_HIP_ENODEV, g_rccl.why);
    RCCL_TRY(nullptr, R, R->GroupEnd());
    return AIRBAND_HIP_OK;
}

/* SUM of the mixer waveforms (src/mixer.cpp:133-140), MAX of the signal flags (channel->axcindicate = SIGNAL if any input had signal, :209), in place */
int airband_hip_allreduce_mixers(airband_hip_handle* h, void* stream) {
    if (!h) return AIRBAND_HIP_EINVAL;
    if (h->n_mixers <= 0) return fail(h, AIRBAND_HIP_EINVAL, "no mixers configured");
    if (!h->comm) return fail(h, AIRBAND_HIP_EINVAL, "no communicator: airband_hip_comm_init_rank / _init_all first");
    Rccl* R = rccl();
    if (!R) return fail(h, AIRBAND_HIP_ENODEV, g_rccl.why);
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    hipStream_t s = stream ? (hipStream_t)stream : results_stream(h);
    if (stream) {
        const int rc = airband_hip_stream_wait_results(h, stream);
        if (rc != AIRBAND_HIP_OK) return rc;
    }
    const size_t n = (size_t)h->n_mixers * h->B;
    RCCL_TRY(h, R, R->GroupStart());
    RCCL_TRY(h, R, R->AllReduce(h->d_mix_left.p, h->d_mix_left.p, n, ncclFloat, ncclSum, h->comm, s));
    RCCL_TRY(h, R, R->AllReduce(h->d_mix_right.p, h->d_mix_right.p, n, ncclFloat, ncclSum, h->comm, s));
    RCCL_TRY(h, R, R->AllReduce(h->d_mix_signal.p, h->d_mix_signal.p, (size_t)h->n_mixers, ncclUint8, ncclMax, h->comm, s));
    RCCL_TRY(h, R, R->GroupEnd());
    if (s != h->stream) { /* (a caller's stream, handed in here or to process_device) whatever the handle does next to these buffers (the next batch's sums, collect_mixers) comes behind the exchange */
        if (!h->ev_last) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
        HIP_TRY(h, hipEventRecord(h->ev_last, s), AIRBAND_HIP_ERUNTIME);
        h->ev_last_pending = true;
    }
    return AIRBAND_HIP_OK;
}

int airband_hip_add_mixers(airband_hip_handle* dst, airband_hip_handle* src) {
    if (!dst || !src || dst == src) return fail(dst, AIRBAND_HIP_EINVAL, "two different handles needed");
    if (dst->n_mixers <= 0 || dst->n_mixers != src->n_mixers || dst->B != src->B) return fail(dst, AIRBAND_HIP_EINVAL, "the handles need the same mixer_count and WAVE_BATCH");
    if (dst->hip_device != src->hip_device) return fail(dst, AIRBAND_HIP_EINVAL, "handles on different GPUs exchange through airband_hip_allreduce_mixers");
    HIP_TRY(dst, hipSetDevice(dst->hip_device), AIRBAND_HIP_ENODEV);
    if (!dst->ev_peer) HIP_TRY(dst, hipEventCreateWithFlags(&dst->ev_peer, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
    hipStream_t s = results_stream(dst);
    HIP_TRY(dst, hipEventRecord(dst->ev_peer, results_stream(src)), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(dst, hipStreamWaitEvent(s, dst->ev_peer, 0), AIRBAND_HIP_ERUNTIME);
    launch_mix_add(dst->d_mix_left.p, dst->d_mix_right.p, dst->d_mix_signal.p, src->d_mix_left.p, src->d_mix_right.p, src->d_mix_signal.p, dst->n_mixers, dst->B, s);
    /* src's next batch must not overwrite its sums before they have been read */
    if (!src->ev_last) HIP_TRY(src, hipEventCreateWithFlags(&src->ev_last, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
    HIP_TRY(src, hipEventRecord(src->ev_last, s), AIRBAND_HIP_ERUNTIME);
    src->ev_last_pending = true;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(dst, AIRBAND_HIP_ERUNTIME, std::string("mixer add launch: ") + hipGetErrorString(e));
    return AIRBAND_HIP_OK;
}

/* A handle that ran no batch this round (every dongle of it switched off) still stands in the exchange: its partial sums are those of a
 * mixer whose inputs are all masked out (mixer_disable_input(), src/mixer.cpp:96-112) -- zeros, no signal.  Its buffers do NOT hold that by
 * themselves: the last batch's sums are still in them, and after an in-place all-reduce the whole node's. */
int airband_hip_clear_mixers(airband_hip_handle* h) {
    if (!h) return AIRBAND_HIP_EINVAL;
    if (h->n_mixers <= 0) return fail(h, AIRBAND_HIP_EINVAL, "no mixers configured");
    HIP_TRY(h, hipSetDevice(h->hip_device), AIRBAND_HIP_ENODEV);
    hipStream_t s = results_stream(h); /* behind whatever wrote or read the sums last: the handle's last batch, an exchange, add_mixers */
    if (h->ev_last && h->ev_last_pending) HIP_TRY(h, hipStreamWaitEvent(s, h->ev_last, 0), AIRBAND_HIP_ERUNTIME); /* a peer's add_mixers still reading them */
    const size_t n = (size_t)h->n_mixers * h->B;
    HIP_TRY(h, hipMemsetAsync(h->d_mix_left.p, 0, n * sizeof(float), s), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipMemsetAsync(h->d_mix_right.p, 0, n * sizeof(float), s), AIRBAND_HIP_ERUNTIME);
    HIP_TRY(h, hipMemsetAsync(h->d_mix_signal.p, 0, (size_t)h->n_mixers, s), AIRBAND_HIP_ERUNTIME);
    if (s != h->stream) { /* a caller's stream (the last batch ran there): collect_mixers, or a later batch on another stream, comes behind the clear -- as behind allreduce_mixers */
        if (!h->ev_last) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming), AIRBAND_HIP_ENODEV);
        HIP_TRY(h, hipEventRecord(h->ev_last, s), AIRBAND_HIP_ERUNTIME);
        h->ev_last_pending = true;
    }
    return AIRBAND_HIP_OK;
}

int airband_hip_comm_destroy(airband_hip_handle* h) {
    if (!h) return AIRBAND_HIP_EINVAL;
    if (!h->comm) return AIRBAND_HIP_OK;
    Rccl* R = rccl();
    if (R) {
        (void)hipSetDevice(h->hip_device);
        (void)hipStreamSynchronize(h->stream);
        (void)R->CommDestroy(h->comm);
    }
    h->comm = nullptr;
    return AIRBAND_HIP_OK;
}

} /* extern "C" */
