/* csrc/misc_kernels.hip -- small helper kernels around the two hot stages: synthetic-dongle generator, mixer
 * sum, and the layout shuffles used by the parity/introspection entry points. */
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace airband {

/* ---- synthetic dongles (integer-only twin of rtlsdr-airband_amd/siggen.py::generate_u8) ------------------- */
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void siggen_kernel(SiggenArgs a) {
    __shared__ short tab[4096];
    __shared__ long long car[16 * 12];
    __shared__ unsigned ph0[16];
    for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = a.sin_tab[i];
    for (int i = threadIdx.x; i < a.n_carriers * 12; i += 256) car[i] = a.carriers[i];
    const int d = blockIdx.y;
    const int dev = a.dev_offset + d;
    if ((int)threadIdx.x < a.n_carriers)
        ph0[threadIdx.x] = (unsigned)mix64((a.seed ^ 0xC0FFEEull) + (unsigned long long)((dev << 8) | (int)threadIdx.x));
    __shared__ unsigned step_of[16];
    if ((int)threadIdx.x < a.n_carriers) { /* the carrier's phase step for THIS dongle: the table's, moved by the dongle's plan (SiggenArgs::n_plans) */
        const unsigned plan = a.n_plans > 1 ? (unsigned)(dev % a.n_plans) : 0u;
        step_of[threadIdx.x] = (unsigned)a.carriers[threadIdx.x * 12] + a.plan_shift_step * ((plan >> (2 * (threadIdx.x & 15))) & 3u);
    }
    __syncthreads();
    uint8_t* out = a.iq + (long)d * a.stride;
    /* each thread produces 8 complex samples = 16 bytes per pass */
    for (long base = ((long)blockIdx.x * 256 + threadIdx.x) * 8; base < a.n_samples; base += (long)gridDim.x * 256 * 8) {
        uint8_t bytes[16];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned long long n = a.start_sample + (unsigned long long)(base + u);
            const unsigned n32 = (unsigned)n;
            long long acc_i = 0, acc_q = 0;
            for (int c = 0; c < a.n_carriers; c++) {
                const long long* k = car + c * 12;
                unsigned ph = step_of[c] * n32 + ph0[c];
                const unsigned pa = (unsigned)k[3] * n32;
                const long long s_mod = tab[pa >> 20];
                long long amp;
                if (k[2] == 0) {
                    amp = (k[1] * (32768 + ((k[4] * s_mod) >> 15))) >> 15;
                } else {
                    amp = k[1];
                    long long dphi = k[5] * s_mod;
                    if (k[6]) {
                        const unsigned pc = (unsigned)k[6] * n32;
                        dphi += k[7] * (long long)tab[pc >> 20];
                    }
                    ph = (unsigned)((long long)ph + dphi);
                }
                if (k[8]) {
                    const unsigned long long period = (unsigned long long)k[8];
                    const unsigned long long t0 = (unsigned long long)(k[11] * ((dev + k[10]) % 8));
                    const bool on = ((n + (period - (t0 % period))) % period) < (unsigned long long)k[9];
                    if (!on) amp = 0;
                }
                const int idx = ph >> 20;
                acc_i += (amp * (long long)tab[(idx + 1024) & 4095]) >> 15;
                acc_q += (amp * (long long)tab[idx]) >> 15;
            }
            const unsigned long long h = mix64((a.seed ^ ((unsigned long long)dev * 0xD1B54A32D192ED03ull)) + n * 0x9E3779B97F4A7C15ull);
            const long long n_i = (long long)((h & 0xff) + ((h >> 8) & 0xff) + ((h >> 16) & 0xff) + ((h >> 24) & 0xff)) - 510;
            const long long n_q = (long long)(((h >> 32) & 0xff) + ((h >> 40) & 0xff) + ((h >> 48) & 0xff) + ((h >> 56) & 0xff)) - 510;
            acc_i += (n_i * a.noise_q8) >> 8;
            acc_q += (n_q * a.noise_q8) >> 8;
            long long vi = 128 + (acc_i >> 8), vq = 128 + (acc_q >> 8);
            vi = vi < 0 ? 0 : (vi > 255 ? 255 : vi);
            vq = vq < 0 ? 0 : (vq > 255 ? 255 : vq);
            bytes[2 * u] = (uint8_t)vi;
            bytes[2 * u + 1] = (uint8_t)vq;
        }
        if (base + 8 <= a.n_samples) {
            *reinterpret_cast<uint4*>(out + 2 * base) = *reinterpret_cast<const uint4*>(bytes);
        } else {
            for (long u = 0; base + u < a.n_samples; u++) {
                out[2 * (base + u)] = bytes[2 * u];
                out[2 * (base + u) + 1] = bytes[2 * u + 1];
            }
        }
    }
}

void launch_siggen(const SiggenArgs& a, hipStream_t stream) {
    long per_dev_threads = (a.n_samples + 7) / 8;
    int bx = (int)((per_dev_threads + 255) / 256);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    /* gridDim.y is limited to 65535: walk the dongles in slabs */
    for (int d0 = 0; d0 < a.n_dev; d0 += 32768) {
        SiggenArgs b = a;
        b.iq = a.iq + (long)d0 * a.stride;
        b.dev_offset = a.dev_offset + d0;
        b.n_dev = a.n_dev - d0 < 32768 ? a.n_dev - d0 : 32768;
        hipLaunchKernelGGL(siggen_kernel, dim3(bx, b.n_dev), dim3(256), 0, stream, b);
    }
}

/* ---- mixer sum (reference: src/mixer.cpp:133-140 mix_waveforms, :201-214) -------------------------------
 * Two stages so that mixers with hundreds of thousands of inputs (BASELINE config #5) parallelise: stage A sums runs of
 * up to MIX_RUN consecutive inputs of one mixer in connection order, stage B adds the run sums in order.  A mixer with
 * <= MIX_RUN inputs (every mixer in the reference's example configs) is therefore summed in exactly the reference's order. */
/* (a thread sums FOUR consecutive samples, and the rows of eight inputs are in flight before the first of them is added: one dependent 4-byte load per
 * input and thread was a memory round trip per input -- 1.46 ms for 524 288 inputs where the bytes take 0.5.  The additions keep the connection order.) */
__global__ __launch_bounds__(256) void mix_runs_kernel(MixArgs a) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int run = blockIdx.y;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4; /* WAVE_BATCH is a multiple of four; rows start 16-byte aligned (AB_OUT_PAD) */
    const int first = a.run_first[run], last = a.run_first[run + 1];
    const bool stereo = a.mixer_stereo[a.run_mixer[run]] != 0;
    const bool mine = t < a.wave_batch;
    v4f l = {0.0f, 0.0f, 0.0f, 0.0f}, r = {0.0f, 0.0f, 0.0f, 0.0f};
    bool any = false;
    constexpr int G = 8;
    for (int i0 = first; i0 < last; i0 += G) {
        v4f w[G];
        float ml[G], mr[G];
        bool use[G];
#pragma unroll
        for (int k = 0; k < G; k++) { /* block-uniform conditions: scalar branches around the loads */
            const int i = i0 + k;
            const int ch = i < last ? a.in_chan[i] : -1; /* < 0: input masked out (mixer_disable_input, src/mixer.cpp:96-110) */
            use[k] = ch >= 0 && a.out_axc[ch] != ' ';    /* has_signal == false: nothing is added (src/mixer.cpp:119-122,203) */
            w[k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            ml[k] = mr[k] = 0.0f;
            if (use[k]) {
                ml[k] = a.in_ml[i];
                mr[k] = a.in_mr[i];
                if (mine) w[k] = *reinterpret_cast<const v4f*>(a.out_wave + (long)ch * a.wave_stride + t);
            }
        }
#pragma unroll
        for (int k = 0; k < G; k++) {
            if (!use[k]) continue;
            any = true;
            if (ml[k] != 0.0f) l += w[k] * ml[k];
            if (stereo && mr[k] != 0.0f) r += w[k] * mr[k];
        }
    }
    if (mine) {
        *reinterpret_cast<v4f*>(a.run_left + (long)run * a.wave_batch + t) = l;
        *reinterpret_cast<v4f*>(a.run_right + (long)run * a.wave_batch + t) = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.run_signal[run] = any ? 1 : 0;
}

__global__ __launch_bounds__(256) void mix_final_kernel(MixArgs a) {
    const int m = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int first = a.mixer_first_run[m], last = a.mixer_first_run[m + 1];
    float l = 0.0f, r = 0.0f;
    bool any = false;
    for (int k = first; k < last; k++) {
        any = any || a.run_signal[k] != 0;
        if (t < a.wave_batch) {
            if (k == first) { /* keeps a single-run mixer bit-identical to the sequential sum (0 + x is exact, but -0.0 would not survive) */
                l = a.run_left[(long)k * a.wave_batch + t];
                r = a.run_right[(long)k * a.wave_batch + t];
            } else {
                l += a.run_left[(long)k * a.wave_batch + t];
                r += a.run_right[(long)k * a.wave_batch + t];
            }
        }
    }
    if (t < a.wave_batch) {
        a.left[(long)m * a.wave_batch + t] = l;
        a.right[(long)m * a.wave_batch + t] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.has_signal[m] = any ? 1 : 0;
}

__global__ __launch_bounds__(256) void mix_add_kernel(float* dl, float* dr, uint8_t* ds, const float* sl, const float* sr, const uint8_t* ss, long n, int n_mixers) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        dl[i] += sl[i];
        dr[i] += sr[i];
    }
    if (i < n_mixers) ds[i] = (ds[i] | ss[i]) ? 1 : 0;
}

void launch_mix_add(float* dl, float* dr, uint8_t* ds, const float* sl, const float* sr, const uint8_t* ss, int n_mixers, int wave_batch, hipStream_t stream) {
    const long n = (long)n_mixers * wave_batch;
    hipLaunchKernelGGL(mix_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dl, dr, ds, sl, sr, ss, n, n_mixers);
}

void launch_mix(const MixArgs& a, hipStream_t stream) {
    const int bx = (a.wave_batch + 255) / 256;
    hipLaunchKernelGGL(mix_runs_kernel, dim3((a.wave_batch / 4 + 255) / 256, a.n_runs), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(mix_final_kernel, dim3(bx, a.n_mixers), dim3(256), 0, stream, a);
}

/* ---- AFC (reference: class AFC, src/rtl_airband.cpp:180-251) ------------------------------------------------
 * After a batch: on a NO_SIGNAL -> signal edge walk to a stronger neighbouring bin of the batch's last FFT (downwards
 * first, then upwards), with the reference's growing threshold; on a signal -> NO_SIGNAL edge return to the base bin. */
__device__ int afc_walk(const float2* fft, int fft_size, int base, float base_value, int afc, int step) {
    float threshold = 0;
    int bin;
    for (bin = base;; bin += step) {
        if (step < 0) {
            if (bin < -step) break;
        } else if (bin + step >= fft_size) {
            break;
        }
        const float2 v = fft[bin + step];
        const float value = v.x * v.x + v.y * v.y;
        if (value <= base_value) break;
        if (base == bin) {
            threshold = (value - base_value) / (float)afc;
        } else {
            if ((value - base_value) < threshold) break;
            threshold = (float)((double)threshold + (double)threshold / 10.0);
        }
    }
    return bin;
}

__global__ void afc_kernel(const ChanConst* cc, ChanState* cs, const float2* spectrum, int fft_size, int n_slots, int* moved_epoch, int epoch) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    const ChanConst c = cc[slot];
    if (!(c.flags & AB_F_VALID) || c.afc == 0) return;
    ChanState* s = cs + slot;
    const int axc = s->axc, prev = s->axc_prev;
    const float2* fft = spectrum + (long)c.dev * fft_size;
    if (axc != ' ' && prev == ' ') {
        const int base = c.base_bin;
        const float2 b = fft[base];
        const float base_value = b.x * b.x + b.y * b.y;
        int bin = afc_walk(fft, fft_size, base, base_value, c.afc, -1);
        if (bin == base) bin = afc_walk(fft, fft_size, base, base_value, c.afc, 1);
        if (s->bin != bin) {
            s->bin = bin;
            if (moved_epoch) *moved_epoch = epoch; /* some channel moved in this batch: the re-tune kernel has work (every writer stores the same number) */
            if (bin > base) s->axc = '<';      /* AFC_UP   (enum status, src/rtl_airband.h:99) */
            else if (bin < base) s->axc = '>'; /* AFC_DOWN */
        }
    } else if (axc == ' ' && prev != ' ') {
        if (s->bin != c.base_bin && moved_epoch) *moved_epoch = epoch;
        s->bin = c.base_bin;
    }
}

void launch_afc(const ChanConst* cc, ChanState* cs, const float* spectrum, int fft_size, int n_slots, int* moved_epoch, int epoch, hipStream_t stream) {
    hipLaunchKernelGGL(afc_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, stream, cc, cs, reinterpret_cast<const float2*>(spectrum), fft_size, n_slots, moved_epoch, epoch);
}

/* One (re, im) column pair of a coefficient table, built on the device: w[n] exp(-2 pi i bin n / N) scaled to 24 bits, three balanced base-256 digits in the MFMA
 * B-fragment layout, plus the column's offset correction -- the host builder's arithmetic (params.cpp, build_dft_tables) with the device's sincospi.  One wavefront
 * works on one table; `sums` = two LDS words of its own. */
__device__ __forceinline__ void build_column_pair(int8_t* bfrag, double* corr, const float* window, int bset, int c, int bin, int N, int lane, long long* sums) {
    const int NP = N > 512 ? N / 512 : 1, NS = N / NP, K = 2 * NS, KS = K / 64;
    const size_t piece_bytes = (size_t)3 * KS * 64 * 16;
    for (int piece = 0; piece < NP; piece++) {
        if (lane < 2) sums[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        int8_t* tab = bfrag + ((size_t)bset * NP + piece) * piece_bytes;
        long long part_re = 0, part_im = 0;
        for (int k = lane; k < K; k += 64) {
            const int n = piece * NS + (k >> 1);
            double sn, cs_;
            sincospi(2.0 * (double)(((long long)bin * n) % N) / (double)N, &sn, &cs_); /* the phase is reduced exactly in integers first */
            const double wc = (double)window[n] * cs_ * AB_DFT_COEF_SCALE, ws = (double)window[n] * sn * AB_DFT_COEF_SCALE;
            /* byte k = 2n + {0: I, 1: Q}:  (I + jQ) w e^{-j th} = (I w cos + Q w sin) + j (Q w cos - I w sin) */
            const int v_re = (int)llround((k & 1) ? ws : wc), v_im = (int)llround((k & 1) ? wc : -ws);
            part_re += v_re;
            part_im += v_im;
            const int s_ = k / 64, gg = (k % 64) / 16, jj = k % 16;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                int rest = half ? v_im : v_re;
                const int ln = gg * 16 + 2 * c + half;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int lo = ((rest + 128) & 255) - 128; /* balanced digit */
                    tab[(((size_t)t * KS + s_) * 64 + ln) * 16 + jj] = (int8_t)lo;
                    rest = (rest - lo) / 256;
                }
            }
        }
        atomicAdd((unsigned long long*)&sums[0], (unsigned long long)part_re); /* integer sums: exact, order-free */
        atomicAdd((unsigned long long*)&sums[1], (unsigned long long)part_im);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); /* LDS operations of one wave complete in order; this keeps the compiler from reordering them */
        __builtin_amdgcn_wave_barrier();
        if (lane < 2) corr[((size_t)bset * NP + piece) * 16 + 2 * c + lane] = 0.5 * (double)sums[lane]; /* (b - 127.5) = (b - 128) + 0.5 */
        __builtin_amdgcn_wave_barrier();
    }
}

/* The same for a CF32 handle's float table (channelizer_f32.hip; layout and arithmetic of params.cpp, build_f32_tables: window x twiddle in double, rounded once):
 * entry [bset][segment][piece][s][lane] is coefficient (k, column lane & 15), k = segment * 2 SEG + piece * VPP + 16 (s / 4) + 4 (lane >> 4) + s % 4 */
__device__ __forceinline__ void build_column_pair_f32(float* ftab, const float* window, int bset, int c, int bin, int N, int lane) {
    const int SEG = N < 2048 ? N : 2048, NSEG = N / SEG, NW = N <= 512 ? 4 : 8, VPP = 2 * SEG / NW, KW = VPP / 4;
    float* tab = ftab + (size_t)bset * NSEG * NW * KW * 64;
    /* the column pair's lanes of every [segment][piece][s] row: lanes g * 16 + 2 c + {0, 1}, g = 0 .. 3 -- eight floats per row; a lane of this wave takes one of them */
    const int rows = NSEG * NW * KW;
    for (int e = lane; e < rows * 8; e += 64) {
        const int row = e >> 3, g = (e >> 1) & 3, half = e & 1;
        const int s_ = row % KW, piece = (row / KW) % NW, seg = row / (KW * NW);
        const int k = seg * 2 * SEG + piece * VPP + 16 * (s_ / 4) + 4 * g + (s_ % 4);
        const int n = k >> 1;
        double sn, cs_;
        sincospi(2.0 * (double)(((long long)bin * n) % N) / (double)N, &sn, &cs_);
        const double wc = (double)window[n] * cs_, ws = (double)window[n] * sn;
        const double v = half ? ((k & 1) ? wc : -ws) : ((k & 1) ? ws : wc);
        tab[(size_t)row * 64 + g * 16 + 2 * c + half] = (float)v;
    }
}
__device__ __forceinline__ void copy_column_pair_f32(float* ftab, int bset, int home, int c, int N, int lane) {
    const int SEG = N < 2048 ? N : 2048, NSEG = N / SEG, NW = N <= 512 ? 4 : 8, KW = 2 * SEG / NW / 4;
    const size_t each = (size_t)NSEG * NW * KW * 64;
    float* tab = ftab + (size_t)bset * each;
    const float* src = ftab + (size_t)home * each;
    const int rows = NSEG * NW * KW;
    for (int e = lane; e < rows * 8; e += 64) {
        const size_t at = (size_t)(e >> 3) * 64 + ((e >> 1) & 3) * 16 + 2 * c + (e & 1);
        tab[at] = src[at];
    }
}

/* Shared coefficient tables beyond the ones the host builds (round 6: a fleet whose dongles do NOT share a channel plan -- every device_t derives its own bins,
 * src/config.cpp:666-667 -- has one table per distinct group of eight bins; the host builds the first few thousand, the rest are built here at prepare() time, one
 * wavefront per table): tables [first, first + n), every column from bset_bin. */
__global__ __launch_bounds__(256) void build_tables_kernel(int8_t* bfrag, double* corr, const float* window, const int* bset_bin, int first, int n, int fft_size) {
    __shared__ long long sums[4][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= n) return;
    const int bset = first + t;
    for (int c = 0; c < 8; c++) {
        const int bin = bset_bin[bset * 8 + c];
        if (bin < 0) continue; /* a group with fewer than eight channels: the column pair stays zero */
        build_column_pair(bfrag, corr, window, bset, c, bin, fft_size, lane, sums[wave]);
    }
}

void launch_build_tables(int8_t* bfrag, double* corr, const float* window, const int* bset_bin, int first, int n, int fft_size, hipStream_t stream) {
    if (n > 0) hipLaunchKernelGGL(build_tables_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, bfrag, corr, window, bset_bin, first, n, fft_size);
}

/* ---- AFC on the matrix-core channelizer: coefficient columns follow the bins ---------------------------------------------------
 * The pruned DFT has the channel's bin baked into its coefficient table (params.cpp, build_dft_tables).  A group of channels with an AFC
 * channel owns its table; one wavefront per such work item compares, a lane per channel, the bin the table is built for with the bin the
 * channel is tuned to (ChanState::bin, moved by afc_kernel) and rewrites the channel's (re, im) column pair where they differ: the same
 * arithmetic as the host builder -- w[n] exp(-2 pi i bin n / N) scaled to 24 bits, three balanced base-256 digits, in the MFMA B-fragment
 * layout -- plus the column's offset correction.  At start-up every column of a private table differs (-1): the kernel is the builder.
 * The work item is pointed at its private table only while one of its channels is away from its base bin; at home it reads the shared
 * table of the base bins like every group without AFC (no moved channel, no extra coefficient traffic). */
__global__ __launch_bounds__(256) void retune_kernel(RetuneArgs a) {
    __shared__ long long sums[4][2];
    /* no channel of the handle moved in this batch (afc_kernel would have stamped the batch's number): nothing to compare, let alone rebuild */
    if (*a.moved_epoch != a.epoch) return;
    /* one WAVEFRONT per work item (four per workgroup, no workgroup barriers): lane c < 8 compares channel c's bin with the bin its column pair is built
     * for -- one round of loads per item; the first form walked the eight channels one after the other with 256 threads looking on, 0.5 ms per batch
     * at 65 536 items whenever anything had moved anywhere */
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.n_items) return;
    const int bset = a.item_private[item];
    if (bset < a.n_shared) return; /* shared table: its channels never move */
    const int d = a.item_dev[item], g = a.item_group[item];
    const DevConst dev = a.dev[d];
    const int N = a.fft_size, NP = N > 512 ? N / 512 : 1, NS = N / NP, K = 2 * NS, KS = K / 64;
    const size_t piece_bytes = (size_t)3 * KS * 64 * 16;
    int my_bin = -1, my_base = -2;
    bool my_away = false, my_stale = false;
    if (lane < 8 && g * 8 + lane < dev.n_ch) {
        const int slot = a.ext_to_slot[dev.chan_base + g * 8 + lane];
        my_bin = a.cs[slot].bin;
        my_base = a.cc[slot].base_bin;
        my_away = my_bin != my_base;
        my_stale = a.bset_bin[bset * 8 + lane] != my_bin;
    }
    const bool away = __ballot(my_away) != 0ull; /* some channel of the group is off its base bin */
    const unsigned stale = (unsigned)__ballot(my_stale);
    for (int c = 0; c < 8; c++) {
        if (!((stale >> c) & 1u)) continue; /* wave-uniform */
        const int bin = __shfl(my_bin, c);
        /* a channel on its base bin takes its column pair from the group's HOME table, byte for byte: the host built that one (cos / sin / llround of the
         * platform's libm, params.cpp), this kernel builds with the device's sincospi, and a coefficient may differ by one unit between the two -- a
         * channel that has not moved must not see a different table because a neighbour's AFC has (its bins would change by ~1e-7) */
        if (a.ftab) { /* (launch-uniform) CF32: the float table's column pair, copied from the home table or built */
            if (bin == __shfl(my_base, c)) copy_column_pair_f32(a.ftab, bset, a.item_home[item], c, N, lane);
            else build_column_pair_f32(a.ftab, a.window, bset, c, bin, N, lane);
            if (lane == 0) a.bset_bin[bset * 8 + c] = bin;
            continue;
        }
        if (bin == __shfl(my_base, c)) {
            const int home = a.item_home[item];
            for (int piece = 0; piece < NP; piece++) {
                int8_t* tab = a.bfrag + ((size_t)bset * NP + piece) * piece_bytes;
                const int8_t* src = a.bfrag + ((size_t)home * NP + piece) * piece_bytes;
                for (int k = lane; k < K; k += 64) {
                    const int s_ = k / 64, gg = (k % 64) / 16, jj = k % 16;
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const int ln = gg * 16 + 2 * c + half;
#pragma unroll
                        for (int t = 0; t < 3; t++) {
                            const size_t at = (((size_t)t * KS + s_) * 64 + ln) * 16 + jj;
                            tab[at] = src[at];
                        }
                    }
                }
                if (lane < 2) a.corr[((size_t)bset * NP + piece) * 16 + 2 * c + lane] = a.corr[((size_t)home * NP + piece) * 16 + 2 * c + lane];
            }
            if (lane == 0) a.bset_bin[bset * 8 + c] = bin;
            continue;
        }
        build_column_pair(a.bfrag, a.corr, a.window, bset, c, bin, N, lane, sums[wave]);
        if (lane == 0) a.bset_bin[bset * 8 + c] = bin;
    }
    /* the table the next batch's stage 1 reads for this work item: the fleet's shared one while the group is at home */
    if (lane == 0) a.item_bset[item] = away ? bset : a.item_home[item];
}

void launch_retune(const RetuneArgs& a, hipStream_t stream) {
    if (a.n_items > 0) hipLaunchKernelGGL(retune_kernel, dim3((a.n_items + 3) / 4), dim3(256), 0, stream, a);
}

/* ---- layout shuffles for the introspection entry points -------------------------------------------------- */
__global__ void scatter_bins_kernel(const float* wavein, const float* iqin, const int* slot_to_ext, const ChanConst* cc, float* mag, float2* iq, int n_slots,
                                    int wave_batch, int row0, int ring_rows) {
    const int slot = blockIdx.x * 64 + (threadIdx.x & 63);
    const int t = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (slot >= n_slots || t >= wave_batch) return;
    const int ext = slot_to_ext[slot];
    if (ext < 0) return;
    int row = row0 + AB_AGC_EXTRA + t;
    if (row >= ring_rows) row -= ring_rows;
    const long off = ab_tile_base(slot, ring_rows / AB_TILE_ROWS) + ab_tile_off(row);
    mag[off] = wavein[(long)ext * wave_batch + t];
    if (cc[slot].flags & AB_F_RAW_IQ) iq[off] = make_float2(iqin[((long)ext * wave_batch + t) * 2], iqin[((long)ext * wave_batch + t) * 2 + 1]);
}

void launch_scatter_bins(const float* wavein, const float* iqin, const int* slot_to_ext, const ChanConst* cc, float* mag, float2* iq, int n_slots,
                         int wave_batch, int row0, int ring_rows, hipStream_t stream) {
    hipLaunchKernelGGL(scatter_bins_kernel, dim3((n_slots + 63) / 64, (wave_batch + 3) / 4), dim3(256), 0, stream, wavein, iqin, slot_to_ext, cc, mag, iq, n_slots,
                       wave_batch, row0, ring_rows);
}

__global__ void gather_channels_kernel(const float* mag, const float2* iq, const uint8_t* trace, const int* ext_to_slot, const ChanConst* cc, int first, float* wavein,
                                       float* iqin, uint8_t* trace_out, int wave_batch, int row0, int ring_rows) {
    const int c = blockIdx.x; /* channel first + c */
    const int t = blockIdx.y * 256 + threadIdx.x;
    if (t >= wave_batch) return;
    const int slot = ext_to_slot[first + c];
    int row = row0 + AB_AGC_EXTRA + t;
    if (row >= ring_rows) row -= ring_rows;
    const long off = ab_tile_base(slot, ring_rows / AB_TILE_ROWS) + ab_tile_off(row);
    const unsigned flags = cc[slot].flags;
    const float2 q = (flags & AB_F_RAW_IQ) ? iq[off] : make_float2(0.0f, 0.0f);
    if (wavein) /* NFM: wavein[j] = sqrtf(re^2 + im^2) (src/rtl_airband.cpp:484-487); stage 1 leaves it to stage 2, so does this */
        wavein[(long)c * wave_batch + t] = (flags & AB_F_NFM) ? __fsqrt_rn(q.x * q.x + q.y * q.y) : mag[off];
    if (iqin) {
        iqin[((long)c * wave_batch + t) * 2] = q.x;
        iqin[((long)c * wave_batch + t) * 2 + 1] = q.y;
    }
    if (trace_out && trace) trace_out[(long)c * wave_batch + t] = trace[ab_ring_base(slot, wave_batch) + (long)t * AB_SLOT_BLOCK];
}

void launch_gather_channels(const float* mag, const float2* iq, const uint8_t* trace, const int* ext_to_slot, const ChanConst* cc, int first, int n, float* wavein,
                            float* iqin, uint8_t* trace_out, int wave_batch, int row0, int ring_rows, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_channels_kernel, dim3(n, (wave_batch + 255) / 256), dim3(256), 0, stream, mag, iq, trace, ext_to_slot, cc, first, wavein, iqin, trace_out,
                       wave_batch, row0, ring_rows);
}

}  // namespace airband
