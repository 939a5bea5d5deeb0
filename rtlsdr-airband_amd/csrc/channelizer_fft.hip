/* csrc/channelizer_fft.hip -- stage 1 of the hot path on gfx950, general variant: sliding windowed FFT per hop.
 *
 * Replaces, for every hop of every dongle (reference: src/rtl_airband.cpp:402-492):
 *   sample -> float (LUT / scale) x window      (:402-455, NEON twin src/rtl_airband_neon.s:28-83)
 *   forward complex FFT of fft_size points      (:460 fftwf_execute, VideoCore twin gpu_fft_execute :458)
 *   per-channel bin magnitude (+ raw bin I/Q)   (:483-489)
 * for what the matrix-core channelizer (channelizer_dft.hip) does not take: f32 samples, hops that are not a multiple of four bytes,
 * AIRBAND_HIP_FLAG_FORCE_FFT, and the whole spectrum of a batch's last hop that AFC looks at.
 *
 * Mapping (wave64, CDNA4), both kernels:
 *   * a 256-thread workgroup owns one dongle and a tile of HOPS_PER_TILE consecutive hops; the raw bytes those
 *     hops cover ((T-1)*hop + N samples; consecutive windows overlap by N-hop samples) are fetched from HBM once,
 *     16 bytes per lane, into LDS;
 *   * each wavefront then transforms whole hops: lane l holds the P = N/64 samples n = r*64 + l, converts and
 *     windows them in registers, runs a P-point radix-2 FFT inside the lane (constant twiddles) and multiplies by the
 *     per-lane twiddles W_N^(l*k1); what is left is a 64-point FFT ACROSS the lanes for each of the P values k1.
 * channelizer_fft8_kernel (every fft_size): the 64-point FFT as 8 x 8 -- two radix-8 passes inside the lanes with two 8 x 8
 *     transposes through a per-wavefront LDS buffer between them (l = 8a + b, k2 = c + 8e: DFT over a, twiddle W_64^(bc), DFT over b); the bins
 *     land in the same buffer and the dongle's channel lanes pick theirs up.  40 eight-byte LDS operations and ~190 vector instructions per
 *     512-point hop, where the shuffle kernel below issues 112 ds_bpermute and ~450.  fft_size 1024 ... 8192 run as 2 ... 16 DECIMATED 512-point
 *     transforms per hop, combined for the channels' bins only (see the kernel).
 * channelizer_fft_kernel (AFC's one-hop spectrum launches; tiles whose raw samples leave no LDS for the exchange buffers): six
 *     radix-2 butterfly stages across lanes, exchanging partners with __shfl_xor (ds_bpermute; no LDS storage).  Bin k = k1 + P*k2 ends up in
 *     register bitrev(k1) of lane bitrev6(k2); the (at most 64) channels of the dongle pull their bin with one more shuffle round.
 * Lanes 0..n_ch-1 write |bin| (and re/im for raw-I/Q channels) time-major into the stage-2 rings.
 *
 * Arithmetic: float32, FMA contraction allowed (stage 1 agrees with FFTW's float FFT to ~1e-7 relative, not
 * bit-wise -- no FFT does; see DESIGN.md "parity definition").
 * tests/test_host_fft.py runs this file on the CPU (lanes as fibers) against a float64 FFT.
 */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace airband {

namespace {

/* the workgroup's dynamic LDS (tests/hostshim_wave64 gives the host build a static array instead) */
#if !defined(AB_DYNAMIC_LDS_BYTES)
#define AB_DYNAMIC_LDS_BYTES(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]
#endif

/* Lanes of ONE wavefront exchange data through LDS: a wavefront's LDS operations execute in order, so what is NEEDED is that the compiler keeps them in
 * program order across the exchange (the two wavefront-scope fences around the wave barrier: no instruction).  tests/hostshim_wave64 makes the lanes, which it
 * runs as fibers, meet here. */
#if !defined(AB_WAVE_SYNC)
/* Round 5: every exchange also WAITS until the wavefront's own LDS operations have completed (s_waitcnt lgkmcnt(0)) before any lane reads what another lane wrote.
 * In-order execution of one wavefront's LDS instructions already orders them; the wait takes the kernel off that assumption, at 0 (u8, fft 512) to 2.3 % (CF32, fft 4096) of
 * its time (profiles/r05_misc/fft_*.json, f32_4096_*.json).  It is NOT a fix for round 4's rare wrong transforms: round 5 reproduced those at will -- they need a SECOND PROCESS
 * running this library's long int8 launches on the same GPU, they happen with this wait and with one wavefront per workgroup, they spare the shuffle kernel, and the same kind of
 * fault then hits the main path's CTCSS chain (profiles/r05_event_hunt.md).  One process per GPU: never seen.  What failed turned out to be packed-f32 instructions (lanes 48 - 63 of a
 * result); the library is built without them (_build.py, DEVICE_FLAGS) and the events are gone.  -DAB_WAVE_SYNC_NO_WAIT builds the kernel without the wait. */
#if defined(AB_WAVE_SYNC_NO_WAIT)
#define AB_WAVE_SYNC_EXTRA() (void)0
#else
#define AB_WAVE_SYNC_EXTRA() __builtin_amdgcn_s_waitcnt(0xc07f) /* vmcnt(63) expcnt(7) lgkmcnt(0) */
#endif
#define AB_WAVE_SYNC()                                           \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        AB_WAVE_SYNC_EXTRA();                                    \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#endif

constexpr int HOPS_PER_TILE = 16;
constexpr int XS = 72;                    /* complex values per row of a wavefront's exchange buffer: 64 + 8, so that two rows land in different banks */
constexpr int XBUF_BYTES = 8 * XS * 8;    /* eight rows */

/* bytes of a tile's raw samples in LDS (+ alignment slack), a multiple of 16 */
__host__ __device__ inline long fft_raw_bytes(int fft_log, int hop_samples, int bytes_per_sample) {
    return ((((long)(HOPS_PER_TILE - 1) * hop_samples + (1L << fft_log)) * 2 * bytes_per_sample + 32) + 15) & ~15L;
}
/* the exchange kernel runs where its four exchange buffers fit a CU's LDS beside the tile's raw samples (wide samples at very high sample rates
 * leave no room: those configurations stay on the shuffle kernel, which needs none) */
inline bool fft_uses_exchange(int fft_log, int hop_samples, int bytes_per_sample) {
    /* AIRBAND_HIP_FFT_SHUFFLE=1 in the environment keeps every launch on the shuffle kernel: the A/B partner of the exchange kernel (scripts/next_round_first.sh) */
    static const bool shuffle_only = std::getenv("AIRBAND_HIP_FFT_SHUFFLE") != nullptr;
    return !shuffle_only && fft_raw_bytes(fft_log, hop_samples, bytes_per_sample) + 4 * XBUF_BYTES <= 160 * 1024;
}
constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ int bitrev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

template <int LOGP>
__global__ __launch_bounds__(256) void channelizer_fft_kernel(ChannelizerArgs a) {
    constexpr int P = 1 << LOGP;
    constexpr int N = P * 64;
    AB_DYNAMIC_LDS_BYTES(lds_raw);

    const int tiles = (a.n_hops + HOPS_PER_TILE - 1) / HOPS_PER_TILE;
    const int d = blockIdx.x / tiles, tile = blockIdx.x - d * tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hop0 = tile * HOPS_PER_TILE;
    const int hops_here = min(HOPS_PER_TILE, a.n_hops - hop0);
    const int bps2 = 2 * a.bytes_per_sample; /* bytes per complex sample */
    const DevConst dev = a.dev[d];
    if (dev.disabled) return; /* a failed / disabled dongle (airband_hip_device_enable): block-uniform, in front of every barrier */
    if (a.spectrum_only && !dev.any_afc) return; /* AFC's look at the batch's last hop: only dongles with an AFC channel need it */

    /* ---- stage the tile's raw bytes: coalesced 16 B per lane, HBM -> LDS -------------------------------- */
    const long span_begin = (long)hop0 * a.hop_samples * bps2; /* byte offset inside this batch's span */
    const long span_bytes = ((long)(hops_here - 1) * a.hop_samples + N) * bps2;
    const uint8_t* src = a.iq + (long)d * a.iq_stride + span_begin;
    const long mis = (long)((uintptr_t)src & 15);
    const uint8_t* src_al = src - mis;
    const long n16 = (span_bytes + mis + 15) >> 4;
    /* 16-byte pieces that would start before the dongle's span or end past the bytes the API promises
     * ((first_)batch_bytes + lookahead_bytes) are fetched byte by byte: nothing outside the documented span is touched */
    const long avail_end = ((long)(a.n_hops - 1) * a.hop_samples + N) * bps2 - span_begin + mis; /* relative to src_al */
    const long avail_begin = span_begin == 0 ? mis : 0;
    for (long i = threadIdx.x; i < n16; i += blockDim.x) { /* (256 threads; 64 in the one-hop spectrum launches) */
        const long o = i << 4;
        if (o >= avail_begin && o + 16 <= avail_end) {
            *reinterpret_cast<uint4*>(lds_raw + o) = *reinterpret_cast<const uint4*>(src_al + o);
        } else {
            for (int b = 0; b < 16; b++) lds_raw[o + b] = (o + b >= avail_begin && o + b < avail_end) ? src_al[o + b] : (uint8_t)0;
        }
    }

    /* ---- per-lane constants ------------------------------------------------------------------------------ */
    /* window, pre-multiplied by the sample scale so conversion is one subtract/convert and one multiply:
     * u8 (b - 127.5)/127.5, s8 i/128 (src/rtl_airband.cpp:316-324), s16/f32 x/fullscale (:403,:421) */
    float win[P];
    /* S16 / F32: 1 / input->fullscale of THIS dongle (src/rtl_airband.cpp:403,421) -- two CS16 sources of one handle may differ */
    const float pre = a.sfmt == AIRBAND_SFMT_U8 ? (1.0f / 127.5f) : a.sfmt == AIRBAND_SFMT_S8 ? (1.0f / 128.0f) : dev.scale;
#pragma unroll
    for (int r = 0; r < P; r++) win[r] = a.window[r * 64 + lane] * pre;
    /* per-lane twiddles of the N = P x 64 decomposition, indexed by register (register rho holds k1 = bitrev(rho)): W_N^(lane k1), from the
     * table the host evaluated in double (round 2 called sincosf P + 6 times per block and lane: a third of the kernel's instructions) */
    float twr[P], twi[P];
#pragma unroll
    for (int rho = 0; rho < P; rho++) {
        const int k1 = bitrev(rho, LOGP);
        const float2 w = a.twiddle[(lane * k1) & (N - 1)];
        twr[rho] = w.x;
        twi[rho] = w.y;
    }
    /* cross-lane stage twiddles: distance dd = 32 >> st, W_(2dd)^(lane mod dd) = W_N^((lane mod dd) N / (2 dd)); lanes with the bit clear use 1 */
    float cwr[6], cwi[6];
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int dd = 32 >> st;
        const float2 w = a.twiddle[(lane & (dd - 1)) * (N / (2 * dd))];
        const bool lower = (lane & dd) != 0;
        cwr[st] = lower ? w.x : 1.0f;
        cwi[st] = lower ? w.y : 0.0f;
    }
    /* which (register, lane) holds this lane's channel bin */
    int my_rho = -1, my_src = 0, my_slot = -1;
    bool my_raw = false, my_mag = true; /* NFM channels: stage 2 recomputes |bin| from the raw I/Q */
    if (lane < dev.n_ch) {
        my_slot = a.ext_to_slot[dev.chan_base + lane];
        const int bin = a.cs[my_slot].bin;
        my_rho = bitrev(bin & (P - 1), LOGP);
        my_src = bitrev(bin >> LOGP, 6);
        my_raw = (a.cc[my_slot].flags & AB_F_RAW_IQ) != 0;
        my_mag = (a.cc[my_slot].flags & AB_F_NFM) == 0;
    }
    __syncthreads();

    const uint8_t* lds = lds_raw + mis;
    for (int h = wave; h < hops_here; h += (int)(blockDim.x >> 6)) {
        float xr[P], xi[P];
        const uint8_t* hp = lds + (long)h * a.hop_samples * bps2;
        /* convert + window (src/rtl_airband.cpp:402-455) */
        if (a.sfmt == AIRBAND_SFMT_U8) {
#pragma unroll
            for (int r = 0; r < P; r++) {
                const unsigned v = *reinterpret_cast<const unsigned short*>(hp + 2 * (r * 64 + lane));
                xr[r] = ((float)(v & 0xffu) - 127.5f) * win[r];
                xi[r] = ((float)(v >> 8) - 127.5f) * win[r];
            }
        } else if (a.sfmt == AIRBAND_SFMT_S8) {
#pragma unroll
            for (int r = 0; r < P; r++) {
                const char2 v = *reinterpret_cast<const char2*>(hp + 2 * (r * 64 + lane));
                /* i / 128 for every byte: the reference never initialises its table entry for -128 (src/rtl_airband.cpp:322-324); -1.0
                 * continues the table's own rule (oracle/airband_oracle.c says the same) */
                xr[r] = (float)v.x * win[r];
                xi[r] = (float)v.y * win[r];
            }
        } else if (a.sfmt == AIRBAND_SFMT_S16) {
#pragma unroll
            for (int r = 0; r < P; r++) {
                const short2 v = *reinterpret_cast<const short2*>(hp + 4 * (r * 64 + lane));
                xr[r] = (float)v.x * win[r];
                xi[r] = (float)v.y * win[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < P; r++) {
                const float2 v = *reinterpret_cast<const float2*>(hp + 8 * (r * 64 + lane));
                xr[r] = v.x * win[r];
                xi[r] = v.y * win[r];
            }
        }
        /* in-lane P-point DIF FFT over r (output in bit-reversed register order) */
#pragma unroll
        for (int half = P / 2; half >= 1; half >>= 1) {
#pragma unroll
            for (int base = 0; base < P; base += 2 * half) {
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const int i0 = base + j, i1 = i0 + half;
                    const float ar = xr[i0], ai = xi[i0], br = xr[i1], bi = xi[i1];
                    xr[i0] = ar + br;
                    xi[i0] = ai + bi;
                    const float dr = ar - br, di = ai - bi;
                    /* twiddle W_(2*half)^j : compile-time constant after unrolling */
                    const float ang = -kPi * (float)j / (float)half;
                    const float wc = __builtin_cosf(ang), ws = __builtin_sinf(ang);
                    if (j == 0) {
                        xr[i1] = dr;
                        xi[i1] = di;
                    } else if (2 * j == half) { /* -j */
                        xr[i1] = di;
                        xi[i1] = -dr;
                    } else {
                        xr[i1] = dr * wc - di * ws;
                        xi[i1] = dr * ws + di * wc;
                    }
                }
            }
        }
        /* per-lane twiddles W_N^(lane*k1) */
#pragma unroll
        for (int rho = 1; rho < P; rho++) {
            const float tr = xr[rho] * twr[rho] - xi[rho] * twi[rho];
            const float ti = xr[rho] * twi[rho] + xi[rho] * twr[rho];
            xr[rho] = tr;
            xi[rho] = ti;
        }
        /* 64-point DIF FFT across lanes: six radix-2 stages with __shfl_xor butterflies */
#pragma unroll
        for (int st = 0; st < 6; st++) {
            const int dd = 32 >> st;
            const float sgn = (lane & dd) ? -1.0f : 1.0f;
#pragma unroll
            for (int rho = 0; rho < P; rho++) {
                const float pr_ = __shfl_xor(xr[rho], dd);
                const float pi_ = __shfl_xor(xi[rho], dd);
                /* upper lane: mine + partner ; lower lane: (partner - mine) * w */
                const float tr = fmaf(xr[rho], sgn, pr_);
                const float ti = fmaf(xi[rho], sgn, pi_);
                xr[rho] = tr * cwr[st] - ti * cwi[st];
                xi[rho] = tr * cwi[st] + ti * cwr[st];
            }
        }
        /* AFC looks at the whole spectrum of the batch's last hop (afc.finalize(dev, i, fftout), src/rtl_airband.cpp:626-630) */
        if (a.last_spectrum && hop0 + h == a.n_hops - 1) {
            float2* sp = reinterpret_cast<float2*>(a.last_spectrum) + (long)d * N;
#pragma unroll
            for (int rho = 0; rho < P; rho++) sp[bitrev(rho, LOGP) + P * bitrev(lane, 6)] = make_float2(xr[rho], xi[rho]);
        }
        /* channel lanes fetch their bin (src/rtl_airband.cpp:483-489) */
        float bre = 0.0f, bim = 0.0f;
#pragma unroll
        for (int rho = 0; rho < P; rho++) {
            const float vr = __shfl(xr[rho], my_src);
            const float vi = __shfl(xi[rho], my_src);
            if (rho == my_rho) {
                bre = vr;
                bim = vi;
            }
        }
        if (my_slot >= 0 && !a.spectrum_only) {
            int row = a.row0 + a.first_row + hop0 + h;
            if (row >= a.ring_rows) row -= a.ring_rows;
            const long off = ab_tile_base(my_slot, a.ring_rows / AB_TILE_ROWS) + ab_tile_off(row);
            if (my_mag) a.mag[off] = sqrtf(bre * bre + bim * bim);
            if (my_raw) a.iq_bins[off] = make_float2(bre, bim);
        }
    }
}

typedef float v2f __attribute__((ext_vector_type(2))); /* (re, im).  Until round 5 arithmetic on these became v_pk_*_f32; the library is now built WITHOUT packed-f32 instructions
                                                         (_build.py, DEVICE_FLAGS: beside another process's long launches they leave lanes 48 - 63 wrong now and then), so a pair is two scalar operations */

/* x * w with the twiddle as the pair w = (c, s), wr = i w = (-s, c): (x.re, x.re) * w + (x.im, x.im) * wr -- two multiplies and two FMAs (one packed
 * multiply and one packed FMA in a build with packed-f32 instructions; left to itself the compiler spends five instructions on a complex product: it does not negate one half of a packed operand) */
__device__ __forceinline__ v2f cmul(const v2f x, const v2f w, const v2f wr) { return __builtin_elementwise_fma(x.xx, w, x.yy * wr); }
__device__ __forceinline__ v2f rot_i(const v2f w) { return v2f{-w.y, w.x}; }

/* P-point radix-2 decimation-in-frequency FFT in registers, constant twiddles; output in bit-reversed register order */
template <int P>
__device__ __forceinline__ void fft_dif(v2f (&x)[P]) {
#pragma unroll
    for (int half = P / 2; half >= 1; half >>= 1) {
#pragma unroll
        for (int base = 0; base < P; base += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const int i0 = base + j, i1 = i0 + half;
                const v2f u = x[i0], v = x[i1];
                x[i0] = u + v;
                const v2f d = u - v;
                const float ang = -kPi * (float)j / (float)half; /* W_(2 half)^j: a compile-time constant after unrolling */
                const float wc = __builtin_cosf(ang), ws = __builtin_sinf(ang);
                if (j == 0) x[i1] = d;
                else if (2 * j == half) x[i1] = d.yx * v2f{1.0f, -1.0f}; /* -i: (im, -re), a packed multiply that contracts into the next butterfly's add */
                else x[i1] = cmul(d, v2f{wc, ws}, v2f{-ws, wc});
            }
        }
    }
}

/* LOGM > 0: fft_size = M x 512 (1024 ... 8192).  With n = M n1 + n2 the transform is M transforms of 512 points over the DECIMATED samples,
 *      X[k] = sum over n2 of  W_N^(n2 k) * F_n2[k mod 512],      F_n2[kk] = sum over n1 of x[M n1 + n2] w[M n1 + n2] W_512^(n1 kk),
 * and since only the dongle's channels' bins are wanted, the outer sum is one complex multiply-add per channel lane and n2: a wavefront runs the M
 * 512-point transforms of its hop one after the other with the register footprint of ONE (the shuffle kernel keeps all fft_size / 64 values per lane in
 * registers: 298 VGPRs at 2048 points, scratch memory beyond).  The window is read per transform from a de-interleaved copy of its table (row n2 = the window at samples n2 + M n1: 256 contiguous bytes per
 * load instruction) instead of held in registers. */
template <int LOGP, int LOGM>
__global__ __launch_bounds__(256) void channelizer_fft8_kernel(ChannelizerArgs a) {
    constexpr int P = 1 << LOGP, M = 1 << LOGM;
    constexpr int NS = P * 64;          /* points per transform */
    constexpr int N = NS * M;           /* fft_size */
    static_assert(LOGP == 2 || LOGP == 3, "transforms of 256 or 512 points: four or eight values per lane, ONE exchange round of P 64-point FFTs, eight lanes each");
    static_assert(LOGM == 0 || LOGP == 3, "decimated transforms are 512 points long");
    /* LDS: the four wavefronts' exchange buffers FIRST, the tile's raw samples behind them.  (Round 4, profiles/r04_experiments.md I: with a dozen PROCESSES
     * time-sharing the GPU one transform in ~10^9 came out wrong -- never with one process, 4.6e9 transforms compared bit for bit.  The layout was changed while
     * the events all sat in configurations whose buffers lay above 64 KiB; a later one did not.  It costs nothing and stays; the cause is outside this file.) */
    AB_DYNAMIC_LDS_BYTES(lds_all);
#if defined(AB_FFT_XB_BEHIND) /* experiment builds only (scripts/r05_lds_layout_ab.sh): the layout before round 4's change, for the A/B statistics */
    uint8_t* const lds_raw = lds_all;
#else
    uint8_t* const lds_raw = lds_all + (blockDim.x >> 6) * XBUF_BYTES;
#endif

    const int tiles = (a.n_hops + HOPS_PER_TILE - 1) / HOPS_PER_TILE;
    const int d = blockIdx.x / tiles, tile = blockIdx.x - d * tiles;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* the same number on every lane: hop counters and addresses stay scalar */
    const int hop0 = tile * HOPS_PER_TILE;
    const int hops_here = min(HOPS_PER_TILE, a.n_hops - hop0);
    const int bps2 = 2 * a.bytes_per_sample; /* bytes per complex sample */
    const DevConst dev = a.dev[d];
    if (dev.disabled) return; /* a failed / disabled dongle (airband_hip_device_enable): block-uniform, in front of every barrier */
    if (a.spectrum_only && !dev.any_afc) return; /* AFC's look at the batch's last hop: only dongles with an AFC channel need it */

    /* ---- stage the tile's raw bytes: coalesced 16 B per lane, HBM -> LDS (as in the shuffle kernel below) ---- */
    const long span_begin = (long)hop0 * a.hop_samples * bps2;
    const long span_bytes = ((long)(hops_here - 1) * a.hop_samples + N) * bps2;
    const uint8_t* src = a.iq + (long)d * a.iq_stride + span_begin;
    const long mis = (long)((uintptr_t)src & 15);
    const uint8_t* src_al = src - mis;
    const long n16 = (span_bytes + mis + 15) >> 4;
    const long avail_end = ((long)(a.n_hops - 1) * a.hop_samples + N) * bps2 - span_begin + mis; /* relative to src_al */
    const long avail_begin = span_begin == 0 ? mis : 0;
    for (long i = threadIdx.x; i < n16; i += blockDim.x) {
        const long o = i << 4;
        if (o >= avail_begin && o + 16 <= avail_end) {
            *reinterpret_cast<uint4*>(lds_raw + o) = *reinterpret_cast<const uint4*>(src_al + o);
        } else { /* nothing outside the span the API promises is touched */
            for (int b = 0; b < 16; b++) lds_raw[o + b] = (o + b >= avail_begin && o + b < avail_end) ? src_al[o + b] : (uint8_t)0;
        }
    }

    /* ---- per-lane constants ---- */
    /* window x sample scale (u8 (b - 127.5)/127.5, s8 i/128, s16 / f32 x/fullscale of THIS dongle: src/rtl_airband.cpp:316-324,403,421) */
    const float pre = a.sfmt == AIRBAND_SFMT_U8 ? (1.0f / 127.5f) : a.sfmt == AIRBAND_SFMT_S8 ? (1.0f / 128.0f) : dev.scale;
    float win[P]; /* (M > 1: re-read per transform from the de-interleaved table) */
#pragma unroll
    for (int r = 0; r < P; r++) win[r] = M > 1 ? 0.0f : a.window[r * 64 + lane] * pre;
    /* W_NS^(lane k1) for the k1 = bitrev(rho) register rho holds after the in-lane FFT; from the table of W_N the host evaluated in double (W_NS = W_N^M) */
    v2f tw[P], twr[P];
#pragma unroll
    for (int rho = 0; rho < P; rho++) {
        const float2 w = a.twiddle[((lane * bitrev(rho, LOGP)) & (NS - 1)) * M];
        tw[rho] = v2f{w.x, w.y};
        twr[rho] = rot_i(tw[rho]);
    }
    const int b8 = lane & 7;                /* b in the first radix-8 pass, c in the second */
    const int jj = (lane >> 3) & (P - 1);   /* the k1 (as its register index) this lane works on after the first transpose (256-point transforms: lanes 32 .. 63 repeat the work of lanes 0 .. 31) */
    v2f cw[8], cwr[8]; /* W_64^(b c) for the c = bitrev3(t) register t holds after the first pass */
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float2 w = a.twiddle[(b8 * bitrev(t, 3) * P * M) & (N - 1)];
        cw[t] = v2f{w.x, w.y};
        cwr[t] = rot_i(cw[t]);
    }
    /* the buffer position in which this lane's channel finds bin kk = k mod NS of a transform: row = the register index of kk mod P, column kk / P */
    int my_slot = -1, my_idx = 0, my_bin = 0;
    bool my_raw = false, my_mag = true; /* NFM channels: stage 2 recomputes |bin| from the raw I/Q */
    float* my_mag_ring = a.mag;
    float2* my_iq_ring = a.iq_bins;
    if (lane < dev.n_ch) {
        my_slot = a.ext_to_slot[dev.chan_base + lane];
        my_bin = a.cs[my_slot].bin;
        const int kk = my_bin & (NS - 1);
        my_idx = bitrev(kk & (P - 1), LOGP) * XS + (kk >> LOGP);
        my_raw = (a.cc[my_slot].flags & AB_F_RAW_IQ) != 0;
        my_mag = (a.cc[my_slot].flags & AB_F_NFM) == 0;
        const long base = ab_tile_base(my_slot, a.ring_rows / AB_TILE_ROWS);
        my_mag_ring += base;
        my_iq_ring += base;
    }
#if defined(AB_FFT_XB_BEHIND)
    v2f* xb = reinterpret_cast<v2f*>(lds_all + fft_raw_bytes(a.fft_log, a.hop_samples, a.bytes_per_sample) + (long)wave * XBUF_BYTES);
#else
    v2f* xb = reinterpret_cast<v2f*>(lds_all + (long)wave * XBUF_BYTES);
#endif
    v2f* x_w1 = xb + lane;                     /* [j][lane]: value k1_j of lane l                     */
    v2f* x_r1 = xb + jj * XS + b8;             /* [jj][8 a + b], a = 0 .. 7                           */
    v2f* x_w2 = xb + jj * XS + b8 * 9;         /* [jj][9 b + c]: rows of nine, a transpose without bank conflicts */
    v2f* x_r2 = xb + jj * XS + b8;             /* [jj][9 b + c], b = 0 .. 7 (this lane's c = lane & 7) */
    v2f* x_w3 = xb + jj * XS + b8;             /* [jj][k2 = c + 8 e]                                  */
    __syncthreads();

    const uint8_t* lds = lds_raw + mis;
    /* the hop loop once per sample format (the format is the launch's, not the hop's: chosen once, below, instead of by three scalar branches per transform) */
    auto hops = [&](auto fmt_tag) {
    constexpr int FMT = decltype(fmt_tag)::value;
    for (int h = wave; h < hops_here; h += (int)(blockDim.x >> 6)) {
        const uint8_t* hp = lds + (long)h * a.hop_samples * bps2;
        v2f bin_sum = v2f{0.0f, 0.0f};
#pragma unroll 1
        for (int n2 = 0; n2 < M; n2++) { /* one pass unless the transform is decimated */
            v2f x[P];
            if (M > 1) {
#pragma unroll
                for (int r = 0; r < P; r++) win[r] = a.window_dec[n2 * NS + r * 64 + lane] * pre; /* = window[M (r 64 + lane) + n2], coalesced */
            }
            /* convert + window (src/rtl_airband.cpp:402-455): this lane's samples n = M (r 64 + lane) + n2 */
            if (FMT == AIRBAND_SFMT_U8) {
#pragma unroll
                for (int r = 0; r < P; r++) {
                    const unsigned v = *reinterpret_cast<const unsigned short*>(hp + 2 * (M * (r * 64 + lane) + n2));
                    x[r] = (v2f{(float)(v & 0xffu), (float)(v >> 8)} - 127.5f) * win[r];
                }
            } else if (FMT == AIRBAND_SFMT_S8) {
#pragma unroll
                for (int r = 0; r < P; r++) {
                    const char2 v = *reinterpret_cast<const char2*>(hp + 2 * (M * (r * 64 + lane) + n2));
                    /* i / 128 for every byte: the reference never initialises its table entry for -128 (src/rtl_airband.cpp:322-324); -1.0
                     * continues the table's own rule (oracle/airband_oracle.c says the same) */
                    x[r] = v2f{(float)v.x, (float)v.y} * win[r];
                }
            } else if (FMT == AIRBAND_SFMT_S16) {
#pragma unroll
                for (int r = 0; r < P; r++) {
                    const short2 v = *reinterpret_cast<const short2*>(hp + 4 * (M * (r * 64 + lane) + n2));
                    x[r] = v2f{(float)v.x, (float)v.y} * win[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < P; r++) {
                    const float2 v = *reinterpret_cast<const float2*>(hp + 8 * (M * (r * 64 + lane) + n2));
                    x[r] = v2f{v.x, v.y} * win[r];
                }
            }
            fft_dif<P>(x); /* over r: register rho now holds k1 = bitrev(rho) */
            { /* (all the products, then all the FMAs: a packed-f32 instruction wants a wait state before a dependent one, and written pair by pair the
               * compiler fills it with a no-op) */
#pragma unroll
                for (int g = 0; g < P; g += 4) { /* four at a time: enough independent work for the wait states, few enough registers for four waves per SIMD */
                    v2f t[4];
#pragma unroll
                    for (int rho = g; rho < g + 4; rho++) t[rho - g] = x[rho].yy * twr[rho];
#pragma unroll
                    for (int rho = g; rho < g + 4; rho++)
                        if (rho > 0) x[rho] = __builtin_elementwise_fma(x[rho].xx, tw[rho], t[rho - g]);
                }
            }

            v2f mine = v2f{0.0f, 0.0f};
            {
                /* 64-point FFT over the lanes for each of the P values k1, l = 8 a + b, k2 = c + 8 e */
#pragma unroll
                for (int j = 0; j < P; j++) x_w1[j * XS] = x[j];
                AB_WAVE_SYNC();
                v2f z[8];
#pragma unroll
                for (int i = 0; i < 8; i++) z[i] = x_r1[8 * i];
                AB_WAVE_SYNC(); /* every lane has its eight values: the buffer may be written again */
                fft_dif<8>(z); /* over a: register t holds c = bitrev3(t) */
                {
#pragma unroll
                    for (int g = 0; g < 8; g += 4) {
                        v2f u[4];
#pragma unroll
                        for (int t = g; t < g + 4; t++) u[t - g] = z[t].yy * cwr[t];
#pragma unroll
                        for (int t = g; t < g + 4; t++)
                            if (t > 0) z[t] = __builtin_elementwise_fma(z[t].xx, cw[t], u[t - g]);
                    }
                }
#pragma unroll
                for (int t = 0; t < 8; t++) x_w2[bitrev(t, 3)] = z[t];
                AB_WAVE_SYNC();
#pragma unroll
                for (int i = 0; i < 8; i++) z[i] = x_r2[9 * i];
                AB_WAVE_SYNC();
                fft_dif<8>(z); /* over b: register t holds e = bitrev3(t); the lane is (k1 of register jj, c = lane & 7) */
#pragma unroll
                for (int t = 0; t < 8; t++) x_w3[8 * bitrev(t, 3)] = z[t];
                AB_WAVE_SYNC();
                /* the buffer now holds the transform's bins k1 + P k2 as [register index of k1][k2]: the channel lanes pick theirs up (src/rtl_airband.cpp:483-489) */
                mine = xb[my_idx];
                /* AFC looks at the whole spectrum of the batch's last hop (afc.finalize(dev, i, fftout), src/rtl_airband.cpp:626-630); decimated transforms
                 * leave that to a one-hop launch of the shuffle kernel (launch_channelizer_fft) */
                if (M == 1 && a.last_spectrum && hop0 + h == a.n_hops - 1) {
                    float2* sp = reinterpret_cast<float2*>(a.last_spectrum) + (long)d * N;
#pragma unroll
                    for (int j = 0; j < P; j++) {
                        const v2f v = xb[j * XS + lane];
                        sp[bitrev(j, LOGP) + P * lane] = make_float2(v.x, v.y);
                    }
                }
                AB_WAVE_SYNC(); /* ... before the next transform (or hop) overwrites it */
            }
            if (M == 1) {
                bin_sum = mine;
            } else { /* X[k] += W_N^(n2 k) F_n2[k mod 512], n2 in ascending order */
                const float2 w = a.twiddle[(n2 * my_bin) & (N - 1)];
                const v2f wv = v2f{w.x, w.y};
                bin_sum += cmul(mine, wv, rot_i(wv));
            }
        }
        if (my_slot >= 0 && !a.spectrum_only) {
            int row = a.row0 + a.first_row + hop0 + h;
            if (row >= a.ring_rows) row -= a.ring_rows;
            const int off = ab_tile_off(row);
            /* (v_sqrt_f32, within one ulp, as the matrix-core channelizer takes it: stage 1 is held to 1e-5 of the bins' RMS, and the correctly rounded sqrtf() is sixteen instructions) */
            if (my_mag) my_mag_ring[off] = __builtin_amdgcn_sqrtf(bin_sum.x * bin_sum.x + bin_sum.y * bin_sum.y);
            if (my_raw) my_iq_ring[off] = make_float2(bin_sum.x, bin_sum.y);
        }
    }
    };
    switch (a.sfmt) {
        case AIRBAND_SFMT_U8: hops(std::integral_constant<int, AIRBAND_SFMT_U8>{}); break;
        case AIRBAND_SFMT_S8: hops(std::integral_constant<int, AIRBAND_SFMT_S8>{}); break;
        case AIRBAND_SFMT_S16: hops(std::integral_constant<int, AIRBAND_SFMT_S16>{}); break;
        default: hops(std::integral_constant<int, AIRBAND_SFMT_F32>{}); break;
    }
}

/* threads per workgroup of the exchange kernel: 256 (four wavefronts share a tile's staged samples), or 64 with AIRBAND_HIP_FFT_THREADS=64 in the environment -- ONE wavefront per
 * workgroup (round 5, profiles/r05_event_hunt.md: the kernel's wrong transforms under multi-process GPU sharing do NOT need several wavefronts per workgroup) */
static int fft8_threads() {
    static const int n = [] {
        const char* e = getenv("AIRBAND_HIP_FFT_THREADS");
        return (e && atoi(e) == 64) ? 64 : 256;
    }();
    return n;
}

template <int LOGP, int LOGM, int LOGP_SHUFFLE>
void launch_one(const ChannelizerArgs& a, hipStream_t stream) {
    const int tiles = (a.n_hops + HOPS_PER_TILE - 1) / HOPS_PER_TILE;
    const bool exchange = fft_uses_exchange(a.fft_log, a.hop_samples, a.bytes_per_sample);
    /* a spectrum-only launch (ONE hop and one wavefront per dongle, on a side stream beside the matrix-core channelizer: airband_hip.cpp,
     * launch_last_hop_spectrum) stays on the shuffle kernel at every size: it needs no exchange buffer, so its 6 KiB workgroups fit beside the channelizer's
     * 144 KiB per CU, and one transform per dongle and batch is no time either way */
    const size_t lds = a.spectrum_only ? (size_t)fft_raw_bytes(a.fft_log, a.hop_samples, a.bytes_per_sample) : fft_lds_bytes(a.fft_log, a.hop_samples, a.bytes_per_sample);
    const long blocks = (long)tiles * a.n_dev;
    /* wide formats at high sample rates: opt in to the CU's full 160 KiB (prepare() has checked the upper bound) */
    /* (should the runtime refuse, the launch below fails with hipErrorInvalidValue and the batch driver reports it: airband_hip.cpp checks hipGetLastError) */
    auto shuffle = [&](const ChannelizerArgs& b, long n_blocks) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&channelizer_fft_kernel<LOGP_SHUFFLE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        /* a spectrum-only launch transforms ONE hop per dongle: one wavefront */
        hipLaunchKernelGGL(channelizer_fft_kernel<LOGP_SHUFFLE>, dim3((unsigned)n_blocks), dim3(b.spectrum_only ? 64 : 256), lds, stream, b);
    };
    if (!exchange || a.spectrum_only) {
        shuffle(a, blocks);
        return;
    }
    ChannelizerArgs b = a;
    if (LOGM > 0) b.last_spectrum = nullptr; /* decimated transforms produce the channels' bins only */
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&channelizer_fft8_kernel<LOGP, LOGM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((channelizer_fft8_kernel<LOGP, LOGM>), dim3((unsigned)blocks), dim3(fft8_threads()), lds, stream, b);
    if (LOGM > 0 && a.last_spectrum) { /* ... and AFC's spectrum of the batch's last hop comes from a one-hop, one-wavefront launch of the shuffle kernel */
        ChannelizerArgs c = a;
        c.iq = a.iq + (long)(a.n_hops - 1) * a.hop_samples * 2 * a.bytes_per_sample;
        c.n_hops = 1;
        c.first_row = 0;
        c.row0 = 0;
        c.spectrum_only = 1;
        shuffle(c, a.n_dev);
    }
}

}  // namespace

/* dynamic LDS of one workgroup: the raw bytes of HOPS_PER_TILE consecutive hops (+ alignment slack), and the four wavefronts' exchange buffers where they fit */
size_t fft_lds_bytes(int fft_log, int hop_samples, int bytes_per_sample) {
    return (size_t)fft_raw_bytes(fft_log, hop_samples, bytes_per_sample) + (fft_uses_exchange(fft_log, hop_samples, bytes_per_sample) ? 4 * XBUF_BYTES : 0);
}

void launch_channelizer_fft(const ChannelizerArgs& a, hipStream_t stream) {
    switch (a.fft_log - 6) { /* <points per transform / 64, decimation, the shuffle kernel's values per lane> */
        case 2: launch_one<2, 0, 2>(a, stream); break;
        case 3: launch_one<3, 0, 3>(a, stream); break;
        case 4: launch_one<3, 1, 4>(a, stream); break;
        case 5: launch_one<3, 2, 5>(a, stream); break;
        case 6: launch_one<3, 3, 6>(a, stream); break;
        case 7: launch_one<3, 4, 7>(a, stream); break;
        default: break;
    }
}

}  // namespace airband
