/* csrc/channelizer_fft.hip -- stage 1 of the hot path on gfx950, general variant: sliding windowed FFT per hop.
 *
 * Replaces, for every hop of every dongle (reference: src/rtl_airband.cpp:402-492):
 *   sample -> float (LUT / scale) x window      (:402-455, NEON twin src/rtl_airband_neon.s:28-83)
 *   forward complex FFT of fft_size points      (:460 fftwf_execute, VideoCore twin gpu_fft_execute :458)
 *   per-channel bin magnitude (+ raw bin I/Q)   (:483-489)
 *
 * Mapping (wave64, CDNA4):
 *   * a 256-thread workgroup owns one dongle and a tile of HOPS_PER_TILE consecutive hops; the raw bytes those
 *     hops cover ((T-1)*hop + N samples; consecutive windows overlap by N-hop samples) are fetched from HBM once,
 *     16 bytes per lane, into LDS;
 *   * each wavefront then transforms whole hops: lane l holds the P = N/64 samples n = r*64 + l, converts and
 *     windows them in registers, runs a P-point radix-2 FFT inside the lane (constant twiddles), multiplies by the
 *     per-lane twiddles W_N^(l*k1) and finishes with six radix-2 butterfly stages ACROSS lanes, exchanging
 *     partners with __shfl_xor (ds_bpermute; no LDS storage).  Bin k = k1 + P*k2 ends up in register
 *     bitrev(k1) of lane bitrev6(k2);
 *   * the (at most 64) channels of the dongle pull their bin with one more shuffle round and lanes 0..n_ch-1
 *     write |bin| (and re/im for raw-I/Q channels) time-major into the stage-2 rings.
 *
 * Arithmetic: float32, FMA contraction allowed (stage 1 agrees with FFTW's float FFT to ~1e-7 relative, not
 * bit-wise -- no FFT does; see DESIGN.md "parity definition").
 */
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace airband {

namespace {

/* the workgroup's dynamic LDS (tests/hostshim_wave64 gives the host build a static array instead) */
#if !defined(AB_DYNAMIC_LDS_BYTES)
#define AB_DYNAMIC_LDS_BYTES(name) extern __shared__ __attribute__((aligned(16))) uint8_t name[]
#endif

constexpr int HOPS_PER_TILE = 16;
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

template <int LOGP>
void launch_one(const ChannelizerArgs& a, hipStream_t stream) {
    const int tiles = (a.n_hops + HOPS_PER_TILE - 1) / HOPS_PER_TILE;
    const size_t lds = fft_lds_bytes(a.fft_log, a.hop_samples, a.bytes_per_sample);
    const long blocks = (long)tiles * a.n_dev;
    /* wide formats at high sample rates: opt in to the CU's full 160 KiB (prepare() has checked the upper bound) */
    /* (should the runtime refuse, the launch below fails with hipErrorInvalidValue and the batch driver reports it: airband_hip.cpp checks hipGetLastError) */
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&channelizer_fft_kernel<LOGP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    /* a spectrum-only launch transforms ONE hop per dongle: one wavefront */
    hipLaunchKernelGGL(channelizer_fft_kernel<LOGP>, dim3((unsigned)blocks), dim3(a.spectrum_only ? 64 : 256), lds, stream, a);
}

}  // namespace

/* dynamic LDS of one workgroup: the raw bytes of HOPS_PER_TILE consecutive hops (+ alignment slack) */
size_t fft_lds_bytes(int fft_log, int hop_samples, int bytes_per_sample) {
    return (size_t)(((long)(HOPS_PER_TILE - 1) * hop_samples + (1L << fft_log)) * 2 * bytes_per_sample + 32);
}

void launch_channelizer_fft(const ChannelizerArgs& a, hipStream_t stream) {
    switch (a.fft_log - 6) {
        case 2: launch_one<2>(a, stream); break;
        case 3: launch_one<3>(a, stream); break;
        case 4: launch_one<4>(a, stream); break;
        case 5: launch_one<5>(a, stream); break;
        case 6: launch_one<6>(a, stream); break;
        case 7: launch_one<7>(a, stream); break;
        default: break;
    }
}

}  // namespace airband
