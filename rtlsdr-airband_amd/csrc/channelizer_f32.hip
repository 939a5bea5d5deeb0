/* csrc/channelizer_f32.hip -- stage 1 of the hot path for CF32 dongles (SoapySDR, src/input-soapysdr.cpp:45-64) on the matrix cores.
 *
 * The same observation as channelizer_dft.hip: per hop and channel the reference keeps ONE bin of its fft_size-point FFT
 * (src/rtl_airband.cpp:460 fftwf_execute, :483-489 bin extract),
 *      X[bin] = sum_n  scale x_n w[n] exp(-2 pi i bin n / N),           x_n = I_n + j Q_n as float32   (src/rtl_airband.cpp:421-455)
 * -- with <= 8 channels a [hops x 2N] by [2N x 16] product whose left operand is a sliding view of the raw float stream.  u8 / s8 / CS16 samples
 * are exact in int8 digits; float samples are not bytes, so this kernel contracts in float32 itself: v_mfma_f32_16x16x4_f32, products and sums
 * in float32 -- the error class of a float32 FFT (the reference's own: fftwf), measured against the float64 oracle like every stage-1 path
 * (tests/test_gpu_parity.py, bar 1e-5 relative RMS).  No digit splitting, no offset column: samples are signed, the table holds w[n] cos / sin.
 *
 * Work per sample is 16x the int8 kernel's per byte moved (K = 4 per instruction instead of 64), so this kernel is bound by the float32 matrix pipe,
 * not by HBM: 256 instructions of 32 cycles per 16-hop tile.  Mapping: one WORKGROUP of four waves per work item (dongle, group of 8 channels) and
 * tile range; the four waves share one staged copy of the stream and split the contraction index -- wave p owns window samples [128 p, 128 p + 128),
 * 64 resident B registers -- and one wave (which one rotates with the workgroup) adds the four partial sums (through LDS) and writes the rings.
 *   A (16 hops x 4 values per MFMA): lane l supplies value k = l >> 4 of hop l & 15.  The contraction index is ORDERED so that one 16-byte LDS read
 *     feeds four consecutive MFMAs: MFMA s = 4 j + i of a wave uses the stream values 16 j + 4 (l >> 4) + i of its piece, i = 0 .. 3.
 *   B: [piece][s][lane] floats, built on the host in double with the same ordering (params.cpp, build_f32_tables).
 *   D: lane l holds column l & 15 = (channel, re | im) of hops (l >> 4) * 4 + {0 .. 3}: the int8 kernel's layout, same stores.
 * Staging: all 256 threads load the tile's bytes with coalesced 16-byte global loads one tile AHEAD (registers), and park them in LDS behind the
 * MFMAs of the current tile.  A hop of H samples is 8 H bytes -- 1 280 at 2.56 MS/s, WAVE_RATE 16000: a multiple of the 256 bytes the 64 LDS banks
 * span, so the 16 hop rows of a fragment read would all start in the same bank.  The staged image therefore carries 16 bytes of padding per hop
 * wherever the hop's own length in 16-byte units is even (row pitch = an odd number of 16-byte units: the 16 rows of a read fall in 16 different
 * bank groups).  A window crosses hop boundaries, so a byte's LDS address is  o + pad * floor(o / hop_bytes); for a lane the second term differs
 * from its row number by a constant per fragment read -- sixteen per-lane offsets, computed once.
 */
#include <hip/hip_runtime.h>
#include <atomic>

#include "common.h"
#include "kernels.h"

namespace airband {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4))); /* (not HIP's uint4: an array of that class type ends up in scratch memory) */

constexpr int TILE_HOPS = 16;

/* (hops of an odd number of samples -- 8 mod 16 bytes -- need no padding: their rows do not start in the same bank to begin with) */
__host__ __device__ constexpr int f32_pad(int hop_bytes) { return (hop_bytes & 8) ? 0 : ((hop_bytes / 16) & 1) ? 0 : 16; }
__host__ __device__ constexpr long f32_padded(long o, int hop_bytes, int pad) { return o + (long)pad * (o / hop_bytes); }

/* FFT_N: 256 ... 2048 (window of 2 FFT_N floats = 8 FFT_N bytes); NW: waves per workgroup = pieces of the contraction index (f32_nw: 4 up to fft 512, 8 for 1024 / 2048,
 * round 5); KW = MFMAs per wave and tile = 2 FFT_N / 4 / NW = resident B registers: 32 (fft 256), 64 (512, 1024), 128 (2048) */
/* MAX_LD: 16-byte pieces of a tile per thread (6: tiles up to 24 KiB at NW = 4 -- 2.56 MS/s at WAVE_RATE 16000 -- three workgroups per CU; 12: up to 48 KiB; twice that at NW = 8) */
/* residency the register allocation is held to: NW = 4: three waves per SIMD (168 VGPRs) for the small tiles, one for the large; NW = 8: a workgroup is two waves per SIMD */
/* AL8 (round 6): hops of an ODD number of samples (2.0 MS/s at WAVE_RATE 16000: 125 samples = 1 000 bytes).  A hop is then 8 mod 16 bytes long: the stream is still staged
 * in aligned 16-byte pieces -- the image starts `delta` = 0 or 8 bytes in front of the tile's first hop, the same for every tile of a launch since a tile is 16 hops --
 * and the A fragments are assembled from two 8-byte LDS reads (this kernel waits for the matrix pipe, 128 cycles per fragment, not for LDS). */
/* LAY = 2 (round 6): hops that are whole multiples of 256 bytes (32 samples: 2.56 MS/s at either WAVE_RATE) carry their padding every 256 stream bytes instead of every
 * hop -- byte o of the tile lands at o + 16 floor(o / 256).  A hop is an odd-or-even number m of such units, rows still fall in sixteen different bank groups ((64 + 4) m
 * dwords apart), and floor() now separates: a lane's READS fragment offsets are base + 64 j + 16 (j >> 2) and a thread's parking offsets base + i * const -- immediates
 * of the LDS instructions instead of 32 + 12 registers computed once and kept.  The fft 2048 variants (and fft 4096 / 8192, which run them) spilled 48 - 126 registers. */
template <int FFT_N, int MAX_LD, int NW, int LAY>
__global__ __launch_bounds__(64 * NW, NW == 4 ? (MAX_LD <= 6 ? 3 : 1) : 2) void channelizer_f32_kernel(F32Args a) {
    constexpr bool AL8 = LAY == 1, P256 = LAY == 2, NOPAD = LAY == 3; /* 3: hops of an odd number of 16-byte units need no padding (f32_pad() returns 0 for them): plain offsets */
    constexpr int WIN_BYTES = 8 * FFT_N;
    constexpr int KW = 2 * FFT_N / 4 / NW;   /* 64 (fft 512) or 32 (fft 256) */
    constexpr int READS = KW / 4;            /* 16-byte fragment reads per wave and tile */
    constexpr int PIECE_BYTES = WIN_BYTES / NW;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];

    const int lane = (int)(threadIdx.x & 63);
    const int piece = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    /* XCD-aware placement, as channelizer_dft.hip: 16 neighbouring work items (their output lines interleave) on one XCD */
    const int wg = blockIdx.x;
    const int i_lin = wg % a.n_items;
    const int g128 = i_lin & ~127, in128 = i_lin & 127;
    const int item = ((a.n_items - g128) >= 128) ? g128 + (in128 & 7) * 16 + (in128 >> 3) : i_lin;
    const int split = wg / a.n_items;
    if (split >= a.splits) return;
    const int d = a.item_dev[item], ch0 = a.item_group[item] * 8;
    const DevConst dev = a.dev[d];
    if (dev.disabled) return; /* workgroup-uniform, in front of every barrier */
    const int hop_bytes = a.hop_bytes, pad = a.pad;

    /* MFMA tiles are aligned to the 16-row tiles of the output rings: tile t covers hops [16 t - shift, 16 t - shift + 16) */
    const int shift = (a.row0 + a.first_row) & 15;
    const int ring_tiles = a.ring_rows / AB_TILE_ROWS;
    const int ring_tiles16 = a.ring_rows / TILE_HOPS;
    const int ptile0 = (a.row0 + a.first_row) >> 4;
    const int tiles_total = (shift + a.n_hops + TILE_HOPS - 1) / TILE_HOPS;
    const int tiles_per_split = (tiles_total + a.splits - 1) / a.splits;
    const int t_begin = split * tiles_per_split;
    const int t_end = min(tiles_total, t_begin + tiles_per_split);
    if (t_begin >= t_end) return;

    /* fft_size 4096 / 8192 (round 6): the window in SEGMENTS of 2 048 samples, one launch each -- segment a.seg contracts window samples [2048 seg, 2048 seg + 2048)
     * against its own table, the first parks its (unscaled) sums in a.partial, the ones in between add theirs, the last adds, scales and writes the rings.  The stream is
     * read once per segment: these sizes are bound by the float32 matrix pipe (16x / 32x the instructions of fft 512 per byte), not by the bytes. */
    const uint8_t* src = a.iq + (long)d * a.iq_stride + (long)a.seg * WIN_BYTES;   /* first byte this launch reads of the batch's first hop (16-byte aligned: airband_hip_process_device checks) */
    /* bytes of the batch span (from src) that may be read; odd hops: rounded up to a whole 16-byte piece (the caller's span is: geometry.lookahead_bytes includes the round-up) */
    const long span_end = AL8 ? (((long)(a.n_hops - 1) * hop_bytes + WIN_BYTES + 15) & ~15L) : (long)(a.n_hops - 1) * hop_bytes + WIN_BYTES;
    const int delta = AL8 ? (int)((-(long)shift * hop_bytes) & 15) : 0; /* 0 or 8 */
    const int tile_bytes = (TILE_HOPS - 1) * hop_bytes + WIN_BYTES;     /* stream bytes a tile looks at */
    const int n16 = (tile_bytes + delta + 15) / 16;                     /* (hop_bytes and WIN_BYTES are multiples of 16 -- of 8 with odd hops) */
    const int n_ld = (n16 + 64 * NW - 1) / (64 * NW);
    const int buf_bytes = a.lds_per_buf;
    uint8_t* lds = lds_all;
    float4* exch = reinterpret_cast<float4*>(lds_all + 2 * buf_bytes); /* [tile parity][NW - 1][64] partial sums on their way to wave 0 */

    /* the wave that adds the partial sums up and writes the rings: a different one from workgroup to workgroup -- wave i of every workgroup sits on SIMD i, and
     * with wave 0 everywhere SIMD 0 of a CU would carry the finishing work of all its workgroups */
    const int fin = wg & (NW - 1);

    /* ---- B fragments: KW registers, resident ---- */
    const float* btab = a.btab + (((long)a.item_bset[item] * a.n_seg + a.seg) * NW + piece) * KW * 64 + lane;
    float b[KW];
#pragma unroll
    for (int s = 0; s < KW; s++) b[s] = btab[s * 64];

    const int col = lane & 15, row_l = lane & 15, grp = lane >> 4;
    const int ch = ch0 + (col >> 1);
    const bool ch_valid = ch < dev.n_ch;
    const int slot = a.ext_to_slot[dev.chan_base + (ch_valid ? ch : 0)];
    const unsigned ch_flags = a.cc[slot].flags;
    const bool want_iq = ch_valid && ((ch_flags & AB_F_RAW_IQ) != 0);
    const bool want_mag = !(ch_flags & AB_F_NFM); /* NFM channels: stage 2 recomputes |bin| from the raw I/Q */
    const long slot_base = ab_tile_base(slot, ring_tiles);
    const bool store_lane = !(col & 1) && ch_valid;
    const long lane_off = slot_base + ab_tile_off(grp * 4);
    float* const mag_lane = a.mag + lane_off;
    float2* const iq_lane = a.iq_bins + lane_off;
    constexpr long TILE16_PITCH = (long)TILE_HOPS * AB_SLOT_BLOCK;
    /* scale of the samples: 1 / input->fullscale of THIS dongle (src/rtl_airband.cpp:421), the same number on every lane */
    const float scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(dev.scale)));

    /* the lane's sixteen fragment reads of a tile, as offsets into the padded image: byte o = row hop_bytes + piece PIECE_BYTES + 64 j + 16 grp of the
     * tile's stream lands at o + pad floor(o / hop_bytes) */
    static_assert(!P256 || PIECE_BYTES % 256 == 0, "a piece starts on a padding unit");
    const int abase0 = row_l * hop_bytes + piece * PIECE_BYTES + grp * 16;
    const int abase = abase0 + 16 * (abase0 >> 8); /* LAY = 2 */
    int aoff[READS];
#pragma unroll
    for (int j = 0; j < READS; j++) {
        const int o = abase0 + j * 64;
        aoff[j] = P256 ? abase + 64 * j + 16 * (j >> 2) : AL8 ? o + delta : NOPAD ? o : o + pad * (o / hop_bytes);
    }

    /* ---- staging: registers one tile ahead ---- */
    v4u stage[MAX_LD];
    int poff[MAX_LD]; /* where this thread's pieces go in the padded image: the same for every tile */
#pragma unroll
    for (int i = 0; i < MAX_LD; i++) {
        const int o = (i * 64 * NW + (int)threadIdx.x) * 16;
        poff[i] = P256 ? 16 * ((int)threadIdx.x + ((int)threadIdx.x >> 4)) + i * (1024 + 64) * NW : NOPAD ? o : o + pad * (o / hop_bytes);
    }
    /* (an interior-tile fast path -- one 64-bit add per tile, pieces at constant offsets -- was tried: the second copy of the loads costs registers,
     * three spilled dwords more and the third workgroup per CU of the large-tile variant; 26.2 ms against 22.4.  One path.) */
    auto load_tile = [&](int t) {
        const long base = ((long)t * TILE_HOPS - shift) * hop_bytes - delta; /* a multiple of 16 */
#pragma unroll
        for (int i = 0; i < MAX_LD; i++) {
            if (i >= n_ld) continue; /* (workgroup-uniform; `continue`, not `break`: with a loop exit the compiler keeps stage[] in scratch memory) */
            const int p = i * 64 * NW + (int)threadIdx.x;
            long so = base + (long)p * 16;
            /* pieces past the tile, before the stream's first byte (tile 0 of a batch with shift > 0) or past the span's end (the last tile's hops >= n_hops)
             * are read from the nearest valid piece: they only feed hops that are never stored */
            if (p >= n16) so = base;
            if (so + 16 > span_end) so = span_end - 16;
            if (so < 0) so = 0;
            stage[i] = *reinterpret_cast<const v4u*>(src + so);
        }
    };
    auto park_tile = [&](uint8_t* buf) {
#pragma unroll
        for (int i = 0; i < MAX_LD; i++) {
            if (i >= n_ld) continue;
            const int p = i * 64 * NW + (int)threadIdx.x;
            if (p < n16) *reinterpret_cast<v4u*>(buf + poff[i]) = stage[i];
        }
    };
    auto frag = [&](const uint8_t* p) { /* four consecutive stream values of the lane's hop */
        if (!AL8) return *reinterpret_cast<const v4f*>(p);
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f lo = *reinterpret_cast<const v2f*>(p), hi = *reinterpret_cast<const v2f*>(p + 8);
        return (v4f){lo.x, lo.y, hi.x, hi.y};
    };
    auto pair_swap = [&](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); };

    load_tile(t_begin);
    park_tile(lds);
    __syncthreads();
    for (int t = t_begin; t < t_end; t++) {
        uint8_t* buf = lds + ((t - t_begin) & 1) * buf_bytes;
        if (t + 1 < t_end) load_tile(t + 1); /* flies under this tile's MFMAs */
        /* two accumulators: the instruction issues every 32 cycles per SIMD but a dependent one waits 40 (MI355X_MICROARCH.md) */
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        /* A fragments are fetched THREE reads (twelve MFMAs, ~400 cycles) ahead of the MFMAs that consume them and a scheduling fence after every read's four
         * MFMAs keeps that distance as written: left alone the compiler sinks each read to just in front of its use, and the matrix pipe then idles for an LDS
         * round trip per four instructions (first measurement: 25.3 ms per launch for 13.9 ms of matrix work) */
        constexpr int AHEAD = 3;
        v4f av[AHEAD + 1];
#pragma unroll
        for (int j = 0; j < AHEAD && j < READS; j++) av[j] = frag(buf + aoff[j]);
#pragma unroll
        for (int j = 0; j < READS; j++) {
            if (j + AHEAD < READS) av[(j + AHEAD) % (AHEAD + 1)] = frag(buf + aoff[j + AHEAD]);
            const v4f x = av[j % (AHEAD + 1)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, b[4 * j + 0], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, b[4 * j + 1], acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, b[4 * j + 2], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, b[4 * j + 3], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc += acc1;
        /* the other pieces' partial sums reach wave 0 through LDS; two areas alternate so that a wave a tile ahead never overwrites what wave 0 still adds up */
        float4* ex = exch + (t & 1) * (NW - 1) * 64;
        if (piece != fin) ex[((piece - fin - 1) & (NW - 1)) * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (t + 1 < t_end) park_tile(lds + ((t + 1 - t_begin) & 1) * buf_bytes); /* the other buffer: every wave left it at the previous tile's barrier */
        __syncthreads();
        if (piece != fin) continue;
        float val[4] = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
        for (int q = 0; q < NW - 1; q++) {
            const float4 o = ex[q * 64 + lane];
            val[0] += o.x; val[1] += o.y; val[2] += o.z; val[3] += o.w;
        }
        if (a.n_seg > 1) { /* (launch-uniform) window segments: whole-wave 1 KiB rows of partial sums, one per (work item, tile) */
            float4* row = a.partial + ((long)item * tiles_total + t) * 64 + lane;
            if (a.seg > 0) {
                const float4 o = *row;
                val[0] += o.x; val[1] += o.y; val[2] += o.z; val[3] += o.w;
            }
            if (a.seg + 1 < a.n_seg) {
                *row = make_float4(val[0], val[1], val[2], val[3]);
                continue;
            }
        }
        float im4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            val[r] *= scale;
            im4[r] = pair_swap(val[r]);
        }
        const bool whole_tile = t * TILE_HOPS - shift >= 0 && t * TILE_HOPS - shift + TILE_HOPS <= a.n_hops; /* wave-uniform */
        int pt = ptile0 + t;
        pt = pt >= ring_tiles16 ? pt - ring_tiles16 : pt;
        if (__builtin_expect(whole_tile, 1)) {
            if (store_lane) {
                const long toff = (long)pt * TILE16_PITCH;
                if (want_mag) {
                    v4f m;
#pragma unroll
                    for (int r = 0; r < 4; r++) m[r] = __builtin_amdgcn_sqrtf(val[r] * val[r] + im4[r] * im4[r]); /* v_sqrt_f32, 1 ulp: stage 1 is tolerance-bound */
                    *reinterpret_cast<v4f*>(mag_lane + toff) = m;
                }
                if (want_iq) {
                    v4f qa = {val[0], im4[0], val[1], im4[1]}, qb = {val[2], im4[2], val[3], im4[3]};
                    v4f* q = reinterpret_cast<v4f*>(iq_lane + toff);
                    q[0] = qa;
                    q[1] = qb;
                }
            }
            continue;
        }
        if (store_lane) { /* first / last tile of a batch: hops outside [0, n_hops) are computed and dropped */
            const long off = slot_base + ab_tile_off(pt * TILE_HOPS + grp * 4);
            const int hop_first = t * TILE_HOPS - shift + grp * 4;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int hop = hop_first + r;
                if (hop >= 0 && hop < a.n_hops) {
                    if (want_mag) a.mag[off + r] = __builtin_amdgcn_sqrtf(val[r] * val[r] + im4[r] * im4[r]);
                    if (want_iq) a.iq_bins[off + r] = make_float2(val[r], im4[r]);
                }
            }
        }
    }
}

}  // namespace

/* CF32 at fft_size 256 ... 8192, any hop (an odd number of samples: the AL8 variants), a tile's bytes within the LDS budget of
 * two workgroups per CU; dongles with an AFC channel stay on the wavefront FFT (their tables move at run time: the re-tune kernel builds int8 tables) */
bool f32_supported(int fft_size, int hop_samples, int sfmt) {
    if (sfmt != AIRBAND_SFMT_F32) return false;
    if (fft_size != 256 && fft_size != 512 && fft_size != 1024 && fft_size != 2048 && fft_size != 4096 && fft_size != 8192) return false; /* (4096, 8192: window segments of 2 048 samples, one launch each) */
    if (hop_samples < 8) return false; /* (odd hops: the AL8 variants) */
    const int NW = f32_nw(fft_size);
    if ((TILE_HOPS - 1) * 8 * hop_samples + 8 * f32_seg_size(fft_size) + ((hop_samples & 1) ? 16 : 0) > 12 * 64 * NW * 16) return false; /* a tile's bytes: twelve 16-byte pieces per thread */
    /* two workgroups per CU up to ~78 KiB each (2.56 MS/s at WAVE_RATE 16000: 54 KiB), one beyond (WAVE_RATE 8000: 92 KiB) */
    return f32_lds_per_buf(fft_size, hop_samples) * 2 + 2 * (NW - 1) * 64 * (int)sizeof(float4) <= 154 * 1024;
}

int f32_pad_bytes(int hop_samples) { return f32_pad(8 * hop_samples); }

int f32_lds_per_buf(int fft_size, int hop_samples) {
    const int hop_bytes = 8 * hop_samples, pad = f32_pad(hop_bytes);
    const long tile_bytes = (long)(TILE_HOPS - 1) * hop_bytes + 8 * f32_seg_size(fft_size); /* (a launch stages one window segment) */
    if (f32_layout(fft_size, hop_bytes) == 2) return (int)((tile_bytes + 16 * ((tile_bytes + 255) / 256) + 16 + 255) / 256 * 256); /* 16 bytes of padding per 256 of stream */
    return (int)((f32_padded(tile_bytes, hop_bytes, pad) + 16 + ((hop_bytes & 8) ? 16 : 0) + 255) / 256 * 256); /* (odd hops: the image starts up to 8 bytes in front of the tile) */
}

template <int FFT_N, int MAX_LD, int NW, int AL8>
static void launch_f32_al(const F32Args& a, hipStream_t stream) {
    const long groups = (long)a.n_items * a.splits;
    const size_t lds = (size_t)2 * a.lds_per_buf + 2 * (NW - 1) * 64 * sizeof(float4);
    /* more than the default 64 KiB of dynamic LDS: opt in to the CU's 160 KiB, once per kernel variant AND device (the attribute belongs to the function as loaded
     * on the current device; a process may drive several GPUs) */
    static std::atomic<bool> big_lds[64]; /* (per instantiation; zero-initialised; the shim launches from one thread per GPU: set twice is harmless, torn is not possible) */
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > 64 * 1024 && (dev >= 64 || !big_lds[dev].load(std::memory_order_acquire))) {
        /* (the CU's whole 160 KiB, not this launch's size: a later handle of the same process may have longer hops) */
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&channelizer_f32_kernel<FFT_N, MAX_LD, NW, AL8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess && dev < 64) big_lds[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((channelizer_f32_kernel<FFT_N, MAX_LD, NW, AL8>), dim3((unsigned)groups), dim3(64 * NW), lds, stream, a);
}

template <int FFT_N, int MAX_LD, int NW>
static void launch_f32(const F32Args& a, hipStream_t stream) {
    const int lay = f32_layout(a.fft_size, a.hop_bytes);
    if (lay == 1) launch_f32_al<FFT_N, MAX_LD, NW, 1>(a, stream);
    else if (lay == 2) launch_f32_al<FFT_N, MAX_LD, NW, 2>(a, stream);
    else if (lay == 3) launch_f32_al<FFT_N, MAX_LD, NW, 3>(a, stream);
    else launch_f32_al<FFT_N, MAX_LD, NW, 0>(a, stream);
}

int f32_partial_tiles(int n_hops_max) { return (15 + n_hops_max + TILE_HOPS - 1) / TILE_HOPS + 1; }

void launch_channelizer_f32(const F32Args& a0, hipStream_t stream) {
    F32Args a = a0;
    a.n_seg = f32_n_seg(a0.fft_size);
    a.seg = 0;
    const int tile_bytes = (TILE_HOPS - 1) * a.hop_bytes + 8 * f32_seg_size(a.fft_size) + ((a.hop_bytes & 8) ? 16 : 0); /* (odd hops: the image may start 8 bytes in front of the tile) */
    const bool small = tile_bytes <= 6 * 64 * f32_nw(a.fft_size) * 16; /* six pieces per thread: the register budget of three workgroups per CU (NW = 4) */
    if (a.n_seg > 1) { /* fft_size 4096 / 8192: the fft 2048 kernel once per window segment */
        for (int seg = 0; seg < a.n_seg; seg++) {
            a.seg = seg;
            if (small) launch_f32<2048, 6, 8>(a, stream);
            else launch_f32<2048, 12, 8>(a, stream);
        }
        return;
    }
    a.partial = nullptr;
    if (a.fft_size == 2048) return small ? launch_f32<2048, 6, 8>(a, stream) : launch_f32<2048, 12, 8>(a, stream);
    if (a.fft_size == 1024) return small ? launch_f32<1024, 6, 8>(a, stream) : launch_f32<1024, 12, 8>(a, stream);
    if (a.fft_size == 512) return small ? launch_f32<512, 6, 4>(a, stream) : launch_f32<512, 12, 4>(a, stream);
    return small ? launch_f32<256, 6, 4>(a, stream) : launch_f32<256, 12, 4>(a, stream);
}

}  // namespace airband
