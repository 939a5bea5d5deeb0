/* csrc/channelizer_dft.hip -- stage 1 of the hot path on gfx950, fast variant for u8/s8 dongles with <= 8 channels.
 *
 * The reference computes a full fft_size-point FFT per hop and then reads ONE bin per channel
 * (src/rtl_airband.cpp:460 fftwf_execute, :483-489 bin extract).  Per hop and channel that is
 *      X[bin] = sum_n  lev[b_n] * w[n] * exp(-2 pi i bin n / N)        (lev, w: src/rtl_airband.cpp:316-351)
 * i.e. a length-2N real dot product of the window's raw bytes with a fixed coefficient row.  With <= 8 channels the
 * 16 output components (re, im) x 8 form a [hops x 2N] by [2N x 16] product whose left operand is a sliding view
 * of the raw byte stream -- a dense contraction the matrix cores do at many times the VALU FFT rate, which turns the
 * channelizer from VALU-bound (36-72 flop/byte) into HBM-bound.
 *
 * Exactness: bytes enter as int8 (b - 128, one XOR); the coefficient table is quantised to a 24-bit fixed point
 * number split into three balanced base-256 digits, so each v_mfma_i32_16x16x64_i8 accumulates EXACT integers
 * (|acc| <= 2^24, exact as floats); the three partial sums are recombined with three f32 FMAs and the (b-127.5) offset of the reference's
 * LUT is restored with a per-column constant.  Result = the exact DFT with coefficients rounded at 2^-24 of full
 * scale -- the same class of error as a float FFT (~1e-7 relative), no accumulation round-off at all.
 *
 * u8, s8 and CS16 take this path (f32 is not bytes), at fft_size 256 ... 8192 (windows longer than 512 samples as pieces of 512 on cooperating
 * waves); dongles with more than 8 channels are processed in groups of 8 (one wavefront per group); f32 samples and odd hop sizes use
 * channelizer_fft.hip.
 *
 * Mapping (wave64, CDNA4): one wavefront owns one dongle and a range of 16-hop tiles.
 *   A (16 hops x 64 bytes per MFMA): lane l reads the 16 consecutive stream bytes at hop (l&15), k-chunk (l>>4)
 *     of the current 64-byte step straight from an LDS copy of the raw stream (ds_read_b128); consecutive
 *     hops overlap, so every HBM byte is fetched once (16 B/lane coalesced) and re-used N/hop times from LDS.
 *   B (64 x 16 per MFMA, 3 digits x 16 steps, all-zero edge digits dropped: 44 fragments): 176 VGPRs, loaded once per wave from a per-bin-set table.
 *   D: lane l holds column (l&15) = (channel, re|im) of hops (l>>4)*4 + {0..3}; |bin| needs the neighbour lane.
 */
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

/* Ablation switches (experiment builds only, AIRBAND_EXTRA_DEFINES: the results are WRONG by construction, only the launch time is of interest --
 * profiles/r04_experiments.md): what the kernel costs without its HBM reads (AB_ABL_NO_DMA), without the matrix pipe (AB_ABL_NO_MFMA), without its
 * output stores (AB_ABL_NO_STORE), without the A-fragment reads from LDS (AB_ABL_NO_LDS). */
#if defined(AB_ABL_NO_DMA) || defined(AB_ABL_NO_MFMA) || defined(AB_ABL_NO_STORE) || defined(AB_ABL_NO_LDS)
#define AB_ABLATION 1
#endif

namespace airband {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int TILE_HOPS = 16;

/* staging geometry as a function of the hop size (host and device, usable in constant expressions):
 * a staging step feeds `sub` consecutive 16-hop MFMA tiles; whole 1 KiB DMA pieces, so the last piece of a step may run past the
 * bytes the step needs, never past its buffer */
constexpr __host__ __device__ int c_sub_tiles(int hop_bytes) {
    const int sub = 640 / hop_bytes;
    return sub < 1 ? 1 : (sub > 4 ? 4 : sub);
}
constexpr __host__ __device__ int c_lds_for(int hop_bytes, int sub, int win_bytes) {
    return ((TILE_HOPS * sub - 1) * hop_bytes + win_bytes + ((hop_bytes & 15) ? 16 : 0) + 1023) / 1024 * 1024; /* + the up-to-15 bytes in front of an unaligned step */
}
/* three small buffers (two steps in flight) when eight waves of them fit a CU's 160 KiB, else two larger ones */
/* (np = wavefronts sharing the buffers: fft_size > 512 runs one wave per window piece of 512 samples, 8 / np workgroups per CU -- fewer
 * streams per CU, so a shared step is made two tiles long when three such buffers fit: more bytes in flight, half the barriers) */
/* (experiment builds, profiles/r06_experiments.md F: -DAB_NP_INFLIGHT=1 -- window pieces with two two-tile buffers, ONE step of 10 KiB in flight per workgroup -- and =2 -- three
 * one-tile buffers, two steps of 5 KiB -- against the product's three two-tile buffers, two steps of 10 KiB: how much of these sizes' launch time is bytes in flight.  Little: fft 1024
 * 15.38 -> 15.6 / 16.3 ms, fft 2048 27.2 -> 27.6 / 29.1) */
#ifndef AB_NP_INFLIGHT
#define AB_NP_INFLIGHT 0
#endif
constexpr __host__ __device__ int c_nbuf(int hop_bytes, int win_bytes = 1024, int np = 1) {
    if (AB_NP_INFLIGHT == 1 && np > 1) return 2;
    return 3 * c_lds_for(hop_bytes, 1, win_bytes) * (8 / np) <= 152 * 1024 ? 3 : 2;
}
constexpr __host__ __device__ bool c_long3(int hop_bytes, int win_bytes, int np) {
    if (AB_NP_INFLIGHT != 0) return false;
    return np > 1 && c_sub_tiles(hop_bytes) >= 2 && 3 * c_lds_for(hop_bytes, 2, win_bytes) * (8 / np) <= 152 * 1024;
}
constexpr __host__ __device__ int c_sub(int hop_bytes, int win_bytes = 1024, int np = 1) {
    return c_long3(hop_bytes, win_bytes, np) ? 2 : c_nbuf(hop_bytes, win_bytes, np) == 3 ? 1 : c_sub_tiles(hop_bytes);
}
constexpr __host__ __device__ int c_lds_per_buf(int hop_bytes, int win_bytes = 1024, int np = 1) { return c_lds_for(hop_bytes, c_sub(hop_bytes, win_bytes, np), win_bytes); }

/* EDGE_HI_ZERO: the most significant coefficient digit is zero for every k-step in which the window is below 2^-8
 * (steps 0,1,14,15 of the 7-term cosine window at N = 512 -- checked on the host, see build_dft_tables): those four
 * MFMAs and their 16 VGPRs are dropped. */
/* HOPB: bytes per hop as a compile-time constant (320: 2.56 MS/s at WAVE_RATE 16000, 640: at 8000 -- the BASELINE configurations), or
 * 0 = run-time value.  With the hop known, the staging geometry (pieces per step, buffers, tiles per step) is constant: the DMA loops
 * unroll, the wait-count switch disappears and the scalar bookkeeping around every tile shrinks by about two thirds -- the kernel
 * issues one instruction per SIMD every four cycles whatever its type, so scalar instructions are not free. */
/* S16: the dongle delivers CS16 (SoapySDR, src/input-soapysdr.cpp:45-64): a sample component is lo + 256 * hi with lo the unsigned low byte
 * and hi the signed high byte, so   sum_n s_n c_n  =  sum_n lo_n c_n  +  256 * sum_n hi_n c_n  -- TWO byte planes against the SAME
 * coefficient table.  The planes are pulled apart with v_perm_b32 on the way from LDS to the MFMA (the raw stream interleaves them),
 * the B fragments are the ones of the u8 case, the number of MFMAs per sample doubles and so does the number of bytes per sample. */
/* AL: alignment every hop start is known to have inside the dongle's span (16, 8 or 4 bytes).  2.4 MS/s, the other common RTL-SDR
 * rate, hops 300 (WAVE_RATE 16000) or 600 bytes (8000): the stream is still staged in aligned 16-byte pieces -- the LDS image simply
 * starts up to 15 bytes before the first hop -- and the A fragments are assembled from 8- or 4-byte LDS reads. */
/* s_waitcnt vmcnt(n) for a run-time (wave-uniform) n: the instruction takes an immediate.  Waiting for FEWER operations than n to be
 * outstanding is always safe (it waits longer), so this is a ladder of compares, not a switch: the compiler lowers a 25-way switch to a
 * cascade of ~25 scalar instructions and ten branches per tile, while the counts that occur in the steady state of a launch are one or
 * two values near the top (hi ladder: three staging buffers, transfers two or three steps ahead) or 0..3 (lo ladder: two buffers). */
#define AB_W(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
__device__ __forceinline__ void wait_vmcnt_lo(int n) {
    if (n <= 0) AB_W(0);
    else if (n == 1) AB_W(1);
    else if (n == 2) AB_W(2);
    else if (n == 3) AB_W(3);
    else if (n < 6) AB_W(4);
    else if (n < 8) AB_W(6);
    else if (n < 12) AB_W(8);
    else AB_W(12);
}
__device__ __forceinline__ void wait_vmcnt(int n) {
    if (n >= 18) AB_W(18);
    else if (n >= 16) AB_W(16);
    else if (n >= 14) AB_W(14);
    else if (n >= 12) AB_W(12);
    else if (n >= 10) AB_W(10);
    else if (n >= 8) AB_W(8);
    else if (n >= 6) AB_W(6);
    else wait_vmcnt_lo(n);
}
#undef AB_W

__device__ __forceinline__ v4i ab_mfma(v4i x, v4i b, v4i acc) {
#if defined(AB_ABL_NO_MFMA)
    asm volatile("" ::"v"(x), "v"(b)); /* the operands stay alive (their loads are not optimised away), the matrix pipe stays idle */
    acc.x ^= x.x;
    return acc;
#else
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(x, b, acc, 0, 0, 0);
#endif
}

template <int AL>
__device__ __forceinline__ v4i lds_read16(const uint8_t* p) {
#if defined(AB_ABL_NO_LDS)
    v4i z = {(int)(uintptr_t)p, 1, 2, 3};
    asm volatile("" : "+v"(z));
    return z;
#endif
    if (AL >= 16) return *reinterpret_cast<const v4i*>(p);
    /* (round 5, measured and dropped: ONE unaligned ds_read_b128 per fragment instead of the assembled reads below -- gfx950 under HSA runs with unaligned DS
     * access and the compiler itself emits that instruction for a 16-byte LDS load of alignment 2 -- is correct and 58 % SLOWER at 250-byte hops: 13.97 against
     * 8.85 ms per launch, profiles/r05_misc/r2000k_*.json; a misaligned wide LDS access is served a few bytes at a time) */
    if (AL == 8) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i lo = *reinterpret_cast<const v2i*>(p), hi = *reinterpret_cast<const v2i*>(p + 8);
        return (v4i){lo.x, lo.y, hi.x, hi.y};
    }
    if (AL == 4) {
        const int* q = reinterpret_cast<const int*>(p);
        return (v4i){q[0], q[1], q[2], q[3]};
    }
    /* AL == 2: any even address (u8 / s8 hops of an odd number of samples: 2.0 MS/s at WAVE_RATE 16000 is 125 samples = 250 bytes).  The five
     * aligned dwords that hold the 16 bytes, funnelled through v_alignbyte_b32 with the lane's own byte offset (0 or 2): 5 LDS reads + 4 vector
     * instructions per fragment where the aligned variants need one read.  The last dword's upper bytes lie past the fragment and are shifted out. */
    const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(p) & 3u);
    const int* q = reinterpret_cast<const int*>(p - sh); /* (pointer arithmetic, not an integer round trip: the compiler must keep seeing an LDS address -- a flat load would also count in vmcnt) */
    const unsigned d0 = (unsigned)q[0], d1 = (unsigned)q[1], d2 = (unsigned)q[2], d3 = (unsigned)q[3], d4 = (unsigned)q[4];
    return (v4i){(int)__builtin_amdgcn_alignbyte(d1, d0, sh), (int)__builtin_amdgcn_alignbyte(d2, d1, sh), (int)__builtin_amdgcn_alignbyte(d3, d2, sh),
                 (int)__builtin_amdgcn_alignbyte(d4, d3, sh)};
}

/* NP: window pieces of FFT_N samples = wavefronts per workgroup.  fft_size 1024 / 2048:
 *   X[bin] = sum over pieces p of  sum_{n in piece p} x[n] w[n] e^{-2 pi i bin n / N}
 * -- every piece is a 512-sample contraction of the same kind with its own coefficient table, 32 / 64 k-steps of B fragments do not
 * fit one wave's registers, so wave p of the workgroup holds piece p's table.  The waves share ONE staged copy of the stream (wave 0
 * runs the DMA; a workgroup barrier hands each step over), compute their piece's partial sums for the same 16 hops, and wave 0
 * adds the partials up (through LDS) and writes the outputs: the stream is read once whatever the window length. */
template <int FFT_N, bool EDGE_HI_ZERO, int HOPB, bool S16, int AL, int NP = 1>
__global__ __launch_bounds__(64 * NP, NP <= 4 ? 2 : 1) void channelizer_dft_kernel(DftArgs a) {
    constexpr int BPS = S16 ? 2 : 1;              /* bytes per sample component        */
    constexpr int WIN_BYTES = 2 * FFT_N * BPS;    /* bytes per window piece            */
    constexpr int WIN_ALL = WIN_BYTES * NP;       /* bytes per window                  */
    static_assert(NP == 1 || (HOPB == 0 && FFT_N == 512), "window pieces are 512 samples long; the hop-specialised variants are single-piece");
    constexpr int KSTEPS = 2 * FFT_N / 64;        /* MFMA k-steps per window and plane */
    static_assert(KSTEPS == 16 || KSTEPS == 8, "B fragments (3 digits x KSTEPS x 4 VGPRs) must fit beside everything else: fft_size 256 or 512");
    static_assert(HOPB == 0 || (FFT_N == 512 && !S16 && AL == 16), "the hop-specialised variants are built for u8 at fft_size 512, 16-byte aligned hops");
    constexpr int EDGE = KSTEPS / 8; /* k-steps at either end of the window whose most significant coefficient digit may be all zero */
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];

    /* one wavefront per workgroup: waves share nothing, and a 64-thread block lets the LDS budget (two staging
     * buffers per wave) rather than the block shape decide how many waves a CU holds */
    const int lane = NP > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    /* wave p of the workgroup: window piece p -- the same number on every lane of the wave; told so, the compiler keeps everything that is
     * decided per piece (who runs the transfers, the store accounting of the waits) in scalar registers and scalar branches */
    const int piece = NP > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    /* XCD-aware placement: workgroup b runs on XCD b % 8 (observed dispatch order; speed only).  16 consecutive
     * dongles write neighbouring slots of the same 128-byte lines, so they are given to the SAME XCD and meet in
     * one L2: inside every group of 128 dongles, workgroup i*8 + x takes dongle x*16 + i. */
    const int wave_global = blockIdx.x;
    /* a work item = (dongle, group of 8 channels): dongles with more than 8 channels appear once per group, side by side, so the
     * groups of a dongle stream the same bytes at the same time through the same L2 */
    const int i_lin = wave_global % a.n_items;
    const int g128 = i_lin & ~127, in128 = i_lin & 127;
    const int item = ((a.n_items - g128) >= 128) ? g128 + (in128 & 7) * 16 + (in128 >> 3) : i_lin;
    const int split = wave_global / a.n_items;
    if (split >= a.splits) return;
    const int d = a.item_dev[item], ch0 = a.item_group[item] * 8;
    if (a.dev[d].disabled) return; /* a failed / disabled dongle (airband_hip_device_enable): workgroup-uniform, in front of every barrier */
    const int hop_bytes = HOPB ? HOPB : a.hop_bytes;
    /* a staging step feeds `sub` consecutive 16-hop MFMA tiles: ~10 KiB of stream per step whatever the hop size, so
     * the bytes a wave keeps in flight (one step ahead) do not shrink when the hop does */
    const int sub = HOPB ? c_sub(HOPB ? HOPB : 64) : a.sub;
    const int lds_per_buf = HOPB ? c_lds_per_buf(HOPB ? HOPB : 64) : a.lds_per_buf;
    const int step_hops = TILE_HOPS * sub;
    const int buf_bytes = (step_hops - 1) * hop_bytes + WIN_ALL + (AL >= 16 ? 0 : 16);
    uint8_t* lds = lds_all;                               /* two buffers of lds_per_buf bytes */

    /* MFMA tiles are aligned to the 16-row tiles of the output rings: tile t covers hops [16 t - shift, 16 t - shift + 16);
     * hops < 0 (first tile) and >= n_hops (last tile) are computed on whatever bytes are there and never stored */
    const int shift = (a.row0 + a.first_row) & 15;
    /* hops that are not multiples of 16 bytes: a step's first byte sits `delta` bytes into its (16-byte aligned) LDS image; a step
     * advances by 16 hops, so delta is the same for every step of the wave */
    const uint8_t* src = a.iq + (long)d * a.iq_stride + (long)a.piece0 * WIN_BYTES;    /* first byte this pass reads of the batch's first hop */
    /* Hops that are not multiples of 16 bytes: a batch of such hops need not START on a 16-byte boundary either (250-byte hops: the second batch of a
     * stream begins 2 100 hops = 525 000 bytes in).  The transfers move aligned 16-byte pieces, so the stream is addressed from the aligned byte at or
     * in front of its first one and every position below carries the `mis` bytes in between. */
    const int mis = AL >= 16 ? 0 : (int)(reinterpret_cast<uintptr_t>(src) & 15u);
    src -= mis;
    const int delta = AL >= 16 ? 0 : (int)((-(long)shift * (HOPB ? HOPB : a.hop_bytes) + mis) & 15);
    const int ring_tiles = a.ring_rows / AB_TILE_ROWS;
    const int ring_tiles16 = a.ring_rows / TILE_HOPS; /* the ring length is a whole number of 16-hop MFMA tiles */
    const int ptile0 = (a.row0 + a.first_row) >> 4;
    const int tiles_total = (shift + a.n_hops + TILE_HOPS - 1) / TILE_HOPS;
    const int steps_total = (tiles_total + sub - 1) / sub;
    const int steps_per_split = (steps_total + a.splits - 1) / a.splits;
    const int st_begin = split * steps_per_split;
    const int st_end = min(steps_total, st_begin + steps_per_split);
    if (st_begin >= st_end) return;


    /* bytes of the batch span that may be read, rounded up to whole 16-byte pieces (geometry.lookahead_bytes includes the round-up).
     * fft_size 8192 runs as two passes of 8 window pieces (a.piece0 = 0, 8): a pass streams the part of every window that starts
     * piece0 pieces in, so its addresses are relative to that byte and the readable span is what is left of the window behind it */
    const long span_end = ((long)(a.n_hops - 1) * hop_bytes + (long)(a.np_total - a.piece0) * WIN_BYTES + mis + 15) & ~15L;

    /* ---- B fragments: 3 digits x 16 k-steps, resident for the whole wave ---------------------------------- */
    /* fft_size > 512: each window piece has its own coefficient table */
    const int bset = a.item_bset[item] * a.np_total + a.piece0 + piece;
    const v4i* btab = reinterpret_cast<const v4i*>(a.bfrag) + (long)bset * 3 * KSTEPS * 64 + lane;
    v4i b0[KSTEPS], b1[KSTEPS], b2[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
        b0[s] = btab[(0 * KSTEPS + s) * 64];
        b1[s] = btab[(1 * KSTEPS + s) * 64];
        if (!(EDGE_HI_ZERO && (s < EDGE || s >= KSTEPS - EDGE))) b2[s] = btab[(2 * KSTEPS + s) * 64];
    }
    const int col = lane & 15;
    const double corr = S16 ? a.corr[bset * 16 + col] * 256.0 : a.corr[bset * 16 + col]; /* table holds 0.5 * sum c (u8: b - 127.5 = (b - 128) + 0.5); CS16 needs 128 * sum c */
    const int ch = ch0 + (col >> 1);
    const DevConst dev = a.dev[d];
    /* u8: a.unscale = 1 / (table scale * 127.5); CS16: 1 / (table scale * this dongle's input->fullscale) (src/rtl_airband.cpp:403) */
    const double unscale = S16 ? a.unscale * (double)dev.scale : a.unscale;
    const bool ch_valid = ch < dev.n_ch;
    const int slot = a.ext_to_slot[dev.chan_base + (ch_valid ? ch : 0)];
    const unsigned ch_flags = a.cc[slot].flags;
    const bool want_iq = ch_valid && ((ch_flags & AB_F_RAW_IQ) != 0);
    /* NFM channels never read |bin| back except as sqrt(re^2 + im^2) of the very I/Q written next to it: stage 2 recomputes it,
     * and stage 1 saves the bytes (writes are the expensive half of this kernel's HBM traffic) */
    const bool want_mag = !(ch_flags & AB_F_NFM);
    const long slot_base = ab_tile_base(slot, ring_tiles);

    /* ---- raw-byte staging: HBM -> LDS without a register round trip (global_load_lds_dwordx4: every lane
     * supplies its own 16-byte source address, the wave's data lands contiguously at an M0-relative LDS base).
     * A step needs buf_bytes of stream; its first WIN_BYTES - hop_bytes were already fetched for the previous step,
     * so that part is an L2 hit -- HBM sees every byte once. Lanes past the end of
     * the batch span re-read its last 16 bytes: they only feed hops >= n_hops, which are never stored. */
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    /* Cache policy of the transfers (the instruction's aux field; 2 = nt, non-temporal).  A staging step's FIRST piece holds the window overlap the previous step
     * fetched (its last piece: the same 1 KiB at hops of 320 bytes) and its LAST piece is what the next step fetches again -- those two are re-read from L2 and keep
     * the default policy; the pieces in between are read exactly once by exactly one CU and go out nt (MI355X_MICROARCH.md "nt-weights" / "ldsdma-fill": read-once
     * LDS-DMA streams 6.4 -> 6.5 - 6.8 TB/s chip-wide).  Round 6, interleaved on three boxes (profiles/r06_experiments.md A, profiles/r06_nt/): hops of 640 bytes
     * (nine of eleven pieces read once) 7.82 -> 7.33 ms per launch on every box; hops of 320 bytes (four of six) -5 %, -3 % and +1 %; EVERY piece nt: +13 % (the
     * overlap is then fetched from memory twice); nt on the first piece as well, nt + sc0, nt + sc1: the same as nt; sc1 / sc0 sc1 alone: nothing. */
#ifndef AB_DMA_NT_AUX
#define AB_DMA_NT_AUX 2 /* interior (read-once) pieces of the hop-specialised variants */
#endif
#ifndef AB_DMA_EDGE_AUX
#define AB_DMA_EDGE_AUX 0 /* first and last piece of a step (the L2-hit overlap), and every piece of the run-time-hop variants */
#endif
#if defined(AB_ABL_NO_DMA)
#define AB_DMA(G, L, OFF, AUX) asm volatile("" ::"v"(G), "v"(L))
#else
#define AB_DMA(G, L, OFF, AUX) __builtin_amdgcn_global_load_lds((G), (L), 16, (OFF), (AUX))
#endif
    const int n_dma = (buf_bytes + 1023) >> 10;
    /* the same number as a constant where the hop is one (pieces per step of the hop-specialised variants: 6 at 320 bytes, 11 at 640) */
    constexpr int N_DMA_C = HOPB ? ((TILE_HOPS * c_sub(HOPB ? HOPB : 64) - 1) * (HOPB ? HOPB : 64) + WIN_ALL + 1023) >> 10 : 0;
#ifndef AB_DMA_FIRST_AUX
#define AB_DMA_FIRST_AUX AB_DMA_EDGE_AUX /* a step's first piece: the LAST use of the overlap bytes */
#endif
#ifndef AB_DMA_LAST_AUX
#define AB_DMA_LAST_AUX AB_DMA_EDGE_AUX /* a step's last piece: the FIRST use of the bytes the next step reads again */
#endif
#define AB_PIECE_AUX(K) (!HOPB ? AB_DMA_EDGE_AUX : (K) == 0 ? AB_DMA_FIRST_AUX : (K) + 1 >= N_DMA_C ? AB_DMA_LAST_AUX : AB_DMA_NT_AUX)
    auto stage = [&](int step, uint8_t* buf) {
        const long base = ((long)step * step_hops - shift) * hop_bytes + mis;
        if (HOPB && base >= 0 && base + (long)n_dma * 1024 <= span_end) {
            /* interior step (wave-uniform test): every piece lies inside the batch span, so no lane needs its address clamped -- one
             * 64-bit add per step, the pieces differ only in the instruction's immediate offset, which moves the global source and
             * the LDS destination alike.  The offset field holds 12 bits: a second base covers the pieces past 4 KiB. */
            const uint8_t* p = src + base + lane * 16;
#define AB_PIECE(K, BASE, OFF) \
    if (n_dma > (K)) AB_DMA((gptr_t)(p + (BASE)), (lptr_t)(uintptr_t)(buf + (BASE)), (OFF), AB_PIECE_AUX(K))
            AB_PIECE(0, 0, 0); AB_PIECE(1, 0, 1024); AB_PIECE(2, 0, 2048); AB_PIECE(3, 0, 3072);
            AB_PIECE(4, 4096, 0); AB_PIECE(5, 4096, 1024); AB_PIECE(6, 4096, 2048); AB_PIECE(7, 4096, 3072);
            AB_PIECE(8, 8192, 0); AB_PIECE(9, 8192, 1024); AB_PIECE(10, 8192, 2048); AB_PIECE(11, 8192, 3072);
#undef AB_PIECE
            return;
        }
        const long base_al = AL >= 16 ? base : ((base >> 4) << 4); /* floor to 16 bytes, also below zero */
        for (int i = 0; i < n_dma; i++) {
            long so = base_al + i * 1024 + lane * 16;
            if (so + 16 > span_end) so = span_end - 16;
            if (so < 0) so = 0;
            AB_DMA((gptr_t)(src + so), (lptr_t)(uintptr_t)(buf + i * 1024), 0, AB_DMA_EDGE_AUX);
        }
    };
    /* staging ring of nbuf buffers: step st lives in buffer (st - st_begin) % nbuf; nbuf - 1 steps are in flight */
    const int nbuf = HOPB ? c_nbuf(HOPB ? HOPB : 64) : a.nbuf;
    /* partial sums of pieces 1 .. NP-1 on their way to wave 0: [tile parity][piece - 1][lane] x 4 floats, behind the staging buffers */
    float4* exch = reinterpret_cast<float4*>(lds_all + nbuf * lds_per_buf);
    /* Vector-memory operations complete in issue order, so "at most N outstanding" proves a transfer has landed as soon as N operations
     * YOUNGER than it are known to have been issued: the pieces of later transfers and the output stores issued since.  Leaving the stores
     * out of N -- they are the youngest of all -- would make every wait sit out their write acknowledgements.  `stores` counts the store
     * instructions issued so far (only those of whole tiles: fewer than the truth is safe), mark[b] what it stood at when the transfer
     * into buffer b was issued. */
    int stores = 0, mark0 = 0, mark1 = 0, mark2 = 0; /* (three scalars, not an array: a run-time index would send it to scratch memory) */
    auto mark_get = [&](int b) { return b == 0 ? mark0 : b == 1 ? mark1 : mark2; };
    auto mark_set = [&](int b, int v) {
        mark0 = b == 0 ? v : mark0;
        mark1 = b == 1 ? v : mark1;
        mark2 = b == 2 ? v : mark2;
    };
    /* ---- round 6: the hop-specialised variants stage through ONE ring without the window overlap (below, "staging ring") ---- */
#ifndef AB_RING
#define AB_RING 0 /* 1: experiment builds (-DAB_RING=1).  Measured, parity-green and NOT adopted (profiles/r06_ring/, profiles/r06_experiments.md J): 8 % fewer L1 -> L2 read
                   * requests, 18 % fewer L2 hits, the same bytes fetched -- and the same launch time, 8.31 / 8.42 / 8.95 against 8.34 / 8.42 / 8.43 ms (hops of 640 bytes:
                   * 7.37 / 7.36 / 7.40 against 7.35 / 7.35 / 7.41): the kernel waits for HBM, not for its requests.  The product stays on round 5's three buffers. */
#endif
#ifndef AB_RING_640
#define AB_RING_640 1 /* hops of 640 bytes (WAVE_RATE 8000): three slots of 10 KiB = 31 KiB per wave, five waves per CU (round 5: two buffers of 11 KiB, seven) */
#endif
    constexpr bool RING = AB_RING != 0 && HOPB != 0 && (HOPB != 640 || AB_RING_640 != 0) && NP == 1 && !S16 && AL >= 16 && (TILE_HOPS * (HOPB ? HOPB : 64)) % 1024 == 0;
    if (!RING && piece == 0) {
        stage(st_begin, lds);
        if (nbuf == 3 && st_begin + 1 < st_end) stage(st_begin + 1, lds + lds_per_buf);
    }
    int cur = 0;

    const int row_l = lane & 15, grp = lane >> 4;
    /* store instructions a whole tile issues (each is skipped when no lane wants it): |bin| = one 16-byte store, raw I/Q = two */
    const int k_tile = ((__ballot(!(col & 1) && ch_valid && want_mag) != 0ull) ? 1 : 0) + ((__ballot(!(col & 1) && ch_valid && want_iq) != 0ull) ? 2 : 0);

    /* Recombination of the digit sums in single precision: every accumulator is an exact integer below 2^24 (exact as a float), and
     *     value = ((acc2 * 2^16 + acc1 * 2^8 + acc0) + corr) * unscale          [+ 2^8 * the same of the high-byte plane for CS16]
     * is three fused multiply-adds with the constants folded -- rounding at 2^-24 of partial sums that are never larger than the
     * result's own full scale, the same error class as the final conversion to float (measured against the float64 oracle:
     * tests/test_gpu_parity.py, stage-1 bar 1e-5 relative RMS).  Round 2 recombined in float64: 36 half- and quarter-rate
     * instructions per tile against 12 full-rate ones. */
    /* (the scale factors are the same number on every lane -- a.unscale, and for CS16 the wave's own dongle's 1 / fullscale: scalar registers) */
    const int flipmask = a.sfmt == AIRBAND_SFMT_S8 ? 0 : (int)0x80808080;
    auto uni = [](double v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)v))); };
    const float u0 = uni(unscale), u1 = uni(unscale * 256.0), u2 = uni(unscale * 65536.0);
    const float cu = a.sfmt == AIRBAND_SFMT_S8 ? 0.0f : (float)(corr * unscale); /* s8 samples are i / 128: nothing to restore */
    const float w0 = u1, w1 = u2, w2 = uni(unscale * 16777216.0); /* CS16 high-byte plane */

    struct TileAcc {
        v4i a0, a1, a2, h0, h1, h2; /* h*: CS16 high-byte plane */
    };
    /* LDS -> MFMA for the 16 hops of tile (step buffer `buf`, sub-tile sb): leaves the integer digit sums in A */
    /* u8 / s8: the lane's row of the staged stream, and the first four k-steps' A fragments -- split off so that the pipelined loop can
     * issue them for tile t + 1 before it recombines and stores tile t */
    auto a_row = [&](const uint8_t* buf, int sb) { return buf + delta + (sb * TILE_HOPS + row_l) * hop_bytes + grp * 16 + piece * WIN_BYTES; };
    auto a_head = [&](const uint8_t* arow, v4i* pre) {
        pre[0] = lds_read16<AL>(arow);
        pre[1] = lds_read16<AL>(arow + 64);
        pre[2] = lds_read16<AL>(arow + 128);
        pre[3] = lds_read16<AL>(arow + 192);
    };
    /* k-step behind which the multi-piece loop finishes the previous tile: late enough that the A-fragment registers fetched ahead are free again (the B fragments of a
     * window piece fill 192 registers: in the middle of the sequence the finish costs spills), early enough that MFMAs are still in the pipe while it runs.  Measured
     * (profiles/r06_fft_pieces/, per launch at 65 536 dongles, before -> 13 | 7 | 11 | 15): fft 1024 15.44 -> 15.2 | 16.2 | 16.1 | 15.3 ms, fft 2048 27.1 -> 26.8 | 29.3 |
     * 28.6 | 26.8; fft 4096 56.8 -> 53.4, fft 8192 124.0 -> 118.5 (13 only). */
#ifndef AB_MID_AT
#define AB_MID_AT (KSTEPS - 3)
#endif
    auto tile_body = [&](const uint8_t* arow, const v4i* pre, TileAcc& A, auto&& mid) {
        A.a0 = (v4i){0, 0, 0, 0}; A.a1 = (v4i){0, 0, 0, 0}; A.a2 = (v4i){0, 0, 0, 0};
        /* A fragments are fetched four k-steps ahead of the MFMAs that consume them, so the LDS latency (and the 2-way bank conflict of
         * the strided rows) hides behind a dozen MFMAs instead of stalling in front of them; a scheduling fence every two k-steps keeps
         * that distance as the source spells it out */
        v4i av[KSTEPS];
        av[0] = pre[0]; av[1] = pre[1]; av[2] = pre[2]; av[3] = pre[3];
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            if ((s & 1) == 0 && s + 4 < KSTEPS) {
                av[s + 4] = lds_read16<AL>(arow + (s + 4) * 64);
                av[s + 5] = lds_read16<AL>(arow + (s + 5) * 64);
            }
            v4i x = av[s];
            x.x ^= flipmask; x.y ^= flipmask; x.z ^= flipmask; x.w ^= flipmask; /* u8 -> b - 128 as int8; s8 (mirisdr, SoapySDR CS8) is int8 already: the mask is zero */
            A.a0 = ab_mfma(x, b0[s], A.a0);
            A.a1 = ab_mfma(x, b1[s], A.a1);
            if (!(EDGE_HI_ZERO && (s < EDGE || s >= KSTEPS - EDGE))) A.a2 = ab_mfma(x, b2[s], A.a2);
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
            if (s == AB_MID_AT) { /* most of the tile's MFMAs are in the pipe: what the caller wants done under them (window pieces: the PREVIOUS tile's sums and stores) */
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto no_mid = [] {};
    auto tile_mfma = [&](const uint8_t* buf, int sb, TileAcc& A, auto&& mid) {
        if (!S16) {
            const uint8_t* arow = a_row(buf, sb);
            v4i pre[4];
            a_head(arow, pre);
            tile_body(arow, pre, A, mid);
        } else {
            /* CS16: plane k-step s of the lane = 16 plane bytes = 8 samples x (I, Q) = 32 raw bytes [Ilo Ihi Qlo Qhi] x 8 */
            /* (round 5, measured and dropped: pulling the planes apart ONCE per landed step, in place in LDS -- 32 raw bytes -> [16 low bytes - 128 | 16 high bytes], 96
             * instructions per step instead of 192 v_perm / v_xor per tile -- is correct and 8 % SLOWER: 21.05 against 19.48 ms per launch, profiles/r05_misc/cs16_*.json.
             * Like the u8 flip moved into LDS in round 3: on this kernel vector instructions are cheaper than LDS traffic) */
            A.a0 = (v4i){0, 0, 0, 0}; A.a1 = (v4i){0, 0, 0, 0}; A.a2 = (v4i){0, 0, 0, 0};
            A.h0 = (v4i){0, 0, 0, 0}; A.h1 = (v4i){0, 0, 0, 0}; A.h2 = (v4i){0, 0, 0, 0};
            const uint8_t* arow = buf + delta + (sb * TILE_HOPS + row_l) * hop_bytes + grp * 32 + piece * WIN_BYTES;
            v4i ra[2], rb[2]; /* raw 32 bytes of k-step s in (ra, rb)[s & 1]; the next k-step is fetched under this one's MFMAs */
            ra[0] = lds_read16<AL>(arow);
            rb[0] = lds_read16<AL>(arow + 16);
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) {
                if (s + 1 < KSTEPS) {
                    ra[(s + 1) & 1] = lds_read16<AL>(arow + (s + 1) * 128);
                    rb[(s + 1) & 1] = lds_read16<AL>(arow + (s + 1) * 128 + 16);
                }
                const v4i p = ra[s & 1], q = rb[s & 1];
                v4i lo, hi; /* v_perm_b32(hi dword, lo dword, selector): selector bytes 0-3 index the second operand, 4-7 the first */
                lo.x = (int)__builtin_amdgcn_perm((unsigned)p.y, (unsigned)p.x, 0x06040200u); hi.x = (int)__builtin_amdgcn_perm((unsigned)p.y, (unsigned)p.x, 0x07050301u);
                lo.y = (int)__builtin_amdgcn_perm((unsigned)p.w, (unsigned)p.z, 0x06040200u); hi.y = (int)__builtin_amdgcn_perm((unsigned)p.w, (unsigned)p.z, 0x07050301u);
                lo.z = (int)__builtin_amdgcn_perm((unsigned)q.y, (unsigned)q.x, 0x06040200u); hi.z = (int)__builtin_amdgcn_perm((unsigned)q.y, (unsigned)q.x, 0x07050301u);
                lo.w = (int)__builtin_amdgcn_perm((unsigned)q.w, (unsigned)q.z, 0x06040200u); hi.w = (int)__builtin_amdgcn_perm((unsigned)q.w, (unsigned)q.z, 0x07050301u);
                lo.x ^= 0x80808080; lo.y ^= 0x80808080; lo.z ^= 0x80808080; lo.w ^= 0x80808080; /* unsigned low byte -> lo - 128 as int8 */
                A.a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo, b0[s], A.a0, 0, 0, 0);
                A.h0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi, b0[s], A.h0, 0, 0, 0);
                A.a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo, b1[s], A.a1, 0, 0, 0);
                A.h1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi, b1[s], A.h1, 0, 0, 0);
                if (!(EDGE_HI_ZERO && (s < EDGE || s >= KSTEPS - EDGE))) {
                    A.a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo, b2[s], A.a2, 0, 0, 0);
                    A.h2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi, b2[s], A.h2, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mid(); /* (CS16: twice the accumulators -- behind the last MFMA, while the pipe drains) */
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    /* digit sums -> the lane's four values (hops grp * 4 .. + 3 of column col): recombine, restore the -127.5 offset of the reference's
     * LUT (u8) / the + 128 of the low byte (CS16), undo the fixed-point scale */
    auto tile_value = [&](const TileAcc& A, int r) {
        float y = __builtin_fmaf((float)A.a0[r], u0, cu);
        y = __builtin_fmaf((float)A.a1[r], u1, y);
        y = __builtin_fmaf((float)A.a2[r], u2, y);
        if (S16) {
            y = __builtin_fmaf((float)A.h0[r], w0, y);
            y = __builtin_fmaf((float)A.h1[r], w1, y);
            y = __builtin_fmaf((float)A.h2[r], w2, y);
        }
        return y;
    };
    /* the neighbour lane's value: a DPP move inside the quad (quad_perm [1, 0, 3, 2]), no LDS round trip */
    auto pair_swap = [&](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); };
    /* values of tile t -> rings.  Lane pairs (2ch, 2ch+1) hold (re, im) of the same hop; even lanes write 4 consecutive rows of their slot */
    typedef float v4f __attribute__((ext_vector_type(4)));
    /* the lane's own part of a ring offset: tile t puts its rows at physical 16-row tile pt = (ptile0 + t) mod ring_tiles16, i.e. element
     * slot_base + ab_tile_off(16 pt + 4 grp) = lane_off + pt * (16 / AB_TILE_ROWS) * AB_SLOT_BLOCK * AB_TILE_ROWS -- a scalar multiple per tile */
    const bool store_lane = !(col & 1) && ch_valid;
    const long lane_off = slot_base + ab_tile_off(grp * 4);
    float* const mag_lane = a.mag + lane_off;
    float2* const iq_lane = a.iq_bins + lane_off;
    constexpr long TILE16_PITCH = (long)TILE_HOPS * AB_SLOT_BLOCK; /* elements between consecutive 16-row tiles of one slot block */
    auto tile_store = [&](int t, const float* val) {
        float im4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) im4[r] = pair_swap(val[r]);
        const bool whole_tile = t * TILE_HOPS - shift >= 0 && t * TILE_HOPS - shift + TILE_HOPS <= a.n_hops; /* wave-uniform: all 16 hops of the tile are stored */
        int pt = ptile0 + t;
        pt = pt >= ring_tiles16 ? pt - ring_tiles16 : pt;
        if (__builtin_expect(whole_tile, 1)) {
            /* the common case, straight-line: no per-hop tests, one scalar offset per tile, the raw I/Q pairs in two register quads so that
             * they leave as two 16-byte stores (the compiler, left alone, split them into 16 + 8 + 8) */
            stores += k_tile;
            if (store_lane) {
                const long toff = (long)pt * TILE16_PITCH;
                if (want_mag) {
                    v4f m;
#pragma unroll
                    for (int r = 0; r < 4; r++) m[r] = __builtin_amdgcn_sqrtf(val[r] * val[r] + im4[r] * im4[r]); /* v_sqrt_f32, 1 ulp: stage 1 is tolerance-bound anyway */
#if defined(AB_ABL_NO_STORE)
                    asm volatile("" ::"v"(m));
#else
                    *reinterpret_cast<v4f*>(mag_lane + toff) = m;
#endif
                }
                if (want_iq) {
                    v4f qa = {val[0], im4[0], val[1], im4[1]}, qb = {val[2], im4[2], val[3], im4[3]};
                    asm volatile("" : "+v"(qa), "+v"(qb));
#if !defined(AB_ABL_NO_STORE)
                    v4f* q = reinterpret_cast<v4f*>(iq_lane + toff);
                    q[0] = qa;
                    q[1] = qb;
#endif
                }
            }
            return;
        }
        if (store_lane) { /* first / last tile of a batch: hops outside [0, n_hops) are computed and dropped */
            const long off = slot_base + ab_tile_off(pt * TILE_HOPS + grp * 4); /* the lane's 4 hops never straddle a ring tile (4, 8 or 16 rows) */
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
    };

    /* ---- u8 / s8, one wave, three staging buffers, one tile per step (the 320-byte hop of BASELINE configs[2] and every other hop whose
     * three buffers fit): software-pipelined across tiles.  Per tile: MFMAs of tile t (its first A fragments were fetched during tile
     * t - 1) -> the buffer tile t read is free: the transfer of step t + 3 goes out (three steps in flight, was two) -> wait for step
     * t + 1's bytes, issue tile t + 1's first A fragments -> recombine and store tile t while those LDS reads fly.  The wave no longer
     * stalls on LDS latency at the top of every tile, nor on the matrix pipe's drain with nothing else to issue. */
    /* ---- staging ring (round 6; review item 3 ii): the stream of a wave lives ONCE in LDS.  Three slots of one 16-hop step each (5 KiB at hops of 320 bytes) form a
     * ring; a step's transfer brings only its NEW bytes, and a tile reads its window tail in place -- the first 704 bytes of the NEXT slot (the ring's first KiB is kept a
     * second time behind its end, so a row never wraps).  Against the three buffers of round 5: five transfers per step instead of six (the sixth re-read the previous
     * step's last KiB out of L2: a sixth of this kernel's requests), none of them read twice, so EVERY piece may go out non-temporal (section A of
     * profiles/r06_experiments.md: nt on every piece cost 13 % while two pieces in six were re-read), 16 KiB of LDS per wave instead of 18.  What it costs: tile st needs
     * the head (first KiB) of step st + 1, so that transfer is waited for one tile earlier than its step used to be.  (Result: see AB_RING above -- no gain.)
     * Order inside a step's transfer: piece 0, its replica (slot 0 only), pieces 1 ...: "the head of step v has landed" = at most tail(v) + everything younger outstanding. */
    if constexpr (RING) {
        constexpr int RS = TILE_HOPS * (HOPB ? HOPB : 64); /* bytes per step = per slot */
        constexpr int RNP = RS / 1024;                     /* transfers per step */
        constexpr int RING_BYTES = 3 * RS;                 /* + 1 KiB replica of the ring's first KiB behind it */
        static_assert(RNP >= 2 && RNP <= 12, "pieces at immediate offsets below 12 KiB");
#ifndef AB_RING_AUX
#define AB_RING_AUX 2 /* nt: every piece of the ring is read once */
#endif
#ifndef AB_RING_HEAD_AUX
#define AB_RING_HEAD_AUX 0 /* piece 0 of a slot-0 step is requested twice in a row (the ring and its replica) */
#endif
        const uint8_t* const rp_lane = src - (long)shift * HOPB + lane * 16; /* (16-byte aligned hops: mis = delta = 0) */
        const int r_in_lo = shift > 0 ? 1 : 0;
        const long room_n = span_end - (long)RNP * 1024, room_1 = span_end - 1024;
        const int r_in_hi_n = __builtin_amdgcn_readfirstlane(room_n < 0 ? -1 : (int)((room_n / HOPB + shift) / TILE_HOPS));
        const int r_in_hi_1 = __builtin_amdgcn_readfirstlane(room_1 < 0 ? -1 : (int)((room_1 / HOPB + shift) / TILE_HOPS));
        /* step v (st_begin ... st_end: the one past the last tile's brings its head only) into ring slot `slot` */
        auto ring_stage = [&](int v, int slot) {
            const bool whole = v < st_end;
            uint8_t* dst = lds + slot * RS;
            if (v >= r_in_lo && v <= (whole ? r_in_hi_n : r_in_hi_1)) { /* wave-uniform: no lane's address needs clamping */
                const uint8_t* p = rp_lane + (unsigned long long)(unsigned)v * (unsigned)RS;
                if (slot == 0) {
                    AB_DMA((gptr_t)p, (lptr_t)(uintptr_t)dst, 0, AB_RING_HEAD_AUX);
                    AB_DMA((gptr_t)p, (lptr_t)(uintptr_t)(lds + RING_BYTES), 0, AB_RING_HEAD_AUX);
                } else {
                    AB_DMA((gptr_t)p, (lptr_t)(uintptr_t)dst, 0, AB_RING_AUX);
                }
                if (whole) {
#define AB_RPIECE(K, BASE, OFF) \
    if (RNP > (K)) AB_DMA((gptr_t)(p + (BASE)), (lptr_t)(uintptr_t)(dst + (BASE)), (OFF), AB_RING_AUX)
                    AB_RPIECE(1, 0, 1024); AB_RPIECE(2, 0, 2048); AB_RPIECE(3, 0, 3072);
                    AB_RPIECE(4, 4096, 0); AB_RPIECE(5, 4096, 1024); AB_RPIECE(6, 4096, 2048); AB_RPIECE(7, 4096, 3072);
                    AB_RPIECE(8, 8192, 0); AB_RPIECE(9, 8192, 1024); AB_RPIECE(10, 8192, 2048); AB_RPIECE(11, 8192, 3072);
#undef AB_RPIECE
                }
                return;
            }
            const long base = ((long)v * TILE_HOPS - shift) * HOPB;
            const int np = whole ? RNP : 1;
            for (int i = 0; i < np; i++) {
                long so = base + i * 1024 + lane * 16;
                if (so + 16 > span_end) so = span_end - 16;
                if (so < 0) so = 0;
                AB_DMA((gptr_t)(src + so), (lptr_t)(uintptr_t)(dst + i * 1024), 0, 0);
                if (i == 0 && slot == 0) AB_DMA((gptr_t)(src + so), (lptr_t)(uintptr_t)(lds + RING_BYTES), 0, 0);
            }
        };
        const int nst = st_end - st_begin;
        ring_stage(st_begin, 0);
        ring_stage(st_begin + 1, 1);
        if (nst >= 2) ring_stage(st_begin + 2, 2);
        /* step st_begin whole + the head of the next: younger are that step's tail and the third transfer */
        wait_vmcnt((nst >= 2 ? RNP - 1 : 0) + (nst >= 3 ? RNP : nst == 2 ? 1 : 0));
        v4i pre[4];
        a_head(a_row(lds, 0), pre);
        int mark2 = 0; /* `stores` when the transfer of step st + 2 was issued */
        int slot = 0;
        for (int st = st_begin; st < st_end; st++) {
            TileAcc now;
            tile_body(a_row(lds + slot * RS, 0), pre, now, no_mid);
            const bool has3 = st + 3 <= st_end;
            if (has3) ring_stage(st + 3, slot); /* every LDS read of this slot has returned (the MFMAs consumed them; the tile before read its first 704 bytes): it takes the step three ahead */
            const int mark3 = stores;
            const int slot1 = slot == 2 ? 0 : slot + 1;
            if (st + 1 < st_end) {
                /* tile st + 1 needs step st + 1 whole and the head of step st + 2 (transfers land in issue order): younger than that head are its step's tail, the
                 * transfer just issued and the stores since step st + 2 went out */
                const int tail2 = st + 2 < st_end ? RNP - 1 : 0;
                const int cnt3 = has3 ? (st + 3 < st_end ? RNP : 1) + (slot == 0 ? 1 : 0) : 0;
                wait_vmcnt(tail2 + cnt3 + (stores - mark2));
                a_head(a_row(lds + slot1 * RS, 0), pre);
            }
            mark2 = mark3;
            float val[4];
#pragma unroll
            for (int r = 0; r < 4; r++) val[r] = tile_value(now, r);
            tile_store(st, val);
            slot = slot1;
        }
        return;
    }

    if (NP == 1 && !S16 && nbuf == 3 && sub == 1) {
        /* steps whose transfer lies wholly inside the batch span (no lane's address needs clamping), as a range worked out once: the per-step
         * test is two scalar compares and the source address one multiply-add */
        const int st_in_lo = shift > 0 ? 1 : 0;
        const long in_room = span_end - (long)n_dma * 1024 - (AL >= 16 ? 0 : 16); /* (hops that are not multiples of 16 bytes: an image starts up to 15 bytes off the step's first hop) */
        const int st_in_hi = __builtin_amdgcn_readfirstlane(in_room < 0 ? -1 : (int)((in_room / hop_bytes + shift) / TILE_HOPS)); /* (the same number on every lane: keep it scalar) */
        /* step st's image starts at p_lane + st * 16 hops; hops that are not multiples of 16 bytes: `delta` bytes in front of the step's first hop, the same
         * for every step (a step is 16 hops), so that every transfer is 16-byte aligned -- the interior test stays valid, it only gets more cautious */
        const uint8_t* const p_lane = src + mis - (long)shift * hop_bytes - delta + lane * 16;
        const int step_bytes = TILE_HOPS * hop_bytes;
        auto stage_fast = [&](int step, uint8_t* buf) {
            if (step >= st_in_lo && step <= st_in_hi) {
                const uint8_t* p = p_lane + (unsigned long long)(unsigned)step * (unsigned)step_bytes;
#define AB_PIECE(K, BASE, OFF) \
    if (n_dma > (K)) AB_DMA((gptr_t)(p + (BASE)), (lptr_t)(uintptr_t)(buf + (BASE)), (OFF), AB_PIECE_AUX(K))
                AB_PIECE(0, 0, 0); AB_PIECE(1, 0, 1024); AB_PIECE(2, 0, 2048); AB_PIECE(3, 0, 3072);
                AB_PIECE(4, 4096, 0); AB_PIECE(5, 4096, 1024); AB_PIECE(6, 4096, 2048); AB_PIECE(7, 4096, 3072);
                AB_PIECE(8, 8192, 0); AB_PIECE(9, 8192, 1024); AB_PIECE(10, 8192, 2048); AB_PIECE(11, 8192, 3072);
#undef AB_PIECE
                return;
            }
            stage(step, buf);
        };
        const int nst = st_end - st_begin;
        if (nst > 2) stage(st_begin + 2, lds + 2 * lds_per_buf);
        wait_vmcnt((nst > 2 ? 2 : nst - 1) * n_dma);
        v4i pre[4];
        a_head(a_row(lds, 0), pre);
        /* `stores` at the time the transfers of steps st + 1 and st + 2 were issued (the three transfers of the prologue: 0) */
        int mark_a = 0, mark_b = 0;
        int buf_off = 0; /* LDS offset of step st's buffer */
        for (int st = st_begin; st < st_end; st++) {
            uint8_t* buf = lds + buf_off;
            TileAcc now;
            tile_body(a_row(buf, 0), pre, now, no_mid);
            const bool more3 = st + 3 < st_end;
            if (more3) stage_fast(st + 3, buf); /* every LDS read of this buffer has returned (the MFMAs consumed them): it takes the step three ahead */
            const int mark_c = stores;
            buf_off = buf_off + lds_per_buf == 3 * lds_per_buf ? 0 : buf_off + lds_per_buf;
            if (st + 1 < st_end) {
                /* younger than step st + 1's transfer: the pieces of steps st + 2 and st + 3, the stores since it was issued */
                wait_vmcnt((st + 2 < st_end ? n_dma : 0) + (more3 ? n_dma : 0) + (stores - mark_a));
                a_head(a_row(lds + buf_off, 0), pre);
            }
            mark_a = mark_b;
            mark_b = mark_c;
            float val[4];
#pragma unroll
            for (int r = 0; r < 4; r++) val[r] = tile_value(now, r);
            tile_store(st, val);
        }
        return;
    }

    constexpr bool PIPE_PIECES = NP > 1 && !S16 && AL >= 4;
    /* window pieces: wave 0's own sums of the tile whose other pieces it has yet to add (finished under the next tile's MFMAs) */
    float pend[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int pend_t = 0;
    bool have_pend = false;
    auto finish_tile = [&](int t, float* val) {
        const float4* ex = exch + (t & 1) * (NP - 1) * 64;
#pragma unroll
        for (int q = 0; q < NP - 1; q++) {
            const float4 o = ex[q * 64 + lane];
            val[0] += o.x; val[1] += o.y; val[2] += o.z; val[3] += o.w;
        }
        if (a.partial) { /* fft_size 8192: the first pass parks its sums (whole-wave 1 KiB rows), the second adds them to its own */
            /* (row base through scalar registers: left to itself the compiler keeps a per-lane 64-bit base alive across the whole loop) */
            const unsigned long long rb = (unsigned long long)(a.partial + ((long)item * tiles_total + t) * 64);
            const unsigned rb_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rb), rb_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rb >> 32));
            float4* row = reinterpret_cast<float4*>(((unsigned long long)rb_hi << 32) | rb_lo) + lane;
            if (a.piece0 == 0) {
                *row = make_float4(val[0], val[1], val[2], val[3]);
                return;
            }
            const float4 o = *row;
            val[0] += o.x; val[1] += o.y; val[2] += o.z; val[3] += o.w;
        }
        tile_store(t, val);
    };
    for (int st = st_begin; st < st_end; st++) {
        uint8_t* buf = lds + cur * lds_per_buf;
        /* NP > 1: wave 0 runs the transfers and the waits; the barrier hands step st to the other waves and tells wave 0 that they
         * are done with the buffer the next transfer overwrites (they read it in step st - 1) */
        if (nbuf == 3) {
            if (piece == 0) wait_vmcnt((st + 1 < st_end ? n_dma : 0) + (stores - mark_get(cur))); /* younger than step st's transfer: step st + 1's pieces, the stores since */
            if (NP > 1) __syncthreads();
            int nb = cur + 2;
            nb = nb >= 3 ? nb - 3 : nb;
            if (piece == 0 && st + 2 < st_end) { /* two steps ahead: the buffer step st - 1 just left */
                stage(st + 2, lds + nb * lds_per_buf);
                mark_set(nb, stores);
            }
        } else {
            if (piece == 0) wait_vmcnt(stores - mark_get(cur)); /* this step's bytes have landed in LDS */
            if (NP > 1) __syncthreads();
            if (piece == 0 && st + 1 < st_end) { /* next step streams in under this step's MFMAs */
                stage(st + 1, lds + (cur ^ 1) * lds_per_buf);
                mark_set(cur ^ 1, stores);
            }
        }
        for (int sb = 0; sb < sub; sb++) {
            const int t = st * sub + sb;
            if (t >= tiles_total) break;
            float val[4];
            TileAcc now;
            if (NP == 1) {
                tile_mfma(buf, sb, now, no_mid);
#pragma unroll
                for (int r = 0; r < 4; r++) val[r] = tile_value(now, r);
                tile_store(t, val);
                continue;
            }
            /* Window pieces (round 6: software-pipelined across the workgroup barrier).  Every wave runs its piece's MFMAs for tile t; the other pieces' partial sums
             * reach wave 0 through LDS (two areas alternate).  Wave 0 used to add them up and store right behind a second barrier per tile, with the other waves
             * already waiting at the next one and the matrix pipe idle (0.40 busy at every window length, profiles/r04_experiments.md G: the review's reading was that this
             * second barrier is what idles it -- it is 1.5 % of fft 1024's launch and 6 % of fft 4096's; the matrix time and the stream time of these sizes ADD UP instead of
             * overlapping, 6.9 + 8.4 ms at fft 1024, and what binds them is a workgroup's own tile latency at the residency the B fragments allow -- half the bytes in flight cost 1.5 - 6 %, three workgroups per CU instead of four 11 %, two 50 %: profiles/r06_experiments.md F).  Now tile t - 1 is finished
             * UNDER tile t's MFMAs -- `mid`, called with half of them issued -- and ONE barrier per tile does both jobs: it hands the staged step over and orders the
             * waves' writes of tile t - 1's sums before wave 0's reads (the area tile t's sums go to was last read under tile t - 1, before this barrier). */
            if (!PIPE_PIECES) { /* CS16 (twice the accumulators) and hops of an odd number of samples (five-dword fragment reads): no registers to spare for a tile in waiting */
                tile_mfma(buf, sb, now, no_mid);
#pragma unroll
                for (int r = 0; r < 4; r++) val[r] = tile_value(now, r);
                if (piece > 0) exch[((t & 1) * (NP - 1) + (piece - 1)) * 64 + lane] = make_float4(val[0], val[1], val[2], val[3]);
                __syncthreads();
                if (piece == 0) finish_tile(t, val);
                continue;
            }
            if (sb > 0) __syncthreads(); /* (sb == 0: the step's barrier above) */
            auto finish_pending = [&] {
                if (piece == 0 && have_pend) finish_tile(pend_t, pend);
            };
            tile_mfma(buf, sb, now, finish_pending);
#pragma unroll
            for (int r = 0; r < 4; r++) val[r] = tile_value(now, r);
            if (piece > 0) {
                exch[((t & 1) * (NP - 1) + (piece - 1)) * 64 + lane] = make_float4(val[0], val[1], val[2], val[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) pend[r] = val[r];
                pend_t = t;
                have_pend = true;
            }
        }
        cur = cur + 1 == nbuf ? 0 : cur + 1;
    }
    if (PIPE_PIECES) { /* the last tile's sums */
        __syncthreads();
        if (piece == 0 && have_pend) finish_tile(pend_t, pend);
    }
}

}  // namespace

bool dft_supported(int fft_size, int hop_bytes, int sfmt, int max_ch) {
    (void)max_ch; /* any channel count: dongles with more than 8 channels are split into groups of 8 */
    if (fft_size < 256 || fft_size > 8192 || (fft_size & (fft_size - 1))) return false; /* >= 1024: window pieces of 512 samples, one wave each (8192: two passes of eight) */
    /* u8 / s8: any hop (2.4 MS/s -> 300 / 600 bytes at 4- / 8-byte alignment, 2.0 MS/s at WAVE_RATE 16000 -> 250 bytes at 2-byte alignment: the AL
     * variants of the A-fragment reads); two staging buffers of 16 hops + one window must leave room for 3+ waves per CU */
    if (sfmt == AIRBAND_SFMT_U8 || sfmt == AIRBAND_SFMT_S8) return (hop_bytes % 2) == 0 && hop_bytes <= 1024 && hop_bytes >= 64; /* (a hop is whole I/Q pairs: always even) */
    if (sfmt == AIRBAND_SFMT_S16) return (hop_bytes % 4) == 0 && hop_bytes <= 1280 && hop_bytes >= 128;
    return false;
}

/* win_bytes = bytes of a whole window, np = its pieces of 512 samples (1 up to fft_size 512) */
int dft_sub(int hop_bytes, int win_bytes, int np) { return c_sub(hop_bytes, win_bytes, np); }
int dft_nbuf(int hop_bytes, int win_bytes, int np) { return c_nbuf(hop_bytes, win_bytes, np); }
int dft_lds_per_buf(int hop_bytes, int win_bytes, int np) { return c_lds_per_buf(hop_bytes, win_bytes, np); }
int dft_partial_tiles(int n_hops_max) { return (15 + n_hops_max + TILE_HOPS - 1) / TILE_HOPS + 1; }

template <int FFT_N, int HOPB, bool S16, int AL, int NP = 1>
static void launch_al(const DftArgs& a, hipStream_t stream) {
    const long groups = (long)a.n_items * a.splits;
    /* AIRBAND_HIP_DFT_EXTRA_LDS=<bytes> (measurements only, profiles/r06_experiments.md G): LDS the launch asks for and never touches, so that fewer wavefronts fit a CU --
     * what the channelizer loses when something else (a fused consumer, a resident stage-2 wavefront) takes a share of the CU */
    static const size_t extra_lds = [] {
        const char* e = getenv("AIRBAND_HIP_DFT_EXTRA_LDS");
        return e ? (size_t)atol(e) : (size_t)0;
    }();
    constexpr bool ring = AB_RING != 0 && HOPB != 0 && (HOPB != 640 || AB_RING_640 != 0) && NP == 1 && !S16 && AL >= 16 && (TILE_HOPS * (HOPB ? HOPB : 64)) % 1024 == 0; /* (the kernel's RING) */
    const size_t lds = (ring ? (size_t)3 * TILE_HOPS * HOPB + 1024 : (size_t)a.nbuf * a.lds_per_buf + (NP > 1 ? 2 * (NP - 1) * 64 * sizeof(float4) : 0)) + extra_lds + (size_t)(a.extra_lds > 0 ? a.extra_lds : 0);
    /* more than the default 64 KiB of dynamic LDS (eight-piece windows): opt in to the CU's 160 KiB, once per kernel variant */
    /* (once per kernel variant AND device: the attribute belongs to the function as loaded on the current device, and a process may drive several GPUs) */
    static std::atomic<bool> big_lds_dev[64][2]; /* (zero-initialised; one launching thread per GPU in the shim: setting it twice is harmless) */
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    const bool tracked = cur_dev >= 0 && cur_dev < 64; /* devices beyond the table opt in on every launch */
    std::atomic<bool>* big_lds = big_lds_dev[tracked ? cur_dev : 0];
    if (lds > 64 * 1024 && (!tracked || !big_lds[a.edge_hi_zero ? 1 : 0].load(std::memory_order_acquire))) {
        const void* fn = a.edge_hi_zero ? reinterpret_cast<const void*>(&channelizer_dft_kernel<FFT_N, true, HOPB, S16, AL, NP>)
                                        : reinterpret_cast<const void*>(&channelizer_dft_kernel<FFT_N, false, HOPB, S16, AL, NP>);
        /* the CU's whole 160 KiB, not this launch's size: the flag is per variant, and a later handle of the same process may need more (runtime hop lengths) */
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess && tracked) big_lds[a.edge_hi_zero ? 1 : 0].store(true, std::memory_order_release);
    }
    if (a.edge_hi_zero)
        hipLaunchKernelGGL((channelizer_dft_kernel<FFT_N, true, HOPB, S16, AL, NP>), dim3((unsigned)groups), dim3(64 * NP), lds, stream, a);
    else
        hipLaunchKernelGGL((channelizer_dft_kernel<FFT_N, false, HOPB, S16, AL, NP>), dim3((unsigned)groups), dim3(64 * NP), lds, stream, a);
}

template <int FFT_N, bool S16, int NP = 1>
static void launch_generic(const DftArgs& a, hipStream_t stream) {
    if ((a.hop_bytes & 15) == 0) return launch_al<FFT_N, 0, S16, 16, NP>(a, stream);
    if ((a.hop_bytes & 7) == 0) return launch_al<FFT_N, 0, S16, 8, NP>(a, stream);
    if ((a.hop_bytes & 3) == 0) return launch_al<FFT_N, 0, S16, 4, NP>(a, stream);
    if constexpr (!S16) launch_al<FFT_N, 0, false, 2, NP>(a, stream); /* u8 / s8 hops of an odd number of samples (CS16 hops are multiples of 4 bytes) */
}

static void launch_one_piece(const DftArgs& a, hipStream_t stream);

void launch_channelizer_dft(const DftArgs& a0, hipStream_t stream) {
    DftArgs a = a0;
    a.np_total = a0.fft_size > 512 ? a0.fft_size / 512 : 1;
    a.piece0 = 0;
    if (a0.fft_size > 512) { /* one wavefront per window piece of 512 samples */
        a.fft_size = 512;
        const bool s16 = a.sfmt == AIRBAND_SFMT_S16;
        if (a0.fft_size == 1024) return s16 ? launch_generic<512, true, 2>(a, stream) : launch_generic<512, false, 2>(a, stream);
        if (a0.fft_size == 2048) return s16 ? launch_generic<512, true, 4>(a, stream) : launch_generic<512, false, 4>(a, stream);
        /* 4096: eight pieces, eight waves -- a whole CU's register file (two waves of B fragments per SIMD).  8192: sixteen pieces do not
         * fit one CU, so two passes of eight; the first parks its partial sums in a.partial, the second adds them and writes the rings.
         * The stream is read once per pass: 2x the bytes of the smaller sizes, and 8x / 16x their matrix work -- these sizes are MFMA-bound. */
        if (a0.fft_size == 4096) a.partial = nullptr;
        for (int pass = 0; pass * 8 < a.np_total; pass++) {
            a.piece0 = pass * 8;
            if (s16) launch_generic<512, true, 8>(a, stream);
            else launch_generic<512, false, 8>(a, stream);
        }
        return;
    }
    a.partial = nullptr;
    launch_one_piece(a, stream);
}

static void launch_one_piece(const DftArgs& a, hipStream_t stream) {
    if (a.sfmt == AIRBAND_SFMT_S16) {
        if (a.fft_size == 256) return launch_generic<256, true>(a, stream);
        return launch_generic<512, true>(a, stream);
    }
    if (a.fft_size == 256) return launch_generic<256, false>(a, stream);
    /* the host derives nbuf / sub / lds_per_buf with the same functions the specialised kernels fold in at compile time */
    if (a.hop_bytes == 320 && a.nbuf == c_nbuf(320) && a.sub == c_sub(320) && a.lds_per_buf == c_lds_per_buf(320)) return launch_al<512, 320, false, 16>(a, stream);
    if (a.hop_bytes == 640 && a.nbuf == c_nbuf(640) && a.sub == c_sub(640) && a.lds_per_buf == c_lds_per_buf(640)) return launch_al<512, 640, false, 16>(a, stream);
    launch_generic<512, false>(a, stream);
}

}  // namespace airband
