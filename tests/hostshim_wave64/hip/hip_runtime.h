// tests/hostshim_wave64/hip/hip_runtime.h -- TEST HARNESS ONLY.  A host stand-in for <hip/hip_runtime.h> that RUNS the stage-2 kernels of csrc/demod.hip
// with their wavefront semantics: every work-item of a block is a fiber (ucontext) of one host thread, a "launch" runs the grid block after block, and the
// cross-lane operations the kernels use -- __ballot, v_readlane / v_readfirstlane, __syncthreads, and the places where the code relies on the 64 lanes
// of a wavefront executing in lockstep (AB_LOCKSTEP() in the source: LDS exchanges without a barrier) -- are rendezvous points of the fibers of a wavefront
// (or block).  A lane that returns from the kernel leaves the wavefront, as on the GPU.  A collective reached by only some of the live lanes of a
// wavefront is a deadlock here and is reported as one: it checks the rule the kernels are written to ("lane masks are only ever assigned in wave-uniform
// control flow").  __HIPCC__ is defined, so csrc/squelch_fsm.h, csrc/exact_math.h and csrc/demod.hip take their DEVICE branches -- the code under test is
// the code the GPU runs, lane masks and all.  Used by tests/host_wave64_harness.cpp (built by tests/test_host_wave64.py into a temporary directory);
// nothing in the library includes or links this.
#ifndef AIRBAND_TESTS_HOSTSHIM_WAVE64_HIP_RUNTIME_H
#define AIRBAND_TESTS_HOSTSHIM_WAVE64_HIP_RUNTIME_H

#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __HIPCC__ 1
#define __device__
#define __host__
#define __global__
#define __shared__ static /* one block runs at a time: a function-local static IS the block's LDS */
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define AB_NEEDED_NOW(...) ((void)0) /* csrc/demod.hip: "these registers are needed now" -- nothing to wait for here */
#define AB_V(x) 0
#define AB_DYNAMIC_LDS(type, name) extern type name[] /* defined by the harness */

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static dim3 threadIdx(0, 0, 0), blockIdx(0, 0, 0), blockDim(64, 1, 1), gridDim(1, 1, 1); /* of the fiber that is running */

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }

namespace ab_emu {

constexpr int MAX_THREADS = 1024, WAVE = 64; /* (1 024: regroup_perm_kernel, sixteen wavefronts per workgroup) */
constexpr size_t STACK_BYTES = 512 * 1024;

struct Group { /* a wavefront or a block: the lanes that rendezvous */
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    unsigned slot[2][MAX_THREADS];
    bool put[2][MAX_THREADS];
};

struct Machine {
    ucontext_t main_ctx, ctx[MAX_THREADS];
    char* stack[MAX_THREADS] = {nullptr};
    bool live[MAX_THREADS];
    int n_threads = 0, cur = -1;
    Group wave[MAX_THREADS / WAVE], block;
    long progress = 0; /* collectives completed + lanes retired: the scheduler's deadlock test */
    std::function<void()> body;
};
static Machine M;

static inline void yield() { swapcontext(&M.ctx[M.cur], &M.main_ctx); }

static inline void complete(Group& g) {
    g.arrived = 0;
    g.gen++;
    M.progress++;
}

/* deposit v, wait for every live lane of the group, return the generation the deposits of this rendezvous live in */
static inline unsigned rendezvous(Group& g, int idx, unsigned v) {
    const unsigned my = g.gen;
    if (g.arrived == 0)
        for (int i = 0; i < MAX_THREADS; i++) g.put[my & 1][i] = false;
    g.slot[my & 1][idx] = v;
    g.put[my & 1][idx] = true;
    if (++g.arrived == g.alive) complete(g);
    else
        while (g.gen == my) yield();
    return my;
}

static inline Group& my_wave() { return M.wave[M.cur / WAVE]; }
static inline int my_lane() { return M.cur % WAVE; }

static inline unsigned long long ballot(bool b) {
    Group& g = my_wave();
    const unsigned my = rendezvous(g, my_lane(), b ? 1u : 0u);
    unsigned long long m = 0;
    for (int i = 0; i < WAVE; i++)
        if (g.put[my & 1][i] && g.slot[my & 1][i]) m |= 1ull << i;
    return m;
}
static inline unsigned readlane(unsigned v, int lane) {
    Group& g = my_wave();
    const unsigned my = rendezvous(g, my_lane(), v);
    return g.put[my & 1][lane] ? g.slot[my & 1][lane] : 0u;
}
static inline unsigned readfirstlane(unsigned v) {
    Group& g = my_wave();
    const unsigned my = rendezvous(g, my_lane(), v);
    for (int i = 0; i < WAVE; i++)
        if (g.put[my & 1][i]) return g.slot[my & 1][i];
    return v;
}
static inline void lockstep() { (void)rendezvous(my_wave(), my_lane(), 0u); }
static inline void syncthreads() { (void)rendezvous(M.block, M.cur, 0u); }

static void fiber_entry() {
    M.body();
    /* the lane retires: whoever waits for it need not any more */
    const int i = M.cur;
    M.live[i] = false;
    M.progress++;
    Group& w = M.wave[i / WAVE];
    if (--w.alive > 0 && w.arrived == w.alive) complete(w);
    if (--M.block.alive > 0 && M.block.arrived == M.block.alive) complete(M.block);
    swapcontext(&M.ctx[i], &M.main_ctx);
}

/* one block: n work-items as fibers, round robin until all have returned */
static inline void run_block(int n, const std::function<void()>& body) {
    if (n > MAX_THREADS || n % WAVE) {
        std::fprintf(stderr, "ab_emu: block of %d work-items\n", n);
        std::abort();
    }
    M.n_threads = n;
    M.body = body;
    M.block = Group();
    M.block.alive = n;
    for (int w = 0; w < n / WAVE; w++) {
        M.wave[w] = Group();
        M.wave[w].alive = WAVE;
    }
    for (int i = 0; i < n; i++) {
        if (!M.stack[i]) M.stack[i] = static_cast<char*>(std::malloc(STACK_BYTES));
        getcontext(&M.ctx[i]);
        M.ctx[i].uc_stack.ss_sp = M.stack[i];
        M.ctx[i].uc_stack.ss_size = STACK_BYTES;
        M.ctx[i].uc_link = &M.main_ctx;
        makecontext(&M.ctx[i], fiber_entry, 0);
        M.live[i] = true;
    }
    int left = n, idle_rounds = 0;
    while (left > 0) {
        const long before = M.progress;
        left = 0;
        for (int i = 0; i < n; i++) {
            if (!M.live[i]) continue;
            M.cur = i;
            threadIdx.x = (unsigned)i;
            swapcontext(&M.main_ctx, &M.ctx[i]);
            if (M.live[i]) left++;
        }
        if (left > 0 && M.progress == before) {
            if (++idle_rounds > 2) {
                std::fprintf(stderr, "ab_emu: deadlock -- a cross-lane operation was reached by only some of the live lanes of a wavefront (block %u)\n", blockIdx.x);
                std::abort();
            }
        } else {
            idle_rounds = 0;
        }
    }
    M.cur = -1;
}

/* a block's dynamic LDS (kernels that declare it with AB_DYNAMIC_LDS_BYTES): the bytes the launch asked for, then a guard zone that must come back untouched */
constexpr size_t LDS_MAX = 160 * 1024, LDS_GUARD = 4096;
alignas(16) static uint8_t dynamic_lds[LDS_MAX + LDS_GUARD];

template <class K, class... A>
static inline void launch(K kernel, dim3 grid, dim3 block, size_t lds_bytes, A... args) {
    gridDim = grid;
    blockDim = block;
    if (lds_bytes > LDS_MAX) {
        std::fprintf(stderr, "ab_emu: launch asks for %zu bytes of LDS\n", lds_bytes);
        std::abort();
    }
    for (unsigned by = 0; by < grid.y; by++)
        for (unsigned bx = 0; bx < grid.x; bx++) {
            blockIdx = dim3(bx, by, 0);
            std::memset(dynamic_lds + lds_bytes, 0xA5, LDS_GUARD);
            run_block((int)block.x, [&]() { kernel(args...); });
            for (size_t i = 0; i < LDS_GUARD; i++)
                if (dynamic_lds[lds_bytes + i] != 0xA5) {
                    std::fprintf(stderr, "ab_emu: block %u wrote LDS byte %zu of a launch with %zu bytes of dynamic LDS\n", bx, lds_bytes + i, lds_bytes);
                    std::abort();
                }
        }
}

}  // namespace ab_emu

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ab_emu::launch(kernel, dim3(grid), dim3(block), (size_t)(lds), __VA_ARGS__)
#define AB_LOCKSTEP() ab_emu::lockstep()

static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned long long __ballot(bool b) { return ab_emu::ballot(b); }
static inline void __syncthreads() { ab_emu::syncthreads(); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline float ab_host_med3(float a, float b, float c) { /* v_med3_f32 for non-NaN operands (the only use clamps a value tested for NaN first) */
    const float lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
static inline bool ab_emu_inverse_ballot(unsigned long long m) { return (m >> ab_emu::my_lane()) & 1ull; }
#define __builtin_amdgcn_fmed3f(a, b, c) ab_host_med3(a, b, c)
#define __builtin_amdgcn_inverse_ballot_w64(m) ab_emu_inverse_ballot(m)
#define __builtin_amdgcn_readfirstlane(v) ((int)ab_emu::readfirstlane((unsigned)(v)))
#define __builtin_amdgcn_readlane(v, l) ab_emu::readlane((unsigned)(v), (l))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x) /* v_sqrt_f32 is within 1 ulp; tests/test_exact_math.py covers what exact_math.h makes of either neighbour */

/* ---- what csrc/channelizer_fft.hip needs on top (tests/host_fft_harness.cpp): shuffles as rendezvous points, a block's dynamic LDS, two vector types */
struct char2 { signed char x, y; };
struct short2 { short x, y; };
static inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
static inline float ab_emu_shfl(float v, int src) { /* ds_bpermute: every live lane deposits, then reads its source lane's deposit */
    ab_emu::Group& g = ab_emu::my_wave();
    const unsigned my = ab_emu::rendezvous(g, ab_emu::my_lane(), __float_as_uint(v));
    src &= 63;
    return g.put[my & 1][src] ? __uint_as_float(g.slot[my & 1][src]) : v;
}
static inline float __shfl(float v, int src) { return ab_emu_shfl(v, src); }
static inline float __shfl_xor(float v, int mask) { return ab_emu_shfl(v, ab_emu::my_lane() ^ mask); }
#define AB_WAVE_SYNC() ab_emu::lockstep() /* csrc/channelizer_fft.hip: lanes of one wavefront exchange data through LDS */
#define AB_DYNAMIC_LDS_BYTES(name) uint8_t* const name = ab_emu::dynamic_lds /* one block runs at a time; writes past the launch's size are caught (ab_emu::launch) */
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }

#endif
