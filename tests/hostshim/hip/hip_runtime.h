// tests/hostshim/hip/hip_runtime.h -- TEST HARNESS ONLY.  What csrc/demod.hip needs of <hip/hip_runtime.h> to compile as plain C++ for the host
// (tests/host_demod_harness.cpp, built by tests/test_host_demod.py into a temporary directory): the kernel-language keywords as nothing, the vector
// types, and the few wavefront intrinsics of the LANE-PER-CHANNEL code with a "wavefront" of ONE lane (as csrc/squelch_fsm.h's own host mode has it).
// Launches are discarded: the harness calls the device functions itself, one lane at a time.  Nothing in the library includes or links this.
#ifndef AIRBAND_TESTS_HOSTSHIM_HIP_RUNTIME_H
#define AIRBAND_TESTS_HOSTSHIM_HIP_RUNTIME_H

#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

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
// the lane the harness is running right now
static thread_local dim3 threadIdx(0, 0, 0), blockIdx(0, 0, 0), blockDim(64, 1, 1), gridDim(1, 1, 1);

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
#define hipLaunchKernelGGL(...) ((void)0)

static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned long long __ballot(bool b) { return b ? 1ull : 0ull; } // one lane: never "all 64", so the cooperative stores stay off
static inline void __syncthreads() {}
static inline int min(int a, int b) { return a < b ? a : b; }
static inline float ab_host_med3(float a, float b, float c) { // v_med3_f32 for non-NaN operands (the only use clamps a value tested for NaN first)
    const float lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
#define __builtin_amdgcn_fmed3f(a, b, c) ab_host_med3(a, b, c)
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_readlane(v, l) (v)

#endif
