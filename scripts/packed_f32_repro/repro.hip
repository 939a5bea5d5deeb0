// Stand-alone reproducer for the platform finding of profiles/r05_event_hunt.md section 4 (MI355X / gfx950, ROCm 7.2): while ANOTHER PROCESS runs long launches on the
// same GPU, a packed-f32 vector instruction (v_pk_mul_f32) executed under a partial EXEC mask now and then returns a wrong value in lanes 48 - 63 of the wavefront.
// No part of libairband_hip.so: the library is built WITHOUT such instructions (rtlsdr-airband_amd/_build.py checks the linked code object).
//   hipcc --offload-arch=gfx950 -O2 repro.hip -o repro        ./repro victim <seconds>   |   ./repro aggressor <seconds>        (scripts/packed_f32_repro/run.sh)
// victim: every wavefront runs the recursive half of a notch filter (the back kernel's, 100 Hz at 16 kHz: a double pole next to z = 1, so ONE wrong value grows
//   1 : 2 : 3 : 4) on the same input in every lane, its multiplies as v_pk_mul_f32 under s_and_saveexec with every other lane switched off (a closed channel);
//   all active lanes must end with bit-identical state -- the host counts wavefronts in which they do not, and which lanes differ.
// aggressor: launches of several milliseconds that fill every CU (144 KiB of LDS, 2 x 250 VGPRs per SIMD: the int8 channelizer's footprint), in a loop.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(64) void victim(const float* in, float* out, int n, int packed) {
    const int lane = threadIdx.x;
    const float d0 = 0.9980404f, d1 = 1.9945439f, d2 = 0.9960809f; // NotchFilter(100 Hz, 16 kHz, q = 10): e, 2 e cos(wo), 2 e - 1
    float x0 = 0, x1 = 0, x2 = 0, y0 = 0, y1 = 0, y2 = 0;
    if (lane & 1) { // every other lane's "channel" is closed: the open lanes run under a half-set EXEC mask
        for (int i = 0; i < n; i++) {
            x0 = x1; x1 = x2; x2 = in[i];
            y0 = y1; y1 = y2;
            float a, b, c, d;
            if (packed) { // the compiler's pairing, spelled out: (d0 x2, d1 x1) and (d1 y1, d2 y0) as two v_pk_mul_f32
                v2f p = {x2, x1}, q = {d0, d1}, r, s = {y1, y0}, t = {d1, d2}, u;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(p), "v"(q));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(s), "v"(t));
                a = r.x; b = r.y; c = u.x; d = u.y;
            } else {
                a = __fmul_rn(d0, x2); b = __fmul_rn(d1, x1); c = __fmul_rn(d1, y1); d = __fmul_rn(d2, y0);
            }
            y2 = __fadd_rn(__fsub_rn(__fadd_rn(__fsub_rn(a, b), __fmul_rn(d0, x0)), -c), -d); // d0 x2 - d1 x1 + d0 x0 + d1 y1 - d2 y0
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = y2;
}

__global__ __launch_bounds__(64, 2) void aggressor(float* sink, int iters) {
    extern __shared__ float lds[]; // 18 KiB per wavefront, eight wavefronts per CU: 144 KiB
    float acc[160];
    for (int k = 0; k < 160; k++) acc[k] = threadIdx.x + k;
    for (int i = 0; i < iters; i++) {
        lds[(threadIdx.x + 64 * (i & 63)) % 4608] = acc[0] + (float)i;
        __builtin_amdgcn_wave_barrier();
        const float v = lds[(threadIdx.x * 7 + i) % 4608];
#pragma unroll
        for (int k = 0; k < 160; k++) acc[k] = acc[k] * 1.0000001f + v;
    }
    float s = 0;
    for (int k = 0; k < 160; k++) s += acc[k];
    sink[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s victim|victim_scalar|aggressor <seconds>\n", argv[0]); return 2; }
    const double secs = atof(argv[2]);
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (!strcmp(argv[1], "aggressor")) {
        float* sink; CK(hipMalloc(&sink, (size_t)8192 * 64 * 4));
        CK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 18432));
        long n = 0;
        while (elapsed() < secs) { hipLaunchKernelGGL(aggressor, dim3(8192), dim3(64), 18432, 0, sink, 6000); CK(hipDeviceSynchronize()); n++; }
        printf("aggressor: %ld launches in %.1f s\n", n, elapsed());
        return 0;
    }
    const int packed = strcmp(argv[1], "victim_scalar") != 0, n = 20000, waves = 16384;
    std::vector<float> in(n);
    unsigned r = 12345;
    for (int i = 0; i < n; i++) { r = r * 1664525u + 1013904223u; in[i] = ((int)(r >> 8) % 2001 - 1000) * 1e-3f; }
    float *d_in, *d_out; CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_out, (size_t)waves * 64 * 4));
    CK(hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> out((size_t)waves * 64);
    long launches = 0, bad_launches = 0, bad_waves = 0, lane_hist[64] = {0};
    unsigned want = 0; bool have = false;
    while (elapsed() < secs) {
        hipLaunchKernelGGL(victim, dim3(waves), dim3(64), 0, 0, d_in, d_out, n, packed);
        CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
        if (!have) { memcpy(&want, &out[1], 4); have = true; } // lane 1 of wavefront 0 of the first launch: every active lane of every launch must equal it
        long bw = 0;
        for (int w = 0; w < waves; w++) {
            bool bad = false;
            for (int l = 1; l < 64; l += 2) { unsigned v; memcpy(&v, &out[(size_t)w * 64 + l], 4); if (v != want) { bad = true; lane_hist[l]++; } }
            bw += bad;
        }
        launches++; bad_launches += bw > 0; bad_waves += bw;
    }
    printf("%s: %ld launches of %d wavefronts x %d samples in %.1f s: %ld launches with a wrong lane, %ld wavefronts; wrong values by lane:", argv[1], launches, waves, n, elapsed(), bad_launches, bad_waves);
    for (int l = 1; l < 64; l += 2) if (lane_hist[l]) printf(" %d:%ld", l, lane_hist[l]);
    printf("\n");
    return 0;
}
