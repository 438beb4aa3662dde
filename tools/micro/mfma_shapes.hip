// mfma_shapes.hip -- cycles and FLOP/s of the bf16 MFMA shapes on random operands (development aid, round 3).
// Question it answers: what does a K = 104 inner product (d = 100 spatial slots + 4 time slots) cost per 64 x 32
// output tile of one wave when built from
//   mode 0: 7 x v_mfma_f32_32x32x16_bf16                    (K padded to 112; the round-2 kernel)
//   mode 1: 4 x v_mfma_f32_16x16x32_bf16                    (K padded to 128)
//   mode 2: 6 x 32x32x16 + 1 x v_mfma_f32_32x32x8_bf16_1k   (K = 104 exactly)
//   mode 3: 3 x 16x16x32 + 1 x v_mfma_f32_16x16x16_bf16_1k  (K = 112)
//   mode 4: 3 x 16x16x32 + 1 x 16x16x16 with half its slots dead is the same instruction count as mode 3: not separate
// Operands in registers, no memory traffic; WPS = waves per SIMD (1: 256 blocks of 256 threads, 2: 512 blocks).
// Prints one JSON line per case: ms, TFLOP/s of ALGORITHMIC flop (2 * 104 per output element), in-kernel clock
// (s_memtime / s_memrealtime) and shader cycles per output tile.
// build: hipcc --offload-arch=gfx950 -O3 -o build_variants/mfma_shapes tools/micro/mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ short rnd_bf16(uint32_t& s)
{
    s = s * 1664525u + 1013904223u;
    return (short)(0x3c00 + ((s >> 9) & 0x01ff) + ((s >> 3) & 0x8000));     // +-[0.0078, 0.0156): finite
}

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void mfma_loop(int iters, uint32_t seed, float* out, unsigned long long* stamps)
{
    uint32_t s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    s16x8 a[8], b[8];
    for (int q = 0; q < 8; ++q)
        for (int k = 0; k < 8; ++k) { a[q][k] = rnd_bf16(s); b[q][k] = rnd_bf16(s); }
    s16x4 a4[2], b4[2];
    for (int q = 0; q < 2; ++q)
        for (int k = 0; k < 4; ++k) { a4[q][k] = rnd_bf16(s); b4[q][k] = rnd_bf16(s); }
    float r = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (MODE == 0 || MODE == 2) {
        f32x16 acc[2];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
                for (int e = 0; e < 16; ++e) acc[tm][e] = 0.0f;
#pragma unroll
            for (int g = 0; g < (MODE == 0 ? 7 : 6); ++g)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(g + tm) & 7]), __builtin_bit_cast(bf16x8, b[g]),
                                                                      acc[tm], 0, 0, 0);
            if constexpr (MODE == 2) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4[tm], b4[0], acc[tm], 0, 0, 0);
            }
            float m = acc[0][0];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int e = 0; e < 16; ++e) m = __builtin_fmaxf(m, acc[tm][e]);
            r = __builtin_fmaxf(r, m);
        }
    } else {
        f32x4 acc[8];                                  // 4 row tiles x 2 column tiles of 16 x 16
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                for (int e = 0; e < 4; ++e) acc[t][e] = 0.0f;
#pragma unroll
            for (int g = 0; g < (MODE == 1 ? 4 : 3); ++g)
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(g + (t >> 1)) & 7]),
                                                                     __builtin_bit_cast(bf16x8, b[(2 * g + (t & 1)) & 7]), acc[t], 0, 0, 0);
            if constexpr (MODE == 3) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4[(t >> 1) & 1], b4[t & 1], acc[t], 0, 0, 0);
            }
            float m = acc[0][0];
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = __builtin_fmaxf(m, acc[t][e]);
            r = __builtin_fmaxf(r, m);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    if (r == 123.456f) out[0] = r;                      // keeps the loop alive
}

static int cmp_d(const void* x, const void* y) { const double a = *(const double*)x, b = *(const double*)y; return a < b ? -1 : a > b; }

template <int MODE, int WPS>
static void run(const char* name, int iters, float* out, unsigned long long* stamps, unsigned long long* h)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * WPS;
    for (int rep = 0; rep < 3; ++rep) {                   // rep 0 warms clocks
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, iters, 12345u + rep, out, stamps);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double clk[1024], cyc[1024];
        for (int b = 0; b < blocks; ++b) { clk[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; cyc[b] = (double)h[2 * b] / iters; }
        qsort(clk, blocks, sizeof(double), cmp_d);
        qsort(cyc, blocks, sizeof(double), cmp_d);
        const double flops = (double)blocks * 4 * iters * 64.0 * 32.0 * 2.0 * 104.0;
        if (rep)
            printf("{\"case\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"algorithmic_tflops\": %.1f, \"frac_of_2500\": %.3f, \"clock_ghz\": %.3f, "
                   "\"cycles_per_64x32_tile\": %.1f}\n",
                   name, WPS, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2500.0, clk[blocks / 2], cyc[blocks / 2]);
    }
    fflush(stdout);
}

int main(int argc, char** argv)
{
    float* out = nullptr;
    unsigned long long *stamps = nullptr, *h = (unsigned long long*)malloc(sizeof(unsigned long long) * 2048);
    if (hipMalloc(&out, 4) != hipSuccess) { printf("{\"error\": \"no device\"}\n"); return 1; }
    hipMalloc(&stamps, sizeof(unsigned long long) * 2048);
    const int iters = argc > 1 ? atoi(argv[1]) : 60000;
    run<0, 1>("7 x 32x32x16 (K 112)", iters, out, stamps, h);
    run<0, 2>("7 x 32x32x16 (K 112)", iters, out, stamps, h);
    run<1, 1>("4 x 16x16x32 (K 128)", iters, out, stamps, h);
    run<1, 2>("4 x 16x16x32 (K 128)", iters, out, stamps, h);
    run<2, 1>("6 x 32x32x16 + 32x32x8 (K 104)", iters, out, stamps, h);
    run<2, 2>("6 x 32x32x16 + 32x32x8 (K 104)", iters, out, stamps, h);
    run<3, 1>("3 x 16x16x32 + 16x16x16 (K 112)", iters, out, stamps, h);
    run<3, 2>("3 x 16x16x32 + 16x16x16 (K 112)", iters, out, stamps, h);
    return 0;
}
