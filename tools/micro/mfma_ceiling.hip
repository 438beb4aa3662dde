// mfma_ceiling.hip -- what back-to-back v_mfma_f32_32x32x16_bf16 alone sustain on this chip (development aid).
// Operands stay in registers, four independent accumulators per wave, 2 x 256-thread blocks per CU like the scan; the
// operand bits are either zero or random finite bf16 values (the data the matrix cores see changes the power they
// draw, and with it the clock).  Prints one JSON line per case: TFLOP/s over a timed launch (HIP events).
// build: hipcc --offload-arch=gfx950 -O3 -o build_variants/mfma_ceiling tools/micro/mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void mfma_loop(int iters, uint32_t seed, int random_bits, float* out)
{
    s16x8 a[2], b[2];
    uint32_t s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    for (int q = 0; q < 2; ++q)
        for (int k = 0; k < 8; ++k) {
            s = s * 1664525u + 1013904223u;
            const short va = random_bits ? (short)(0x3c00 + ((s >> 8) & 0x3ff) * 0 + ((s >> 9) & 0x01ff) + ((s >> 3) & 0x8000)) : (short)0;   // +-[0.0078, 0.0156): finite
            s = s * 1664525u + 1013904223u;
            const short vb = random_bits ? (short)(0x3c00 + ((s >> 9) & 0x01ff) + ((s >> 3) & 0x8000)) : (short)0;
            a[q][k] = va; b[q][k] = vb;
        }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q & 1]), __builtin_bit_cast(bf16x8, b[q >> 1]), acc[q], 0, 0, 0);
    }
    float r = 0.0f;
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 16; ++e) r += acc[q][e];
    if (r == 123.456f) out[0] = r;                      // keeps the loop alive
}

int main(int argc, char** argv)
{
    float* out = nullptr;
    if (hipMalloc(&out, 4) != hipSuccess) { printf("{\"error\": \"no device\"}\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 2;
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int rnd = 0; rnd < 2; ++rnd) {
        for (int rep = 0; rep < 3; ++rep) {               // rep 0 warms clocks and caches
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, iters, 12345u + rep, rnd, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 2.0 * 32 * 32 * 16;
            if (rep) printf("{\"kernel\": \"v_mfma_f32_32x32x16_bf16 only\", \"operands\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f, \"frac_of_2500\": %.3f}\n",
                            rnd ? "random" : "zero", ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 2500.0);
        }
    }
    return 0;
}
