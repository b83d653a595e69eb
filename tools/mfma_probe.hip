// Sustained fp32 matrix rate of gfx950 under load: v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32, operands in
// registers (random data, so the power model sees real toggling), W waves per SIMD.  Both have the same peak (64 FLOP / clk /
// SIMD); the question is which clock the chip holds under each.  usage: ./mfma_probe   (prints TF/s per variant)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(t * 16 + i) & 0xffff]; b[i] = src[(t * 16 + 8 + i) & 0xffff]; }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + j) & 7], b[k], acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    dst[t] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(t * 16 + i) & 0xffff]; b[i] = src[(t * 16 + 8 + i) & 0xffff]; }
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + j) & 7], b[k], acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    dst[t] = s;
}

template <typename F> double run(F launch, double flop_per_launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return flop_per_launch * 20 / (ms * 1e-3) / 1e12;
}

int main() {
    float *src, *dst;
    hipMalloc(&src, 65536 * 4); hipMalloc(&dst, 256 * 256 * 16 * 4);
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int wps = 1; wps <= 3; ++wps) {          // waves per SIMD = blocks per CU (4 waves per block)
        const int blocks = 256 * wps;
        const double w = (double)blocks * 4;       // waves
        double t32 = run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, src, dst, iters); },
                         w * iters * 8 * 4 * 2.0 * 32 * 32 * 2);
        double t16 = run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, src, dst, iters * 2); },
                         w * iters * 2 * 8 * 8 * 2.0 * 16 * 16 * 4);
        printf("%d wave(s)/SIMD: 32x32x2 (4 acc) %.1f TF/s   16x16x4 (8 acc) %.1f TF/s\n", wps, t32, t16);
    }
    return 0;
}
