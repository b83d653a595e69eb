// Write-bandwidth probe: how fast can a 128 x 128-tile epilogue stream a [M][C] tensor to HBM, as a function of the
// store shape?  MODE 0: the MFMA accumulator layout of the convolution kernels (a lane owns one column: 2-byte (bf16) or
// 4-byte (fp32) elements, 32 lanes = one 64 / 128-byte row segment, 64 store instructions per lane and tile);
// MODE 1: 16 bytes per lane, a wave writes whole rows (what an LDS-transposed epilogue would issue: 8 / 16 instructions).
// Buffers are 2 GiB (no cache warmth).  build: hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ES, int MODE, int NT>   // ES: element bytes; NT: non-temporal aux
__global__ __launch_bounds__(256, 3) void store_kernel(char* out, int C, int ntn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x, mt = tile / ntn, n0 = (tile - mt * ntn) * 128, m0 = mt * 128;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m0 * C * ES, 0, 128 * C * ES, 0x00020000);
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                    const unsigned off = (unsigned)(row * C + col) * ES;
                    if (ES == 2) __builtin_amdgcn_raw_buffer_store_b16((short)(row + col), rs, off, 0, NT);
                    else __builtin_amdgcn_raw_buffer_store_b32(row + col, rs, off, 0, NT);
                }
            }
    } else {
        // the wave's 64 x 64 quadrant, 16 bytes per lane: ES = 2: a row is 128 B = 8 lanes, 8 rows per instruction;
        // ES = 4: 256 B = 16 lanes, 4 rows per instruction
        constexpr int LPR = 64 * ES / 16, RPI = 64 / LPR, NI = 64 / RPI;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int row = wm * 64 + k * RPI + lane / LPR, col = n0 + wn * 64 + (lane % LPR) * (16 / ES);
            const u32x4 v = {(unsigned)row, (unsigned)col, 0u, 1u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)(row * C + col) * ES, 0, NT);
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int ES, int MODE, int NT> float run(char* buf, int M, int C) {
    const int ntn = C / 128, tiles = (M / 128) * ntn;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((store_kernel<ES, MODE, NT>), dim3(tiles), dim3(256), 0, 0, buf, C, ntn);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    char* buf;
    const size_t bytes = (size_t)2 << 30;
    CK(hipMalloc(&buf, bytes));
    for (int C : {256, 1024}) {
        {
            const int M = (int)(bytes / ((size_t)C * 2));
            const double gb = (double)M * C * 2 / 1e9;
            printf("bf16 [%d][%d]: lane-per-column 2 B stores %.2f TB/s (nt %.2f) | 16 B per lane %.2f TB/s (nt %.2f)\n", M, C,
                   gb / run<2, 0, 0>(buf, M, C), gb / run<2, 0, 2>(buf, M, C), gb / run<2, 1, 0>(buf, M, C), gb / run<2, 1, 2>(buf, M, C));
        }
        {
            const int M = (int)(bytes / ((size_t)C * 4));
            const double gb = (double)M * C * 4 / 1e9;
            printf("fp32 [%d][%d]: lane-per-column 4 B stores %.2f TB/s (nt %.2f) | 16 B per lane %.2f TB/s (nt %.2f)\n", M, C,
                   gb / run<4, 0, 0>(buf, M, C), gb / run<4, 0, 2>(buf, M, C), gb / run<4, 1, 0>(buf, M, C), gb / run<4, 1, 2>(buf, M, C));
        }
    }
    return 0;
}
