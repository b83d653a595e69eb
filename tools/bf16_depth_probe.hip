// How much of the bf16 NT kernel's distance from the matrix pipe (0.8-0.9 PF/s on the 3x3 / long-K layers against 2.5 PF/s) is
// the depth of its global prefetch?  A k-tile of 64 bf16 is 16 MFMAs = 512 cycles per wave, an HBM / L2 round trip is several
// thousand: with one k-tile in flight per block and three blocks per CU the loop waits for memory most of the time.
//   C[M][N] = A[M][K] * B[N][K]^T, bf16 in, fp32 accumulate, bf16 out; 128 x 128 tile, 4 waves of 64 x 64, 64-k tiles, ONE LDS
//   buffer (two barriers per k-tile); DEPTH k-tiles of register prefetch (1: the library's form), MINB blocks per CU.
// usage: ./bf16_depth_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BK = 64, P = 72;      // LDS rows: 64 bf16 + 8 pad = 144 bytes

// TI x TJ: 32 x 32 MFMA tiles per wave (2 x 2 waves per block): block tile 64 TI x 64 TJ.  A fragment read serves TJ MFMAs, a B
// fragment read TI: (TI + TJ) / (TI TJ) 16-byte reads per MFMA -- 1 for 2 x 2, 0.75 for 2 x 4, 0.5 for 4 x 4
template <int DEPTH, int MINB, int TI = 2, int TJ = 2>
__global__ __launch_bounds__(256, MINB) void gemm(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                  bf16_t* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 64 * TI, BN = 64 * TJ, AR = BM / 32, BR = BN / 32;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* sA = smem;
    bf16_t* sB = smem + BM * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN, mt = blockIdx.x / ntn, m0 = mt * BM, n0 = (blockIdx.x - mt * ntn) * BN;
    const int lr = tid >> 3, kq = tid & 7;          // row lr + 32 j, 16-byte chunk kq of the 8 per row
    u32x4 ra[DEPTH][AR], rb[DEPTH][BR];
    const int nk = K / BK;
    auto load = [&](int kt, u32x4 (&a)[AR], u32x4 (&b)[BR]) {
        const bool live = kt < nk;
        const int k0 = (live ? kt : 0) * BK;
#pragma unroll
        for (int j = 0; j < AR; ++j) a[j] = *reinterpret_cast<const u32x4*>(A + (size_t)(m0 + lr + 32 * j) * K + k0 + kq * 8);
#pragma unroll
        for (int j = 0; j < BR; ++j) b[j] = *reinterpret_cast<const u32x4*>(B + (size_t)(n0 + lr + 32 * j) * K + k0 + kq * 8);
    };
    auto store = [&](u32x4 (&a)[AR], u32x4 (&b)[BR]) {
#pragma unroll
        for (int j = 0; j < AR; ++j) *reinterpret_cast<u32x4*>(sA + (lr + 32 * j) * P + kq * 8) = a[j];
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<u32x4*>(sB + (lr + 32 * j) * P + kq * 8) = b[j];
    };
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int a_off = (wm * 32 * TI + (lane & 31)) * P + (lane >> 5) * 8, b_off = (wn * 32 * TJ + (lane & 31)) * P + (lane >> 5) * 8;
    auto mma_tile = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sA + a_off + i * 32 * P + s * 16);
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sB + b_off + j * 32 * P + s * 16);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    // set d holds tile kt + 1 + d when tile kt is being multiplied
    load(0, ra[0], rb[0]);
    store(ra[0], rb[0]);
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(1 + d, ra[d], rb[d]);
    for (int kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {           // multiplies tile kt + d; then refills LDS from set d (tile kt + d + 1)
            if (kt + d >= nk) break;
            mma_tile();
            __syncthreads();
            store(ra[d], rb[d]);
            __syncthreads();
            load(kt + d + 1 + DEPTH, ra[d], rb[d]);
        }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const f32x2 v = {acc[i][j][r], 0.f};
                C[(size_t)row * N + n0 + wn * 32 * TJ + j * 32 + (lane & 31)] =
                    (bf16_t)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)) & 0xffffu);
            }
}

template <typename F> double time_ms(F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms / 10 < best ? ms / 10 : best;
    }
    return best;
}

__global__ void fill(bf16_t* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned s = (unsigned)i * 2654435761u + seed;
        s ^= s >> 13; s *= 1274126177u; s ^= s >> 16;
        const float v = (float)(s & 0xffff) / 65536.0f - 0.5f;
        p[i] = (bf16_t)(__builtin_bit_cast(unsigned, v) >> 16);
    }
}

int main() {
    // 3x3 256 -> 256 as a GEMM (K = 2304), 1x1 1024 -> 256, 1x1 256 -> 1024, 1x1 512 -> 2048 at 8x8, 1x1 128 -> 512 at 32x32
    const int shapes[5][3] = {{131072, 256, 2304}, {131072, 256, 1024}, {131072, 1024, 256}, {32768, 2048, 512}, {524288, 512, 128}};
    for (int si = 0; si < 5; ++si) {
        const int M = shapes[si][0], N = shapes[si][1], K = shapes[si][2];
        bf16_t *dA[4], *dC[4], *dB;
        for (int r = 0; r < 4; ++r) {
            CK(hipMalloc(&dA[r], (size_t)M * K * 2)); CK(hipMalloc(&dC[r], (size_t)M * N * 2));
            hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, dA[r], (size_t)M * K, 17u + r);
        }
        CK(hipMalloc(&dB, (size_t)N * K * 2));
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, dB, (size_t)N * K, 99u);

        const double flop = 2.0 * M * N * K, bytes = 2.0 * ((double)M * K + (double)M * N + (double)N * K);
        int rot = 0;
        printf("M=%7d N=%5d K=%5d :", M, N, K);
#define RUN(D_, B_, TI_, TJ_)                                                                                                  \
        {                                                                                                                     \
            const size_t lds = (size_t)(64 * TI_ + 64 * TJ_) * P * 2;                                                         \
            CK(hipFuncSetAttribute((const void*)(gemm<D_, B_, TI_, TJ_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            const dim3 grid((M / (64 * TI_)) * (N / (64 * TJ_)));                                                             \
            double ms = time_ms([&]() { hipLaunchKernelGGL((gemm<D_, B_, TI_, TJ_>), grid, dim3(256), lds, 0, dA[rot & 3], dB, dC[rot & 3], M, N, K); ++rot; }); \
            printf("  %dx%d tiles/wave, depth %d, %d blocks/CU %6.3f ms %5.0f TF/s |", TI_, TJ_, D_, B_, ms, flop / ms / 1e9);    \
        }
        RUN(1, 3, 2, 2) RUN(2, 2, 2, 2) RUN(1, 2, 2, 4) RUN(2, 2, 2, 4) RUN(1, 1, 4, 4) RUN(2, 1, 4, 4)
        printf("\n");
        for (int r = 0; r < 4; ++r) { CK(hipFree(dA[r])); CK(hipFree(dC[r])); }
        CK(hipFree(dB));
    }
    return 0;
}
