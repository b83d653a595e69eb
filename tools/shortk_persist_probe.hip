// Short-K 1x1 layers (the p -> 4p convolutions of ResNet-50 at the bench batch: K = 64 / 128 / 256, output four times the
// input) sit at ~60 % of BOTH roofs in the step: a block loads, multiplies, stores, and only retires when its stores are
// acknowledged, so neither the read nor the write latency of a tile is hidden by anything but the other two blocks of the CU.
// This probe measures what a PERSISTENT form of the same fp32 GEMM buys on those shapes: each block walks a list of tiles,
// the first k-tile of the next tile is fetched before the epilogue of the current one, and the epilogue's stores are
// fire-and-forget (the accumulators are free again as soon as the store instructions have issued).
//   C[M][N] = A[M][K] * B[N][K]^T, fp32, 128 x 128 tile, 4 waves of 64 x 64, 32-k tiles, one LDS buffer, 3 blocks per CU.
// usage: ./shortk_persist_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;

// MODE 1: no MFMAs (loads, LDS traffic and stores only); MODE 2: no output stores (loads, LDS, MFMAs) -- timing only
template <bool PERSIST, int MODE = 0>
__global__ __launch_bounds__(256, 3) void gemm(const float* __restrict__ A, const float* __restrict__ B,
                                               float* __restrict__ C, int M, int N, int K, int ntiles) {
    __shared__ __attribute__((aligned(16))) float sA[BM * LDT], sB[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN, nk = K / BK;
    const int lr = tid >> 3, kq = tid & 7;
    f32x4 ra[4], rb[4];
    auto load = [&](int m0, int n0, int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A + (size_t)(m0 + lr + 32 * j) * K + kt * BK + kq * 4));
            rb[j] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + lr + 32 * j) * K + kt * BK + kq * 4);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4*>(sA + (lr + 32 * j) * LDT + kq * 4) = ra[j];
            *reinterpret_cast<f32x4*>(sB + (lr + 32 * j) * LDT + kq * 4) = rb[j];
        }
    };
    const int a_off = (wm * 64 + (lane & 31)) * LDT + (lane >> 5) * 4, b_off = (wn * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;
    f32x16 acc[2][2];
    auto mma_tile = [&]() {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(sA + a_off + i * 32 * LDT + kk * 8);
                b[i] = *reinterpret_cast<const f32x4*>(sB + b_off + i * 32 * LDT + kk * 8);
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][tt], b[j][tt], acc[i][j], 0, 0, 0);
        }
    };
    const int step = PERSIST ? gridDim.x : ntiles;
    int t = blockIdx.x;
    if (t >= ntiles) return;
    {
        const int mt = t / ntn;
        load(mt * BM, (t - mt * ntn) * BN, 0);
    }
    for (; t < ntiles; t += step) {
        const int mt = t / ntn, m0 = mt * BM, n0 = (t - mt * ntn) * BN;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        store();
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                load(m0, n0, kt + 1);
            } else if (PERSIST && t + step < ntiles) {       // the first k-tile of the NEXT tile, ahead of this tile's epilogue
                const int t2 = t + step, mt2 = t2 / ntn;
                load(mt2 * BM, (t2 - mt2 * ntn) * BN, 0);
            }
            if constexpr (MODE != 1) mma_tile();
            else acc[0][0][0] += sA[a_off] + sB[b_off];
            __syncthreads();
            if (kt + 1 < nk) {
                store();
                __syncthreads();
            }
        }
        if constexpr (MODE == 2) {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
            if (sacc == 12345.678f) C[t] = sacc;
            continue;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    __builtin_nontemporal_store(acc[i][j][r], C + (size_t)row * N + n0 + wn * 64 + j * 32 + (lane & 31));
                }
    }
}

// K = 64 only: the WHOLE reduction range of a tile resident in LDS (A 128 x 64 + B 128 x 64, 68-word pitch: 69.6 KB, two blocks
// per CU) -- one fetch per tile, issued a full tile ahead (before the 128 MFMAs of the current tile), one barrier pair per tile
__global__ __launch_bounds__(256, 2) void gemm_k64(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ C, int M, int N, int ntiles) {
    constexpr int K = 64, LP = 68;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sA = sm;
    float* sB = sm + BM * LP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN;
    const int lr = tid >> 4, kq = tid & 15;          // row lr + 16 j, float4 kq of the 16 per row
    f32x4 ra[8], rb[8];
    auto load = [&](int m0, int n0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ra[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A + (size_t)(m0 + lr + 16 * j) * K + kq * 4));
            rb[j] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + lr + 16 * j) * K + kq * 4);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *reinterpret_cast<f32x4*>(sA + (lr + 16 * j) * LP + kq * 4) = ra[j];
            *reinterpret_cast<f32x4*>(sB + (lr + 16 * j) * LP + kq * 4) = rb[j];
        }
    };
    const int a_off = (wm * 64 + (lane & 31)) * LP + (lane >> 5) * 4, b_off = (wn * 64 + (lane & 31)) * LP + (lane >> 5) * 4;
    f32x16 acc[2][2];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    {
        const int mt = t / ntn;
        load(mt * BM, (t - mt * ntn) * BN);
    }
    for (; t < ntiles; t += gridDim.x) {
        const int mt = t / ntn, m0 = mt * BM, n0 = (t - mt * ntn) * BN;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        store();
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) {
            const int t2 = t + gridDim.x, mt2 = t2 / ntn;
            load(mt2 * BM, (t2 - mt2 * ntn) * BN);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(sA + a_off + i * 32 * LP + kk * 8);
                b[i] = *reinterpret_cast<const f32x4*>(sB + b_off + i * 32 * LP + kk * 8);
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][tt], b[j][tt], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                 // every wave has read its last fragments
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    __builtin_nontemporal_store(acc[i][j][r], C + (size_t)row * N + n0 + wn * 64 + j * 32 + (lane & 31));
                }
    }
}

// Persistent, 32-k tiles, and the epilogue of tile i INTERLEAVED into the MFMAs of tile i + 1 (two accumulator sets: 64 / (4 NK)
// stores behind each group of 16 MFMAs) -- no phase in which a wave only stores, so the blocks of a CU cannot fall into a
// common store-then-multiply rhythm
template <int NK>
__global__ __launch_bounds__(256, 2) void gemm_il(const float* __restrict__ A, const float* __restrict__ B,
                                                  float* __restrict__ C, int M, int N, int ntiles) {
    constexpr int K = NK * BK, SPG = 64 / (4 * NK);      // stores per MFMA group
    __shared__ __attribute__((aligned(16))) float sA[BM * LDT], sB[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN;
    const int lr = tid >> 3, kq = tid & 7;
    f32x4 ra[4], rb[4];
    auto load = [&](int m0, int n0, int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A + (size_t)(m0 + lr + 32 * j) * K + kt * BK + kq * 4));
            rb[j] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + lr + 32 * j) * K + kt * BK + kq * 4);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4*>(sA + (lr + 32 * j) * LDT + kq * 4) = ra[j];
            *reinterpret_cast<f32x4*>(sB + (lr + 32 * j) * LDT + kq * 4) = rb[j];
        }
    };
    const int a_off = (wm * 64 + (lane & 31)) * LDT + (lane >> 5) * 4, b_off = (wn * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;
    f32x16 accA[2][2], accB[2][2];
    auto body = [&](f32x16 (&cur)[2][2], f32x16 (&old)[2][2], float* oldcb, int t) {
        const int mt = t / ntn, m0 = mt * BM, n0 = (t - mt * ntn) * BN;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[i][j][r] = 0.f;
        store();
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            if (kt + 1 < NK) {
                load(m0, n0, kt + 1);
            } else if (t + (int)gridDim.x < ntiles) {
                const int t2 = t + gridDim.x, mt2 = t2 / ntn;
                load(mt2 * BM, (t2 - mt2 * ntn) * BN, 0);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const f32x4*>(sA + a_off + i * 32 * LDT + kk * 8);
                    b[i] = *reinterpret_cast<const f32x4*>(sB + b_off + i * 32 * LDT + kk * 8);
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            cur[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][tt], b[j][tt], cur[i][j], 0, 0, 0);
                if (oldcb) {
#pragma unroll
                    for (int q = 0; q < SPG; ++q) {
                        const int e = (kt * 4 + kk) * SPG + q, i = e >> 5, j = (e >> 4) & 1, r = e & 15;     // compile-time
                        __builtin_nontemporal_store(old[i][j][r], oldcb + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (kt + 1 < NK) {
                store();
                __syncthreads();
            }
        }
    };
    auto cbase = [&](int t) {
        const int mt = t / ntn, m0 = mt * BM, n0 = (t - mt * ntn) * BN;
        return C + (size_t)(m0 + wm * 64 + 4 * (lane >> 5)) * N + n0 + wn * 64 + (lane & 31);
    };
    int t = blockIdx.x;
    if (t >= ntiles) return;
    {
        const int mt = t / ntn;
        load(mt * BM, (t - mt * ntn) * BN, 0);
    }
    float* pend = nullptr;          // output position of the tile whose stores are still owed (in accB before body(accA ..), ...)
    bool inA = false;
    for (; t < ntiles; t += 2 * gridDim.x) {
        body(accA, accB, pend, t);
        pend = cbase(t); inA = true;
        const int t1 = t + gridDim.x;
        if (t1 < ntiles) {
            body(accB, accA, pend, t1);
            pend = cbase(t1); inA = false;
        }
    }
#pragma unroll
    for (int e = 0; e < 64; ++e) {
        const int i = e >> 5, j = (e >> 4) & 1, r = e & 15;
        const float v = inA ? accA[i][j][r] : accB[i][j][r];
        __builtin_nontemporal_store(v, pend + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N + j * 32);
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

__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned s = (unsigned)i * 2654435761u + seed;
        s ^= s >> 13; s *= 1274126177u; s ^= s >> 16;
        p[i] = (float)(s & 0xffff) / 65536.0f - 0.5f;
    }
}

int main() {
    const int shapes[4][3] = {{2097152, 256, 64}, {524288, 512, 128}, {131072, 1024, 256}, {32768, 2048, 512}};
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    for (int si = 0; si < 4; ++si) {
        const int M = shapes[si][0], N = shapes[si][1], K = shapes[si][2];
        // four rotating (A, C) sets so that nothing is served from the 256 MB MALL
        float *dA[4], *dC[4], *dB;
        for (int r = 0; r < 4; ++r) {
            CK(hipMalloc(&dA[r], (size_t)M * K * 4)); CK(hipMalloc(&dC[r], (size_t)M * N * 4));
            hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, dA[r], (size_t)M * K, 17u + r);
        }
        CK(hipMalloc(&dB, (size_t)N * K * 4));
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, dB, (size_t)N * K, 99u);
        const int ntiles = (M / BM) * (N / BN);
        const double flop = 2.0 * M * N * K, bytes = 4.0 * ((double)M * K + (double)M * N + (double)N * K);
        int rot = 0;
        double ms0 = time_ms([&]() { hipLaunchKernelGGL(gemm<false>, dim3(ntiles), dim3(256), 0, 0, dA[rot & 3], dB, dC[rot & 3], M, N, K, ntiles); ++rot; });
        printf("M=%8d K=%4d N=%5d   one tile per block  %6.3f ms  %6.1f TF/s  %5.2f TB/s", M, K, N, ms0, flop / ms0 / 1e9, bytes / ms0 / 1e9);
        for (int bpc = 2; bpc <= 3; ++bpc) {
            const int grid = ncu * bpc < ntiles ? ncu * bpc : ntiles;
            double ms1 = time_ms([&]() { hipLaunchKernelGGL(gemm<true>, dim3(grid), dim3(256), 0, 0, dA[rot & 3], dB, dC[rot & 3], M, N, K, ntiles); ++rot; });
            printf(" | persistent x%d  %6.3f ms  %6.1f TF/s  %5.2f TB/s", bpc, ms1, flop / ms1 / 1e9, bytes / ms1 / 1e9);
        }
        {
            const int grid = ncu * 2 < ntiles ? ncu * 2 : ntiles;
            double m1 = time_ms([&]() { hipLaunchKernelGGL((gemm<true, 1>), dim3(grid), dim3(256), 0, 0, dA[rot & 3], dB, dC[rot & 3], M, N, K, ntiles); ++rot; });
            double m2 = time_ms([&]() { hipLaunchKernelGGL((gemm<true, 2>), dim3(grid), dim3(256), 0, 0, dA[rot & 3], dB, dC[rot & 3], M, N, K, ntiles); ++rot; });
            printf(" | no MFMAs %6.3f ms (%5.2f TB/s), no stores %6.3f ms (%6.1f TF/s)", m1, bytes / m1 / 1e9, m2, flop / m2 / 1e9);
        }
        if (K == 64) {
            const size_t lds = (size_t)2 * 128 * 68 * 4;
            CK(hipFuncSetAttribute((const void*)gemm_k64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = ncu * 2 < ntiles ? ncu * 2 : ntiles;
            double ms2 = time_ms([&]() { hipLaunchKernelGGL(gemm_k64, dim3(grid), dim3(256), lds, 0, dA[rot & 3], dB, dC[rot & 3], M, N, ntiles); ++rot; });
            printf(" | whole-K persistent x2  %6.3f ms  %6.1f TF/s  %5.2f TB/s", ms2, flop / ms2 / 1e9, bytes / ms2 / 1e9);

        }
        if (K <= 256) {
            const int grid = ncu * 2 < ntiles ? ncu * 2 : ntiles;
            auto run_il = [&](float* a, float* c) {
                if (K == 64) hipLaunchKernelGGL(gemm_il<2>, dim3(grid), dim3(256), 0, 0, a, dB, c, M, N, ntiles);
                else if (K == 128) hipLaunchKernelGGL(gemm_il<4>, dim3(grid), dim3(256), 0, 0, a, dB, c, M, N, ntiles);
                else hipLaunchKernelGGL(gemm_il<8>, dim3(grid), dim3(256), 0, 0, a, dB, c, M, N, ntiles);
            };
            double ms3 = time_ms([&]() { run_il(dA[rot & 3], dC[rot & 3]); ++rot; });
            printf(" | persistent x2 + interleaved epilogue  %6.3f ms  %6.1f TF/s  %5.2f TB/s", ms3, flop / ms3 / 1e9, bytes / ms3 / 1e9);
            run_il(dA[0], dC[2]);
            hipLaunchKernelGGL(gemm<false>, dim3(ntiles), dim3(256), 0, 0, dA[0], dB, dC[3], M, N, K, ntiles);
            CK(hipDeviceSynchronize());
            float g0[64], g1[64];
            int badil = 0;
            for (int probe = 0; probe < 4; ++probe) {
                const size_t off = (size_t)((long)(M - 1) * probe / 3) * N + (probe * 64) % (N - 63);
                CK(hipMemcpy(g0, dC[2] + off, 256, hipMemcpyDeviceToHost));
                CK(hipMemcpy(g1, dC[3] + off, 256, hipMemcpyDeviceToHost));
                for (int i = 0; i < 64; ++i) badil += g0[i] != g1[i];
            }
            if (badil) printf(" [interleaved form MISMATCH %d]", badil);
        }
        printf("\n");
        // spot check: the two forms agree bit for bit
        hipLaunchKernelGGL(gemm<false>, dim3(ntiles), dim3(256), 0, 0, dA[0], dB, dC[0], M, N, K, ntiles);
        hipLaunchKernelGGL(gemm<true>, dim3(ncu * 3 < ntiles ? ncu * 3 : ntiles), dim3(256), 0, 0, dA[0], dB, dC[1], M, N, K, ntiles);
        float h0[64], h1[64];
        CK(hipMemcpy(h0, dC[0] + (size_t)(M - 1) * N + N - 64, 256, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1, dC[1] + (size_t)(M - 1) * N + N - 64, 256, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 64; ++i) bad += h0[i] != h1[i];
        if (bad) printf("  MISMATCH between the two forms (%d of 64)\n", bad);
        for (int r = 0; r < 4; ++r) { CK(hipFree(dA[r])); CK(hipFree(dC[r])); }
        CK(hipFree(dB));
    }
    return 0;
}
