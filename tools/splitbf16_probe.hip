// Verdict item 8 (round 3): what would an fp32 GEMM cost on gfx950 if every fp32 product were SPLIT into bf16 terms --
// x = hi + mid + lo (three bf16 values, 24 mantissa bits together), x * y ~ hi*hi' + hi*mid' + mid*hi' + mid*mid' + hi*lo' +
// lo*hi' (six v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulation; the three dropped cross terms are below 2^-24 of the
// product) -- against the native v_mfma_f32_32x32x2_f32 GEMM of the SAME structure?  A microbenchmark, NOT the product path:
// the headline stays on the native fp32 MFMA.
//
// Both kernels: C[M][N] = A[M][K] * B[N][K]^T, fp32 in / fp32 out, 128 x 128 block tile, 4 waves of 64 x 64, 32-k tiles
// through one LDS buffer (two barriers per k-tile), global loads of tile k+1 issued ahead of the MFMAs of tile k.
//   native : LDS holds fp32 rows (36-word pitch), a 16-byte fragment read feeds 4 MFMAs (k = 2 each): 64 MFMAs of 64 cycles
//            per wave and k-tile.
//   split  : A is split while it is staged (global fp32 -> three bf16 planes in LDS: 2 cvt_pk + 2 shifts + 2 subs per pair
//            and plane), B arrives pre-split ([3][N][K] bf16 -- filters are re-laid out once per step anyway);
//            a 16-byte fragment read (8 bf16) feeds one MFMA of k = 16: 48 MFMAs of 32 cycles per wave and k-tile,
//            i.e. 0.375 of the native matrix time.
// Shapes: the 3x3 256 -> 256 layer at the bench batch as a plain GEMM (M = 131072, N = 256, K = 2304) and the 1x1
// 1024 -> 256 layer (M = 131072, N = 256, K = 1024).  Reported: fp32-equivalent TF/s (2 M N K / time) and the error of
// 512 sampled outputs against fp64 on the host (max and rms, relative to the rms of the output).
// usage: ./splitbf16_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32: round to nearest even
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// x -> (hi, mid, lo) for a pair of values; packed results (low half = first value)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pk_bf16(a, b);
    const float ra = a - bf_lo(hi), rb = b - bf_hi(hi);
    mid = pk_bf16(ra, rb);
    lo = pk_bf16(ra - bf_lo(mid), rb - bf_hi(mid));
}

__global__ void split_rows_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, size_t n) {    // out[3][n]
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * 256) {
        unsigned hi, mid, lo;
        split2(w[2 * i], w[2 * i + 1], hi, mid, lo);
        reinterpret_cast<unsigned*>(out)[i] = hi;
        reinterpret_cast<unsigned*>(out + n)[i] = mid;
        reinterpret_cast<unsigned*>(out + 2 * n)[i] = lo;
    }
}

constexpr int BM = 128, BN = 128, BK = 32;

// ---- native fp32 -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_native(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int M, int N, int K) {
    constexpr int LDT = 36;
    __shared__ __attribute__((aligned(16))) float sA[BM * LDT], sB[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN, mt = blockIdx.x / ntn, m0 = mt * BM, n0 = (blockIdx.x - mt * ntn) * BN;
    const int lr = tid >> 3, kq = tid & 7;
    f32x4 ra[4], rb[4];
    auto load = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + lr + 32 * j) * K + kt * BK + kq * 4);
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
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;
    load(0);
    store();
    __syncthreads();
    const int a_off = (wm * 64 + (lane & 31)) * LDT + (lane >> 5) * 4, b_off = (wn * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load(kt + 1);
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
        __syncthreads();
        if (kt + 1 < nk) store();
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(size_t)row * N + n0 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
}

// ---- split bf16 ------------------------------------------------------------------------------------------------
// LDS planes [3][128 rows][40 bf16]: 80-byte rows (16 consecutive rows of one 16-byte column hit 16 distinct bank quads)
// PRE: A arrives pre-split too ([3][M][K] bf16, 6 bytes per element instead of 4: what the producing layer's epilogue would write) --
// the bound of the form without the split VALU work in the GEMM
template <int TERMS, bool PRE = false>      // 6: the form above; 3: hi*hi' + hi*mid' + mid*hi' only (16 mantissa bits: the usual "bf16x3")
__global__ __launch_bounds__(256, 2) void gemm_split(const float* __restrict__ A, const bf16_t* __restrict__ As,
                                                     const bf16_t* __restrict__ Bs, float* __restrict__ C, int M, int N, int K) {
    constexpr int P = 40;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* sA = smem;                  // [3][BM][P]
    bf16_t* sB = smem + 3 * BM * P;     // [3][BN][P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN, mt = blockIdx.x / ntn, m0 = mt * BM, n0 = (blockIdx.x - mt * ntn) * BN;
    const int lr = tid >> 3, kq = tid & 7;          // A: row lr + 32 j, float4 kq
    const int br = tid >> 2, bq = tid & 3;          // B: row br + 64 j, 16-byte chunk bq (8 bf16), per plane
    const size_t plane = (size_t)N * K;
    f32x4 ra[4];
    u32x4 rb[6], rp[PRE ? 6 : 1];
    const size_t aplane = (size_t)M * K;
    auto load = [&](int kt) {
        if constexpr (PRE) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    rp[p * 2 + j] = *reinterpret_cast<const u32x4*>(As + p * aplane + (size_t)(m0 + br + 64 * j) * K + kt * BK + bq * 8);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ra[j] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + lr + 32 * j) * K + kt * BK + kq * 4);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                rb[p * 2 + j] = *reinterpret_cast<const u32x4*>(Bs + p * plane + (size_t)(n0 + br + 64 * j) * K + kt * BK + bq * 8);
    };
    auto store = [&]() {
        if constexpr (PRE) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<u32x4*>(sA + p * BM * P + (br + 64 * j) * P + bq * 8) = rp[p * 2 + j];
        } else
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split2(ra[j][0], ra[j][1], h0, m0_, l0);
            split2(ra[j][2], ra[j][3], h1, m1, l1);
            bf16_t* d = sA + (lr + 32 * j) * P + kq * 4;
            *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(d + BM * P) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(d + 2 * BM * P) = u32x2{l0, l1};
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<u32x4*>(sB + p * BN * P + (br + 64 * j) * P + bq * 8) = rb[p * 2 + j];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;
    load(0);
    store();
    __syncthreads();
    const int a_off = (wm * 64 + (lane & 31)) * P + (lane >> 5) * 8, b_off = (wn * 64 + (lane & 31)) * P + (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load(kt + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[p][i] = *reinterpret_cast<const bf16x8*>(sA + p * BM * P + a_off + i * 32 * P + s * 16);
                    b[p][i] = *reinterpret_cast<const bf16x8*>(sB + p * BN * P + b_off + i * 32 * P + s * 16);
                }
            // small terms first
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (TERMS == 6) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kt + 1 < nk) store();
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(size_t)row * N + n0 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
}


// ---- round 6: the same product on the tile structure of csrc/conv_p256.hip ------------------------------------------------
// 256 x 128 block tile, 8 waves as 4 x 2 (wave = 64 rows x 64 columns), 32-k tiles, TWO LDS stages, both operands global -> LDS by
// LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass): A lands as RAW fp32 (128-byte rows, the XOR-swizzled
// image of conv_p256) and is split into its three bf16 terms AT FRAGMENT-READ TIME (two 16-byte reads -> 8 floats -> 3 x bf16x8:
// 44 VALU per fragment, under 24 MFMAs per 16-k step); B arrives pre-split ([3][N][K] bf16: 64-byte rows per plane and k-tile,
// slot = chunk ^ ((row >> 2) & 3)).  One barrier per k-tile, next k-tile in flight under the MFMAs of this one.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t p_rsrc(const void* p, size_t bytes) {
    const unsigned long long a = (unsigned long long)p;
    const u32x4_t r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes, 0x00020000u};
    return r;
}
__device__ __forceinline__ void p_dma16(u32x4_t rs, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    unsigned hh[4], mm[4], ll[4];
    split2(x0[0], x0[1], hh[0], mm[0], ll[0]);
    split2(x0[2], x0[3], hh[1], mm[1], ll[1]);
    split2(x1[0], x1[1], hh[2], mm[2], ll[2]);
    split2(x1[2], x1[3], hh[3], mm[3], ll[3]);
    const u32x4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
    hi = __builtin_bit_cast(bf16x8, h);
    mid = __builtin_bit_cast(bf16x8, m);
    lo = __builtin_bit_cast(bf16x8, l);
}
template <int TERMS>
__global__ __launch_bounds__(512, 2) void gemm_split_p256(const float* __restrict__ A, const bf16_t* __restrict__ Bs,
                                                          float* __restrict__ C, int M, int N, int K) {
    constexpr int PBM = 256, PBN = 128;
    constexpr int A_BYTES = PBM * 128, B_PLANE = PBN * 64, STAGE = A_BYTES + 3 * B_PLANE;      // 32 KB + 24 KB
    extern __shared__ __attribute__((aligned(1024))) char psm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / PBN, mt = blockIdx.x / ntn, m0 = mt * PBM, n0 = (blockIdx.x - mt * ntn) * PBN;
    const unsigned lds0 = (unsigned)(size_t)psm;
    const u32x4_t rsA = p_rsrc(A + (size_t)m0 * K, (size_t)PBM * K * 4);
    const size_t plane = (size_t)N * K;
    // A: an instruction fills 8 rows x 128 B; lane i -> row i >> 3, slot i & 7 holds chunk (i & 7) ^ ((row >> 1) & 7)
    const int ar = lane >> 3, as = lane & 7;
    // B: an instruction fills 16 rows x 64 B; lane i -> row i >> 2, slot i & 3 holds chunk (i & 3) ^ ((row >> 2) & 3)
    const int br = lane >> 2, bs = lane & 3;
    auto issue = [&](int stage, int kt) {
        const unsigned sb = lds0 + (unsigned)(stage * STAGE);
#pragma unroll
        for (int u = 0; u < 4; ++u) {                 // wave w: A chunks 4w .. 4w+3 (8 rows each)
            const int ch = wave * 4 + u, row = ch * 8 + ar;
            const unsigned voff = (unsigned)(row * K * 4) + (unsigned)((as ^ ((row >> 1) & 7)) << 4);
            p_dma16(rsA, sb + (unsigned)(ch * 1024), voff, (unsigned)(kt * 128));
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {                 // wave w: B pieces 3w .. 3w+2 of the 24 (plane p = piece / 8, 16 rows each)
            const int pc = wave * 3 + u, pl = pc >> 3, row = (pc & 7) * 16 + br;
            const u32x4_t rsB = p_rsrc(Bs + pl * plane + (size_t)n0 * K, (size_t)PBN * K * 2);
            const unsigned voff = (unsigned)(row * K * 2) + (unsigned)((bs ^ ((row >> 2) & 3)) << 4);
            p_dma16(rsB, sb + (unsigned)(A_BYTES + pc * 1024), voff, (unsigned)(kt * 64));
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / 32;
    issue(0, 0);
    const int r32 = lane & 31, half = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
        const char* sb = psm + (kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + r32, j2 = 2 * (2 * s + half), sw = (row >> 1) & 7;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(sb + row * 128 + ((j2 ^ sw) << 4));
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(sb + row * 128 + (((j2 + 1) ^ sw) << 4));
                split8(x0, x1, a[0][i], a[1][i], a[2][i]);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wn * 64 + j * 32 + r32, c = 2 * s + half;
                    b[p][j] = *reinterpret_cast<const bf16x8*>(sb + A_BYTES + p * B_PLANE + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
                }
            // term-major: consecutive MFMAs write DIFFERENT accumulators (small terms first)
            constexpr int NT = TERMS;
            constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 6 - NT; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta[t]][i], b[tb[t]][j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(size_t)row * N + n0 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
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

static void errors(const std::vector<float>& hA, const std::vector<float>& hB, const float* dC, int M, int N, int K,
                   double& emax, double& erms) {
    const int NS = 512;
    double se = 0.0, sr = 0.0;
    emax = 0.0;
    std::vector<double> ref(NS), got(NS);
    unsigned s = 12345u;
    for (int t = 0; t < NS; ++t) {
        s = s * 1664525u + 1013904223u; const int m = (int)((s >> 8) % (unsigned)M);
        s = s * 1664525u + 1013904223u; const int n = (int)((s >> 8) % (unsigned)N);
        double a = 0.0;
        for (int k = 0; k < K; ++k) a += (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
        float g;
        CK(hipMemcpy(&g, dC + (size_t)m * N + n, 4, hipMemcpyDeviceToHost));
        ref[t] = a; got[t] = g;
        sr += a * a;
    }
    const double scale = sqrt(sr / NS);
    for (int t = 0; t < NS; ++t) {
        const double e = fabs(got[t] - ref[t]) / scale;
        emax = e > emax ? e : emax;
        se += e * e;
    }
    erms = sqrt(se / NS);
}

int main() {
    const int NSH = 3;
    const int shapes[NSH][3] = {{131072, 256, 2304}, {131072, 256, 1024}, {131072, 1024, 256}};
    const char* names[NSH] = {"3x3 256->256 at 16x16 x 512 (K = 2304)", "1x1 1024->256 at 16x16 x 512 (K = 1024)",
                              "1x1 256->1024 at 16x16 x 512 (K = 256)"};
    const size_t lds_p256 = (size_t)2 * (256 * 128 + 3 * 128 * 64);
    CK(hipFuncSetAttribute((const void*)gemm_split_p256<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p256));
    CK(hipFuncSetAttribute((const void*)gemm_split_p256<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p256));
    const size_t lds_split = (size_t)6 * 128 * 40 * 2;
    CK(hipFuncSetAttribute((const void*)gemm_split<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split));
    CK(hipFuncSetAttribute((const void*)gemm_split<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split));
    CK(hipFuncSetAttribute((const void*)(gemm_split<6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split));
    for (int si = 0; si < NSH; ++si) {
        const int M = shapes[si][0], N = shapes[si][1], K = shapes[si][2];
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 777u + si;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
        // activations after a ReLU-like non-negative-heavy law, filters at the He scale
        for (auto& v : hA) { const float r = rnd(); v = r > -0.3f ? r + 0.3f : 0.f; }
        for (auto& v : hB) v = rnd() * sqrtf(6.0f / K);
        float *dA, *dB, *dC;
        bf16_t *dBs, *dAs;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMalloc(&dBs, hB.size() * 2 * 3));
        CK(hipMalloc(&dAs, hA.size() * 2 * 3));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(split_rows_kernel, dim3(1024), dim3(256), 0, 0, dB, dBs, hB.size());
        hipLaunchKernelGGL(split_rows_kernel, dim3(8192), dim3(256), 0, 0, dA, dAs, hA.size());
        const dim3 grid((M / BM) * (N / BN));
        const double flop = 2.0 * M * N * K;
        printf("%s\n", names[si]);
        double emax, erms;
        double ms = time_ms([&]() { hipLaunchKernelGGL(gemm_native, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K); });
        errors(hA, hB, dC, M, N, K, emax, erms);
        printf("  native v_mfma_f32_32x32x2_f32      %7.3f ms  %6.1f TF/s   error vs fp64: max %.2e rms %.2e\n", ms, flop / ms / 1e9, emax, erms);
        ms = time_ms([&]() { hipLaunchKernelGGL(gemm_split<6>, grid, dim3(256), lds_split, 0, dA, dAs, dBs, dC, M, N, K); });
        errors(hA, hB, dC, M, N, K, emax, erms);
        printf("  split bf16, 6 terms (24 bits)      %7.3f ms  %6.1f TF/s   error vs fp64: max %.2e rms %.2e\n", ms, flop / ms / 1e9, emax, erms);
        ms = time_ms([&]() { hipLaunchKernelGGL(gemm_split<3>, grid, dim3(256), lds_split, 0, dA, dAs, dBs, dC, M, N, K); });
        errors(hA, hB, dC, M, N, K, emax, erms);
        printf("  split bf16, 3 terms (16 bits)      %7.3f ms  %6.1f TF/s   error vs fp64: max %.2e rms %.2e\n", ms, flop / ms / 1e9, emax, erms);
        ms = time_ms([&]() { hipLaunchKernelGGL((gemm_split<6, true>), grid, dim3(256), lds_split, 0, dA, dAs, dBs, dC, M, N, K); });
        errors(hA, hB, dC, M, N, K, emax, erms);
        printf("  split bf16, 6 terms, A pre-split   %7.3f ms  %6.1f TF/s   error vs fp64: max %.2e rms %.2e\n", ms, flop / ms / 1e9, emax, erms);
        {
            const dim3 gridp((M / 256) * (N / 128));
            ms = time_ms([&]() { hipLaunchKernelGGL(gemm_split_p256<6>, gridp, dim3(512), lds_p256, 0, dA, dBs, dC, M, N, K); });
            errors(hA, hB, dC, M, N, K, emax, erms);
            printf("  split 6 terms, 256x128 tile, LDS-DMA, split at fragment read  %7.3f ms  %6.1f TF/s   error vs fp64: max %.2e rms %.2e\n",
                   ms, flop / ms / 1e9, emax, erms);
            ms = time_ms([&]() { hipLaunchKernelGGL(gemm_split_p256<3>, gridp, dim3(512), lds_p256, 0, dA, dBs, dC, M, N, K); });
            errors(hA, hB, dC, M, N, K, emax, erms);
            printf("  split 3 terms, 256x128 tile, LDS-DMA, split at fragment read  %7.3f ms  %6.1f TF/s   error vs fp64: max %.2e rms %.2e\n",
                   ms, flop / ms / 1e9, emax, erms);
        }
        ms = time_ms([&]() { hipLaunchKernelGGL(split_rows_kernel, dim3(1024), dim3(256), 0, 0, dB, dBs, hB.size()); });
        printf("  (filter split [3][N][K], once per step: %.3f ms)\n", ms);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dBs)); CK(hipFree(dAs));
    }
    return 0;
}
