// Implicit-GEMM convolutions on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Everything the ResNet-50 path needs from "conv" is one of two GEMM shapes over NHWC data:
//
//   NT  out[m][o] = sum_{tap,c} In[pix(m,tap)][c] * W[o][tap][c]      (+ add[m][o])
//       forward conv (resnet_cls.py:23-31, :140) and both data-gradient forms.  Both operands
//       are K-contiguous, so the k index is permuted freely: each lane fetches 4 consecutive k
//       with one ds_read_b128 and feeds them to 4 MFMAs.
//   TN  dW[o][tap][c] = sum_m dY[m][o] * In[pix(m,tap)][c]
//       weight gradient; the reduction index m is the slow (row) dimension of both operands, so
//       tiles are stored as loaded ([m][channel]) and fragments are column slices (ds_read_b32,
//       consecutive lanes -> consecutive banks).  Split over m with a deterministic second pass.
//
// fp32 MFMA runs at the vector rate (64 FLOP/clk/SIMD), so the kernels are compute bound by a
// wide margin (LDS needs ~16 B/clk/CU of 256); the design goal is simply to keep one MFMA chain
// per SIMD busy: 128x128x32 block tile, 4 waves in 2x2, 2x2 MFMA tiles per wave (4 independent
// accumulators), register-staged double buffering (global loads of tile k+1 fly under the 64
// MFMAs of tile k), 73 KiB LDS -> 2 blocks/CU.
#include "io_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// bijective XCD-aware remap of a 1-D grid: blocks that run on one XCD (b % 8) get a contiguous
// range of tile ids, so neighbouring tiles (same A rows, different output-channel tile) share
// that XCD's L2.  Only affects speed.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------
template <int BN, bool STEM>
__global__ __launch_bounds__(kThreads) void conv_nt_kernel(IoConvGeom g, const float* __restrict__ in,
                                                          const float* __restrict__ wgt,
                                                          float* __restrict__ out,
                                                          const float* __restrict__ add, int ntn) {
    constexpr int BM = 128, BK = 32, LDT = BK + 4;
    constexpr int TI = 2, TJ = BN / 64;     // MFMA tiles per wave (wave tile 64 x BN/2)
    constexpr int BR = BN / 32;             // weight rows loaded per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                       // [2][BM][LDT]
    float* sB = smem + 2 * BM * LDT;        // [2][BN][LDT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int HoWo = g.Ho * g.Wo;
    const int M = g.N * HoWo;
    const int lr = tid >> 3, kq = tid & 7;

    const float* rowbase[4];
    int hi0[4], wi0[4];
    bool rvalid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + lr + 32 * j;
        rvalid[j] = m < M;
        const int mm = rvalid[j] ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
        hi0[j] = ho * g.is;
        wi0[j] = wo * g.is;
        rowbase[j] = in + (size_t)n * g.Hi * g.Wi * g.Ci;
    }
    const float* wrow[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) wrow[j] = wgt + (size_t)(n0 + lr + 32 * j) * g.wT * g.Ci;

    const int nkc = STEM ? 1 : g.Ci / BK;
    const int nk = STEM ? (g.wT + 3) / 4 : g.Th * g.Tw * nkc;

    f32x4 ra[4], rb[BR];
    int th = 0, tw = 0, cc = 0;   // running tap / channel-chunk counters of the tile being LOADED

    auto load_tile = [&](int kt) {
        int dh, dw, widx, coff;
        bool tapok = true;
        if (STEM) {
            const int tap = kt * 4 + (kq >> 1);
            tapok = tap < g.wT;
            const int r = tap / g.S, s = tap - r * g.S;
            dh = g.dh0 + g.dhs * r;
            dw = g.dw0 + g.dws * s;
            widx = tap;
            coff = (kq & 1) * 4;
        } else {
            dh = g.dh0 + g.dhs * th;
            dw = g.dw0 + g.dws * tw;
            widx = (g.r0 + g.rs * th) * g.S + (g.s0 + g.ss * tw);
            coff = cc * BK + kq * 4;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hi = hi0[j] + dh, wi = wi0[j] + dw;
            const bool ok = tapok && rvalid[j] && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[j] = ok ? ld4(rowbase[j] + ((size_t)(hi * g.Wi + wi)) * g.Ci + coff) : z;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rb[j] = tapok ? ld4(wrow[j] + (size_t)widx * g.Ci + coff) : z;
        }
        if (!STEM) {
            if (++cc == nkc) {
                cc = 0;
                if (++tw == g.Tw) { tw = 0; ++th; }
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* a = sA + buf * BM * LDT + lr * LDT + kq * 4;
        float* b = sB + buf * BN * LDT + lr * LDT + kq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) st4(a + 32 * j * LDT, ra[j]);
#pragma unroll
        for (int j = 0; j < BR; ++j) st4(b + 32 * j * LDT, rb[j]);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* a_lds = sA + buf * BM * LDT + (wm * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;
        const float* b_lds = sB + buf * BN * LDT + (wn * (BN / 2) + (lane & 31)) * LDT + (lane >> 5) * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = ld4(a_lds + i * 32 * LDT + kk * 8);
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = ld4(b_lds + j * 32 * LDT + kk * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D layout col = lane&31 (output channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool dense = (g.os == 1) && (g.Ho == g.outH) && (g.Wo == g.outW);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= M) continue;
            size_t pix;
            if (dense) {
                pix = (size_t)m;
            } else {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
                pix = ((size_t)n * g.outH + (ho * g.os + g.ooh)) * g.outW + (wo * g.os + g.oow);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const size_t idx = pix * g.Co + n0 + wn * (BN / 2) + j * 32 + (lane & 31);
                float v = acc[i][j][r];
                if (add) v += add[idx];
                out[idx] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN (weight gradient) kernel
// ------------------------------------------------------------------------------------------
template <int BMO, int BNC, bool STEM>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(IoConvGeom g, const float* __restrict__ in,
                                                             const float* __restrict__ dy,
                                                             float* __restrict__ dst, int ntile_c, int kps) {
    constexpr int BKM = 32;
    constexpr int TI = BMO / 64, TJ = BNC / 64;
    constexpr int QA = BMO / 4, QB = BNC / 4;          // float4 per tile row
    constexpr int RA = (BKM * QA) / kThreads;          // rows per thread (A)
    constexpr int RB = (BKM * QB) / kThreads;
    constexpr int SA = kThreads / QA, SB = kThreads / QB;   // row step between a thread's rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                    // [2][BKM][BMO]
    float* sB = smem + 2 * BKM * BMO;    // [2][BKM][BNC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = g.Ho * g.Wo;
    const int M = g.N * HoWo;
    const int T = g.Th * g.Tw;

    // tile decode: blockIdx.x -> (o tile, tap, channel tile)
    const int per_o = STEM ? ntile_c : T * ntile_c;
    const int ot = blockIdx.x / per_o;
    const int rem0 = blockIdx.x - ot * per_o;
    const int o0 = ot * BMO;
    int tap = 0, c0;
    if (STEM) {
        c0 = rem0 * BNC;                 // column in the flattened (tap, 8 channels) axis
    } else {
        tap = rem0 / ntile_c;
        c0 = (rem0 - tap * ntile_c) * BNC;
    }
    const int qa = tid % QA, ra0 = tid / QA;
    const int qb = tid % QB, rb0 = tid / QB;

    int dh, dw, widx, coff;
    bool tapok = true;
    if (STEM) {
        const int col = c0 + qb * 4;
        const int tp = col >> 3;
        tapok = tp < g.wT;
        const int r = tp / g.S, s = tp - r * g.S;
        dh = g.dh0 + g.dhs * r;
        dw = g.dw0 + g.dws * s;
        widx = tp;
        coff = col & 7;
    } else {
        const int th = tap / g.Tw, tw = tap - th * g.Tw;
        dh = g.dh0 + g.dhs * th;
        dw = g.dw0 + g.dws * tw;
        widx = (g.r0 + g.rs * th) * g.S + (g.s0 + g.ss * tw);
        coff = c0 + qb * 4;
    }

    const int nkt = (M + BKM - 1) / BKM;
    const int kt0 = blockIdx.y * kps;
    const int kt1 = min(kt0 + kps, nkt);

    f32x4 ra[RA], rb[RB];
    auto load_tile = [&](int kt) {
        const int mb = kt * BKM;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int m = mb + ra0 + SA * j;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[j] = (m < M) ? ld4(dy + (size_t)m * g.Co + o0 + qa * 4) : z;
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int m = mb + rb0 + SB * j;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bool ok = tapok && m < M;
            const int mm = ok ? m : 0;
            const int n = mm / HoWo, rem = mm - n * HoWo;
            const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
            const int hi = ho * g.is + dh, wi = wo * g.is + dw;
            ok = ok && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
            rb[j] = ok ? ld4(in + (((size_t)n * g.Hi + hi) * g.Wi + wi) * g.Ci + coff) : z;
        }
    };
    auto store_tile = [&](int buf) {
        float* a = sA + buf * BKM * BMO + ra0 * BMO + qa * 4;
        float* b = sB + buf * BKM * BNC + rb0 * BNC + qb * 4;
#pragma unroll
        for (int j = 0; j < RA; ++j) st4(a + SA * j * BMO, ra[j]);
#pragma unroll
        for (int j = 0; j < RB; ++j) st4(b + SB * j * BNC, rb[j]);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tile(kt + 1);
        const float* a_lds = sA + buf * BKM * BMO + (lane >> 5) * BMO + wm * (BMO / 2) + (lane & 31);
        const float* b_lds = sB + buf * BKM * BNC + (lane >> 5) * BNC + wn * (BNC / 2) + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BKM / 2; ++kk) {
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = a_lds[kk * 2 * BMO + i * 32];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = b_lds[kk * 2 * BNC + j * 32];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt1) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: rows = output channel o, cols = input channel (or flattened stem column)
    const size_t wrow = (size_t)g.wT * g.Ci;
    float* base = dst + (size_t)blockIdx.y * g.Co * wrow;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wm * (BMO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int cl = wn * (BNC / 2) + j * 32 + (lane & 31);
                if (STEM) {
                    const int col = c0 + cl;
                    if (col < (int)wrow) base[(size_t)o * wrow + col] = acc[i][j][r];
                } else {
                    base[(size_t)o * wrow + (size_t)widx * g.Ci + c0 + cl] = acc[i][j][r];
                }
            }
        }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dst, size_t n4,
                                     int splits) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 s = reinterpret_cast<const f32x4*>(partial)[i];
        for (int z = 1; z < splits; ++z) s += reinterpret_cast<const f32x4*>(partial)[(size_t)z * n4 + i];
        reinterpret_cast<f32x4*>(dst)[i] = s;
    }
}

struct WgradPlan {
    int bmo, bnc, ntile_c, tiles, splits, kps;
};

WgradPlan plan_wgrad(const IoConvGeom& g, int stem) {
    WgradPlan p;
    const long M = (long)g.N * g.Ho * g.Wo;
    p.bmo = (g.Co % 128 == 0) ? 128 : 64;
    if (stem) {
        p.bnc = 64;
        p.ntile_c = io_cdiv((long)g.wT * g.Ci, 64);
        p.tiles = (g.Co / p.bmo) * p.ntile_c;
    } else {
        p.bnc = (g.Ci % 128 == 0) ? 128 : 64;
        p.ntile_c = g.Ci / p.bnc;
        p.tiles = (g.Co / p.bmo) * g.Th * g.Tw * p.ntile_c;
    }
    const int nkt = io_cdiv(M, 32);
    int want = io_cdiv(1024, p.tiles);          // ~4 resident block-waves over 256 CUs
    int maxs = nkt / 8 > 0 ? nkt / 8 : 1;       // at least 8 k-tiles (256 rows) per split
    p.splits = want < maxs ? want : maxs;
    if (p.splits < 1) p.splits = 1;
    p.kps = io_cdiv(nkt, p.splits);
    p.splits = io_cdiv(nkt, p.kps);
    return p;
}

}  // namespace

size_t io_conv_wgrad_partial_bytes(const IoConvGeom& g, int stem) {
    WgradPlan p = plan_wgrad(g, stem);
    if (p.splits == 1) return 0;
    return (size_t)p.splits * g.Co * g.wT * g.Ci * sizeof(float);
}

int io_launch_conv_nt(const IoConvGeom& g, const float* in, const float* wgt, float* out, const float* add,
                      int stem, hipStream_t st) {
    IO_REQUIRE(g.Co % 64 == 0, IO_ERR_SHAPE, "conv_nt: Co=%d must be a multiple of 64", g.Co);
    if (stem)
        IO_REQUIRE(g.Ci == 8, IO_ERR_SHAPE, "conv_nt(stem): Ci=%d must be 8 (5 channels padded)", g.Ci);
    else
        IO_REQUIRE(g.Ci % 32 == 0, IO_ERR_SHAPE, "conv_nt: Ci=%d must be a multiple of 32", g.Ci);
    const long M = (long)g.N * g.Ho * g.Wo;
    IO_REQUIRE(M > 0 && M < (1L << 31), IO_ERR_SHAPE, "conv_nt: bad M=%ld", M);
    const int bn = (g.Co % 128 == 0) ? 128 : 64;
    const int ntn = g.Co / bn;
    const long tiles = (long)io_cdiv(M, 128) * ntn;
    IO_REQUIRE(tiles < (1L << 31), IO_ERR_SHAPE, "conv_nt: grid too large");
    const size_t lds = (size_t)2 * (128 + bn) * 36 * sizeof(float);
    dim3 grid((unsigned)tiles), block(kThreads);
    // algorithmic work: real taps x real channels (the stem's 3 padding channels do not count)
    const double kred = stem ? (double)g.wT * 5.0 : (double)g.Th * g.Tw * g.Ci;
    IoProfScope prof(stem ? IO_PROF_CONV_STEM : (bn == 128 ? IO_PROF_CONV_NT128 : IO_PROF_CONV_NT64),
                     2.0 * (double)M * g.Co * kred,
                     4.0 * ((double)M * g.Co + (double)g.N * g.Hi * g.Wi * g.Ci + (double)g.Co * kred), st);
#define IO_LAUNCH_NT(BN_, STEM_)                                                                             \
    do {                                                                                                     \
        static bool attr_done = false;                                                                       \
        if (!attr_done) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)conv_nt_kernel<BN_, STEM_>,                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + BN_) * 36 * 4);       \
            attr_done = true;                                                                                \
        }                                                                                                    \
        hipLaunchKernelGGL((conv_nt_kernel<BN_, STEM_>), grid, block, lds, st, g, in, wgt, out, add, ntn);   \
    } while (0)
    if (stem) {
        IO_REQUIRE(bn == 64, IO_ERR_SHAPE, "conv_nt(stem): Co must be 64");
        IO_LAUNCH_NT(64, true);
    } else if (bn == 128) {
        IO_LAUNCH_NT(128, false);
    } else {
        IO_LAUNCH_NT(64, false);
    }
#undef IO_LAUNCH_NT
    return io_check_launch("conv_nt");
}

int io_launch_conv_wgrad(const IoConvGeom& g, const float* in, const float* dy, float* dw, float* partial,
                         size_t partial_bytes, int stem, hipStream_t st) {
    IO_REQUIRE(g.Co % 64 == 0, IO_ERR_SHAPE, "conv_wgrad: Co=%d must be a multiple of 64", g.Co);
    if (stem)
        IO_REQUIRE(g.Ci == 8 && g.Co == 64, IO_ERR_SHAPE, "conv_wgrad(stem): need Ci=8, Co=64");
    else
        IO_REQUIRE(g.Ci % 64 == 0, IO_ERR_SHAPE, "conv_wgrad: Ci=%d must be a multiple of 64", g.Ci);
    IO_REQUIRE(g.os == 1 && g.Ho == g.outH && g.Wo == g.outW, IO_ERR_SHAPE, "conv_wgrad: dY must be dense");
    WgradPlan p = plan_wgrad(g, stem);
    const size_t need = io_conv_wgrad_partial_bytes(g, stem);
    IO_REQUIRE(partial_bytes >= need && (need == 0 || partial), IO_ERR_WORKSPACE,
               "conv_wgrad: workspace %zu < %zu bytes", partial_bytes, need);
    float* dst = p.splits == 1 ? dw : partial;
    dim3 grid((unsigned)p.tiles, (unsigned)p.splits), block(kThreads);
    const double Md = (double)g.N * g.Ho * g.Wo;
    const double kred = stem ? (double)g.wT * 5.0 : (double)g.Th * g.Tw * g.Ci;
    IoProfScope prof(stem ? IO_PROF_WGRAD_STEM : IO_PROF_WGRAD, 2.0 * Md * g.Co * kred,
                     4.0 * (Md * g.Co + (double)g.N * g.Hi * g.Wi * g.Ci + (double)g.Co * kred), st);
#define IO_LAUNCH_WG(BMO_, BNC_, STEM_)                                                                   \
    do {                                                                                                  \
        const size_t lds = (size_t)2 * 32 * (BMO_ + BNC_) * sizeof(float);                                \
        hipLaunchKernelGGL((conv_wgrad_kernel<BMO_, BNC_, STEM_>), grid, block, lds, st, g, in, dy, dst,  \
                           p.ntile_c, p.kps);                                                             \
    } while (0)
    if (stem)
        IO_LAUNCH_WG(64, 64, true);
    else if (p.bmo == 128 && p.bnc == 128)
        IO_LAUNCH_WG(128, 128, false);
    else if (p.bmo == 128)
        IO_LAUNCH_WG(128, 64, false);
    else if (p.bnc == 128)
        IO_LAUNCH_WG(64, 128, false);
    else
        IO_LAUNCH_WG(64, 64, false);
#undef IO_LAUNCH_WG
    int rc = io_check_launch("conv_wgrad");
    if (rc) return rc;
    if (p.splits > 1) {
        const size_t n4 = (size_t)g.Co * g.wT * g.Ci / 4;
        int blocks = io_cdiv((long)n4, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, partial, dw, n4, p.splits);
        rc = io_check_launch("splitk_reduce");
    }
    return rc;
}
