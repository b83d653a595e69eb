// bf16 implicit-GEMM forward convolutions / data gradients on 256-row tiles, persistent blocks, LDS-DMA staging.
//
//   out[m][o] = sum_{tap,c} In[pix(m,tap)][c] * W[o][tap][c]   (+ add[m][o]) -- the NT shape of conv_igemm.hip
//   (models/backbone/resnet_cls.py:23-31: conv3x3 / conv1x1, and their data gradients as gather convolutions)
//
// Why a second kernel next to conv_nt_kernel: in bf16 the 128 x 128 / 4-wave / register-staged structure is bound by LDS
// traffic (one 16-byte fragment read per MFMA, a ds_write pass per k-tile) on the long-K layers and by what one k-tile per
// block keeps in flight on the HBM-bound ones (profiles/r04_pmc_bf16.json: matrix pipe 19 % busy at 0.46 of the HBM peak).
// tools/bf16_dma_probe.hip measured the alternatives on the GEMM shapes of the step (profiles/r05_bf16_dma_probe.txt):
//   * 256 x 256 x 64 block tile, 8 waves of 128 x 64 (0.75 fragment reads per MFMA), both operands global -> LDS by
//     LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass) into an XOR-swizzled image, 2 stages;
//   * PERSISTENT blocks (one per CU): a block walks tiles b, b + grid, ... and treats their k-tiles as one stream, so the
//     first k-tile of the next tile is in flight under the last MFMAs and the epilogue of the current one;
//   -13 .. -31 % against the library's structure on every shape with N >= 256, -6 .. -11 % at N = 128 (256 x 128 tile).
// This file is that kernel with the convolution's addressing and the epilogues the executor needs:
//   GATHER  per-row pixel decode + tap validity for anything that is not a dense 1x1 GEMM (3x3, strided inputs);
//   EPI_STATS  per-(128-row tile, channel) mean / M2 of the output (training forward, bn.hip finalizes them);
//   EPI_BWE    the fused BatchNorm-backward epilogue of a data gradient (IoBwStats: mask recomputed from y or read, residual
//              gradient added, per-tile sums of dz and dz * xhat) -- on ROWS: a wave turns its accumulators through LDS 16
//              rows at a time, so y / add / mask arrive as 16-byte row pieces instead of 2-byte column loads;
//   EPI_PLAIN  optional add / mask / bias (+ ReLU).
// Operand transforms (XF / XB / XR) stay on conv_nt_kernel.  bf16 in, bf16 out, fp32 accumulate.
#include <stdlib.h>
#include <string.h>


#include "io_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFFFFu;

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
__device__ __forceinline__ int fdiv(int n, IoFastDiv f) {
    return f.shift < 0 ? n : (int)(__umulhi((unsigned)n, f.magic) >> f.shift);
}
// descriptor over p[base, total): 32-bit offsets are relative to base; num_records saturates at 2^32 - 1
__device__ __forceinline__ u32x4 dma_rsrc(const void* p, size_t base, size_t total) {
    const unsigned long long a = (unsigned long long)p + base;
    const size_t rest = total > base ? total - base : 0;
    const u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, rest > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)rest,
                     0x00020000u};
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const void* p, size_t base, size_t total) {
    const size_t rest = total > base ? total - base : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p)) + base, 0,
                                             rest > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)rest, 0x00020000);
}
// One LDS-DMA instruction: 16 bytes per lane from rs at voff (per lane, range-checked: out of range = zeros in LDS) + soff
// (wave-uniform, NOT range-checked) to LDS byte address lds_addr + 16 * lane.  Inline asm: hipcc would put s_waitcnt vmcnt(0)
// in front of every LDS read that may alias a pending builtin DMA (conv_igemm.hip: conv_wgrad_bf16_tr_kernel).
__device__ __forceinline__ void dma16(u32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// EPI_BWE: the ReLU mask of the BatchNorm behind the gradient is RECOMPUTED from y (mscale / mshift); EPI_BWE_READ: no such mask
// (add / mask tensors are read instead) -- two instantiations because the first keeps three more coefficient tables in
// registers and the second three operand loads per pass in flight
enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_BWE = 2, EPI_BWE_READ = 3 };
constexpr int kXopMaxCi = 2048;       // XOP: three coefficient rows of Ci floats wait in LDS behind the two stages (<= 24 KB)

struct P256Args {
    const bf16_t* in;
    const bf16_t* wgt;
    bf16_t* out;
    const bf16_t* add;      // optional, shaped like out
    const bf16_t* mask;     // optional, shaped like out: out = mask > 0 ? out : 0
    size_t in_bytes, out_bytes;
    unsigned w_bytes;
    float *st_mean, *st_m2; // EPI_STATS: [M / 128][Co]
    IoBwStats bw;           // EPI_BWE: y, mean, rstd, mscale, mshift, p1, p2, Mg.  EPI_PLAIN: bias, relu
    int ntn, ntiles;
    // XOP (operand transform in LDS, dense 1x1 only): the A operand is a function of `in` and a second tensor xy of the same
    // shape, evaluated on the k-tile AFTER its DMA has landed and written back in place; the blocks of the first
    // output-channel tile also write it out (xout; xbits: [value > 0] as one bit per element).  Tables [G][Ci].
    //   xmode 1: a * in + (b * xy + c)               BatchNorm-backward apply (IoBwStats::xb_a, conv_igemm.hip XB = 1)
    //   xmode 2: relu((in - b) * a + c + xy)         block output relu(bn3(y3) + identity)               (XB = 2)
    //   xmode 3: relu(a * in + (b * xy + c))         block output with a downsample branch                (XB = 3)
    const bf16_t* xy;
    const float *xa, *xb, *xc;
    bf16_t* xout;
    uint8_t* xbits;
    int xmode, xMg;
};

// Block = 512 threads = 8 waves as 2 (rows) x 4 (columns): a wave owns 128 rows (ONE BatchNorm statistics tile) x BN / 4
// columns = TM x TN = 4 x (BN / 128) MFMA tiles of 32 x 32.  LDS: 2 stages of (256 + BN) rows x 128 bytes; row r of an operand
// tile at r * 128, its 16-byte k-chunk c in slot c ^ ((r >> 1) & 7) (the 16 rows a ds_read_b128 lane group touches -- distinct
// mod 16 -- land in 16 distinct bank quads).  A DMA instruction fills 8 rows: lane i fetches what belongs in slot i & 7 of row
// i >> 3.  Wave w fetches A chunks 4w .. 4w+3 and B chunks (BN / 64) w .. of every k-tile.
// XOP: 0 none; 1 the affine forms (xmode 1, and 3 = the same followed by a ReLU); 2 bn_apply_kernel's form (xmode 2)
template <int BN, int EPI, bool GATHER, int XOP = 0>
__global__ __launch_bounds__(512, 2) void conv_p256_kernel(IoConvGeom g, P256Args a) {
    static_assert(!XOP || !GATHER, "the in-LDS operand transform is for dense 1x1 launches");
    constexpr int BM = 256, TM = 4, TN = BN / 128, NW = 8;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int CA = 4, CB = BN / 64;          // DMA chunks (8 rows) per wave and k-tile
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int HoWo = g.Ho * g.Wo;
    const int kcn = g.Ci / 64;                   // 64-channel chunks per tap
    const int nk = g.Th * g.Tw * kcn;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int r8 = lane >> 3, p8 = lane & 7;
    const unsigned rowB = (unsigned)(g.wT * g.Ci * 2);         // bytes per filter row
    // lane parts of the fetches: the in-chunk row and the k-chunk that belongs in this lane's slot (odd chunks: slot ^ 4)
    const unsigned slot_e = (unsigned)((p8 ^ (r8 >> 1)) << 4), slot_o = slot_e ^ 64u;
    const unsigned vb_e = (unsigned)r8 * rowB + slot_e, vb_o = (unsigned)r8 * rowB + slot_o;
    const unsigned va_lin_e = (unsigned)(r8 * g.Ci * 2) + slot_e, va_lin_o = (unsigned)(r8 * g.Ci * 2) + slot_o;

    // per-tile state of the A fetches
    u32x4 rsA, rsB = dma_rsrc(a.wgt, 0, a.w_bytes);
    unsigned rowv[CA];          // GATHER: byte offset of the lane's row (pixel (hi0, wi0), channel 0, relative to the tile's base)
    int rhw[CA];                // GATHER: hi0 | wi0 << 16 (kOob-marked rows beyond M: hi0 = 0x7fff)
    int cur_m0 = 0, cur_n0 = 0;
    auto setup_tile = [&](int tile) {
        const int mt = tile / a.ntn;
        cur_m0 = mt * BM;
        cur_n0 = (tile - mt * a.ntn) * BN;
        if constexpr (!GATHER) {
            rsA = dma_rsrc(a.in, (size_t)cur_m0 * (size_t)(g.Ci * 2), a.in_bytes);
        } else {
            const int n_lo = fdiv(cur_m0, g.fd_howo);
            rsA = dma_rsrc(a.in, (size_t)n_lo * (size_t)(g.Hi * g.Wi) * (size_t)(g.Ci * 2), a.in_bytes);
#pragma unroll
            for (int u = 0; u < CA; ++u) {
                const int m = cur_m0 + (wave * CA + u) * 8 + r8;          // (whole tiles: always < M)
                const int n = fdiv(m, g.fd_howo), rem = m - n * HoWo;
                const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
                const int hi0 = ho * g.is, wi0 = wo * g.is;
                rowv[u] = (unsigned)((((n - n_lo) * g.Hi + hi0) * g.Wi + wi0) * g.Ci) * 2u;
                rhw[u] = hi0 | (wi0 << 16);
            }
        }
    };
    // tap / channel-chunk counters of the k-tile being FETCHED
    int f_th = 0, f_tw = 0, f_cc = 0;
    auto issue = [&](int stage) {
        const unsigned sb = lds0 + (unsigned)(stage * STAGE);
        const int widx = (g.r0 + g.rs * f_th) * g.S + (g.s0 + g.ss * f_tw);
        const unsigned bo = (unsigned)((widx * g.Ci + f_cc * 64) * 2);
        if constexpr (!GATHER) {
            const unsigned ao = (unsigned)(f_cc * 128);
#pragma unroll
            for (int u = 0; u < CA; ++u) {
                const int ch = wave * CA + u;
                dma16(rsA, sb + (unsigned)(ch * 1024), (ch & 1) ? va_lin_o : va_lin_e, ao + (unsigned)(ch * 8 * g.Ci * 2));
            }
        } else {
            const int dh = g.dh0 + g.dhs * f_th, dw = g.dw0 + g.dws * f_tw;
            // (unsigned wrap-around: a negative tap shift on an offset that stays inside the descriptor whenever it is used)
            const unsigned ao = (unsigned)(((dh * g.Wi + dw) * g.Ci + f_cc * 64) * 2);
#pragma unroll
            for (int u = 0; u < CA; ++u) {
                const int ch = wave * CA + u;
                const int hi = (rhw[u] & 0xffff) + dh, wi = (rhw[u] >> 16) + dw;
                const bool ok = (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                // (the tap shift rides in the vector offset: a scalar offset takes no part in the range check and must not wrap)
                dma16(rsA, sb + (unsigned)(ch * 1024), ok ? rowv[u] + ao + ((ch & 1) ? slot_o : slot_e) : kOob, 0u);
            }
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int ch = wave * CB + u;
            dma16(rsB, sb + (unsigned)(BM * 128 + ch * 1024), (ch & 1) ? vb_o : vb_e,
                  bo + (unsigned)(cur_n0 + ch * 8) * rowB);
        }
        // advance the fetch counters
        if (++f_cc == kcn) {
            f_cc = 0;
            if (++f_tw == g.Tw) {
                f_tw = 0;
                ++f_th;
            }
        }
    };

    // fragment reads: lane l reads row (l & 31) of a 32-row tile, k-chunk 2 kk + (l >> 5) -> slot that ^ ((l >> 1) & 7)
    const int swz = (lane >> 1) & 7;
    unsigned ko[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ko[kk] = (unsigned)(((2 * kk + (lane >> 5)) ^ swz) << 4);
    const unsigned a_row = (unsigned)((wm * 128 + (lane & 31)) * 128);
    const unsigned b_row = (unsigned)(BM * 128 + (wn * TN * 32 + (lane & 31)) * 128);

    // XOP: thread t owns the 16-byte chunk t & 7 (8 channels) of rows (t >> 3) + 64 u, u = 0..3, of every A k-tile -- the
    // pattern of the DMA (8 lanes = one 128-byte row piece: coalesced for xy and the side output, conflict-free in LDS).
    // The second operand travels one k-tile ahead in registers, like the DMA: requested (xyn) right behind the fetches of
    // k-tile q + 1, i.e. a whole k-tile before its sweep.  (Requested behind the sweep of k-tile q instead -- only that tile's
    // MFMAs to land in -- the launches took the same time, 0.238 ms on layer 3's 1024 -> 256: they are not paced by this load;
    // kept because it frees the 24 coefficient registers the first form held across the MFMAs.)  The three coefficient rows of
    // the tile's BatchNorm group wait in LDS behind the two stages ([3][Ci] floats, loaded once per group).
    u32x4 xyv[XOP ? 4 : 1], xyn[XOP ? 4 : 1];
    float* const xtab = reinterpret_cast<float*>(smem + 2 * STAGE);
    int xgrp = -1;
    const unsigned x_lds = (unsigned)(((tid >> 3) * 128) + (((tid & 7) ^ ((tid >> 4) & 7)) << 4));   // (+ 8192 u: same swizzle bits)
    const unsigned x_goff = (unsigned)((tid >> 3) * g.Ci * 2 + (tid & 7) * 16);                     // (+ 64 u rows, + 128 kt)
    auto xop_fetch = [&](int m0f, int ktf) {
        if constexpr (XOP != 0) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_at(a.xy, (size_t)m0f * (size_t)(g.Ci * 2), a.in_bytes);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                xyn[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, x_goff + (unsigned)(u * 64 * g.Ci * 2) + (unsigned)(ktf * 128), 0, 0);
        }
    };

    f32x16 acc[TM][TN];
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    int kt = 0, q = 0;
    if (tile < a.ntiles) {
        setup_tile(tile);
        issue(0);
        xop_fetch(cur_m0, 0);
    }
    while (tile < a.ntiles) {
        if (kt == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        const int m0 = cur_m0, n0 = cur_n0;      // of the tile being multiplied (setup_tile below moves on to the next one)
        // this wave's share of k-tile q has landed ...  (XOP: this also waits for the eight side-output stores of the previous
        // sweep; a counted wait that leaves them in flight -- s_waitcnt vmcnt(8), the counter retires in issue order -- was
        // measured: 47.13 / 47.27 vs 47.12 / 47.39 ms per step, no difference)
        dma_wait_all();
        __syncthreads();                         // ... everybody's has, and every wave is done reading the other stage
        const bool last = kt + 1 == nk;
        const int ntile = last ? tile + (int)gridDim.x : tile;
        if (last) {
            f_th = f_tw = f_cc = 0;
            if (ntile < a.ntiles) setup_tile(ntile);
        }
        if constexpr (XOP != 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) xyv[u] = xyn[u];     // (landed: requested a k-tile ago, waited for above)
        }
        if (ntile < a.ntiles) {
            issue((q + 1) & 1);
            xop_fetch(last ? cur_m0 : m0, last ? 0 : kt + 1);
        }
        if constexpr (XOP != 0) {
            if (kt == 0 && m0 / a.xMg != xgrp) {          // (block-uniform; at most once per group and block: tiles ascend)
                xgrp = m0 / a.xMg;
                for (int i = tid; i < g.Ci; i += 512) {
                    xtab[i] = a.xa[xgrp * g.Ci + i];
                    xtab[g.Ci + i] = a.xb[xgrp * g.Ci + i];
                    xtab[2 * g.Ci + i] = a.xc[xgrp * g.Ci + i];
                }
                __syncthreads();
            }
            // ---- the operand transform of k-tile kt of tile (m0, n0), in place in stage q & 1
            f32x4 xca[2], xcb[2], xcc[2];
            {
                const float* tb = xtab + kt * 64 + (tid & 7) * 8;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    xca[h] = *reinterpret_cast<const f32x4*>(tb + 4 * h);
                    xcb[h] = *reinterpret_cast<const f32x4*>(tb + g.Ci + 4 * h);
                    xcc[h] = *reinterpret_cast<const f32x4*>(tb + 2 * g.Ci + 4 * h);
                }
            }
            char* sa = smem + (q & 1) * STAGE + x_lds;
            const bool side = n0 == 0;
            const size_t sbase = (size_t)m0 * (size_t)(g.Ci * 2);
            const __amdgpu_buffer_rsrc_t rs_side = rsrc_at(a.xout ? (const void*)a.xout : (const void*)a.in, sbase,
                                                           (side && a.xout) ? a.in_bytes : sbase);
            const __amdgpu_buffer_rsrc_t rs_xb = rsrc_at(a.xbits ? (const void*)a.xbits : (const void*)a.in, sbase / 16,
                                                         (side && a.xbits) ? a.in_bytes / 16 : sbase / 16);
            const bool xrelu = a.xmode >= 2;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32x4 av = *reinterpret_cast<const u32x4*>(sa + u * 8192);
                const u32x4 yv = xyv[u];
                u32x4 o;
                unsigned bits = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int h = d >> 1, e0 = (d & 1) * 2;
                    const float lo = bf_lo(av[d]), hi = bf_hi(av[d]), yl = bf_lo(yv[d]), yh = bf_hi(yv[d]);
                    float vl, vh;
                    if constexpr (XOP == 2) {   // bn_apply_kernel's expression: the tensor a separate pass would have written
                        vl = __builtin_fmaf(lo - xcb[h][e0], xca[h][e0], xcc[h][e0]) + yl;
                        vh = __builtin_fmaf(hi - xcb[h][e0 + 1], xca[h][e0 + 1], xcc[h][e0 + 1]) + yh;
                    } else {
                        vl = __builtin_fmaf(lo, xca[h][e0], __builtin_fmaf(yl, xcb[h][e0], xcc[h][e0]));
                        vh = __builtin_fmaf(hi, xca[h][e0 + 1], __builtin_fmaf(yh, xcb[h][e0 + 1], xcc[h][e0 + 1]));
                    }
                    vl = (xrelu && !(vl > 0.f)) ? 0.f : vl;
                    vh = (xrelu && !(vh > 0.f)) ? 0.f : vh;
                    bits |= (vl > 0.f ? 1u : 0u) << (2 * d) | (vh > 0.f ? 1u : 0u) << (2 * d + 1);
                    o[d] = io_f2bf2(vl, vh);
                }
                *reinterpret_cast<u32x4*>(sa + u * 8192) = o;
                // (the k-tile's offset rides in the VECTOR offset, scalar offset 0: with an SGPR there hipcc schedules a VALU
                // write of the store's first data register right behind the 16-byte store -- its hazard table has no wait state
                // for that form -- and on gfx950 the store then wrote that VALU result for a quarter of the lanes: found by
                // tests/test_gpu_xop.py, ISA inspected)
                const unsigned go = x_goff + (unsigned)(u * 64 * g.Ci * 2) + (unsigned)(kt * 128);
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_side, go, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, rs_xb, go >> 4, 0, 0);
            }
            __syncthreads();                     // every wave's share is transformed before anybody multiplies
        }
        const char* sb = smem + (q & 1) * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sb + a_row + i * 4096 + ko[kk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(sb + b_row + j * 4096 + ko[kk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (last) {
            // ---- epilogue of tile (m0, n0): D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) per 32 x 32 tile
            const int mt128 = (m0 >> 7) + wm;                    // the wave's 128 rows are one statistics tile
            if constexpr (EPI == EPI_STATS) {
                // per column: mean of the 128 rows, then the sum of squared deviations from it -- two passes over the
                // accumulators, the two lane halves combined with one shuffle (as conv_nt_kernel does per 64 rows)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
                    s += __shfl_xor(s, 32, 64);
                    const float mean = s * (1.0f / 128.0f);
                    float d2 = 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) d2 += (acc[i][j][r] - mean) * (acc[i][j][r] - mean);
                    d2 += __shfl_xor(d2, 32, 64);
                    if (lane < 32) {
                        const size_t o = (size_t)mt128 * g.Co + n0 + wn * TN * 32 + j * 32 + lane;
                        a.st_mean[o] = mean;
                        a.st_m2[o] = d2;
                    }
                }
            }
            __syncthreads();                     // every wave is done with stage q & 1: it carries the output now
            // A wave turns its tile through its own LDS slice, 16 rows x WC columns at a time: fp32 in column order, out as 8
            // consecutive channels of a row per lane (16-byte pieces of whole rows).
            constexpr int WC = TN * 32, EPP = WC + 4, ER = 16;
            constexpr int LPR = WC / 8, RPI = 64 / LPR, NI = ER / RPI;      // lanes per row, rows per pass of the wave
            static_assert(NW * ER * EPP * 4 <= STAGE, "epilogue slice must fit one stage");
            float* ep = reinterpret_cast<float*>(smem + (q & 1) * STAGE) + wave * (ER * EPP);
            const int ecol = n0 + wn * WC + (lane % LPR) * 8;            // first of the lane's 8 output channels
            const size_t out_base = (size_t)m0 * (size_t)(g.Co * 2);
            const __amdgpu_buffer_rsrc_t rs_out = rsrc_at(a.out, out_base, a.out_bytes);
            const __amdgpu_buffer_rsrc_t rs_add = rsrc_at(a.add ? (const void*)a.add : (const void*)a.out, out_base,
                                                          a.add ? a.out_bytes : out_base);
            const __amdgpu_buffer_rsrc_t rs_msk = rsrc_at(a.mask ? (const void*)a.mask : (const void*)a.out, out_base,
                                                          a.mask ? a.out_bytes : out_base);
            constexpr bool BWE = EPI == EPI_BWE || EPI == EPI_BWE_READ;
            const __amdgpu_buffer_rsrc_t rs_y = rsrc_at(BWE ? a.bw.y : (const void*)a.out, out_base, BWE ? a.out_bytes : out_base);
            // BWE: per channel sum(dz) and the second BatchNorm-backward sum over the wave's 128 rows.  EPI_BWE (the mean is in
            // registers for the mask): sum(dz * (y - mean)), centred per element as conv_nt_kernel does -- the two routes of a launch
            // round alike also where |mean| >> std.  EPI_BWE_READ: sum(dz * y) with the mean taken out at the end, sum(dz * xhat) =
            // rstd * (sum(dz y) - mean sum(dz)) -- eight more registers for the means spill that instantiation (8 / 18 VGPRs, dense /
            // gather) and cost the bf16 step 0.3 ms (47.2 vs 46.9 ms, same box): measured, not taken
            float t_mu[8], t_sc[8], t_sh[8], s1[8], s2[8], t_bias[8];
            const bool nomask = a.mask == nullptr && a.bw.maskbits == nullptr;
            // the mask as one bit per element (IoBwStats::maskbits): a lane's 8 channels are one byte
            const __amdgpu_buffer_rsrc_t rs_bits = rsrc_at(a.bw.maskbits ? (const void*)a.bw.maskbits : (const void*)a.out,
                                                           out_base / 16, a.bw.maskbits ? a.out_bytes / 16 : out_base / 16);
            const int gcol = BWE ? (m0 / a.bw.Mg) * g.Co + ecol : 0;     // (a 256-row tile never straddles two groups)
            if constexpr (BWE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
            }
            if constexpr (EPI == EPI_BWE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    t_mu[e] = a.bw.mean[gcol + e];
                    t_sc[e] = a.bw.mscale[gcol + e];
                    t_sh[e] = a.bw.mshift[gcol + e];
                }
            }

            if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) t_bias[e] = a.bw.bias ? a.bw.bias[ecol + e] : 0.f;
            }
            // The passes of the wave -- (row tile i, half h, row group k): 8 or 16 rows x WC columns each -- as ONE unrolled
            // sequence with the global loads of the epilogue operands (residual gradient, mask, BatchNorm input) issued PD
            // passes ahead: a pass that loaded and waited on its own measured 13-18 % SLOWER than conv_nt_kernel's 2-byte
            // column loads on the data gradients of conv1 (three 16-byte loads in flight per lane, one memory latency per pass).
            constexpr int NP = TM * 2 * NI, PD = (EPI == EPI_STATS) ? 1 : (EPI == EPI_BWE ? (XOP != 0 ? 2 : 4) : 3);   // (XOP: xyn is alive across the epilogue)
            auto pass_off = [&](int p) -> unsigned {
                const int i = p / (2 * NI), h = (p / NI) & 1, k = p % NI;
                return (unsigned)((wm * 128 + i * 32 + h * 16 + k * RPI + lane / LPR) * g.Co + ecol) * 2u;
            };
            u32x4 pav[PD], pmv[PD], pyv[PD];
            unsigned pbv[PD];
            auto pass_load = [&](int p, int sl) {
                const unsigned off = pass_off(p);
                if constexpr (EPI == EPI_PLAIN || EPI == EPI_BWE_READ) {
                    // absent operands have zero-length descriptors: their loads return 0 without touching memory
                    pav[sl] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, off, 0, 0);
                    pmv[sl] = __builtin_amdgcn_raw_buffer_load_b128(rs_msk, off, 0, 0);
                    pbv[sl] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_bits, off >> 4, 0, 0);      // (off = element index * 2)
                }
                if constexpr (BWE) pyv[sl] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, off, 0, 0);
            };
#pragma unroll
            for (int d = 0; d < PD; ++d)
                if (d < NP) pass_load(d, d);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int i = p / (2 * NI), h = (p / NI) & 1, k = p % NI;
                if (k == 0) {
                    // rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32: r >> 3 = h picks 16 of them.  (LDS operations of one
                    // wave stay in order: the reads of the previous 16 rows are done before these writes land.)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r8i = 0; r8i < 8; ++r8i) {
                            const int r = h * 8 + r8i;
                            ep[((r & 3) + 8 * ((r >> 2) & 1) + 4 * (lane >> 5)) * EPP + j * 32 + (lane & 31)] = acc[i][j][r];
                        }
                }
                const int row = k * RPI + lane / LPR;
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(ep + row * EPP + (lane % LPR) * 8);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(ep + row * EPP + (lane % LPR) * 8 + 4);
                float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
                const unsigned off = pass_off(p);
                const int sl = p % PD;
                if constexpr (EPI == EPI_PLAIN || EPI == EPI_BWE_READ) {
                    const u32x4 av = pav[sl], mv = pmv[sl];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        v[2 * d] += bf_lo(av[d]);
                        v[2 * d + 1] += bf_hi(av[d]);
                    }
                    if constexpr (EPI == EPI_PLAIN) {
                        if (a.bw.bias) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                v[e] += t_bias[e];
                                v[e] = (a.bw.relu && v[e] < 0.f) ? 0.f : v[e];
                            }
                        }
                    }
                    // branch-free over the two forms of the mask: the absent one loads zeros (a runtime branch around the uses lets
                    // hipcc sink the loads into it and wait for each: +7 ms on the step, measured)
                    const unsigned bb = pbv[sl];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        v[2 * d] = (nomask || ((bb >> (2 * d)) & 1u) || bf_lo(mv[d]) > 0.f) ? v[2 * d] : 0.f;
                        v[2 * d + 1] = (nomask || ((bb >> (2 * d + 1)) & 1u) || bf_hi(mv[d]) > 0.f) ? v[2 * d + 1] : 0.f;
                    }
                }
                if constexpr (BWE) {
                    const u32x4 yv = pyv[sl];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int e = 2 * d + hh;
                            const float y = hh ? bf_hi(yv[d]) : bf_lo(yv[d]);
                            if constexpr (EPI == EPI_BWE) {
                                const float t = __builtin_fmaf(y - t_mu[e], t_sc[e], t_sh[e]);   // bn(y), bn_apply's fma
                                v[e] = t > 0.f ? v[e] : 0.f;
                            }
                            s1[e] += v[e];
                            if constexpr (EPI == EPI_BWE) s2[e] = __builtin_fmaf(v[e], y - t_mu[e], s2[e]);   // (centred: the mean is in registers anyway)
                            else s2[e] = __builtin_fmaf(v[e], y, s2[e]);
                        }
                    }
                }
                if (p + PD < NP) pass_load(p + PD, sl);
                const u32x4 pk = {io_f2bf2(v[0], v[1]), io_f2bf2(v[2], v[3]), io_f2bf2(v[4], v[5]), io_f2bf2(v[6], v[7])};
                __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, off, 0, 0);
            }
            if constexpr (BWE) {
                // the lanes that hold the same 8 channels (lane % LPR equal) cover the wave's 128 rows between them
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int sft = LPR; sft < 64; sft <<= 1) {
                        s1[e] += __shfl_xor(s1[e], sft, 64);
                        s2[e] += __shfl_xor(s2[e], sft, 64);
                    }
                }
                if (lane < LPR) {
                    const size_t o = (size_t)mt128 * g.Co + ecol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        a.bw.p1[o + e] = s1[e];
                        if constexpr (EPI == EPI_BWE) a.bw.p2[o + e] = a.bw.rstd[gcol + e] * s2[e];
                        else a.bw.p2[o + e] = a.bw.rstd[gcol + e] * (s2[e] - a.bw.mean[gcol + e] * s1[e]);
                    }
                }
            }
        }
        tile = ntile;
        kt = last ? 0 : kt + 1;
        ++q;
    }
}

int p256_ncu() { return io_device_cu_count(); }

std::atomic<int> g_p256{-1};          // -1: not read from the environment yet
int p256_enabled() {
    int on = g_p256.load(std::memory_order_relaxed);
    if (on < 0) {
        const char* e = getenv("IO_P256");
        on = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : ((e && e[0] == '3') ? 3 : 1));
        g_p256.store(on, std::memory_order_relaxed);
    }
    return on;
}

// One 512-thread block per CU walks the tiles: with few tiles, or a last round that leaves most CUs idle, the 128-row
// kernel (four times the blocks, three per CU) fills the chip better -- measured on the MiDaS step at 16 pairs (24 x 24
// and 12 x 12 maps: 9..144 row tiles), where taking every eligible launch cost 2.8 %.
bool p256_rounds_ok(long tiles) {
    const int ncu = p256_ncu();
    const long rounds = (tiles + ncu - 1) / ncu;
    return p256_enabled() == 3 || tiles * 10 >= rounds * ncu * 8;          // at least 80 % of the rounds' slots used
}

// IO_P256_XOP=0 / io_set_bf16_p256_xop(0): the operand forms stay on conv_nt_kernel / separate BatchNorm passes (A/B runs)
std::atomic<int> g_xop{-1};
int xop_enabled() {
    int on = g_xop.load(std::memory_order_relaxed);
    if (on < 0) {
        const char* e = getenv("IO_P256_XOP");
        on = (e && e[0] == '0') ? 0 : 1;
        g_xop.store(on, std::memory_order_relaxed);
    }
    return on;
}

}  // namespace

extern "C" int io_get_bf16_p256_xop(void) { return xop_enabled(); }
extern "C" int io_set_bf16_p256_xop(int on) {
    const int prev = xop_enabled();
    g_xop.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}
// Would io_launch_conv_p256 take a dense 1x1 launch [M x Ci] -> [M x Co] WITH an operand form (executor: is the BatchNorm
// pass worth leaving to the consumer's operand load in bf16)?  Same tests as the launcher, shape part only.
bool io_conv_p256_takes_xop(long M, int Ci, int Co, int Mg) {
    if (!p256_enabled() || !xop_enabled()) return false;
    if (M % 256 != 0 || Ci % 64 != 0 || Ci > kXopMaxCi || Co % 128 != 0 || Mg <= 0 || Mg % 256 != 0 || M % Mg != 0) return false;
    if (256.0 * Ci * 2.0 >= 4.0e9 || 256.0 * Co * 2.0 >= 4.0e9) return false;
    return p256_rounds_ok(M / 256 * (Co / (Co % 256 == 0 ? 256 : 128)));
}

extern "C" int io_get_bf16_p256(void) { return p256_enabled(); }
extern "C" int io_set_bf16_p256(int on) {
    const int prev = p256_enabled();
    g_p256.store((on == 2 || on == 3) ? on : (on ? 1 : 0), std::memory_order_relaxed);
    return prev;
}
int io_bf16_persist_mode() { return p256_enabled(); }

// Returns IO_OK when the launch was taken, 1 when the shape / form is not this kernel's (the caller falls through to
// conv_nt_kernel), < 0 on a launch error.
int io_launch_conv_p256(const IoConvGeom& g, const void* in, const void* wgt, void* out, const void* add, const void* mask,
                        hipStream_t st, float* st_mean, float* st_m2, const IoBwStats* bw, size_t in_bytes,
                        unsigned w_bytes, size_t out_bytes) {
    if (!p256_enabled()) return 1;
    const long M = (long)g.N * g.Ho * g.Wo;
    const bool dense_out = g.os == 1 && g.Ho == g.outH && g.Wo == g.outW;
    if (g.gw || g.cr || !dense_out || M % 256 != 0 || g.Ci % 64 != 0 || g.Co % 128 != 0 || g.Th * g.Tw < 1) return 1;
    if (bw && (bw->in_scale || bw->a_out || bw->wino_u)) return 1;
    const bool lin = g.Th * g.Tw == 1 && g.is == 1 && g.dh0 == 0 && g.dw0 == 0 && g.Hi == g.Ho && g.Wi == g.Wo;
    // the operand forms (IoBwStats::xb_a) as an in-LDS transform: dense 1x1 launches, groups of whole 256-row tiles
    const bool xop = bw && bw->xb_a;
    if (xop && (!lin || !xop_enabled() || g.Ci > kXopMaxCi || !bw->xb_b || !bw->xb_c || !bw->xb_y || bw->xb_Mg <= 0 || bw->xb_Mg % 256 != 0 ||
                M % bw->xb_Mg != 0 || add || mask || bw->maskbits || bw->bias))
        return 1;
    if (st_mean && (add || mask || (bw && (bw->y || bw->bias)))) return 1;
    if (bw && bw->y && (bw->Mg % 256 != 0 || bw->bias)) return 1;
    if (bw && bw->y && bw->mscale && (add || mask || bw->maskbits)) return 1;         // (the executor never combines them)
    // 32-bit offsets inside a tile's reach: 256 rows of the output, the samples a tile touches of the input
    const double span = 256.0 / ((double)g.Ho * g.Wo) + 2.0;
    if (span * 2.0 * g.Hi * g.Wi * g.Ci >= 4.0e9 || 256.0 * g.Co * 2.0 >= 4.0e9) return 1;
    if (g.Hi >= 32768 || g.Wi >= 32768) return 1;
    // where the 256-wide tile pays (tools/bf16_dma_probe.hip): N >= 256; 128-wide tiles for N = 128 (mod 256)
    const int bn = g.Co % 256 == 0 ? 256 : 128;
    const int epi = st_mean ? EPI_STATS : ((bw && bw->y) ? (bw->mscale ? EPI_BWE : EPI_BWE_READ) : EPI_PLAIN);
    P256Args a;
    memset(&a, 0, sizeof(a));
    a.in = (const bf16_t*)in;
    a.wgt = (const bf16_t*)wgt;
    a.out = (bf16_t*)out;
    a.add = (const bf16_t*)add;
    a.mask = (bw && bw->maskbits) ? nullptr : (const bf16_t*)mask;      // (the bit form of the same mask, when there is one)
    a.in_bytes = in_bytes;
    a.out_bytes = out_bytes;
    a.w_bytes = w_bytes;
    a.st_mean = st_mean;
    a.st_m2 = st_m2;
    if (bw) a.bw = *bw;
    if (xop) {
        a.xy = (const bf16_t*)bw->xb_y;
        a.xa = bw->xb_a;
        a.xb = bw->xb_b;
        a.xc = bw->xb_c;
        a.xout = (bf16_t*)bw->xb_out;
        a.xbits = (uint8_t*)bw->xb_bits;
        a.xmode = bw->xb_res == 2 ? 3 : (bw->xb_res ? 2 : 1);
        a.xMg = bw->xb_Mg;
    }
    a.ntn = g.Co / bn;
    const long tiles = (M / 256) * a.ntn;
    if (tiles >= (1L << 31)) return 1;
    a.ntiles = (int)tiles;
    const int ncu = p256_ncu();
    if (!p256_rounds_ok(tiles)) return 1;
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    const double kred = (double)g.Th * g.Tw * g.Ci;
    IoProfScope prof(bn == 256 ? IO_PROF_CONV_NT128 : IO_PROF_CONV_NT64, 2.0 * (double)M * g.Co * kred,
                     2.0 * M * g.Co * (1.0 + (add ? 1.0 : 0.0) + (mask ? 1.0 : 0.0) + ((bw && bw->y) ? 1.0 : 0.0)) +
                         2.0 * ((double)g.N * g.Hi * g.Wi * g.Ci * (1.0 + (xop ? 1.0 + (bw->xb_out ? 1.0 : 0.0) : 0.0)) +
                                (double)g.Co * kred),
                     st);
#define IO_P256_LAUNCH(BN_, EPI_, ...)                                                                              \
    do {                                                                                                            \
        const size_t lds = (size_t)2 * (256 + BN_) * 128 + (xop ? (size_t)12 * g.Ci : 0);                           \
        static std::atomic<unsigned long long> attr_done{0};                                                        \
        if (io_first_on_device(attr_done))                                                                          \
            (void)hipFuncSetAttribute((const void*)conv_p256_kernel<BN_, EPI_, __VA_ARGS__>,                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize,                                   \
                                      (int)(2 * (256 + BN_) * 128 + (xop ? 12 * kXopMaxCi : 0)));                   \
        hipLaunchKernelGGL((conv_p256_kernel<BN_, EPI_, __VA_ARGS__>), dim3((unsigned)grid), dim3(512), lds, st, g, a); \
    } while (0)
#define IO_P256_EPI(BN_, G_)                                         \
    do {                                                             \
        if (epi == EPI_STATS) IO_P256_LAUNCH(BN_, EPI_STATS, G_);    \
        else if (epi == EPI_BWE) IO_P256_LAUNCH(BN_, EPI_BWE, G_);   \
        else if (epi == EPI_BWE_READ) IO_P256_LAUNCH(BN_, EPI_BWE_READ, G_); \
        else IO_P256_LAUNCH(BN_, EPI_PLAIN, G_);                     \
    } while (0)
#define IO_P256_XOP(BN_)                                                   \
    do {                                                                   \
        if (a.xmode == 2) {                                                \
            if (epi == EPI_STATS) IO_P256_LAUNCH(BN_, EPI_STATS, false, 2); \
            else IO_P256_LAUNCH(BN_, EPI_PLAIN, false, 2);                 \
        } else if (epi == EPI_STATS) IO_P256_LAUNCH(BN_, EPI_STATS, false, 1); \
        else if (epi == EPI_BWE) IO_P256_LAUNCH(BN_, EPI_BWE, false, 1);   \
        else IO_P256_LAUNCH(BN_, EPI_PLAIN, false, 1);                     \
    } while (0)
    if (xop && (epi == EPI_BWE_READ || (epi == EPI_BWE && a.xmode == 2))) return 1;   // (not instantiated: the executor has no such launch)
    if (xop) {
        if (bn == 256) IO_P256_XOP(256);
        else IO_P256_XOP(128);
    } else if (bn == 256) {
        if (lin) IO_P256_EPI(256, false);
        else IO_P256_EPI(256, true);
    } else {
        if (lin) IO_P256_EPI(128, false);
        else IO_P256_EPI(128, true);
    }
#undef IO_P256_XOP
#undef IO_P256_EPI
#undef IO_P256_LAUNCH
    return io_check_launch("conv_p256");
}
