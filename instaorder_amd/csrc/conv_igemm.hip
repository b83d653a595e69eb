// Implicit-GEMM convolutions on the matrix cores of gfx950: v_mfma_f32_32x32x2_f32 on fp32 data (the reference's
// precision) and v_mfma_f32_32x32x16_bf16 on bf16 data (dtype switch), fp32 accumulation in both.
//
// Everything the ResNet / ResNeXt paths need from "conv" is one of two GEMM shapes over NHWC data:
//
//   NT  out[m][o] = sum_{tap,c} In[pix(m,tap)][c] * W[o][tap][c]      (+ add[m][o])
//       forward conv (resnet_cls.py:23-31, :140) and both data-gradient forms; with IoConvGeom::gw also the grouped
//       3x3 convolution as a block-diagonal one.  Both operands are K-contiguous, so the k index is permuted
//       freely: each lane fetches 16 bytes of consecutive k with one ds_read_b128 and feeds them to 4 fp32 MFMAs or
//       1 bf16 MFMA.  Optional epilogues: BatchNorm statistics of the output, residual add, ReLU mask, BatchNorm
//       backward reductions.
//   TN  dW[o][tap][c] = sum_m dY[m][o] * In[pix(m,tap)][c]
//       weight gradient; the reduction index m is the slow (row) dimension of both operands.  fp32: tiles are
//       stored as loaded ([m][channel]) and fragments are column slices (ds_read_b32, consecutive lanes ->
//       consecutive banks).  bf16: the staging pass transposes 8x8 blocks in registers so that the LDS image is the
//       NT kernel's.  Split over m with a deterministic second pass.
//
// fp32 MFMA runs at the vector rate (64 FLOP/clk/SIMD), so the fp32 kernels are compute bound by a wide margin
// (LDS needs ~16 B/clk/CU of 256); the design goal is simply to keep one MFMA chain per SIMD busy: 128x128x32 block
// tile, 4 waves in 2x2, 2x2 MFMA tiles per wave (4 independent accumulators), register-staged double buffering
// (global loads of tile k+1 fly under the 64 MFMAs of tile k), 73 KiB LDS -> 2 blocks/CU.  Operands are addressed
// with 32-bit offsets through buffer descriptors rebased per tile, so tensors may exceed 4 GiB.
#include <string.h>
#include <stdlib.h>

#include "io_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Pins a register value at this point of the program: the LDS read that produces it must have been ISSUED (and waited
// for) before.  The GEMM loops prefetch the last fragment group of a k-tile and then pass a barrier behind which the
// other waves overwrite the tile -- and hipcc sinks such ds_reads below the s_barrier (they are only used after the
// NEXT barrier; observed in conv_wgrad_wino_kernel, where the sunk reads raced with the refill whenever two blocks shared
// a CU).  __syncthreads() does not stop that; a value that is "modified" by an empty asm statement before it does.
template <typename T> __device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// bijective XCD-aware remap of a 1-D grid: blocks that run on one XCD (b % 8) get a contiguous
// range of tile ids, so neighbouring tiles (same A rows, different output-channel tile) share
// that XCD's L2.  Only affects speed.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// n / d for small non-negative n (< 2^16) with rd = 1.0f / d: (n + 0.5) * rd is at least 0.5 / d away from an integer
__device__ __forceinline__ int sdiv(int n, float rd) { return (int)(((float)n + 0.5f) * rd); }
__device__ __forceinline__ int stem_kp(int wT, int cr) { return (wT * cr + 31) / 32 * 32; }   // = io_stem_kp
__device__ __forceinline__ int fdiv(int n, IoFastDiv f) {
    return f.shift < 0 ? n : (int)(__umulhi((unsigned)n, f.magic) >> f.shift);
}

// 16-byte load through a buffer descriptor: 32-bit byte offset, and an offset past the end of the tensor
// (kInvalidOff, or rows beyond M) returns zeros in hardware -- no 64-bit address arithmetic, no selects.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kInvalidOff = 0xFFFFFFFFu;
// Output stores of the convolution epilogues are NON-TEMPORAL (aux bit 1 = nt).  What is known about that bit here
// (profiles/r02_store_and_shortk_probes.txt): on ONE re-used buffer pair (cache-warm) nt looks 31-47 % faster on the
// bf16 1x1 layers; on rotating buffers plain stores win by 8-25 % on the write-heavy layers, because a lane owns one
// column of the accumulator layout and nt sends its 64-byte half lines to memory unmerged (write-only probe: 3.3 TB/s
// nt vs 5.5-6.0 plain for 2-byte stores; no such gap for 4-byte stores or whole lines); in the network the two builds are
// indistinguishable (the BatchNorm pass that follows pays back what the convolution gains).  nt is kept; the bf16
// epilogues that matter store whole rows through LDS instead.  (sc0 / sc1 bits: no gain / -15 %.)
constexpr int kStAux = 2;       // aux bits of the epilogue stores: nt
constexpr int kTrBkm = 64;      // rows (output pixels) per k-tile of the LDS-DMA filter-gradient kernel ...
constexpr int kTrMinB = 2;      // ... the blocks per CU asked of the register allocator ...
constexpr int kTrStages = 2;    // ... and its LDS stages.  Round 5, same-box A/B on the bf16 step (filter-gradient class 9.6 ms): 3 or
                                // 4 stages of 32 rows at two blocks per CU 9.7-9.9 ms, 3 or 4 stages of 64 rows at one block 10.3-10.5
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// Descriptor over the tail of a tensor that starts `base` bytes in: a block addresses only the samples its tile
// touches, relative to the first of them, so 32-bit offsets never limit the size of the whole tensor (activations of
// 1024-pair batches exceed 4 GiB).  num_records saturates at 2^32-1; kInvalidOff stays out of range either way.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_at(const void* p, size_t base, size_t total) {
    const size_t rest = total > base ? total - base : 0;
    return make_rsrc(static_cast<const char*>(p) + base, rest > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)rest);
}
__device__ __forceinline__ f32x4 bld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
// 4 consecutive elements at byte offset `off`, widened to fp32 (filter-gradient kernel: fp32 MFMA on bf16 data)
template <typename T> __device__ __forceinline__ f32x4 bldv(__amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ f32x4 bldv<float>(__amdgpu_buffer_rsrc_t r, unsigned off) { return bld4(r, off); }
template <> __device__ __forceinline__ f32x4 bldv<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    f32x4 v;
    v[0] = __builtin_bit_cast(float, q[0] << 16);
    v[1] = __builtin_bit_cast(float, q[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, q[1] << 16);
    v[3] = __builtin_bit_cast(float, q[1] & 0xffff0000u);
    return v;
}
// one element
template <typename T> __device__ __forceinline__ float ld_el(__amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ float ld_el<float>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
template <> __device__ __forceinline__ float ld_el<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return io_bf2f(__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
}
template <typename T> __device__ __forceinline__ void st_el(float v, __amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ void st_el<float>(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, kStAux);
}
template <> __device__ __forceinline__ void st_el<bf16_t>(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b16(io_f2bf(v), r, off, 0, kStAux);
}
// the same with a wave-uniform (SGPR) offset added on top of the per-lane one.  NOTE: the scalar offset takes no part in
// the bounds check of the descriptor -- only for accesses known to be in range.
template <typename T> __device__ __forceinline__ float ld_el_s(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff);
template <> __device__ __forceinline__ float ld_el_s<float>(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}
template <> __device__ __forceinline__ float ld_el_s<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    return io_bf2f(__builtin_amdgcn_raw_buffer_load_b16(r, off, soff, 0));
}
template <typename T> __device__ __forceinline__ void st_el_s(float v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff);
template <> __device__ __forceinline__ void st_el_s<float>(float v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, soff, kStAux);
}
template <> __device__ __forceinline__ void st_el_s<bf16_t>(float v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b16(io_f2bf(v), r, off, soff, kStAux);
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------
// NT kernel.  NW waves per block share one 128 x BN tile: NW = 4 -> 2x2 waves of 64 x BN/2 (what is
// launched); NW = 8 -> 2x4 waves of 64 x BN/4 measured identical (the kernel is not latency bound).
// ------------------------------------------------------------------------------------------
// TA: storage type of the operands (in, wgt): float -> v_mfma_f32_32x32x2_f32, bf16 -> v_mfma_f32_32x32x16_bf16.
// A k-tile row is always 128 bytes (32 floats or 64 bf16) and a lane always moves 16-byte chunks, so the LDS
// image, the addressing and the fragment reads are byte-for-byte the same in both modes.  TO: storage type of
// out / add / mask / bw.y (the fp32 stem writes bf16 activations in bf16 mode).
// STEM: 0 regular conv, 1 stem on the packed x8 input (k = tap x 8 channels, 3 of them padding), 2 stem in exact-K
// mode (fp32 only): k = tap * g.cr + channel over the real channels, gathered one dword at a time, filters pre-packed
// as [Co][kp] -- 8 k-tiles instead of 13 for the 7x7 / 5-channel stem.
// NBUF = 1: single LDS buffer (a second barrier per k-tile) + MINB blocks per CU requested from the register allocator --
// more tiles in flight per CU for the latency-bound bf16 shapes.
// BWE: the instantiation that carries the fused BatchNorm-backward epilogue (bw.y); every other launch -- forward
// convolutions, plain data gradients -- runs the BWE = false build, whose register allocation does not pay for it.
// XF: the instantiation whose A operand goes through an input transform (IoBwStats::in_scale): BatchNorm scale / shift +
// ReLU applied to the staged chunk between its global load and its LDS store.
// XB: the BACKWARD operand transform of a data-gradient launch (IoBwStats::xb_a): the A operand is the gradient of a
// BatchNorm INPUT, dy = a[c] * dz + b[c] * y + c[c], evaluated from the masked gradient dz (= `in`) and that BatchNorm's
// input y while the chunk is staged -- two 16-byte loads per chunk instead of one -- so BatchNorm backward has no apply
// pass of its own; the blocks of the first output-channel tile also write dy out (centre tap) for the filter gradient.
// XB = 2: the FORWARD sibling with the same two-load staging and side output -- the A operand of a block's conv1 is the
// previous block's output relu(bn3(y3) + identity), evaluated from y3 (= `in`) and the identity tensor (xb_y) with
// bn_apply_kernel's own expression (tables a = scale, b = mean, c = shift) and written out once as that block's output
// tensor: the residual BatchNorm pass `out = relu(bn3(conv3(.)) + identity)` (resnet_cls.py:108-114) has no launch of its own.
// LIN: dense 1x1 stride-1 GEMM on whole tiles (row m of the output IS pixel m of the input, 128 | M): no row decoding, no
// validity, k offsets and output row steps ride in the scalar offset of the buffer instructions.  The generic path spends
// ~800 VALU + ~500 SALU instructions per wave and tile on addressing; with a bf16 tile of K <= 512 worth only 16..128
// MFMAs the SIMDs were instruction-issue bound on exactly these layers (measured: 69 % issue utilisation, 19 % MFMA).
// WINO (fp32, BN = 64, 3x3 stride-1 same-size convolutions / data gradients on whole 128-pixel tiles, Wo even): the
// convolution along the image ROW in Winograd's minimal form F(2, 3).  A pair of horizontally adjacent output pixels
// (w, w + 1), w even, needs per filter row 4 products per (input channel, output channel) instead of 6: with d0..d3 the
// inputs at columns w - 1 .. w + 2 of that row and g0..g2 the filter row,
//     V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)        U = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)
//     m_f = sum over filter rows and input channels of V_f * U_f            (f = 0..3: FOUR independent GEMMs)
//     y(w) = m0 + m1 + m2        y(w + 1) = m1 - m2 - m3
// so the 3x3 layer runs 2/3 of the MFMAs of the direct form.  The block tile is the same 128 output pixels (= 64 pairs) x
// 64 output channels; a wave owns 32 pairs x 32 channels x the FOUR frequencies -- four 32x32 accumulators, the same 64
// registers as the direct kernel's 2 x 2 tile -- and because the four accumulators of a lane hold the SAME (pair, channel)
// entries, the output transform is plain register arithmetic.  U is prepared by wino_filter_kernel (layout
// [filter row][f][Co][Ci]); V is formed between the global loads and the LDS stores of the A operand (after the optional
// input transform XF), i.e. the LDS image holds 4 x 64 rows of V and 4 x 64 rows of U per k-tile, and a k-tile is one
// filter row x 32 input channels: K = 3 Ci per frequency.  Everything behind the accumulators -- statistics, the fused
// BatchNorm-backward epilogue, side outputs -- is the direct kernel's code on (even pixel, odd pixel) tiles.
template <typename TA, typename TO, int BN, int STEM, int NW, int NBUF = 2, int MINB = 1, bool BWE = false,
          bool XF = false, bool LIN = false, int XB = 0, bool WINO = false>
__global__ __launch_bounds__(NW * 64, MINB) void conv_nt_kernel(IoConvGeom g, const TA* __restrict__ in,
                                                         const TA* __restrict__ wgt, TO* __restrict__ out,
                                                         const TO* __restrict__ add,
                                                         const TO* __restrict__ mask, int ntn, size_t in_bytes,
                                                         unsigned w_bytes, size_t out_bytes,
                                                         float* __restrict__ st_mean, float* __restrict__ st_m2,
                                                         IoBwStats bw) {
    constexpr int ES = sizeof(TA), OS = sizeof(TO);
    constexpr int VE = 16 / ES;             // elements per 16-byte chunk
    // LDS rows: BN = 128 pads them to 36 words (conflict-free 16-byte fragment reads, 73.7 KB, 2 blocks per CU either
    // way); BN = 64 keeps them at 32 words and XORs the 16-byte chunk index with bits 1..3 of the row instead -- equally
    // conflict-free (16 consecutive rows of one logical chunk hit 16 distinct bank quads) and 48 KB, so THREE blocks
    // share a CU: the short-K / HBM-bound 64-channel layers and the stem want tiles in flight, not a bigger tile.
    constexpr bool SWZ = BN == 64;
    constexpr int BM = 128, BK = 128 / ES, LDT = SWZ ? 32 : 32 + 4, NT = NW * 64;   // LDT in 4-byte words
    constexpr int CPT = ES / 2;             // stem: 16-byte chunks per 8-channel tap (fp32 2, bf16 1)
    constexpr int TPT = 8 / CPT;            // stem: taps per k-tile
    constexpr bool XK = STEM == 2;          // exact-K stem
    static_assert(!XK || ES == 4, "the exact-K stem is an fp32 path");
    constexpr int WN = NW / 2;              // waves along N (2 along M)
    static_assert(!WINO || (ES == 4 && OS == 4 && BN == 64 && NW == 4 && STEM == 0 && NBUF == 1 && !LIN && XB == 0),
                  "WINO: fp32, 64-wide tiles, single LDS buffer, plain / XF / BWE forms");
    constexpr int TI = WINO ? 4 : 2, TJ = BN / (32 * WN);   // WINO: the 4 frequencies of 32 pairs
    constexpr int ETI = 2;                  // row blocks of a wave's OUTPUT tile (WINO: even / odd pixels of its 32 pairs)
    constexpr int AROWS = WINO ? 2 * BM : BM, BROWS = WINO ? 4 * BN : BN;   // LDS rows per operand tile
    constexpr int AR = (BM * 8) / NT;       // A rows loaded per thread (8 float4 per row)
    constexpr int BR = (BN * 8) / NT;       // weight rows loaded per thread
    constexpr int RS = NT / 8;              // row step between a thread's rows
    static_assert(TJ >= 1 && BR >= 1, "tile too small for this wave count");
    // the row-wise bf16 epilogues turn 32 x (BN / WN) fp32 values per wave through the operand LDS
    static_assert(NBUF * (BM + BN) * LDT >= NW * 32 * (BN / WN + 4), "operand LDS too small for the row epilogue");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                       // [2][AROWS][LDT]
    float* sB = smem + NBUF * AROWS * LDT;  // [NBUF][BROWS][LDT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile / ntn;
    const int m0 = mt * BM, n0 = (tile - mt * ntn) * BN;
    const int HoWo = g.Ho * g.Wo;
    const int M = g.N * HoWo;
    const int lr = tid >> 3, kq = tid & 7;
    // grouped (block-diagonal) mode: this tile's output channels read the gw input channels starting at n0
    const int kw = g.gw ? g.gw : g.Ci;           // filter row length per tap
    const int cbase = g.gw ? n0 : 0;             // first input channel of the k range
    const int nkc = STEM ? 1 : kw / BK;
    const int kp = XK ? stem_kp(g.wT, g.cr) : 0;
    const int nk = XK ? kp / 32 : STEM ? (g.wT + TPT - 1) / TPT : (WINO ? g.Th : g.Th * g.Tw) * nkc;

    // Addressing: every operand row gets ONE 32-bit byte offset per tile (rowv / wv); a k-tile adds a
    // wave-uniform tap/channel offset to it.  Invalid rows / padding taps get kInvalidOff and the buffer
    // unit returns zeros.
    static_assert(!LIN || STEM == 0, "LIN: regular 1x1 convolutions");
    // first sample of this tile: every offset below is relative to it (LIN: relative to the tile's first row)
    const int n_lo = LIN ? 0 : fdiv(m0, g.fd_howo);
    const __amdgpu_buffer_rsrc_t rs_in =
        make_rsrc_at(in, LIN ? (size_t)m0 * (size_t)(g.Ci * ES) : (size_t)n_lo * (size_t)(g.Hi * g.Wi) * (size_t)(g.Ci * ES),
                     in_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wgt, w_bytes);
    unsigned rowv[AR];
    int hi0[AR], wi0[AR];
    bool rvalid[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        if constexpr (LIN) {
            rvalid[j] = true;
            hi0[j] = wi0[j] = 0;
            rowv[j] = (unsigned)((lr + RS * j) * g.Ci + kq * VE) * (unsigned)ES;
        } else {
            const int m = m0 + lr + RS * j;
            rvalid[j] = m < M;
            const int mm = rvalid[j] ? m : 0;
            const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
            const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
            hi0[j] = ho * g.is;
            wi0[j] = wo * g.is;
            rowv[j] = (unsigned)((((n - n_lo) * g.Hi + hi0[j]) * g.Wi + wi0[j]) * g.Ci + (STEM ? 0 : kq * VE)) * (unsigned)ES;
        }
    }
    // WINO: a thread stages AI = 2 (pair, 16-byte channel chunk) items of the A operand -- pairs lr and lr + 32 of the tile --
    // from the 4 pixels w - 1 .. w + 2 of the pair's row, and 8 rows (f, channel) of U
    constexpr int AI = 2, WBR = 8;
    unsigned wrow[AI];
    int whi[AI];
    bool wl[AI], wr[AI];
    unsigned wvb[WBR];
    f32x4 wa[WINO ? AI : 1][4], wb[WINO ? WBR : 1];
    if constexpr (WINO) {
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            const int m = m0 + 2 * (lr + 32 * j);                      // whole tiles: always < M
            const int n = fdiv(m, g.fd_howo), rem = m - n * HoWo;
            const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
            whi[j] = ho;
            wl[j] = wo > 0;
            wr[j] = wo + 2 < g.Wi;
            wrow[j] = (unsigned)((((n - n_lo) * g.Hi + ho) * g.Wi + wo) * g.Ci + kq * 4) * 4u;
        }
#pragma unroll
        for (int j = 0; j < WBR; ++j)     // LDS row lr + 32 j = (f = j >> 1, channel lr + 32 (j & 1)); U is [th][f][Co][Ci]
            wvb[j] = (unsigned)(((j >> 1) * g.Co + n0 + lr + 32 * (j & 1)) * g.Ci + kq * 4) * 4u;
    }
    unsigned wv[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j)
        wv[j] = XK ? (unsigned)((n0 + lr + RS * j) * kp + kq * VE) * (unsigned)ES
                   : (unsigned)((n0 + lr + RS * j) * g.wT * kw + (STEM ? 0 : kq * VE)) * (unsigned)ES;
    const bool nopad = g.Th == 1 && g.Tw == 1 && g.dh0 == 0 && g.dw0 == 0 && g.is == 1 && g.Hi >= g.Ho &&
                       g.Wi >= g.Wo;   // 1x1 stride-1: a row is valid for every k-tile or for none

    f32x4 ra[AR], rb[BR];
    // XF: per-channel coefficients of the chunk being loaded (they change with the k-tile) and the validity of its rows
    static_assert(!XF || STEM == 0, "input transform: regular convolutions");
    static_assert(!XB || (STEM == 0 && !XF), "backward operand transform: regular data gradients, not combined with XF");
    static_assert(XB < 2 || !BWE, "the residual forms are forward paths");
    constexpr bool XT = XF || XB;           // some operand transform
    constexpr int XC = ES == 4 ? 1 : 2;     // 16-byte coefficient loads per 16-byte operand chunk (4 floats or 8 bf16)
    f32x4 xm[XC], xs[XC], xh[XC];           // XF: mean, scale, shift.  XB: b (times y), a (times dz), c
    f32x4 ry[XB ? AR : 1];                  // XB: the chunk of y that goes with ra
    unsigned xok = 0, xaoff = 0;            // XB: xaoff = the k-tile's wave-uniform offset (side output goes where dz came from)
    bool xside = false;                     // XB: this k-tile is the centre tap (every pixel exactly once)
    const int xgrp = XF ? m0 / bw.in_Mg : XB ? m0 / bw.xb_Mg : 0;
    const float* const xt_s = XF ? bw.in_scale : bw.xb_a;
    const float* const xt_h = XF ? bw.in_shift : bw.xb_c;
    const float* const xt_m = XF ? bw.in_mean : bw.xb_b;
    const __amdgpu_buffer_rsrc_t rs_xs = make_rsrc(XT ? (const void*)(xt_s + (size_t)xgrp * g.Ci) : (const void*)wgt,
                                                   XT ? (unsigned)g.Ci * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_xh = make_rsrc(XT ? (const void*)(xt_h + (size_t)xgrp * g.Ci) : (const void*)wgt,
                                                   XT ? (unsigned)g.Ci * 4u : 0u);
    // (no mean table: a zero-length descriptor, whose loads return 0)
    const __amdgpu_buffer_rsrc_t rs_xm = make_rsrc((XT && xt_m) ? (const void*)(xt_m + (size_t)xgrp * g.Ci)
                                                                : (const void*)wgt,
                                                   (XT && xt_m) ? (unsigned)g.Ci * 4u : 0u);
    // XB: y is addressed exactly like `in`; the side output too, through a descriptor that is EMPTY (stores dropped)
    // unless this block owns the first output-channel tile
    const size_t xb_base = LIN ? (size_t)m0 * (size_t)(g.Ci * ES) : (size_t)n_lo * (size_t)(g.Hi * g.Wi) * (size_t)(g.Ci * ES);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc_at(XB ? bw.xb_y : (const void*)in, xb_base, XB ? in_bytes : xb_base);
    const __amdgpu_buffer_rsrc_t rs_side =
        make_rsrc_at((XB && bw.xb_out) ? bw.xb_out : (void*)out, xb_base, (XB && bw.xb_out && n0 == 0) ? in_bytes : xb_base);
    int th = 0, tw = 0, cc = 0;   // tap / channel-chunk counters of the k-tile being LOADED (non-stem)
    // Loads are branch-free and the loop body below is ONE basic block (the last iteration simply
    // re-fetches the final k-tile and discards it), so the scheduler is free to sink the address
    // arithmetic and the global loads of tile k+1 into the shadow of the 64 MFMAs of tile k.
    const float rcr = 1.0f / (float)(XK ? g.cr : 1), rS = 1.0f / (float)g.S;
    auto load_tile = [&](int kt) {
        if constexpr (XK) {
            // a lane's 4 consecutive k indices belong to (up to) two taps: one dword load per element
            unsigned eoff[4];
            int edh[4], edw[4];
            bool eok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = kt * 32 + kq * 4 + i;
                const int tp = sdiv(e, rcr), ch = e - tp * g.cr;
                eok[i] = tp < g.wT;
                const int r = sdiv(tp, rS), s = tp - r * g.S;
                edh[i] = g.dh0 + g.dhs * r;
                edw[i] = g.dw0 + g.dws * s;
                eoff[i] = (unsigned)((edh[i] * g.Wi + edw[i]) * g.Ci + ch) * 4u;
            }
#pragma unroll
            for (int j = 0; j < AR; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hi = hi0[j] + edh[i], wi = wi0[j] + edw[i];
                    const bool ok = eok[i] && rvalid[j] && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                    ra[j][i] = ld_el<float>(rs_in, ok ? rowv[j] + eoff[i] : kInvalidOff);
                }
#pragma unroll
            for (int j = 0; j < BR; ++j) rb[j] = bld4(rs_w, wv[j] + (unsigned)kt * 128u);
            return;
        }
        if constexpr (LIN) {
            const unsigned koff = (unsigned)(cc * BK) * (unsigned)ES;        // scalar: the channel chunk of this k-tile
            if constexpr (XB) {
                xaoff = koff;
#pragma unroll
                for (int j = 0; j < AR; ++j)
                    ry[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, rowv[j], koff, 0));
            }
            if constexpr (XT) {
                const unsigned coff = (unsigned)(cc * BK + kq * VE) * 4u;
#pragma unroll
                for (int q = 0; q < XC; ++q) {
                    xm[q] = bld4(rs_xm, coff + 16u * q);
                    xs[q] = bld4(rs_xs, coff + 16u * q);
                    xh[q] = bld4(rs_xh, coff + 16u * q);
                }
                xok = ~0u;
            }
#pragma unroll
            for (int j = 0; j < AR; ++j)
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, rowv[j], koff, 0));
#pragma unroll
            for (int j = 0; j < BR; ++j)
                rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv[j], koff, 0));
            return;
        }
        if constexpr (WINO) {
            const int dh = g.dh0 + g.dhs * th;
            // (unsigned wrap-around: "- 1 pixel" on an offset that stays inside the descriptor whenever it is used)
            const unsigned aoff = (unsigned)(((dh * g.Wi - 1) * g.Ci + cc * BK) * 4);
            const unsigned woff = (unsigned)((th * 4 * g.Co * g.Ci + cc * BK) * 4);
            if constexpr (XF) {
                const unsigned coff = (unsigned)(cc * BK + kq * VE) * 4u;
                xm[0] = bld4(rs_xm, coff);
                xs[0] = bld4(rs_xs, coff);
                xh[0] = bld4(rs_xh, coff);
            }
            xok = 0;
#pragma unroll
            for (int j = 0; j < AI; ++j) {
                const bool rowok = (unsigned)(whi[j] + dh) < (unsigned)g.Hi;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = rowok & (i == 0 ? wl[j] : i == 3 ? wr[j] : true);
                    xok |= ok ? (1u << (j * 4 + i)) : 0u;
                    wa[j][i] = bld4(rs_in, ok ? wrow[j] + aoff + (unsigned)(i * g.Ci * 4) : kInvalidOff);
                }
            }
#pragma unroll
            for (int j = 0; j < WBR; ++j) wb[j] = bld4(rs_w, wvb[j] + woff);
            return;
        }
        int dh, dw;
        unsigned aoff, woff;     // wave-uniform for the regular path
        bool tapok = true;
        if (STEM) {
            const int tap = kt * TPT + kq / CPT;
            tapok = tap < g.wT;
            const int r = tap / g.S, s = tap - r * g.S;
            dh = g.dh0 + g.dhs * r;
            dw = g.dw0 + g.dws * s;
            aoff = (unsigned)((dh * g.Wi + dw) * g.Ci + (kq % CPT) * VE) * (unsigned)ES;
            woff = (unsigned)((tapok ? tap : 0) * g.Ci + (kq % CPT) * VE) * (unsigned)ES;
        } else {
            dh = g.dh0 + g.dhs * th;
            dw = g.dw0 + g.dws * tw;
            const int widx = (g.r0 + g.rs * th) * g.S + (g.s0 + g.ss * tw);
            aoff = (unsigned)((dh * g.Wi + dw) * g.Ci + cbase + cc * BK) * (unsigned)ES;
            woff = (unsigned)(widx * kw + cc * BK) * (unsigned)ES;
        }
        if constexpr (XT) {
            const unsigned coff = (unsigned)(cbase + cc * BK + kq * VE) * 4u;
#pragma unroll
            for (int q = 0; q < XC; ++q) {
                xm[q] = bld4(rs_xm, coff + 16u * q);
                xs[q] = bld4(rs_xs, coff + 16u * q);
                xh[q] = bld4(rs_xh, coff + 16u * q);
            }
            xok = 0;
        }
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            bool ok = tapok && rvalid[j];
            if (!nopad) {
                const int hi = hi0[j] + dh, wi = wi0[j] + dw;
                ok = ok && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
            }
            if constexpr (XT) xok |= ok ? (1u << j) : 0u;
            ra[j] = bld4(rs_in, ok ? rowv[j] + aoff : kInvalidOff);
            if constexpr (XB) ry[j] = bld4(rs_y, ok ? rowv[j] + aoff : kInvalidOff);
        }
        if constexpr (XB) {
            xaoff = aoff;
            xside = dh == 0 && dw == 0;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) rb[j] = bld4(rs_w, tapok ? wv[j] + woff : kInvalidOff);
    };
    auto advance = [&](bool really) {    // step the (th, tw, cc) counters unless we are re-fetching
        if constexpr (LIN) {
            cc += really ? 1 : 0;
            return;
        }
        if (!STEM) {
            const int c1 = cc + 1;
            const bool wrapc = c1 == nkc;
            const int t1 = tw + (wrapc ? 1 : 0);
            const bool wrapt = t1 == (WINO ? 1 : g.Tw);
            cc = really ? (wrapc ? 0 : c1) : cc;
            tw = really ? (wrapt ? 0 : t1) : tw;
            th = really ? th + (wrapt ? 1 : 0) : th;
        }
    };
    const int wchunk = SWZ ? (kq ^ ((lr >> 1) & 7)) : kq;      // (the row step RS = 32 leaves bits 1..3 alone)
    // XB: the operand transform of the tile in (ra, ry) + its side output.  Its own step so that the loop can run it under
    // the MFMAs of the previous tile instead of inside the barrier-to-barrier section of store_tile.
    auto xform_tile = [&]() {
        if constexpr (WINO) {
#pragma unroll
            for (int j = 0; j < AI; ++j) {
                if constexpr (XF) {
                    // relu(bn(x)) of the 4 pixels first (bn_apply_kernel's expression; padding stays zero)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool ok = (xok >> (j * 4 + i)) & 1u;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = fmaxf(__builtin_fmaf(wa[j][i][e] - xm[0][e], xs[0][e], xh[0][e]), 0.f);
                            wa[j][i][e] = ok ? v : 0.f;
                        }
                    }
                }
                const f32x4 d0 = wa[j][0], d1 = wa[j][1], d2 = wa[j][2], d3 = wa[j][3];
                wa[j][0] = d0 - d2;
                wa[j][1] = d1 + d2;
                wa[j][2] = d2 - d1;
                wa[j][3] = d1 - d3;
            }
            return;
        }
        if constexpr (XF) {
            // relu((x - mean) * scale + shift) on the staged chunk (bn_apply_kernel's expression); rows that are padding (or past M) stay zero -- the transform of
            // the zeros the buffer unit returned would be relu(shift)
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const bool ok = (xok >> j) & 1u;
                if constexpr (ES == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fmaxf(__builtin_fmaf(ra[j][e] - xm[0][e], xs[0][e], xh[0][e]), 0.f);
                        ra[j][e] = ok ? v : 0.f;
                    }
                } else {
                    const u32x4 raw = __builtin_bit_cast(u32x4, ra[j]);
                    u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {     // dword d holds elements 2d (low half) and 2d + 1 (high half)
                        const int q = d >> 1, e0 = (d & 1) * 2;
                        const float lo = __builtin_bit_cast(float, raw[d] << 16);
                        const float hi = __builtin_bit_cast(float, raw[d] & 0xffff0000u);
                        const float vl = fmaxf(__builtin_fmaf(lo - xm[q][e0], xs[q][e0], xh[q][e0]), 0.f);
                        const float vh = fmaxf(__builtin_fmaf(hi - xm[q][e0 + 1], xs[q][e0 + 1], xh[q][e0 + 1]), 0.f);
                        o[d] = ok ? io_f2bf2(vl, vh) : 0u;
                    }
                    ra[j] = __builtin_bit_cast(f32x4, o);
                }
            }
        }
        if constexpr (XB != 0) {
            // dy = a * dz + (b * y + c) on the staged chunk; padding rows stay zero (their transform would be c)
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const bool ok = (xok >> j) & 1u;
                if constexpr (XB == 2 && ES == 4) {
                    // relu(bn(y) + identity): bn_apply_kernel's expression (MODE 1), so the tensor is bit for bit the one a
                    // separate pass would have written
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = __builtin_fmaf(ra[j][e] - xm[0][e], xs[0][e], xh[0][e]) + ry[j][e];
                        ra[j][e] = (ok && v > 0.f) ? v : 0.f;
                    }
                } else if constexpr (ES == 4) {
                    // XB = 3: the affine form followed by a ReLU -- the output of a block with a downsample branch,
                    // relu(bn3(y3) + bnd(yd)) = relu(a * y3 + b * yd + c) with the two BatchNorms folded into one table set
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = __builtin_fmaf(ra[j][e], xs[0][e], __builtin_fmaf(ry[j][e], xm[0][e], xh[0][e]));
                        ra[j][e] = (ok && (XB != 3 || v > 0.f)) ? v : 0.f;
                    }
                } else {
                    const u32x4 raw = __builtin_bit_cast(u32x4, ra[j]), rwy = __builtin_bit_cast(u32x4, ry[j]);
                    u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {     // dword d holds elements 2d (low half) and 2d + 1 (high half)
                        const int q = d >> 1, e0 = (d & 1) * 2;
                        const float lo = __builtin_bit_cast(float, raw[d] << 16);
                        const float hi = __builtin_bit_cast(float, raw[d] & 0xffff0000u);
                        const float yl = __builtin_bit_cast(float, rwy[d] << 16);
                        const float yh = __builtin_bit_cast(float, rwy[d] & 0xffff0000u);
                        float vl, vh;
                        if constexpr (XB == 2) {        // relu(bn(y) + identity), bn_apply_kernel's expression
                            vl = __builtin_fmaf(lo - xm[q][e0], xs[q][e0], xh[q][e0]) + yl;
                            vh = __builtin_fmaf(hi - xm[q][e0 + 1], xs[q][e0 + 1], xh[q][e0 + 1]) + yh;
                        } else {
                            vl = __builtin_fmaf(lo, xs[q][e0], __builtin_fmaf(yl, xm[q][e0], xh[q][e0]));
                            vh = __builtin_fmaf(hi, xs[q][e0 + 1], __builtin_fmaf(yh, xm[q][e0 + 1], xh[q][e0 + 1]));
                        }
                        if constexpr (XB >= 2) {
                            vl = vl > 0.f ? vl : 0.f;
                            vh = vh > 0.f ? vh : 0.f;
                        }
                        o[d] = ok ? io_f2bf2(vl, vh) : 0u;
                    }
                    ra[j] = __builtin_bit_cast(f32x4, o);
                }
            }
            // side output: the transformed chunk goes back out where dz came from (first output-channel tile, centre tap)
            if constexpr (LIN) {
#pragma unroll
                for (int j = 0; j < AR; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ra[j]), rs_side, rowv[j], xaoff, 0);
            } else {
#pragma unroll
                for (int j = 0; j < AR; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ra[j]), rs_side,
                                                           (xside && ((xok >> j) & 1u)) ? rowv[j] + xaoff : kInvalidOff, 0, 0);
            }
        }
    };
    auto store_tile = [&](int buf, bool xdone = false) {
        if constexpr (WINO) {
            if (!xdone) xform_tile();
            float* a = sA + lr * LDT + wchunk * 4;      // rows f * 64 + pair (pairs lr, lr + 32: the same swizzle bits)
            float* b = sB + lr * LDT + wchunk * 4;      // rows lr + 32 j
#pragma unroll
            for (int j = 0; j < AI; ++j)
#pragma unroll
                for (int f = 0; f < 4; ++f) st4(a + (f * 64 + 32 * j) * LDT, wa[j][f]);
#pragma unroll
            for (int j = 0; j < WBR; ++j) st4(b + 32 * j * LDT, wb[j]);
            return;
        }
        float* a = sA + buf * BM * LDT + lr * LDT + wchunk * 4;
        float* b = sB + buf * BN * LDT + lr * LDT + wchunk * 4;
        if constexpr (XT) {
            if (!xdone) xform_tile();
        }
#pragma unroll
        for (int j = 0; j < AR; ++j) st4(a + RS * j * LDT, ra[j]);
#pragma unroll
        for (int j = 0; j < BR; ++j) st4(b + RS * j * LDT, rb[j]);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Software pipeline.  Two waves share each SIMD's matrix pipe; if a wave had a long MFMA-free phase
    // per k-tile (address math, global loads, LDS refill, barrier) the two waves fall into lock-step
    // and the pipe idles during that phase (measured: 78 % MfmaUtil).  So the MFMA stream of a wave is
    // kept continuous: fragments are prefetched one 16-MFMA group ahead, the loads of tile k+1 are
    // issued under the first 48 MFMAs of tile k, and the LDS refill + barrier sit between MFMA groups
    // 3 and 4 of tile k -- whose operands are already in registers -- with the first fragments of
    // tile k+1 fetched right behind the barrier, under that last group.
    const int rsw = SWZ ? (lane >> 1) & 7 : 0;                 // all fragment rows of a lane are = lane mod 32
    const int a_off = (wm * (WINO ? 32 : 64) + (lane & 31)) * LDT + (SWZ ? 0 : (lane >> 5) * 4);
    const int b_off = (wn * (BN / WN) + (lane & 31)) * LDT + (SWZ ? 0 : (lane >> 5) * 4);
    constexpr int FB = WINO ? 4 : TJ;       // B fragments per read (WINO: one per frequency)
    auto read_frags = [&](int buf, int kk, f32x4 (&a)[TI], f32x4 (&b)[FB]) {
        const int koff = SWZ ? ((((lane >> 5) + kk * 2) ^ rsw) * 4) : kk * 8;
        if constexpr (WINO) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                a[f] = ld4(sA + a_off + koff + f * 64 * LDT);
                b[f] = ld4(sB + b_off + koff + f * 64 * LDT);
            }
            return;
        }
        const float* al = sA + buf * BM * LDT + a_off + koff;
        const float* bl = sB + buf * BN * LDT + b_off + koff;
#pragma unroll
        for (int i = 0; i < TI; ++i) a[i] = ld4(al + i * 32 * LDT);
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[j] = ld4(bl + j * 32 * LDT);
    };
    // one fragment read (16 B per lane and tile) feeds 4 fp32 MFMAs (k = 2 each) or 1 bf16 MFMA (k = 16)
    auto mma16 = [&](const f32x4 (&a)[TI], const f32x4 (&b)[FB]) {
        if constexpr (WINO) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f][tt], b[f][tt], acc[f][0], 0, 0, 0);
        } else if constexpr (ES == 4) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][tt], b[j][tt], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                        __builtin_bit_cast(bf16x8, b[j]), acc[i][j],
                                                                        0, 0, 0);
        }
    };

    f32x4 fa[TI], fb[FB];
    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    if (nk > 0) read_frags(0, 0, fa, fb);
    // All k-tiles but the last: fetch tile kt+1 under the MFMAs of tile kt.  The last tile runs in a peeled block without
    // loads, LDS refill or barriers: the loop used to be ONE body whose last trip re-fetched the final tile and waited
    // for it before the refill it then discarded -- one full memory latency per output tile, which is 1/3 of the chain
    // of a K = 64 fp32 tile and 1/2 of a K = 64 bf16 one.
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int buf = NBUF == 2 ? kt & 1 : 0;
        const int nbuf = NBUF == 2 ? buf ^ 1 : 0;
        advance(true);
        // (fp32 only: in bf16 the transform is VALU-bound whatever its place and the longer live ranges spill --
        // profiles/r03_xb_microbench_bf16.txt; the bf16 step does not use the operand forms)
        constexpr bool PIPE = ES == 4 && (XB != 0 || XF || WINO);
        if constexpr (!PIPE) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            f32x4 na[TI], nb[FB];
            read_frags(buf, kk + 1, na, nb);
            mma16(fa, fb);
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[i] = na[i];
#pragma unroll
            for (int j = 0; j < FB; ++j) fb[j] = nb[j];
            if constexpr (PIPE) {
                // operand forms: the fetches of tile kt+1 go out behind the first MFMA group, and the transform of what
                // they bring runs under the third -- not between the barriers, where all four waves of the block wait
                if (kk == 0) {
                    load_tile(kt + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kk == 1) xform_tile();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TI; ++i) pin(fa[i]);      // the group-3 fragments are in registers BEFORE the barrier
#pragma unroll
        for (int j = 0; j < FB; ++j) pin(fb[j]);
        if (NBUF == 1) __syncthreads();        // every wave has read the last fragments of tile kt
        store_tile(nbuf, PIPE);
        __syncthreads();
        f32x4 na[TI], nb[FB];
        read_frags(nbuf, 0, na, nb);           // first fragments of tile kt+1
        __builtin_amdgcn_sched_barrier(0);
        mma16(fa, fb);                         // group 4 of tile kt, operands already in registers
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[i] = na[i];
#pragma unroll
        for (int j = 0; j < FB; ++j) fb[j] = nb[j];
    }
    if (nk > 0) {
        const int buf = NBUF == 2 ? (nk - 1) & 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            f32x4 na[TI], nb[FB];
            read_frags(buf, kk + 1, na, nb);
            mma16(fa, fb);
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[i] = na[i];
#pragma unroll
            for (int j = 0; j < FB; ++j) fb[j] = nb[j];
        }
        mma16(fa, fb);
    }

    // WINO: the output transform -- y(even pixel) = m0 + m1 + m2, y(odd pixel) = m1 - m2 - m3 -- in place: from here on
    // acc[0] / acc[1] are the wave's OUTPUT tiles (32 even / 32 odd pixels x 32 channels).  Row of element r of tile i inside
    // the wave's 64 rows: direct i * 32 + q, WINO 2 q + i, with q = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    if constexpr (WINO) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m1 = acc[1][0][r], m2 = acc[2][0][r];
            acc[0][0][r] = acc[0][0][r] + m1 + m2;
            acc[1][0][r] = m1 - m2 - acc[3][0][r];
        }
    }
    constexpr int RQ = WINO ? 2 : 1, RI = WINO ? 1 : 32;      // row = RI * i + RQ * q
    // Fused BatchNorm statistics (forward convs in training mode): per (row tile, channel) the mean of the
    // tile's valid rows and the sum of squared deviations from it -- two in-register passes over the
    // accumulators, combined across the two row-waves through LDS.  Merged later with Chan's update.
    if (st_mean) {
        __syncthreads();                     // every wave is done with the operand tiles: LDS is free
        float* red = smem;                   // [2][BN]
        const int nvalid = min(BM, M - m0);
        float cmean[TJ];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float sacc = 0.f;
#pragma unroll
                for (int i = 0; i < ETI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + RI * i + RQ * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
                        const float v = acc[i][j][r];
                        const float d = pass == 0 ? v : (v - cmean[j]) * (v - cmean[j]);
                        sacc += m < M ? d : 0.f;
                    }
                sacc += __shfl_xor(sacc, 32, 64);
                if (lane < 32) red[wm * BN + wn * (BN / WN) + j * 32 + lane] = sacc;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int cl = wn * (BN / WN) + j * 32 + (lane & 31);
                const float tot = red[cl] + red[BN + cl];
                if (pass == 0) {
                    cmean[j] = tot / (float)nvalid;
                } else if (wm == 0 && lane < 32) {
                    const size_t o = (size_t)mt * g.Co + n0 + cl;
                    st_mean[o] = cmean[j];
                    st_m2[o] = tot;
                }
            }
            __syncthreads();
        }
    }

    // epilogue: D layout col = lane&31 (output channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // One 32-bit byte offset per (i, r) row through a buffer descriptor (rows past M get kInvalidOff:
    // their stores are dropped and their `add` loads return 0); the `add` variant issues all its loads
    // before the stores.
    const bool dense = (g.os == 1) && (g.Ho == g.outH) && (g.Wo == g.outW);
    const int opix_lo = LIN ? m0 : n_lo * g.outH * g.outW;   // < 2^31: it is a pixel count, not a byte count
    const size_t out_base = (size_t)opix_lo * (size_t)(g.Co * OS);
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_at(out, out_base, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_add = make_rsrc_at(add ? (const void*)add : (const void*)out, out_base, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_mask = make_rsrc_at(mask ? (const void*)mask : (const void*)out, out_base, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_bwy = make_rsrc_at(bw.y ? bw.y : (const void*)out, out_base, out_bytes);
    float bw_mu[TJ], bw_rs[TJ], bw_sc[TJ], bw_sh[TJ], bw_s1[TJ], bw_s2[TJ];
    if constexpr (BWE) {
        const int gcol = (m0 / bw.Mg) * g.Co + n0 + wn * (BN / WN) + (lane & 31);   // group is uniform per tile
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            bw_mu[j] = bw.mean[gcol + j * 32];
            bw_rs[j] = bw.rstd[gcol + j * 32];
            bw_sc[j] = bw.mscale ? bw.mscale[gcol + j * 32] : 0.f;
            bw_sh[j] = bw.mscale ? bw.mshift[gcol + j * 32] : 0.f;
            bw_s1[j] = 0.f;
            bw_s2[j] = 0.f;
        }
        // The launcher guarantees a dense output and 128 | rows per group here: row m IS pixel m and no tile is
        // partial, so there is nothing to validate.  One VGPR offset per lane; the 16 row steps of the accumulator
        // layout ride in the scalar offset of the buffer instructions (no per-row offset registers, no selects).
        const unsigned colb2 = (unsigned)(n0 + wn * (BN / WN) + (lane & 31)) * (unsigned)OS;
        const unsigned rowstep = (unsigned)g.Co * (unsigned)OS;
        const unsigned lane_base = (unsigned)(m0 - opix_lo + wm * 64 + RQ * 4 * (lane >> 5)) * rowstep + colb2;
        const __amdgpu_buffer_rsrc_t rs_add0 = add ? rs_add : make_rsrc(out, 0);
        const __amdgpu_buffer_rsrc_t rs_mask0 = mask ? rs_mask : make_rsrc(out, 0);
        const bool nomask = mask == nullptr;
        // side output (optional): relu(bn(y)) of the BatchNorm whose mask is recomputed here -- the activation the
        // forward pass never stored (its consumer read y through the input transform); the filter gradient of that
        // consumer wants it as a plain tensor.  y is in registers anyway: one more store stream, no extra read.
        const __amdgpu_buffer_rsrc_t rs_aout = make_rsrc_at(bw.a_out ? bw.a_out : (void*)out, out_base,
                                                            bw.a_out ? out_bytes : out_base);
        // bf16, 128-wide tiles: dz leaves through the wave's LDS slice as whole rows (see the plain 1x1 epilogue below for
        // the why); the optional activation side output keeps its column stores
        constexpr bool ROWS = OS == 2 && BN == 128;
        constexpr int WC = BN / WN, EPP = WC + 4, LPR = WC / 4, RPI = 64 / LPR, NI = 32 / RPI;
        float* ep = smem + wave * (32 * EPP);
        if (ROWS) __syncthreads();               // every wave is done with the operand tiles
#pragma unroll
        for (int i = 0; i < ETI; ++i) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const unsigned voff = lane_base + (unsigned)(i * RI) * rowstep + (unsigned)j * 32u * OS;
                // All three tensors are fetched unconditionally -- a per-element "load or constant" on a runtime pointer
                // makes hipcc branch around every load and drain the queue each time; an absent `add` / `mask` has a
                // zero-length descriptor instead, whose loads return 0 without touching memory.
                float av[16], mv[16], yv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = ld_el_s<TO>(rs_bwy, voff, (unsigned)(RQ * ((r & 3) + 8 * (r >> 2))) * rowstep);
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = ld_el_s<TO>(rs_add0, voff, (unsigned)(RQ * ((r & 3) + 8 * (r >> 2))) * rowstep);
#pragma unroll
                for (int r = 0; r < 16; ++r) mv[r] = ld_el_s<TO>(rs_mask0, voff, (unsigned)(RQ * ((r & 3) + 8 * (r >> 2))) * rowstep);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + av[r];
                    v = (mv[r] > 0.f || nomask) ? v : 0.f;
                    const float t = __builtin_fmaf(yv[r] - bw_mu[j], bw_sc[j], bw_sh[j]);     // bn(y), bn_apply's fma
                    if (bw.mscale) v = t > 0.f ? v : 0.f;
                    bw_s1[j] += v;
                    bw_s2[j] += v * ((yv[r] - bw_mu[j]) * bw_rs[j]);
                    if constexpr (ROWS) ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPP + j * 32 + (lane & 31)] = v;
                    else st_el_s<TO>(v, rs_out, voff, (unsigned)(RQ * ((r & 3) + 8 * (r >> 2))) * rowstep);
                    // (no a_out: a zero-length descriptor drops the store)
                    st_el_s<TO>(fmaxf(t, 0.f), rs_aout, voff, (unsigned)(RQ * ((r & 3) + 8 * (r >> 2))) * rowstep);
                }
            }
            if constexpr (ROWS) {
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int row = k * RPI + lane / LPR, cc = (lane % LPR) * 4;
                    const f32x4 q = ld4(ep + row * EPP + cc);
                    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                    const u32x2_ pk = {io_f2bf2(q[0], q[1]), io_f2bf2(q[2], q[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(
                        pk, rs_out, (unsigned)((m0 - opix_lo + wm * 64 + i * 32 + row) * g.Co + n0 + wn * WC + cc) * 2u, 0,
                        kStAux);
                }
            }
        }
        // (the reduction of the partial sums below starts with a barrier before it reuses LDS)
    }
    if constexpr (!BWE) {
    float ep_bias[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) ep_bias[j] = bw.bias ? bw.bias[n0 + wn * (BN / WN) + j * 32 + (lane & 31)] : 0.f;
    if constexpr (LIN) {
        // dense output, whole tiles: one VGPR offset per lane and column block, the 16 row steps of the accumulator
        // layout in the scalar offset (as in the fused BatchNorm-backward epilogue above)
        const unsigned rowstep = (unsigned)g.Co * (unsigned)OS;
        const unsigned lane_base = (unsigned)((wm * 64 + 4 * (lane >> 5)) * g.Co + n0 + wn * (BN / WN) + (lane & 31)) * (unsigned)OS;
        if constexpr (OS == 2 && BN == 128) {
            // bf16 output without residual / mask: the accumulator layout gives a lane ONE column, i.e. 2-byte stores
            // that reach memory as 64-byte half lines -- 3.3 TB/s with the non-temporal bit against 5.6+ for whole
            // lines (profiles/r02_store_and_shortk_probes.txt), and the p -> 4p layers write four times what they read.
            // So each wave turns its quadrant through its own slice of the (now idle) operand LDS, 32 rows at a time: fp32
            // in column order, out as 4 consecutive channels of a row per lane -> 8-byte stores, 16 lanes = one 128-byte
            // line.  LDS traffic of a wave stays in order, so the slice needs no barrier after the first one.  (128-wide
            // tiles only: a 64-wide tile's rows are 64-byte segments either way, measured 4 % slower this way.)
            // Isolated, rotating buffers: 64 -> 256 0.384 -> 0.306 ms, 128 -> 512 0.258 -> 0.183, 256 -> 1024 0.128 ->
            // 0.111; in the bf16 step the NT class 21.49 -> 20.97 ms (a timing-only build with 16-byte stores and no LDS
            // trip bounds the idea at 20.41).
            if (!add && !mask) {
                constexpr int WC = BN / WN, EPP = WC + 4, LPR = WC / 4, RPI = 64 / LPR, NI = 32 / RPI;
                __syncthreads();                 // every wave is done with the operand tiles
                float* ep = smem + wave * (32 * EPP);
#pragma unroll
                for (int i = 0; i < ETI; ++i) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[i][j][r];
                            if (bw.bias) {
                                v += ep_bias[j];
                                v = (bw.relu && v < 0.f) ? 0.f : v;
                            }
                            ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPP + j * 32 + (lane & 31)] = v;
                        }
#pragma unroll
                    for (int k = 0; k < NI; ++k) {
                        const int row = k * RPI + lane / LPR, cc = (lane % LPR) * 4;
                        const f32x4 q = ld4(ep + row * EPP + cc);
                        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                        const u32x2_ pk = {io_f2bf2(q[0], q[1]), io_f2bf2(q[2], q[3])};
                        __builtin_amdgcn_raw_buffer_store_b64(
                            pk, rs_out, (unsigned)((wm * 64 + i * 32 + row) * g.Co + n0 + wn * WC + cc) * 2u, 0, kStAux);
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < ETI; ++i) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const unsigned voff = lane_base + (unsigned)(i * 32) * rowstep + (unsigned)j * 32u * OS;
                if (add) {
                    float av[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) av[r] = ld_el_s<TO>(rs_add, voff, (unsigned)((r & 3) + 8 * (r >> 2)) * rowstep);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += av[r];
                }
                if (bw.bias) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r] + ep_bias[j];
                        acc[i][j][r] = (bw.relu && v < 0.f) ? 0.f : v;
                    }
                }
                if (mask) {
                    float mv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) mv[r] = ld_el_s<TO>(rs_mask, voff, (unsigned)((r & 3) + 8 * (r >> 2)) * rowstep);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = mv[r] > 0.f ? acc[i][j][r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r];
                    st_el_s<TO>(v, rs_out, voff, (unsigned)((r & 3) + 8 * (r >> 2)) * rowstep);
                }
            }
        }
    } else {
    const unsigned colb = (unsigned)(n0 + wn * (BN / WN) + (lane & 31)) * (unsigned)OS;
    constexpr unsigned JS = 32u * OS;     // byte step between a lane's column blocks
#pragma unroll
    for (int i = 0; i < ETI; ++i) {
        unsigned rowb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + RI * i + RQ * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
            const bool ok = m < M;
            int pix = ok ? m : m0;
            if (!dense) {
                const int n = fdiv(pix, g.fd_howo), rem = pix - n * HoWo;
                const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
                pix = (n * g.outH + (ho * g.os + g.ooh)) * g.outW + (wo * g.os + g.oow);
            }
            rowb[r] = ok ? (unsigned)((pix - opix_lo) * g.Co) * (unsigned)OS + colb : kInvalidOff;
        }
        if (add) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float av[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    av[r] = ld_el<TO>(rs_add, rowb[r] == kInvalidOff ? kInvalidOff : rowb[r] + j * JS);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += av[r];
            }
        }
        if (bw.bias) {   // folded BatchNorm of an inference forward: + bias[o], optional ReLU
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] + ep_bias[j];
                    acc[i][j][r] = (bw.relu && v < 0.f) ? 0.f : v;
                }
        }
        if (mask) {      // ReLU backward of the tensor this gradient belongs to: zero where it was clipped
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float mv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mv[r] = ld_el<TO>(rs_mask, rowb[r] == kInvalidOff ? kInvalidOff : rowb[r] + j * JS);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = mv[r] > 0.f ? acc[i][j][r] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const float v = acc[i][j][r];      // (bit_cast straight from the vector element miscompiles)
                st_el<TO>(v, rs_out, rowb[r] == kInvalidOff ? kInvalidOff : rowb[r] + j * JS);
            }
        }
    }
    }   // !LIN
    }   // generic epilogue
    if constexpr (BWE) {
        __syncthreads();
        float* red = smem;                   // [2 sums][2 row-waves][BN]
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const float a = bw_s1[j] + __shfl_xor(bw_s1[j], 32, 64);
            const float b = bw_s2[j] + __shfl_xor(bw_s2[j], 32, 64);
            if (lane < 32) {
                const int cl = wn * (BN / WN) + j * 32 + lane;
                red[wm * BN + cl] = a;
                red[2 * BN + wm * BN + cl] = b;
            }
        }
        __syncthreads();
        if (wm == 0 && lane < 32) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int cl = wn * (BN / WN) + j * 32 + lane;
                const size_t o = (size_t)mt * g.Co + n0 + cl;
                bw.p1[o] = red[cl] + red[BN + cl];
                bw.p2[o] = red[2 * BN + cl] + red[3 * BN + cl];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN (weight gradient) kernel
// ------------------------------------------------------------------------------------------
// fp32 MFMA.  TX / TDY: storage of the conv input and of dY; bf16 operands are widened when they are staged (used
// for the stem in bf16 mode: fp32 packed input x bf16 dY; all other bf16 convs run conv_wgrad_bf16_kernel below).
// STEM as in the NT kernel; 2 (exact-K, fp32 input): columns are k = tap * g.cr + channel, dW is the packed [Co][kp].
// TR (fp32 operands, not the stem): transposed staging.  A thread fetches 4 CONSECUTIVE rows m of its 4-channel chunk
// and writes the 4x4 block transposed -- pure register renaming -- so the LDS image is [channel][32 m] (36-word
// pitch), k-contiguous like the NT kernel's: fragments are one ds_read_b128 per 4 MFMAs instead of four
// ds_read_b32 (24 instead of 136 LDS instructions per thread and k-tile).  The 16-byte chunk index is XORed with
// bits 4..5 of the channel so that both the transposed writes (16 lanes = 16 channel chunks, 4 channels apart) and
// the fragment reads (16 consecutive channels) touch 16 distinct bank quads.
// W4: the row decode of prep() shared by a thread's rows -- with TR: 4 | Wo (4 consecutive wo); without TR: 32 | Wo (the
// 32 rows of a k-tile lie in one output row, whose (n, ho) is wave-uniform).
template <typename TX, typename TDY, int BMO, int BNC, int STEM, bool TR = false, bool W4 = false, int NBUF = 2,
          int MINB = 1>
__global__ __launch_bounds__(kThreads, MINB) void conv_wgrad_kernel(IoConvGeom g, const TX* __restrict__ in,
                                                             const TDY* __restrict__ dy,
                                                             float* __restrict__ dst, int ntile_c, int tiles,
                                                             int kps, size_t in_bytes, size_t dy_bytes) {
    constexpr int BKM = 32;
    constexpr int TI = BMO / 64, TJ = BNC / 64;
    constexpr int QA = BMO / 4, QB = BNC / 4;          // float4 per tile row
    constexpr int RA = (BKM * QA) / kThreads;          // rows per thread (A)
    constexpr int RB = (BKM * QB) / kThreads;
    constexpr int SA = kThreads / QA, SB = kThreads / QB;   // row step between a thread's rows
    static_assert(!TR || (STEM == 0 && sizeof(TX) == 4 && sizeof(TDY) == 4), "transposed staging: fp32, not the stem");
    constexpr int LDT = 36;                            // TR: words per LDS row (32 m + 4 pad)
    constexpr int NA = TR ? 4 : RA, NB = TR ? 4 : RB;  // 16-byte loads per thread and operand
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                        // [2][BKM][BMO]   (TR: [2][BMO][LDT])
    float* sB = smem + (TR ? NBUF * BMO * LDT : NBUF * BKM * BMO);   // [2][BKM][BNC]   (TR: [2][BNC][LDT])

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = g.Ho * g.Wo;
    const int M = g.N * HoWo;
    const int T = g.Th * g.Tw;

    // 1-D grid, XCD-remapped: the blocks of one m-range (all taps / channel tiles of a split) share an L2.
    // tile decode: tile -> (o tile, tap, channel tile)
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / tiles, tile = logical - split * tiles;
    const int per_o = STEM ? ntile_c : T * ntile_c;
    const int ot = tile / per_o;
    const int rem0 = tile - ot * per_o;
    const int o0 = ot * BMO;
    int tap = 0, c0;
    if (STEM) {
        c0 = rem0 * BNC;                 // column in the flattened (tap, 8 channels) axis
    } else {
        tap = rem0 / ntile_c;
        c0 = g.gw ? o0 : (rem0 - tap * ntile_c) * BNC;      // grouped: only the diagonal (o tile == c tile)
    }
    const int qa = tid % QA, ra0 = tid / QA;
    const int qb = tid % QB, rb0 = tid / QB;
    // TR: ra0 / rb0 number the 4-row groups of a k-tile (8 of them); with 64-wide tiles only half the threads stage
    const bool actA = !TR || ra0 < BKM / 4, actB = !TR || rb0 < BKM / 4;

    constexpr bool XK = STEM >= 2;            // exact-K stem; STEM == 3: 32 | Wo (uniform row decode per k-tile)
    static_assert(!XK || sizeof(TX) == 4, "the exact-K stem gathers fp32 dwords");
    int dh = 0, dw = 0, widx = 0, coff = 0;
    bool tapok = true;
    int xdh[4], xdw[4], xch[4];       // exact-K: tap offsets / channel of this thread's 4 columns
    bool xok[4];
    if (XK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = c0 + qb * 4 + i;
            const int tp = col / g.cr;
            xch[i] = col - tp * g.cr;
            xok[i] = tp < g.wT;
            const int r = tp / g.S, s = tp - r * g.S;
            xdh[i] = g.dh0 + g.dhs * r;
            xdw[i] = g.dw0 + g.dws * s;
        }
    } else if (STEM) {
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
    const int kt0 = split * kps;
    const int kt1 = min(kt0 + kps, nkt);

    // 32-bit byte offsets through buffer descriptors that start at this split's first row (dY) / first sample
    // (In); rows past M fall off the end of dY / In and read 0.
    // `lin`: 1x1 stride-1 (the gathered pixel of row m is pixel m): no per-k-tile decoding at all.
    const int mfirst = min(kt0 * BKM, M - 1);
    const int n_lo = fdiv(mfirst, g.fd_howo);
    const int ipix_lo = n_lo * g.Hi * g.Wi;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc_at(in, (size_t)ipix_lo * (size_t)(g.Ci * (int)sizeof(TX)), in_bytes);
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc_at(dy, (size_t)mfirst * (size_t)(g.Co * (int)sizeof(TDY)), dy_bytes);
    const bool lin = !STEM && T == 1 && g.is == 1 && g.dh0 == 0 && g.dw0 == 0 && g.Hi == g.Ho && g.Wi == g.Wo;
    f32x4 ra[NA], rb[NB];
    // The gather offsets of a k-tile are computed one iteration ahead of its loads (`prep`), so that the loads
    // themselves are the first thing a loop iteration issues and have the whole MFMA stream to land under.
    unsigned offb[XK ? 4 * RB : NB];
    // TR with 4 | Wo (every ResNet shape): a thread's 4 consecutive rows are 4 consecutive wo of ONE output row, so a
    // k-tile costs one row decode and four adds instead of four decodes
    static_assert(!W4 || STEM == 0, "W4: regular convolutions");
    auto prep = [&](int kt) {
        const int mb = kt * BKM;
        if constexpr (W4 && !TR) {
            const bool ok0 = tapok && mb < M;                  // uniform: scalar decode of the k-tile's first row
            const int mm = ok0 ? mb : 0;
            const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
            const int ho = fdiv(rem, g.fd_wo), wo_b = rem - ho * g.Wo;
            const int hi = ho * g.is + dh;
            const bool okh = ok0 && (unsigned)hi < (unsigned)g.Hi;
            const int base = (((n - n_lo) * g.Hi + hi) * g.Wi + dw) * g.Ci + coff;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int wis = (wo_b + rb0 + SB * j) * g.is;
                const bool ok = okh && (unsigned)(wis + dw) < (unsigned)g.Wi;
                offb[j] = ok ? (unsigned)(base + wis * g.Ci) * (unsigned)sizeof(TX) : kInvalidOff;
            }
            return;
        }
        if constexpr (W4 && TR) {
            const int m0 = mb + 4 * rb0;
            const bool ok0 = tapok && actB && m0 < M;
            const int mm = ok0 ? m0 : 0;
            const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
            const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
            const int hi = ho * g.is + dh, wi0 = wo * g.is + dw;
            const bool okh = ok0 && (unsigned)hi < (unsigned)g.Hi;
            const int base = (((n - n_lo) * g.Hi + hi) * g.Wi + wi0) * g.Ci + coff;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const bool ok = okh && (unsigned)(wi0 + j * g.is) < (unsigned)g.Wi;
                offb[j] = ok ? (unsigned)(base + j * g.is * g.Ci) * (unsigned)sizeof(TX) : kInvalidOff;
            }
            return;        // (1x1 stride-1 layers are this form too: dh = dw = 0, is = 1)
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int m = TR ? mb + 4 * rb0 + j : mb + rb0 + SB * j;
            if constexpr (STEM == 3) {
                if (j == 0) {
                    // 32 | Wo: the k-tile's 32 rows lie in one output row -- (n, ho) decoded once on the scalar unit,
                    // the 4 taps of this thread's columns checked once for both of its rows
                    const bool ok0 = mb < M;
                    const int mm = ok0 ? mb : 0;
                    const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
                    const int ho = fdiv(rem, g.fd_wo), wo_b = rem - ho * g.Wo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int hi = ho * g.is + xdh[i];
                        const bool okh = ok0 && xok[i] && (unsigned)hi < (unsigned)g.Hi;
                        const int base = (((n - n_lo) * g.Hi + hi) * g.Wi + xdw[i]) * g.Ci + xch[i];
#pragma unroll
                        for (int jj = 0; jj < RB; ++jj) {
                            const int wis = (wo_b + rb0 + SB * jj) * g.is;
                            const bool ok = okh && (unsigned)(wis + xdw[i]) < (unsigned)g.Wi;
                            offb[4 * jj + i] = ok ? (unsigned)(base + wis * g.Ci) * 4u : kInvalidOff;
                        }
                    }
                }
            } else if constexpr (XK) {
                const bool rok = m < M;
                const int mm = rok ? m : 0;
                const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
                const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hi = ho * g.is + xdh[i], wi = wo * g.is + xdw[i];
                    const bool ok = rok && xok[i] && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                    offb[4 * j + i] = ok ? (unsigned)((((n - n_lo) * g.Hi + hi) * g.Wi + wi) * g.Ci + xch[i]) * 4u
                                         : kInvalidOff;
                }
            } else if (lin) {
                offb[j] = actB ? (unsigned)((m - ipix_lo) * g.Ci + coff) * (unsigned)sizeof(TX) : kInvalidOff;
            } else {
                bool ok = tapok && m < M && actB;
                const int mm = ok ? m : 0;
                const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
                const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
                const int hi = ho * g.is + dh, wi = wo * g.is + dw;
                ok = ok && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                offb[j] = ok ? (unsigned)((((n - n_lo) * g.Hi + hi) * g.Wi + wi) * g.Ci + coff) * (unsigned)sizeof(TX)
                             : kInvalidOff;
            }
        }
    };
    auto load_tile = [&](int kt) {
        const int mb = kt * BKM;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int m = TR ? mb + 4 * ra0 + j : mb + ra0 + SA * j;
            ra[j] = bldv<TDY>(rs_dy, actA ? (unsigned)((m - mfirst) * g.Co + o0 + qa * 4) * (unsigned)sizeof(TDY)
                                          : kInvalidOff);                                        // m >= M -> 0
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (XK) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rb[j][i] = ld_el<float>(rs_in, offb[4 * j + i]);
            } else {
                rb[j] = bldv<TX>(rs_in, offb[j]);
            }
        }
    };
    auto store_tile = [&](int buf) {
        if constexpr (TR) {
            // rows = channels qa*4 + i, 16-byte chunk = the row group, XORed with bits 4..5 of the channel
            float* a = sA + buf * BMO * LDT + (qa * 4) * LDT + ((ra0 ^ ((qa >> 2) & 3)) * 4);
            float* b = sB + buf * BNC * LDT + (qb * 4) * LDT + ((rb0 ^ ((qb >> 2) & 3)) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 ta = {ra[0][i], ra[1][i], ra[2][i], ra[3][i]};
                const f32x4 tb = {rb[0][i], rb[1][i], rb[2][i], rb[3][i]};
                if (actA) st4(a + i * LDT, ta);
                if (actB) st4(b + i * LDT, tb);
            }
            return;
        }
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

    // same software pipeline as the NT kernel: fragments one 16-MFMA group (4 k-steps) ahead, LDS refill
    // and barrier between groups 3 and 4 of a k-tile.
    const int a_off = (lane >> 5) * BMO + wm * (BMO / 2) + (lane & 31);
    const int b_off = (lane >> 5) * BNC + wn * (BNC / 2) + (lane & 31);
    auto read_frags = [&](int buf, int grp, float (&a)[4][TI], float (&b)[4][TJ]) {
        if constexpr (TR) {
            // one 16-byte read per operand tile: 4 consecutive m of the lane's channel (lanes >= 32: the next 4),
            // element s4 feeds MFMA s4 of the group -- the k index is permuted alike in both operands
            const int chunk = grp * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int row = wm * (BMO / 2) + i * 32 + (lane & 31);
                const f32x4 v = ld4(sA + buf * BMO * LDT + row * LDT + ((chunk ^ ((row >> 4) & 3)) * 4));
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) a[s4][i] = v[s4];
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int row = wn * (BNC / 2) + j * 32 + (lane & 31);
                const f32x4 v = ld4(sB + buf * BNC * LDT + row * LDT + ((chunk ^ ((row >> 4) & 3)) * 4));
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) b[s4][j] = v[s4];
            }
            return;
        }
        const float* al = sA + buf * BKM * BMO + a_off + grp * 8 * BMO;
        const float* bl = sB + buf * BKM * BNC + b_off + grp * 8 * BNC;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int i = 0; i < TI; ++i) a[s4][i] = al[s4 * 2 * BMO + i * 32];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[s4][j] = bl[s4 * 2 * BNC + j * 32];
        }
    };
    auto mma16 = [&](const float (&a)[4][TI], const float (&b)[4][TJ]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s4][i], b[s4][j], acc[i][j], 0, 0, 0);
    };
    float fa[4][TI], fb[4][TJ];
    if (kt0 < kt1) {
        prep(kt0);
        load_tile(kt0);
        store_tile(0);
        prep(kt0 + 1 < kt1 ? kt0 + 1 : kt0);
    }
    __syncthreads();
    if (kt0 < kt1) read_frags(0, 0, fa, fb);
    // (the last k-tile runs in a peeled block without loads, LDS refill or barriers, as in the NT kernel)
    for (int kt = kt0; kt + 1 < kt1; ++kt) {
        const int buf = NBUF == 2 ? (kt - kt0) & 1 : 0;
        const int nbuf = NBUF == 2 ? buf ^ 1 : 0;
        load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        prep(kt + 2 < kt1 ? kt + 2 : kt);           // rows past M read zeros anyway; clamp keeps it branch-free
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            float na[4][TI], nb[4][TJ];
            read_frags(buf, grp + 1, na, nb);
            mma16(fa, fb);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                for (int i = 0; i < TI; ++i) fa[s4][i] = na[s4][i];
#pragma unroll
                for (int j = 0; j < TJ; ++j) fb[s4][j] = nb[s4][j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {              // the group-3 fragments are in registers BEFORE the barrier
#pragma unroll
            for (int i = 0; i < TI; ++i) pin(fa[s4][i]);
#pragma unroll
            for (int j = 0; j < TJ; ++j) pin(fb[s4][j]);
        }
        if (NBUF == 1) __syncthreads();        // every wave has read the last fragments of tile kt
        store_tile(nbuf);
        __syncthreads();
        float na[4][TI], nb[4][TJ];
        read_frags(nbuf, 0, na, nb);
        __builtin_amdgcn_sched_barrier(0);
        mma16(fa, fb);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[s4][i] = na[s4][i];
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[s4][j] = nb[s4][j];
        }
    }
    if (kt0 < kt1) {
        const int buf = NBUF == 2 ? (kt1 - 1 - kt0) & 1 : 0;
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            float na[4][TI], nb[4][TJ];
            read_frags(buf, grp + 1, na, nb);
            mma16(fa, fb);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                for (int i = 0; i < TI; ++i) fa[s4][i] = na[s4][i];
#pragma unroll
                for (int j = 0; j < TJ; ++j) fb[s4][j] = nb[s4][j];
            }
        }
        mma16(fa, fb);
    }

    // epilogue: rows = output channel o, cols = input channel (or flattened stem column)
    const int kwid = g.gw ? g.gw : g.Ci;          // filter row length per tap (grouped: the window)
    const size_t wrow = XK ? (size_t)stem_kp(g.wT, g.cr) : (size_t)g.wT * kwid;
    float* base = dst + (size_t)split * g.Co * wrow;
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
                    base[(size_t)o * wrow + (size_t)widx * kwid + (g.gw ? 0 : c0) + cl] = acc[i][j][r];
                }
            }
        }
}

// ------------------------------------------------------------------------------------------
// TN kernel on the bf16 MFMA (bf16 mode, every conv but the fp32 stem)
// ------------------------------------------------------------------------------------------
// dW[o][tap][c] = sum_m dY[m][o] * X[gather(m, tap)][c].  v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE values of the
// reduction index per lane, but m is the slow index of both operands in memory.  So the staging pass transposes:
// a thread fetches an 8(m) x 8(channel) block with eight 16-byte row loads, interleaves the 16-bit halves with 32
// v_perm_b32 (the dword-level part of the transpose is register renaming) and writes eight 16-byte LDS rows of an
// image [channel][64 m] -- byte for byte the image the NT kernel builds ([row][64 k], 36-word pitch), so the fragment
// reads, the MFMA stream and the software pipeline are the NT kernel's.  Thread -> block mapping: m-group fastest
// (8 lanes cover 8 m-groups of one channel group): the LDS writes of a wave then hit all 64 banks 4 lanes deep (the
// b128 minimum) and each global row still gets 128 contiguous bytes from 8 lanes.
// Grid: 1-D, XCD-remapped so that the blocks of one m-range (all taps / channel tiles of a split) share an L2.
// W8: 8 | Wo -- a thread's 8 rows are 8 consecutive wo of one output row: one row decode per k-tile instead of eight.
template <int BMO, int BNC, bool STEM, bool W8 = false, int NBUF = 2, int MINB = 1>
__global__ __launch_bounds__(kThreads, MINB) void conv_wgrad_bf16_kernel(IoConvGeom g, const bf16_t* __restrict__ in,
                                                                  const bf16_t* __restrict__ dy,
                                                                  float* __restrict__ dst, int ntile_c, int tiles,
                                                                  int kps, size_t in_bytes, size_t dy_bytes) {
    constexpr int BKM = 64, LDT = 36;
    constexpr int TI = BMO / 64, TJ = BNC / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                      // [2][BMO][LDT]
    float* sB = smem + NBUF * BMO * LDT;   // [NBUF][BNC][LDT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = g.Ho * g.Wo;
    const int M = g.N * HoWo;
    const int T = g.Th * g.Tw;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / tiles, tile = logical - split * tiles;
    // staging role (wave-uniform): threads [0, BMO) move dY blocks, [BMO, BMO + BNC) move X blocks
    const bool role_a = wave * 64 < BMO;
    const bool role_b = !role_a && wave * 64 < BMO + BNC;
    const int blk = role_a ? tid : tid - BMO;
    const int mg = blk & 7, cg = blk >> 3;

    // tile -> (o tile, tap, channel tile); stem: the columns are the flattened (tap, 8 channels) axis, so each
    // 8-column group of a thread is one tap of its own
    const int per_o = STEM ? ntile_c : T * ntile_c;
    const int ot = tile / per_o, rem0 = tile - ot * per_o;
    const int o0 = ot * BMO;
    int c0, dh, dw, widx, xcol;
    bool tapok = true;
    if (STEM) {
        c0 = rem0 * BNC;
        const int tp = (c0 >> 3) + cg;
        tapok = tp < g.wT;
        const int r = tp / g.S, sx = tp - r * g.S;
        dh = g.dh0 + g.dhs * r;
        dw = g.dw0 + g.dws * sx;
        widx = 0;
        xcol = 0;
    } else {
        const int tap = rem0 / ntile_c;
        c0 = g.gw ? o0 : (rem0 - tap * ntile_c) * BNC;      // grouped: only the diagonal (o tile == c tile)
        const int th = tap / g.Tw, tw = tap - th * g.Tw;
        dh = g.dh0 + g.dhs * th;
        dw = g.dw0 + g.dws * tw;
        widx = (g.r0 + g.rs * th) * g.S + (g.s0 + g.ss * tw);
        xcol = c0 + cg * 8;
    }

    const int nkt = (M + BKM - 1) / BKM;
    const int kt0 = split * kps;
    const int kt1 = min(kt0 + kps, nkt);

    const int mfirst = min(kt0 * BKM, M - 1);    // descriptors start at this split's first row / first sample
    const int n_lo = fdiv(mfirst, g.fd_howo);
    const int ipix_lo = n_lo * g.Hi * g.Wi;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc_at(in, (size_t)ipix_lo * (size_t)(g.Ci * 2), in_bytes);
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc_at(dy, (size_t)mfirst * (size_t)(g.Co * 2), dy_bytes);
    const bool lin = !STEM && T == 1 && g.is == 1 && g.dh0 == 0 && g.dw0 == 0 && g.Hi == g.Ho && g.Wi == g.Wo;
    u32x4 rr[8];
    auto load_tile = [&](int kt) {
        const int mrow = kt * BKM + mg * 8;
        if (role_a) {
#pragma unroll
            for (int r = 0; r < 8; ++r)     // rows past M lie past the end of dY -> zeros
                rr[r] = __builtin_amdgcn_raw_buffer_load_b128(
                    rs_dy, (unsigned)((mrow + r - mfirst) * g.Co + o0 + cg * 8) * 2u, 0, 0);
        } else if (role_b) {
            if constexpr (W8) {
                const bool ok0 = tapok && mrow < M;
                const int mm = ok0 ? mrow : 0;
                const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
                const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
                const int hi = ho * g.is + dh, wi0 = wo * g.is + dw;
                const bool okh = ok0 & ((unsigned)hi < (unsigned)g.Hi);
                const int base = (((n - n_lo) * g.Hi + hi) * g.Wi + wi0) * g.Ci + xcol;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    // `&`, not `&&`: the short-circuit form makes hipcc split the r = 0 load into an if / else pair
                    // of loads to the same registers with an s_waitcnt vmcnt(0) between them
                    const bool ok = okh & ((unsigned)(wi0 + r * g.is) < (unsigned)g.Wi);
                    rr[r] = __builtin_amdgcn_raw_buffer_load_b128(
                        rs_in, ok ? (unsigned)(base + r * g.is * g.Ci) * 2u : kInvalidOff, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int m = mrow + r;
                unsigned off;
                if (lin) {
                    off = (unsigned)((m - ipix_lo) * g.Ci + xcol) * 2u;
                } else {
                    bool ok = tapok && m < M;
                    const int mm = ok ? m : 0;
                    const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
                    const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
                    const int hi = ho * g.is + dh, wi = wo * g.is + dw;
                    ok = ok && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                    off = ok ? (unsigned)((((n - n_lo) * g.Hi + hi) * g.Wi + wi) * g.Ci + xcol) * 2u : kInvalidOff;
                }
                rr[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
            }
        }
    };
    auto store_tile = [&](int buf) {
        if (role_a || role_b) {
            float* base = (role_a ? sA + buf * BMO * LDT : sB + buf * BNC * LDT) + cg * 8 * LDT + mg * 4;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                // channel c of the block: halves (c & 1) of dword (c >> 1) of the eight rows, rows pairwise packed
                const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
                u32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_perm(rr[2 * j + 1][c >> 1], rr[2 * j][c >> 1], sel);
                *reinterpret_cast<u32x4*>(base + c * LDT) = v;
            }
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = (wm * (BMO / 2) + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int b_off = (wn * (BNC / 2) + (lane & 31)) * LDT + (lane >> 5) * 4;
    auto read_frags = [&](int buf, int kk, f32x4 (&a)[TI], f32x4 (&b)[TJ]) {
        const float* al = sA + buf * BMO * LDT + a_off + kk * 8;
        const float* bl = sB + buf * BNC * LDT + b_off + kk * 8;
#pragma unroll
        for (int i = 0; i < TI; ++i) a[i] = ld4(al + i * 32 * LDT);
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[j] = ld4(bl + j * 32 * LDT);
    };
    auto mma = [&](const f32x4 (&a)[TI], const f32x4 (&b)[TJ]) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                    __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    };

    f32x4 fa[TI], fb[TJ];
    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
    }
    __syncthreads();
    if (kt0 < kt1) read_frags(0, 0, fa, fb);
    for (int kt = kt0; kt + 1 < kt1; ++kt) {
        const int buf = NBUF == 2 ? (kt - kt0) & 1 : 0;
        const int nbuf = NBUF == 2 ? buf ^ 1 : 0;
        load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            f32x4 na[TI], nb[TJ];
            read_frags(buf, kk + 1, na, nb);
            mma(fa, fb);
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[i] = na[i];
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[j] = nb[j];
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) pin(fa[i]);      // the group-3 fragments are in registers BEFORE the barrier
#pragma unroll
        for (int j = 0; j < TJ; ++j) pin(fb[j]);
        if (NBUF == 1) __syncthreads();        // every wave has read the last fragments of tile kt
        store_tile(nbuf);
        __syncthreads();
        f32x4 na[TI], nb[TJ];
        read_frags(nbuf, 0, na, nb);
        mma(fa, fb);
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[i] = na[i];
#pragma unroll
        for (int j = 0; j < TJ; ++j) fb[j] = nb[j];
    }
    if (kt0 < kt1) {                           // last k-tile: peeled, nothing to fetch
        const int buf = NBUF == 2 ? (kt1 - 1 - kt0) & 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            f32x4 na[TI], nb[TJ];
            read_frags(buf, kk + 1, na, nb);
            mma(fa, fb);
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[i] = na[i];
#pragma unroll
            for (int j = 0; j < TJ; ++j) fb[j] = nb[j];
        }
        mma(fa, fb);
    }

    const int kwid = g.gw ? g.gw : g.Ci;
    const size_t wrow = (size_t)g.wT * kwid;
    float* base = dst + (size_t)split * g.Co * wrow;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wm * (BMO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int cl = wn * (BNC / 2) + j * 32 + (lane & 31);
                if (STEM) {
                    if (c0 + cl < (int)wrow) base[(size_t)o * wrow + c0 + cl] = acc[i][j][r];
                } else {
                    base[(size_t)o * wrow + (size_t)widx * kwid + (g.gw ? 0 : c0) + cl] = acc[i][j][r];
                }
            }
        }
}

// ------------------------------------------------------------------------------------------
// TN kernel on the bf16 MFMA, LDS-DMA + transpose-read form
// ------------------------------------------------------------------------------------------
// Same product as conv_wgrad_bf16_kernel, without its register staging: the two operands go from global memory
// STRAIGHT into LDS (`buffer_load_dwordx4 ... lds`, 16 bytes per lane, no VGPRs, no LDS store instructions) as
// row-major images [64 m][channels] -- exactly what memory holds -- and the MFMA fragments (8 consecutive m per lane)
// come out of `ds_read_b64_tr_b16`, gfx950's transposing LDS read: per 16-lane group, lane 4 j + q points at columns
// 4q..4q+3 of row j of a [4 m][16 channel] block and lane c receives column c of the four rows.  That removes the 32
// v_perm_b32 + 8 ds_write_b128 + the per-row gather arithmetic per k-tile and thread that made the staged kernel
// instruction-issue bound (profiles/r02_pmc_bf16_wgrad_3x3.txt).
//   * An LDS-DMA instruction writes 64 lanes x 16 B = 1 KiB contiguously (lane i -> M0 base + 16 i), a "chunk": 4 rows
//     of a 128-channel operand, 8 rows of a 64-channel one.  WHICH (row, 16-byte channel slot) a lane fetches is free,
//     so the image is swizzled on the way in (tr_chunk_pos): the four rows a transpose-read touches, 64 bytes each, then
//     sit in four different 64-byte bank groups -- conflict-free (tools/tr_probe.hip measures it on the hardware).
//   * Addresses: the lane part of a fetch (in-chunk row, channel slot) is constant for the whole kernel (voffset), the
//     k-tile / chunk part is wave-uniform and goes into the scalar offset -- computed on the scalar unit, zero vector
//     instructions for dY, three (the left / right image border) for X.  soffset is not range-checked and must not be
//     negative: the X descriptor starts `padpx` pixels early; border lanes get an out-of-range voffset = zeros in LDS.
//   * One barrier per k-tile: wait for the own fetches of tile kt, barrier (tile kt complete, every wave done with
//     tile kt - 1), request tile kt + 1 into the other stage, multiply tile kt.
// Shapes: 64 | Ho*Wo (a k-tile lies in one sample), rows-per-chunk | Wo (a chunk lies in one output row), 64 | M; the
// launcher sends everything else (and the stem) to the staged kernel.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char* lds_cptr;

// One LDS-DMA instruction: 16 bytes per lane from descriptor `rs` at voff (per lane, range-checked) + soff (uniform,
// not range-checked) to LDS byte address lds_addr + 16 * lane.  Inline asm, NOT __builtin_amdgcn_raw_ptr_buffer_load_lds:
// hipcc tracks the builtin as a pending LDS write and puts s_waitcnt vmcnt(0) in front of the next LDS read that may
// alias it -- every read of the OTHER stage -- which serialises fetch and multiply.  The asm form is invisible to that
// pass, so the one wait it needs (before the barrier that publishes the tile) is written by hand: dma_wait_all().
__device__ __forceinline__ u32x4 dma_rsrc(const void* p, size_t bytes) {
    const unsigned long long a = (unsigned long long)p;
    const u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes,
                     0x00020000u};
    return r;
}
__device__ __forceinline__ void dma16(u32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr)
                 : "memory");          // (m0 is reserved: hipcc never keeps a value in it across statements)
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 16-byte slot of (in-chunk row r, channel slot cs) inside a chunk of a W-channel image
template <int W> __device__ __forceinline__ int tr_chunk_pos(int r, int cs) {
    if (W == 128) return r * 16 + ((cs + 4 * r) & 15);                     // 4 rows x 16 slots, row r rotated by 4 r
    return (2 * r + ((cs >> 2) ^ ((r >> 1) & 1))) * 4 + (cs & 3);           // 8 rows x 2 halves, halves swapped in rows 2,3,6,7
}
// ... and its inverse: which (row, slot) DMA lane `pos` fetches
template <int W> __device__ __forceinline__ void tr_chunk_src(int pos, int& r, int& cs) {
    if (W == 128) {
        r = pos >> 4;
        cs = ((pos & 15) - 4 * r) & 15;
    } else {
        const int P = pos >> 2;
        r = P >> 1;
        cs = (((P & 1) ^ ((r >> 1) & 1)) << 2) + (pos & 3);
    }
}

// STAGES: LDS stages.  With 2, one k-tile is in flight per block and every k-tile pays a full memory latency (the launch of
// layer 3's 1024 -> 256 gradient: 64 k-tiles x 2 us = its 0.12 ms); with more, STAGES - 1 k-tiles are in flight and the wait
// in front of the barrier is for the OLDEST of them only (counted vmcnt).
template <int N> __device__ __forceinline__ void dma_wait_left() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int BMO, int BNC, bool STEM = false, int MINB = 2, int BKM = 64, int STAGES = 2>
__global__ __launch_bounds__(kThreads, MINB) void conv_wgrad_bf16_tr_kernel(IoConvGeom g, const bf16_t* __restrict__ in,
                                                                     const bf16_t* __restrict__ dy,
                                                                     float* __restrict__ dst, int ntile_c, int tiles,
                                                                     int kps, size_t in_bytes, size_t dy_bytes) {
    constexpr int TI = BMO / 64, TJ = BNC / 64;
    constexpr int A_BYTES = BMO * BKM * 2, B_BYTES = BNC * BKM * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int RA = 512 / BMO, RB = 512 / BNC;              // rows per 1 KiB chunk
    constexpr int CA = BKM / RA / 4, CB = BKM / RB / 4;        // chunks per wave and k-tile
    constexpr int KKA = BMO == 128 ? 4096 : 2048, HA = BMO == 128 ? 1024 : 512;   // bytes per 16-row k-step / 4-row half
    constexpr int KKB = BNC == 128 ? 4096 : 2048, HB = BNC == 128 ? 1024 : 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const lds_cptr lds = (lds_cptr)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = g.Ho * g.Wo;
    const int M = g.N * HoWo;
    const int T = g.Th * g.Tw;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / tiles, tile = logical - split * tiles;
    // tile -> (o tile, tap, channel tile); stem: the columns are the flattened (tap, 8 channels) axis of the packed
    // input, a 16-byte slot of the X image is one tap of one pixel, so the tap is a LANE constant there
    const int per_o = STEM ? ntile_c : T * ntile_c;
    const int ot = tile / per_o, rem0 = tile - ot * per_o;
    const int o0 = ot * BMO;
    const int tap = STEM ? 0 : rem0 / ntile_c;
    const int c0 = STEM ? rem0 * BNC : g.gw ? o0 : (rem0 - tap * ntile_c) * BNC;   // grouped: only the diagonal tiles
    const int th = tap / g.Tw, tw = tap - th * g.Tw;
    const int widx = STEM ? 0 : (g.r0 + g.rs * th) * g.S + (g.s0 + g.ss * tw);

    const int nkt = M / BKM;
    const int kt0 = split * kps;
    const int kt1 = min(kt0 + kps, nkt);
    const int mfirst = kt0 * BKM;                // descriptors start at this split's first row / first sample
    const int n_lo = fdiv(mfirst, g.fd_howo);
    // the tap shift (dh, dw) lives in the lane offset, which must not be negative: the X descriptor starts `padh` rows
    // and `padw` pixels before the split's first sample (address arithmetic only -- lanes that would read there are
    // border lanes and get an out-of-range offset)
    const int padh = g.dh0 < 0 ? -g.dh0 : 0, padw = g.dw0 < 0 ? -g.dw0 : 0;
    u32x4 rs_dy, rs_in;
    {
        const size_t abase = (size_t)mfirst * (size_t)(g.Co * 2);
        rs_dy = dma_rsrc(reinterpret_cast<const char*>(dy) + abase, dy_bytes > abase ? dy_bytes - abase : 0);
        const size_t base = (size_t)n_lo * (size_t)(g.Hi * g.Wi) * (size_t)(g.Ci * 2);
        const size_t pad = (size_t)(padh * g.Wi + padw) * (size_t)(g.Ci * 2);
        rs_in = dma_rsrc(reinterpret_cast<const char*>(in) + base - pad, (in_bytes > base ? in_bytes - base : 0) + pad);
    }
    const unsigned lds0 = (unsigned)(size_t)lds;       // LDS byte address of the dynamic region

    // lane constants of the fetches
    int ra_, csa, rb_, csb;
    tr_chunk_src<BMO>(lane, ra_, csa);
    tr_chunk_src<BNC>(lane, rb_, csb);
    int dh, dw;
    bool tapok = true;
    if (STEM) {
        const int tp = (c0 >> 3) + csb;
        tapok = tp < g.wT;
        const int r = tp / g.S, sx = tp - r * g.S;
        dh = g.dh0 + g.dhs * r;
        dw = g.dw0 + g.dws * sx;
    } else {
        dh = g.dh0 + g.dhs * th;
        dw = g.dw0 + g.dws * tw;
    }
    const unsigned va = (unsigned)(ra_ * g.Co + o0 + csa * 8) * 2u;
    const unsigned vb =
        (unsigned)(((dh + padh) * g.Wi + rb_ * g.is + dw + padw) * g.Ci + (STEM ? 0 : c0 + csb * 8)) * 2u;
    const int wib = rb_ * g.is + dw;             // + wo * is = input column of the lane's row

    // Scalar side of the fetches.  The wave's chunks sit at fixed pixel offsets q inside a k-tile (64 consecutive output
    // pixels of one sample, starting at a multiple of 64).  When Wo divides 64 or 64 divides Wo (every power-of-two map)
    // a chunk's (row, column) is the tile's plus a per-chunk CONSTANT, so a k-tile costs two divisions and each fetch one
    // scalar add; other widths decode every chunk (the general path below).  This matters: with the decode per chunk the
    // kernel issued 7.4 scalar instructions per MFMA and spent 27 % of its wave cycles on them.
    const bool fastrow = (g.Wo % BKM == 0) || (BKM % g.Wo == 0);
    const bool wo_small = BKM % g.Wo == 0;
    int dho[CB], wou[CB];
    unsigned cso[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
        const int q = (wave * CB + u) * RB;
        dho[u] = wo_small ? q / g.Wo : 0;
        wou[u] = wo_small ? q - dho[u] * g.Wo : q;
        cso[u] = (unsigned)((dho[u] * g.is * g.Wi + wou[u] * g.is) * g.Ci) * 2u;
    }
    auto issue = [&](int kt, int stage) {
        const unsigned sb = lds0 + (unsigned)(stage * STAGE);
        const int m0 = kt * BKM;
        const unsigned asoff = (unsigned)((m0 - mfirst) * g.Co) * 2u;
#pragma unroll
        for (int u = 0; u < CA; ++u) {
            const int chunk = wave * CA + u;
            dma16(rs_dy, sb + (unsigned)(chunk * 1024), va, asoff + (unsigned)(chunk * RA * g.Co) * 2u);
        }
        const int n = fdiv(m0, g.fd_howo), p_tile = m0 - n * HoWo;
        if (fastrow) {
            const int ho_t = fdiv(p_tile, g.fd_wo), wo_t = p_tile - ho_t * g.Wo;
            const unsigned tsoff = (unsigned)((((n - n_lo) * g.Hi + ho_t * g.is) * g.Wi + wo_t * g.is) * g.Ci) * 2u;
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int chunk = wave * CB + u;
                const int ho = ho_t + dho[u], wo = wo_t + wou[u];
                const bool ok = tapok & ((unsigned)(ho * g.is + dh) < (unsigned)g.Hi) &
                                ((unsigned)(wo * g.is + wib) < (unsigned)g.Wi);
                dma16(rs_in, sb + (unsigned)(A_BYTES + chunk * 1024), ok ? vb : kInvalidOff, tsoff + cso[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int chunk = wave * CB + u;
                const int pc = p_tile + chunk * RB;
                const int ho = fdiv(pc, g.fd_wo), wo = pc - ho * g.Wo;
                const unsigned soff = (unsigned)((((n - n_lo) * g.Hi + ho * g.is) * g.Wi + wo * g.is) * g.Ci) * 2u;
                const bool ok = tapok & ((unsigned)(ho * g.is + dh) < (unsigned)g.Hi) &
                                ((unsigned)(wo * g.is + wib) < (unsigned)g.Wi);
                dma16(rs_in, sb + (unsigned)(A_BYTES + chunk * 1024), ok ? vb : kInvalidOff, soff);
            }
        }
    };

    // lane constants of the fragment reads
    const int g4 = lane >> 4, fj = (lane >> 2) & 3, fq = lane & 3;
    unsigned fa[TI], fb[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int c = wm * (BMO / 2) + i * 32 + (g4 & 1) * 16 + 4 * fq;
        fa[i] = (unsigned)((g4 >> 1) * (BMO == 128 ? 2048 : 1024) + tr_chunk_pos<BMO>(fj, c >> 3) * 16 + (c & 7) * 2);
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int c = wn * (BNC / 2) + j * 32 + (g4 & 1) * 16 + 4 * fq;
        fb[j] = (unsigned)(A_BYTES + (g4 >> 1) * (BNC == 128 ? 2048 : 1024) + tr_chunk_pos<BNC>(fj, c >> 3) * 16 + (c & 7) * 2);
    }
    auto frag = [&](lds_cptr sb, unsigned off, int half_bytes) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sb + off));
        const s16x4 hi =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sb + off + half_bytes));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int DPT = CA + CB;         // DMA instructions per wave and k-tile
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0)
        if (kt0 + s0 < kt1) issue(kt0 + s0, s0);
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        // the own fetches of tile kt have landed when at most the younger STAGES - 2 tiles are outstanding ...
        if (STAGES > 2 && kt + STAGES - 2 < kt1) dma_wait_left<DPT*(STAGES > 2 ? STAGES - 2 : 0)>();
        else dma_wait_all();
        __syncthreads();                 // ... everybody's have, and every wave is done reading the stage refilled next
        if (kt + STAGES - 1 < kt1) issue(kt + STAGES - 1, stage == 0 ? STAGES - 1 : stage - 1);
        const lds_cptr sb = lds + stage * STAGE;
        stage = stage + 1 == STAGES ? 0 : stage + 1;
#pragma unroll
        for (int kk = 0; kk < BKM / 16; ++kk) {
            bf16x8 a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = frag(sb, fa[i] + kk * KKA, HA);
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = frag(sb, fb[j] + kk * KKB, HB);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    const int kwid = g.gw ? g.gw : g.Ci;
    const size_t wrow = (size_t)g.wT * kwid;
    float* base = dst + (size_t)split * g.Co * wrow;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wm * (BMO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int cl = wn * (BNC / 2) + j * 32 + (lane & 31);
                if (STEM) {
                    if (c0 + cl < (int)wrow) base[(size_t)o * wrow + c0 + cl] = acc[i][j][r];
                } else {
                    base[(size_t)o * wrow + (size_t)widx * kwid + (g.gw ? 0 : c0) + cl] = acc[i][j][r];
                }
            }
        }
}


// ------------------------------------------------------------------------------------------
// TN kernel, Winograd F(2, 3) row form (fp32; 3x3 stride-1 same-size convolutions, 8 | Wo, 64 | M)
// ------------------------------------------------------------------------------------------
// The transpose of the forward form of conv_nt_kernel<..., WINO>: for a pair of horizontally adjacent output pixels (w, w+1)
// and one filter row r, the three taps dW[r][0..2] += (dy0 d0 + dy1 d1, dy0 d1 + dy1 d2, dy0 d2 + dy1 d3) (d0..d3 = the
// inputs at columns w - 1 .. w + 2 of input row h + r - 1) cost 4 products instead of 6:
//     Y = (dy0, dy0 + dy1, dy0 - dy1, -dy1)        V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
//     P_f = sum over all pairs of Y_f * V_f                       (f = 0..3: four independent GEMMs, reduction index = pair)
//     dW[r][0] = P0 + (P1 + P2) / 2      dW[r][1] = (P1 - P2) / 2      dW[r][2] = (P1 + P2) / 2 + P3
// A block owns (output-channel tile of 64, filter row r, input-channel tile of 64, split of the pairs); a wave owns 32 x 32
// channels x the four frequencies -- four 32x32 accumulators whose lanes hold the same (o, c) entry, so the combination above
// is register arithmetic in the epilogue and the block writes its three taps straight into the [Co][9][Ci] partial.
// Staging: a k-tile is 32 pairs = 64 consecutive output pixels.  Threads 0..127 stage dY -- a thread fetches 8 consecutive
// pixels (4 pairs) of a 4-channel chunk, forms Y and writes it transposed ([f][channel][4 pairs]: register renaming, as in the
// TR form above) -- threads 128..255 stage the input the same way from the 10 pixels w - 1 .. w + 8 of row h + r - 1 (8 | Wo:
// a thread's pixels lie in one image row, one row decode per k-tile).  LDS image [f][64 channels][32 pairs], 36-word pitch,
// chunk index XORed with channel bits 4..5 (conflict-free transposed writes and fragment reads, see conv_wgrad_kernel).
template <int MINB = 2>
__global__ __launch_bounds__(kThreads, MINB) void conv_wgrad_wino_kernel(IoConvGeom g, const float* __restrict__ in,
                                                                  const float* __restrict__ dy,
                                                                  float* __restrict__ dst, int ntile_c, int tiles,
                                                                  int kps, size_t in_bytes, size_t dy_bytes) {
    constexpr int BC = 64, BKP = 32, LDT = 36;          // channels per operand tile, pairs per k-tile, LDS pitch
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                   // [4 f][64 o][LDT]
    float* sB = smem + 4 * BC * LDT;                    // [4 f][64 c][LDT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = g.Ho * g.Wo;
    const int M = g.N * HoWo;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / tiles, tile = logical - split * tiles;
    const int per_o = 3 * ntile_c;
    const int ot = tile / per_o, rem0 = tile - ot * per_o;
    const int o0 = ot * BC;
    const int fr = rem0 / ntile_c;                      // filter row of this block
    const int c0 = (rem0 - fr * ntile_c) * BC;
    const int dh = g.dh0 + g.dhs * fr;                  // input row offset of that filter row (forward geometry: fr - 1)

    const int nkt = M / 64;
    const int kt0 = split * kps;
    const int kt1 = min(kt0 + kps, nkt);
    const int mfirst = min(kt0 * 64, M - 1);
    const int n_lo = fdiv(mfirst, g.fd_howo);
    const int ipix_lo = n_lo * g.Hi * g.Wi;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc_at(in, (size_t)ipix_lo * (size_t)(g.Ci * 4), in_bytes);
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc_at(dy, (size_t)mfirst * (size_t)(g.Co * 4), dy_bytes);

    // staging roles (wave-uniform): waves 0, 1 move dY, waves 2, 3 move the input
    const bool role_a = wave < 2;
    const int t7 = tid & 127;
    const int q4 = t7 & 15, pg = t7 >> 4;               // 4-channel chunk, pair-quad (8 pixels) of the k-tile
    f32x4 px[10];                                       // dY: 8 pixels; input: the 10 pixels w - 1 .. w + 8
    auto load_tile = [&](int kt) {
        const int m = kt * 64 + 8 * pg;                 // first pixel of this thread's pair-quad
        if (role_a) {
#pragma unroll
            for (int i = 0; i < 8; ++i)                 // rows past M fall off the descriptor -> zeros
                px[i] = bld4(rs_dy, (unsigned)((m + i - mfirst) * g.Co + o0 + q4 * 4) * 4u);
            px[8] = px[9] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const bool ok0 = m < M;
            const int mm = ok0 ? m : 0;
            const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
            const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
            const int hi = ho + dh;
            const bool okh = ok0 & ((unsigned)hi < (unsigned)g.Hi);
            // (unsigned wrap-around for the pixel at w - 1: the offset is only used when that pixel exists)
            const unsigned base = (unsigned)(((((n - n_lo) * g.Hi + hi) * g.Wi + wo - 1) * g.Ci + c0 + q4 * 4) * 4);
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const bool ok = okh & (i == 0 ? wo > 0 : i == 9 ? wo + 8 < g.Wi : true);
                px[i] = bld4(rs_in, ok ? base + (unsigned)(i * g.Ci * 4) : kInvalidOff);
            }
        }
    };
    auto store_tile = [&]() {
        float* base = (role_a ? sA : sB) + (q4 * 4) * LDT + ((pg ^ ((q4 >> 2) & 3)) * 4);
        f32x4 v[4][4];                                  // [f][pair]
        if (role_a) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 y0 = px[2 * t], y1 = px[2 * t + 1];
                v[0][t] = y0;
                v[1][t] = y0 + y1;
                v[2][t] = y0 - y1;
                v[3][t] = -y1;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 d0 = px[2 * t], d1 = px[2 * t + 1], d2 = px[2 * t + 2], d3 = px[2 * t + 3];
                v[0][t] = d0 - d2;
                v[1][t] = d1 + d2;
                v[2][t] = d2 - d1;
                v[3][t] = d1 - d3;
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) {               // channel i of the chunk: its 4 pairs as one 16-byte row segment
                const f32x4 row = {v[f][0][i], v[f][1][i], v[f][2][i], v[f][3][i]};
                st4(base + (f * BC + i) * LDT, row);
            }
    };

    f32x16 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31);
    auto read_frags = [&](int grp, f32x4 (&a)[4], f32x4 (&b)[4]) {
        const int chunk = grp * 2 + (lane >> 5);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            a[f] = ld4(sA + (f * BC + arow) * LDT + ((chunk ^ ((arow >> 4) & 3)) * 4));
            b[f] = ld4(sB + (f * BC + brow) * LDT + ((chunk ^ ((brow >> 4) & 3)) * 4));
        }
    };
    auto mma16 = [&](const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f][tt], b[f][tt], acc[f], 0, 0, 0);
    };

    f32x4 fa[4], fb[4];
    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile();
    }
    __syncthreads();
    if (kt0 < kt1) read_frags(0, fa, fb);
    for (int kt = kt0; kt + 1 < kt1; ++kt) {
        load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            f32x4 na[4], nb[4];
            read_frags(grp + 1, na, nb);
            mma16(fa, fb);
#pragma unroll
            for (int f = 0; f < 4; ++f) { fa[f] = na[f]; fb[f] = nb[f]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 4; ++f) {                 // the group-3 fragments are in registers BEFORE the barrier
            pin(fa[f]);
            pin(fb[f]);
        }
        __syncthreads();                       // every wave has read the last fragments of tile kt
        store_tile();
        __syncthreads();
        f32x4 na[4], nb[4];
        read_frags(0, na, nb);
        __builtin_amdgcn_sched_barrier(0);
        mma16(fa, fb);
#pragma unroll
        for (int f = 0; f < 4; ++f) { fa[f] = na[f]; fb[f] = nb[f]; }
    }
    if (kt0 < kt1) {
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            f32x4 na[4], nb[4];
            read_frags(grp + 1, na, nb);
            mma16(fa, fb);
#pragma unroll
            for (int f = 0; f < 4; ++f) { fa[f] = na[f]; fb[f] = nb[f]; }
        }
        mma16(fa, fb);
    }

    // epilogue: rows = output channel, cols = input channel; the three taps of filter row fr
    const size_t wrow = (size_t)g.wT * g.Ci;
    float* base = dst + (size_t)split * g.Co * wrow + (size_t)(g.r0 + g.rs * fr) * g.S * g.Ci + c0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float h12 = 0.5f * (acc[1][r] + acc[2][r]);
        float* p = base + (size_t)o * wrow;
        p[0] = acc[0][r] + h12;
        p[g.Ci] = 0.5f * (acc[1][r] - acc[2][r]);
        p[2 * g.Ci] = h12 + acc[3][r];
    }
}

// ------------------------------------------------------------------------------------------
// TN kernel, Winograd F(4, 3) row form (fp32; 3x3 stride-1 same-size convolutions, 16 | Wo, 64 | M)
// ------------------------------------------------------------------------------------------
// The transpose of conv_wino4_kernel, as conv_wgrad_wino_kernel is of the F(2, 3) form: four horizontally adjacent output
// pixels and one filter row give the three taps dW[r][s] += sum_i dy_i d_(i+s) (d0..d5 = inputs at columns w - 1 .. w + 4 of
// input row h + r - 1) with SIX products instead of twelve:
//     Y = (dy0,  dy0 + dy1 + dy2 + dy3,  dy0 - dy1 + dy2 - dy3,  dy0 + 2 dy1 + 4 dy2 + 8 dy3,  dy0 - 2 dy1 + 4 dy2 - 8 dy3,  dy3)
//     V = as in conv_wino4_kernel          P_f = sum over all quads of Y_f * V_f       (six GEMMs, reduction index = quad)
//     dW[r][0] = P0 / 4 - (P1 + P2) / 6 + (P3 + P4) / 24      dW[r][1] = (P2 - P1) / 6 + (P3 - P4) / 12
//     dW[r][2] = -(P1 + P2) / 6 + (P3 + P4) / 6 + P5
// Block = (64 output channels, filter row, 64 input channels, split of the quads); wave = 32 x 32 channels x six frequencies
// (96 accumulator registers), combination in registers in the epilogue.  k-tile = 16 quads = 64 consecutive output pixels;
// LDS image [f][64 channels][16 quads] per operand (48 KB, the layout and swizzle of conv_wino4_kernel).  Staging: threads
// 0..127 take dY, 128..255 the input; a thread owns TWO channels (8-byte loads) and 4 quads: 16 (18) pixels of one image row.
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_wino4_kernel(IoConvGeom g, const float* __restrict__ in,
                                                                const float* __restrict__ dy,
                                                                float* __restrict__ dst, int ntile_c, int tiles,
                                                                int kps, size_t in_bytes, size_t dy_bytes) {
    constexpr int BC = 64, LDT = 16, NF = 6;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                   // [6 f][64 o][16 quads]
    float* sB = smem + NF * BC * LDT;                   // [6 f][64 c][16 quads]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = g.Ho * g.Wo;
    const int M = g.N * HoWo;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / tiles, tile = logical - split * tiles;
    const int per_o = 3 * ntile_c;
    const int ot = tile / per_o, rem0 = tile - ot * per_o;
    const int o0 = ot * BC;
    const int fr = rem0 / ntile_c;
    const int c0 = (rem0 - fr * ntile_c) * BC;
    const int dh = g.dh0 + g.dhs * fr;

    const int nkt = M / 64;
    const int kt0 = split * kps;
    const int kt1 = min(kt0 + kps, nkt);
    const int mfirst = min(kt0 * 64, M - 1);
    const int n_lo = fdiv(mfirst, g.fd_howo);
    const int ipix_lo = n_lo * g.Hi * g.Wi;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc_at(in, (size_t)ipix_lo * (size_t)(g.Ci * 4), in_bytes);
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc_at(dy, (size_t)mfirst * (size_t)(g.Co * 4), dy_bytes);

    const bool role_a = wave < 2;
    const int t7 = tid & 127;
    const int cp = t7 >> 2, qg = t7 & 3;                // channel pair, quad group (16 pixels) of the k-tile
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto ld2 = [&](__amdgpu_buffer_rsrc_t r, unsigned off) -> f32x2 {
        return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    };
    f32x2 px[18];
    auto load_tile = [&](int kt) {
        const int m = kt * 64 + 16 * qg;
        if (role_a) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                px[i] = ld2(rs_dy, (unsigned)((m + i - mfirst) * g.Co + o0 + cp * 2) * 4u);
            px[16] = px[17] = f32x2{0.f, 0.f};
        } else {
            const bool ok0 = m < M;
            const int mm = ok0 ? m : 0;
            const int n = fdiv(mm, g.fd_howo), rem = mm - n * HoWo;
            const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
            const int hi = ho + dh;
            const bool okh = ok0 & ((unsigned)hi < (unsigned)g.Hi);
            const unsigned base = (unsigned)(((((n - n_lo) * g.Hi + hi) * g.Wi + wo - 1) * g.Ci + c0 + cp * 2) * 4);
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const bool ok = okh & (i == 0 ? wo > 0 : i == 17 ? wo + 16 < g.Wi : true);
                px[i] = ld2(rs_in, ok ? base + (unsigned)(i * g.Ci * 4) : kInvalidOff);
            }
        }
    };
    auto store_tile = [&]() {
        f32x2 v[NF][4];                                 // [f][quad]
        if (role_a) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x2 y0 = px[4 * t], y1 = px[4 * t + 1], y2 = px[4 * t + 2], y3 = px[4 * t + 3];
                const f32x2 e = y0 + y2, o = y1 + y3, e4 = y0 + 4.f * y2, o4 = y1 + 4.f * y3;
                v[0][t] = y0;
                v[1][t] = e + o;
                v[2][t] = e - o;
                v[3][t] = e4 + 2.f * o4;
                v[4][t] = e4 - 2.f * o4;
                v[5][t] = y3;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x2 d0 = px[4 * t], d1 = px[4 * t + 1], d2 = px[4 * t + 2], d3 = px[4 * t + 3], d4 = px[4 * t + 4],
                            d5 = px[4 * t + 5];
                const f32x2 e42 = d4 - 4.f * d2, e31 = d3 - 4.f * d1, f42 = d4 - d2, f31 = d3 - d1;
                v[0][t] = 4.f * d0 - 5.f * d2 + d4;
                v[1][t] = e42 + e31;
                v[2][t] = e42 - e31;
                v[3][t] = f42 + 2.f * f31;
                v[4][t] = f42 - 2.f * f31;
                v[5][t] = 4.f * d1 - 5.f * d3 + d5;
            }
        }
        float* base = role_a ? sA : sB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = cp * 2 + j;
            float* p = base + row * LDT + ((qg ^ ((row >> 2) & 3)) * 4);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const f32x4 q = {v[f][0][j], v[f][1][j], v[f][2][j], v[f][3][j]};
                st4(p + f * BC * LDT, q);
            }
        }
    };

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    const int frow = lane & 31;
    const int a_base = (wm * 32 + frow) * LDT, b_base = (wn * 32 + frow) * LDT;
    const int fsw = (frow >> 2) & 3;
    auto read_frags = [&](int gi, f32x4 (&a)[3], f32x4 (&b)[3]) {
        const int chunk = ((gi >> 1) * 2 + (lane >> 5)) ^ fsw;
        const int f0 = (gi & 1) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a[j] = ld4(sA + (f0 + j) * BC * LDT + a_base + chunk * 4);
            b[j] = ld4(sB + (f0 + j) * BC * LDT + b_base + chunk * 4);
        }
    };
    auto mma12 = [&](int gi, const f32x4 (&a)[3], const f32x4 (&b)[3]) {
        const int f0 = (gi & 1) * 3;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[f0 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][tt], b[j][tt], acc[f0 + j], 0, 0, 0);
    };

    f32x4 fa[3], fb[3];
    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile();
    }
    __syncthreads();
    if (kt0 < kt1) read_frags(0, fa, fb);
    for (int kt = kt0; kt + 1 < kt1; ++kt) {
        load_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            f32x4 na[3], nb[3];
            read_frags(gi + 1, na, nb);
            mma12(gi, fa, fb);
#pragma unroll
            for (int j = 0; j < 3; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) { pin(fa[j]); pin(fb[j]); }      // group-3 fragments in registers BEFORE the barrier
        __syncthreads();
        store_tile();
        __syncthreads();
        f32x4 na[3], nb[3];
        read_frags(0, na, nb);
        __builtin_amdgcn_sched_barrier(0);
        mma12(3, fa, fb);
#pragma unroll
        for (int j = 0; j < 3; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
    }
    if (kt0 < kt1) {
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            f32x4 na[3], nb[3];
            read_frags(gi + 1, na, nb);
            mma12(gi, fa, fb);
#pragma unroll
            for (int j = 0; j < 3; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
        }
        mma12(3, fa, fb);
    }

    const size_t wrow = (size_t)g.wT * g.Ci;
    float* base = dst + (size_t)split * g.Co * wrow + (size_t)(g.r0 + g.rs * fr) * g.S * g.Ci + c0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float s12 = acc[1][r] + acc[2][r], s34 = acc[3][r] + acc[4][r];
        float* p = base + (size_t)o * wrow;
        p[0] = 0.25f * acc[0][r] - (1.0f / 6.0f) * s12 + (1.0f / 24.0f) * s34;
        p[g.Ci] = (1.0f / 6.0f) * (acc[2][r] - acc[1][r]) + (1.0f / 12.0f) * (acc[3][r] - acc[4][r]);
        p[2 * g.Ci] = (1.0f / 6.0f) * (s34 - s12) + acc[5][r];
    }
}

// (An fp32 sibling -- LDS-DMA into a row-major [m][channels] image, ds_read_b32 fragments, no transposes at all because
// v_mfma_f32_32x32x2_f32 takes one reduction index per lane -- was built and measured: 112.9 TF/s with 32-row k-tiles at
// two blocks per CU, 117.8 with 16-row k-tiles at four, against 118.3 for the register-staged kernel at three; the
// fp32 filter gradient is bound by the matrix pipe, not by its staging, so it stays on conv_wgrad_kernel.)


// ------------------------------------------------------------------------------------------
// NT kernel, Winograd F(4, 3) row form (fp32; 3x3 stride-1 same-size convolutions / data gradients, 4 | Wo, 256 | M)
// ------------------------------------------------------------------------------------------
// The next member of the family of conv_nt_kernel<..., WINO> (F(2, 3): 4 products per 2 outputs and filter row): FOUR
// horizontally adjacent outputs from the six inputs d0..d5 at columns w - 1 .. w + 4 with SIX products -- half the MFMAs
// of the direct form.  With g0..g2 the filter row (g_j meets input column w + j - 1):
//     V = (4 d0 - 5 d2 + d4,  -4 d1 - 4 d2 + d3 + d4,  4 d1 - 4 d2 - d3 + d4,  -2 d1 - d2 + 2 d3 + d4,
//          2 d1 - d2 - 2 d3 + d4,  4 d1 - 5 d3 + d5)
//     U = (g0 / 4,  -(g0 + g1 + g2) / 6,  -(g0 - g1 + g2) / 6,  (g0 + 2 g1 + 4 g2) / 24,  (g0 - 2 g1 + 4 g2) / 24,  g2)
//     m_f = sum over filter rows and input channels of V_f * U_f        (f = 0..5: six independent GEMMs)
//     y0 = m0 + m1 + m2 + m3 + m4     y1 = m1 - m2 + 2 (m3 - m4)     y2 = m1 + m2 + 4 (m3 + m4)     y3 = m1 - m2 + 8 (m3 - m4) + m5
// (Lavin & Gray's matrices; in fp32 the form is about eight times the rounding error of the direct product -- 6e-7 of
// the output scale at K = 576 -- three orders inside the 1e-3 bar.)
// Block tile: 256 output pixels (64 quads) x 64 output channels; wave (wm, wn) owns quads [32 wm, 32 wm + 32) = the 128
// pixels of ONE statistics tile x channels [32 wn, 32 wn + 32) x the six frequencies: six 32x32 accumulators (96
// registers) whose lanes hold the same (quad, channel) entry, so the output transform is register arithmetic and the
// per-128-row reductions of the epilogues (BatchNorm statistics, BatchNorm-backward sums) never leave the wave.
// A k-tile is one filter row x 16 input channels (K = 3 Ci per frequency): LDS image 6 x 64 rows of V + 6 x 64 rows of U,
// 64 bytes per row, chunk index XORed with row bits 2..3 (conflict-free 16-byte stores and fragment reads), 48 KB, one
// buffer, two blocks per CU.  Fragments are prefetched one 12-MFMA group (three frequencies x 8 k) ahead.
// XF: the A operand goes through relu(bn(x)) while it is staged (IoBwStats::in_scale, as in conv_nt_kernel).
// BWE: the fused BatchNorm-backward epilogue (IoBwStats::y ...: ReLU mask recomputed from y, per-tile sums, activation side
// output).  Both need whole 256-row tiles per BatchNorm group.
// HALO (a tile = whole image rows: 256 % Wo == 0, Wo <= 64, and either 256 | Ho Wo -- rows of ONE sample -- or whole small
// samples whose halo images fit 96 quad rows): the three filter rows of a tile read
// the input rows h - 1, h, h + 1 of its R = 256 / Wo output rows, i.e. R + 2 distinct input rows -- so the k loop runs
// channel chunk OUTER, filter row INNER, the V image of all R + 2 rows ((R + 2) Wo / 4 <= 96 quad rows per frequency) is
// staged ONCE per channel chunk and the MFMAs of filter row r read it at an offset of (1 + dh) Wo / 4 quad rows; only U is
// restaged per filter row.  The operand transforms (relu(bn(x)) + V: the largest part of the staging cost, see the ablations
// in profiles/r04_pmc_wino4_vs_direct.txt) and the A loads drop by 3 R / (R + 2) = 2 .. 2.7x.
template <bool BWE, bool XF, bool HALO = false>
__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(IoConvGeom g, const float* __restrict__ in,
                                                        const float* __restrict__ U, float* __restrict__ out, int ntn,
                                                        size_t in_bytes, unsigned u_bytes, size_t out_bytes,
                                                        float* __restrict__ st_mean, float* __restrict__ st_m2,
                                                        IoBwStats bw) {
    constexpr int BM = 256, BN = 64, BK = 16, LDT = 16, NF = 6;
    constexpr int AQ = HALO ? 96 : 64;      // quad rows of the A image per frequency
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                       // [6 f][AQ quad rows][16]
    float* sB = smem + NF * AQ * LDT;       // [6 f][64 channels][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile / ntn;
    const int m0 = mt * BM, n0 = (tile - mt * ntn) * BN;
    const int HoWo = g.Ho * g.Wo;
    const int nkc = g.Ci / BK, nk = 3 * nkc;
    const int n_lo = fdiv(m0, g.fd_howo);
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc_at(in, (size_t)n_lo * (size_t)(g.Hi * g.Wi) * (size_t)(g.Ci * 4), in_bytes);
    const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(U, u_bytes);

    // staging: thread = (row sr = tid >> 2, 16-byte chunk sc = tid & 3): quad sr of the A operand (6 pixels) and the six
    // rows (f, channel sr) of U
    const int sr = tid >> 2, sc = tid & 3;
    unsigned arow, urow;
    int aho;
    bool aleft, aright;
    // HALO: NI = 2 items per thread -- halo quad rows sr and 64 + sr (the second exists for sr < 2 Q); item validity is per
    // input row, independent of the filter row
    constexpr int NI = HALO ? 2 : 1;
    const int Q = g.Wo >> 2;                                            // quads per image row
    // the tile holds whole image rows: RS rows of each of its ns samples (ns = 1: R rows of one sample; ns > 1: whole samples);
    // a sample's block of the halo image has RS + 2 rows = HB quad rows
    const int ns = HALO ? (HoWo >= BM ? 1 : BM / HoWo) : 1;
    const int RS = HALO ? (64 / Q) / ns : 1;
    const int HB = (RS + 2) * Q, hq_n = ns * HB;
    unsigned hrow[NI];
    bool hvalid[NI], hleft[NI], hright[NI];
    const bool has2 = HALO && 64 + sr < hq_n;
    const bool wave_has2 = HALO && 64 + wave * 16 < hq_n;               // wave-uniform: some lane of this wave has a second item
    if constexpr (HALO) {
        const int rem0 = m0 - n_lo * HoWo;
        const int h0 = ns > 1 ? 0 : fdiv(rem0, g.fd_wo);                // first output row of the tile inside its sample
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int hq = sr + 64 * it;
            const int sm = hq / HB, hl = hq - sm * HB;
            const int hr = hl / Q, cq = hl - hr * Q;
            const int hi = h0 - 1 + hr, wo = 4 * cq;
            hvalid[it] = (it == 0 || has2) && (unsigned)hi < (unsigned)g.Hi;
            hleft[it] = wo > 0;
            hright[it] = wo + 4 < g.Wi;
            // (wraps for hi = -1: used only when valid)
            hrow[it] = (unsigned)((((sm * g.Hi + hi) * g.Wi + wo) * g.Ci + sc * 4) * 4);
        }
        aho = 0; aleft = aright = false; arow = 0;
        urow = (unsigned)((n0 + sr) * g.Ci + sc * 4) * 4u;
    } else {
        const int m = m0 + 4 * sr;                                      // whole tiles: always < M
        const int n = fdiv(m, g.fd_howo), rem = m - n * HoWo;
        const int ho = fdiv(rem, g.fd_wo), wo = rem - ho * g.Wo;
        aho = ho;
        aleft = wo > 0;
        aright = wo + 4 < g.Wi;
        arow = (unsigned)((((n - n_lo) * g.Hi + ho) * g.Wi + wo) * g.Ci + sc * 4) * 4u;
        urow = (unsigned)((n0 + sr) * g.Ci + sc * 4) * 4u;
    }
    const unsigned uplane = (unsigned)(g.Co * g.Ci) * 4u;               // bytes per (filter row, f) plane of U
    const int xgrp = XF ? m0 / bw.in_Mg : 0;
    const __amdgpu_buffer_rsrc_t rs_xs = make_rsrc(XF ? (const void*)(bw.in_scale + (size_t)xgrp * g.Ci) : (const void*)U,
                                                   XF ? (unsigned)g.Ci * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_xh = make_rsrc(XF ? (const void*)(bw.in_shift + (size_t)xgrp * g.Ci) : (const void*)U,
                                                   XF ? (unsigned)g.Ci * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_xm = make_rsrc((XF && bw.in_mean) ? (const void*)(bw.in_mean + (size_t)xgrp * g.Ci)
                                                                      : (const void*)U,
                                                   (XF && bw.in_mean) ? (unsigned)g.Ci * 4u : 0u);
    f32x4 pa[6], pu[6], xm, xs, xh;
    f32x4 pa2[HALO ? 6 : 1];                 // HALO: the second item
    unsigned xok = 0, xok2 = 0;
    int th = 0, cc = 0;
    auto load_tile = [&]() {                 // the k-tile (th, cc)
        const unsigned uoff = (unsigned)(th * NF) * uplane + (unsigned)(cc * BK) * 4u;
        if constexpr (HALO) {
#pragma unroll
            for (int f = 0; f < NF; ++f) pu[f] = bld4(rs_u, urow + uoff + (unsigned)f * uplane);
            if (th != 0) return;             // the V image of this channel chunk is already in LDS
            if constexpr (XF) {
                const unsigned coff = (unsigned)(cc * BK + sc * 4) * 4u;
                xm = bld4(rs_xm, coff);
                xs = bld4(rs_xs, coff);
                xh = bld4(rs_xh, coff);
            }
            const unsigned aoff = (unsigned)((cc * BK - g.Ci) * 4);          // channel chunk, one pixel to the left
            xok = xok2 = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool ok = hvalid[0] & (i == 0 ? hleft[0] : i == 5 ? hright[0] : true);
                xok |= ok ? (1u << i) : 0u;
                pa[i] = bld4(rs_in, ok ? hrow[0] + aoff + (unsigned)(i * g.Ci * 4) : kInvalidOff);
            }
            if (wave_has2) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const bool ok = hvalid[NI - 1] & (i == 0 ? hleft[NI - 1] : i == 5 ? hright[NI - 1] : true);
                    xok2 |= ok ? (1u << i) : 0u;
                    pa2[i] = bld4(rs_in, ok ? hrow[NI - 1] + aoff + (unsigned)(i * g.Ci * 4) : kInvalidOff);
                }
            }
            return;
        }
        const int dh = g.dh0 + g.dhs * th;
        const unsigned aoff = (unsigned)(((dh * g.Wi - 1) * g.Ci + cc * BK) * 4);      // (wraps; used only where the pixel exists)
        if constexpr (XF) {
            const unsigned coff = (unsigned)(cc * BK + sc * 4) * 4u;
            xm = bld4(rs_xm, coff);
            xs = bld4(rs_xs, coff);
            xh = bld4(rs_xh, coff);
        }
        const bool rowok = (unsigned)(aho + dh) < (unsigned)g.Hi;
        xok = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool ok = rowok & (i == 0 ? aleft : i == 5 ? aright : true);
            xok |= ok ? (1u << i) : 0u;
            pa[i] = bld4(rs_in, ok ? arow + aoff + (unsigned)(i * g.Ci * 4) : kInvalidOff);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) pu[f] = bld4(rs_u, urow + uoff + (unsigned)f * uplane);
    };
    auto advance = [&]() {
        if constexpr (HALO) {                // channel chunk outer, filter row inner
            const bool wrap = th == 2;
            th = wrap ? 0 : th + 1;
            cc += wrap ? 1 : 0;
            return;
        }
        const int c1 = cc + 1;
        const bool wrap = c1 == nkc;
        cc = wrap ? 0 : c1;
        th += wrap ? 1 : 0;
    };
    auto xform_item = [&](f32x4 (&p)[6], unsigned okbits) {
        if constexpr (XF) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool ok = (okbits >> i) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(__builtin_fmaf(p[i][e] - xm[e], xs[e], xh[e]), 0.f);     // bn_apply's expression
                    p[i][e] = ok ? v : 0.f;
                }
            }
        }
        const f32x4 d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4], d5 = p[5];
        const f32x4 e42 = d4 - 4.f * d2, e31 = d3 - 4.f * d1, f42 = d4 - d2, f31 = d3 - d1;
        p[0] = 4.f * d0 - 5.f * d2 + d4;
        p[1] = e42 + e31;
        p[2] = e42 - e31;
        p[3] = f42 + 2.f * f31;
        p[4] = f42 - 2.f * f31;
        p[5] = 4.f * d1 - 5.f * d3 + d5;
    };
    // (th, cc) here = the tile that was LOADED last: HALO transforms / stores the V image only at the start of a channel chunk
    auto xform_tile = [&]() {
        if constexpr (HALO) {
            if (th != 0) return;
            xform_item(pa, xok);
            if (wave_has2) xform_item(pa2, xok2);
            return;
        }
        xform_item(pa, xok);
    };
    const int swz = sc ^ ((sr >> 2) & 3);
    auto store_tile = [&]() {
        float* a = sA + sr * LDT + swz * 4;
        float* b = sB + sr * LDT + swz * 4;
        if constexpr (HALO) {
#pragma unroll
            for (int f = 0; f < NF; ++f) st4(b + f * 64 * LDT, pu[f]);
            if (th != 0) return;
#pragma unroll
            for (int f = 0; f < NF; ++f) st4(a + f * AQ * LDT, pa[f]);
            if (has2) {                      // quad row 64 + sr: the same swizzle bits (64 is a multiple of 16)
#pragma unroll
                for (int f = 0; f < NF; ++f) st4(a + (f * AQ + 64) * LDT, pa2[f]);
            }
            return;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            st4(a + f * 64 * LDT, pa[f]);
            st4(b + f * 64 * LDT, pu[f]);
        }
    };

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // fragment group gi = (kk = gi >> 1, frequencies 3 (gi & 1) .. + 2): lane reads 4 consecutive k of its row
    const int frow = lane & 31;
    const int b_base = (wn * 32 + frow) * LDT;
    const int fsw = (frow >> 2) & 3;
    // HALO: the A rows of the tile being multiplied start (1 + dh) Q quad rows into the halo image (hoff, set per k-tile)
    int arow_cur = wm * 32 + frow;
    const int arow_lane = HALO ? (wm * 32 + frow) + ((wm * 32 + frow) / (RS * Q)) * 2 * Q : 0;    // + the halo rows of earlier samples
    auto set_hoff = [&](int thc) {
        if constexpr (HALO) arow_cur = arow_lane + (1 + g.dh0 + g.dhs * thc) * Q;
    };
    auto read_frags = [&](int gi, f32x4 (&a)[3], f32x4 (&b)[3]) {
        const int cl = (gi >> 1) * 2 + (lane >> 5);
        const int chunk = cl ^ fsw;
        const int chunk_a = HALO ? (cl ^ ((arow_cur >> 2) & 3)) : chunk;
        const int f0 = (gi & 1) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a[j] = ld4(sA + ((f0 + j) * AQ + arow_cur) * LDT + chunk_a * 4);
            b[j] = ld4(sB + (f0 + j) * 64 * LDT + b_base + chunk * 4);
        }
    };
    auto mma12 = [&](int gi, const f32x4 (&a)[3], const f32x4 (&b)[3]) {
        const int f0 = (gi & 1) * 3;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[f0 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][tt], b[j][tt], acc[f0 + j], 0, 0, 0);
    };

    f32x4 fa[3], fb[3];
    load_tile();
    xform_tile();
    store_tile();
    __syncthreads();
    int thc = 0;                             // HALO: filter row of the tile being multiplied
    set_hoff(0);
    read_frags(0, fa, fb);
    for (int kt = 0; kt + 1 < nk; ++kt) {
        advance();
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            f32x4 na[3], nb[3];
            read_frags(gi + 1, na, nb);
            mma12(gi, fa, fb);
#pragma unroll
            for (int j = 0; j < 3; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
            if (gi == 0) {                       // the fetches of tile kt + 1 go out behind the first MFMA group ...
                load_tile();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (gi == 1) xform_tile();           // ... and their transform runs under the third
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) { pin(fa[j]); pin(fb[j]); }      // group-3 fragments in registers BEFORE the barrier
        __syncthreads();                         // every wave has read the last fragments of tile kt
        store_tile();
        __syncthreads();
        thc = thc == 2 ? 0 : thc + 1;
        set_hoff(thc);                           // (the fragments read next belong to tile kt + 1)
        f32x4 na[3], nb[3];
        read_frags(0, na, nb);
        __builtin_amdgcn_sched_barrier(0);
        mma12(3, fa, fb);
#pragma unroll
        for (int j = 0; j < 3; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
    }
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
        f32x4 na[3], nb[3];
        read_frags(gi + 1, na, nb);
        mma12(gi, fa, fb);
#pragma unroll
        for (int j = 0; j < 3; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
    }
    mma12(3, fa, fb);

    // output transform: acc[0..3] <- y0..y3 (the wave's 32 quads x 32 channels; pixel = 4 q + i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
        const float s34 = acc[3][r] + acc[4][r], d34 = acc[3][r] - acc[4][r];
        const float y0 = acc[0][r] + s12 + s34, y3 = d12 + 8.f * d34 + acc[5][r];
        acc[1][r] = d12 + 2.f * d34;
        acc[2][r] = s12 + 4.f * s34;
        acc[0][r] = y0;
        acc[3][r] = y3;
    }
    // this wave's 128 rows are statistics tile st of the tensor; its column: co
    const int st = 2 * mt + wm;
    const int co = n0 + wn * 32 + (lane & 31);
    if (st_mean) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[i][r];
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / 128.0f);
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) m2 += (acc[i][r] - mean) * (acc[i][r] - mean);
        m2 += __shfl_xor(m2, 32, 64);
        if (lane < 32) {
            st_mean[(size_t)st * g.Co + co] = mean;
            st_m2[(size_t)st * g.Co + co] = m2;
        }
    }
    // dense output, whole tiles: one VGPR offset per lane, the row steps in the scalar offset of the buffer instructions
    const size_t out_base = (size_t)(m0 + wm * 128) * (size_t)(g.Co * 4);
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_at(out, out_base, out_bytes);
    const unsigned rowstep = (unsigned)g.Co * 4u;
    const unsigned lane_off = (unsigned)(16 * (lane >> 5)) * rowstep + (unsigned)co * 4u;      // q = .. + 4 (lane >> 5): 4 q rows
    if constexpr (BWE) {
        const __amdgpu_buffer_rsrc_t rs_y = make_rsrc_at(bw.y, out_base, out_bytes);
        const __amdgpu_buffer_rsrc_t rs_ao = make_rsrc_at(bw.a_out ? bw.a_out : (void*)out, out_base,
                                                           bw.a_out ? out_bytes : out_base);     // absent: stores dropped
        const int gcol = ((m0 + wm * 128) / bw.Mg) * g.Co + co;
        const float mu = bw.mean[gcol], rs = bw.rstd[gcol];
        const float sc_ = bw.mscale ? bw.mscale[gcol] : 0.f, sh_ = bw.mscale ? bw.mshift[gcol] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                yv[r] = ld_el_s<float>(rs_y, lane_off, (unsigned)(4 * ((r & 3) + 8 * (r >> 2)) + i) * rowstep);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned so = (unsigned)(4 * ((r & 3) + 8 * (r >> 2)) + i) * rowstep;
                const float t = __builtin_fmaf(yv[r] - mu, sc_, sh_);          // bn(y), bn_apply's fma
                float v = acc[i][r];
                if (bw.mscale) v = t > 0.f ? v : 0.f;
                s1 += v;
                s2 += v * ((yv[r] - mu) * rs);
                st_el_s<float>(v, rs_out, lane_off, so);
                st_el_s<float>(fmaxf(t, 0.f), rs_ao, lane_off, so);
            }
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lane < 32) {
            bw.p1[(size_t)st * g.Co + co] = s1;
            bw.p2[(size_t)st * g.Co + co] = s2;
        }
    } else {
        const float bias = bw.bias ? bw.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][r];
                if (bw.bias) {
                    v += bias;
                    v = (bw.relu && v < 0.f) ? 0.f : v;
                }
                st_el_s<float>(v, rs_out, lane_off, (unsigned)(4 * ((r & 3) + 8 * (r >> 2)) + i) * rowstep);
            }
    }
}

// Filter transform of conv_wino4_kernel: w [Co][9][Ci] -> U [filter-row step th][f][Co][Ci], f = 0..5 (see the kernel);
// tap mapping as in wino_filter_kernel.
__global__ __launch_bounds__(256) void wino4_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Co,
                                                          int Ci, int S, int r0, int rs, int s0, int ss, int dw0,
                                                          int dws) {
    const int th = blockIdx.y, c4n = Ci / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Co * c4n) return;
    const int o = idx / c4n, c = (idx - o * c4n) * 4;
    f32x4 gq[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tw = (j - 1 - dw0) * dws;
        const int tap = (r0 + rs * th) * S + s0 + ss * tw;
        gq[j] = ld4(w + ((size_t)o * (3 * S) + tap) * Ci + c);
    }
    const size_t plane = (size_t)Co * Ci;
    float* dst = U + (size_t)th * 6 * plane + (size_t)o * Ci + c;
    const f32x4 s02 = gq[0] + gq[2], q02 = gq[0] + 4.f * gq[2];
    st4(dst, 0.25f * gq[0]);
    st4(dst + plane, (-1.0f / 6.0f) * (s02 + gq[1]));
    st4(dst + 2 * plane, (-1.0f / 6.0f) * (s02 - gq[1]));
    st4(dst + 3 * plane, (1.0f / 24.0f) * (q02 + 2.f * gq[1]));
    st4(dst + 4 * plane, (1.0f / 24.0f) * (q02 - 2.f * gq[1]));
    st4(dst + 5 * plane, gq[2]);
}

// Winograd F(2, 3) filter transform along the filter ROW for conv_nt_kernel<..., WINO>: w [Co][9][Ci] (the operand layout
// of the direct kernel; for a data gradient the transposed filter) -> U [filter-row step th][f][Co][Ci] with
// U = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2), where g_j is the tap that meets the input column w + j - 1 in the
// launch's gather geometry (forward: tap j; data gradient: tap 2 - j) and th walks the geometry's row taps in kernel order.
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Co,
                                                         int Ci, int S, int r0, int rs, int s0, int ss, int dw0,
                                                         int dws) {
    const int th = blockIdx.y, c4n = Ci / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Co * c4n) return;
    const int o = idx / c4n, c = (idx - o * c4n) * 4;
    f32x4 gq[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tw = (j - 1 - dw0) * dws;                    // dws = +-1
        const int tap = (r0 + rs * th) * S + s0 + ss * tw;
        gq[j] = ld4(w + ((size_t)o * (3 * S) + tap) * Ci + c);
    }
    const size_t plane = (size_t)Co * Ci;
    float* dst = U + (size_t)th * 4 * plane + (size_t)o * Ci + c;
    st4(dst, gq[0]);
    st4(dst + plane, 0.5f * (gq[0] + gq[1] + gq[2]));
    st4(dst + 2 * plane, 0.5f * (gq[0] - gq[1] + gq[2]));
    st4(dst + 3 * plane, gq[2]);
}

// dst[i] = sum_z partial[z][i]: block = 32 float4 columns x 8 split lanes (8 loads in flight per lane),
// fixed summation order -> deterministic
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial,
                                                           float* __restrict__ dst, size_t n4, int splits) {
    __shared__ f32x4 red[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + tx;
    const bool ok = i < n4;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(partial);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int z0 = ty; z0 < splits; z0 += 64) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = z0 + 8 * u;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                v[u] = z < splits ? p4[(size_t)z * n4 + i] : zero;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && ok) {
#pragma unroll
        for (int k = 1; k < 8; ++k) s += red[k][tx];
        reinterpret_cast<f32x4*>(dst)[i] = s;
    }
}

// floats per output channel of the filter (gradient) a launch addresses
inline size_t io_filter_row(const IoConvGeom& g) {
    return g.cr ? (size_t)io_stem_kp(g.wT, g.cr) : (size_t)g.wT * (g.gw ? g.gw : g.Ci);
}

struct WgradPlan {
    int bmo, bnc, ntile_c, tiles, splits, kps;
};

WgradPlan plan_wgrad(const IoConvGeom& g, int stem, bool fp32 = true) {
    WgradPlan p;
    const long M = (long)g.N * g.Ho * g.Wo;
    p.bmo = (g.Co % 128 == 0) ? 128 : 64;
    if (stem) {
        p.bnc = 64;
        p.ntile_c = g.cr ? io_stem_kp(g.wT, g.cr) / 64 + (io_stem_kp(g.wT, g.cr) % 64 != 0)
                         : io_cdiv((long)g.wT * g.Ci, 64);
        p.tiles = (g.Co / p.bmo) * p.ntile_c;
    } else if (g.gw) {
        p.bmo = p.bnc = g.gw;
        p.ntile_c = 1;
        p.tiles = (g.Co / p.bmo) * g.Th * g.Tw;
    } else {
        p.bnc = (g.Ci % 128 == 0) ? 128 : 64;
        p.ntile_c = g.Ci / p.bnc;
        p.tiles = (g.Co / p.bmo) * g.Th * g.Tw * p.ntile_c;
    }
    if (p.tiles < 1) p.tiles = 1;               // unsupported channel counts are rejected by the launcher, not here
    const bool fp32_tr = fp32 && !stem && !g.gw && p.bmo == 128 && p.bnc == 128;
    const int nkt = io_cdiv(M, 32);
    // 2 blocks fit a CU (LDS), so 512 run at once: fill at most two full rounds -- one block more than that would
    // cost a third, almost empty round (a 3x3 conv with 36 tiles: 29 splits = 1044 blocks ran 25 % slower than 28)
    // the fp32 128 x 128 kernel runs three blocks per CU (768 at once): one full round -- three for the layers with many
    // tiles, whose blocks would otherwise be few and long (same-box A/B: +1..4 % per layer against 1024)
    int want = (fp32_tr ? (p.tiles >= 128 ? 2304 : 768) : 1024) / p.tiles;
    int maxs = nkt / 8 > 0 ? nkt / 8 : 1;       // at least 8 k-tiles (256 rows) per split
    p.splits = want < maxs ? want : maxs;
    if (p.splits < 1) p.splits = 1;
    p.kps = io_cdiv(nkt, p.splits);
    p.splits = io_cdiv(nkt, p.kps);
    return p;
}

// the Winograd row form of the fp32 filter gradient: 3x3 stride-1 same-size convolutions with 8 | Wo and 64 | M
bool wgrad_wino_shape_ok(const IoConvGeom& g, int stem) {
    return !stem && !g.gw && !g.cr && g.Th == 3 && g.Tw == 3 && g.S == 3 && g.wT == 9 && g.is == 1 &&
           g.os == 1 && g.Hi == g.Ho && g.Wi == g.Wo && g.dh0 == -1 && g.dhs == 1 && g.dw0 == -1 && g.dws == 1 &&
           g.rs == 1 && g.ss == 1 && g.r0 == 0 && g.s0 == 0 && g.Wo % 8 == 0 && ((long)g.N * g.Ho * g.Wo) % 64 == 0 &&
           g.Ci % 64 == 0 && g.Co % 64 == 0;
}
bool wgrad_wino_ok(const IoConvGeom& g, int stem) { return io_wino_on() && wgrad_wino_shape_ok(g, stem); }
WgradPlan plan_wgrad_wino(const IoConvGeom& g) {
    WgradPlan p;
    p.bmo = p.bnc = 64;
    p.ntile_c = g.Ci / 64;
    p.tiles = (g.Co / 64) * 3 * p.ntile_c;
    const int nkt = (int)(((long)g.N * g.Ho * g.Wo) / 64);
    // two blocks per CU: 512 resident -- two full rounds
    const int want = 1024 / p.tiles, maxs = nkt / 8 > 0 ? nkt / 8 : 1;
    p.splits = want < maxs ? want : maxs;
    if (p.splits < 1) p.splits = 1;
    p.kps = io_cdiv(nkt, p.splits);
    p.splits = io_cdiv(nkt, p.kps);
    return p;
}

}  // namespace

size_t io_conv_wgrad_partial_bytes(const IoConvGeom& g, int stem) {
    // sized for either storage type: the fp32 and the bf16 kernels split differently
    const int s0 = plan_wgrad(g, stem, true).splits, s1 = plan_wgrad(g, stem, false).splits;
    int splits = s0 > s1 ? s0 : s1;
    if (wgrad_wino_shape_ok(g, stem)) {          // (by shape alone: the size must not depend on the run-time switch)
        const int sw = plan_wgrad_wino(g).splits;
        if (sw > splits) splits = sw;
    }
    size_t need = splits == 1 ? 0 : (size_t)splits * g.Co * io_filter_row(g) * sizeof(float);
    if (stem && g.cr && io_stem_rows_ok(g)) {      // one partial per block of the row-persistent kernel
        const size_t rows = (size_t)io_stem_wgrad_rows_max_blocks() * g.Co * io_filter_row(g) * sizeof(float);
        if (rows > need) need = rows;
    }
    if (stem && !g.cr && g.Ci == 8 && g.Co == 64 && g.Wo == 128)      // the bf16 stem's one partial per block (conv_halo3.hip)
        if (io_stem_wgrad_halo_partial_bytes() > need) need = io_stem_wgrad_halo_partial_bytes();
    if (!stem && io_wgrad_halo3_shape(g) && io_wgrad_halo3_partial_bytes() > need) need = io_wgrad_halo3_partial_bytes();
    return need;
}

static std::atomic<int> g_last_wgrad_route{0};      // tests: 1 = the last filter gradient ran on stem_wgrad_halo_kernel, 2 = conv_wgrad_halo3_kernel
extern "C" int io_debug_last_wgrad_route(void) { return g_last_wgrad_route.load(std::memory_order_relaxed); }
// which kernel family the last forward / data-gradient launch went to (tests: 0 = conv_nt_kernel, 1 = conv_p256, 2 = conv_halo3, 3 = stem_halo)
static std::atomic<int> g_last_route{0};
extern "C" int io_debug_last_nt_route(void) { return g_last_route.load(std::memory_order_relaxed); }

int io_launch_conv_nt(const IoConvGeom& g, const void* in, const void* wgt, void* out, const void* add,
                      const void* mask, int stem, hipStream_t st, float* st_mean, float* st_m2,
                      const IoBwStats* bw, int dt_in, int dt_out) {
    IoBwStats bws;
    memset(&bws, 0, sizeof(bws));
    if (bw) {
        bws = *bw;
        const long Mchk = (long)g.N * g.Ho * g.Wo;
        // the fused epilogue addresses whole tiles without validating rows: no partial tile, no tile across two groups
        IO_REQUIRE(!bws.y || (g.os == 1 && g.Ho == g.outH && g.Wo == g.outW && bws.Mg > 0 && bws.Mg % 128 == 0 &&
                              Mchk % 128 == 0 && Mchk % bws.Mg == 0),
                   IO_ERR_SHAPE, "conv_nt: fused BN-backward reductions need a dense output and 128 | rows per group | M");
        IO_REQUIRE(!bws.xb_res || (bws.xb_a && !bws.y && dt_in == dt_out), IO_ERR_SHAPE,
                   "conv_nt: the residual operand form is a forward path (no BatchNorm-backward epilogue)");
        // (the output may sit on a strided lattice -- the one class of a strided 1x1 data gradient that has a tap; the operand
        // side only needs the gathered grid to BE the logical output grid)
        IO_REQUIRE(!bws.xb_a || (bws.xb_b && bws.xb_c && bws.xb_y && !bws.in_scale && !stem && !g.gw && g.is == 1 &&
                                 g.Hi == g.Ho && g.Wi == g.Wo &&
                                 bws.xb_Mg > 0 && bws.xb_Mg % 128 == 0 && Mchk % bws.xb_Mg == 0 && g.dhs * g.dhs == 1 &&
                                 g.dws * g.dws == 1 && g.dh0 * (g.dh0 + g.dhs * (g.Th - 1)) <= 0 &&
                                 g.dw0 * (g.dw0 + g.dws * (g.Tw - 1)) <= 0),
                   IO_ERR_SHAPE,
                   "conv_nt: the backward operand transform needs a stride-1 same-size data gradient whose taps include "
                   "the centre, and 128 | rows per group | M");
    }
    IO_REQUIRE(!(stem && bws.y), IO_ERR_SHAPE, "conv_nt: the stem has no BatchNorm-backward epilogue");
    IO_REQUIRE(!bws.a_out || (bws.y && bws.mscale && bws.mshift), IO_ERR_SHAPE,
               "conv_nt: the activation side output needs the BatchNorm-backward epilogue with its mask tables");
    IO_REQUIRE(!bws.in_scale || (!stem && !bws.y && bws.in_shift && bws.in_Mg > 0 && bws.in_Mg % 128 == 0 && !g.gw),
               IO_ERR_SHAPE, "conv_nt: the input transform needs a regular forward convolution and 128 | rows per group");
    IO_REQUIRE((st_mean == nullptr) == (st_m2 == nullptr), IO_ERR_SHAPE, "conv_nt: statistics outputs come in pairs");
    IO_REQUIRE(!st_mean || (g.os == 1 && g.Ho == g.outH && g.Wo == g.outW && !add && !mask), IO_ERR_SHAPE,
               "conv_nt: fused statistics need a plain dense forward convolution");
    IO_REQUIRE(g.Co % 64 == 0, IO_ERR_SHAPE, "conv_nt: Co=%d must be a multiple of 64", g.Co);
    const int es = io_dtype_bytes(dt_in), os = io_dtype_bytes(dt_out);
    IO_REQUIRE(!g.cr || (stem && dt_in == IO_F32 && dt_out == IO_F32 && g.cr > 0 && g.cr <= 8), IO_ERR_SHAPE,
               "conv_nt: the exact-K mode is for the fp32 stem (1..8 real channels)");
    if (stem)
        IO_REQUIRE(g.Ci == 8, IO_ERR_SHAPE, "conv_nt(stem): needs the packed input with Ci=8 (5 padded)");
    else
        IO_REQUIRE(g.Ci % (128 / es) == 0, IO_ERR_SHAPE, "conv_nt: Ci=%d must be a multiple of %d", g.Ci, 128 / es);
    IO_REQUIRE(g.N > 0 && g.Ho > 0 && g.Wo > 0 && g.Hi > 0 && g.Wi > 0, IO_ERR_SHAPE,
               "conv_nt: empty tensor (N=%d, in %dx%d, out %dx%d)", g.N, g.Hi, g.Wi, g.Ho, g.Wo);
    const long M = (long)g.N * g.Ho * g.Wo;
    IO_REQUIRE(M > 0 && M < (1L << 31), IO_ERR_SHAPE, "conv_nt: bad M=%ld", M);
    // whole tensors may exceed 4 GiB (descriptors are rebased per tile); what a tile spans -- the samples of 128
    // consecutive rows, relative to the first -- must fit 32-bit byte offsets, and pixel counts must fit an int
    const double w_b = (double)es * g.Co * (double)io_filter_row(g);
    const double span = 128.0 / ((double)g.Ho * g.Wo) + 2.0;
    IO_REQUIRE(w_b < 4.0e9 && span * es * g.Hi * g.Wi * g.Ci < 4.0e9 && span * os * g.outH * g.outW * g.Co < 4.0e9,
               IO_ERR_SHAPE, "conv_nt: filter or per-tile sample span larger than 4 GB (32-bit offsets)");
    IO_REQUIRE((double)g.N * g.Hi * g.Wi < 2.0e9 && (double)g.N * g.outH * g.outW < 2.0e9, IO_ERR_SHAPE,
               "conv_nt: more than 2^31 pixels");
    const size_t in_bytes = (size_t)es * g.N * g.Hi * g.Wi * g.Ci;
    const size_t out_bytes = (size_t)os * g.N * g.outH * g.outW * g.Co;
    const unsigned w_bytes = (unsigned)w_b;
    // bf16 stem on 128-wide output rows: the patch-in-LDS / filters-in-registers kernel of conv_halo3.hip
    if (dt_in == IO_BF16 && dt_out == IO_BF16 && stem && !add && !mask) {
        const int rs = io_launch_conv_stem_halo(g, in, wgt, out, st, st_mean, st_m2, bw ? &bws : nullptr, in_bytes, w_bytes,
                                                out_bytes);
        if (rs <= 0) {
            g_last_route.store(3, std::memory_order_relaxed);
            return rs;
        }
    }
    // bf16: the persistent 256-row LDS-DMA kernel where the shape and the form are its own (conv_p256.hip)
    if (dt_in == IO_BF16 && dt_out == IO_BF16 && !stem) {
        const int rh = io_launch_conv_halo3(g, in, wgt, out, add, mask, st, st_mean, st_m2, bw ? &bws : nullptr, in_bytes,
                                            w_bytes, out_bytes);
        if (rh <= 0) {
            g_last_route.store(2, std::memory_order_relaxed);
            return rh;
        }
        const int rc = io_launch_conv_p256(g, in, wgt, out, add, mask, st, st_mean, st_m2, bw ? &bws : nullptr, in_bytes,
                                           w_bytes, out_bytes);
        if (rc <= 0) {
            g_last_route.store(1, std::memory_order_relaxed);
            return rc;
        }
    }
    g_last_route.store(0, std::memory_order_relaxed);
    // Output-channel tile: 128 wide where that leaves enough tiles to fill the chip, 64 wide otherwise -- a small per-GPU batch
    // (the reference's own 32 pairs per GPU, or a strong-scaling rank) gives layers 3-4 only 32..128 row tiles, and 128-wide
    // tiles then occupy a fraction of the 256 CUs with one block each (measured at 32 pairs: 43-51 TF/s on the 8 x 8 maps).
    static std::atomic<int> small_tiles_c{-1};
    int small_tiles = small_tiles_c.load(std::memory_order_relaxed);
    if (small_tiles < 0) {
        const char* e = getenv("IO_NT_SMALL_TILES");        // (experiments: the largest 128-wide tile count that still goes 64 wide)
        small_tiles = e ? atoi(e) : 256;   // one 128-wide tile per CU or fewer (same-box, 32 pairs: 1388 -> 1428 pairs/s fp32, 3014 -> 3090 bf16)
        small_tiles_c.store(small_tiles, std::memory_order_relaxed);
    }
    const long tiles128 = (long)io_cdiv(M, 128) * (g.Co / 128);
    const int bn = g.gw ? g.gw : ((g.Co % 128 == 0 && tiles128 > small_tiles) ? 128 : 64);
    IO_REQUIRE(!g.gw || (g.gw == 64 && !stem && g.Ci == g.Co), IO_ERR_SHAPE,
               "conv_nt: grouped mode needs a 64-channel window and Ci == Co");
    const int ntn = g.Co / bn;
    const long tiles = (long)io_cdiv(M, 128) * ntn;
    IO_REQUIRE(tiles < (1L << 31), IO_ERR_SHAPE, "conv_nt: grid too large");
    dim3 grid((unsigned)tiles), block(kThreads);
    // dense 1x1 stride-1 GEMM on whole tiles: the addressing-free instantiation (LIN)
    const bool lin = !stem && !g.gw && g.Th * g.Tw == 1 && g.is == 1 && g.os == 1 && g.dh0 == 0 && g.dw0 == 0 &&
                     g.Hi == g.Ho && g.Wi == g.Wo && g.outH == g.Ho && g.outW == g.Wo && M % 128 == 0 &&
                     128.0 * g.Ci * es < 4.0e9 && 128.0 * g.Co * os < 4.0e9;
    IO_REQUIRE(!bws.xb_res || lin, IO_ERR_SHAPE, "conv_nt: the residual operand form needs a dense 1x1 stride-1 GEMM on whole tiles");
    // 3x3 stride-1 same-size convolutions / data gradients in fp32 with a scratch for the transformed filters: the
    // Winograd F(2, 3) row form (2/3 of the MFMAs).  Forms it carries: plain (incl. the inference epilogue bias + ReLU), input
    // transform (XF), fused BatchNorm-backward epilogue (BWE), each with optional statistics; everything else stays on the
    // direct kernel.
    if (io_wino_on() && bws.wino_u && !stem && !g.gw && dt_in == IO_F32 && dt_out == IO_F32 && g.Th == 3 && g.Tw == 3 &&
        g.S == 3 && g.wT == 9 && g.is == 1 && g.os == 1 && g.Hi == g.Ho && g.Wi == g.Wo && g.outH == g.Ho &&
        g.outW == g.Wo && g.Wo % 2 == 0 && M % 128 == 0 && g.Ci % 32 == 0 && !add && !mask && !bws.xb_a &&
        g.dhs * g.dhs == 1 && g.dws * g.dws == 1 && g.dw0 * (g.dw0 + 2 * g.dws) == -1 && g.rs == 1 && g.ss == 1 &&
        !(bws.y && bws.in_scale) && 12.0 * g.Co * g.Ci * 4.0 < 4.0e9) {
        const double kred9 = 9.0 * g.Ci;
        // F(4, 3): 4 | Wo, whole 256-row tiles (per BatchNorm group where tables are indexed by group)
        const bool wino4 = g.Wo % 4 == 0 && M % 256 == 0 && g.Ci % 16 == 0 && (!bws.y || bws.Mg % 256 == 0) &&
                           (!bws.in_scale || bws.in_Mg % 256 == 0) && 18.0 * g.Co * g.Ci * 4.0 < 4.0e9;
        const double fl9 = 2.0 * (double)M * g.Co * kred9;
        IoProfScope prof(IO_PROF_CONV_WINO, fl9,
                         (double)os * M * g.Co * (1.0 + (bws.y ? 1.0 : 0.0)) +
                             (double)es * ((double)g.N * g.Hi * g.Wi * g.Ci + (double)g.Co * kred9), st,
                         wino4 ? fl9 * 0.5 : fl9 * (2.0 / 3.0));
        if (wino4) {
            hipLaunchKernelGGL(wino4_filter_kernel, dim3((unsigned)io_cdiv((long)g.Co * g.Ci / 4, 256), 3), dim3(256), 0, st,
                               (const float*)wgt, bws.wino_u, g.Co, g.Ci, g.S, g.r0, g.rs, g.s0, g.ss, g.dw0, g.dws);
            const unsigned u_bytes = (unsigned)(18.0 * g.Co * g.Ci * 4.0);
            const int ntw = g.Co / 64;
            dim3 grid4((unsigned)((M / 256) * ntw));
            // tiles of whole image rows of one sample: the V image is staged once per channel chunk for all three filter rows
            const long hw = (long)g.Ho * g.Wo;
            const bool halo = g.Wo <= 64 && 256 % g.Wo == 0 &&
                              (hw % 256 == 0 || (256 % hw == 0 && (256 / hw) * (g.Ho + 2) * (g.Wo / 4) <= 96));
            const size_t lds4 = (size_t)6 * ((halo ? 96 : 64) + 64) * 16 * sizeof(float);
#define IO_LAUNCH_W4_(BWE_, XF_, HALO_)                                                                                 \
    hipLaunchKernelGGL((conv_wino4_kernel<BWE_, XF_, HALO_>), grid4, block, lds4, st, g, (const float*)in,              \
                       (const float*)bws.wino_u, (float*)out, ntw, in_bytes, u_bytes, out_bytes, st_mean, st_m2, bws)
            if (halo) {
                if (bws.y) IO_LAUNCH_W4_(true, false, true);
                else if (bws.in_scale) IO_LAUNCH_W4_(false, true, true);
                else IO_LAUNCH_W4_(false, false, true);
            } else {
                if (bws.y) IO_LAUNCH_W4_(true, false, false);
                else if (bws.in_scale) IO_LAUNCH_W4_(false, true, false);
                else IO_LAUNCH_W4_(false, false, false);
            }
#undef IO_LAUNCH_W4_
            return io_check_launch("conv_nt(wino4)");
        }
        hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)io_cdiv((long)g.Co * g.Ci / 4, 256), 3), dim3(256), 0, st,
                           (const float*)wgt, bws.wino_u, g.Co, g.Ci, g.S, g.r0, g.rs, g.s0, g.ss, g.dw0, g.dws);
        const unsigned wu_bytes = (unsigned)(12.0 * g.Co * g.Ci * 4.0);
        const int ntw = g.Co / 64;
        dim3 gridw((unsigned)((M / 128) * ntw));
        const size_t ldsz = (size_t)(256 + 256) * 32 * sizeof(float);
#define IO_LAUNCH_WINO_(BWE_, XF_)                                                                                    \
    do {                                                                                                             \
        static std::atomic<unsigned long long> attr_done{0};                                                         \
        if (io_first_on_device(attr_done))                                                                           \
            (void)hipFuncSetAttribute(                                                                               \
                (const void*)conv_nt_kernel<float, float, 64, 0, 4, 1, 2, BWE_, XF_, false, 0, true>,                \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);                                              \
        hipLaunchKernelGGL((conv_nt_kernel<float, float, 64, 0, 4, 1, 2, BWE_, XF_, false, 0, true>), gridw, block,  \
                           ldsz, st, g, (const float*)in, (const float*)bws.wino_u, (float*)out, (const float*)nullptr, \
                           (const float*)nullptr, ntw, in_bytes, wu_bytes, out_bytes, st_mean, st_m2, bws);          \
    } while (0)
        if (bws.y) IO_LAUNCH_WINO_(true, false);
        else if (bws.in_scale) IO_LAUNCH_WINO_(false, true);
        else IO_LAUNCH_WINO_(false, false);
#undef IO_LAUNCH_WINO_
        return io_check_launch("conv_nt(wino)");
    }
    // algorithmic work: real taps x real channels (the stem's 3 padding channels do not count)
    const double kred = stem ? (double)g.wT * 5.0 : (double)g.Th * g.Tw * (g.gw ? g.gw : g.Ci);
    IoProfScope prof(stem ? IO_PROF_CONV_STEM : (bn == 128 ? IO_PROF_CONV_NT128 : IO_PROF_CONV_NT64),
                     2.0 * (double)M * g.Co * kred,
                     (double)os * M * g.Co * (1.0 + (add ? 1.0 : 0.0) + (mask ? 1.0 : 0.0) + ((bw && bw->y) ? 1.0 : 0.0)) +
                         // (a lattice class of a strided data gradient that no tap reaches reads nothing: it only writes zeros)
                         (double)es * ((kred > 0.0 ? (double)g.N * g.Hi * g.Wi * g.Ci : 0.0) *
                                           ((bw && bw->xb_a) ? (bw->xb_out ? 3.0 : 2.0) : 1.0) +
                                       (double)g.Co * kred),
                     st);
#define IO_LAUNCH_NT_(TI_, TO_, BN_, STEM_, NBUF_, MINB_, BWE_, XF_, LIN_, XB_)                               \
    do {                                                                                                     \
        const size_t ldsz = (size_t)NBUF_ * (128 + BN_) * (BN_ == 64 ? 32 : 36) * sizeof(float);             \
        static std::atomic<unsigned long long> attr_done{0};                                                                       \
        if (io_first_on_device(attr_done)) {                                                                                    \
            (void)hipFuncSetAttribute(                                                                       \
                (const void*)conv_nt_kernel<TI_, TO_, BN_, STEM_, 4, NBUF_, MINB_, BWE_, XF_, LIN_, XB_>,    \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);                                      \
        }                                                                                                    \
        hipLaunchKernelGGL((conv_nt_kernel<TI_, TO_, BN_, STEM_, 4, NBUF_, MINB_, BWE_, XF_, LIN_, XB_>),    \
                           grid, block, ldsz, st, g, (const TI_*)in, (const TI_*)wgt, (TO_*)out,             \
                           (const TO_*)add, (const TO_*)mask, ntn, in_bytes, w_bytes, out_bytes, st_mean,    \
                           st_m2, bws);                                                                      \
    } while (0)
#define IO_LAUNCH_NT__(TI_, TO_, BN_, STEM_, NBUF_, MINB_, LIN_)                                             \
    do {                                                                                                     \
        constexpr bool R_ = STEM_ == 0, XBOK_ = R_ && sizeof(TI_) == sizeof(TO_);                            \
        constexpr int XB1_ = XBOK_ ? 1 : 0, XB2_ = (XBOK_ && LIN_) ? 2 : XB1_;                               \
        constexpr int XB3_ = XB2_ == 2 ? 3 : XB1_;                                                           \
        if (XBOK_ && bws.xb_a && bws.xb_res == 2) IO_LAUNCH_NT_(TI_, TO_, BN_, STEM_, NBUF_, MINB_, false, false, LIN_, XB3_); \
        else if (XBOK_ && bws.xb_a && bws.xb_res) IO_LAUNCH_NT_(TI_, TO_, BN_, STEM_, NBUF_, MINB_, false, false, LIN_, XB2_); \
        else if (XBOK_ && bws.xb_a && bws.y) IO_LAUNCH_NT_(TI_, TO_, BN_, STEM_, NBUF_, MINB_, R_, false, LIN_, XB1_); \
        else if (XBOK_ && bws.xb_a) IO_LAUNCH_NT_(TI_, TO_, BN_, STEM_, NBUF_, MINB_, false, false, LIN_, XB1_);  \
        else if (R_ && bws.y) IO_LAUNCH_NT_(TI_, TO_, BN_, STEM_, NBUF_, MINB_, R_, false, LIN_, false);     \
        else if (R_ && bws.in_scale) IO_LAUNCH_NT_(TI_, TO_, BN_, STEM_, NBUF_, MINB_, false, R_, LIN_, false); \
        else IO_LAUNCH_NT_(TI_, TO_, BN_, STEM_, NBUF_, MINB_, false, false, LIN_, false);                   \
    } while (0)
#define IO_LAUNCH_NT(TI_, TO_, BN_, STEM_, NBUF_, MINB_)                                                     \
    do {                                                                                                     \
        if (STEM_ == 0 && lin) IO_LAUNCH_NT__(TI_, TO_, BN_, STEM_, NBUF_, MINB_, (STEM_ == 0));             \
        else IO_LAUNCH_NT__(TI_, TO_, BN_, STEM_, NBUF_, MINB_, false);                                      \
    } while (0)
    // 128-wide tiles run single-buffered (36.9 KB of LDS, a second barrier per k-tile) with the register allocator held
    // to three blocks per CU: three waves per SIMD keep the matrix pipe fuller than two even on the MFMA-bound layers
    // (fp32 3x3 256->256: 132 -> 139 TF/s) and put more loads in flight on the short-K ones (same-box A/B over all
    // ResNet-50 shapes: forward 46.2 -> 43.5 ms, data gradient 46.3 -> 44.0 ms in fp32; 11.5 -> 10.3 and 11.9 -> 10.7 ms
    // in bf16).
    if (stem) {
        IO_REQUIRE(bn == 64, IO_ERR_SHAPE, "conv_nt(stem): Co must be 64");
        if (dt_in == IO_BF16) {
            IO_REQUIRE(dt_out == IO_BF16, IO_ERR_SHAPE, "conv_nt: bf16 operands write bf16 outputs");
            IO_LAUNCH_NT(bf16_t, bf16_t, 64, 1, 1, 4);
        } else if (dt_out == IO_BF16) {
            IO_LAUNCH_NT(float, bf16_t, 64, 1, 2, 1);
        } else if (g.cr && io_stem_rows_ok(g) && !add && !mask && !bws.y && !bws.in_scale) {
            // whole 128-pixel output rows: the row-persistent kernel of stem.hip (filter resident in LDS, A fragments read
            // straight out of a compact input patch)
            return io_launch_stem_rows(g, (const float*)in, (const float*)wgt, (float*)out, st_mean, st_m2, bws.bias,
                                       bws.relu, st);
        } else if (g.cr) {
            IO_LAUNCH_NT(float, float, 64, 2, 1, 4);      // (3.3 -> 3.1 ms against the double-buffered form)
        } else {
            IO_LAUNCH_NT(float, float, 64, 1, 2, 1);
        }
    } else if (dt_in == IO_BF16) {
        IO_REQUIRE(dt_out == IO_BF16, IO_ERR_SHAPE, "conv_nt: bf16 operands write bf16 outputs");
        if (bn == 128) IO_LAUNCH_NT(bf16_t, bf16_t, 128, 0, 1, 3);
        else IO_LAUNCH_NT(bf16_t, bf16_t, 64, 0, 1, 4);      // 64-wide: 24.6 KB, four blocks per CU
    } else {
        IO_REQUIRE(dt_out == IO_F32, IO_ERR_SHAPE, "conv_nt: fp32 operands write fp32 outputs (except the stem)");
        // a grid of 769..1024 tiles is two full rounds of the two-block kernel but 1 1/3 rounds of the three-block one:
        // the 1x1 GEMMs of the last stage (M = 32768 rows, 512 output channels) run 12 % faster on the former
        if (bn == 128 && tiles > 768 && tiles <= 1024 && g.Th * g.Tw == 1) IO_LAUNCH_NT(float, float, 128, 0, 2, 1);
        else if (bn == 128) IO_LAUNCH_NT(float, float, 128, 0, 1, 3);
        else IO_LAUNCH_NT(float, float, 64, 0, 1, 4);
    }
#undef IO_LAUNCH_NT
#undef IO_LAUNCH_NT__
#undef IO_LAUNCH_NT_
    return io_check_launch("conv_nt");
}

int io_launch_conv_wgrad(const IoConvGeom& g, const void* in, const void* dy, float* dw, float* partial,
                         size_t partial_bytes, int stem, hipStream_t st, int dt_in, int dt_dy) {
    IO_REQUIRE(g.Co % 64 == 0, IO_ERR_SHAPE, "conv_wgrad: Co=%d must be a multiple of 64", g.Co);
    IO_REQUIRE(!g.cr || (stem && dt_in == IO_F32 && g.cr > 0 && g.cr <= 8), IO_ERR_SHAPE,
               "conv_wgrad: the exact-K mode is for the fp32 stem (1..8 real channels)");
    if (stem)
        IO_REQUIRE(g.Ci == 8 && g.Co == 64 && (dt_in == IO_F32 || dt_dy == IO_BF16), IO_ERR_SHAPE,
                   "conv_wgrad(stem): need Ci=8, Co=64 (and bf16 dY with a bf16 input)");
    else
        IO_REQUIRE(g.Ci % 64 == 0 && dt_in == dt_dy, IO_ERR_SHAPE,
                   "conv_wgrad: Ci=%d must be a multiple of 64 (and one storage type)", g.Ci);
    IO_REQUIRE(g.N > 0 && g.Ho > 0 && g.Wo > 0 && g.Hi > 0 && g.Wi > 0, IO_ERR_SHAPE,
               "conv_wgrad: empty tensor (N=%d, in %dx%d, out %dx%d)", g.N, g.Hi, g.Wi, g.Ho, g.Wo);
    IO_REQUIRE(g.os == 1 && g.Ho == g.outH && g.Wo == g.outW, IO_ERR_SHAPE, "conv_wgrad: dY must be dense");
    IO_REQUIRE(!g.gw || (g.gw == 64 && !stem && g.Ci == g.Co), IO_ERR_SHAPE,
               "conv_wgrad: grouped mode needs a 64-channel window and Ci == Co");
    WgradPlan p = plan_wgrad(g, stem, dt_in == IO_F32 && dt_dy == IO_F32);
    const size_t need = io_conv_wgrad_partial_bytes(g, stem);
    IO_REQUIRE(partial_bytes >= need && (need == 0 || partial), IO_ERR_WORKSPACE,
               "conv_wgrad: workspace %zu < %zu bytes", partial_bytes, need);
    // the fp32 exact-K stem on whole 128-pixel output rows: the row-persistent kernel of stem.hip, one partial per block
    if (stem && dt_in == IO_F32 && dt_dy == IO_F32 && g.cr && io_stem_rows_ok(g))
        return io_launch_stem_wgrad_rows(g, (const float*)in, (const float*)dy, dw, partial, partial_bytes, st);
    // the bf16 stem on 128-wide output rows: patch in LDS, transposing fragment reads (conv_halo3.hip)
    if (stem && dt_in == IO_BF16 && dt_dy == IO_BF16) {
        const int rh = io_launch_stem_wgrad_halo(g, in, dy, dw, partial, partial_bytes, st, nullptr);
        if (rh <= 0) {
            g_last_wgrad_route.store(1, std::memory_order_relaxed);
            return rh;
        }
    }
    // the bf16 3x3 stride-1 64 -> 64 layer on 64-wide maps: all nine taps from one halo image (conv_halo3.hip)
    if (!stem && dt_in == IO_BF16 && dt_dy == IO_BF16) {
        const int rh = io_launch_conv_wgrad_halo3(g, in, dy, dw, partial, partial_bytes, st);
        if (rh <= 0) {
            g_last_wgrad_route.store(2, std::memory_order_relaxed);
            return rh;
        }
    }
    g_last_wgrad_route.store(0, std::memory_order_relaxed);
    if (dt_in == IO_F32 && dt_dy == IO_F32 && wgrad_wino_ok(g, stem)) {
        const WgradPlan pw = plan_wgrad_wino(g);
        float* dstw = pw.splits == 1 ? dw : partial;
        const size_t in_b = (size_t)4 * g.N * g.Hi * g.Wi * g.Ci, dy_b = (size_t)4 * g.N * g.Ho * g.Wo * g.Co;
        const double rows = (double)pw.kps * 64.0 + 64.0, samples = rows / ((double)g.Ho * g.Wo) + 2.0;
        IO_REQUIRE(rows * g.Co * 4.0 < 4.0e9 && samples * g.Hi * g.Wi * g.Ci * 4.0 < 4.0e9 && (double)g.N * g.Hi * g.Wi < 2.0e9,
                   IO_ERR_SHAPE, "conv_wgrad(wino): one split spans more than 4 GB (32-bit offsets)");
        const double Mdw = (double)g.N * g.Ho * g.Wo;
        IoProfScope prof(IO_PROF_WGRAD_WINO, 2.0 * Mdw * g.Co * 9.0 * g.Ci,
                         4.0 * (Mdw * g.Co + (double)g.N * g.Hi * g.Wi * g.Ci + 9.0 * g.Co * g.Ci), st,
                         2.0 * Mdw * g.Co * 9.0 * g.Ci * (g.Wo % 16 == 0 ? 0.5 : 2.0 / 3.0));
        if (g.Wo % 16 == 0) {          // F(4, 3): a staging thread's 16 pixels lie in one image row
            hipLaunchKernelGGL(conv_wgrad_wino4_kernel, dim3((unsigned)(pw.tiles * pw.splits)), dim3(kThreads),
                               (size_t)2 * 6 * 64 * 16 * sizeof(float), st, g, (const float*)in, (const float*)dy, dstw,
                               pw.ntile_c, pw.tiles, pw.kps, in_b, dy_b);
        } else {
        const size_t lds = (size_t)2 * 4 * 64 * 36 * sizeof(float);
        static std::atomic<unsigned long long> attr_done{0};
        if (io_first_on_device(attr_done))
            (void)hipFuncSetAttribute((const void*)conv_wgrad_wino_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        hipLaunchKernelGGL((conv_wgrad_wino_kernel<2>), dim3((unsigned)(pw.tiles * pw.splits)), dim3(kThreads), lds, st, g,
                           (const float*)in, (const float*)dy, dstw, pw.ntile_c, pw.tiles, pw.kps, in_b, dy_b);
        }
        int rcw = io_check_launch("conv_wgrad(wino)");
        if (rcw) return rcw;
        if (pw.splits > 1) rcw = io_splitk_reduce(partial, dw, (size_t)g.Co * io_filter_row(g) / 4, pw.splits, st);
        return rcw;
    }
    float* dst = p.splits == 1 ? dw : partial;
    int splits = p.splits;
    dim3 grid((unsigned)(p.tiles * p.splits)), block(kThreads);
    // descriptors are rebased per split: what one split spans must fit 32-bit offsets, not the whole tensors
    const size_t in_bytes = (size_t)io_dtype_bytes(dt_in) * g.N * g.Hi * g.Wi * g.Ci;
    const size_t dy_bytes = (size_t)io_dtype_bytes(dt_dy) * g.N * g.Ho * g.Wo * g.Co;
    {
        const double rows = (double)p.kps * 32.0 + 64.0, samples = rows / ((double)g.Ho * g.Wo) + 2.0;
        IO_REQUIRE(rows * g.Co * io_dtype_bytes(dt_dy) < 4.0e9 &&
                       samples * g.Hi * g.Wi * g.Ci * io_dtype_bytes(dt_in) < 4.0e9 &&
                       (double)g.N * g.Hi * g.Wi < 2.0e9,
                   IO_ERR_SHAPE, "conv_wgrad: one split spans more than 4 GB (32-bit offsets)");
    }
    const double Md = (double)g.N * g.Ho * g.Wo;
    const double kred = stem ? (double)g.wT * 5.0 : (double)g.Th * g.Tw * (g.gw ? g.gw : g.Ci);
    // 64-wide tiles (row layout): scalar row decode per k-tile when 32 | Wo.  Measured on the 64-channel 3x3 layer:
    // 104 -> 113 TF/s; the 1x1 stride-1 layers (HBM-bound, already without per-row divisions) gain nothing from it
    const bool lin1x1 = g.Th * g.Tw == 1 && g.is == 1 && g.dh0 == 0 && g.dw0 == 0 && g.Hi == g.Ho && g.Wi == g.Wo;
    const bool w32 = !stem && !lin1x1 && g.Wo % 32 == 0;
    IoProfScope prof(stem ? IO_PROF_WGRAD_STEM : IO_PROF_WGRAD, 2.0 * Md * g.Co * kred,
                     io_dtype_bytes(dt_dy) * Md * g.Co + io_dtype_bytes(dt_in) * (double)g.N * g.Hi * g.Wi * g.Ci +
                         4.0 * g.Co * kred, st);
#define IO_LAUNCH_WG(TX_, TDY_, BMO_, BNC_, STEM_)                                                              \
    do {                                                                                                        \
        const size_t lds = (size_t)2 * 32 * (BMO_ + BNC_) * sizeof(float);                                      \
        if (STEM_ == 0 && w32)                                                                                  \
            hipLaunchKernelGGL((conv_wgrad_kernel<TX_, TDY_, BMO_, BNC_, 0, false, true>), grid, block, lds, st, \
                               g, (const TX_*)in, (const TDY_*)dy, dst, p.ntile_c, p.tiles, p.kps, in_bytes,    \
                               dy_bytes);                                                                       \
        else                                                                                                    \
            hipLaunchKernelGGL((conv_wgrad_kernel<TX_, TDY_, BMO_, BNC_, STEM_>), grid, block, lds, st, g,      \
                               (const TX_*)in, (const TDY_*)dy, dst, p.ntile_c, p.tiles, p.kps, in_bytes,       \
                               dy_bytes);                                                                       \
    } while (0)
#define IO_LAUNCH_WGT_(BMO_, BNC_, W4_, NBUF_, MINB_)                                                           \
    do {                                                                                                        \
        const size_t lds = (size_t)NBUF_ * (BMO_ + BNC_) * 36 * sizeof(float);                                  \
        static std::atomic<unsigned long long> attr_done{0};                                                                          \
        if (io_first_on_device(attr_done)) {                                                                                       \
            (void)hipFuncSetAttribute(                                                                          \
                (const void*)conv_wgrad_kernel<float, float, BMO_, BNC_, 0, true, W4_, NBUF_, MINB_>,           \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                          \
        }                                                                                                       \
        hipLaunchKernelGGL((conv_wgrad_kernel<float, float, BMO_, BNC_, 0, true, W4_, NBUF_, MINB_>), grid,     \
                           block, lds, st, g, (const float*)in, (const float*)dy, dst, p.ntile_c, p.tiles,      \
                           p.kps, in_bytes, dy_bytes);                                                          \
    } while (0)
#define IO_LAUNCH_WGT(BMO_, BNC_)                                                                               \
    do {                                                                                                        \
        /* single LDS buffer + three blocks per CU, as in the NT kernel: 3x3 layers 124 -> 128 TF/s */          \
        if (g.Wo % 4 == 0) IO_LAUNCH_WGT_(BMO_, BNC_, true, 1, 3);                                              \
        else IO_LAUNCH_WGT_(BMO_, BNC_, false, 1, 3);                                                           \
    } while (0)
#define IO_LAUNCH_WG_SHAPES(TX_, TDY_)                                                 \
    do {                                                                               \
        /* transposed staging pays on 128 x 128 tiles (3x3 layers 112 -> 117 TF/s); with a 64-wide operand half   \
         * the threads have nothing to stage and it loses (l1 3x3: 103 -> 85 TF/s), so those keep the row layout */ \
        if (p.bmo == 128 && p.bnc == 128) IO_LAUNCH_WGT(128, 128);                     \
        else if (p.bmo == 128) IO_LAUNCH_WG(TX_, TDY_, 128, 64, 0);                    \
        else if (p.bnc == 128) IO_LAUNCH_WG(TX_, TDY_, 64, 128, 0);                    \
        else IO_LAUNCH_WG(TX_, TDY_, 64, 64, 0);                                       \
    } while (0)
    if (stem && dt_in == IO_F32) {
        if (g.cr && g.Wo % 32 == 0) {
            if (dt_dy == IO_BF16) IO_LAUNCH_WG(float, bf16_t, 64, 64, 3);
            else IO_LAUNCH_WG(float, float, 64, 64, 3);
        } else if (g.cr) {
            if (dt_dy == IO_BF16) IO_LAUNCH_WG(float, bf16_t, 64, 64, 2);
            else IO_LAUNCH_WG(float, float, 64, 64, 2);
        } else if (dt_dy == IO_BF16) IO_LAUNCH_WG(float, bf16_t, 64, 64, 1);
        else IO_LAUNCH_WG(float, float, 64, 64, 1);
    } else if (dt_in == IO_BF16) {
        // 64-row k-tiles; the split count can only shrink, so the fp32 plan's partial buffer is large enough
        const int nkt64 = io_cdiv((long)Md, 64), kps64 = io_cdiv(p.kps, 2);
        splits = io_cdiv(nkt64, kps64);
        dst = splits == 1 ? dw : partial;
        dim3 grid1((unsigned)(p.tiles * splits));
        const bool w8 = g.Wo % 8 == 0;       // measured: 3x3 / strided layers 400-470 -> 490-540 TF/s
#define IO_LAUNCH_WGB_(BMO_, BNC_, STEM_, W8_, NBUF_, MINB_)                                                     \
    do {                                                                                                         \
        const size_t lds = (size_t)NBUF_ * (BMO_ + BNC_) * 36 * sizeof(float);                                   \
        static std::atomic<unsigned long long> attr_done{0};                                                                           \
        if (io_first_on_device(attr_done)) {                                                                                        \
            (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<BMO_, BNC_, STEM_, W8_, NBUF_, MINB_>, \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
        }                                                                                                        \
        hipLaunchKernelGGL((conv_wgrad_bf16_kernel<BMO_, BNC_, STEM_, W8_, NBUF_, MINB_>), grid1, block, lds,    \
                           st, g, (const bf16_t*)in, (const bf16_t*)dy, dst, p.ntile_c, p.tiles, kps64,          \
                           in_bytes, dy_bytes);                                                                  \
    } while (0)
#define IO_LAUNCH_WGB(BMO_, BNC_, STEM_)                                                                         \
    do {                                                                                                         \
        /* (single-buffered with three blocks per CU, what the other GEMMs run, measured 15 % slower here) */    \
        if (!STEM_ && w8) IO_LAUNCH_WGB_(BMO_, BNC_, STEM_, !STEM_, 2, 1);                                       \
        else IO_LAUNCH_WGB_(BMO_, BNC_, STEM_, false, 2, 1);                                                     \
    } while (0)
        // LDS-DMA + transpose-read form where its shape conditions hold (see the kernel)
        const int rb_rows = 512 / p.bnc;
        const bool trk = !(stem && g.cr) && (long)Md % 64 == 0 && (g.Ho * g.Wo) % 64 == 0 &&
                         g.Wo % rb_rows == 0;
#define IO_LAUNCH_WGTR(BMO_, BNC_, STEM_)                                                                               \
    do {                                                                                                         \
        const size_t lds = (size_t)kTrStages * (BMO_ + BNC_) * kTrBkm * 2;                                    \
        static std::atomic<unsigned long long> attr_done{0};                                                                           \
        if (io_first_on_device(attr_done)) {                                                                                        \
            (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_tr_kernel<BMO_, BNC_, STEM_, kTrMinB, kTrBkm, kTrStages>, \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
        }                                                                                                        \
        hipLaunchKernelGGL((conv_wgrad_bf16_tr_kernel<BMO_, BNC_, STEM_, kTrMinB, kTrBkm, kTrStages>), grid1, block,  \
                           lds, st, g, (const bf16_t*)in, (const bf16_t*)dy, dst, p.ntile_c, p.tiles,            \
                           kps64 * (64 / kTrBkm), in_bytes, dy_bytes);                                        \
    } while (0)
        if (trk && stem) IO_LAUNCH_WGTR(64, 64, true);
        else if (trk && p.bmo == 128 && p.bnc == 128) IO_LAUNCH_WGTR(128, 128, false);
        else if (trk && p.bmo == 128) IO_LAUNCH_WGTR(128, 64, false);
        else if (trk && p.bnc == 128) IO_LAUNCH_WGTR(64, 128, false);
        else if (trk) IO_LAUNCH_WGTR(64, 64, false);
        else if (stem) IO_LAUNCH_WGB(64, 64, true);
        else if (p.bmo == 128 && p.bnc == 128) IO_LAUNCH_WGB(128, 128, false);
        else if (p.bmo == 128) IO_LAUNCH_WGB(128, 64, false);
        else if (p.bnc == 128) IO_LAUNCH_WGB(64, 128, false);
        else IO_LAUNCH_WGB(64, 64, false);
#undef IO_LAUNCH_WGTR
#undef IO_LAUNCH_WGB
#undef IO_LAUNCH_WGB_
    } else {
        IO_LAUNCH_WG_SHAPES(float, float);
    }
#undef IO_LAUNCH_WG_SHAPES
#undef IO_LAUNCH_WGT
#undef IO_LAUNCH_WGT_
#undef IO_LAUNCH_WG
    int rc = io_check_launch("conv_wgrad");
    if (rc) return rc;
    if (splits > 1) rc = io_splitk_reduce(partial, dw, (size_t)g.Co * io_filter_row(g) / 4, splits, st);
    return rc;
}

int io_splitk_reduce(const float* partial, float* dst, size_t n4, int splits, hipStream_t st) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(io_cdiv((long)n4, 32)), dim3(256), 0, st, partial, dst, n4, splits);
    return io_check_launch("splitk_reduce");
}
