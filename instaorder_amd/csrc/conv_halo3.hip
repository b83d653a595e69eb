// bf16 3x3 stride-1 same-size convolutions / data gradients with 64 output channels (conv2 of the layer-1 Bottlenecks and
// its data gradient, models/backbone/resnet_cls.py:23-26, 88) with the input staged ONCE per tile as a halo image.
//
// The implicit-GEMM kernels (conv_nt_kernel, conv_p256.hip) fetch the A operand once per filter tap: nine times the input
// through L2 and LDS.  On the wide layers that hides behind the matrix pipe; with 64 output channels a k-tile is 64 MFMAs per
// CU and the nine-fold operand traffic -- and the per-k-tile barrier, fetch issue and scalar bookkeeping that come with it --
// is what the launch waits for (layer 1 at the bench batch: 0.287 ms forward, 0.378 ms data gradient against 0.09 / 0.14 ms of
// HBM time and 0.06 ms of matrix time).  Here a tile is 256 output pixels = R = 256 / W whole rows of one image times all 64
// output channels, and its input -- rows h0 - 1 .. h0 + R, columns -1 .. W, zeros outside the image -- goes to LDS once per
// 64-channel chunk ((R + 2)(W + 2) pixels x 128 bytes, by LDS-DMA, XOR-swizzled); the nine taps read it at nine row offsets.
// Blocks are persistent (one per CU, 8 waves): the halo image of the next (tile, channel chunk) is in flight while the current
// one is multiplied (two A buffers), and the output leaves as 16-byte row pieces through LDS, as in conv_p256.hip.
//   64 input channels (the ResNet case): the FILTERS LIVE IN REGISTERS -- 36 k-steps x one 32-column fragment = 144 VGPRs
//   per lane, loaded once per block.  Only halo images stream, a tile is ONE barrier interval of 72 MFMAs per wave, and the
//   A fragments are requested three k-steps ahead.  Measured on the bench step (256 pairs, 512 x 64 x 64 x 64): forward with
//   statistics 0.287 -> 0.174 ms, data gradient with the BatchNorm-backward epilogue 0.378 -> 0.316 ms.
//   128 input channels: the filter taps stream through a two-stage LDS ring, three taps per stage.
//   EPI_STATS  per-(128-row tile, channel) mean / M2 of the output (training forward);
//   EPI_BWE    fused BatchNorm-backward epilogue with the ReLU mask recomputed from y (IoBwStats: the data gradient of conv2);
//   EPI_PLAIN  optional bias (+ ReLU) (+ add).
// A wave owns 64 rows x 32 columns (8 waves = 4 x 2): the two waves of a 128-row statistics tile combine their halves
// through LDS (Chan's update for the statistics, plain sums for the BatchNorm-backward partials).
// (What did NOT help, measured: a second accumulator set per wave against dependent-MFMA pacing; 128 output channels --
// 0.190 against conv_p256's 0.184 ms; spreading the fetches through the k-steps instead of issuing them behind the barrier.)
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
__device__ __forceinline__ u32x4 dma_rsrc_raw(unsigned long long a, size_t bytes) {
    const u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes, 0x00020000u};
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const void* p, size_t base, size_t total) {
    const size_t rest = total > base ? total - base : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p)) + base, 0,
                                             rest > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)rest, 0x00020000);
}
__device__ __forceinline__ void dma16(u32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr)
                 : "memory");
}
template <int N> __device__ __forceinline__ void dma_wait_left() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_BWE = 2 };

struct Halo3Args {
    const bf16_t* in;
    const bf16_t* wgt;
    bf16_t* out;
    const bf16_t* add;
    size_t in_bytes, out_bytes;
    unsigned w_bytes;
    float *st_mean, *st_m2;
    IoBwStats bw;
    int ntiles, tiles_per_img;
};

// The epilogue of one 256-row x BN tile (rows m0 .. m0 + 255 of the output matrix) out of the waves' accumulators: statistics
// or BatchNorm-backward partials per 128-row tile, then the rows as 16-byte pieces through the LDS scratch `scr` (SCRB bytes,
// free once every wave has passed the barrier at the top).  D layout of a 32 x 32 MFMA tile: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  LTAB: the mean / scale / shift rows of the BatchNorm-backward epilogue are
// read per pass from `ltab` ([3][BN] floats of LDS) instead of living in 24 registers.
template <int TP, int BN, int WM, int WN, int EPI, bool LTAB, int SCRB>
__device__ __forceinline__ void halo_epilogue(f32x16 (&acc)[TP / WM / 32][BN / WN / 32], float* scr, float* ltab,
                                              const IoConvGeom& g, const Halo3Args& a, int m0, int tid) {
    constexpr int NW = WM * WN, TM = TP / WM / 32, TN = BN / WN / 32, ABUF = SCRB;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int mt128 = (m0 >> 7) + (wm * (TP / WM)) / 128;          // the statistics tile this wave's rows belong to
    constexpr int WPT = 128 / (TP / WM);                            // waves (along rows) per statistics tile
    if constexpr (LTAB) {
        if (tid < 3 * BN) {
            const int which = tid / BN, c = tid - which * BN;
            const float* src = which == 0 ? a.bw.mean : (which == 1 ? a.bw.mscale : a.bw.mshift);
            ltab[tid] = src[(m0 / a.bw.Mg) * g.Co + c];
        }
    }
    __syncthreads();                     // every wave is done with the A buffer just read: it becomes scratch
    if constexpr (EPI == EPI_STATS) {
        // per wave: mean / M2 of its TP / WM rows per column; the WPT waves of a tile merged with Chan's update
        constexpr int RW = TP / WM;
        float wmean[TN], wm2[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
            s += __shfl_xor(s, 32, 64);
            wmean[j] = s * (1.0f / RW);
            float d2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) d2 += (acc[i][j][r] - wmean[j]) * (acc[i][j][r] - wmean[j]);
            d2 += __shfl_xor(d2, 32, 64);
            wm2[j] = d2;
            if (lane < 32) {
                scr[(wave * TN + j) * 64 + lane] = wmean[j];
                scr[(wave * TN + j) * 64 + 32 + lane] = d2;
            }
        }
        __syncthreads();
        if (wm % WPT == 0 && lane < 32) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float n = (float)RW, mu = wmean[j], m2 = wm2[j];
#pragma unroll
                for (int k = 1; k < WPT; ++k) {
                    const int ow = (wm + k) * WN + wn;
                    const float mb = scr[(ow * TN + j) * 64 + lane], m2b = scr[(ow * TN + j) * 64 + 32 + lane];
                    const float tot = n + (float)RW, delta = mb - mu;
                    mu += delta * ((float)RW / tot);
                    m2 += m2b + delta * delta * (n * (float)RW / tot);
                    n = tot;
                }
                const size_t o = (size_t)mt128 * g.Co + wn * TN * 32 + j * 32 + lane;
                a.st_mean[o] = mu;
                a.st_m2[o] = m2;
            }
        }
        __syncthreads();
    }
    // rows through LDS: 16 rows x WC columns of the wave at a time -> 8 consecutive channels of a row per lane
    constexpr int WC = TN * 32, EPP = WC + 4, ER = 16;
    constexpr int LPR = WC / 8, RPI = 64 / LPR, NI = (ER + RPI - 1) / RPI;
    static_assert(NW * ER * EPP * 4 + NW * 2 * LPR * 8 * 4 <= ABUF, "scratch");
    float* ep = scr + wave * (ER * EPP);
    float* red = scr + NW * ER * EPP;                               // [wave][2 sums][LPR lanes][8] for the BWE partials
    const int ecol = wn * WC + (lane % LPR) * 8;
    const size_t out_base = (size_t)m0 * (size_t)(g.Co * 2);
    const __amdgpu_buffer_rsrc_t rs_out = rsrc_at(a.out, out_base, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rs_add = rsrc_at(a.add ? (const void*)a.add : (const void*)a.out, out_base,
                                                  a.add ? a.out_bytes : out_base);
    const __amdgpu_buffer_rsrc_t rs_y = rsrc_at(EPI == EPI_BWE ? a.bw.y : (const void*)a.out, out_base,
                                                EPI == EPI_BWE ? a.out_bytes : out_base);
    float t_mu[8], t_sc[8], t_sh[8], s1[8], s2[8], t_bias[8];
    const int gcol = EPI == EPI_BWE ? (m0 / a.bw.Mg) * g.Co + ecol : 0;
    if constexpr (EPI == EPI_BWE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (!LTAB) {
                t_mu[e] = a.bw.mean[gcol + e];
                t_sc[e] = a.bw.mscale[gcol + e];
                t_sh[e] = a.bw.mshift[gcol + e];
            }
            s1[e] = s2[e] = 0.f;
        }
    }
    if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) t_bias[e] = a.bw.bias ? a.bw.bias[ecol + e] : 0.f;
    }
    constexpr int NP = TM * 2 * NI, PD = (EPI == EPI_STATS || LTAB) ? 1 : 3;
    auto pass_row = [&](int p) -> int {
        const int i = p / (2 * NI), h = (p / NI) & 1, k = p % NI;
        return wm * (TP / WM) + i * 32 + h * 16 + k * RPI + lane / LPR;
    };
    const bool rowok = NI * RPI == ER || (lane / LPR) < ER;          // (RPI = 16 = ER here: always true)
    u32x4 pav[PD], pyv[PD];
    auto pass_load = [&](int p, int sl) {
        const unsigned off = (unsigned)(pass_row(p) * g.Co + ecol) * 2u;
        if constexpr (EPI == EPI_PLAIN) pav[sl] = __builtin_amdgcn_raw_buffer_load_b128(rs_add, off, 0, 0);
        if constexpr (EPI == EPI_BWE) pyv[sl] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, off, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < NP) pass_load(d, d);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = p / (2 * NI), h = (p / NI) & 1, k = p % NI;
        if (k == 0) {
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
        const unsigned off = (unsigned)(pass_row(p) * g.Co + ecol) * 2u;
        const int sl = p % PD;
        if constexpr (EPI == EPI_PLAIN) {
            const u32x4 av = pav[sl];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                v[2 * d] += bf_lo(av[d]);
                v[2 * d + 1] += bf_hi(av[d]);
            }
            if (a.bw.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] += t_bias[e];
                    v[e] = (a.bw.relu && v[e] < 0.f) ? 0.f : v[e];
                }
            }
        }
        if constexpr (EPI == EPI_BWE) {
            const u32x4 yv = pyv[sl];
            int tc = ecol;                   // (opaque per pass: the table reads stay inside the pass)
            if constexpr (LTAB) asm volatile("" : "+v"(tc));
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int e = 2 * d + hh;
                    const float y = hh ? bf_hi(yv[d]) : bf_lo(yv[d]);
                    const float yc = y - (LTAB ? ltab[tc + e] : t_mu[e]);                 // (centred: also what the second sum takes)
                    const float t = LTAB ? __builtin_fmaf(yc, ltab[BN + tc + e], ltab[2 * BN + tc + e])
                                         : __builtin_fmaf(yc, t_sc[e], t_sh[e]);              // bn(y), bn_apply's fma
                    v[e] = t > 0.f ? v[e] : 0.f;
                    s1[e] += v[e];
                    s2[e] = __builtin_fmaf(v[e], yc, s2[e]);      // sum(dz (y - mean)), per element as conv_nt_kernel: no cancellation
                }
        }
        if (p + PD < NP) pass_load(p + PD, sl);
        const u32x4 pk = {io_f2bf2(v[0], v[1]), io_f2bf2(v[2], v[3]), io_f2bf2(v[4], v[5]), io_f2bf2(v[6], v[7])};
        if (rowok) __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, off, 0, 0);
    }
    if constexpr (EPI == EPI_BWE) {
        // lanes with the same lane % LPR hold the same 8 channels: sum them, then the WPT waves of the statistics tile
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int sft = LPR; sft < 64; sft <<= 1) {
                s1[e] += __shfl_xor(s1[e], sft, 64);
                s2[e] += __shfl_xor(s2[e], sft, 64);
            }
        if (lane < LPR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[((wave * 2 + 0) * LPR + lane) * 8 + e] = s1[e];
                red[((wave * 2 + 1) * LPR + lane) * 8 + e] = s2[e];
            }
        }
        __syncthreads();
        if (wm % WPT == 0 && lane < LPR) {
            const size_t o = (size_t)mt128 * g.Co + ecol;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t1 = s1[e], t2 = s2[e];
#pragma unroll
                for (int k = 1; k < WPT; ++k) {
                    const int ow = (wm + k) * WN + wn;
                    t1 += red[((ow * 2 + 0) * LPR + lane) * 8 + e];
                    t2 += red[((ow * 2 + 1) * LPR + lane) * 8 + e];
                }
                a.bw.p1[o + e] = t1;
                a.bw.p2[o + e] = a.bw.rstd[gcol + e] * t2;
            }
        }
    }
}

// W: map width (32 | W, W | 256).  BN = Co (64 or 128).  WM x WN = 8 waves of (256 / WM) x (BN / WN).  TPS: filter taps per
// B stage (1 or 3) -- or 0: the filters live in REGISTERS (64 input channels only: 36 k-steps x TN fragments per lane, loaded
// once per block), nothing but the halo images streams, and a tile is one barrier interval instead of 9 / TPS.
template <int W, int BN, int WM, int WN, int TPS, int EPI>
__global__ __launch_bounds__(512, 2) void conv_halo3_kernel(IoConvGeom g, Halo3Args a) {
    constexpr int TP = 256, R = TP / W, NW = 8;
    static_assert(WM * WN == NW && TP % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave layout");
    constexpr bool BREG = TPS == 0;
    constexpr int TPI = BREG ? 9 : TPS;                    // taps per barrier interval
    constexpr int TM = TP / WM / 32, TN = BN / WN / 32;
    constexpr int WP = W + 2, HP = (R + 2) * WP;           // halo image: (R + 2) rows of W + 2 pixels
    constexpr int NCH = (HP + 7) / 8;                      // 8-pixel DMA chunks of it
    constexpr int NAW = (NCH + NW - 1) / NW;               // ... per wave (every wave issues exactly NAW: see the counted wait)
    constexpr int ABUF = NAW * NW * 1024;                  // bytes of one A buffer (padded to whole chunks per wave)
    constexpr int BST = TPS * BN * 128;                    // bytes of one B stage
    constexpr int NBW = BREG ? 1 : TPS * BN / 8 / NW;      // B chunks per wave and stage
    static_assert(TPS * BN % (8 * NW) == 0, "B chunks per wave");
    constexpr int NST = 9 / TPI;                           // B stages (barrier intervals) per channel chunk
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // LDS: [A buffer 0][A buffer 1][B stage 0][B stage 1]
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ncc = g.Ci / 64;                             // channel chunks
    const int r8 = lane >> 3, p8 = lane & 7;
    const unsigned rowB = (unsigned)(g.wT * g.Ci * 2);
    const unsigned slot_e = (unsigned)((p8 ^ (r8 >> 1)) << 4), slot_o = slot_e ^ 64u;
    const unsigned vb_e = (unsigned)r8 * rowB + slot_e, vb_o = (unsigned)r8 * rowB + slot_o;
    const u32x4 rsB = dma_rsrc_raw((unsigned long long)a.wgt, a.w_bytes);

    // ---- A fetches: halo pixel hp = chunk * 8 + r8 of the image is (hy, hx) = (hp / WP, hp % WP) -> input pixel
    // (h0 - 1 + hy, hx - 1); the decomposition does not depend on the tile
    const unsigned padpx = (unsigned)(W + 1);              // the descriptor of a tile starts this many pixels before its first
    unsigned arel[NAW];       // byte offset of the lane's pixel relative to that start (+ slot): a multiple of 16, with hy in
                              // the low four bits -- 15 where the pixel does not exist / is left-right padding
    static_assert(R + 2 < 15, "hy code");
#pragma unroll
    for (int u = 0; u < NAW; ++u) {
        const int ch = wave * NAW + u, hp = ch * 8 + r8;
        const int hy = hp / WP, hx = hp - hy * WP;
        const bool colok = hp < HP && hx >= 1 && hx <= W;
        arel[u] = ((unsigned)(((hy - 1) * W + (hx - 1) + (int)padpx) * g.Ci * 2) + ((ch & 1) ? slot_o : slot_e)) |
                  (colok ? (unsigned)hy : 15u);
    }
    // chunks u0 .. u1 - 1 of the wave's share of A item (tile, cc) into A buffer `buf`
    auto issue_a = [&](int tile, int cc, int buf, int u0, int u1) {
        const int img = tile / a.tiles_per_img, h0 = (tile - img * a.tiles_per_img) * R;
        // (address arithmetic only for the pixels before the tensor: lanes that would read there are out of the image)
        const long long start = ((long long)img * g.Hi * g.Wi + (long long)h0 * W - (long long)padpx) * (long long)(g.Ci * 2);
        const u32x4 rsA = dma_rsrc_raw((unsigned long long)((const char*)a.in + start),
                                       (size_t)((long long)a.in_bytes - start));
        const unsigned sb = lds0 + (unsigned)(buf * ABUF);
        const unsigned coff = (unsigned)(cc * 128);
#pragma unroll
        for (int u = 0; u < NAW; ++u) {
            if (u < u0 || u >= u1) continue;
            const unsigned hy = arel[u] & 15u;
            const bool ok = hy != 15u && (unsigned)(h0 - 1 + (int)hy) < (unsigned)g.Hi;
            dma16(rsA, sb + (unsigned)((wave * NAW + u) * 1024), ok ? (arel[u] & ~15u) + coff : kOob, 0u);
        }
    };
    auto issue_b = [&](int cc, int st, int stage) {       // the TPS taps of B stage `st` of channel chunk cc
        const unsigned sb = lds0 + (unsigned)(2 * ABUF + stage * BST);
#pragma unroll
        for (int u = 0; u < NBW; ++u) {
            const int ch = wave * NBW + u;                 // chunk of the stage: tap ch / (BN / 8), rows (ch % (BN / 8)) * 8 ..
            const int t = ch / (BN / 8), rc = ch - t * (BN / 8);
            const int tap = st * TPS + t, th = tap / 3, tw = tap - th * 3;
            const int widx = (g.r0 + g.rs * th) * g.S + (g.s0 + g.ss * tw);
            dma16(rsB, sb + (unsigned)(ch * 1024), (rc & 1) ? vb_o : vb_e,
                  (unsigned)((widx * g.Ci + cc * 64) * 2) + (unsigned)(rc * 8) * rowB);
        }
    };

    // ---- fragment reads
    // A: MFMA row tile i of the wave = 32 consecutive output pixels of one image row: halo pixel hpb[i] + (lane & 31) + tap shift
    int hpb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = wm * (TP / WM) + i * 32;             // tile-local pixel
        hpb[i] = (p / W + 1) * WP + (p % W) + 1 + (lane & 31);
    }
    const int swzb = (lane >> 1) & 7;                      // B rows: row & 31 = lane & 31
    unsigned kob[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kob[kk] = (unsigned)(((2 * kk + (lane >> 5)) ^ swzb) << 4);
    const unsigned b_row = (unsigned)((wn * TN * 32 + (lane & 31)) * 128);

    f32x16 acc[TM][TN];
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    int cc = 0, abuf = 0, bstage = 0;
    // BREG: the B fragments of all 36 k-steps (tap s / 4, channels 16 (s % 4) ..): lane l holds filter row wn TN 32 + 32 j +
    // (l & 31), channels 8 (2 kk + (l >> 5)) .. + 7 of tap t -- the MFMA's B layout, straight from global memory
    bf16x8 breg[BREG ? 36 * TN : 1];
    if constexpr (BREG) {
        const __amdgpu_buffer_rsrc_t rw = rsrc_at(a.wgt, 0, a.w_bytes);
#pragma unroll
        for (int s_ = 0; s_ < 36; ++s_) {
            const int t = s_ / 4, kk = s_ % 4, th = t / 3, tw = t - th * 3;
            const int widx = (g.r0 + g.rs * th) * g.S + (g.s0 + g.ss * tw);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned off = (unsigned)(wn * TN * 32 + j * 32 + (lane & 31)) * rowB +
                                     (unsigned)((widx * g.Ci + (2 * kk + (lane >> 5)) * 8) * 2);
                breg[s_ * TN + j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
            }
        }
    }
    // The stream of B stages (tile, cc, st), one per iteration; A item (tile, cc) is read from st == 0 of its chunk on.
    // Issue order inside iteration k: [B stage k + 1][a piece of the NEXT A item].  The wait of iteration k + 1 needs stage
    // k + 1 and everything older, i.e. it may leave exactly the piece issued after it outstanding (`pend` fetches): a piece
    // has two iterations to land, a B stage one.  The pieces go out in the first NST - 1 iterations of a chunk, so the
    // wait at st == 0 of the next chunk (pend == 0) covers the whole item.  (BREG: one iteration per tile, the next tile's
    // image issued whole behind the barrier.)
    constexpr int NPC = BREG ? 1 : NST - 1, PMAX = (NAW + NPC - 1) / NPC;       // iterations that carry a piece; fetches per piece
    static_assert(BREG || PMAX <= 4, "piece size");
    if (tile < a.ntiles) {
        issue_a(tile, 0, 0, 0, NAW);
        if constexpr (!BREG) issue_b(0, 0, 0);
    }
    int st = 0, pend = 0;
    while (tile < a.ntiles) {
        if (cc == 0 && st == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        if (BREG || pend == 0) dma_wait_left<0>();
        else if (pend == 1) dma_wait_left<1>();
        else if (pend == 2) dma_wait_left<2>();
        else if (pend == 3) dma_wait_left<3>();
        else dma_wait_left<4>();
        __syncthreads();
        // what comes next in the stream
        const bool last_st = st + 1 == NST, last_cc = cc + 1 == ncc;
        const int nst = last_st ? 0 : st + 1;
        const int ncc_ = last_st ? (last_cc ? 0 : cc + 1) : cc;
        const int ntile = (last_st && last_cc) ? tile + (int)gridDim.x : tile;
        if constexpr (!BREG) {
            if (ntile < a.ntiles) issue_b(ncc_, nst, bstage ^ 1);
        }
        pend = 0;
        if (BREG || st < NPC) {
            // the A item AFTER the current one goes into the other A buffer (free: its last reader was the previous chunk)
            const int acc_ = last_cc ? 0 : cc + 1;
            const int atile = last_cc ? tile + (int)gridDim.x : tile;
            const int u0 = BREG ? 0 : (NAW * st) / NPC, u1 = BREG ? NAW : (NAW * (st + 1)) / NPC;
            if (atile < a.ntiles) {
                issue_a(atile, acc_, abuf ^ 1, u0, u1);
                pend = u1 - u0;
            }
        }
        const char* sa = smem + abuf * ABUF;
        const char* sbp = smem + 2 * ABUF + bstage * BST;
        // TPI x 4 k-steps of 16 channels; the fragments of step s + PF are requested before the products of step s (a ring of
        // PF + 1 register sets: with TM x TN = 2 .. 4 MFMAs per step the LDS latency spans several steps)
        constexpr int S = TPI * 4, PF = 3;
        // (the row bases pass through an empty asm per iteration: with the filters in registers nothing else in the k-steps
        // varies from tile to tile, and the compiler would otherwise keep all 72 fragment addresses in registers across tiles)
        int hb[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            hb[i] = hpb[i];
            asm volatile("" : "+v"(hb[i]));
        }
        int shs[TPI];                                        // halo-pixel shift of each tap (uniform)
#pragma unroll
        for (int t = 0; t < TPI; ++t) {
            const int tap = st * TPI + t, th = tap / 3, tw = tap - th * 3;
            shs[t] = (g.dh0 + g.dhs * th) * WP + (g.dw0 + g.dws * tw);
        }
        bf16x8 fa[PF + 1][TM], fb[BREG ? 1 : PF + 1][TN];
        auto frag_load = [&](int s_, int slot) {
            const int t = s_ / 4, kk = s_ % 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned hp = (unsigned)(hb[i] + shs[t]);
                fa[slot][i] = *reinterpret_cast<const bf16x8*>(sa + hp * 128u + ((((unsigned)(2 * kk + (lane >> 5))) ^ ((hp >> 1) & 7u)) << 4));
            }
            if constexpr (!BREG) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[slot][j] = *reinterpret_cast<const bf16x8*>(sbp + t * (BN * 128) + b_row + j * 4096 + kob[kk]);
            }
        };
#pragma unroll
        for (int s_ = 0; s_ < PF; ++s_) frag_load(s_, s_ % (PF + 1));
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            if (s_ + PF < S) frag_load(s_ + PF, (s_ + PF) % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);       // (left alone the scheduler sinks every read to just before its use)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_ % (PF + 1)][i], BREG ? breg[s_ * TN + j] : fb[s_ % (PF + 1)][j],
                                                                        acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (last_st && last_cc) {
            // (filters in registers: no room for the 24 table values of the BatchNorm-backward epilogue next to them -- the
            // tile's mean / scale / shift rows go to LDS behind the A buffers and are read per pass)
            halo_epilogue<TP, BN, WM, WN, EPI, BREG && EPI == EPI_BWE, ABUF>(
                acc, reinterpret_cast<float*>(smem + abuf * ABUF), reinterpret_cast<float*>(smem + 2 * ABUF + 2 * BST), g, a,
                tile * TP, tid);
            // (the next iteration's barrier orders the scratch uses before the buffer is refilled)
        }
        // advance the stream
        if (last_st) abuf ^= 1;
        bstage ^= 1;
        st = nst;
        cc = ncc_;
        tile = ntile;
    }
}


// ---- the stem (7x7 stride 2 pad 3 on the packed 8-channel input -> 64 channels; resnet_cls.py:155 `conv1`) ----------------
// The same recipe: a tile is 256 output pixels = TWO output rows of one image (Wo = 128), whose 9 input rows x 261 input
// pixels x 16 bytes go to LDS once, split by pixel PARITY -- plane (row, parity) holds pixels ix = 2 j + parity - 4, j = 0 ..
// 131 -- so that the stride-2 walk of 32 consecutive output pixels reads 32 consecutive 16-byte entries (conflict-free).  A
// pixel's 8 channels are one MFMA operand chunk; a k-step is two filter taps (lane half = tap parity), 25 steps for the 49
// taps (the 50th has a zero filter and re-reads tap 48's pixels).  The filters live in registers (25 fragments = 100 VGPRs per
// lane), one barrier interval per tile, epilogue shared with conv_halo3_kernel.
// Measured at the bench batch (512 x 256 x 256): see DESIGN.md (the 64-wide implicit-GEMM kernel it replaces: 0.89 ms).
template <int EPI>
__global__ __launch_bounds__(512, 2) void stem_halo_kernel(IoConvGeom g, Halo3Args a) {
    constexpr int TP = 256, BN = 64, WM = 4, WN = 2, NW = 8, TM = 2;
    constexpr int PJ = 132, ROWS = 9, SLOTS = ROWS * 2 * PJ;       // 16-byte entries of one patch
    constexpr int NCH = (SLOTS + 63) / 64, NAW = (NCH + NW - 1) / NW, ABUF = NAW * NW * 1024;
    constexpr int S = 25, PF = 3;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hh = lane >> 5;
    // ---- fetches: entry q = chunk * 64 + lane of the patch is (plane rp = q / PJ, j = q % PJ) -> input row 2 oy0 - 3 + (rp >> 1),
    // pixel 2 j + (rp & 1) - 4.  arel: byte offset from the tile's descriptor start (pixel (2 oy0 - 3, -4)), the patch row in
    // the low four bits (15: no such pixel)
    unsigned arel[NAW];
#pragma unroll
    for (int u = 0; u < NAW; ++u) {
        const int q = (wave * NAW + u) * 64 + lane;
        const int rp = q / PJ, j = q - rp * PJ, rr = rp >> 1, ix = 2 * j + (rp & 1) - 4;
        const bool ok = q < SLOTS && ix >= 0 && ix < g.Wi;
        arel[u] = ((unsigned)((rr * g.Wi + ix + 4) * 16)) | (ok ? (unsigned)rr : 15u);
    }
    const int tpi = g.Ho >> 1;                              // tiles per image
    auto issue_a = [&](int tile, int buf) {
        const int img = tile / tpi, oy0 = (tile - img * tpi) * 2;
        const long long start = (((long long)img * g.Hi + (2 * oy0 - 3)) * g.Wi - 4) * 16;
        const u32x4 rsA = dma_rsrc_raw((unsigned long long)((const char*)a.in + start), (size_t)((long long)a.in_bytes - start));
        const unsigned sb = lds0 + (unsigned)(buf * ABUF);
#pragma unroll
        for (int u = 0; u < NAW; ++u) {
            const unsigned rr = arel[u] & 15u;
            const bool ok = rr != 15u && (unsigned)(2 * oy0 - 3 + (int)rr) < (unsigned)g.Hi;
            dma16(rsA, sb + (unsigned)((wave * NAW + u) * 1024), ok ? (arel[u] & ~15u) : kOob, 0u);
        }
    };
    // ---- the filters: lane (column wn 32 + l31, tap parity hh) holds the 8 channels of tap 2 s + hh for every k-step s
    bf16x8 breg[S];
    {
        const __amdgpu_buffer_rsrc_t rw = rsrc_at(a.wgt, 0, a.w_bytes);
        const unsigned rowB = (unsigned)(g.wT * 8 * 2);
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            const int t0 = 2 * s_, t1 = 2 * s_ + 1 < 49 ? 2 * s_ + 1 : -1;
            const int w0 = (g.r0 + g.rs * (t0 / 7)) * g.S + (g.s0 + g.ss * (t0 % 7));
            const int w1 = t1 < 0 ? 0 : (g.r0 + g.rs * (t1 / 7)) * g.S + (g.s0 + g.ss * (t1 % 7));
            const unsigned off = (unsigned)(wn * 32 + l31) * rowB + (unsigned)((hh ? w1 : w0) * 16);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, (t1 < 0 && hh) ? kOob : off, 0, 0);      // (out of range: zeros)
            breg[s_] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // A fragments: output pixel p = wm 64 + 32 i + l31 of the tile = (row p >> 7, column p & 127); under tap (ty, tx) it reads
    // entry ((2 (p >> 7) + ty) 2 + (tx + 1 & 1)) PJ + (p & 127) + (tx + 1 >> 1) -- a per-lane base plus a per-tap constant
    int ab[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pp = wm * 64 + i * 32 + l31;
        ab[i] = ((4 * (pp >> 7)) * PJ + (pp & 127)) * 16;
    }
    f32x16 acc[TM][1];
    int tile = xcd_remap(blockIdx.x, gridDim.x), abuf = 0;
    if (tile < a.ntiles) issue_a(tile, 0);
    while (tile < a.ntiles) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        dma_wait_left<0>();
        __syncthreads();
        const int ntile = tile + (int)gridDim.x;
        if (ntile < a.ntiles) issue_a(ntile, abuf ^ 1);
        const char* sa = smem + abuf * ABUF;
        int hb[TM];                                          // (opaque per tile, as in conv_halo3_kernel)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            hb[i] = ab[i];
            asm volatile("" : "+v"(hb[i]));
        }
        bf16x8 fa[PF + 1][TM];
        auto frag_load = [&](int s_, int slot) {
            const int t0 = 2 * s_, t1 = 2 * s_ + 1 < 49 ? 2 * s_ + 1 : 48;
            const int c0 = ((2 * (t0 / 7) + ((t0 % 7 + 1) & 1)) * PJ + ((t0 % 7 + 1) >> 1)) * 16;
            const int c1 = ((2 * (t1 / 7) + ((t1 % 7 + 1) & 1)) * PJ + ((t1 % 7 + 1) >> 1)) * 16;
            const int cs = hh ? c1 : c0;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const bf16x8*>(sa + hb[i] + cs);
        };
#pragma unroll
        for (int s_ = 0; s_ < PF; ++s_) frag_load(s_, s_ % (PF + 1));
#pragma unroll
        for (int s_ = 0; s_ < S; ++s_) {
            if (s_ + PF < S) frag_load(s_ + PF, (s_ + PF) % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s_ % (PF + 1)][i], breg[s_], acc[i][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        halo_epilogue<TP, BN, WM, WN, EPI, false, ABUF>(acc, reinterpret_cast<float*>(smem + abuf * ABUF), nullptr, g, a, tile * TP,
                                                        tid);
        abuf ^= 1;
        tile = ntile;
    }
}


// ---- the stem's filter gradient (bf16), with bn1's backward folded in ------------------------------------------------------
// dW[co][tap][c] = sum over output pixels of dy[px][co] * x8[2 px + tap][c]: the reduction index of the MFMA is the PIXEL, so
// both operands are needed with 8 consecutive pixels per lane -- the transpose of what memory holds.  `ds_read_b64_tr_b16`
// delivers exactly that from pixel-major LDS images (per 16-lane group: lane 4 j + q points at 4 channels of pixel j, lane c
// receives channel c of the four pixels), and because every lane brings its own address the "16 columns" of a read can be
// two filter taps x 8 channels taken from two different places of the parity-split patch of stem_halo_kernel.  Tile = two
// output rows (256 pixels) of one image: the patch by LDS-DMA as in the forward kernel, the 256 x 64 dy rows through
// registers -- because with XB they are not read but COMPUTED while staged: dy = a * (dz where relu(bn1(y)) > 0) + b * y + c
// with the [G][64] tables of io_bn_bwd_coefs_t (resnet_cls.py:157; the stem has no data gradient, so this kernel is dy's
// only reader and the apply pass -- read dz, read y, write dy, read dy -- does not exist).  Output: D[64 co][392 -> 416
// columns (tap, c)] = 2 x 13 MFMA tiles, wave w owns column tiles w and w + 8 for both row tiles; the accumulators stay in
// registers across all tiles of the (persistent) block, which leaves one [64][392] partial for io_splitk_reduce.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_cptr;

struct StemWgArgs {
    const bf16_t* x8;
    const bf16_t* dz;       // XB: gradient of relu(bn1(y)); else dy itself
    const bf16_t* y;        // XB
    const float *ta, *tb, *tc, *mean, *scale, *shift;       // XB: [G][64]
    float* partial;         // [gridDim.x][64][392]
    size_t x_bytes, dy_bytes;
    int ntiles, tiles_per_img, tiles_per_group;
};

template <bool XB>
__global__ __launch_bounds__(512, 2) void stem_wgrad_halo_kernel(IoConvGeom g, StemWgArgs a) {
    constexpr int NW = 8, PJ = 132, ROWS = 9, SLOTS = ROWS * 2 * PJ;
    constexpr int NCH = (SLOTS + 63) / 64, NAW = (NCH + NW - 1) / NW, ABUF = NAW * NW * 1024;       // 40 KB patch buffer
    constexpr int DBUF = 256 * 128;                                                                    // 32 KB dy buffer
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // LDS: [patch 0][patch 1][dy 0][dy 1]
    const lds_cptr lds = (lds_cptr)smem;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- patch fetches (as stem_halo_kernel)
    unsigned arel[NAW];
#pragma unroll
    for (int u = 0; u < NAW; ++u) {
        const int q = (wave * NAW + u) * 64 + lane;
        const int rp = q / PJ, j = q - rp * PJ, rr = rp >> 1, ix = 2 * j + (rp & 1) - 4;
        const bool ok = q < SLOTS && ix >= 0 && ix < g.Wi;
        arel[u] = ((unsigned)((rr * g.Wi + ix + 4) * 16)) | (ok ? (unsigned)rr : 15u);
    }
    auto issue_a = [&](int tile, int buf) {
        const int img = tile / a.tiles_per_img, oy0 = (tile - img * a.tiles_per_img) * 2;
        const long long start = (((long long)img * g.Hi + (2 * oy0 - 3)) * g.Wi - 4) * 16;
        const u32x4 rsA = dma_rsrc_raw((unsigned long long)((const char*)a.x8 + start), (size_t)((long long)a.x_bytes - start));
        const unsigned sb = lds0 + (unsigned)(buf * ABUF);
#pragma unroll
        for (int u = 0; u < NAW; ++u) {
            const unsigned rr = arel[u] & 15u;
            const bool ok = rr != 15u && (unsigned)(2 * oy0 - 3 + (int)rr) < (unsigned)g.Hi;
            dma16(rsA, sb + (unsigned)((wave * NAW + u) * 1024), ok ? (arel[u] & ~15u) : kOob, 0u);
        }
    };
    // ---- dy rows: thread = (pixel tid >> 3 + 64 i, channels 8 (tid & 7) ..), 16 bytes per tensor and i
    const int spx = tid >> 3, scg = tid & 7;
    u32x4 rdz[4], ry[4];
    float t_a[8], t_b[8], t_c[8], t_mu[8], t_sc[8], t_sh[8];
    auto dy_load = [&](int tile) {
        // (descriptors rebased per TILE -- wave-uniform; the thread's part rides in the vector offset)
        const size_t base = (size_t)tile * 256 * 128;
        const __amdgpu_buffer_rsrc_t rz = rsrc_at(a.dz, base, a.dy_bytes);
        const __amdgpu_buffer_rsrc_t rq = rsrc_at(XB ? a.y : a.dz, base, a.dy_bytes);
        const unsigned vo = (unsigned)(spx * 128 + scg * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rdz[i] = __builtin_amdgcn_raw_buffer_load_b128(rz, vo + (unsigned)(i * 64 * 128), 0, 0);
            if constexpr (XB) ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vo + (unsigned)(i * 64 * 128), 0, 0);
        }
        if constexpr (XB) {
            const int go = (tile / a.tiles_per_group) * 64 + scg * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t_a[e] = a.ta[go + e];
                t_b[e] = a.tb[go + e];
                t_c[e] = a.tc[go + e];
                t_mu[e] = a.mean[go + e];
                t_sc[e] = a.scale[go + e];
                t_sh[e] = a.shift[go + e];
            }
        }
    };
    // the LDS image of dy: pixel px at px * 128, its 32-byte block b (16 channels) in slot b ^ ((px >> 1) & 1): the four pixels
    // a transposing read touches (128 bytes apart) then sit in four different 64-byte bank groups
    auto dy_store = [&](int buf) {
        char* dst = smem + 2 * ABUF + buf * DBUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = spx + 64 * i;
            u32x4 v = rdz[i];
            if constexpr (XB) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    float o[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int e = 2 * d + hh;
                        const float yv = hh ? bf_hi(ry[i][d]) : bf_lo(ry[i][d]);
                        const float zv = hh ? bf_hi(rdz[i][d]) : bf_lo(rdz[i][d]);
                        const float act = __builtin_fmaf(yv - t_mu[e], t_sc[e], t_sh[e]);       // bn_apply's fma: the mask the forward saw
                        const float dzm = act > 0.f ? zv : 0.f;
                        o[hh] = __builtin_fmaf(t_a[e], dzm, __builtin_fmaf(t_b[e], yv, t_c[e]));
                    }
                    v[d] = io_f2bf2(o[0], o[1]);
                }
            }
            *reinterpret_cast<u32x4*>(dst + px * 128 + (((scg >> 1) ^ ((px >> 1) & 1)) << 5) + ((scg & 1) << 4)) = v;
        }
    };
    // ---- fragment addresses.  16-lane group g4 = lane >> 4: (g4 & 1) = which 16 of a tile's 32 rows / columns, (g4 >> 1) = k half;
    // within the group lane 4 fj + fq points at 4 channels (quad fq) of k row fj
    const int g4 = lane >> 4, fj = (lane >> 2) & 3, fq = lane & 3;
    const unsigned krow = (unsigned)((g4 >> 1) * 8 + fj);                   // pixel of the k-step this lane points at (+ 4 for the 2nd read)
    unsigned fa[2];                                                         // A = dy^T: row tile rt = 32 output channels
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int co = rt * 32 + (g4 & 1) * 16 + 4 * fq;
        // (the slot swizzle depends on bit 1 of the pixel = bit 1 of fj: k-steps start at multiples of 8)
        fa[rt] = (unsigned)(2 * ABUF) + krow * 128u + (unsigned)((((co >> 4) ^ ((fj >> 1) & 1)) << 5) + (co & 15) * 2);
    }
    const int nct = wave + 8 < 13 ? 2 : 1;                                  // column tiles of this wave: wave, wave + 8
    unsigned fb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = (wave + 8 * j) * 32 + (g4 & 1) * 16 + 4 * fq;      // first of the lane's four columns
        int tap = col >> 3;
        if (tap > 48) tap = 48;                                             // padding columns 392 .. 415: any valid pixels
        const int ty = tap / 7, tx = tap - ty * 7;
        fb[j] = (unsigned)((((2 * ty + ((tx + 1) & 1)) * PJ + ((tx + 1) >> 1)) * 16) + (col & 7) * 2) + krow * 16u;
    }
    auto tr8 = [&](unsigned off, unsigned second) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off + second));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int tile = xcd_remap(blockIdx.x, gridDim.x), buf = 0;
    if (tile < a.ntiles) {
        issue_a(tile, 0);
        dy_load(tile);
        dy_store(0);
    }
    while (tile < a.ntiles) {
        dma_wait_left<0>();
        __syncthreads();                 // patch and dy rows of this tile are in LDS; every wave is done with the other buffers
        const int ntile = tile + (int)gridDim.x;
        if (ntile < a.ntiles) {
            issue_a(ntile, buf ^ 1);
            dy_load(ntile);
        }
        const unsigned pa = (unsigned)(buf * ABUF), pd = (unsigned)(buf * DBUF);
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            // k-step s: pixels 16 s .. 16 s + 15 of the tile = row s >> 3, columns 16 (s & 7) ..
            const unsigned ka = pd + (unsigned)(s_ * 16 * 128);
            const unsigned kb = pa + (unsigned)(((4 * (s_ >> 3)) * PJ + (s_ & 7) * 16) * 16);
            bf16x8 A[2], B[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) A[rt] = tr8(fa[rt] + ka, 4 * 128);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (j < nct) B[j] = tr8(fb[j] + kb, 4 * 16);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (j < nct) acc[rt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[rt], B[j], acc[rt][j], 0, 0, 0);
        }
        if (ntile < a.ntiles) dy_store(buf ^ 1);
        buf ^= 1;
        tile = ntile;
    }
    // the block's partial: D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* dst = a.partial + (size_t)blockIdx.x * 64 * 392;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = (wave + 8 * j) * 32 + (lane & 31);
        if (j < nct && n < 392) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    dst[(size_t)co * 392 + n] = acc[rt][j][r];
                }
        }
    }
}


// ---- filter gradient of the 3x3 stride-1 layers (conv2 of the layer-1 .. layer-3 Bottlenecks), bf16 --------------------------
// The recipe of stem_wgrad_halo_kernel on the geometry of conv_halo3_kernel: dW[co][tap][ci] = sum over pixels of dy[px][co] *
// x[px + tap][ci].  conv_wgrad_bf16_tr_kernel gives every tap its own blocks -- x and dy go through L2 nine times and the
// launch runs at 0.5 .. 0.9 PF/s (layer 1: 0.29 ms against 0.09 of HBM time at 256 pairs).  Here a tile is 128 output pixels =
// 128 / W whole image rows: the halo image of a 64-channel slice of x (the swizzled image of conv_halo3_kernel) and the tile's
// rows of a 64-channel slice of dy go to LDS once, by LDS-DMA, and all nine taps are multiplied from them: D[64 co][576 columns
// (tap, ci)] = 2 x 18 MFMA tiles, wave w owns column tiles w, w + 8, w + 16 for both row tiles (<= 96 accumulator registers,
// resident across all tiles a block walks: one partial per block).  A transposing read's 16 columns are 16 input channels of ONE
// tap: 32 contiguous bytes of a halo pixel.  With more than 64 channels the (Co / 64) x (Ci / 64) slice pairs are separate
// SUB-PROBLEMS: block b works on pair b % nsp and walks the tiles (b / nsp) + k (grid / nsp) -- x is read Co / 64 times and dy
// Ci / 64 times instead of nine times each.
struct Wg3Args {
    const bf16_t* x;
    const bf16_t* dy;
    float* partial;         // [gridDim.x][64][576]
    size_t x_bytes, dy_bytes;
    int ntiles, tiles_per_img, nci, nsp;
};

template <int W>
__global__ __launch_bounds__(512, 2) void conv_wgrad_halo3_kernel(IoConvGeom g, Wg3Args a) {
    constexpr int NW = 8, R = 128 / W, WP = W + 2, HP = (R + 2) * WP;      // halo image: rows h0 - 1 .. h0 + R
    constexpr int NAW = 5, ABUF = NAW * NW * 1024;                           // 40 KB (<= 320 halo pixels)
    static_assert((HP + 7) / 8 <= NAW * NW, "halo image");
    constexpr int DBUF = 128 * 128;                                          // 16 KB of dy rows
    constexpr int STG = ABUF + DBUF;
    constexpr int LW = W == 64 ? 6 : (W == 32 ? 5 : 4);                      // log2 W
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // LDS: [x halo 0][dy 0][x halo 1][dy 1]
    const lds_cptr lds = (lds_cptr)smem;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r8 = lane >> 3, p8 = lane & 7;
    const int sp = (int)blockIdx.x % a.nsp, gsz = (int)gridDim.x / a.nsp;
    // the blocks of one tile (its nsp slice pairs) sit on nsp consecutive XCDs (block b runs on XCD b % 8): 8 / nsp groups of XCDs
    // take turns through the tile ranks -- remapped so that each group walks a CONTIGUOUS run of tiles (neighbouring tiles share
    // two of their halo rows: out of the same L2)
    int rank = (int)blockIdx.x / a.nsp;
    {
        const int ng = a.nsp <= 8 ? 8 / a.nsp : 1;
        const int xg = rank % ng, ig = rank / ng, qg = gsz / ng, rg = gsz % ng;
        rank = (xg < rg ? xg * (qg + 1) : rg * (qg + 1) + (xg - rg) * qg) + ig;
    }
    const int co0 = (sp / a.nci) * 64, ci0 = (sp % a.nci) * 64;
    const unsigned xpix = (unsigned)(g.Ci * 2), dpix = (unsigned)(g.Co * 2);      // bytes per pixel of x / dy
    // ---- x fetches (as conv_halo3_kernel: halo pixel hp = chunk * 8 + r8 -> (hy, hx), slot p8 holds chunk p8 ^ ((hp >> 1) & 7))
    const unsigned slot_e = (unsigned)((p8 ^ (r8 >> 1)) << 4), slot_o = slot_e ^ 64u;
    const unsigned padpx = (unsigned)(W + 1);
    unsigned arel[NAW];
#pragma unroll
    for (int u = 0; u < NAW; ++u) {
        const int ch = wave * NAW + u, hp = ch * 8 + r8;
        const int hy = hp / WP, hx = hp - hy * WP;
        const bool colok = hp < HP && hx >= 1 && hx <= W;
        arel[u] = ((unsigned)((hy - 1) * W + (hx - 1) + (int)padpx) * xpix + (unsigned)(ci0 * 2) + ((ch & 1) ? slot_o : slot_e)) |
                  (colok ? (unsigned)hy : 15u);
    }
    // ---- dy fetches: 16 chunks of 8 pixels, two per wave; pixel px at px * 128, its 32-byte block b in slot b ^ ((px >> 1) & 1)
    // (stem_wgrad_halo_kernel's image): LDS position p8 of a row holds the 16-byte chunk (((p8 >> 1) ^ ((r8 >> 1) & 1)) << 1) | (p8 & 1)
    const unsigned dvo = (unsigned)r8 * dpix + (unsigned)(co0 * 2) + (unsigned)(((((p8 >> 1) ^ ((r8 >> 1) & 1)) << 1) | (p8 & 1)) << 4);
    auto issue = [&](int tile, int buf) {
        const int img = tile / a.tiles_per_img, h0 = (tile - img * a.tiles_per_img) * R;
        const long long start = ((long long)img * g.Hi * g.Wi + (long long)h0 * W - (long long)padpx) * (long long)xpix;
        const u32x4 rsA = dma_rsrc_raw((unsigned long long)((const char*)a.x + start), (size_t)((long long)a.x_bytes - start));
        const unsigned sb = lds0 + (unsigned)(buf * STG);
#pragma unroll
        for (int u = 0; u < NAW; ++u) {
            const unsigned hy = arel[u] & 15u;
            const bool ok = hy != 15u && (unsigned)(h0 - 1 + (int)hy) < (unsigned)g.Hi;
            dma16(rsA, sb + (unsigned)((wave * NAW + u) * 1024), ok ? (arel[u] & ~15u) : kOob, 0u);
        }
        const size_t dbase = (size_t)tile * 128 * dpix;
        const u32x4 rsD = dma_rsrc_raw((unsigned long long)((const char*)a.dy + dbase), a.dy_bytes - dbase);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = wave * 2 + u;                        // pixels 8 ch .. 8 ch + 7 (8 ch is a multiple of 8: the swizzle bit is r8's)
            dma16(rsD, sb + (unsigned)(ABUF + ch * 1024), dvo, (unsigned)(ch * 8) * dpix);
        }
    };
    // ---- fragment addresses (16-lane group g4: (g4 & 1) = which 16 of a tile's 32 rows / columns, (g4 >> 1) = k half; lane 4 fj + fq
    // of the group points at 4 channels (quad fq) of k row fj)
    const int g4 = lane >> 4, fj = (lane >> 2) & 3, fq = lane & 3;
    const int krow = (g4 >> 1) * 8 + fj;
    unsigned fa[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int co = rt * 32 + (g4 & 1) * 16 + 4 * fq;
        fa[rt] = (unsigned)ABUF + (unsigned)krow * 128u + (unsigned)((((co >> 4) ^ ((fj >> 1) & 1)) << 5) + (co & 15) * 2);
    }
    const int nct = wave + 16 < 18 ? 3 : 2;                     // column tiles of this wave: wave, wave + 8 (, wave + 16)
    int bl[3];                                                  // halo pixel of pixel 0 of the tile under the lane's tap, + the lane's k row
    unsigned bc[3];                                             // the lane's 16-byte chunk (bits 4..) and byte inside it
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int col = (wave + 8 * j) * 32 + (g4 & 1) * 16 + 4 * fq;          // (j = 2 of the waves without a third tile: unused)
        const int tap = (col >> 6) < 9 ? (col >> 6) : 8, ci = col & 63;
        const int th = tap / 3, tw = tap - th * 3;
        // pixel 16 s + krow of the tile = image row (16 s + krow) >> LW, column (16 s + krow) & (W - 1): W >= 16 keeps krow (< 16)
        // inside the row of pixel 16 s
        bl[j] = (1 + g.dh0 + g.dhs * th) * WP + 1 + (g.dw0 + g.dws * tw) + (krow & (W - 1));
        bc[j] = (unsigned)(((ci >> 3) << 4) | ((ci & 7) * 2));
    }
    auto tr_a = [&](unsigned off) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off + 4 * 128));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto tr_b = [&](unsigned base, int hp, unsigned cb) -> bf16x8 {       // k rows hp .. hp + 3 and hp + 4 .. hp + 7 (this lane: one of each)
        const unsigned o0 = base + (unsigned)hp * 128u + ((((cb >> 4) ^ (((unsigned)hp >> 1) & 7u)) << 4) | (cb & 15u));
        const unsigned h1 = (unsigned)hp + 4u;
        const unsigned o1 = base + h1 * 128u + ((((cb >> 4) ^ ((h1 >> 1) & 7u)) << 4) | (cb & 15u));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + o0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + o1));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int tile = rank, buf = 0;
    if (tile < a.ntiles) issue(tile, 0);
    while (tile < a.ntiles) {
        dma_wait_left<0>();
        __syncthreads();
        const int ntile = tile + gsz;
        if (ntile < a.ntiles) issue(ntile, buf ^ 1);
        const unsigned sb = (unsigned)(buf * STG);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            // k-step s: pixels 16 s .. 16 s + 15 of the tile = image row (16 s) >> LW, columns (16 s) & (W - 1) ..
            const int hps = ((16 * s_) >> LW) * WP + ((16 * s_) & (W - 1));
            bf16x8 A[2], B[3];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) A[rt] = tr_a(sb + fa[rt] + (unsigned)(s_ * 16 * 128));
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < nct) B[j] = tr_b(sb, bl[j] + hps, bc[j]);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < nct) acc[rt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[rt], B[j], acc[rt][j], 0, 0, 0);
        }
        buf ^= 1;
        tile = ntile;
    }
    float* dst = a.partial + (size_t)blockIdx.x * 64 * 576;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nct) {
            const int n = (wave + 8 * j) * 32 + (lane & 31);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    dst[(size_t)co * 576 + n] = acc[rt][j][r];
                }
        }
    }
}

// dw[co][tap][ci] = sum over the blocks of slice pair (co / 64, ci / 64) of their partial [64][9 x 64], fixed order (the shape of
// splitk_reduce_kernel: 32 outputs x 8 partial groups per block, eight loads in flight per thread)
__global__ __launch_bounds__(256) void wgrad_halo3_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Ci,
                                                                 int Co, int nci, int nsp, int gsz) {
    __shared__ f32x4 red[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t i4 = (size_t)blockIdx.x * 32 + tx;                    // float4 index into dw
    const size_t n4 = (size_t)Co * 9 * Ci / 4;
    const bool ok = i4 < n4;
    const size_t e = ok ? i4 * 4 : 0;
    const int ci = (int)(e % (size_t)Ci), rest = (int)(e / (size_t)Ci), tap = rest % 9, co = rest / 9;
    const int sp = (co >> 6) * nci + (ci >> 6);
    const f32x4* p4 = reinterpret_cast<const f32x4*>(partial) + (size_t)sp * (64 * 576 / 4) +
                      ((size_t)(co & 63) * 576 + tap * 64 + (ci & 63)) / 4;
    const size_t stride = (size_t)nsp * (64 * 576 / 4);               // between the partials of consecutive blocks of the pair
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (ok)
        for (int z0 = ty; z0 < gsz; z0 += 64) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = z0 + 8 * u;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                v[u] = z < gsz ? p4[(size_t)z * stride] : zero;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && ok) {
#pragma unroll
        for (int k = 1; k < 8; ++k) s += red[k][tx];
        reinterpret_cast<f32x4*>(dw)[i4] = s;
    }
}

}  // namespace

// IO_OK = launched, 1 = not this kernel's shape / form (the caller falls through), < 0 = error
int io_launch_conv_halo3(const IoConvGeom& g, const void* in, const void* wgt, void* out, const void* add, const void* mask,
                         hipStream_t st, float* st_mean, float* st_m2, const IoBwStats* bw, size_t in_bytes,
                         unsigned w_bytes, size_t out_bytes) {
    const int mode = io_bf16_persist_mode();
    if (mode != 1 && mode != 3) return 1;
    const bool same3 = g.Th == 3 && g.Tw == 3 && g.S == 3 && g.wT == 9 && g.is == 1 && g.os == 1 && g.Hi == g.Ho &&
                       g.Wi == g.Wo && g.outH == g.Ho && g.outW == g.Wo && !g.gw && !g.cr && g.dhs * g.dhs == 1 &&
                       g.dws * g.dws == 1 && g.dh0 == -g.dhs && g.dw0 == -g.dws;
    if (!same3 || mask) return 1;
    // (128 output channels run, but no faster than conv_p256: 0.190 against 0.184 ms forward, 0.228 against 0.234 ms data
    // gradient on layer 2 of the bench step -- not routed here, not instantiated)
    if (!(g.Co == 64 && (g.Ci == 64 || g.Ci == 128))) return 1;
    if (!(g.Wo == 64 || g.Wo == 32) || (g.Ho * g.Wo) % 256 != 0) return 1;
    if (bw && (bw->in_scale || bw->xb_a || bw->a_out || bw->wino_u || bw->maskbits)) return 1;
    if (st_mean && (add || (bw && (bw->y || bw->bias)))) return 1;
    if (bw && bw->y && (!bw->mscale || add || bw->bias || bw->Mg % 256 != 0)) return 1;
    const long M = (long)g.N * g.Ho * g.Wo;
    if ((double)g.Hi * g.Wi * g.Ci * 2.0 >= 2.0e9 || 256.0 * g.Co * 2.0 >= 4.0e9) return 1;
    const int epi = st_mean ? EPI_STATS : ((bw && bw->y) ? EPI_BWE : EPI_PLAIN);
    Halo3Args a;
    memset(&a, 0, sizeof(a));
    a.in = (const bf16_t*)in;
    a.wgt = (const bf16_t*)wgt;
    a.out = (bf16_t*)out;
    a.add = (const bf16_t*)add;
    a.in_bytes = in_bytes;
    a.out_bytes = out_bytes;
    a.w_bytes = w_bytes;
    a.st_mean = st_mean;
    a.st_m2 = st_m2;
    if (bw) a.bw = *bw;
    a.ntiles = (int)(M / 256);
    a.tiles_per_img = g.Ho * g.Wo / 256;
    const int ncu = io_device_cu_count();
    {
        const long rounds = ((long)a.ntiles + ncu - 1) / ncu;
        if (mode != 3 && (long)a.ntiles * 10 < rounds * ncu * 8) return 1;     // (conv_p256.hip's rule)
    }
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    const double kred = 9.0 * g.Ci;
    IoProfScope prof(IO_PROF_CONV_NT64, 2.0 * (double)M * g.Co * kred,
                     2.0 * M * g.Co * (1.0 + (add ? 1.0 : 0.0) + ((bw && bw->y) ? 1.0 : 0.0)) +
                         2.0 * ((double)g.N * g.Hi * g.Wi * g.Ci + (double)g.Co * kred),
                     st);
#define IO_HALO3_LAUNCH(W_, BN_, WM_, WN_, TPS_, EPI_)                                                                    \
    do {                                                                                                                  \
        constexpr int HP_ = (256 / W_ + 2) * (W_ + 2), NAW_ = ((HP_ + 7) / 8 + 7) / 8;                                    \
        const size_t lds = (size_t)2 * NAW_ * 8 * 1024 + (size_t)2 * TPS_ * BN_ * 128 + (TPS_ == 0 ? 3 * BN_ * 4 : 0);                                    \
        static std::atomic<unsigned long long> attr_done{0};                                                              \
        if (io_first_on_device(attr_done))                                                                                \
            (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<W_, BN_, WM_, WN_, TPS_, EPI_>,                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        hipLaunchKernelGGL((conv_halo3_kernel<W_, BN_, WM_, WN_, TPS_, EPI_>), dim3((unsigned)grid), dim3(512), lds, st,  \
                           g, a);                                                                                         \
    } while (0)
#define IO_HALO3_EPI(W_, BN_, WM_, WN_, TPS_)                                     \
    do {                                                                          \
        if (epi == EPI_STATS) IO_HALO3_LAUNCH(W_, BN_, WM_, WN_, TPS_, EPI_STATS);  \
        else if (epi == EPI_BWE) IO_HALO3_LAUNCH(W_, BN_, WM_, WN_, TPS_, EPI_BWE); \
        else IO_HALO3_LAUNCH(W_, BN_, WM_, WN_, TPS_, EPI_PLAIN);                   \
    } while (0)
    // 64 output channels: 4 x 2 waves of 64 x 32, three taps per stage; 128: 4 x 2 waves of 64 x 64, one tap per stage
    static std::atomic<int> breg_c{-1};
    int breg = breg_c.load(std::memory_order_relaxed);
    if (breg < 0) {
        const char* e = getenv("IO_HALO3_BREG");
        breg = (e && e[0] == '0') ? 0 : 1;
        breg_c.store(breg, std::memory_order_relaxed);
    }
    // 4 x 2 waves of 64 x 32; 64 -> 64 channels: the filters in registers, otherwise three taps per LDS stage
    if (breg && g.Wo == 64 && g.Ci == 64) IO_HALO3_EPI(64, 64, 4, 2, 0);
    else if (breg && g.Wo == 32 && g.Ci == 64) IO_HALO3_EPI(32, 64, 4, 2, 0);
    else if (g.Wo == 64) IO_HALO3_EPI(64, 64, 4, 2, 3);
    else IO_HALO3_EPI(32, 64, 4, 2, 3);
#undef IO_HALO3_EPI
#undef IO_HALO3_LAUNCH
    return io_check_launch("conv_halo3");
}

// The stem in bf16 (packed 8-channel input, 7x7 stride 2, 64 channels, 128-wide output rows): same contract.
int io_launch_conv_stem_halo(const IoConvGeom& g, const void* in, const void* wgt, void* out, hipStream_t st, float* st_mean,
                             float* st_m2, const IoBwStats* bw, size_t in_bytes, unsigned w_bytes, size_t out_bytes) {
    const int mode = io_bf16_persist_mode();
    if (mode != 1 && mode != 3) return 1;
    if (!(g.Th == 7 && g.Tw == 7 && g.S == 7 && g.wT == 49 && g.is == 2 && g.os == 1 && g.Ci == 8 && g.Co == 64 && !g.gw && !g.cr &&
          g.Wo == 128 && g.Wi == 256 && g.Hi == 2 * g.Ho && (g.Ho & 1) == 0 && g.outH == g.Ho && g.outW == g.Wo && g.dh0 == -3 &&
          g.dw0 == -3 && g.dhs == 1 && g.dws == 1))
        return 1;
    if (bw && (bw->in_scale || bw->xb_a || bw->a_out || bw->wino_u || bw->maskbits || bw->y)) return 1;
    if (st_mean && bw && bw->bias) return 1;
    if ((double)g.Hi * g.Wi * 16.0 >= 2.0e9) return 1;
    const long M = (long)g.N * g.Ho * g.Wo;
    Halo3Args a;
    memset(&a, 0, sizeof(a));
    a.in = (const bf16_t*)in;
    a.wgt = (const bf16_t*)wgt;
    a.out = (bf16_t*)out;
    a.in_bytes = in_bytes;
    a.out_bytes = out_bytes;
    a.w_bytes = w_bytes;
    a.st_mean = st_mean;
    a.st_m2 = st_m2;
    if (bw) a.bw = *bw;
    a.ntiles = (int)(M / 256);
    a.tiles_per_img = g.Ho / 2;
    const int ncu = io_device_cu_count();
    {
        const long rounds = ((long)a.ntiles + ncu - 1) / ncu;
        if (mode != 3 && (long)a.ntiles * 10 < rounds * ncu * 8) return 1;
    }
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    IoProfScope prof(IO_PROF_CONV_STEM, 2.0 * (double)M * g.Co * 49.0 * 5.0,
                     2.0 * M * g.Co + 2.0 * ((double)g.N * g.Hi * g.Wi * 8 + 64.0 * 49 * 8), st);
    constexpr size_t lds = (size_t)2 * 5 * 8 * 1024;
    static_assert((9 * 2 * 132 + 63) / 64 <= 5 * 8, "patch chunks");
    if (st_mean) {
        static std::atomic<unsigned long long> done{0};
        if (io_first_on_device(done))
            (void)hipFuncSetAttribute((const void*)stem_halo_kernel<EPI_STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((stem_halo_kernel<EPI_STATS>), dim3((unsigned)grid), dim3(512), lds, st, g, a);
    } else {
        static std::atomic<unsigned long long> done{0};
        if (io_first_on_device(done))
            (void)hipFuncSetAttribute((const void*)stem_halo_kernel<EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((stem_halo_kernel<EPI_PLAIN>), dim3((unsigned)grid), dim3(512), lds, st, g, a);
    }
    return io_check_launch("stem_halo");
}

// The stem's filter gradient in bf16 (see stem_wgrad_halo_kernel): dw[64][49][8] fp32 = reduction of one partial per block.
// xb: bn1's backward folded in (dz / y / tables), or null (dz is dy).  IO_OK, 1 = not this kernel's shape / workspace, < 0 error.
size_t io_stem_wgrad_halo_partial_bytes() { return (size_t)io_stem_wgrad_rows_max_blocks() * 64 * 392 * sizeof(float); }
static int stem_wg_ncu() {
    const int ncu = io_device_cu_count();
    return ncu;
}
// would io_launch_stem_wgrad_halo take this launch (shape, mode, enough tiles for its persistent blocks, workspace)?
bool io_stem_wgrad_halo_ok(const IoConvGeom& g, size_t partial_bytes, int G) {
    const int mode = io_bf16_persist_mode();
    if (mode != 1 && mode != 3) return false;
    if (!(g.Th == 7 && g.Tw == 7 && g.S == 7 && g.wT == 49 && g.is == 2 && g.Ci == 8 && g.Co == 64 && !g.gw && !g.cr && g.Wo == 128 &&
          g.Wi == 256 && g.Hi == 2 * g.Ho && (g.Ho & 1) == 0 && g.dh0 == -3 && g.dw0 == -3 && g.dhs == 1 && g.dws == 1))
        return false;
    if ((double)g.Hi * g.Wi * 16.0 >= 2.0e9 || G < 1 || g.N % G != 0) return false;
    const long ntiles = (long)g.N * g.Ho * g.Wo / 256;
    long grid = ntiles < stem_wg_ncu() ? ntiles : stem_wg_ncu();
    if (grid > io_stem_wgrad_rows_max_blocks()) grid = io_stem_wgrad_rows_max_blocks();
    if ((size_t)grid * 64 * 392 * sizeof(float) > partial_bytes) return false;
    const long rounds = (ntiles + grid - 1) / grid;
    return mode == 3 || ntiles * 10 >= rounds * grid * 8;
}
int io_launch_stem_wgrad_halo(const IoConvGeom& g, const void* x8, const void* dz, float* dw, float* partial, size_t partial_bytes,
                              hipStream_t st, const IoStemXb* xb) {
    if (!io_stem_wgrad_halo_ok(g, partial_bytes, xb ? xb->G : 1)) return 1;
    const long M = (long)g.N * g.Ho * g.Wo;
    StemWgArgs a;
    memset(&a, 0, sizeof(a));
    a.x8 = (const bf16_t*)x8;
    a.dz = (const bf16_t*)dz;
    a.partial = partial;
    a.x_bytes = (size_t)g.N * g.Hi * g.Wi * 16;
    a.dy_bytes = (size_t)M * 128;
    a.ntiles = (int)(M / 256);
    a.tiles_per_img = g.Ho / 2;
    if (xb) {
        a.y = (const bf16_t*)xb->y;
        a.ta = xb->a;
        a.tb = xb->b;
        a.tc = xb->c;
        a.mean = xb->mean;
        a.scale = xb->scale;
        a.shift = xb->shift;
        a.tiles_per_group = (g.N / xb->G) * a.tiles_per_img;
    }
    int grid = a.ntiles < stem_wg_ncu() ? a.ntiles : stem_wg_ncu();
    if (grid > io_stem_wgrad_rows_max_blocks()) grid = io_stem_wgrad_rows_max_blocks();
    {
        IoProfScope prof(IO_PROF_WGRAD_STEM, 2.0 * (double)M * 64 * 49.0 * 5.0,
                         (xb ? 4.0 : 2.0) * M * 64 + 2.0 * (double)g.N * g.Hi * g.Wi * 8, st);
        constexpr size_t lds = (size_t)2 * 5 * 8 * 1024 + (size_t)2 * 256 * 128;
        if (xb) {
            static std::atomic<unsigned long long> done{0};
            if (io_first_on_device(done))
                (void)hipFuncSetAttribute((const void*)stem_wgrad_halo_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((stem_wgrad_halo_kernel<true>), dim3((unsigned)grid), dim3(512), lds, st, g, a);
        } else {
            static std::atomic<unsigned long long> done{0};
            if (io_first_on_device(done))
                (void)hipFuncSetAttribute((const void*)stem_wgrad_halo_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((stem_wgrad_halo_kernel<false>), dim3((unsigned)grid), dim3(512), lds, st, g, a);
        }
        const int rc = io_check_launch("stem_wgrad_halo");
        if (rc) return rc;
    }
    return io_splitk_reduce(partial, dw, (size_t)64 * 392 / 4, grid, st);
}

// Filter gradient of a bf16 3x3 stride-1 convolution with 64 / 128 / 256 channels on 64- / 32- / 16-wide maps
// (conv_wgrad_halo3_kernel).  IO_OK / 1 = not this kernel's / < 0.
size_t io_wgrad_halo3_partial_bytes() { return (size_t)io_stem_wgrad_rows_max_blocks() * 64 * 576 * sizeof(float); }
bool io_wgrad_halo3_shape(const IoConvGeom& g) {
    return g.Th == 3 && g.Tw == 3 && g.S == 3 && g.wT == 9 && g.is == 1 && g.os == 1 && g.Hi == g.Ho && g.Wi == g.Wo && !g.gw && !g.cr &&
           g.Ci == g.Co && (g.Ci == 64 || g.Ci == 128 || g.Ci == 256) && (g.Wo == 64 || g.Wo == 32 || g.Wo == 16) &&
           (g.Ho * g.Wo) % 128 == 0 && g.dhs == 1 && g.dws == 1 && g.dh0 == -1 && g.dw0 == -1 && g.r0 == 0 && g.rs == 1 && g.s0 == 0 &&
           g.ss == 1;
}
int io_launch_conv_wgrad_halo3(const IoConvGeom& g, const void* x, const void* dy, float* dw, float* partial, size_t partial_bytes,
                               hipStream_t st) {
    const int mode = io_bf16_persist_mode();
    if (mode != 1 && mode != 3) return 1;
    if (!io_wgrad_halo3_shape(g) || (double)g.Hi * g.Wi * g.Ci * 2.0 >= 2.0e9 || 128.0 * g.Co * 2.0 >= 2.0e9) return 1;
    const long M = (long)g.N * g.Ho * g.Wo;
    Wg3Args a;
    memset(&a, 0, sizeof(a));
    a.x = (const bf16_t*)x;
    a.dy = (const bf16_t*)dy;
    a.partial = partial;
    a.x_bytes = (size_t)g.N * g.Hi * g.Wi * g.Ci * 2;
    a.dy_bytes = (size_t)M * g.Co * 2;
    a.ntiles = (int)(M / 128);
    a.tiles_per_img = g.Ho * g.Wo / 128;
    a.nci = g.Ci / 64;
    a.nsp = a.nci * (g.Co / 64);
    int cap = stem_wg_ncu() < io_stem_wgrad_rows_max_blocks() ? stem_wg_ncu() : io_stem_wgrad_rows_max_blocks();
    int gsz = cap / a.nsp;                                  // blocks per slice pair
    if (gsz > a.ntiles) gsz = a.ntiles;
    if (gsz < 1) return 1;
    const int grid = gsz * a.nsp;
    if ((size_t)grid * 64 * 576 * sizeof(float) > partial_bytes) return 1;
    {
        const long rounds = ((long)a.ntiles + gsz - 1) / gsz;
        if (mode != 3 && ((long)a.ntiles * 10 < rounds * gsz * 8 || grid * 10 < cap * 8)) return 1;
    }
    {
        IoProfScope prof(IO_PROF_WGRAD, 2.0 * (double)M * g.Co * 9.0 * g.Ci, 2.0 * ((double)M * (g.Ci + g.Co) + 9.0 * g.Co * g.Ci), st);
        constexpr size_t lds = (size_t)2 * (5 * 8 * 1024 + 128 * 128);
#define IO_WG3_LAUNCH(W_)                                                                                                  \
    do {                                                                                                                  \
        static std::atomic<unsigned long long> done{0};                                                                   \
        if (io_first_on_device(done))                                                                                     \
            (void)hipFuncSetAttribute((const void*)conv_wgrad_halo3_kernel<W_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                          \
        hipLaunchKernelGGL((conv_wgrad_halo3_kernel<W_>), dim3((unsigned)grid), dim3(512), lds, st, g, a);                \
    } while (0)
        if (g.Wo == 64) IO_WG3_LAUNCH(64);
        else if (g.Wo == 32) IO_WG3_LAUNCH(32);
        else IO_WG3_LAUNCH(16);
#undef IO_WG3_LAUNCH
        int rc = io_check_launch("conv_wgrad_halo3");
        if (rc) return rc;
        const size_t n4 = (size_t)g.Co * 9 * g.Ci / 4;
        hipLaunchKernelGGL(wgrad_halo3_reduce_kernel, dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, st, partial, dw, g.Ci, g.Co,
                           a.nci, a.nsp, gsz);
        rc = io_check_launch("wgrad_halo3_reduce");
        return rc;
    }
}
