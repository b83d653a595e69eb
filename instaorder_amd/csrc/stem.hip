// The stem convolution (7x7 stride 2, 5 real input channels -> 64; resnet_cls.py:155-156 `conv1`) as a ROW-PERSISTENT kernel.
//
// The generic implicit-GEMM kernel (conv_igemm.hip, exact-K mode) treats the stem as a GEMM with K = 245 -> 256: per 128-row
// output tile it gathers eight k-tiles of the A operand pixel by pixel (4 B per address computation: five channels of a
// pixel are not a power of two), re-fetches the 64 KB filter, and pays two barriers per k-tile for 32 k of work: 3.1 ms at
// 512 x 256 x 256 (85 TF/s), of which its own ablation says the MFMAs are 1.9.
//
// Here the reduction index is laid out the way the INPUT sits in memory.  One tile is one output row (Wo = 128 pixels of
// one sample; 128 | Wo): the 7 input rows x 261 input pixels x 5 channels that row reads are brought into LDS ONCE as a
// compact [7][262][5] image (36.7 KB).  For filter row r, output pixel w reads the 35 contiguous floats that start at
// pixel 2w of patch row r -- so the A fragments are plain LDS reads at a 40-byte lane stride (conflict-free for 8-byte
// reads), no gather and no im2col copy.  The filter is re-packed once per BLOCK into [7][64][36] (k = 35 -> 36, one zero
// column; 64.5 KB, 16-byte fragment reads at a 144-byte lane stride: conflict-free) and stays resident while the block
// walks its share of the rows; the patch of the next row is fetched into registers under the MFMAs of this one and lands
// in the other half of a double buffer.  ONE barrier per output row: the BatchNorm statistics are reduced per wave and
// merged a row later (see park / merge below).
// Per row: 7 x 18 = 126 MFMAs per wave (K = 252 against the 245 real), eight waves of 32 pixels x 32 channels, one block
// of 512 threads per CU (148 KB of LDS).
//
// Blocks take CONTIGUOUS runs of rows, so the five input rows two consecutive output rows share come out of the L2 of the
// XCD that just fetched them.
//
// (filter gradient: below, with its own header)
//
// Measured (512 x 256 x 256, tools/one_stem.py, profiles/r03_stem_rows.txt): 2.37 ms = 111 TF/s against 3.1 ms; with
// stores, fetch + staging and statistics compiled out (IO_STEM_ABLATE=7) the MFMA + fragment-read loop alone is 1.95 ms =
// 135 TF/s, i.e. at what v_mfma_f32_32x32x2_f32 sustains on this part -- the remaining 0.4 ms is the epilogue work that
// does not hide: a wave's MFMAs form ONE dependent chain (a 32 x 32 tile per wave), each holds the wave's in-order issue
// for the 64 cycles of its predecessor, and whatever the scheduler puts between two of them in runs longer than that
// leaves the pipe idle on both waves of the SIMD, which the barrier keeps in phase.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "io_common.h"

#ifndef IO_STEM_ABLATE
#define IO_STEM_ABLATE 0      // measurement builds only: 1 no output stores, 2 no patch fetch / staging, 4 no statistics
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kCR = 5;                      // real input channels
constexpr int kPW = 262;                    // patch pixels per row: 2 * 128 + 5, + the column the zero k reads
constexpr int kPitch = kPW * kCR;           // 1310 floats (even: 8-byte fragment reads stay aligned)
constexpr int kKB = 36;                     // k per filter row in LDS (35 real + 1 zero)
constexpr int kBImg = 7 * 64 * kKB;         // floats of the resident filter image
constexpr int kPix = 7 * kPW;               // patch pixels
constexpr int kNT = 512;
constexpr int kRounds = (kPix + kNT - 1) / kNT;   // 4
constexpr int kPatch = kRounds * kNT * kCR; // floats per patch buffer: the staging loop writes ALL its slots (no branch);
                                            // the ones past the 7 x 262 pixels land in this tail
constexpr int kRed = 2 * 4 * 64 * 2;        // statistics scratch: [row parity][pixel group][channel](mean, M2)
constexpr size_t kLds = (size_t)(kBImg + 2 * kPatch + kRed) * sizeof(float);

template <bool STATS>
__global__ __launch_bounds__(kNT, 1) void stem_rows_kernel(const float* __restrict__ x8, const float* __restrict__ wp,
                                                           int kp, float* __restrict__ out,
                                                           float* __restrict__ st_mean, float* __restrict__ st_m2,
                                                           const float* __restrict__ bias, int relu, int Hi, int Wi,
                                                           int Ho, int Wo, int ntiles, int per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sB = smem;                       // [7][64][36]
    float* sP = smem + kBImg;               // [2][kPatch]: [7][262][5] + tail
    float* sR = sP + 2 * kPatch;            // [2][4][64][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rg = wave & 3, cg = wave >> 2;            // 32-pixel group, 32-channel group of this wave
    const int l32 = lane & 31, h = lane >> 5;
    const int t_lo = blockIdx.x * per_block, t_hi = min(ntiles, t_lo + per_block);
    if (t_lo >= t_hi) return;
    const int cbn = Wo >> 7;                // 128-pixel column blocks per output row

    for (int idx = tid; idx < kBImg; idx += kNT) {
        const int k = idx % kKB, rc = idx / kKB, ch = rc & 63, r = rc >> 6;
        sB[idx] = k < 7 * kCR ? wp[(size_t)ch * kp + r * 7 * kCR + k] : 0.f;
    }

    // what a thread fetches is the same patch position for every row it stages: decode it once
    float pr[kRounds][kCR];
    int p_r[kRounds], p_c[kRounds], p_off[kRounds];
    bool p_ok[kRounds];
#pragma unroll
    for (int i = 0; i < kRounds; ++i) {
        const int p = tid + i * kNT;
        p_r[i] = p / kPW;
        p_c[i] = p - p_r[i] * kPW;
        p_off[i] = (p_r[i] * Wi + p_c[i]) * 8;
        if (p >= kPix || p_c[i] >= kPW - 1) p_r[i] = 1 << 20;      // never a valid input row
    }
    // (n, ho, column block) of the row the next fetch brings in: stepped, not divided out of t -- scalar divisions by run-time
    // values are ~60 SALU instructions the matrix pipe would sit idle behind
    int f_cb = t_lo % cbn, f_ho = (t_lo / cbn) % Ho, f_n = (t_lo / cbn) / Ho;
    auto fetch = [&]() {
        const int hi0 = 2 * f_ho - 3, wi0 = 256 * f_cb - 3;
        const float* base = x8 + (((ptrdiff_t)f_n * Hi + hi0) * Wi + wi0) * 8;
        if (++f_cb == cbn) {
            f_cb = 0;
            if (++f_ho == Ho) {
                f_ho = 0;
                ++f_n;
            }
        }
#pragma unroll
        for (int i = 0; i < kRounds; ++i) {
            const bool ok = (unsigned)(hi0 + p_r[i]) < (unsigned)Hi && (unsigned)(wi0 + p_c[i]) < (unsigned)Wi;
            const float* src = ok ? base + p_off[i] : x8;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src);
            pr[i][0] = v.x; pr[i][1] = v.y; pr[i][2] = v.z; pr[i][3] = v.w;
            pr[i][4] = src[4];
            p_ok[i] = ok;
        }
    };
    // (the zeroing of the padding sits HERE, not next to the loads: the first use of a loaded register is where the wave
    // waits for memory, and that must not be in the middle of the MFMA stream)
    auto stage = [&](int buf) {
        float* dst = sP + buf * kPatch + tid * kCR;
#pragma unroll
        for (int i = 0; i < kRounds; ++i)
#pragma unroll
            for (int c = 0; c < kCR; ++c) dst[i * kNT * kCR + c] = p_ok[i] ? pr[i][c] : 0.f;
    };

    fetch();
    stage(0);
    __syncthreads();

    const int a_lane = kCR * 2 * (rg * 32 + l32);                   // float offset of this lane's pixel in a patch row
    const float* b_lane = sB + (cg * 32 + l32) * kKB;
    const int ch = cg * 32 + l32;
    const float bv = bias ? bias[ch] : 0.f;
    const float floor_v = relu ? 0.f : -__builtin_inff();
    f32x16 acc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
    int cur = 0;
    // one reduction row of the filter: 18 MFMAs of this wave's 32 pixels x 32 channels
    auto mma_row = [&](int r) {
        const float* ar = sP + cur * kPatch + a_lane + r * kPitch;
        const float* br = b_lane + r * 64 * kKB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 a0 = *reinterpret_cast<const f32x2*>(ar + 8 * q + 4 * h);
            const f32x2 a1 = *reinterpret_cast<const f32x2*>(ar + 8 * q + 4 * h + 2);
            const f32x4 b = *reinterpret_cast<const f32x4*>(br + 8 * q + 4 * h);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.w, acc, 0, 0, 0);
        }
        const f32x2 a = *reinterpret_cast<const f32x2*>(ar + 32 + 2 * h);
        const f32x2 b = *reinterpret_cast<const f32x2*>(br + 32 + 2 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    };
    // Statistics without a barrier of their own: every wave reduces ITS 32 pixels to (mean, sum of squared deviations from
    // it) per channel -- registers and one cross-half shuffle -- and parks the pair; one barrier later the four pairs of a
    // row are merged (Chan's update with equal counts) into the 128-row partial the finalize kernel expects.  sR is
    // double-buffered by row parity: the merge of row t reads while row t+1 parks.  Both steps run on every lane of every
    // wave -- lanes l and l + 32 hold the same channel, and all four pixel groups write the same merged pair to the same
    // address -- because a predicated store is a branch, and a branch ends the basic block the MFMAs are scheduled in.
    auto park = [&](int t) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += pacc[r];
        s += __shfl_xor(s, 32, 64);
        const float m = s * (1.f / 32.f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) q += (pacc[r] - m) * (pacc[r] - m);
        q += __shfl_xor(q, 32, 64);
        *reinterpret_cast<f32x2*>(sR + (((t & 1) * 4 + rg) * 64 + ch) * 2) = f32x2{m, q};
    };
    auto merge = [&](int t) {
        const float* rr = sR + ((t & 1) * 4 * 64 + ch) * 2;
        const f32x2 p0 = *reinterpret_cast<const f32x2*>(rr), p1 = *reinterpret_cast<const f32x2*>(rr + 128);
        const f32x2 p2 = *reinterpret_cast<const f32x2*>(rr + 256), p3 = *reinterpret_cast<const f32x2*>(rr + 384);
        const float mean = ((p0.x + p1.x) + (p2.x + p3.x)) * 0.25f;
        const float d0 = p0.x - mean, d1 = p1.x - mean, d2 = p2.x - mean, d3 = p3.x - mean;
        st_mean[(size_t)t * 64 + ch] = mean;
        st_m2[(size_t)t * 64 + ch] = ((p0.y + p1.y) + (p2.y + p3.y)) + 32.f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    };
    auto store_half = [&](int t, int half) {
        float* o = out + ((size_t)t * 128 + rg * 32 + 4 * h) * 64 + ch;
#pragma unroll
        for (int r = 8 * half; r < 8 * half + 8; ++r) {
            const float v = pacc[r] + bv;
            o[(size_t)((r & 3) + 8 * (r >> 2)) * 64] = v < floor_v ? floor_v : v;      // (a NaN stays a NaN)
        }
    };
    // The epilogue of row t-1 (statistics, stores), the merge of row t-2 and the fetch + staging of row t+1 are threaded
    // between the seven MFMA groups of row t.  Both waves of a SIMD sit in the same phase (the barrier sees to that), so
    // an epilogue run on its own is a stretch with nothing in the matrix pipe: measured 2.53 ms of which the bare
    // MFMA + fragment-read loop is 1.97.  STEADY (all four rows exist) is branch-free, ONE basic block for the scheduler.
    auto body = [&](auto steady, int t) {
        constexpr bool STEADY = decltype(steady)::value;
        const bool have = STEADY || t < t_hi, prev = STEADY || (t > t_lo && t <= t_hi), prev2 = STEADY || t > t_lo + 1;
        const bool more = STEADY || t + 1 < t_hi;
        if (more && !(IO_STEM_ABLATE & 2)) fetch();
        __builtin_amdgcn_sched_barrier(0);      // the loads go out HERE; left alone the scheduler sinks them to their use
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (have) mma_row(0);
        if (STATS && prev && !(IO_STEM_ABLATE & 4)) park(t - 1);
        if (have) mma_row(1);
        if (STATS && prev2 && !(IO_STEM_ABLATE & 4)) merge(t - 2);
        if (have) mma_row(2);
        if (prev && (!(IO_STEM_ABLATE & 1) || pacc[0] == 123.456f)) store_half(t - 1, 0);
        if (have) mma_row(3);
        if (prev && (!(IO_STEM_ABLATE & 1) || pacc[0] == 123.456f)) store_half(t - 1, 1);
        if (have) mma_row(4);
        if (more && !(IO_STEM_ABLATE & 2)) stage(cur ^ 1);   // (every wave is past the MFMAs of row t-1, that buffer's last readers)
        if (have) mma_row(5);
        if (have) mma_row(6);
        __syncthreads();
        pacc = acc;
        cur ^= 1;
    };
    for (int t = t_lo; t <= t_hi + 1; ++t) {
        if (t > t_lo + 1 && t + 1 < t_hi) body(std::true_type{}, t);
        else body(std::false_type{}, t);
    }
}

// ---- filter gradient -------------------------------------------------------------------------------------------------
// dW[ch][r][k'] = sum over output pixels of dy[pixel][ch] * patch[r][10 * w + k'] -- the same patch image, now the B
// operand: the reduction index of the MFMA is the PIXEL (two per instruction, lane half = pixel parity), the 32 columns
// of a tile are 32 consecutive entries of the packed filter row (k = r * 35 + k', 245 -> 256: exactly the [64][256]
// layout of the generic exact-K kernel, so the split-K reduction and the unpack that follow are unchanged).  A lane's
// column is fixed for the whole kernel: its B address is a per-lane constant plus 40 bytes per pixel, one ds_read_b32
// (the patch rows sit at a pitch of 1315 floats = 35 mod 32: the 32 columns of a tile that straddles two filter rows
// still hit 32 distinct banks).  dy rows are staged as they are ([128][64], A fragment = 32 consecutive channels of one
// pixel: conflict-free).  The 64 x 256 accumulators -- two 32 x 32 tiles per wave -- stay in registers across ALL rows of
// the block; there is no epilogue per row at all, only the fetch of the next row's patch and dy under the MFMAs of
// this one.  Each block leaves one [64][256] partial; splitk_reduce_kernel sums them in a fixed order.
constexpr int kWPitch = 1315;
constexpr int kWPatch = kRounds * kNT * kCR + 40;       // floats per patch buffer (slots of p >= kPix land behind the image)
constexpr int kWDy = 128 * 64;
constexpr size_t kWLds = (size_t)2 * (kWPatch + kWDy) * sizeof(float);

// XB: `dy` is the gradient of relu(bn1(y)) as the pooling backward leaves it, and the BatchNorm backward is applied
// while the rows are staged: dz = da where fma(y - mean, scale, shift) > 0 (bn_affine of bn.hip: the mask bit the
// reduction pass saw), dy = a * dz + b * y + c with the [G][64] tables of io_bn_bwd_coefs_t -- the stem has no data
// gradient, so this kernel is dy's only reader and the apply pass (read dz, read y, write dy) has no reason to exist.
template <bool XB>
__global__ __launch_bounds__(kNT, 1) void stem_wgrad_rows_kernel(const float* __restrict__ x8, const float* __restrict__ dy,
                                                                 float* __restrict__ partial, int Hi, int Wi, int Ho, int Wo,
                                                                 int ntiles, int per_block, IoStemXb xb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sP = smem;                       // [2][kWPatch]
    float* sD = smem + 2 * kWPatch;         // [2][128][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const int t_lo = blockIdx.x * per_block, t_hi = min(ntiles, t_lo + per_block);
    const int cbn = Wo >> 7;

    float pr[kRounds][kCR];
    f32x4 dr[4], yr[4], tb[6];              // tb: a, b, c, mean, scale, shift of this thread's 4 channels (XB)
    int p_r[kRounds], p_c[kRounds], p_off[kRounds], p_dst[kRounds];
    bool p_ok[kRounds];
#pragma unroll
    for (int i = 0; i < kRounds; ++i) {
        const int p = tid + i * kNT;
        p_r[i] = p / kPW;
        p_c[i] = p - p_r[i] * kPW;
        p_off[i] = (p_r[i] * Wi + p_c[i]) * 8;
        p_dst[i] = p_r[i] * kWPitch + p_c[i] * kCR;
        if (p >= kPix || p_c[i] >= kPW - 1) p_r[i] = 1 << 20;
    }
    int f_cb = t_lo % cbn, f_ho = (t_lo / cbn) % Ho, f_n = (t_lo / cbn) / Ho;
    const float* f_dy = dy + (size_t)t_lo * kWDy + tid * 4;
    const float* f_y = XB ? xb.y + (size_t)t_lo * kWDy + tid * 4 : nullptr;
    int f_t = t_lo;
    auto fetch = [&]() {
        const int hi0 = 2 * f_ho - 3, wi0 = 256 * f_cb - 3;
        const float* base = x8 + (((ptrdiff_t)f_n * Hi + hi0) * Wi + wi0) * 8;
        if (++f_cb == cbn) {
            f_cb = 0;
            if (++f_ho == Ho) {
                f_ho = 0;
                ++f_n;
            }
        }
#pragma unroll
        for (int i = 0; i < kRounds; ++i) {
            const bool ok = (unsigned)(hi0 + p_r[i]) < (unsigned)Hi && (unsigned)(wi0 + p_c[i]) < (unsigned)Wi;
            const float* src = ok ? base + p_off[i] : x8;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src);
            pr[i][0] = v.x; pr[i][1] = v.y; pr[i][2] = v.z; pr[i][3] = v.w;
            pr[i][4] = src[4];
            p_ok[i] = ok;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dr[i] = *reinterpret_cast<const f32x4*>(f_dy + i * kNT * 4);
        f_dy += kWDy;
        if constexpr (XB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) yr[i] = *reinterpret_cast<const f32x4*>(f_y + i * kNT * 4);
            f_y += kWDy;
            const int go = (f_t / xb.tiles_per_group) * 64 + (tid & 15) * 4;     // a row belongs to one sample group
            ++f_t;
            tb[0] = *reinterpret_cast<const f32x4*>(xb.a + go);
            tb[1] = *reinterpret_cast<const f32x4*>(xb.b + go);
            tb[2] = *reinterpret_cast<const f32x4*>(xb.c + go);
            tb[3] = *reinterpret_cast<const f32x4*>(xb.mean + go);
            tb[4] = *reinterpret_cast<const f32x4*>(xb.scale + go);
            tb[5] = *reinterpret_cast<const f32x4*>(xb.shift + go);
        }
    };
    auto stage = [&](int buf) {
        float* dst = sP + buf * kWPatch;
#pragma unroll
        for (int i = 0; i < kRounds; ++i)
#pragma unroll
            for (int c = 0; c < kCR; ++c) dst[p_dst[i] + c] = p_ok[i] ? pr[i][c] : 0.f;
        float* dd = sD + buf * kWDy + tid * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = dr[i];
            if constexpr (XB) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float act = __builtin_fmaf(yr[i][e] - tb[3][e], tb[4][e], tb[5][e]);
                    const float dz = act > 0.f ? v[e] : 0.f;
                    v[e] = __builtin_fmaf(tb[0][e], dz, __builtin_fmaf(tb[1][e], yr[i][e], tb[2][e]));
                }
            }
            *reinterpret_cast<f32x4*>(dd + i * kNT * 4) = v;
        }
    };

    fetch();
    stage(0);
    __syncthreads();

    // this lane's column of the packed filter row, and where that column starts in a patch image
    const int j = wave * 32 + l32;
    const int jr = min(j, 7 * 7 * kCR - 1) / (7 * kCR), jk = min(j, 7 * 7 * kCR - 1) - jr * 7 * kCR;
    const int b_lane = jr * kWPitch + jk + 2 * kCR * h;         // + 20 floats per pixel pair
    const int a_lane = h * 64 + l32;                            // + 128 floats per pixel pair; channels l32, l32 + 32
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    int cur = 0;
    // eight pixel pairs per chunk; the operands of chunk c + 1 are read while the 16 MFMAs of chunk c run (left to the
    // scheduler the reads of a chunk sit right in front of their first use: one exposed LDS latency per 8 MFMAs, 20 %)
    auto mma_tile = [&]() {
        const float* pb = sP + cur * kWPatch + b_lane;
        const float* pa = sD + cur * kWDy + a_lane;
        float fb[2][8], fa0[2][8], fa1[2][8];
        auto read_chunk = [&](int c, int s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                fb[s][i] = pb[(c * 8 + i) * 4 * kCR];
                fa0[s][i] = pa[(c * 8 + i) * 128];
                fa1[s][i] = pa[(c * 8 + i) * 128 + 32];
            }
        };
        read_chunk(0, 0);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c + 1 < 8) read_chunk(c + 1, (c + 1) & 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[c & 1][i], fb[c & 1][i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[c & 1][i], fb[c & 1][i], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t = t_lo; t < t_hi; ++t) {
        if (t + 1 < t_hi) {
            fetch();
            __builtin_amdgcn_sched_barrier(0);      // the loads go out here, ahead of the MFMAs they hide under
            mma_tile();
            stage(cur ^ 1);
        } else {
            mma_tile();
        }
        __syncthreads();
        cur ^= 1;
    }
    // D layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* o = partial + ((size_t)blockIdx.x * 64 + 4 * h) * 256 + j;
    const bool real = j < 7 * 7 * kCR;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        o[(size_t)row * 256] = real ? acc0[r] : 0.f;
        o[(size_t)(row + 32) * 256] = real ? acc1[r] : 0.f;
    }
}
}  // namespace

// 1 when io_launch_stem_rows takes this geometry: the 7x7 stride-2 pad-3 stem on the packed x8 input with 5 real channels,
// whole 128-pixel output rows.
bool io_stem_rows_ok(const IoConvGeom& g) {
    return g.cr == kCR && g.Ci == 8 && g.Co == 64 && g.Th == 7 && g.Tw == 7 && g.S == 7 && g.wT == 49 && g.is == 2 &&
           g.os == 1 && g.dh0 == -3 && g.dw0 == -3 && g.dhs == 1 && g.dws == 1 && g.r0 == 0 && g.rs == 1 && g.s0 == 0 &&
           g.ss == 1 && g.Ho == g.outH && g.Wo == g.outW && g.Wo % 128 == 0 && 2 * g.Ho == g.Hi && 2 * g.Wo == g.Wi &&
           (double)g.N * g.Ho * g.Wo < 2.0e9;
}

int io_launch_stem_rows(const IoConvGeom& g, const float* x8, const float* wp, float* out, float* st_mean, float* st_m2,
                        const float* bias, int relu, hipStream_t st) {
    IO_REQUIRE(io_stem_rows_ok(g), IO_ERR_SHAPE, "stem_rows: not the 7x7 stride-2 stem on whole 128-pixel output rows");
    const int ntiles = g.N * g.Ho * (g.Wo / 128);
    static std::atomic<unsigned long long> seen{0};
    static int ncu_of[64];
    int dev = 0;
    if (io_first_on_device(seen, &dev)) {
        hipDeviceProp_t prop;
        int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
        ncu_of[dev & 63] = n > 0 ? n : 256;
        (void)hipFuncSetAttribute((const void*)stem_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
        (void)hipFuncSetAttribute((const void*)stem_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    }
    const int ncu = ncu_of[dev & 63] > 0 ? ncu_of[dev & 63] : 256;
    const int blocks = ntiles < ncu ? ntiles : ncu;
    const int per_block = (ntiles + blocks - 1) / blocks;
    const dim3 grid((unsigned)((ntiles + per_block - 1) / per_block));
    if (st_mean)
        hipLaunchKernelGGL(stem_rows_kernel<true>, grid, dim3(kNT), kLds, st, x8, wp, io_stem_kp(g.wT, g.cr), out, st_mean,
                           st_m2, bias, relu, g.Hi, g.Wi, g.Ho, g.Wo, ntiles, per_block);
    else
        hipLaunchKernelGGL(stem_rows_kernel<false>, grid, dim3(kNT), kLds, st, x8, wp, io_stem_kp(g.wT, g.cr), out, st_mean,
                           st_m2, bias, relu, g.Hi, g.Wi, g.Ho, g.Wo, ntiles, per_block);
    return io_check_launch("stem_rows");
}

// blocks (= partial filter gradients) io_launch_stem_wgrad_rows may use: what the caller's workspace is sized for
int io_stem_wgrad_rows_max_blocks() { return 256; }

// The filter gradient of the same geometry into the packed exact-K rows dwp[64][256]: one partial per block, summed in a
// fixed order (io_splitk_reduce).  xb != nullptr: dy is the pooling backward's output and the BatchNorm backward rides
// in the staging (see the kernel).
int io_launch_stem_wgrad_rows(const IoConvGeom& g, const float* x8, const float* dy, float* dwp, float* partial,
                              size_t partial_bytes, hipStream_t st, const IoStemXb* xb) {
    IO_REQUIRE(io_stem_rows_ok(g), IO_ERR_SHAPE, "stem_wgrad_rows: not the 7x7 stride-2 stem on whole 128-pixel output rows");
    const int ntiles = g.N * g.Ho * (g.Wo / 128);
    static std::atomic<unsigned long long> seen{0};
    static int ncu_of[64];
    int dev = 0;
    if (io_first_on_device(seen, &dev)) {
        hipDeviceProp_t prop;
        int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
        ncu_of[dev & 63] = (n <= 0 || n > io_stem_wgrad_rows_max_blocks()) ? io_stem_wgrad_rows_max_blocks() : n;
        (void)hipFuncSetAttribute((const void*)stem_wgrad_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWLds);
        (void)hipFuncSetAttribute((const void*)stem_wgrad_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWLds);
    }
    const int ncu = ncu_of[dev & 63] > 0 ? ncu_of[dev & 63] : io_stem_wgrad_rows_max_blocks();
    const int blocks = ntiles < ncu ? ntiles : ncu;
    const int per_block = (ntiles + blocks - 1) / blocks;
    const int grid = (ntiles + per_block - 1) / per_block;
    IO_REQUIRE(partial && partial_bytes >= (size_t)grid * 64 * 256 * sizeof(float), IO_ERR_WORKSPACE,
               "stem_wgrad_rows: workspace %zu < %zu bytes", partial_bytes, (size_t)grid * 64 * 256 * sizeof(float));
    IoStemXb x{};
    if (xb) {
        x = *xb;
        IO_REQUIRE(x.y && x.a && x.b && x.c && x.mean && x.scale && x.shift && x.G >= 1 && g.N % x.G == 0, IO_ERR_SHAPE,
                   "stem_wgrad_rows: the fused BatchNorm backward needs y, six [G][64] tables and G | N");
        x.tiles_per_group = ntiles / x.G;
    }
    const double Md = (double)g.N * g.Ho * g.Wo;
    {
        IoProfScope prof(IO_PROF_WGRAD_STEM, 2.0 * Md * 64 * 49.0 * kCR,
                         4.0 * (Md * 64 * (xb ? 2.0 : 1.0) + (double)g.N * g.Hi * g.Wi * 8 + 64.0 * 49 * kCR), st);
        if (xb)
            hipLaunchKernelGGL(stem_wgrad_rows_kernel<true>, dim3((unsigned)grid), dim3(kNT), kWLds, st, x8, dy, partial, g.Hi,
                               g.Wi, g.Ho, g.Wo, ntiles, per_block, x);
        else
            hipLaunchKernelGGL(stem_wgrad_rows_kernel<false>, dim3((unsigned)grid), dim3(kNT), kWLds, st, x8, dy, partial, g.Hi,
                               g.Wi, g.Ho, g.Wo, ntiles, per_block, x);
        const int rc = io_check_launch("stem_wgrad_rows");
        if (rc) return rc;
        return io_splitk_reduce(partial, dwp, (size_t)64 * 256 / 4, grid, st);
    }
}
