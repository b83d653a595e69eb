// Operators the MiDaS branch (InstaDepthNet_od / _d) needs on top of the ResNet-50 path:
//   * grouped 3x3 convolution of ResNeXt-101 32x8d (resnet_cls.py:309-320 via midas/blocks.py:85-87), run by the
//     implicit-GEMM kernels as a block-diagonal convolution over 64-channel windows (IoConvGeom::gw): the filter
//     [C][cg][3][3] is expanded to [C][9][64] (zeros outside the group), so the waste is 64 / cg of a small layer
//     instead of C / cg of a dense emulation;
//   * bilinear x2 up-sampling, align_corners True (FeatureFusionBlock, midas/blocks.py:186-190) and False
//     (Interpolate in output_conv, midas/blocks.py:97-118), forward and exact adjoint (gather form, deterministic);
//   * bias (+ReLU), ReLU backward, add, column sums (bias gradients) for the biased 3x3 convolutions of the decoder
//     (midas/blocks.py:121-160, midas_net.py:134-141);
//   * the final 1x1 convolution to ONE channel (+ReLU) and its backward.
// All tensors NHWC fp32.
#include "io_common.h"

#ifndef IO_HEAD_ROWS
#define IO_HEAD_ROWS 1
#endif

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return io_bf2f(*p); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = io_f2bf(v); }

int ew_blocks(size_t n) {
    size_t b = (n + kThreads - 1) / kThreads;
    return (int)(b > 8192 ? 8192 : (b ? b : 1));
}

// ---- grouped filter <-> block-diagonal windows ---------------------------------------------------------------
// w [C][cg][T] (OIHW with T = R*S taps).  wc [C][T][64]: row o holds, for its 64-channel window wb = 64*(o/64), the
// taps of input channel wb + cl (zero unless that channel is in o's group).  wtc [C][T][64]: the same for the data
// gradient, row = INPUT channel c, column = output channel wb + cl.
template <typename T_>
__global__ __launch_bounds__(kThreads) void gconv_pack_kernel(const float* __restrict__ w, int C, int cg, int T,
                                                             T_* __restrict__ wc, T_* __restrict__ wtc) {
    const size_t total = (size_t)C * T * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cl = (int)(i & 63);
        const int tap = (int)((i >> 6) % T);
        const int o = (int)(i / ((size_t)T * 64));
        const int other = (o & ~63) + cl;
        const bool same = (other / cg) == (o / cg);
        st1(wc + i, same ? w[((size_t)o * cg + other % cg) * T + tap] : 0.f);
        st1(wtc + i, same ? w[((size_t)other * cg + o % cg) * T + tap] : 0.f);
    }
}

// dw [C][cg][T] <- dwc [C][T][64] (entries outside the group are gradients of structural zeros: dropped)
__global__ __launch_bounds__(kThreads) void gconv_unpack_kernel(const float* __restrict__ dwc, int C, int cg, int T,
                                                               float* __restrict__ dw) {
    const size_t total = (size_t)C * cg * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % T);
        const int cil = (int)((i / T) % cg);
        const int o = (int)(i / ((size_t)T * cg));
        const int ci = (o / cg) * cg + cil;
        dw[i] = dwc[((size_t)o * T + tap) * 64 + (ci - (o & ~63))];
    }
}

// ---- bilinear x2 ----------------------------------------------------------------------------------------------
// source index of nn.functional.interpolate(scale_factor=2, mode='bilinear'): align_corners -> dst*(H-1)/(2H-1);
// otherwise max(0.5*(dst+0.5)-0.5, 0)
struct Lerp {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ Lerp lerp_of(int dst, int H, int align) {
    const int OH = 2 * H;
    float src;
    if (align) {
        const float sc = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
        src = sc * (float)dst;
    } else {
        src = 0.5f * ((float)dst + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
    }
    Lerp l;
    l.i0 = (int)src;
    if (l.i0 > H - 1) l.i0 = H - 1;
    l.i1 = l.i0 + (l.i0 < H - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

// V channels per thread: 16 bytes of either storage type where the channel count allows (fp32: 4, bf16: 8; else 4).  The grid
// is (chunks of one output / input image row, image rows): the row decode and the row's interpolation weights are block
// uniform, a thread divides once by the chunks per pixel.  (The first form -- 8 bytes per lane in bf16, three 64-bit
// divisions per element -- ran the 384 x 384 x 128 map of the MiDaS output head at 2.0 TB/s.)
template <typename T_, int V> struct UpVec {
    f32x4 v[V / 4];
};
template <typename T_, int V> __device__ __forceinline__ UpVec<T_, V> up_ld(const T_* p) {
    UpVec<T_, V> r;
    if constexpr (V == 4) {
        r.v[0] = io_ldv(p);
    } else {
        static_assert(sizeof(T_) == 2 && V == 8, "8 channels per thread: bf16");
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        r.v[0][0] = __builtin_bit_cast(float, q.x << 16); r.v[0][1] = __builtin_bit_cast(float, q.x & 0xffff0000u);
        r.v[0][2] = __builtin_bit_cast(float, q.y << 16); r.v[0][3] = __builtin_bit_cast(float, q.y & 0xffff0000u);
        r.v[1][0] = __builtin_bit_cast(float, q.z << 16); r.v[1][1] = __builtin_bit_cast(float, q.z & 0xffff0000u);
        r.v[1][2] = __builtin_bit_cast(float, q.w << 16); r.v[1][3] = __builtin_bit_cast(float, q.w & 0xffff0000u);
    }
    return r;
}
template <typename T_, int V> __device__ __forceinline__ void up_st(T_* p, const UpVec<T_, V>& r) {
    if constexpr (V == 4) {
        io_stv(p, r.v[0]);
    } else {
        uint4 q;
        q.x = io_f2bf2(r.v[0][0], r.v[0][1]); q.y = io_f2bf2(r.v[0][2], r.v[0][3]);
        q.z = io_f2bf2(r.v[1][0], r.v[1][1]); q.w = io_f2bf2(r.v[1][2], r.v[1][3]);
        *reinterpret_cast<uint4*>(p) = q;
    }
}

template <typename T_, int V>
__global__ __launch_bounds__(kThreads) void upsample2x_fwd_kernel(const T_* __restrict__ x, int N, int H, int W,
                                                                 int CV, int align, T_* __restrict__ out) {
    const int OH = 2 * H, OW = 2 * W, rowlen = OW * CV;
    for (int row = blockIdx.y; row < N * OH; row += gridDim.y) {
        const int n = row / OH, oh = row - n * OH;
        const Lerp lh = lerp_of(oh, H, align);
        const T_* x0 = x + ((size_t)n * H + lh.i0) * W * CV * V;
        const T_* x1 = x + ((size_t)n * H + lh.i1) * W * CV * V;
        T_* o = out + (size_t)row * rowlen * V;
        for (int j = blockIdx.x * kThreads + threadIdx.x; j < rowlen; j += gridDim.x * kThreads) {
            const int ow = j / CV, q = j - ow * CV;
            const Lerp lw = lerp_of(ow, W, align);
            const UpVec<T_, V> a = up_ld<T_, V>(x0 + ((size_t)lw.i0 * CV + q) * V), b = up_ld<T_, V>(x0 + ((size_t)lw.i1 * CV + q) * V);
            const UpVec<T_, V> c = up_ld<T_, V>(x1 + ((size_t)lw.i0 * CV + q) * V), d = up_ld<T_, V>(x1 + ((size_t)lw.i1 * CV + q) * V);
            UpVec<T_, V> r;
            // same association as PyTorch's CPU kernel: h0lambda*(w0lambda*a + w1lambda*b) + h1lambda*(...)
#pragma unroll
            for (int k = 0; k < V / 4; ++k)
                r.v[k] = lh.w0 * (lw.w0 * a.v[k] + lw.w1 * b.v[k]) + lh.w1 * (lw.w0 * c.v[k] + lw.w1 * d.v[k]);
            up_st<T_, V>(o + (size_t)j * V, r);
        }
    }
}

// exact adjoint in gather form: input pixel (h, w) collects from every output pixel whose stencil touches it
template <typename T_, int V>
__global__ __launch_bounds__(kThreads) void upsample2x_bwd_kernel(const T_* __restrict__ dy, int N, int H, int W,
                                                                 int CV, int align, T_* __restrict__ dx) {
    const int OH = 2 * H, OW = 2 * W, rowlen = W * CV;
    for (int row = blockIdx.y; row < N * H; row += gridDim.y) {
        const int n = row / H, h = row - n * H;
        const T_* dp = dy + (size_t)n * OH * OW * CV * V;
        const int oh0 = max(0, 2 * h - 3), oh1 = min(OH - 1, 2 * h + 4);
        T_* o = dx + (size_t)row * rowlen * V;
        for (int j = blockIdx.x * kThreads + threadIdx.x; j < rowlen; j += gridDim.x * kThreads) {
            const int w = j / CV, q = j - w * CV;
            UpVec<T_, V> acc;
#pragma unroll
            for (int k = 0; k < V / 4; ++k) acc.v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the column weights do not depend on the row: 8 stencil evaluations per element, not 64 (same products, same
            // order of accumulation)
            float wws[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ow = 2 * w - 3 + k;
                wws[k] = 0.f;
                if ((unsigned)ow < (unsigned)OW) {
                    const Lerp lw = lerp_of(ow, W, align);
                    wws[k] = (lw.i0 == w ? lw.w0 : 0.f) + (lw.i1 == w ? lw.w1 : 0.f);
                }
            }
            for (int oh = oh0; oh <= oh1; ++oh) {
                const Lerp lh = lerp_of(oh, H, align);
                const float wh = (lh.i0 == h ? lh.w0 : 0.f) + (lh.i1 == h ? lh.w1 : 0.f);
                if (wh == 0.f) continue;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (wws[k] == 0.f) continue;
                    const UpVec<T_, V> g = up_ld<T_, V>(dp + (((size_t)oh * OW + (2 * w - 3 + k)) * CV + q) * V);
#pragma unroll
                    for (int e = 0; e < V / 4; ++e) acc.v[e] += (wh * wws[k]) * g.v[e];
                }
            }
            up_st<T_, V>(o + (size_t)j * V, acc);
        }
    }
}

// ---- elementwise ------------------------------------------------------------------------------------------------
template <typename T_>
__global__ __launch_bounds__(kThreads) void bias_act_kernel(const T_* x, const float* __restrict__ bias, size_t total4,
                                                           int C4, int relu, T_* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = io_ldv(x + i * 4);
        if (bias) v += reinterpret_cast<const f32x4*>(bias)[i % C4];
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        }
        io_stv(out + i * 4, v);
    }
}

template <typename T_>
__global__ __launch_bounds__(kThreads) void relu_bwd_kernel(const T_* dy, const T_* __restrict__ act, size_t total4,
                                                           T_* dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 g = io_ldv(dy + i * 4);
        const f32x4 a = io_ldv(act + i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = a[k] > 0.f ? g[k] : 0.f;
        io_stv(dx + i * 4, g);
    }
}

template <typename T_>
__global__ __launch_bounds__(kThreads) void add_kernel(const T_* a, const T_* b, size_t total4, T_* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x)
        io_stv(out + i * 4, io_ldv(a + i * 4) + io_ldv(b + i * 4));
}

// ---- column sums: out[c] = sum_m x[m][c] (bias gradients), two levels, fixed order ---------------------------------
// block b sums rows [b*rpb, ...) into partial[b][C]; thread = (column tx = tid % C, row lane ty = tid / C)
template <typename T_>
__global__ __launch_bounds__(kThreads) void colsum_partial_kernel(const T_* __restrict__ x, int M, int C, int rpb,
                                                                 float* __restrict__ partial) {
    __shared__ float red[kThreads];
    const int TY = kThreads / C, tx = threadIdx.x % C, ty = threadIdx.x / C;
    const int r0 = blockIdx.x * rpb, r1 = min(r0 + rpb, M);
    float s = 0.f;
    for (int r = r0 + ty; r < r1; r += TY) s += ld1(x + (size_t)r * C + tx);
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0) {
        for (int k = 1; k < TY; ++k) s += red[k * C + tx];
        partial[(size_t)blockIdx.x * C + tx] = s;
    }
}
// one 64-lane wave per column: lanes stride over the nb partials (fp64), then a butterfly reduction -- fixed order
__global__ void colsum_final_kernel(const float* __restrict__ partial, int nb, int stride, int C,
                                    float* __restrict__ out) {
    const int c = blockIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) s += (double)partial[(size_t)b * stride + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) out[c] = (float)s;
}

// ---- 1x1 convolution to one channel (+ReLU), midas_net.py:139-140 ---------------------------------------------------
// out[m] = act(b + sum_c x[m*pitch + c] * w[c]); one thread per row (C <= 64 floats, contiguous)
template <typename T_>
__global__ __launch_bounds__(kThreads) void head1_fwd_kernel(const T_* __restrict__ x, int M, int pitch, int C,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            int relu, float* __restrict__ out) {
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < (size_t)M; m += (size_t)gridDim.x * blockDim.x) {
        const T_* xp = x + m * pitch;
        float s = b ? b[0] : 0.f;
        for (int q = 0; q < C / 4; ++q) {
            const f32x4 v = io_ldv(xp + q * 4), ww = reinterpret_cast<const f32x4*>(w)[q];
            s += v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3];
        }
        out[m] = (relu && s < 0.f) ? 0.f : s;
    }
}

// dz = dy * [out > 0]; dx[m][c] = dz * w[c] (zero in the padding channels C..pitch); partial[b][C+1] = block sums of
// dz * x[m][c] and of dz
template <typename T_>
__global__ __launch_bounds__(kThreads) void head1_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                                                            const T_* __restrict__ x, int M, int pitch, int C,
                                                            const float* __restrict__ w, int relu, int rpb,
                                                            T_* __restrict__ dx, float* __restrict__ partial) {
    __shared__ float red[kThreads];
    const int TXN = 64;                                  // column lanes (>= C, padded)
    const int TY = kThreads / TXN, tx = threadIdx.x % TXN, ty = threadIdx.x / TXN;
    const int r0 = blockIdx.x * rpb, r1 = min(r0 + rpb, M);
    const float wc = tx < C ? w[tx] : 0.f;
    float s = 0.f, sb = 0.f;
    for (int r = r0 + ty; r < r1; r += TY) {
        float dz = dy[r];
        if (relu && !(out[r] > 0.f)) dz = 0.f;
        if (tx < pitch) st1(dx + (size_t)r * pitch + tx, dz * wc);
        if (tx < C) s += dz * ld1(x + (size_t)r * pitch + tx);
        sb += dz;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0) {
        for (int k = 1; k < TY; ++k) s += red[k * TXN + tx];
        if (tx < C) partial[(size_t)blockIdx.x * (C + 1) + tx] = s;
    }
    __syncthreads();
    red[threadIdx.x] = sb;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < TY; ++k) t += red[k * TXN];
        partial[(size_t)blockIdx.x * (C + 1) + C] = t;
    }
}

// Row forms of the two kernels above for rows that are whole 16-byte chunks (pitch = NQ chunks of VEC = 4 fp32 / 8 bf16
// channels): one thread per row, 16-byte loads / stores, the filter in registers, and -- backward -- the C sums of
// dz * x in registers, reduced once per block (wave shuffles, then LDS across the four waves).  The column-lane form
// above moves 2..4 bytes per lane.
template <typename T_> struct HeadChunk;
template <> struct HeadChunk<float> {
    static constexpr int VEC = 4;
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p);
        v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
};
template <> struct HeadChunk<bf16_t> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __builtin_bit_cast(float, w[k] << 16);
            v[2 * k + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
        uint4 r;
        r.x = io_f2bf2(v[0], v[1]); r.y = io_f2bf2(v[2], v[3]); r.z = io_f2bf2(v[4], v[5]); r.w = io_f2bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = r;
    }
};

template <typename T_, int NQ>
__global__ __launch_bounds__(kThreads) void head1_fwd_rows_kernel(const T_* __restrict__ x, int M, int C,
                                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                                 int relu, float* __restrict__ out) {
    constexpr int VEC = HeadChunk<T_>::VEC, P = NQ * VEC;
    float wr[P];
#pragma unroll
    for (int c = 0; c < P; ++c) wr[c] = c < C ? w[c] : 0.f;
    const float b0 = b ? b[0] : 0.f;
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < (size_t)M; m += (size_t)gridDim.x * blockDim.x) {
        const T_* xp = x + m * P;
        float s = b0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float v[VEC];
            HeadChunk<T_>::load(xp + q * VEC, v);
            // (the element order and the grouping by four of head1_fwd_kernel: the same sums)
#pragma unroll
            for (int e = 0; e < VEC; e += 4)
                if (q * VEC + e < C)                   // (the padding channels are not read into the sum)
                    s += v[e] * wr[q * VEC + e] + v[e + 1] * wr[q * VEC + e + 1] + v[e + 2] * wr[q * VEC + e + 2] +
                         v[e + 3] * wr[q * VEC + e + 3];
        }
        out[m] = (relu && s < 0.f) ? 0.f : s;
    }
}

template <typename T_, int NQ>
__global__ __launch_bounds__(kThreads) void head1_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                                                                 const T_* __restrict__ x, int M, int C,
                                                                 const float* __restrict__ w, int relu, int rpb,
                                                                 T_* __restrict__ dx, float* __restrict__ partial) {
    constexpr int VEC = HeadChunk<T_>::VEC, P = NQ * VEC;
    __shared__ float red[kThreads / 64][P + 1];
    float wr[P], acc[P];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        wr[c] = c < C ? w[c] : 0.f;
        acc[c] = 0.f;
    }
    float sb = 0.f;
    const int r0 = blockIdx.x * rpb, r1 = min(r0 + rpb, M);
    for (int r = r0 + threadIdx.x; r < r1; r += kThreads) {
        float dz = dy[r];
        if (relu && !(out[r] > 0.f)) dz = 0.f;
        sb += dz;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float v[VEC], o[VEC];
            HeadChunk<T_>::load(x + (size_t)r * P + q * VEC, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                acc[q * VEC + e] += dz * v[e];
                o[e] = dz * wr[q * VEC + e];           // (zero in the padding channels: wr is)
            }
            HeadChunk<T_>::store(dx + (size_t)r * P + q * VEC, o);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c <= P; ++c) {
        float v = c < P ? acc[c < P ? c : 0] : sb;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][c] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x <= C) {
        const int c = (int)threadIdx.x < C ? (int)threadIdx.x : P;     // entry C of the partial row = the sum of dz
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kThreads / 64; ++k) t += red[k][c];
        partial[(size_t)blockIdx.x * (C + 1) + threadIdx.x] = t;
    }
}

// column sums with 16-byte chunks per lane: thread = (chunk q = tid % (C / VEC), row lane tid / (C / VEC)); the element
// form (colsum_partial_kernel) moves 2..4 bytes per lane
template <typename T_>
__global__ __launch_bounds__(kThreads) void colsum_partial_rows_kernel(const T_* __restrict__ x, int M, int C, int rpb,
                                                                      float* __restrict__ partial) {
    constexpr int VEC = HeadChunk<T_>::VEC;
    __shared__ float red[kThreads][VEC + 1];
    const int CV = C / VEC, TY = kThreads / CV, q = threadIdx.x % CV, ty = threadIdx.x / CV;
    const int r0 = blockIdx.x * rpb, r1 = min(r0 + rpb, M);
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    for (int r = r0 + ty; r < r1; r += TY) {
        float v[VEC];
        HeadChunk<T_>::load(x + (size_t)r * C + q * VEC, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    if (ty == 0) {
        for (int k = 1; k < TY; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += red[k * CV + q][e];
#pragma unroll
        for (int e = 0; e < VEC; ++e) partial[(size_t)blockIdx.x * C + q * VEC + e] = s[e];
    }
}

int rows_per_block(int M, int* nb) {
    int want = 1024;
    int rpb = (M + want - 1) / want;
    if (rpb < 64) rpb = 64;
    *nb = (M + rpb - 1) / rpb;
    return rpb;
}

}  // namespace

#define IO_DT_REQUIRE(dt_) IO_REQUIRE((dt_) == IO_F32 || (dt_) == IO_BF16, IO_ERR_SHAPE, "unknown dtype %d", (dt_))
#define IO_BY_DTYPE(dt_, CALL_)          \
    do {                                 \
        if ((dt_) == IO_BF16) {          \
            typedef bf16_t T_;           \
            CALL_;                       \
        } else {                         \
            typedef float T_;            \
            CALL_;                       \
        }                                \
    } while (0)

extern "C" int io_gconv_pack(const float* w, int C, int cg, int taps, void* wc, void* wtc, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(C % 64 == 0 && cg >= 1 && cg <= 64 && 64 % cg == 0, IO_ERR_SHAPE,
               "gconv_pack: C=%d must be a multiple of 64 and the group width %d must divide 64", C, cg);
    const size_t total = (size_t)C * taps * 64;
    IoProfScope prof(IO_PROF_TRANSPOSE, 0.0, (4.0 + 2.0 * io_dtype_bytes(dt)) * total, st);
    IO_BY_DTYPE(dt, hipLaunchKernelGGL(gconv_pack_kernel<T_>, dim3(ew_blocks(total)), dim3(kThreads), 0, st, w, C, cg,
                                       taps, (T_*)wc, (T_*)wtc));
    return io_check_launch("gconv_pack");
}

extern "C" int io_gconv_unpack_grad(const float* dwc, int C, int cg, int taps, float* dw, hipStream_t st) {
    IO_REQUIRE(C % 64 == 0 && cg >= 1 && cg <= 64 && 64 % cg == 0, IO_ERR_SHAPE, "gconv_unpack_grad: C=%d cg=%d", C, cg);
    const size_t total = (size_t)C * cg * taps;
    IoProfScope prof(IO_PROF_TRANSPOSE, 0.0, 8.0 * total, st);
    hipLaunchKernelGGL(gconv_unpack_kernel, dim3(ew_blocks(total)), dim3(kThreads), 0, st, dwc, C, cg, taps, dw);
    return io_check_launch("gconv_unpack_grad");
}

// ---- all filters of a module tree in one launch ------------------------------------------------------------------
// The op-by-op graphs (instaorder_amd.ops) re-lay every filter out per call: OIHW fp32 master -> [Cout'][taps][Cin']
// operand (activation type, channels zero-padded) for the forward / filter-gradient kernels, its transpose
// [Cin'][taps][Cout'] for the data gradient, and the filter gradient back to OIHW -- three to six tiny launches per
// convolution, ~1300 per step of the MiDaS-based nets.  One table-driven launch per direction does all of them.
struct IoWeightDesc {
    long src;            // float offset of the OIHW master in the flat parameter / gradient buffer
    long dst_op, dst_t;  // element offsets of the operand and of its transpose in the operand buffer (dst_t < 0: none)
    long dst_g;          // float offset of the [Cout'][taps][Cin'] filter gradient in the gradient staging buffer
    int Co, Ci, T, Cop, Cip;
};

// Tiles of 32 output channels x 32 input channels x up to 9 taps go through LDS so that the read of the OIHW master
// (runs of 32 * T floats) and both writes ([o][t][c]: c fastest, [c][t][o]: o fastest) are contiguous; the first,
// element-per-thread version wrote the transpose 2 bytes at a time with a stride of Cout' and took 2.3 ms per call for
// the 105 M parameters of the MiDaS tree (3.4 % of its training step).
// Round 4: the tile body is instantiated for the two tap counts that carry the parameters (T = 1: 64 x 64 channel tiles,
// 128-byte bf16 runs on both writes; T = 9) so that every index split is by a compile-time constant -- the generic body
// spends ~40 integer instructions per element on divisions by run-time tile extents (0.84 ms per step for 105 M parameters).
template <typename T, int NTC>      // NTC: taps per tile at compile time (0: run-time, tiles of up to 9 taps)
__device__ __forceinline__ void weights_tile(const IoWeightDesc& d, const float* __restrict__ params, T* __restrict__ ops,
                                             float* s, int tile, int ntc, int ntt, int tid) {
    constexpr int TO = NTC == 1 ? 64 : 32, TC = NTC == 1 ? 64 : 32, TT = NTC ? NTC : 9, PITCH = TC * TT + 1;
    auto put = [&](long idx, float v) {
        if constexpr (sizeof(T) == 2) ops[idx] = io_f2bf(v);
        else ops[idx] = v;
    };
    const int tt = tile % ntt, r = tile / ntt;
    const int o0 = (r / ntc) * TO, c0 = (r % ntc) * TC, t0 = tt * TT;
    const int nt = NTC ? NTC : min(TT, d.T - t0), row = TC * nt, n = TO * row;
    __syncthreads();                                   // the previous tile has been written out
    for (int e = tid; e < n; e += 256) {               // read: (c, t) fastest = contiguous in the master when nt == T
        const int ol = e / row, rest = e - ol * row;
        const int cl = rest / nt, tl = rest - cl * nt;
        const int o = o0 + ol, c = c0 + cl;
        s[ol * PITCH + rest] = (o < d.Co && c < d.Ci) ? params[d.src + ((long)o * d.Ci + c) * d.T + t0 + tl] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < n; e += 256) {               // operand [o][t][c]
        const int ol = e / row, rest = e - ol * row;
        const int tl = rest / TC, cl = rest - tl * TC;
        const int o = o0 + ol, c = c0 + cl;
        if (o < d.Cop && c < d.Cip) put(d.dst_op + ((long)o * d.T + t0 + tl) * d.Cip + c, s[ol * PITCH + cl * nt + tl]);
    }
    if (d.dst_t >= 0) {
        for (int e = tid; e < n; e += 256) {           // transpose [c][t][o]
            const int cl = e / (nt * TO), rest = e - cl * (nt * TO);
            const int tl = rest / TO, ol = rest - tl * TO;
            const int o = o0 + ol, c = c0 + cl;
            if (o < d.Cop && c < d.Cip) put(d.dst_t + ((long)c * d.T + t0 + tl) * d.Cop + o, s[ol * PITCH + cl * nt + tl]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void weights_prepare_kernel(const IoWeightDesc* __restrict__ tab,
                                                             const float* __restrict__ params, T* __restrict__ ops) {
    __shared__ float s[64 * 65 > 32 * (32 * 9 + 1) ? 64 * 65 : 32 * (32 * 9 + 1)];
    const IoWeightDesc d = tab[blockIdx.y];
    const int tid = threadIdx.x;
    if (d.T == 1) {
        const int nto = (d.Cop + 63) / 64, ntc = (d.Cip + 63) / 64;
        for (int tile = blockIdx.x; tile < nto * ntc; tile += gridDim.x) weights_tile<T, 1>(d, params, ops, s, tile, ntc, 1, tid);
    } else if (d.T == 9) {
        const int nto = (d.Cop + 31) / 32, ntc = (d.Cip + 31) / 32;
        for (int tile = blockIdx.x; tile < nto * ntc; tile += gridDim.x) weights_tile<T, 9>(d, params, ops, s, tile, ntc, 1, tid);
    } else {
        const int nto = (d.Cop + 31) / 32, ntc = (d.Cip + 31) / 32, ntt = (d.T + 8) / 9;
        for (int tile = blockIdx.x; tile < nto * ntc * ntt; tile += gridDim.x)
            weights_tile<T, 0>(d, params, ops, s, tile, ntc, ntt, tid);
    }
}

__global__ __launch_bounds__(256) void weights_unpack_grads_kernel(const IoWeightDesc* __restrict__ tab,
                                                                  const float* __restrict__ gk,
                                                                  float* __restrict__ grads) {
    const IoWeightDesc d = tab[blockIdx.y];
    const long total = (long)d.Co * d.Ci * d.T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % d.T);
        const long r = i / d.T;
        const int c = (int)(r % d.Ci), o = (int)(r / d.Ci);
        grads[d.src + i] = gk[d.dst_g + ((long)o * d.T + t) * d.Cip + c];
    }
}

extern "C" int io_weights_prepare(const void* table, int n, const float* params, void* ops, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(n > 0 && n < 65536 && table && params && ops, IO_ERR_SHAPE, "weights_prepare: bad table (n=%d)", n);
    IoProfScope prof(IO_PROF_TRANSPOSE, 0.0, 0.0, st);
    IO_BY_DTYPE(dt, hipLaunchKernelGGL(weights_prepare_kernel<T_>, dim3(64, n), dim3(256), 0, st,
                                       (const IoWeightDesc*)table, params, (T_*)ops));
    return io_check_launch("weights_prepare");
}

extern "C" int io_weights_unpack_grads(const void* table, int n, const float* gk, float* grads, hipStream_t st) {
    IO_REQUIRE(n > 0 && n < 65536 && table && gk && grads, IO_ERR_SHAPE, "weights_unpack_grads: bad table (n=%d)", n);
    IoProfScope prof(IO_PROF_TRANSPOSE, 0.0, 0.0, st);
    hipLaunchKernelGGL(weights_unpack_grads_kernel, dim3(64, n), dim3(256), 0, st, (const IoWeightDesc*)table, gk, grads);
    return io_check_launch("weights_unpack_grads");
}

static IoConvGeom gconv_geom(int N, int H, int W, int C, int R, int S, int stride, int pad) {
    IoConvGeom g = io_geom_fwd(N, H, W, C, C, R, S, stride, pad);
    g.gw = 64;
    return g;
}

extern "C" int io_gconv2d_fwd(const void* x, const void* wc, void* y, int N, int H, int W, int C, int R, int S,
                              int stride, int pad, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    return io_launch_conv_nt(gconv_geom(N, H, W, C, R, S, stride, pad), x, wc, y, nullptr, nullptr, 0, st, nullptr,
                             nullptr, nullptr, dt, dt);
}

extern "C" int io_gconv2d_dgrad(const void* dy, const void* wtc, void* dx, int N, int H, int W, int C, int R, int S,
                                int stride, int pad, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    return io_run_dgrad(dy, wtc, dx, nullptr, nullptr, N, H, W, C, C, R, S, stride, pad, st, nullptr, dt, 64);
}

extern "C" size_t io_gconv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int R, int S, int stride, int pad) {
    return io_conv_wgrad_partial_bytes(gconv_geom(N, H, W, C, R, S, stride, pad), 0);
}

extern "C" int io_gconv2d_wgrad(const void* x, const void* dy, float* dwc, int N, int H, int W, int C, int R, int S,
                                int stride, int pad, void* ws, size_t ws_bytes, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    return io_launch_conv_wgrad(gconv_geom(N, H, W, C, R, S, stride, pad), x, dy, dwc, (float*)ws, ws_bytes, 0, st, dt,
                                dt);
}

// grid of the row-decomposed upsample kernels: x over the chunks of one image row, y over the image rows
static dim3 up_grid(long rowlen, long rows) {
    long gx = (rowlen + kThreads - 1) / kThreads;
    return dim3((unsigned)(gx < 1 ? 1 : (gx > 64 ? 64 : gx)), (unsigned)(rows > 65535 ? 65535 : rows));
}
#define IO_UPSAMPLE_LAUNCH(K_, ROWLEN_, ROWS_, ...)                                                                    \
    do {                                                                                                              \
        if (dt == IO_BF16 && C % 8 == 0)                                                                              \
            hipLaunchKernelGGL((K_<bf16_t, 8>), up_grid((long)(ROWLEN_) * (C / 8), ROWS_), dim3(kThreads), 0, st,     \
                               (const bf16_t*)src, N, H, W, C / 8, align_corners, (bf16_t*)dst);                      \
        else if (dt == IO_BF16)                                                                                       \
            hipLaunchKernelGGL((K_<bf16_t, 4>), up_grid((long)(ROWLEN_) * (C / 4), ROWS_), dim3(kThreads), 0, st,     \
                               (const bf16_t*)src, N, H, W, C / 4, align_corners, (bf16_t*)dst);                      \
        else                                                                                                          \
            hipLaunchKernelGGL((K_<float, 4>), up_grid((long)(ROWLEN_) * (C / 4), ROWS_), dim3(kThreads), 0, st,      \
                               (const float*)src, N, H, W, C / 4, align_corners, (float*)dst);                        \
    } while (0)

extern "C" int io_upsample2x_bilinear_fwd(const void* x, int N, int H, int W, int C, int align_corners, void* out,
                                          int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(C % 4 == 0 && N > 0 && H > 0 && W > 0 && (double)N * H * 2 < 2.0e9 && (double)W * 2 * C < 2.0e9, IO_ERR_SHAPE,
               "upsample2x: N=%d H=%d W=%d C=%d", N, H, W, C);
    IoProfScope prof(IO_PROF_POOL_HEAD, 0.0, 5.0 * io_dtype_bytes(dt) * N * H * W * C, st);
    const void* src = x;
    void* dst = out;
    IO_UPSAMPLE_LAUNCH(upsample2x_fwd_kernel, 2 * W, (long)N * 2 * H);
    return io_check_launch("upsample2x_fwd");
}

extern "C" int io_upsample2x_bilinear_bwd(const void* dy, int N, int H, int W, int C, int align_corners, void* dx,
                                          int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(C % 4 == 0 && N > 0 && H > 0 && W > 0 && (double)N * H * 2 < 2.0e9 && (double)W * 2 * C < 2.0e9, IO_ERR_SHAPE,
               "upsample2x: N=%d H=%d W=%d C=%d", N, H, W, C);
    IoProfScope prof(IO_PROF_POOL_HEAD, 0.0, 5.0 * io_dtype_bytes(dt) * N * H * W * C, st);
    const void* src = dy;
    void* dst = dx;
    IO_UPSAMPLE_LAUNCH(upsample2x_bwd_kernel, W, (long)N * H);
    return io_check_launch("upsample2x_bwd");
}
#undef IO_UPSAMPLE_LAUNCH

extern "C" int io_bias_act(const void* x, const float* bias, int M, int C, int relu, void* out, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(C % 4 == 0 && M > 0, IO_ERR_SHAPE, "bias_act: M=%d C=%d", M, C);
    const size_t total4 = (size_t)M * (C / 4);
    IoProfScope prof(IO_PROF_BN_APPLY, 0.0, 2.0 * io_dtype_bytes(dt) * M * C, st);
    IO_BY_DTYPE(dt, hipLaunchKernelGGL(bias_act_kernel<T_>, dim3(ew_blocks(total4)), dim3(kThreads), 0, st, (const T_*)x,
                                       bias, total4, C / 4, relu, (T_*)out));
    return io_check_launch("bias_act");
}

extern "C" int io_relu_bwd(const void* dy, const void* act, size_t n, void* dx, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(n % 4 == 0 && n > 0, IO_ERR_SHAPE, "relu_bwd: n=%zu", n);
    IoProfScope prof(IO_PROF_BN_BWD, 0.0, 3.0 * io_dtype_bytes(dt) * n, st);
    IO_BY_DTYPE(dt, hipLaunchKernelGGL(relu_bwd_kernel<T_>, dim3(ew_blocks(n / 4)), dim3(kThreads), 0, st, (const T_*)dy,
                                       (const T_*)act, n / 4, (T_*)dx));
    return io_check_launch("relu_bwd");
}

extern "C" int io_add(const void* a, const void* b, size_t n, void* out, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(n % 4 == 0 && n > 0, IO_ERR_SHAPE, "add: n=%zu", n);
    IoProfScope prof(IO_PROF_BN_APPLY, 0.0, 3.0 * io_dtype_bytes(dt) * n, st);
    IO_BY_DTYPE(dt, hipLaunchKernelGGL(add_kernel<T_>, dim3(ew_blocks(n / 4)), dim3(kThreads), 0, st, (const T_*)a,
                                       (const T_*)b, n / 4, (T_*)out));
    return io_check_launch("add");
}

extern "C" size_t io_colsum_partial_floats(int M, int C) {
    int nb;
    rows_per_block(M, &nb);
    return (size_t)nb * (C + 1);
}

extern "C" int io_colsum(const void* x, int M, int C, float* out, float* partial, size_t partial_floats, int dt,
                         hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(C >= 1 && C <= kThreads && kThreads % C == 0 && M > 0, IO_ERR_SHAPE, "colsum: C=%d must divide %d", C,
               kThreads);
    int nb;
    const int rpb = rows_per_block(M, &nb);
    IO_REQUIRE(partial_floats >= (size_t)nb * C, IO_ERR_WORKSPACE, "colsum: workspace %zu < %zu floats", partial_floats,
               (size_t)nb * C);
    IoProfScope prof(IO_PROF_BN_BWD, 0.0, (double)io_dtype_bytes(dt) * M * C, st);
    const int vec = 16 / io_dtype_bytes(dt);
    if (IO_HEAD_ROWS && C % vec == 0 && kThreads % (C / vec) == 0)
        IO_BY_DTYPE(dt, hipLaunchKernelGGL(colsum_partial_rows_kernel<T_>, dim3(nb), dim3(kThreads), 0, st, (const T_*)x, M,
                                           C, rpb, partial));
    else
        IO_BY_DTYPE(dt, hipLaunchKernelGGL(colsum_partial_kernel<T_>, dim3(nb), dim3(kThreads), 0, st, (const T_*)x, M, C,
                                           rpb, partial));
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, st, partial, nb, C, C, out);
    return io_check_launch("colsum");
}

extern "C" int io_head1_fwd(const void* x, int M, int pitch, int C, const float* w, const float* b, int relu,
                            float* out, int dt, hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(C % 4 == 0 && C <= 64 && pitch % 4 == 0 && pitch >= C && pitch <= 64 && M > 0, IO_ERR_SHAPE,
               "head1: C=%d pitch=%d (C <= pitch <= 64, multiples of 4)", C, pitch);
    IoProfScope prof(IO_PROF_POOL_HEAD, 2.0 * M * C, (double)M * (io_dtype_bytes(dt) * pitch + 4.0), st);
    const int nq = pitch * io_dtype_bytes(dt) / 16;        // whole 16-byte chunks per row: the row form
#define IO_H1F(NQ_)                                                                                                  \
    IO_BY_DTYPE(dt, hipLaunchKernelGGL((head1_fwd_rows_kernel<T_, NQ_>), dim3(ew_blocks((size_t)M)), dim3(kThreads), 0, \
                                       st, (const T_*)x, M, C, w, b, relu, out))
    if (IO_HEAD_ROWS && pitch * io_dtype_bytes(dt) % 16 == 0 && nq == 4) IO_H1F(4);
    else if (IO_HEAD_ROWS && pitch * io_dtype_bytes(dt) % 16 == 0 && nq == 8) IO_H1F(8);
    else
        IO_BY_DTYPE(dt, hipLaunchKernelGGL(head1_fwd_kernel<T_>, dim3(ew_blocks((size_t)M)), dim3(kThreads), 0, st,
                                           (const T_*)x, M, pitch, C, w, b, relu, out));
#undef IO_H1F
    return io_check_launch("head1_fwd");
}

extern "C" int io_head1_bwd(const float* dy, const float* out, const void* x, int M, int pitch, int C, const float* w,
                            int relu, void* dx, float* dw, float* db, float* partial, size_t partial_floats, int dt,
                            hipStream_t st) {
    IO_DT_REQUIRE(dt);
    IO_REQUIRE(C % 4 == 0 && C <= 64 && pitch % 4 == 0 && pitch >= C && pitch <= 64 && M > 0, IO_ERR_SHAPE,
               "head1: C=%d pitch=%d (C <= pitch <= 64, multiples of 4)", C, pitch);
    int nb;
    const int rpb = rows_per_block(M, &nb);
    IO_REQUIRE(partial_floats >= (size_t)nb * (C + 1), IO_ERR_WORKSPACE, "head1_bwd: workspace %zu < %zu floats",
               partial_floats, (size_t)nb * (C + 1));
    IoProfScope prof(IO_PROF_POOL_HEAD, 4.0 * M * C, (double)M * (2.0 * io_dtype_bytes(dt) * pitch + 8.0), st);
    const int nq = pitch * io_dtype_bytes(dt) / 16;
#define IO_H1B(NQ_)                                                                                                  \
    IO_BY_DTYPE(dt, hipLaunchKernelGGL((head1_bwd_rows_kernel<T_, NQ_>), dim3(nb), dim3(kThreads), 0, st, dy, out,   \
                                       (const T_*)x, M, C, w, relu, rpb, (T_*)dx, partial))
    if (IO_HEAD_ROWS && pitch * io_dtype_bytes(dt) % 16 == 0 && nq == 4) IO_H1B(4);
    else if (IO_HEAD_ROWS && pitch * io_dtype_bytes(dt) % 16 == 0 && nq == 8) IO_H1B(8);
    else
        IO_BY_DTYPE(dt, hipLaunchKernelGGL(head1_bwd_kernel<T_>, dim3(nb), dim3(kThreads), 0, st, dy, out, (const T_*)x, M,
                                           pitch, C, w, relu, rpb, (T_*)dx, partial));
#undef IO_H1B
    // the per-block sums are [nb][C+1]: column sums give dw[0..C) and db
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, st, partial, nb, C + 1, C, dw);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(64), 0, st, partial + C, nb, C + 1, 1, db);
    return io_check_launch("head1_bwd");
}

// =====================================================================================================================
// Losses of the MiDaS-based nets that are not order-head losses (models/supervised_order.py:152-173, 214-235)
// =====================================================================================================================
namespace {

constexpr float kSmoothEps = 1e-7f;

struct MinMaxIdx {
    float mn, mx, sum;
    int imn, imx;      // keys w * H + h: torch reduces dim 2 (h) first, then dim 3 (w) -- the first w, then the first h wins a tie
};
__device__ __forceinline__ void mm_merge(MinMaxIdx& a, const MinMaxIdx& b) {
    if (b.mn < a.mn || (b.mn == a.mn && b.imn < a.imn)) { a.mn = b.mn; a.imn = b.imn; }
    if (b.mx > a.mx || (b.mx == a.mx && b.imx < a.imx)) { a.mx = b.mx; a.imx = b.imx; }
    a.sum += b.sum;
}

constexpr int kStatThreads = 1024;
// one block of 1024 threads per sample (16 samples of 147 k pixels: the block size is the parallelism): min / max (+ the
// element torch's chained min(2).min(3) selects), sum.  stats[b][8] = {mn, mx, sum, key_min, key_max, -, -, -} (keys as
// float bit patterns of ints)
__global__ __launch_bounds__(kStatThreads) void smooth_stats_kernel(const float* __restrict__ disp, int H, int W,
                                                               float* __restrict__ stats) {
    __shared__ MinMaxIdx sh[kStatThreads];
    const int b = blockIdx.x, N = H * W;
    const float* d = disp + (size_t)b * N;
    MinMaxIdx a{INFINITY, -INFINITY, 0.f, 0x7fffffff, 0x7fffffff};
    for (int i = threadIdx.x; i < N; i += kStatThreads) {
        const int h = i / W, w = i - h * W;
        const float v = d[i];
        MinMaxIdx e{v, v, v, w * H + h, w * H + h};
        mm_merge(a, e);
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = kStatThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            MinMaxIdx t = sh[threadIdx.x];
            mm_merge(t, sh[threadIdx.x + off]);
            sh[threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* s = stats + (size_t)b * 8;
        s[0] = sh[0].mn; s[1] = sh[0].mx; s[2] = sh[0].sum;
        s[3] = __int_as_float(sh[0].imn); s[4] = __int_as_float(sh[0].imx);
    }
}

// n = ((disp - mn) * s) * r with s = 1 / (mx + eps), r = 1 / (mean((disp - mn) * s) + eps)
struct SmoothNorm { float mn, s, r; };
__device__ __forceinline__ SmoothNorm smooth_norm(const float* st, int N) {
    SmoothNorm q;
    q.mn = st[0];
    q.s = 1.f / (st[1] + kSmoothEps);
    const double m = ((double)st[2] - (double)N * (double)st[0]) * (double)q.s / (double)N;
    q.r = (float)(1.0 / (m + (double)kSmoothEps));
    return q;
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// edge weight exp(-mean_c |img(c, p) - img(c, q)|) of the pixel pair (p, q), img NCHW with 3 channels
__device__ __forceinline__ float edge_w(const float* img, int HW, int p, int q) {
    const float a = fabsf(img[p] - img[q]) + fabsf(img[HW + p] - img[HW + q]) + fabsf(img[2 * HW + p] - img[2 * HW + q]);
    return expf(-a * (1.f / 3.f));
}

// grid (blocks, B).  Per pixel: its two forward edge terms (loss) and g = dL/dn from its up-to-four incident edges;
// per block: partial sums {Lx, Ly, sum g, sum g * n} -> part[b][blk][4]
__global__ __launch_bounds__(kThreads) void smooth_fwd_kernel(const float* __restrict__ disp, const float* __restrict__ img,
                                                             int H, int W, const float* __restrict__ stats, float cx,
                                                             float cy, float* __restrict__ g, float* __restrict__ part) {
    __shared__ float red[4][kThreads];
    const int b = blockIdx.y, N = H * W;
    const float* d = disp + (size_t)b * N;
    const float* im = img + (size_t)b * 3 * N;
    const SmoothNorm q = smooth_norm(stats + (size_t)b * 8, N);
    const float k = q.s * q.r;
    float lx = 0.f, ly = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < N; i += gridDim.x * kThreads) {
        const int h = i / W, w = i - h * W;
        const float n0 = (d[i] - q.mn) * k;
        float gi = 0.f;
        if (w + 1 < W) {
            const float df = n0 - (d[i + 1] - q.mn) * k, e = edge_w(im, N, i, i + 1);
            lx += fabsf(df) * e;
            gi += sgn(df) * e * cx;
        }
        if (w > 0) gi -= sgn((d[i - 1] - q.mn) * k - n0) * edge_w(im, N, i - 1, i) * cx;
        if (h + 1 < H) {
            const float df = n0 - (d[i + W] - q.mn) * k, e = edge_w(im, N, i, i + W);
            ly += fabsf(df) * e;
            gi += sgn(df) * e * cy;
        }
        if (h > 0) gi -= sgn((d[i - W] - q.mn) * k - n0) * edge_w(im, N, i - W, i) * cy;
        g[(size_t)b * N + i] = gi;
        a1 += gi;
        a2 += gi * n0;
    }
    red[0][threadIdx.x] = lx; red[1][threadIdx.x] = ly; red[2][threadIdx.x] = a1; red[3][threadIdx.x] = a2;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int t = 0; t < 4; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 4) part[((size_t)b * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// one block: loss = sum_b,blk (Lx * cx + Ly * cy) * out_scale; stats[b][5] = sum g, stats[b][6] = sum g * n
__global__ __launch_bounds__(kThreads) void smooth_finalize_kernel(const float* __restrict__ part, int B, int nblk, float cx,
                                                                  float cy, float out_scale, float* __restrict__ stats,
                                                                  float* __restrict__ loss) {
    __shared__ double red[kThreads];
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        double l = 0.0, a1 = 0.0, a2 = 0.0;
        for (int j = threadIdx.x; j < nblk; j += kThreads) {
            const float* p = part + ((size_t)b * nblk + j) * 4;
            l += (double)p[0] * cx + (double)p[1] * cy;
            a1 += p[2];
            a2 += p[3];
        }
        double v[3] = {l, a1, a2};
        for (int t = 0; t < 3; ++t) {
            red[threadIdx.x] = v[t];
            __syncthreads();
            for (int off = kThreads / 2; off > 0; off >>= 1) {
                if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
                __syncthreads();
            }
            v[t] = red[0];
            __syncthreads();
        }
        tot += v[0];
        if (threadIdx.x == 0) { stats[(size_t)b * 8 + 5] = (float)v[1]; stats[(size_t)b * 8 + 6] = (float)v[2]; }
    }
    if (threadIdx.x == 0) *loss = (float)(tot * out_scale);
}

// ddisp (+)= gscale[0] * scale * dL/ddisp: through the mean normalisation, the min / max normalisation (their
// selected elements get the sums), see the derivation in DESIGN.md 3b
__global__ __launch_bounds__(kThreads) void smooth_bwd_kernel(const float* __restrict__ g, const float* __restrict__ stats,
                                                             const float* __restrict__ gscale, float scale, int H, int W,
                                                             int accumulate, float* __restrict__ ddisp) {
    const int b = blockIdx.y, N = H * W;
    const float* st = stats + (size_t)b * 8;
    const SmoothNorm q = smooth_norm(st, N);
    const float a1 = st[5], a2 = st[6];
    const int kmn = __float_as_int(st[3]), kmx = __float_as_int(st[4]);
    const float f = gscale[0] * scale, k = q.s * q.r, mean_g = a2 / (float)N;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < N; i += gridDim.x * kThreads) {
        const int h = i / W, w = i - h * W, key = w * H + h;
        float v = k * (g[(size_t)b * N + i] - mean_g);
        if (key == kmn) v -= k * (a1 - a2);
        if (key == kmx) v -= q.s * a2 * kSmoothEps * q.r;
        float* o = ddisp + (size_t)b * N + i;
        *o = accumulate ? *o + f * v : f * v;
    }
}

// ---- disparity-order count (supervised_order.py:152-173): no gradient, scipy.ndimage.binary_erosion (3x3 cross, border 0)
__device__ __forceinline__ bool eroded(const float* m, int H, int W, int h, int w) {
    if (h == 0 || w == 0 || h == H - 1 || w == W - 1) return false;
    const int i = h * W + w;
    return m[i] != 0.f && m[i - 1] != 0.f && m[i + 1] != 0.f && m[i - W] != 0.f && m[i + W] != 0.f;
}

// The batch is 8..64 samples of up to 147 k pixels, so a sample is spread over DB = disp_blocks(N) blocks and the two
// data-dependent phases are two launches: pass 0 leaves each block's (max over eroded mask 2, min over eroded mask 1) of
// both disparity maps, pass 1 folds a sample's DB partials (fixed order: exact for max / min) and leaves each block's four
// counts per map, the finalize launch turns the integer counts into the batch's scalar.  (One 1024-thread block per
// sample, the first form, took 0.5 ms of the 48 ms InstaDepthNet_od step: 16 of 256 CUs walking 147 k pixels twice.)
constexpr int kDispThreads = 256, kDispMaxBlocks = 64;
int disp_blocks(int N) {
    int b = (N + kDispThreads * 8 - 1) / (kDispThreads * 8);
    return b < 1 ? 1 : (b > kDispMaxBlocks ? kDispMaxBlocks : b);
}

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ext[b][blk][6]: max2 of map 1 / 2, min1 of map 1 / 2, any eroded pixel in mask 1 / 2 (as floats)
__global__ __launch_bounds__(kDispThreads) void disp_order_ext_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                                 const float* __restrict__ m1, const float* __restrict__ m2,
                                                                 int H, int W, float* __restrict__ ext) {
    __shared__ float sh[kDispThreads / 64][6];
    const int b = blockIdx.y, N = H * W;
    const float *p1 = d1 + (size_t)b * N, *p2 = d2 + (size_t)b * N, *q1 = m1 + (size_t)b * N, *q2 = m2 + (size_t)b * N;
    float mx2[2] = {-INFINITY, -INFINITY}, mn1[2] = {INFINITY, INFINITY};
    int any1 = 0, any2 = 0;
    for (int i = blockIdx.x * kDispThreads + threadIdx.x; i < N; i += gridDim.x * kDispThreads) {
        const int h = i / W, w = i - h * W;
        const bool e1 = eroded(q1, H, W, h, w), e2 = eroded(q2, H, W, h, w);
        const float v[2] = {p1[i], p2[i]};
        for (int t = 0; t < 2; ++t) {
            if (e2) mx2[t] = fmaxf(mx2[t], v[t]);
            if (e1) mn1[t] = fminf(mn1[t], v[t]);
        }
        any1 |= e1; any2 |= e2;
    }
    const float r[6] = {wave_max(mx2[0]), wave_max(mx2[1]), wave_min(mn1[0]), wave_min(mn1[1]),
                        wave_max((float)any1), wave_max((float)any2)};
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int t = 0; t < 6; ++t) sh[wave][t] = r[t];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int t = threadIdx.x;
        float v = sh[0][t];
        for (int k = 1; k < kDispThreads / 64; ++k) v = (t == 2 || t == 3) ? fminf(v, sh[k][t]) : fmaxf(v, sh[k][t]);
        ext[((size_t)b * gridDim.x + blockIdx.x) * 6 + t] = v;
    }
}

// cnt[b][blk][4]: le / ge counts of map 1, le / ge counts of map 2 (zeros when the pair is skipped)
__global__ __launch_bounds__(kDispThreads) void disp_order_cnt_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                                 const float* __restrict__ m1, const float* __restrict__ m2,
                                                                 const long* __restrict__ order, const long* __restrict__ ovl,
                                                                 int H, int W, const float* __restrict__ ext,
                                                                 int* __restrict__ cnt) {
    __shared__ float shx[6];
    __shared__ int shc[kDispThreads / 64][4];
    const int b = blockIdx.y, N = H * W, DB = gridDim.x;
    if (threadIdx.x < 64) {          // one wave folds the sample's DB <= 64 partials
        float r[6];
        for (int t = 0; t < 6; ++t) {
            const bool mn = t == 2 || t == 3;
            float v = (int)threadIdx.x < DB ? ext[((size_t)b * DB + threadIdx.x) * 6 + t] : (mn ? INFINITY : -INFINITY);
            r[t] = mn ? wave_min(v) : wave_max(v);
        }
        if (threadIdx.x == 0)
            for (int t = 0; t < 6; ++t) shx[t] = r[t];
    }
    __syncthreads();
    const float MX[2] = {shx[0], shx[1]}, MN[2] = {shx[2], shx[3]};
    const bool have = shx[4] > 0.f && shx[5] > 0.f;
    const long od = order[b];
    // (the reference's .max() / .min() of an empty selection would raise; such pairs are skipped)
    const bool use = ovl[b] == 0 && (od == 0 || od == 1) && have;
    int c[4] = {0, 0, 0, 0};
    if (use) {
        const float *p1 = d1 + (size_t)b * N, *p2 = d2 + (size_t)b * N, *q1 = m1 + (size_t)b * N, *q2 = m2 + (size_t)b * N;
        for (int i = blockIdx.x * kDispThreads + threadIdx.x; i < N; i += DB * kDispThreads) {
            const int h = i / W, w = i - h * W;
            const bool e1 = eroded(q1, H, W, h, w), e2 = eroded(q2, H, W, h, w);
            const float v[2] = {p1[i], p2[i]};
            for (int t = 0; t < 2; ++t) {
                c[2 * t] += (e1 && v[t] <= MX[t]) + (e2 && MN[t] <= v[t]);
                c[2 * t + 1] += (e1 && v[t] >= MX[t]) + (e2 && MN[t] >= v[t]);
            }
        }
    }
    const int wave = threadIdx.x >> 6;
    for (int t = 0; t < 4; ++t) {
        const int v = wave_sum(c[t]);
        if ((threadIdx.x & 63) == 0) shc[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        int v = 0;
        for (int k = 0; k < kDispThreads / 64; ++k) v += shc[k][threadIdx.x];
        cnt[((size_t)b * DB + blockIdx.x) * 4 + threadIdx.x] = v;
    }
}

// out = scale * sum over samples of (the two counts the pair's depth order selects)
__global__ __launch_bounds__(kThreads) void disp_order_finalize_kernel(const int* __restrict__ cnt, const long* __restrict__ order,
                                                                      int B, int DB, int le_order, float scale,
                                                                      float* __restrict__ out) {
    __shared__ double red[kThreads];
    double a = 0.0;
    for (int b = threadIdx.x; b < B; b += kThreads) {
        int c[4] = {0, 0, 0, 0};
        for (int k = 0; k < DB; ++k)
            for (int t = 0; t < 4; ++t) c[t] += cnt[((size_t)b * DB + k) * 4 + t];
        // disp1 uses `<=` when depth_order1 == le_order and `>=` otherwise; disp2 the other way round
        const bool le1 = order[b] == le_order;
        a += (double)(float)((le1 ? c[0] : c[1]) + (le1 ? c[3] : c[2]));
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (float)(red[0] * scale);
}

int smooth_blocks(int N) {
    int b = (N + kThreads * 4 - 1) / (kThreads * 4);
    return b < 1 ? 1 : (b > 512 ? 512 : b);
}

}  // namespace

extern "C" size_t io_smooth_loss_workspace_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * 8 + (size_t)B * smooth_blocks(H * W) * 4;
}

extern "C" int io_smooth_loss_fwd(const float* disp, const float* img, int B, int H, int W, float out_scale, float* loss,
                                  float* g, float* workspace, size_t workspace_floats, hipStream_t st) {
    IO_REQUIRE(B > 0 && H > 1 && W > 1 && (double)H * W < 2.0e9, IO_ERR_SHAPE, "smooth_loss: B=%d H=%d W=%d", B, H, W);
    IO_REQUIRE(workspace_floats >= io_smooth_loss_workspace_floats(B, H, W), IO_ERR_WORKSPACE, "smooth_loss: workspace too small");
    const int N = H * W, nblk = smooth_blocks(N);
    float* stats = workspace;
    float* part = workspace + (size_t)B * 8;
    const float cx = 1.f / ((float)B * H * (W - 1)), cy = 1.f / ((float)B * (H - 1) * W);
    hipLaunchKernelGGL(smooth_stats_kernel, dim3(B), dim3(kStatThreads), 0, st, disp, H, W, stats);
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nblk, B), dim3(kThreads), 0, st, disp, img, H, W, stats, cx, cy, g, part);
    hipLaunchKernelGGL(smooth_finalize_kernel, dim3(1), dim3(kThreads), 0, st, part, B, nblk, cx, cy, out_scale, stats, loss);
    return io_check_launch("smooth_loss_fwd");
}

extern "C" int io_smooth_loss_bwd(const float* g, const float* workspace, const float* grad_out, float scale, int B, int H,
                                  int W, int accumulate, float* ddisp, hipStream_t st) {
    IO_REQUIRE(B > 0 && H > 1 && W > 1, IO_ERR_SHAPE, "smooth_loss_bwd: B=%d H=%d W=%d", B, H, W);
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3(smooth_blocks(H * W), B), dim3(kThreads), 0, st, g, workspace, grad_out, scale,
                       H, W, accumulate, ddisp);
    return io_check_launch("smooth_loss_bwd");
}

extern "C" size_t io_disp_order_workspace_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * disp_blocks(H * W) * 10;
}

extern "C" int io_disp_order_count(const float* disp1, const float* disp2, const float* modal1, const float* modal2,
                                   const long* depth_order1, const long* is_overlap, int B, int H, int W, int le_order,
                                   float out_scale, float* out, float* workspace, size_t workspace_floats, hipStream_t st) {
    IO_REQUIRE(B > 0 && H > 2 && W > 2 && (double)H * W < 2.0e9, IO_ERR_SHAPE, "disp_order_count: B=%d H=%d W=%d", B, H, W);
    IO_REQUIRE(workspace && workspace_floats >= io_disp_order_workspace_floats(B, H, W), IO_ERR_WORKSPACE,
               "disp_order_count: workspace too small");
    const int DB = disp_blocks(H * W);
    float* ext = workspace;
    int* cnt = reinterpret_cast<int*>(workspace + (size_t)B * DB * 6);
    hipLaunchKernelGGL(disp_order_ext_kernel, dim3(DB, B), dim3(kDispThreads), 0, st, disp1, disp2, modal1, modal2, H, W, ext);
    hipLaunchKernelGGL(disp_order_cnt_kernel, dim3(DB, B), dim3(kDispThreads), 0, st, disp1, disp2, modal1, modal2,
                       depth_order1, is_overlap, H, W, ext, cnt);
    hipLaunchKernelGGL(disp_order_finalize_kernel, dim3(1), dim3(kThreads), 0, st, cnt, depth_order1, B, DB, le_order,
                       out_scale / ((float)H * (float)W), out);
    return io_check_launch("disp_order_count");
}
