// Small HBM-bound / latency-bound kernels of the order-prediction path: input packing, max-pool,
// global-average-pool + FC heads, order losses, momentum SGD, filter transposition.
#include "io_common.h"

#ifndef IO_POOL_ROWS
#define IO_POOL_ROWS 1
#endif

namespace {

constexpr int kThreads = 256;

template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p) { return io_ldv(p); }
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v) { io_stv(p, v); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return io_bf2f(*p); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = io_f2bf(v); }

int ew_blocks(size_t n) {
    size_t b = (n + kThreads - 1) / kThreads;
    return (int)(b > 8192 ? 8192 : (b ? b : 1));
}

// ---- pack: up to 5 NCHW planes -> NHWC with 8 channels (5 real + 3 zero) --------------------
// The reference feeds torch.cat([modal_a, modal_b, rgb], 1) (supervised_order.py:537-538): channel
// order (mask_a, mask_b, R, G, B).  Planes are given per channel with a per-sample stride so the
// same kernel serves the concatenated [N,5,S,S] tensor and the separate rgb / mask tensors.
struct PackArgs {
    const float* plane[5];
    long stride[5];   // floats between consecutive samples of that plane
};
template <typename T>
__global__ __launch_bounds__(kThreads) void pack_planes_kernel(PackArgs a, int nplanes, int N, int HW,
                                                              T* __restrict__ out) {
    const size_t total = (size_t)N * HW;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t n = i / HW, p = i - n * HW;
        float v[8];
#pragma unroll
        for (int c = 0; c < 5; ++c) v[c] = c < nplanes ? a.plane[c][n * a.stride[c] + p] : 0.f;
        v[5] = v[6] = v[7] = 0.f;
        const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
        io_stv(out + i * 8, lo);
        io_stv(out + i * 8 + 4, hi);
    }
}

// ---- max-pool 3x3 stride 2 pad 1 (resnet_cls.py:144), NHWC -----------------------------------
// idx keeps, per output element, which of the 9 window taps won (first maximum in (kh,kw) scan
// order, as the PyTorch CPU kernel); one byte per element, 4 channels packed per uint32.
// XF: the pooled tensor is relu((x - xm[g][c]) * xs[g][c] + xh[g][c]) -- the BatchNorm + ReLU in front of the pooling
// (resnet_cls.py:205-208: bn1 -> relu -> maxpool) evaluated on the fly, so that activation is never stored; g = sample
// n / (N / G).  The arg-max is taken over the transformed values (scale may be negative).
template <typename T, bool XF = false>
__global__ __launch_bounds__(kThreads) void maxpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W,
                                                              int C, T* __restrict__ out,
                                                              uint32_t* __restrict__ idx,
                                                              const float* __restrict__ xs,
                                                              const float* __restrict__ xh, int npg,
                                                              const float* __restrict__ xm) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C >> 2;
    const size_t total = (size_t)N * Ho * Wo * C4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int q = (int)(i % C4);
        size_t t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if constexpr (XF) {
            const int gi = n / npg;
            if (xm) mu = *reinterpret_cast<const f32x4*>(xm + (size_t)gi * C + q * 4);
            sc = *reinterpret_cast<const f32x4*>(xs + (size_t)gi * C + q * 4);
            sh = *reinterpret_cast<const f32x4*>(xh + (size_t)gi * C + q * 4);
        }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        uint32_t bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * 2 - 1 + kh;
            if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * 2 - 1 + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                f32x4 v = ld4(x + (((size_t)n * H + h) * W + w) * C + q * 4);
                if constexpr (XF) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(__builtin_fmaf(v[k] - mu[k], sc[k], sh[k]), 0.f);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; bi[k] = kh * 3 + kw; }
            }
        }
        st4(out + i * 4, best);
        if (idx) idx[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool_bwd_kernel(const T* __restrict__ dy,
                                                              const uint32_t* __restrict__ idx, int N, int H, int W,
                                                              int C, T* __restrict__ dx) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C >> 2;
    const size_t total = (size_t)N * H * W * C4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int q = (int)(i % C4);
        size_t t = i / C4;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // windows (ho,wo) with 2*ho-1 <= h <= 2*ho+1
        for (int ho = h >> 1; ho <= (h + 1) >> 1; ++ho) {
            if (ho >= Ho) continue;
            const int kh = h - (2 * ho - 1);
            for (int wo = w >> 1; wo <= (w + 1) >> 1; ++wo) {
                if (wo >= Wo) continue;
                const int kw = w - (2 * wo - 1);
                const uint32_t k = kh * 3 + kw;
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C4 + q;
                const uint32_t id = idx[o];
                const f32x4 g = ld4(dy + o * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((id >> (8 * e)) & 0xffu) == k) acc[e] += g[e];
            }
        }
        st4(dx + i * 4, acc);
    }
}

// Row forms of the two pooling kernels: one block per output row (forward) / input row (backward), threads over
// (column, 16-byte channel chunk).  The element-indexed forms above decode (n, h, w, chunk) from a 64-bit flat index with
// three divisions by run-time values per element and move 8 bytes per lane on bf16 tensors; here the row is decoded once
// per block on the scalar unit, the column split is a shift when C / VEC is a power of two, and a lane always moves 16
// bytes (4 fp32 / 8 bf16 channels).
template <typename T> struct PoolChunk;
template <> struct PoolChunk<float> {
    static constexpr int NV = 1;
    static __device__ __forceinline__ void load(const float* p, f32x4* v) { v[0] = *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, const f32x4* v) { *reinterpret_cast<f32x4*>(p) = v[0]; }
};
template <> struct PoolChunk<bf16_t> {
    static constexpr int NV = 2;
    static __device__ __forceinline__ void load(const bf16_t* p, f32x4* v) {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k >> 1][(k & 1) * 2] = __builtin_bit_cast(float, w[k] << 16);
            v[k >> 1][(k & 1) * 2 + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const f32x4* v) {
        uint4 r;
        r.x = io_f2bf2(v[0][0], v[0][1]); r.y = io_f2bf2(v[0][2], v[0][3]);
        r.z = io_f2bf2(v[1][0], v[1][1]); r.w = io_f2bf2(v[1][2], v[1][3]);
        *reinterpret_cast<uint4*>(p) = r;
    }
};

template <typename T, bool XF>
__global__ __launch_bounds__(kThreads) void maxpool_fwd_rows_kernel(const T* __restrict__ x, int N, int H, int W, int C,
                                                                   T* __restrict__ out, uint32_t* __restrict__ idx,
                                                                   const float* __restrict__ xs,
                                                                   const float* __restrict__ xh, int npg,
                                                                   const float* __restrict__ xm, int cv_shift) {
    constexpr int NV = PoolChunk<T>::NV, VEC = 4 * NV;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, CV = C / VEC;
    const int ho = (int)blockIdx.x % Ho, n = (int)blockIdx.x / Ho;
    const int gi = XF ? n / npg : 0;
    const T* xn = x + (size_t)n * H * W * C;
    const size_t orow = (size_t)blockIdx.x * Wo;
    for (int j = threadIdx.x; j < Wo * CV; j += kThreads) {
        const int wo = cv_shift >= 0 ? j >> cv_shift : j / CV;
        const int q = j - wo * CV;
        f32x4 mu[NV], sc[NV], sh[NV], best[NV];
        uint32_t bi[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            mu[k] = 0.f; sc[k] = 1.f; sh[k] = 0.f; best[k] = -INFINITY; bi[k] = 0;
            if constexpr (XF) {
                const size_t to = (size_t)gi * C + q * VEC + 4 * k;
                if (xm) mu[k] = *reinterpret_cast<const f32x4*>(xm + to);
                sc[k] = *reinterpret_cast<const f32x4*>(xs + to);
                sh[k] = *reinterpret_cast<const f32x4*>(xh + to);
            }
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * 2 - 1 + kh;
            if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * 2 - 1 + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                f32x4 v[NV];
                PoolChunk<T>::load(xn + ((size_t)h * W + w) * C + q * VEC, v);
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = v[k][e];
                        if constexpr (XF) a = fmaxf(__builtin_fmaf(a - mu[k][e], sc[k][e], sh[k][e]), 0.f);
                        if (a > best[k][e] || a != a) {
                            best[k][e] = a;
                            bi[k] = (bi[k] & ~(0xffu << (8 * e))) | ((uint32_t)(kh * 3 + kw) << (8 * e));
                        }
                    }
            }
        }
        const size_t o = (orow + wo) * CV + q;
        PoolChunk<T>::store(out + o * VEC, best);
        if (idx) {
#pragma unroll
            for (int k = 0; k < NV; ++k) idx[o * NV + k] = bi[k];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool_bwd_rows_kernel(const T* __restrict__ dy,
                                                                   const uint32_t* __restrict__ idx, int N, int H, int W,
                                                                   int C, T* __restrict__ dx, int cv_shift) {
    constexpr int NV = PoolChunk<T>::NV, VEC = 4 * NV;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, CV = C / VEC;
    const int h = (int)blockIdx.x % H, n = (int)blockIdx.x / H;
    const size_t irow = (size_t)blockIdx.x * W;
    for (int j = threadIdx.x; j < W * CV; j += kThreads) {
        const int w = cv_shift >= 0 ? j >> cv_shift : j / CV;
        const int q = j - w * CV;
        f32x4 acc[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] = 0.f;
        // windows (ho, wo) with 2*ho-1 <= h <= 2*ho+1
        for (int ho = h >> 1; ho <= (h + 1) >> 1; ++ho) {
            if (ho >= Ho) continue;
            const int kh = h - (2 * ho - 1);
            for (int wo = w >> 1; wo <= (w + 1) >> 1; ++wo) {
                if (wo >= Wo) continue;
                const uint32_t kk = kh * 3 + (w - (2 * wo - 1));
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * CV + q;
                f32x4 g[NV];
                PoolChunk<T>::load(dy + o * VEC, g);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const uint32_t id = idx[o * NV + k];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (((id >> (8 * e)) & 0xffu) == kk) acc[k][e] += g[k][e];
                }
            }
        }
        PoolChunk<T>::store(dx + ((irow + w) * CV + q) * VEC, acc);
    }
}

// log2(v) when v is a power of two, else -1
int pow2_shift(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

// ---- global average pool + FC heads (resnet_cls.py:152-160, 214-222) ----------------------------
// one block per sample: pooled[n][c] = mean_p x[n][p][c]; logits[n][k] = pooled . W[k] + b[k]
template <typename T>
__global__ __launch_bounds__(kThreads) void avgpool_fc_kernel(const T* __restrict__ x, int HW, int C,
                                                             const float* __restrict__ w0,
                                                             const float* __restrict__ b0, int K0,
                                                             const float* __restrict__ w1,
                                                             const float* __restrict__ b1, int K1,
                                                             float* __restrict__ pooled,
                                                             float* __restrict__ logits) {
    extern __shared__ float sp[];   // C floats + 8*4 reduction slots
    const int n = blockIdx.x, C4 = C >> 2;
    const float inv = 1.f / (float)HW;
    const T* xn = x + (size_t)n * HW * C;
    for (int q = threadIdx.x; q < C4; q += blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < HW; ++p) s += ld4(xn + (size_t)p * C + q * 4);
        s *= inv;
        st4(sp + q * 4, s);
        st4(pooled + (size_t)n * C + q * 4, s);
    }
    __syncthreads();
    const int K = K0 + K1;
    float* red = sp + C;
    for (int k = 0; k < K; ++k) {
        const float* wk = k < K0 ? w0 + (size_t)k * C : w1 + (size_t)(k - K0) * C;
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += blockDim.x) s += sp[c] * wk[c];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
            logits[(size_t)n * K + k] = t + (k < K0 ? b0[k] : b1[k - K0]);
        }
        __syncthreads();
    }
}

// d_x[n][p][c] = (sum_k dlogits[n][k] W[k][c]) / HW
template <typename T>
__global__ __launch_bounds__(kThreads) void avgpool_fc_bwd_data_kernel(const float* __restrict__ dlogits,
                                                                      const float* __restrict__ w0, int K0,
                                                                      const float* __restrict__ w1, int K1, int HW,
                                                                      int C, const T* __restrict__ mask,
                                                                      T* __restrict__ dx, size_t total4) {
    // one thread = 4 consecutive channels of one pixel: 16-byte mask load and store, the K <= 5 head rows from L1/L2
    // (the first version walked the HW pixels of a channel serially, one 4-byte load + store at a time: 292 us for the
    // 268 MB of the bench batch, a latency chain)
    const int K = K0 + K1, C4 = C >> 2;
    const float inv = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % C4);
        const size_t pix = i / C4;
        const int n = (int)(pix / HW);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; ++k) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>((k < K0 ? w0 + (size_t)k * C : w1 + (size_t)(k - K0) * C) + q * 4);
            const float d = dlogits[(size_t)n * K + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += d * wv[e];
        }
        f32x4 m = {1.f, 1.f, 1.f, 1.f};
        if (mask) m = ld4(mask + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = m[e] > 0.f ? s[e] * inv : 0.f;
        st4(dx + i * 4, s);
    }
}

// dW[k][c] = sum_n dlogits[n][k] pooled[n][c];  db[k] = sum_n dlogits[n][k].  Block = 32 channels x 8 slices of n
// (fixed summation order: deterministic); the first version had ONE thread walk all N samples of a channel.
__global__ __launch_bounds__(kThreads) void fc_bwd_weight_kernel(const float* __restrict__ dlogits,
                                                                const float* __restrict__ pooled, int N, int C,
                                                                int K, int kofs, int Kh, float* __restrict__ dw,
                                                                float* __restrict__ db) {
    __shared__ float red[8][33];
    const int k = blockIdx.y;   // row of this head
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f, sb = 0.f;
    if (c < C)
        for (int n = sl; n < N; n += 8) {
            const float d = dlogits[(size_t)n * K + kofs + k];
            s += d * pooled[(size_t)n * C + c];
            sb += d;
        }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j][cl];
        dw[(size_t)k * C + c] = t;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        if (cl == 0) red[sl][0] = sb;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += red[j][0];
            db[k] = t;
        }
    }
    (void)Kh;
}

// ---- order losses (supervised_order.py:59-95, 413-438, 481-493, 535-548) ---------------------------
// Rows are G direction-groups of B samples.  Occlusion head: sigmoid -> BCELoss (mean over B*2 per
// direction, log clamped at -100).  Depth head: softmax -> CrossEntropyLoss *on the probabilities*
// (log-softmax of a softmax), optionally split into overlap / distinct subsets, each a mean over its
// own size, weighted.  losses[0] = total * inv_world, losses[1] = occlusion, losses[2] = depth.
// dlogits = d(total * inv_world)/dlogits.
__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(kThreads) void order_loss_kernel(const float* __restrict__ logits, int N, int B, int K,
                                                             int Kocc, int Kdep,
                                                             const float* __restrict__ occ_t,
                                                             const long* __restrict__ dep_t,
                                                             const long* __restrict__ is_overlap, float w_ov,
                                                             float w_di, float inv_world,
                                                             float* __restrict__ losses,
                                                             float* __restrict__ dlogits) {
    __shared__ float red[8];
    // subset sizes as the reference's boolean masks give them (supervised_order.py:62-73): is_overlap == 1 / == 0; a row
    // with any other value belongs to neither subset and contributes nothing
    float n_ov = 0.f, n_di = 0.f;
    if (Kdep && is_overlap) {
        float c = 0.f, d = 0.f;
        for (int n = threadIdx.x; n < B; n += blockDim.x) {
            c += is_overlap[n] == 1 ? 1.f : 0.f;
            d += is_overlap[n] == 0 ? 1.f : 0.f;
        }
        n_ov = block_sum(c, red);
        __syncthreads();
        n_di = block_sum(d, red);
    }
    float l_occ = 0.f, l_dep = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* z = logits + (size_t)n * K;
        float* dz = dlogits ? dlogits + (size_t)n * K : nullptr;
        if (Kocc) {
            const float wrow = 1.f / (float)(B * Kocc);
            for (int k = 0; k < Kocc; ++k) {
                const float p = 1.f / (1.f + expf(-z[k]));
                const float y = occ_t[(size_t)n * Kocc + k];
                const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
                l_occ += -(y * lp + (1.f - y) * lq) * wrow;
                if (dz) {
                    // BCELoss backward then sigmoid backward, as autograd composes them
                    const float gp = (p - y) / fmaxf((1.f - p) * p, 1e-12f) * wrow * inv_world;
                    dz[k] = gp * p * (1.f - p);
                }
            }
        }
        if (Kdep) {
            const float* zd = z + Kocc;
            float wrow;
            if (is_overlap) {
                const long io = is_overlap[n % B];
                const bool ov = io == 1;
                const float cnt = ov ? n_ov : n_di;
                wrow = ((io == 0 || io == 1) && cnt > 0.f) ? (ov ? w_ov : w_di) / cnt : 0.f;
            } else {
                wrow = 1.f / (float)B;
            }
            float q[4], s[4];
            float mx = zd[0];
            for (int k = 1; k < Kdep; ++k) mx = fmaxf(mx, zd[k]);
            float sum = 0.f;
            for (int k = 0; k < Kdep; ++k) { q[k] = expf(zd[k] - mx); sum += q[k]; }
            for (int k = 0; k < Kdep; ++k) q[k] /= sum;
            float mq = q[0];
            for (int k = 1; k < Kdep; ++k) mq = fmaxf(mq, q[k]);
            float s2 = 0.f;
            for (int k = 0; k < Kdep; ++k) { s[k] = expf(q[k] - mq); s2 += s[k]; }
            // a class id outside [0, Kdep) is an error (nn.CrossEntropyLoss device-asserts on it): poison the loss so
            // that it cannot pass silently -- never index past the local arrays
            const long tl = dep_t[n];
            const bool tok = tl >= 0 && tl < Kdep;
            const int t = tok ? (int)tl : 0;
            // (a row of weight 0 -- is_overlap outside {0, 1}: in neither subset, datasets/reader.py:363-380 with
            // remove_depth_overlap -- takes no part in the loss, so its label is not looked at either)
            l_dep += tok ? -(q[t] - mq - logf(s2)) * wrow : (wrow != 0.f ? __builtin_nanf("") : 0.f);
            if (dz) {
                float dq[4], dot = 0.f;
                for (int k = 0; k < Kdep; ++k) {
                    dq[k] = (s[k] / s2 - (k == t ? 1.f : 0.f)) * wrow * inv_world;
                    dot += q[k] * dq[k];
                }
                for (int k = 0; k < Kdep; ++k) dz[Kocc + k] = q[k] * (dq[k] - dot);
            }
        }
    }
    const float so = block_sum(l_occ, red);
    const float sd = block_sum(l_dep, red);
    if (threadIdx.x == 0) {
        losses[0] = (so + sd) * inv_world;
        losses[1] = so;
        losses[2] = sd;
    }
}

// ---- momentum SGD over the flat parameter buffer (single_stage_model.py:35-38) -------------------
// d = g + wd*p ; buf = momentum*buf + d ; p -= lr*buf      (buf starts at 0 == "first step buf=d")
__global__ __launch_bounds__(kThreads) void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                               float* __restrict__ buf, size_t n4, float lr,
                                                               float momentum, float wd) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 pv = ld4(p + i * 4);
        const f32x4 d = ld4(g + i * 4) + wd * pv;
        const f32x4 b = momentum * ld4(buf + i * 4) + d;
        st4(buf + i * 4, b);
        st4(p + i * 4, pv - lr * b);
    }
}

// ---- filter transpose  W[O][T][C] -> Wt[C][T][O]  (operand of the data-gradient GEMM); the source is the
// fp32 master filter, the destination has the GEMM operand type (bf16 in bf16 mode) --------------------
template <typename T>
__global__ void filter_transpose_kernel(const float* __restrict__ w, int O, int T_, int C, T* __restrict__ wt) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int o0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, c = c0 + tx;
        tile[r][tx] = (o < O && c < C) ? w[((size_t)o * T_ + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, o = o0 + tx;
        if (c < C && o < O) st1(wt + ((size_t)c * T_ + t) * O + o, tile[tx][r]);
    }
}

// the same for EVERY filter of a network in one launch: 52 launches of a few microseconds each were 0.4 ms (fp32) / 0.55 ms
// (bf16) of a step.  The table rides in the kernel arguments; a block finds its layer by a scalar scan over <= 64 starts.
template <typename T>
__global__ __launch_bounds__(256) void filter_transpose_all_kernel(IoFilterTable tab, const float* __restrict__ params,
                                                                   T* __restrict__ wt_all) {
    __shared__ float tile[32][33];
    int l = 0;
    while (l + 1 < tab.n && (int)blockIdx.x >= tab.start[l + 1]) ++l;
    const int O = tab.O[l], T_ = tab.T[l], C = tab.C[l];
    const int ncx = (C + 31) >> 5, noy = (O + 31) >> 5;
    int id = (int)blockIdx.x - tab.start[l];
    const int t = id / (ncx * noy);
    id -= t * ncx * noy;
    const int o0 = (id / ncx) * 32, c0 = (id % ncx) * 32;
    const float* w = params + tab.off[l];
    T* wt = wt_all + tab.off[l];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, c = c0 + tx;
        tile[r][tx] = (o < O && c < C) ? w[((size_t)o * T_ + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, o = o0 + tx;
        if (c < C && o < O) st1(wt + ((size_t)c * T_ + t) * O + o, tile[tx][r]);
    }
}

// fp32 -> bf16 copy (operand copies of the fp32 master filters)
__global__ __launch_bounds__(kThreads) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                            size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        st4(dst + i * 4, ld4(src + i * 4));
}

}  // namespace

int io_pack_planes_t(const float* const* planes, const long* sample_strides, int nplanes, int N, int H, int W,
                     void* out, hipStream_t st, int dt) {
    IO_REQUIRE(nplanes >= 1 && nplanes <= 5, IO_ERR_SHAPE, "pack: nplanes=%d (1..5)", nplanes);
    PackArgs a;
    for (int c = 0; c < 5; ++c) {
        a.plane[c] = c < nplanes ? planes[c] : nullptr;
        a.stride[c] = c < nplanes ? sample_strides[c] : 0;
    }
    const size_t total = (size_t)N * H * W;
    IoProfScope prof(IO_PROF_PACK, 0.0, (double)total * (4.0 * nplanes + 8.0 * io_dtype_bytes(dt)), st);
    if (dt == IO_BF16)
        hipLaunchKernelGGL(pack_planes_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(kThreads), 0, st, a, nplanes, N,
                           H * W, (bf16_t*)out);
    else
        hipLaunchKernelGGL(pack_planes_kernel<float>, dim3(ew_blocks(total)), dim3(kThreads), 0, st, a, nplanes, N, H * W,
                           (float*)out);
    return io_check_launch("pack_planes");
}

extern "C" int io_pack_planes_nhwc8(const float* const* planes, const long* sample_strides, int nplanes, int N,
                                    int H, int W, float* out, hipStream_t st) {
    return io_pack_planes_t(planes, sample_strides, nplanes, N, H, W, out, st, IO_F32);
}

int io_maxpool_fwd_t(const void* x, int N, int H, int W, int C, void* out, uint32_t* idx, hipStream_t st, int dt,
                     const float* xs, const float* xh, int G, const float* xm) {
    IO_REQUIRE(C % 4 == 0, IO_ERR_SHAPE, "maxpool: C=%d", C);
    IO_REQUIRE(!xs || (xh && G >= 1 && N % G == 0), IO_ERR_SHAPE, "maxpool: input transform needs both tables and G | N");
    const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    IoProfScope prof(IO_PROF_POOL_HEAD, 0.0, (double)io_dtype_bytes(dt) * N * H * W * C * 1.3125, st);
    const int npg = xs ? N / G : 1;
#define IO_MP(T_, XF_)                                                                                              \
    hipLaunchKernelGGL((maxpool_fwd_kernel<T_, XF_>), dim3(ew_blocks(total)), dim3(kThreads), 0, st, (const T_*)x, N, H, \
                       W, C, (T_*)out, idx, xs, xh, npg, xm)
    const int vec = dt == IO_BF16 ? 8 : 4;
    const long rows = (long)N * ((H + 1) / 2);
#define IO_MPR(T_, XF_)                                                                                             \
    hipLaunchKernelGGL((maxpool_fwd_rows_kernel<T_, XF_>), dim3((unsigned)rows), dim3(kThreads), 0, st, (const T_*)x, N, H, \
                       W, C, (T_*)out, idx, xs, xh, npg, xm, pow2_shift(C / vec))
    if (IO_POOL_ROWS && C % vec == 0 && rows < (1L << 31)) {
        if (dt == IO_BF16) {
            if (xs) IO_MPR(bf16_t, true);
            else IO_MPR(bf16_t, false);
        } else {
            if (xs) IO_MPR(float, true);
            else IO_MPR(float, false);
        }
    } else if (dt == IO_BF16) {
        if (xs) IO_MP(bf16_t, true);
        else IO_MP(bf16_t, false);
    } else {
        if (xs) IO_MP(float, true);
        else IO_MP(float, false);
    }
#undef IO_MPR
#undef IO_MP
    return io_check_launch("maxpool_fwd");
}

extern "C" int io_maxpool_fwd_xf_dt(const void* x, int N, int H, int W, int C, void* out, uint32_t* idx, int G,
                                    const float* in_mean, const float* in_scale, const float* in_shift, int dtype,
                                    hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "maxpool_fwd_xf: unknown dtype %d", dtype);
    IO_REQUIRE(in_scale && in_shift, IO_ERR_SHAPE, "maxpool_fwd_xf: in_scale / in_shift are required");
    return io_maxpool_fwd_t(x, N, H, W, C, out, idx, st, dtype, in_scale, in_shift, G, in_mean);
}

extern "C" int io_maxpool_fwd(const float* x, int N, int H, int W, int C, float* out, uint32_t* idx,
                              hipStream_t st) {
    return io_maxpool_fwd_t(x, N, H, W, C, out, idx, st, IO_F32);
}

int io_maxpool_bwd_t(const void* dy, const uint32_t* idx, int N, int H, int W, int C, void* dx, hipStream_t st,
                     int dt) {
    IO_REQUIRE(C % 4 == 0, IO_ERR_SHAPE, "maxpool: C=%d", C);
    const size_t total = (size_t)N * H * W * (C / 4);
    IoProfScope prof(IO_PROF_POOL_HEAD, 0.0, (double)io_dtype_bytes(dt) * N * H * W * C * 1.3125, st);
    const int vec = dt == IO_BF16 ? 8 : 4;
    const long rows = (long)N * H;
    if (IO_POOL_ROWS && C % vec == 0 && rows < (1L << 31)) {
        if (dt == IO_BF16)
            hipLaunchKernelGGL(maxpool_bwd_rows_kernel<bf16_t>, dim3((unsigned)rows), dim3(kThreads), 0, st,
                               (const bf16_t*)dy, idx, N, H, W, C, (bf16_t*)dx, pow2_shift(C / vec));
        else
            hipLaunchKernelGGL(maxpool_bwd_rows_kernel<float>, dim3((unsigned)rows), dim3(kThreads), 0, st, (const float*)dy,
                               idx, N, H, W, C, (float*)dx, pow2_shift(C / vec));
    } else if (dt == IO_BF16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(kThreads), 0, st, (const bf16_t*)dy,
                           idx, N, H, W, C, (bf16_t*)dx);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(kThreads), 0, st, (const float*)dy,
                           idx, N, H, W, C, (float*)dx);
    return io_check_launch("maxpool_bwd");
}

extern "C" int io_maxpool_bwd(const float* dy, const uint32_t* idx, int N, int H, int W, int C, float* dx,
                              hipStream_t st) {
    return io_maxpool_bwd_t(dy, idx, N, H, W, C, dx, st, IO_F32);
}

int io_avgpool_fc_fwd_t(const void* x, int N, int HW, int C, const float* w0, const float* b0, int K0,
                        const float* w1, const float* b1, int K1, float* pooled, float* logits, hipStream_t st,
                        int dt) {
    IO_REQUIRE(C % 4 == 0 && K0 >= 1 && K1 >= 0, IO_ERR_SHAPE, "avgpool_fc: C=%d K0=%d K1=%d", C, K0, K1);
    const size_t lds = (size_t)(C + 32) * sizeof(float);
    IoProfScope prof(IO_PROF_POOL_HEAD, 2.0 * N * C * (K0 + K1), (double)io_dtype_bytes(dt) * N * HW * C, st);
    if (dt == IO_BF16)
        hipLaunchKernelGGL(avgpool_fc_kernel<bf16_t>, dim3(N), dim3(kThreads), lds, st, (const bf16_t*)x, HW, C, w0, b0,
                           K0, w1, b1, K1, pooled, logits);
    else
        hipLaunchKernelGGL(avgpool_fc_kernel<float>, dim3(N), dim3(kThreads), lds, st, (const float*)x, HW, C, w0, b0,
                           K0, w1, b1, K1, pooled, logits);
    return io_check_launch("avgpool_fc_fwd");
}

extern "C" int io_avgpool_fc_fwd(const float* x, int N, int HW, int C, const float* w0, const float* b0, int K0,
                                 const float* w1, const float* b1, int K1, float* pooled, float* logits,
                                 hipStream_t st) {
    return io_avgpool_fc_fwd_t(x, N, HW, C, w0, b0, K0, w1, b1, K1, pooled, logits, st, IO_F32);
}

int io_avgpool_fc_bwd_t(const float* dlogits, const float* pooled, int N, int HW, int C, const float* w0, int K0,
                        const float* w1, int K1, const void* relu_mask, void* dx, float* dw0, float* db0, float* dw1,
                        float* db1, hipStream_t st, int dt) {
    const int K = K0 + K1;
    IO_REQUIRE(C % 4 == 0, IO_ERR_SHAPE, "avgpool_fc_bwd: C=%d must be a multiple of 4", C);
    const size_t total4 = (size_t)N * HW * (C / 4);
    IoProfScope prof(IO_PROF_POOL_HEAD, 4.0 * N * C * K, (relu_mask ? 2.0 : 1.0) * io_dtype_bytes(dt) * N * HW * C, st);
    if (dt == IO_BF16)
        hipLaunchKernelGGL(avgpool_fc_bwd_data_kernel<bf16_t>, dim3(ew_blocks(total4)), dim3(kThreads), 0, st, dlogits, w0,
                           K0, w1, K1, HW, C, (const bf16_t*)relu_mask, (bf16_t*)dx, total4);
    else
        hipLaunchKernelGGL(avgpool_fc_bwd_data_kernel<float>, dim3(ew_blocks(total4)), dim3(kThreads), 0, st, dlogits, w0,
                           K0, w1, K1, HW, C, (const float*)relu_mask, (float*)dx, total4);
    hipLaunchKernelGGL(fc_bwd_weight_kernel, dim3(io_cdiv(C, 32), K0), dim3(kThreads), 0, st, dlogits, pooled,
                       N, C, K, 0, K0, dw0, db0);
    if (K1 > 0)
        hipLaunchKernelGGL(fc_bwd_weight_kernel, dim3(io_cdiv(C, 32), K1), dim3(kThreads), 0, st, dlogits,
                           pooled, N, C, K, K0, K1, dw1, db1);
    return io_check_launch("avgpool_fc_bwd");
}

extern "C" int io_avgpool_fc_bwd(const float* dlogits, const float* pooled, int N, int HW, int C, const float* w0,
                                 int K0, const float* w1, int K1, const float* relu_mask, float* dx, float* dw0,
                                 float* db0, float* dw1, float* db1, hipStream_t st) {
    return io_avgpool_fc_bwd_t(dlogits, pooled, N, HW, C, w0, K0, w1, K1, relu_mask, dx, dw0, db0, dw1, db1, st, IO_F32);
}

extern "C" int io_order_loss(const float* logits, int N, int B, int Kocc, int Kdep, const float* occ_target,
                             const long* depth_target, const long* is_overlap, float overlap_weight,
                             float distinct_weight, float inv_world, float* losses, float* dlogits,
                             hipStream_t st) {
    IO_REQUIRE(B > 0 && N % B == 0, IO_ERR_SHAPE, "order_loss: N=%d B=%d", N, B);
    IO_REQUIRE((Kocc == 0 || Kocc == 2) && Kdep >= 0 && Kdep <= 4 && Kocc + Kdep > 0, IO_ERR_SHAPE,
               "order_loss: Kocc=%d Kdep=%d", Kocc, Kdep);
    IoProfScope prof(IO_PROF_LOSS, 0.0, 8.0 * N * (Kocc + Kdep), st);
    hipLaunchKernelGGL(order_loss_kernel, dim3(1), dim3(kThreads), 0, st, logits, N, B, Kocc + Kdep, Kocc, Kdep,
                       occ_target, depth_target, is_overlap, overlap_weight, distinct_weight, inv_world, losses,
                       dlogits);
    return io_check_launch("order_loss");
}

extern "C" int io_sgd_momentum(float* params, const float* grads, float* momentum_buf, size_t n, float lr,
                               float momentum, float weight_decay, hipStream_t st) {
    IO_REQUIRE(n % 4 == 0, IO_ERR_SHAPE, "sgd: n=%zu must be a multiple of 4", n);
    const size_t n4 = n / 4;
    IoProfScope prof(IO_PROF_SGD, 0.0, 20.0 * n, st);
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3(ew_blocks(n4)), dim3(kThreads), 0, st, params, grads, momentum_buf,
                       n4, lr, momentum, weight_decay);
    return io_check_launch("sgd_momentum");
}

int io_filter_prepare_t(const float* w, int O, int T, int C, void* dst, int transpose, hipStream_t st, int dt) {
    IoProfScope prof(IO_PROF_TRANSPOSE, 0.0, (4.0 + io_dtype_bytes(dt)) * O * T * C, st);
    if (transpose) {
        dim3 grid(io_cdiv(C, 32), io_cdiv(O, 32), T);
        if (dt == IO_BF16)
            hipLaunchKernelGGL(filter_transpose_kernel<bf16_t>, grid, dim3(256), 0, st, w, O, T, C, (bf16_t*)dst);
        else
            hipLaunchKernelGGL(filter_transpose_kernel<float>, grid, dim3(256), 0, st, w, O, T, C, (float*)dst);
    } else {
        IO_REQUIRE(dt == IO_BF16 && ((size_t)O * T * C) % 4 == 0, IO_ERR_SHAPE, "filter_prepare: nothing to do");
        const size_t n4 = (size_t)O * T * C / 4;
        hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_blocks(n4)), dim3(kThreads), 0, st, w, (bf16_t*)dst, n4);
    }
    return io_check_launch("filter_prepare");
}

int io_filter_table_add(IoFilterTable& tab, long off, int O, int T, int C) {
    IO_REQUIRE(tab.n < IoFilterTable::kMax && off >= 0 && off < (1L << 32) && off % 8 == 0 && O > 0 && T > 0 && C > 0,
               IO_ERR_SHAPE, "filter table: entry %d (offset %ld, %d x %d x %d)", tab.n, off, O, T, C);
    const int l = tab.n++;
    tab.off[l] = (unsigned)off;
    tab.O[l] = O; tab.T[l] = T; tab.C[l] = C;
    tab.start[l + 1] = tab.start[l] + io_cdiv(C, 32) * io_cdiv(O, 32) * T;
    return IO_OK;
}

// Wt of every table entry at the entry's own element offset inside wt_all (a buffer shaped like the parameter buffer, in
// the operand type)
int io_filter_transpose_all(const IoFilterTable& tab, const float* params, void* wt_all, hipStream_t st, int dt) {
    if (tab.n == 0) return IO_OK;
    double elems = 0.0;
    for (int l = 0; l < tab.n; ++l) elems += (double)tab.O[l] * tab.T[l] * tab.C[l];
    IoProfScope prof(IO_PROF_TRANSPOSE, 0.0, (4.0 + io_dtype_bytes(dt)) * elems, st);
    const dim3 grid((unsigned)tab.start[tab.n]);
    if (dt == IO_BF16)
        hipLaunchKernelGGL(filter_transpose_all_kernel<bf16_t>, grid, dim3(256), 0, st, tab, params, (bf16_t*)wt_all);
    else
        hipLaunchKernelGGL(filter_transpose_all_kernel<float>, grid, dim3(256), 0, st, tab, params, (float*)wt_all);
    return io_check_launch("filter_transpose_all");
}

// Exact-K stem (conv_igemm.hip): filter [O][T][8] <-> packed rows [O][kp], k = tap * cr + channel over the cr real
// channels (the columns past T * cr are zero).
namespace {
__global__ __launch_bounds__(256) void stem_pack_filter_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                              int O, int T, int cr, int kp) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= O * kp) return;
    const int o = idx / kp, k = idx - o * kp;
    const int tp = k / cr, ch = k - tp * cr;
    wp[idx] = tp < T ? w[((size_t)o * T + tp) * 8 + ch] : 0.f;
}
__global__ __launch_bounds__(256) void stem_unpack_grad_kernel(const float* __restrict__ dwp, float* __restrict__ dw,
                                                              int O, int T, int cr, int kp) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= O * T * 8) return;
    const int ch = idx & 7, ot = idx >> 3;
    const int o = ot / T, tp = ot - o * T;
    dw[idx] = ch < cr ? dwp[(size_t)o * kp + tp * cr + ch] : 0.f;
}
}  // namespace

int io_stem_pack_filter(const float* w, float* wp, int O, int T, int cr, hipStream_t st) {
    const int kp = io_stem_kp(T, cr);
    hipLaunchKernelGGL(stem_pack_filter_kernel, dim3(io_cdiv((long)O * kp, 256)), dim3(256), 0, st, w, wp, O, T, cr, kp);
    return io_check_launch("stem_pack_filter");
}

int io_stem_unpack_grad(const float* dwp, float* dw, int O, int T, int cr, hipStream_t st) {
    const int kp = io_stem_kp(T, cr);
    hipLaunchKernelGGL(stem_unpack_grad_kernel, dim3(io_cdiv((long)O * T * 8, 256)), dim3(256), 0, st, dwp, dw, O, T,
                       cr, kp);
    return io_check_launch("stem_unpack_grad");
}

extern "C" int io_filter_transpose(const float* w, int O, int T, int C, float* wt, hipStream_t st) {
    return io_filter_prepare_t(w, O, T, C, wt, 1, st, IO_F32);
}
