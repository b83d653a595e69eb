// BatchNorm2d (train + eval) over NHWC activations (fp32 or bf16 storage; statistics, tables and parameter gradients
// are always fp32), with "groups": the N samples are split
// into G equal consecutive groups that are normalised with SEPARATE batch statistics.  G = 2 is how
// one launch serves the reference's two directional passes (supervised_order.py:537-538), which are
// separate module calls and therefore separate BN batches; running statistics are advanced group by
// group so they end up exactly as after two sequential calls.
//
// All kernels are HBM-bound streaming passes: each thread owns one 16-byte chunk of channels (4 fp32 / 8 bf16) and
// strides over rows, so a wave reads whole contiguous NHWC rows (1 KiB per wave-instruction for C >= 256).
// Reductions are two-level (per-block fp32 partials, fp64 finalize) and deterministic.
// Reference semantics: nn.BatchNorm2d, resnet_cls.py:142, 87-92, 189 (eps 1e-5, momentum 0.1,
// biased variance for normalisation, unbiased for the running estimate).
#include "io_common.h"

namespace {

constexpr int kThreads = 256;

// 4 consecutive elements <-> float4, for fp32 and bf16 tensors alike (tables / partials are always fp32)
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p) { return io_ldv(p); }
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v) { io_stv(p, v); }

// BN affine output; the SAME expression is used by the forward apply and by the backward kernels that
// recompute the ReLU mask from y, so the mask bit is reproduced exactly
__device__ __forceinline__ f32x4 bn_affine(f32x4 y, f32x4 mean, f32x4 scale, f32x4 shift) {
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __builtin_fmaf(y[k] - mean[k], scale[k], shift[k]);
    return r;
}

// One 16-byte chunk of a storage-typed tensor = VEC consecutive channels (4 fp32 / 8 bf16), held as NV float4.
// The streaming (apply) kernels move whole chunks so that a bf16 lane issues the same 16-byte accesses as fp32.
template <typename T> struct Chunk {
    static constexpr int VEC = 16 / (int)sizeof(T), NV = VEC / 4;
    f32x4 v[NV];
};
__device__ __forceinline__ Chunk<float> ldc(const float* p) {
    Chunk<float> c;
    c.v[0] = *reinterpret_cast<const f32x4*>(p);
    return c;
}
__device__ __forceinline__ Chunk<bf16_t> ldc(const bf16_t* p) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    Chunk<bf16_t> c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c.v[k >> 1][(k & 1) * 2] = __builtin_bit_cast(float, w[k] << 16);
        c.v[k >> 1][(k & 1) * 2 + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
    }
    return c;
}
__device__ __forceinline__ void stc(float* p, const Chunk<float>& c) { *reinterpret_cast<f32x4*>(p) = c.v[0]; }
__device__ __forceinline__ void stc(bf16_t* p, const Chunk<bf16_t>& c) {
    uint4 r;
    r.x = io_f2bf2(c.v[0][0], c.v[0][1]);
    r.y = io_f2bf2(c.v[0][2], c.v[0][3]);
    r.z = io_f2bf2(c.v[1][0], c.v[1][1]);
    r.w = io_f2bf2(c.v[1][2], c.v[1][3]);
    *reinterpret_cast<uint4*>(p) = r;
}
// per-channel table entries for the channels of one chunk
template <int NV> struct Tab {
    f32x4 v[NV];
};
template <int NV> __device__ __forceinline__ Tab<NV> ldt(const float* p) {
    Tab<NV> t;
#pragma unroll
    for (int k = 0; k < NV; ++k) t.v[k] = *reinterpret_cast<const f32x4*>(p + 4 * k);
    return t;
}

struct ColMap {
    int TX, TY, tx, ty, nq;   // column threads, row lanes, my coords, quads per thread
};
__device__ __forceinline__ ColMap col_map(int C4) {
    ColMap m;
    m.TX = C4 < kThreads ? C4 : kThreads;
    m.TY = kThreads / m.TX;
    m.tx = threadIdx.x % m.TX;
    m.ty = threadIdx.x / m.TX;
    m.nq = (C4 + m.TX - 1) / m.TX;
    return m;
}

// block-level reduction over the TY row lanes of two float4 accumulators per owned quad; result is
// written by ty == 0 to dst0/dst1[(quad)*4 ..]
template <int MAXQ>
__device__ __forceinline__ void reduce_rows_store(const ColMap& cm, f32x4 (&s0)[MAXQ], f32x4 (&s1)[MAXQ],
                                                  float* dst0, float* dst1, int C4, int qs = 4) {
    __shared__ f32x4 red[2][kThreads];
    for (int i = 0; i < MAXQ; ++i) {
        if (i >= cm.nq) break;
        red[0][threadIdx.x] = s0[i];
        red[1][threadIdx.x] = s1[i];
        __syncthreads();
        for (int off = cm.TY >> 1; off > 0; off >>= 1) {
            if (cm.ty < off) {
                red[0][threadIdx.x] += red[0][threadIdx.x + off * cm.TX];
                red[1][threadIdx.x] += red[1][threadIdx.x + off * cm.TX];
            }
            __syncthreads();
        }
        const int q = cm.tx + cm.TX * i;
        if (cm.ty == 0 && q < C4) {
            st4(dst0 + q * qs, red[0][threadIdx.x]);
            st4(dst1 + q * qs, red[1][threadIdx.x]);
        }
        __syncthreads();
    }
}

// ---- forward statistics --------------------------------------------------------------------
// grid (nb, G): block b of group g reduces rows [g*Mg + b*rpb, ...).  Sums are taken of d = x - pivot
// with pivot = the block's first row (per channel): conv outputs whose |mean| >> std would otherwise
// lose the variance in E[x^2] - E[x]^2.  Partials: sum d, sum d^2, pivot, per (g, b, c); merged in
// fp64 with the pairwise (Chan) update -- the same robustness as the reference's two-pass CPU kernel.
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const T* __restrict__ x, int Mg, int C, int rpb,
                                                           float* __restrict__ psum, float* __restrict__ psq,
                                                           float* __restrict__ ppiv) {
    const int C4 = C >> 2;
    const ColMap cm = col_map(C4);
    const int g = blockIdx.y, b = blockIdx.x, nb = gridDim.x;
    const int r0 = b * rpb, r1 = min(r0 + rpb, Mg);
    const T* xg = x + (size_t)g * Mg * C;
    f32x4 s[2], ss[2], pv[2];
    for (int i = 0; i < 2; ++i) {
        s[i] = 0.f; ss[i] = 0.f; pv[i] = 0.f;
        const int q = cm.tx + cm.TX * i;
        if (i < cm.nq && q < C4) pv[i] = ld4(xg + (size_t)r0 * C + q * 4);
    }
    constexpr int U = 4;   // rows in flight per thread (memory-level parallelism)
    for (int r = r0 + cm.ty; r < r1; r += U * cm.TY) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = cm.tx + cm.TX * i;
            if (i < cm.nq && q < C4) {
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int rr = r + u * cm.TY;
                    v[u] = rr < r1 ? ld4(xg + (size_t)rr * C + q * 4) : pv[i];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const f32x4 d = v[u] - pv[i];
                    s[i] += d;
                    ss[i] += d * d;
                }
            }
        }
    }
    const size_t o = ((size_t)g * nb + b) * C;
    if (cm.ty == 0)
        for (int i = 0; i < 2; ++i) {
            const int q = cm.tx + cm.TX * i;
            if (i < cm.nq && q < C4) st4(ppiv + o + q * 4, pv[i]);
        }
    reduce_rows_store<2>(cm, s, ss, psum + o, psq + o, C4);
}

// block = 32 channels x 8 lanes over the per-block partials (coalesced 128-B reads, short serial chains);
// groups are processed in order so the running statistics see group 0, then group 1, ...
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ psum,
                                                         const float* __restrict__ psq,
                                                         const float* __restrict__ ppiv, int nb, int rpb, int G,
                                                         int Mg, int C, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         float* __restrict__ run_mean, float* __restrict__ run_var,
                                                         float momentum, float eps, float* __restrict__ mean,
                                                         float* __restrict__ rstd, float* __restrict__ scale,
                                                         float* __restrict__ shift) {
    __shared__ double sh[3][8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const bool ok = c < C;
    float rm = 0.f, rv = 0.f;
    if (ok && ty == 0) {
        rm = run_mean ? run_mean[c] : 0.f;
        rv = run_var ? run_var[c] : 0.f;
    }
    for (int g = 0; g < G; ++g) {
        double n = 0.0, mu = 0.0, m2 = 0.0;
        if (ok)
            for (int b0 = ty; b0 < nb; b0 += 64) {
                float vs[8], vq[8], vp[8];     // 24 independent loads in flight, then the serial merges
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = b0 + 8 * u;
                    const size_t o = ((size_t)g * nb + (b < nb ? b : 0)) * C + c;
                    vs[u] = psum ? psum[o] : 0.f; vq[u] = psq[o]; vp[u] = ppiv[o];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = b0 + 8 * u;
                    if (b < nb) {
                        const double nbk = (double)(min((b + 1) * rpb, Mg) - b * rpb);
                        const double sd = (double)vs[u], sq = (double)vq[u];
                        const double mb = (double)vp[u] + sd / nbk;
                        const double m2b = sq - sd * sd / nbk;
                        const double tot = n + nbk, delta = mb - mu;
                        mu += delta * (nbk / tot);
                        m2 += m2b + delta * delta * (n * nbk / tot);
                        n = tot;
                    }
                }
            }
        sh[0][ty][tx] = n;
        sh[1][ty][tx] = mu;
        sh[2][ty][tx] = m2;
        __syncthreads();
        if (ok && ty == 0) {
            for (int k = 1; k < 8; ++k) {
                const double nk = sh[0][k][tx];
                if (nk > 0.0) {
                    const double tot = n + nk, delta = sh[1][k][tx] - mu;
                    mu += delta * (nk / tot);
                    m2 += sh[2][k][tx] + delta * delta * (n * nk / tot);
                    n = tot;
                }
            }
            double var = m2 / Mg;
            if (var < 0.0) var = 0.0;
            const float r = (float)(1.0 / sqrt(var + (double)eps));
            mean[g * C + c] = (float)mu;
            rstd[g * C + c] = r;
            scale[g * C + c] = gamma[c] * r;
            shift[g * C + c] = beta[c];
            const float unb = (float)(Mg > 1 ? var * ((double)Mg / (double)(Mg - 1)) : var);
            rm = (1.f - momentum) * rm + momentum * (float)mu;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
        __syncthreads();
    }
    if (ok && ty == 0) {
        if (run_mean) run_mean[c] = rm;
        if (run_var) run_var[c] = rv;
    }
}

// eval mode: tables from the running statistics (one "group")
__global__ void bn_eval_prepare_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                       float eps, float* __restrict__ mean, float* __restrict__ scale,
                                       float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    scale[c] = gamma[c] / sqrtf(run_var[c] + eps);
    shift[c] = beta[c];
}

// ---- apply: out = [relu]( (y-mean)*scale+shift  (+ id | + (yd-mean2)*scale2+shift2) ) -----------
// sg = stride (in channels) between groups of the tables: C in training, 0 in eval.
struct BnTab {
    const float *mean, *scale, *shift;
};
// Grid (x, G): blockIdx.y is the BN group, the x blocks grid-stride over the group's chunks with a stride that is a
// multiple of the chunks per row -- so a thread keeps ONE channel chunk for its whole life and its table entries
// are loaded once; the loop body is nothing but 4 independent chunk loads per tensor, the arithmetic and the stores.
// BITS: also the sign mask of the output, one bit per element (IoBwStats::maskbits): the 32 / VEC consecutive lanes that hold
// the 32 channels of one word combine their pieces with shuffles, the first of them stores the word.
template <int MODE, typename T, bool BITS = false>   // MODE 0 none, 1 identity tensor, 2 second BN (downsample branch)
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const T* __restrict__ y, size_t per_group, int cvmask,
                                                           int sg, BnTab t, const T* __restrict__ idt, BnTab t2,
                                                           int relu, T* __restrict__ out, uint32_t* __restrict__ bits = nullptr) {
    constexpr int VEC = Chunk<T>::VEC, NV = Chunk<T>::NV, U = 4;
    const int g = blockIdx.y;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int co = g * sg + (int)(i & (size_t)cvmask) * VEC;
    const Tab<NV> mu = ldt<NV>(t.mean + co), sc = ldt<NV>(t.scale + co), sh = ldt<NV>(t.shift + co);
    Tab<NV> mu2 = mu, sc2 = sc, sh2 = sh;
    if (MODE == 2) {
        mu2 = ldt<NV>(t2.mean + co);
        sc2 = ldt<NV>(t2.scale + co);
        sh2 = ldt<NV>(t2.shift + co);
    }
    const size_t base = (size_t)g * per_group;
    for (; i < per_group; i += U * stride) {
        Chunk<T> a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j < per_group) {
                a[u] = ldc(y + (base + j) * VEC);
                if (MODE != 0) b[u] = ldc(idt + (base + j) * VEC);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j < per_group) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    f32x4 v = bn_affine(a[u].v[k], mu.v[k], sc.v[k], sh.v[k]);
                    if (MODE == 1) v += b[u].v[k];
                    if (MODE == 2) v += bn_affine(b[u].v[k], mu2.v[k], sc2.v[k], sh2.v[k]);
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    a[u].v[k] = v;
                }
                stc(out + (base + j) * VEC, a[u]);
                if constexpr (BITS) {
                    // (the 32 / VEC lanes of a word are all inside the `j < per_group` branch: per_group is a multiple of it)
                    constexpr int LPW = 32 / VEC;
                    unsigned w = 0;
#pragma unroll
                    for (int k = 0; k < NV; ++k)
#pragma unroll
                        for (int e = 0; e < 4; ++e) w |= (a[u].v[k][e] > 0.f ? 1u : 0u) << (k * 4 + e);
                    w <<= (threadIdx.x & (LPW - 1)) * VEC;
#pragma unroll
                    for (int sft = 1; sft < LPW; sft <<= 1) w |= (unsigned)__shfl_xor((int)w, sft, 64);
                    if ((threadIdx.x & (LPW - 1)) == 0) bits[((base + j) * VEC) >> 5] = w;
                }
            }
        }
    }
}

// ---- backward ----------------------------------------------------------------------------------
// dz = dout * [act > 0] (act may be null: no ReLU behind this BN), xhat = (y - mean) * rstd
// partial sums of dz and dz*xhat per (group, block, channel)
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const T* __restrict__ dout,
                                                                const T* __restrict__ act,
                                                                const T* __restrict__ y, int Mg, int C, int rpb,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ mscale,
                                                                const float* __restrict__ mshift,
                                                                float* __restrict__ p1, float* __restrict__ p2) {
    // a thread owns up to two 16-byte channel chunks (4 fp32 / 8 bf16) and strides over rows, 4 rows in flight
    constexpr int VEC = Chunk<T>::VEC, NV = Chunk<T>::NV, U = 4;
    const int CV = C / VEC;
    const ColMap cm = col_map(CV);
    const int g = blockIdx.y, b = blockIdx.x, nb = gridDim.x;
    const int r0 = b * rpb, r1 = min(r0 + rpb, Mg);
    const size_t goff = (size_t)g * Mg * C;
    f32x4 s1[2][NV], s2[2][NV];
    Tab<NV> mu[2], rs[2], msc[2], msh[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            s1[i][k] = 0.f; s2[i][k] = 0.f; mu[i].v[k] = 0.f; rs[i].v[k] = 0.f; msc[i].v[k] = 0.f; msh[i].v[k] = 0.f;
        }
        const int q = cm.tx + cm.TX * i;
        if (i < cm.nq && q < CV) {
            mu[i] = ldt<NV>(mean + g * C + q * VEC);
            rs[i] = ldt<NV>(rstd + g * C + q * VEC);
            if (mscale) {
                msc[i] = ldt<NV>(mscale + g * C + q * VEC);
                msh[i] = ldt<NV>(mshift + g * C + q * VEC);
            }
        }
    }
    for (int r = r0 + cm.ty; r < r1; r += U * cm.TY) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = cm.tx + cm.TX * i;
            if (i < cm.nq && q < CV) {
                Chunk<T> d[U], a[U], yv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int rr = r + u * cm.TY;
                    if (rr < r1) {
                        const size_t ro = goff + (size_t)rr * C + q * VEC;
                        d[u] = ldc(dout + ro);
                        yv[u] = ldc(y + ro);
                        if (act) a[u] = ldc(act + ro);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (r + u * cm.TY < r1) {
#pragma unroll
                        for (int k = 0; k < NV; ++k) {
                            f32x4 dd = d[u].v[k];
                            if (act || mscale) {
                                const f32x4 av = act ? a[u].v[k] : bn_affine(yv[u].v[k], mu[i].v[k], msc[i].v[k], msh[i].v[k]);
#pragma unroll
                                for (int e = 0; e < 4; ++e) dd[e] = av[e] > 0.f ? dd[e] : 0.f;
                            }
                            s1[i][k] += dd;
                            s2[i][k] += dd * ((yv[u].v[k] - mu[i].v[k]) * rs[i].v[k]);
                        }
                    }
                }
            }
        }
    }
    const size_t o = ((size_t)g * nb + b) * C;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        f32x4 t1[2] = {s1[0][k], s1[1][k]}, t2[2] = {s2[0][k], s2[1][k]};
        reduce_rows_store<2>(cm, t1, t2, p1 + o + 4 * k, p2 + o + 4 * k, CV, VEC);
    }
}

// block = 8 channels x 32 lanes over the partials; ALL loads of a lane are issued before the first use (the
// partials were written by other XCDs in the previous kernel, so every dependent round trip is an HBM/MALL
// miss -- the kernel is pure latency, and serial batches were what made it take 20-60 us)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ p1,
                                                             const float* __restrict__ p2, int nb, int G, int Mg,
                                                             int C, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ c1,
                                                             float* __restrict__ c2,
                                                             const float* __restrict__ xgamma = nullptr,
                                                             const float* __restrict__ xmean = nullptr,
                                                             const float* __restrict__ xrstd = nullptr,
                                                             float* __restrict__ c3 = nullptr) {
    constexpr int U = 16, MAXG = 8;
    __shared__ double sh[2][32][8];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;     // tx: channel, ty: partial lane (0..31)
    const int c = blockIdx.x * 8 + tx;
    const bool ok = c < C;
    double dg = 0.0, db = 0.0;
    for (int g = 0; g < G && g < MAXG; ++g) {
        double a = 0.0, b2 = 0.0;
        for (int b0 = ty; b0 < nb; b0 += 32 * U) {
            float v1[U], v2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + 32 * u;
                const bool in = ok && b < nb;
                const size_t o = ((size_t)g * nb + (in ? b : 0)) * C + (ok ? c : 0);
                v1[u] = in ? p1[o] : 0.f;
                v2[u] = in ? p2[o] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { a += (double)v1[u]; b2 += (double)v2[u]; }
        }
        sh[0][ty][tx] = a;
        sh[1][ty][tx] = b2;
        __syncthreads();
        if (ok && ty == 0) {
            for (int k = 1; k < 32; ++k) { a += sh[0][k][tx]; b2 += sh[1][k][tx]; }
            db += a;
            dg += b2;
            if (c3) {
                // coefficient form for the operand transform of the data-gradient kernel (IoBwStats::xb_a/b/c):
                // dy = gamma*rstd*(dz - m1 - (y - mean)*rstd*m2) = A*dz + B*y + Cc
                const double rs = (double)xrstd[g * C + c];
                const double A = (double)xgamma[c] * rs, B = -A * rs * (b2 / Mg);
                c1[g * C + c] = (float)A;
                c2[g * C + c] = (float)B;
                c3[g * C + c] = (float)(-A * (a / Mg) - B * (double)xmean[g * C + c]);
            } else {
                c1[g * C + c] = (float)(a / Mg);
                c2[g * C + c] = (float)(b2 / Mg);
            }
        }
        __syncthreads();
    }
    if (ok && ty == 0) {
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
    }
}

// dy = gamma*rstd*(dz - c1 - xhat*c2); optionally also stores dz (may alias dout).  Same launch geometry as
// bn_apply_kernel: one channel chunk per thread, every per-channel coefficient in registers.
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const T* dout, const T* __restrict__ act,
                                                               const T* __restrict__ y, size_t per_group,
                                                               int cvmask, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const float* __restrict__ c1,
                                                               const float* __restrict__ c2,
                                                               const float* __restrict__ mscale,
                                                               const float* __restrict__ mshift,
                                                               T* __restrict__ dy, T* dz_out) {
    constexpr int VEC = Chunk<T>::VEC, NV = Chunk<T>::NV, U = 4;
    const int g = blockIdx.y;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(i & (size_t)cvmask) * VEC, co = g * C + q;
    const Tab<NV> mu = ldt<NV>(mean + co), rs = ldt<NV>(rstd + co), k1 = ldt<NV>(c1 + co), k2 = ldt<NV>(c2 + co);
    Tab<NV> gr = ldt<NV>(gamma + q), msc = mu, msh = mu;
#pragma unroll
    for (int k = 0; k < NV; ++k) gr.v[k] *= rs.v[k];
    if (mscale) {
        msc = ldt<NV>(mscale + co);
        msh = ldt<NV>(mshift + co);
    }
    const size_t base = (size_t)g * per_group;
    for (; i < per_group; i += U * stride) {
        Chunk<T> d[U], yv[U], av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j < per_group) {
                d[u] = ldc(dout + (base + j) * VEC);
                yv[u] = ldc(y + (base + j) * VEC);
                if (act) av[u] = ldc(act + (base + j) * VEC);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + u * stride;
            if (j < per_group) {
                Chunk<T> r;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    f32x4 dd = d[u].v[k];
                    if (act || mscale) {
                        const f32x4 a = act ? av[u].v[k] : bn_affine(yv[u].v[k], mu.v[k], msc.v[k], msh.v[k]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dd[e] = a[e] > 0.f ? dd[e] : 0.f;
                    }
                    const f32x4 xh = (yv[u].v[k] - mu.v[k]) * rs.v[k];
                    r.v[k] = (dd - k1.v[k] - xh * k2.v[k]) * gr.v[k];
                    d[u].v[k] = dd;
                }
                if (dz_out) stc(dz_out + (base + j) * VEC, d[u]);
                stc(dy + (base + j) * VEC, r);
            }
        }
    }
}

// launch geometry of the two streaming kernels above: x blocks times G groups.  The grid stride (x blocks * 256
// chunks) must be a multiple of the chunks per row (<= 512) -- any even block count -- and should NOT be a power of
// two: a thread's 4 chunks in flight would sit exactly 2^k bytes apart, on the same HBM channel.
static int stream_blocks(size_t per_group, int cv, int G) {
    size_t want = (per_group + (size_t)kThreads * 4 - 1) / ((size_t)kThreads * 4);   // 4 chunks per thread and trip
    size_t cap = 8192 / (size_t)(G > 0 ? G : 1);
    if (want > cap) want = cap;
    int b = (int)(want & ~(size_t)1);
    if (b < 2) b = 2;
    if (b >= 6 && ((b / 2) & 1) == 0) b -= 2;      // 2 * odd
    return b;
}

int ilog2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

int ew_blocks(size_t total4) {
    size_t b = (total4 + kThreads - 1) / kThreads;
    return (int)(b > 4096 ? 4096 : (b ? b : 1));
}

}  // namespace

namespace {
// level-1 merge of per-tile (mean, M2) partials: block = 32 channels x 8 lanes, each block folds `chunk`
// consecutive tiles of one group into one partial (same representation, count implied by position)
__global__ __launch_bounds__(256) void bn_merge_tiles_kernel(const float* __restrict__ tmean,
                                                            const float* __restrict__ tm2, int nt, int rows,
                                                            int Mg, int C, int chunk, float* __restrict__ omean,
                                                            float* __restrict__ om2) {
    __shared__ double sh[3][8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx, g = blockIdx.y, ch = blockIdx.z;
    const bool ok = c < C;
    const int t0 = ch * chunk, t1 = min(t0 + chunk, nt);
    double n = 0.0, mu = 0.0, m2 = 0.0;
    if (ok)
        for (int b0 = t0 + ty; b0 < t1; b0 += 64) {
            float vm[8], vq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 8 * u;
                const size_t o = ((size_t)g * nt + (b < t1 ? b : t0)) * C + c;
                vm[u] = tmean[o]; vq[u] = tm2[o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 8 * u;
                if (b < t1) {
                    const double nbk = (double)(min((b + 1) * rows, Mg) - b * rows);
                    const double tot = n + nbk, delta = (double)vm[u] - mu;
                    mu += delta * (nbk / tot);
                    m2 += (double)vq[u] + delta * delta * (n * nbk / tot);
                    n = tot;
                }
            }
        }
    sh[0][ty][tx] = n; sh[1][ty][tx] = mu; sh[2][ty][tx] = m2;
    __syncthreads();
    if (ok && ty == 0) {
        for (int k = 1; k < 8; ++k) {
            const double nk = sh[0][k][tx];
            if (nk > 0.0) {
                const double tot = n + nk, delta = sh[1][k][tx] - mu;
                mu += delta * (nk / tot);
                m2 += sh[2][k][tx] + delta * delta * (n * nk / tot);
                n = tot;
            }
        }
        const size_t o = ((size_t)g * gridDim.z + ch) * C + c;
        omean[o] = (float)mu;
        om2[o] = (float)m2;
    }
}
}  // namespace

// Statistics from the per-tile partials the conv epilogue wrote (tile_mean/tile_m2: [M/128][C], tiles of
// kIoStatTileRows rows, M/G a multiple of it).  The partial arrays are scratch: the level-1 merge output is
// written behind the tile partials (caller provides room for (nt + nt/64 + G) * C floats in each array).
int io_bn_finalize_tiles(float* tile_mean, float* tile_m2, int M, int C, int G, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         float* mean, float* rstd, float* scale, float* shift, hipStream_t st) {
    const int rows = kIoStatTileRows;
    IO_REQUIRE(G >= 1 && M % G == 0 && (M / G) % rows == 0, IO_ERR_SHAPE,
               "bn_finalize_tiles: rows per group %d must be a multiple of %d", G ? M / G : 0, rows);
    const int Mg = M / G, nt = Mg / rows;
    IoProfScope prof(IO_PROF_BN_STATS, 0.0, 8.0 * (double)(M / rows) * C, st);
    const float* pm = tile_mean;
    const float* pq = tile_m2;
    int nb = nt, rpb = rows;
    if (nt > 128) {
        const int chunk = 64, nch = io_cdiv(nt, chunk);
        float* om = tile_mean + (size_t)G * nt * C;
        float* oq = tile_m2 + (size_t)G * nt * C;
        hipLaunchKernelGGL(bn_merge_tiles_kernel, dim3(io_cdiv(C, 32), G, nch), dim3(256), 0, st, tile_mean, tile_m2,
                           nt, rows, Mg, C, chunk, om, oq);
        pm = om; pq = oq; nb = nch; rpb = rows * chunk;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(io_cdiv(C, 32)), dim3(256), 0, st, (const float*)nullptr, pq, pm, nb,
                       rpb, G, Mg, C, gamma, beta, running_mean, running_var, momentum, eps, mean, rstd, scale, shift);
    return io_check_launch("bn_finalize_tiles");
}

namespace {
// plain sums of per-tile partials: block = 32 channels x 8 lanes, `chunk` consecutive tiles of one group
__global__ __launch_bounds__(256) void bn_sum_tiles_kernel(const float* __restrict__ t1, const float* __restrict__ t2,
                                                          int nt, int C, int chunk, float* __restrict__ o1,
                                                          float* __restrict__ o2) {
    __shared__ double sh[2][8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx, g = blockIdx.y, ch = blockIdx.z;
    const bool ok = c < C;
    const int t0 = ch * chunk, t1e = min(t0 + chunk, nt);
    double a = 0.0, b2 = 0.0;
    if (ok)
        for (int b0 = t0 + ty; b0 < t1e; b0 += 64) {
            float v1[8], v2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 8 * u;
                const size_t o = ((size_t)g * nt + (b < t1e ? b : t0)) * C + c;
                v1[u] = b < t1e ? t1[o] : 0.f;
                v2[u] = b < t1e ? t2[o] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += (double)v1[u]; b2 += (double)v2[u]; }
        }
    sh[0][ty][tx] = a; sh[1][ty][tx] = b2;
    __syncthreads();
    if (ok && ty == 0) {
        for (int k = 1; k < 8; ++k) { a += sh[0][k][tx]; b2 += sh[1][k][tx]; }
        const size_t o = ((size_t)g * gridDim.z + ch) * C + c;
        o1[o] = (float)a;
        o2[o] = (float)b2;
    }
}
}  // namespace

// BN backward when sum(dz), sum(dz*xhat) per (128-row tile, channel) were already produced by the epilogue of
// the kernel that wrote dz (IoBwStats): merge the tile partials, finalize, apply.  dz is already masked.
// p1/p2 need room for (tiles + tiles/64 + G) * C floats each.
int io_bn_bwd_from_tiles(float* p1, float* p2, const void* dz, const void* y, int M, int C, int G,
                         const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                         void* dy, float* coef, hipStream_t st, int dt) {
    const int sh = ilog2_exact(C / 4);
    IO_REQUIRE(C % 4 == 0 && sh >= 0 && C <= 2048, IO_ERR_SHAPE, "bn_bwd_from_tiles: C=%d unsupported", C);
    IO_REQUIRE(G >= 1 && M % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "bn_bwd_from_tiles: rows per group must be a multiple of %d", kIoStatTileRows);
    const int Mg = M / G, nt = Mg / kIoStatTileRows;
    IoProfScope prof(IO_PROF_BN_BWD, 0.0, (double)io_dtype_bytes(dt) * M * C * 3.0, st);
    const float* q1 = p1;
    const float* q2 = p2;
    int nb = nt;
    if (nt > 64) {
        const int chunk = 64, nch = io_cdiv(nt, chunk);
        float* o1 = p1 + (size_t)G * nt * C;
        float* o2 = p2 + (size_t)G * nt * C;
        hipLaunchKernelGGL(bn_sum_tiles_kernel, dim3(io_cdiv(C, 32), G, nch), dim3(256), 0, st, p1, p2, nt, C, chunk, o1,
                           o2);
        q1 = o1; q2 = o2; nb = nch;
    }
    float* c1 = coef;
    float* c2 = coef + (size_t)G * C;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(io_cdiv(C, 8)), dim3(256), 0, st, q1, q2, nb, G, Mg, C, dgamma,
                       dbeta, c1, c2);
    const int vec = 16 / io_dtype_bytes(dt), cv = C / vec;
    const size_t per_group = (size_t)Mg * cv;
    dim3 agrid(stream_blocks(per_group, cv, G), G);
    if (dt == IO_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, agrid, dim3(kThreads), 0, st, (const bf16_t*)dz,
                           (const bf16_t*)nullptr, (const bf16_t*)y, per_group, cv - 1, C, gamma, mean, rstd, c1, c2,
                           (const float*)nullptr, (const float*)nullptr, (bf16_t*)dy, (bf16_t*)nullptr);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, agrid, dim3(kThreads), 0, st, (const float*)dz,
                           (const float*)nullptr, (const float*)y, per_group, cv - 1, C, gamma, mean, rstd, c1, c2,
                           (const float*)nullptr, (const float*)nullptr, (float*)dy, (float*)nullptr);
    return io_check_launch("bn_bwd_from_tiles");
}

int io_bn_bwd_coefs_from_tiles(float* p1, float* p2, int M, int C, int G, const float* gamma, const float* mean,
                               const float* rstd, float* dgamma, float* dbeta, float* coef, hipStream_t st) {
    IO_REQUIRE(C % 4 == 0 && C <= 2048, IO_ERR_SHAPE, "bn_bwd_coefs_from_tiles: C=%d unsupported", C);
    IO_REQUIRE(G >= 1 && M % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "bn_bwd_coefs_from_tiles: rows per group must be a multiple of %d", kIoStatTileRows);
    const int Mg = M / G, nt = Mg / kIoStatTileRows;
    IoProfScope prof(IO_PROF_BN_BWD, 0.0, 8.0 * (double)(M / kIoStatTileRows) * C, st);
    const float* q1 = p1;
    const float* q2 = p2;
    int nb = nt;
    if (nt > 64) {
        const int chunk = 64, nch = io_cdiv(nt, chunk);
        float* o1 = p1 + (size_t)G * nt * C;
        float* o2 = p2 + (size_t)G * nt * C;
        hipLaunchKernelGGL(bn_sum_tiles_kernel, dim3(io_cdiv(C, 32), G, nch), dim3(256), 0, st, p1, p2, nt, C, chunk, o1,
                           o2);
        q1 = o1; q2 = o2; nb = nch;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(io_cdiv(C, 8)), dim3(256), 0, st, q1, q2, nb, G, Mg, C, dgamma,
                       dbeta, coef, coef + (size_t)G * C, gamma, mean, rstd, coef + (size_t)2 * G * C);
    return io_check_launch("bn_bwd_coefs_from_tiles");
}

static int bn_rows_per_block(int Mg, int G, int* nb) {
    int want = 1024 / (G > 0 ? G : 1);
    if (want < 1) want = 1;
    int rpb = io_cdiv(Mg, want);
    if (rpb < 64) rpb = Mg < 64 ? Mg : 64;
    *nb = io_cdiv(Mg, rpb);
    return rpb;
}

extern "C" size_t io_bn_partial_floats(int M, int C, int G) {
    if (M <= 0 || G <= 0) return 0;
    int nb;
    bn_rows_per_block(M / G, G, &nb);
    return (size_t)3 * G * nb * C;
}

int io_bn_stats_finalize_t(const void* y, int M, int C, int G, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* mean,
                           float* rstd, float* scale, float* shift, float* partial, size_t partial_floats,
                           hipStream_t st, int dt) {
    IO_REQUIRE(C % 4 == 0 && C <= 2048 && ilog2_exact(C / 4) >= 0, IO_ERR_SHAPE, "bn_stats: C=%d unsupported", C);
    IO_REQUIRE(G >= 1 && M % G == 0, IO_ERR_SHAPE, "bn_stats: M=%d not divisible by G=%d", M, G);
    IO_REQUIRE(partial_floats >= io_bn_partial_floats(M, C, G), IO_ERR_WORKSPACE, "bn_stats: partial too small");
    const int Mg = M / G;
    int nb;
    const int rpb = bn_rows_per_block(Mg, G, &nb);
    float* psum = partial;
    float* psq = partial + (size_t)G * nb * C;
    float* ppiv = partial + (size_t)2 * G * nb * C;
    IoProfScope prof(IO_PROF_BN_STATS, 0.0, (double)io_dtype_bytes(dt) * M * C, st);
    if (dt == IO_BF16)
        hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(nb, G), dim3(kThreads), 0, st, (const bf16_t*)y, Mg, C, rpb,
                           psum, psq, ppiv);
    else
        hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(nb, G), dim3(kThreads), 0, st, (const float*)y, Mg, C, rpb,
                           psum, psq, ppiv);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(io_cdiv(C, 32)), dim3(256), 0, st, psum, psq, ppiv, nb, rpb, G, Mg,
                       C, gamma, beta, running_mean, running_var, momentum, eps, mean, rstd, scale, shift);
    return io_check_launch("bn_stats_finalize");
}

extern "C" int io_bn_stats_finalize(const float* y, int M, int C, int G, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps,
                                    float* mean, float* rstd, float* scale, float* shift, float* partial,
                                    size_t partial_floats, hipStream_t st) {
    return io_bn_stats_finalize_t(y, M, C, G, gamma, beta, running_mean, running_var, momentum, eps, mean, rstd,
                                  scale, shift, partial, partial_floats, st, IO_F32);
}

extern "C" int io_bn_eval_prepare(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* mean, float* scale, float* shift,
                                  hipStream_t st) {
    hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(io_cdiv(C, 256)), dim3(256), 0, st, C, gamma, beta,
                       running_mean, running_var, eps, mean, scale, shift);
    return io_check_launch("bn_eval_prepare");
}

int io_bn_apply_t(const void* y, int M, int C, int G, int per_group_tables, const float* mean, const float* scale,
                  const float* shift, const void* identity, const float* mean2, const float* scale2,
                  const float* shift2, int relu, void* out, hipStream_t st, int dt, uint32_t* bits) {
    const int vec = 16 / io_dtype_bytes(dt);
    IO_REQUIRE(!bits || (C % 32 == 0 && relu), IO_ERR_SHAPE, "bn_apply: the sign mask needs C %% 32 == 0 and a ReLU");
    IO_REQUIRE(C % vec == 0 && ilog2_exact(C / vec) >= 0, IO_ERR_SHAPE, "bn_apply: C=%d must be %d*2^k", C, vec);
    IO_REQUIRE(G >= 1 && M % G == 0, IO_ERR_SHAPE, "bn_apply: M=%d not divisible by G=%d", M, G);
    const int Mg = M / G, sg = per_group_tables ? C : 0, cv = C / vec;
    const size_t per_group = (size_t)Mg * cv;
    dim3 grid(stream_blocks(per_group, cv, G), G), block(kThreads);
    IoProfScope prof(IO_PROF_BN_APPLY, 0.0, (double)io_dtype_bytes(dt) * M * C * (identity ? 3.0 : 2.0), st);
    const BnTab t{mean, scale, shift}, t2{mean2, scale2, shift2};
    const int mode = (identity && scale2) ? 2 : (identity ? 1 : 0);
#define IO_BN_APPLY(MODE_, T_)                                                                                     \
    do {                                                                                                           \
        if (bits)                                                                                                  \
            hipLaunchKernelGGL((bn_apply_kernel<MODE_, T_, true>), grid, block, 0, st, (const T_*)y, per_group,    \
                               cv - 1, sg, t, (const T_*)identity, t2, relu, (T_*)out, bits);                      \
        else                                                                                                       \
            hipLaunchKernelGGL((bn_apply_kernel<MODE_, T_, false>), grid, block, 0, st, (const T_*)y, per_group,   \
                               cv - 1, sg, t, (const T_*)identity, t2, relu, (T_*)out, (uint32_t*)nullptr);        \
    } while (0)
    if (dt == IO_BF16) {
        if (mode == 2) IO_BN_APPLY(2, bf16_t);
        else if (mode == 1) IO_BN_APPLY(1, bf16_t);
        else IO_BN_APPLY(0, bf16_t);
    } else {
        if (mode == 2) IO_BN_APPLY(2, float);
        else if (mode == 1) IO_BN_APPLY(1, float);
        else IO_BN_APPLY(0, float);
    }
#undef IO_BN_APPLY
    return io_check_launch("bn_apply");
}

extern "C" int io_bn_apply(const float* y, int M, int C, int G, int per_group_tables, const float* mean,
                           const float* scale, const float* shift, const float* identity, const float* mean2,
                           const float* scale2, const float* shift2, int relu, float* out, hipStream_t st) {
    return io_bn_apply_t(y, M, C, G, per_group_tables, mean, scale, shift, identity, mean2, scale2, shift2, relu, out,
                         st, IO_F32);
}

int io_bn_bwd_t(const void* dout, const void* act, const float* mask_scale, const float* mask_shift, const void* y,
                int M, int C, int G, const float* gamma, const float* mean, const float* rstd, float* dgamma,
                float* dbeta, void* dy, void* dz_out, float* partial, size_t partial_floats, float* coef,
                hipStream_t st, int dt) {
    IO_REQUIRE(!(act && mask_scale), IO_ERR_SHAPE, "bn_bwd: give the activation OR the mask tables, not both");
    IO_REQUIRE((mask_scale == nullptr) == (mask_shift == nullptr), IO_ERR_SHAPE, "bn_bwd: mask tables come in pairs");
    const int sh = ilog2_exact(C / 4);
    IO_REQUIRE(C % 4 == 0 && sh >= 0 && C <= 2048, IO_ERR_SHAPE, "bn_bwd: C=%d unsupported", C);
    IO_REQUIRE(G >= 1 && M % G == 0, IO_ERR_SHAPE, "bn_bwd: M=%d not divisible by G=%d", M, G);
    IO_REQUIRE(partial_floats >= io_bn_partial_floats(M, C, G), IO_ERR_WORKSPACE, "bn_bwd: partial too small");
    const int Mg = M / G;
    int nb;
    const int rpb = bn_rows_per_block(Mg, G, &nb);
    float* p1 = partial;
    float* p2 = partial + (size_t)G * nb * C;
    float* c1 = coef;
    float* c2 = coef + (size_t)G * C;
    IoProfScope prof(IO_PROF_BN_BWD, 0.0,
                     (double)io_dtype_bytes(dt) * M * C * ((act ? 6.0 : 4.0) + 1.0 + (dz_out ? 1.0 : 0.0)), st);
    const int vec = 16 / io_dtype_bytes(dt), cv = C / vec;
    const size_t per_group = (size_t)Mg * cv;
    dim3 agrid(stream_blocks(per_group, cv, G), G);
#define IO_BN_BWD(T_)                                                                                               \
    do {                                                                                                            \
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<T_>, dim3(nb, G), dim3(kThreads), 0, st, (const T_*)dout,           \
                           (const T_*)act, (const T_*)y, Mg, C, rpb, mean, rstd, mask_scale, mask_shift, p1, p2);   \
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(io_cdiv(C, 8)), dim3(256), 0, st, p1, p2, nb, G, Mg, C,     \
                           dgamma, dbeta, c1, c2);                                                                  \
        hipLaunchKernelGGL(bn_bwd_apply_kernel<T_>, agrid, dim3(kThreads), 0, st, (const T_*)dout,                  \
                           (const T_*)act, (const T_*)y, per_group, cv - 1, C, gamma, mean, rstd, c1, c2,           \
                           mask_scale, mask_shift, (T_*)dy, (T_*)dz_out);                                           \
    } while (0)
    if (dt == IO_BF16) IO_BN_BWD(bf16_t);
    else IO_BN_BWD(float);
#undef IO_BN_BWD
    return io_check_launch("bn_bwd");
}

extern "C" int io_bn_bwd(const float* dout, const float* act, const float* mask_scale, const float* mask_shift,
                         const float* y, int M, int C, int G, const float* gamma, const float* mean,
                         const float* rstd, float* dgamma, float* dbeta, float* dy, float* dz_out, float* partial,
                         size_t partial_floats, float* coef, hipStream_t st) {
    return io_bn_bwd_t(dout, act, mask_scale, mask_shift, y, M, C, G, gamma, mean, rstd, dgamma, dbeta, dy, dz_out,
                       partial, partial_floats, coef, st, IO_F32);
}

// mask_scale / mask_shift != nullptr: dz is the gradient of relu(bn(y)) and the ReLU mask is recomputed from y with the
// forward tables (as io_bn_bwd_t's mask mode does); whoever applies the coefficients must apply the same mask
int io_bn_bwd_coefs_t(const void* dz, const void* y, int M, int C, int G, const float* gamma, const float* mean,
                      const float* rstd, float* dgamma, float* dbeta, float* coef, float* partial,
                      size_t partial_floats, hipStream_t st, int dt, const float* mask_scale, const float* mask_shift) {
    const int sh = ilog2_exact(C / 4);
    IO_REQUIRE(C % 4 == 0 && sh >= 0 && C <= 2048, IO_ERR_SHAPE, "bn_bwd_coefs: C=%d unsupported", C);
    IO_REQUIRE(G >= 1 && M % G == 0, IO_ERR_SHAPE, "bn_bwd_coefs: M=%d not divisible by G=%d", M, G);
    IO_REQUIRE((mask_scale == nullptr) == (mask_shift == nullptr), IO_ERR_SHAPE, "bn_bwd_coefs: the mask tables come in pairs");
    IO_REQUIRE(partial_floats >= io_bn_partial_floats(M, C, G), IO_ERR_WORKSPACE, "bn_bwd_coefs: partial too small");
    const int Mg = M / G;
    int nb;
    const int rpb = bn_rows_per_block(Mg, G, &nb);
    float* p1 = partial;
    float* p2 = partial + (size_t)G * nb * C;
    IoProfScope prof(IO_PROF_BN_BWD, 0.0, (double)io_dtype_bytes(dt) * M * C * 2.0, st);
    if (dt == IO_BF16)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nb, G), dim3(kThreads), 0, st, (const bf16_t*)dz,
                           (const bf16_t*)nullptr, (const bf16_t*)y, Mg, C, rpb, mean, rstd, mask_scale, mask_shift, p1, p2);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nb, G), dim3(kThreads), 0, st, (const float*)dz,
                           (const float*)nullptr, (const float*)y, Mg, C, rpb, mean, rstd, mask_scale, mask_shift, p1, p2);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(io_cdiv(C, 8)), dim3(256), 0, st, p1, p2, nb, G, Mg, C, dgamma, dbeta,
                       coef, coef + (size_t)G * C, gamma, mean, rstd, coef + (size_t)2 * G * C);
    return io_check_launch("bn_bwd_coefs");
}

namespace {
__global__ __launch_bounds__(256) void bn_resid2_tables_kernel(const float* __restrict__ m3, const float* __restrict__ s3,
                                                              const float* __restrict__ h3, const float* __restrict__ md,
                                                              const float* __restrict__ sd, const float* __restrict__ hd,
                                                              int n, float* __restrict__ coef) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    coef[i] = s3[i];
    coef[n + i] = sd[i];
    coef[2 * n + i] = (float)(((double)h3[i] - (double)m3[i] * (double)s3[i]) + ((double)hd[i] - (double)md[i] * (double)sd[i]));
}
}  // namespace

int io_bn_resid2_tables(const float* mean3, const float* scale3, const float* shift3, const float* meand,
                        const float* scaled, const float* shiftd, int G, int C, float* coef, hipStream_t st) {
    const int n = G * C;
    IoProfScope prof(IO_PROF_BN_STATS, 0.0, 36.0 * n, st);
    hipLaunchKernelGGL(bn_resid2_tables_kernel, dim3(io_cdiv(n, 256)), dim3(256), 0, st, mean3, scale3, shift3, meand, scaled,
                       shiftd, n, coef);
    return io_check_launch("bn_resid2_tables");
}
