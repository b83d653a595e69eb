// Shared declarations for the instaorder_hip library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/instaorder_hip.h"

// ---- error plumbing (thread-local last-error string, never throws) ----------
void io_set_error(const char* fmt, ...);
int io_check_launch(const char* what);

#define IO_REQUIRE(cond, code, ...)        \
    do {                                   \
        if (!(cond)) {                     \
            io_set_error(__VA_ARGS__);     \
            return (code);                 \
        }                                  \
    } while (0)

static inline int io_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// exact n / d for 0 <= n < 2^31 by multiply-high: q = umulhi(n, magic) >> shift (shift < 0: d == 1)
struct IoFastDiv {
    unsigned magic;
    int shift;
};
static inline IoFastDiv io_fastdiv(int d) {
    IoFastDiv f;
    if (d <= 1) { f.magic = 0; f.shift = -1; return f; }
    int s = 0;
    while ((1LL << s) < d) ++s;                       // 2^(s-1) < d <= 2^s
    const unsigned long long num = 1ULL << (31 + s);
    f.magic = (unsigned)((num + (unsigned long long)d - 1) / (unsigned long long)d);   // ceil, < 2^32
    f.shift = s - 1;
    return f;
}

// Geometry of one implicit-GEMM launch ("gather conv"): forward convolution and
// both data-gradient forms are instances of it (see conv_igemm.hip).
struct IoConvGeom {
    int N, Hi, Wi, Ci;   // gathered tensor  [N,Hi,Wi,Ci]   (NHWC)
    int Ho, Wo, Co;      // logical output grid of THIS launch, Co channels
    int outH, outW;      // spatial dims of the output tensor in memory
    int os, ooh, oow;    // output lattice: mem (h,w) = (ho*os+ooh, wo*os+oow)
    int is;              // hi = ho*is + dh,  wi = wo*is + dw
    int Th, Tw;          // tap grid of this launch
    int dh0, dhs, dw0, dws;   // dh = dh0 + dhs*th ; dw = dw0 + dws*tw
    int r0, rs, s0, ss;       // filter tap (r,s) = (r0+rs*th, s0+ss*tw)
    int S, wT;           // filter width S and total taps wT=R*S: weights [Co][wT][Ci]
    IoFastDiv fd_howo, fd_wo;   // division by Ho*Wo and by Wo (filled by io_geom_finish)
};
static inline void io_geom_finish(IoConvGeom& g) {
    g.fd_howo = io_fastdiv(g.Ho * g.Wo);
    g.fd_wo = io_fastdiv(g.Wo);
}

// Optional epilogue of a data-gradient launch: the BN-backward reductions of the BN layer that produced the
// tensor this gradient belongs to (sum dz, sum dz*xhat per 128-row tile and channel), so the separate reduce
// pass over (dz, y) disappears.  y: that BN's input (the conv output); mean/rstd (+ optional scale/shift to
// recompute the ReLU mask from y) are its [G][C] tables; Mg rows per group (a multiple of 128).
struct IoBwStats {
    const float* y;
    const float *mean, *rstd, *mscale, *mshift;
    float *p1, *p2;     // [M/128][C] tile partials
    int Mg;
};

// internal launchers shared between the C ABI and the network executor
int io_launch_conv_nt(const IoConvGeom& g, const float* in, const float* wgt, float* out,
                      const float* add, const float* mask, int stem, hipStream_t st, float* st_mean = nullptr,
                      float* st_m2 = nullptr, const IoBwStats* bw = nullptr);
int io_bn_bwd_from_tiles(float* p1, float* p2, const float* dz, const float* y, int M, int C, int G,
                         const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                         float* dy, float* coef, hipStream_t st);
constexpr int kIoStatTileRows = 128;   // row-tile height of the conv kernel = granule of fused BN statistics
int io_bn_finalize_tiles(float* tile_mean, float* tile_m2, int M, int C, int G, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         float* mean, float* rstd, float* scale, float* shift, hipStream_t st);
int io_launch_conv_wgrad(const IoConvGeom& g, const float* in, const float* dy, float* dw,
                         float* partial, size_t partial_bytes, int stem, hipStream_t st);
size_t io_conv_wgrad_partial_bytes(const IoConvGeom& g, int stem);

IoConvGeom io_geom_fwd(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad);
IoConvGeom io_geom_dgrad(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int ph, int pw);
int io_run_dgrad(const float* dy, const float* wt, float* dx, const float* add, const float* mask, int N, int H,
                 int W, int Cin, int Cout, int R, int S, int stride, int pad, hipStream_t st,
                 const IoBwStats* bw = nullptr);

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench / profiling) ----
enum IoProfClass {
    IO_PROF_CONV_NT128 = 0, IO_PROF_CONV_NT64, IO_PROF_CONV_STEM, IO_PROF_WGRAD, IO_PROF_WGRAD_STEM,
    IO_PROF_BN_STATS, IO_PROF_BN_APPLY, IO_PROF_BN_BWD, IO_PROF_POOL_HEAD, IO_PROF_TRANSPOSE, IO_PROF_PACK,
    IO_PROF_LOSS, IO_PROF_SGD, IO_PROF_NCLASS
};
struct IoProfScope {
    int idx;
    hipStream_t st;
    IoProfScope(int cls, double flops, double bytes, hipStream_t stream);
    ~IoProfScope();
};
