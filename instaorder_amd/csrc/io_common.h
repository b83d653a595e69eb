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

// Per-device one-time set-up of a launcher (the dynamic-LDS opt-in of a kernel, the CU count): `mask` holds one bit per
// device id; returns true the first time it is asked on the CURRENT device (a process may launch on several devices; two
// threads racing here both do the idempotent set-up).  *dev_out: the current device id.
#include <atomic>
// CUs of the current device, cached per device ordinal (the persistent grids of conv_p256.hip / conv_halo3.hip are sized by it)
static inline int io_device_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        hipDeviceProp_t p;
        n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
static inline bool io_first_on_device(std::atomic<unsigned long long>& mask, int* dev_out = nullptr) {
    int d = 0;
    (void)hipGetDevice(&d);
    if (dev_out) *dev_out = d;
    const unsigned long long bit = 1ull << (d & 63);
    if (mask.load(std::memory_order_relaxed) & bit) return false;
    mask.fetch_or(bit, std::memory_order_relaxed);
    return true;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short bf16_t;          // bfloat16 storage (activations / activation gradients in bf16 mode)

// storage type of activation-like tensors.  fp32: everything as in the reference.  bf16: activations, their
// gradients and the GEMM operands are bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate); parameters, their
// gradients, BatchNorm statistics, losses and the optimiser stay fp32.
enum IoDType { IO_F32 = 0, IO_BF16 = 1 };
static inline int io_dtype_bytes(int dt) { return dt == IO_BF16 ? 2 : 4; }

#ifdef __HIPCC__
__device__ __forceinline__ float io_bf2f(bf16_t v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
// round to nearest even in hardware (v_cvt_pk_bf16_f32 on gfx950: one instruction per PAIR of values, NaN stays NaN)
typedef __bf16 io_bf16x2 __attribute__((ext_vector_type(2)));
typedef float io_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t io_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned io_f2bf2(float lo, float hi) {     // lo in bits [15:0], hi in bits [31:16]
    const io_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, io_bf16x2));
}
// 4 consecutive elements <-> float4
__device__ __forceinline__ f32x4 io_ldv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 io_ldv(const bf16_t* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    f32x4 v;
    v[0] = __builtin_bit_cast(float, r.x << 16);
    v[1] = __builtin_bit_cast(float, r.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, r.y << 16);
    v[3] = __builtin_bit_cast(float, r.y & 0xffff0000u);
    return v;
}
__device__ __forceinline__ void io_stv(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void io_stv(bf16_t* p, f32x4 v) {
    uint2 r;
    r.x = io_f2bf2(v[0], v[1]);
    r.y = io_f2bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = r;
}
#endif

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
    int gw;              // 0: dense.  > 0: grouped convolution run as a block-diagonal one: output channels
                         // [t*gw, (t+1)*gw) read only input channels [t*gw, (t+1)*gw); weights [Co][wT][gw]
    int cr;              // stem only.  > 0: exact-K mode: the reduction runs over the wT x cr REAL channels of the packed
                         // x8 input (k = tap * cr + channel), filters / filter gradients are [Co][io_stem_kp(wT, cr)]
    IoFastDiv fd_howo, fd_wo;   // division by Ho*Wo and by Wo (filled by io_geom_finish)
};
static inline int io_stem_kp(int wT, int cr) { return (wT * cr + 31) / 32 * 32; }   // packed row length (k-tiles of 32)
static inline void io_geom_finish(IoConvGeom& g) {
    g.fd_howo = io_fastdiv(g.Ho * g.Wo);
    g.fd_wo = io_fastdiv(g.Wo);
}

// Optional epilogue of a data-gradient launch: the BN-backward reductions of the BN layer that produced the
// tensor this gradient belongs to (sum dz, sum dz*xhat per 128-row tile and channel), so the separate reduce
// pass over (dz, y) disappears.  y: that BN's input (the conv output); mean/rstd (+ optional scale/shift to
// recompute the ReLU mask from y) are its [G][C] tables; Mg rows per group (a multiple of 128).
struct IoBwStats {
    const void* y;       // same storage type as the gradient tensor being written
    const float *mean, *rstd, *mscale, *mshift;
    float *p1, *p2;     // [M/128][C] tile partials
    int Mg;
    void* a_out;        // optional (needs mscale / mshift): relu(bn(y)) written next to the gradient, same shape and type
    // Independent of the above (y may be null): inference epilogue out = [relu](acc + bias[o] (+ add)) -- the
    // BatchNorm of an eval-mode forward folded into the convolution (filters pre-scaled by gamma * rstd).
    const float* bias;
    int relu;
    // Independent again: INPUT transform of a forward convolution.  The A operand is read through the BatchNorm + ReLU of
    // the layer that produced it -- in = relu((in_raw - in_mean[g][c]) * in_scale[g][c] + in_shift[g][c]), zero in the
    // padding -- so that layer's normalised activation is never written to memory.  Tables [G][Ci] exactly as
    // bn_apply_kernel takes them (mean, scale = gamma * rstd, shift = beta; in_mean may be null = 0), evaluated with the
    // same fma, so the ReLU mask the backward recomputes from the raw tensor agrees bit for bit; in_Mg: OUTPUT rows per
    // BatchNorm group (a multiple of 128: a tile never straddles two groups; the rows a tile gathers belong to the
    // samples of its output rows).
    const float *in_mean, *in_scale, *in_shift;
    int in_Mg;
    // Independent again: BACKWARD operand transform of a stride-1 data-gradient launch.  `in` is then not the gradient dy
    // of the convolution's output but the (ReLU-masked) gradient dz of the BatchNorm output behind it, and the A operand
    // is that BatchNorm's input gradient evaluated while the chunk is staged: dy = xb_a[g][c] * dz + xb_b[g][c] * y +
    // xb_c[g][c] with y (xb_y, shaped like `in`) the BatchNorm input and the [G][Ci] tables of io_bn_bwd_coefs* (a =
    // gamma * rstd, b = -a * rstd * mean(dz * xhat), c = -a * mean(dz) - b * mean) -- zero in the padding -- so the
    // BatchNorm backward has no apply pass.  xb_out (optional): dy as a tensor shaped like `in`, written by the blocks of
    // the first output-channel tile (centre tap), for the filter gradient that runs next.  xb_Mg: rows per BatchNorm
    // group (a multiple of 128; input and output grids of the launch coincide).
    const void* xb_y;
    const float *xb_a, *xb_b, *xb_c;
    void* xb_out;
    int xb_Mg;
    // xb_res != 0: the same staging for a FORWARD 1x1 convolution (fp32, dense, whole tiles): `in` is the raw output y3 of
    // the previous block's conv3, xb_y that block's identity tensor, and the operand is the block output relu((y3 -
    // xb_b) * xb_a + xb_c + identity) -- xb_a / xb_b / xb_c = the scale / mean / shift tables of bn3, bn_apply's expression
    // -- with xb_out receiving it as a tensor (the next block's identity, the backward's mask and filter-gradient operand).
    // xb_res == 2: the block has a downsample branch -- operand = relu(xb_a * y3 + xb_b * yd + xb_c) with xb_y = yd (the
    // downsample convolution's output) and the two BatchNorms folded into one table set (io_bn_resid2_tables).
    int xb_res;
    // xb_res != 0 on the 256-row bf16 kernel (conv_p256.hip): [operand > 0] of the tensor written to xb_out as one bit per
    // element (the layout of maskbits below), for the data gradient that later masks by this block output.  Optional.
    uint32_t* xb_bits;
    // Independent again: scratch for the Winograd form of 3x3 stride-1 same-size launches (fp32, Wo even, whole 128-row
    // tiles, no add / mask): 18 * Co * Ci floats that the launcher fills with the transformed filters ([filter row][4 | 6][Co][Ci],
    // wino_filter_kernel) before it starts conv_nt_kernel<..., WINO>.  Null: the direct form.  (A caller-owned buffer
    // because no entry point of the library allocates.)
    float* wino_u;
    // Independent again: the ReLU mask of a data-gradient epilogue ALSO as one bit per element, next to `mask` (same information):
    // word (m * Co + c) / 32, bit c % 32 set where the masking activation was > 0 (io_bn_apply_t writes it).  The 256-row bf16
    // kernel (conv_p256.hip) reads it instead of the tensor -- 1/16 of the bytes; conv_nt_kernel ignores it and reads `mask`.
    const uint32_t* maskbits;
};

// run-time switch of the Winograd row forms (io_set_winograd / IO_WINOGRAD, capi.hip)
bool io_wino_on();

// internal launchers shared between the C ABI and the network executor
// dt_in: storage of `in` and `wgt`; dt_out: storage of out / add / mask / bw.y
int io_launch_conv_nt(const IoConvGeom& g, const void* in, const void* wgt, void* out, const void* add,
                      const void* mask, int stem, hipStream_t st, float* st_mean = nullptr,
                      float* st_m2 = nullptr, const IoBwStats* bw = nullptr, int dt_in = IO_F32,
                      int dt_out = IO_F32);
// conv_p256.hip: would the 256-row bf16 kernel take a dense 1x1 launch [M x Ci] -> [M x Co] with an operand form (xb_a)?
bool io_conv_p256_takes_xop(long M, int Ci, int Co, int Mg);
// conv_p256.hip: IO_OK = launched, 1 = not this kernel's shape / form (fall through), < 0 = error
int io_launch_conv_p256(const IoConvGeom& g, const void* in, const void* wgt, void* out, const void* add, const void* mask,
                        hipStream_t st, float* st_mean, float* st_m2, const IoBwStats* bw, size_t in_bytes,
                        unsigned w_bytes, size_t out_bytes);
int io_bf16_persist_mode();   // io_set_bf16_p256's value: 0 = neither kernel, 1 = both, 2 = conv_p256 without conv_halo3, 3 = both, every eligible shape
// conv_halo3.hip: the same contract for the narrow 3x3 stride-1 layers (input staged once per tile as a halo image)
int io_launch_conv_halo3(const IoConvGeom& g, const void* in, const void* wgt, void* out, const void* add, const void* mask,
                         hipStream_t st, float* st_mean, float* st_m2, const IoBwStats* bw, size_t in_bytes,
                         unsigned w_bytes, size_t out_bytes);
int io_launch_conv_stem_halo(const IoConvGeom& g, const void* in, const void* wgt, void* out, hipStream_t st, float* st_mean,
                             float* st_m2, const IoBwStats* bw, size_t in_bytes, unsigned w_bytes, size_t out_bytes);
int io_bn_bwd_from_tiles(float* p1, float* p2, const void* dz, const void* y, int M, int C, int G,
                         const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                         void* dy, float* coef, hipStream_t st, int dt = IO_F32);
// the same without the apply pass: dgamma / dbeta and the three [G][C] coefficient tables of IoBwStats::xb_a/b/c
// (coef: 3*G*C floats, a | b | c)
int io_bn_bwd_coefs_from_tiles(float* p1, float* p2, int M, int C, int G, const float* gamma, const float* mean,
                               const float* rstd, float* dgamma, float* dbeta, float* coef, hipStream_t st);
// ... and from (dz, y) themselves: reduction pass + finalize, no apply.  dz must already carry the ReLU mask.
int io_bn_bwd_coefs_t(const void* dz, const void* y, int M, int C, int G, const float* gamma, const float* mean,
                      const float* rstd, float* dgamma, float* dbeta, float* coef, float* partial,
                      size_t partial_floats, hipStream_t st, int dt, const float* mask_scale = nullptr,
                      const float* mask_shift = nullptr);
// storage-typed internals behind the fp32 C entry points of the same name (dt: IoDType of the tensors)
int io_bn_stats_finalize_t(const void* y, int M, int C, int G, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* mean,
                           float* rstd, float* scale, float* shift, float* partial, size_t partial_floats,
                           hipStream_t st, int dt);
// bits (optional, C % 32 == 0): [M][C / 32] words, bit c % 32 of word (m C + c) / 32 = (out[m][c] > 0) -- IoBwStats::maskbits
int io_bn_apply_t(const void* y, int M, int C, int G, int per_group_tables, const float* mean, const float* scale,
                  const float* shift, const void* identity, const float* mean2, const float* scale2,
                  const float* shift2, int relu, void* out, hipStream_t st, int dt, uint32_t* bits = nullptr);
int io_bn_bwd_t(const void* dout, const void* act, const float* mask_scale, const float* mask_shift, const void* y,
                int M, int C, int G, const float* gamma, const float* mean, const float* rstd, float* dgamma,
                float* dbeta, void* dy, void* dz_out, float* partial, size_t partial_floats, float* coef,
                hipStream_t st, int dt);
// xm / xs / xh (optional): [G][C] tables -- the pooled tensor is relu((x - xm) * xs + xh), evaluated on the fly
int io_maxpool_fwd_t(const void* x, int N, int H, int W, int C, void* out, uint32_t* idx, hipStream_t st, int dt,
                     const float* xs = nullptr, const float* xh = nullptr, int G = 1, const float* xm = nullptr);
int io_maxpool_bwd_t(const void* dy, const uint32_t* idx, int N, int H, int W, int C, void* dx, hipStream_t st, int dt);
int io_avgpool_fc_fwd_t(const void* x, int N, int HW, int C, const float* w0, const float* b0, int K0,
                        const float* w1, const float* b1, int K1, float* pooled, float* logits, hipStream_t st,
                        int dt);
int io_avgpool_fc_bwd_t(const float* dlogits, const float* pooled, int N, int HW, int C, const float* w0, int K0,
                        const float* w1, int K1, const void* relu_mask, void* dx, float* dw0, float* db0, float* dw1,
                        float* db1, hipStream_t st, int dt);
int io_pack_planes_t(const float* const* planes, const long* sample_strides, int nplanes, int N, int H, int W,
                     void* out, hipStream_t st, int dt);
// fp32 master filter -> filter of storage type dt, either as is (transpose = 0) or as W^T [C][T][O]
int io_filter_prepare_t(const float* w, int O, int T, int C, void* dst, int transpose, hipStream_t st, int dt);
// exact-K stem: fp32 filter [O][T][8] -> [O][io_stem_kp(T, cr)] and the packed filter gradient back (pad channels 0)
int io_stem_pack_filter(const float* w, float* wp, int O, int T, int cr, hipStream_t st);
int io_stem_unpack_grad(const float* dwp, float* dw, int O, int T, int cr, hipStream_t st);
// tables of IoBwStats::xb_res == 2 from the forward tables of bn3 and the downsample BatchNorm ([G][C] each, training
// layout): a = scale3, b = scaled, c = shift3 - mean3 * scale3 + shiftd - meand * scaled; coef: 3*G*C floats (a | b | c)
int io_bn_resid2_tables(const float* mean3, const float* scale3, const float* shift3, const float* meand,
                        const float* scaled, const float* shiftd, int G, int C, float* coef, hipStream_t st);
constexpr int kIoStatTileRows = 128;   // row-tile height of the conv kernel = granule of fused BN statistics
int io_bn_finalize_tiles(float* tile_mean, float* tile_m2, int M, int C, int G, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         float* mean, float* rstd, float* scale, float* shift, hipStream_t st);
// dt_in: storage of `in` (the conv input), dt_dy: storage of dY; dW and the partials are fp32
int io_launch_conv_wgrad(const IoConvGeom& g, const void* in, const void* dy, float* dw, float* partial,
                         size_t partial_bytes, int stem, hipStream_t st, int dt_in = IO_F32, int dt_dy = IO_F32);
size_t io_conv_wgrad_partial_bytes(const IoConvGeom& g, int stem);

IoConvGeom io_geom_fwd(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad);
// stem.hip: the exact-K stem forward on whole 128-pixel output rows (row-persistent; statistics partials in the tile
// format of io_launch_conv_nt, or the inference epilogue relu(acc + bias))
bool io_stem_rows_ok(const IoConvGeom& g);
int io_launch_stem_rows(const IoConvGeom& g, const float* x8, const float* wp, float* out, float* st_mean, float* st_m2,
                        const float* bias, int relu, hipStream_t st);
int io_stem_wgrad_rows_max_blocks();
// BatchNorm backward of the stem's bn1 evaluated in the staging of its filter gradient (stem.hip): y = the raw conv output,
// a / b / c = the coefficient tables of io_bn_bwd_coefs_t, mean / scale / shift = the forward tables the ReLU mask is
// recomputed from; all [G][64]
struct IoStemXb {
    const float *y, *a, *b, *c, *mean, *scale, *shift;
    int G, tiles_per_group;
};
int io_launch_stem_wgrad_rows(const IoConvGeom& g, const float* x8, const float* dy, float* dwp, float* partial,
                              size_t partial_bytes, hipStream_t st, const IoStemXb* xb = nullptr);
// conv_halo3.hip: the bf16 stem's filter gradient (xb: bn1's backward folded in; y is bf16 there).  IO_OK / 1 = not its shape / < 0
size_t io_stem_wgrad_halo_partial_bytes();
bool io_stem_wgrad_halo_ok(const IoConvGeom& g, size_t partial_bytes, int G);
int io_launch_stem_wgrad_halo(const IoConvGeom& g, const void* x8, const void* dz, float* dw, float* partial, size_t partial_bytes,
                              hipStream_t st, const IoStemXb* xb);
// ... and the filter gradient of the bf16 3x3 stride-1 64 -> 64 layer on 64-wide maps
size_t io_wgrad_halo3_partial_bytes();
bool io_wgrad_halo3_shape(const IoConvGeom& g);
int io_launch_conv_wgrad_halo3(const IoConvGeom& g, const void* x, const void* dy, float* dw, float* partial, size_t partial_bytes,
                               hipStream_t st);
// dst[i] = sum over `splits` slabs of n4 float4, fixed order (conv_igemm.hip)
int io_splitk_reduce(const float* partial, float* dst, size_t n4, int splits, hipStream_t st);
// all filter transposes of a network in one launch (misc.hip): entry l = filter [O][T][C] at element offset off[l] of the
// parameter buffer; its transpose lands at the same element offset of wt_all
struct IoFilterTable {
    static constexpr int kMax = 64;
    int n;
    int start[kMax + 1];     // first 32 x 32 tile of entry l (start[n] = grid size)
    int O[kMax], T[kMax], C[kMax];
    unsigned off[kMax];
};
int io_filter_table_add(IoFilterTable& tab, long off, int O, int T, int C);
int io_filter_transpose_all(const IoFilterTable& tab, const float* params, void* wt_all, hipStream_t st, int dt);
IoConvGeom io_geom_dgrad(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int ph, int pw);
int io_run_dgrad(const void* dy, const void* wt, void* dx, const void* add, const void* mask, int N, int H,
                 int W, int Cin, int Cout, int R, int S, int stride, int pad, hipStream_t st,
                 const IoBwStats* bw = nullptr, int dt = IO_F32, int gw = 0);

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench / profiling) ----
enum IoProfClass {
    IO_PROF_CONV_NT128 = 0, IO_PROF_CONV_NT64, IO_PROF_CONV_STEM, IO_PROF_WGRAD, IO_PROF_WGRAD_STEM,
    IO_PROF_BN_STATS, IO_PROF_BN_APPLY, IO_PROF_BN_BWD, IO_PROF_POOL_HEAD, IO_PROF_TRANSPOSE, IO_PROF_PACK,
    IO_PROF_LOSS, IO_PROF_SGD, IO_PROF_CONV_WINO, IO_PROF_WGRAD_WINO, IO_PROF_NCLASS
};
struct IoProfScope {
    int idx;
    hipStream_t st;
    // flops_exec < 0: the launch executes its algorithmic flops
    IoProfScope(int cls, double flops, double bytes, hipStream_t stream, double flops_exec = -1.0);
    ~IoProfScope();
};
