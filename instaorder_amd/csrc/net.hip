// Whole-network executor for resnet50_cls (models/backbone/resnet_cls.py:121-222, 259-268):
// stem 7x7/2 -> BN -> ReLU -> maxpool -> Bottleneck x [3,4,6,3] (stride on the 3x3, downsample =
// strided 1x1 + BN on the first block of each stage) -> avgpool -> fc | (fc_occ, fc_depth).
//
// The executor owns no device memory: parameters, gradients, running statistics and the workspace
// arena are caller buffers; the plan (offsets into the arena) is a pure function of (N, S, mode),
// so forward and backward agree on it without shared state.  One forward = ~220 launches on one
// stream, no host synchronisation.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "io_common.h"

namespace {

constexpr int kMaxGroups = 8;
constexpr float kBnEps = 1e-5f, kBnMomentum = 0.1f;   // nn.BatchNorm2d defaults (resnet_cls.py:142)

struct ConvL {
    int cin, cout, k, stride, pad, cin_store;
    long w_off;   // float offset in params / grads
};
struct BnL {
    int C;
    long g_off, b_off, run_off;   // gamma, beta offsets in params; running_mean offset (var at +C)
    int index;
};
struct Block {
    ConvL c1, c2, c3, cd;
    BnL b1, b2, b3, bd;
    bool down;
    int stride, inC, planes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct io_net {
    int dtype = IO_F32;       // storage of activations / activation gradients / GEMM operands (IoDType)
    int in_ch, n_heads, head_dims[2];
    ConvL stem;
    BnL bn1;
    std::vector<Block> blocks;
    long fcw_off[2], fcb_off[2];
    long param_floats, running_floats;
    int n_bn;
    long bn_channels;
    std::vector<io_tensor_info> tensors;
};

namespace {

void add_tensor(io_net* net, const char* name, int kind, int ndim, const long* shape, long numel_storage,
                int cin_storage, long* off_out, int bn_index = -1, long run_off = -1) {
    io_tensor_info t;
    memset(&t, 0, sizeof(t));
    snprintf(t.name, sizeof(t.name), "%s", name);
    t.kind = kind;
    t.ndim = ndim;
    for (int i = 0; i < ndim; ++i) t.shape[i] = shape[i];
    t.offset = net->param_floats;
    t.numel_storage = numel_storage;
    t.cin_storage = cin_storage;
    t.bn_index = bn_index;
    t.running_offset = run_off;
    *off_out = net->param_floats;
    net->param_floats = (long)align_up((size_t)(net->param_floats + numel_storage), 64);
    net->tensors.push_back(t);
}

ConvL add_conv(io_net* net, const char* name, int cout, int cin, int k, int stride, int pad) {
    ConvL c;
    c.cin = cin; c.cout = cout; c.k = k; c.stride = stride; c.pad = pad;
    c.cin_store = (cin % 32 == 0) ? cin : 8;
    char nm[64];
    snprintf(nm, sizeof(nm), "%s.weight", name);
    const long shape[4] = {cout, cin, k, k};
    add_tensor(net, nm, 0, 4, shape, (long)cout * k * k * c.cin_store, c.cin_store, &c.w_off);
    return c;
}

BnL add_bn(io_net* net, const char* name, int C) {
    BnL b;
    b.C = C;
    b.index = net->n_bn++;
    b.run_off = net->running_floats;
    net->running_floats += 2L * C;
    net->bn_channels += C;
    char nm[64];
    const long shape[1] = {C};
    snprintf(nm, sizeof(nm), "%s.weight", name);
    add_tensor(net, nm, 1, 1, shape, C, C, &b.g_off, b.index, b.run_off);
    snprintf(nm, sizeof(nm), "%s.bias", name);
    add_tensor(net, nm, 2, 1, shape, C, C, &b.b_off, b.index, b.run_off);
    return b;
}

// ---- workspace plan ---------------------------------------------------------------------------
constexpr size_t kNoBuf = ~(size_t)0;
struct BlockBufs {
    size_t y1, a1, y2, a2, y3, yd, out;   // byte offsets (a1 / a2 = kNoBuf: never materialised, see fuse_in)
    size_t bits;                          // training: [out > 0] as one bit per element (IoBwStats::maskbits), kNoBuf: none
};
struct Plan {
    size_t y0, a0, p0, idx0, pooled;
    std::vector<BlockBufs> blk;
    size_t tables;       // per-BN [4][kMaxGroups][C] floats, BN i at tables + 4*kMaxGroups*chan_prefix[i]
    size_t bn_partial, bn_partial_floats, coef;
    size_t tile_mean, tile_m2;   // fused-statistics partials written by the conv epilogue (training)
    size_t gbuf[5];      // gradient scratch (training only)
    size_t aside;        // fp32 training: relu(bn(y)) of ONE layer at a time, rebuilt by the data-gradient epilogue for
                         // the filter gradient that wants it (see fuse_in)
    size_t wt, wg_partial, wg_partial_bytes;
    size_t wop;          // bf16 mode: operand copy of the whole flat parameter buffer (same element offsets)
    size_t stem_wp, stem_dwp;   // fp32: exact-K stem filter / filter gradient, [64][io_stem_kp]
    size_t wt_all;              // transposed copies of all filters (kNoBuf: one transpose launch per data gradient)
    size_t wfold, fbias; // eval: filters with the BatchNorm scale folded in (storage type, parameter offsets) + biases
    size_t wino_u;       // fp32: transformed filters of the 3x3 launch in flight (Winograd row form), 18 * 512 * 512 floats
    size_t total;
};

struct Arena {
    size_t top = 0;
    size_t take(size_t bytes) {
        const size_t o = top;
        top = align_up(top + bytes, 256);
        return o;
    }
};

// SW: input width when it differs from the height S (eval only: the 'orig' inference mode feeds whole images at their
// own aspect ratio); 0 = square
Plan make_plan(const io_net* net, int N, int S, bool training, int SW = 0) {
    Plan p;
    Arena a;
    if (SW <= 0) SW = S;
    const int H0 = S / 2, H1 = S / 4, W0 = SW / 2;
    const size_t f = sizeof(float);
    const size_t e = (size_t)io_dtype_bytes(net->dtype);   // activation element size
    const size_t maxact = (size_t)N * H0 * W0 * 64 * e;   // == N*H1*W1*256*e, the largest activations
    p.wop = net->dtype == IO_BF16 ? a.take((size_t)net->param_floats * sizeof(bf16_t)) : 0;
    {
        const size_t sb = (size_t)64 * io_stem_kp(49, net->in_ch) * f;
        p.stem_wp = a.take(sb);
        p.stem_dwp = a.take(sb);
    }
    p.wfold = training ? 0 : a.take((size_t)net->param_floats * e);
    p.fbias = training ? 0 : a.take((size_t)net->bn_channels * f);
    p.wino_u = net->dtype == IO_F32 ? a.take((size_t)18 * 512 * 512 * f) : kNoBuf;
    p.tables = a.take((size_t)4 * kMaxGroups * net->bn_channels * f);
    p.bn_partial_floats = (size_t)3 * 1100 * 2048;
    p.bn_partial = a.take(p.bn_partial_floats * f);
    p.coef = a.take((size_t)3 * kMaxGroups * 2048 * f);
    p.pooled = a.take((size_t)N * 2048 * f);
    {
        // per (128-row tile, channel) partials of the widest conv output + room for the level-1 merge
        const size_t tiles_x_c = (size_t)N * H0 * W0 * 64 / kIoStatTileRows + (size_t)kMaxGroups * 2048;
        const size_t fl = training ? tiles_x_c + tiles_x_c / 32 + (size_t)kMaxGroups * 2048 : 0;
        p.tile_mean = a.take(fl * f);
        p.tile_m2 = a.take(fl * f);
    }
    p.blk.resize(net->blocks.size());
    if (training) {
        p.y0 = a.take(maxact);
        p.a0 = kNoBuf;          // relu(bn1(conv1)) is evaluated inside the max-pooling kernel, never stored
        p.p0 = a.take((size_t)N * H1 * H1 * 64 * e);
        p.idx0 = a.take((size_t)N * H1 * H1 * 16 * sizeof(uint32_t));
        int H = H1;
        for (size_t i = 0; i < net->blocks.size(); ++i) {
            const Block& b = net->blocks[i];
            const int Ho = H / b.stride;
            BlockBufs& bb = p.blk[i];
            // fp32: relu(bn1(y1)) / relu(bn2(y2)) are applied to the operand of conv2 / conv3 (and of their filter
            // gradients) as it is staged; the tensors exist only where that cannot run -- bf16 mode (the VALU work
            // does not hide under the 16x faster MFMA: measured) and shapes whose BatchNorm groups do not
            // fill whole 128-row tiles for some group count G | 8
            const long Mo = (long)N * Ho * Ho;
            const bool never = net->dtype == IO_F32 && Mo % kMaxGroups == 0 && (Mo / kMaxGroups) % kIoStatTileRows == 0;
            bb.y1 = a.take((size_t)N * H * H * b.planes * e);
            bb.a1 = (never && b.stride == 1) ? kNoBuf : a.take((size_t)N * H * H * b.planes * e);
            bb.y2 = a.take((size_t)N * Ho * Ho * b.planes * e);
            bb.a2 = never ? kNoBuf : a.take((size_t)N * Ho * Ho * b.planes * e);
            bb.y3 = a.take((size_t)N * Ho * Ho * b.planes * 4 * e);
            bb.yd = b.down ? a.take((size_t)N * Ho * Ho * b.planes * 4 * e) : 0;
            bb.out = a.take((size_t)N * Ho * Ho * b.planes * 4 * e);
            // bf16: the ReLU mask of the block output as one bit per element, for the data gradient that completes d(out) on
            // the 256-row kernel (conv_p256.hip): 1/16 of the tensor it would otherwise read.  Written by the BatchNorm pass
            // that builds the output; outputs built inside the next conv1 on conv_nt_kernel (xr_route 1) have none.  fp32 never: measured round 5 --
            // packing the bits in conv_nt_kernel's operand staging and reading them in its column-layout epilogue cost the
            // fp32 step 3 % (59.7 -> 64.6 ms in the 128-wide class) for 4 of 136 bytes per element saved.
            bb.bits = (net->dtype == IO_BF16 && Mo % 256 == 0) ? a.take((size_t)Mo * b.planes * 4 / 8) : kNoBuf;
            H = Ho;
        }
        for (int i = 0; i < 5; ++i) p.gbuf[i] = a.take(maxact);
        p.aside = net->dtype == IO_F32 ? a.take(maxact / 2) : kNoBuf;    // the largest a1 / a2: N x (S/4)^2 x 128 (layer2.0)
        // filter-gradient split-K partials: largest over all convs
        size_t wg = 0, wtmax = 0;
        {
            IoConvGeom g = io_geom_fwd(N, S, S, 8, 64, 7, 7, 2, 3);
            if (net->dtype == IO_F32) g.cr = net->in_ch;
            wg = io_conv_wgrad_partial_bytes(g, 1);
        }
        H = H1;
        for (const Block& b : net->blocks) {
            const ConvL* cs[4] = {&b.c1, &b.c2, &b.c3, b.down ? &b.cd : nullptr};
            for (int j = 0; j < 4; ++j) {
                if (!cs[j]) continue;
                const ConvL& c = *cs[j];
                const int Hin = (j == 2) ? H / b.stride : H;
                IoConvGeom g = io_geom_fwd(N, Hin, Hin, c.cin, c.cout, c.k, c.k, c.stride, c.pad);
                const size_t need = io_conv_wgrad_partial_bytes(g, 0);
                if (need > wg) wg = need;
                const size_t wsz = (size_t)c.cout * c.k * c.k * c.cin * e;
                if (wsz > wtmax) wtmax = wsz;
            }
            H /= b.stride;
        }
        p.wg_partial_bytes = wg;
        p.wg_partial = a.take(wg);
        p.wt = a.take(wtmax);
        // every filter's transpose (the B operand of its data gradient) at the filter's own offset: one launch per
        // backward pass instead of one per layer
        p.wt_all = a.take((size_t)net->param_floats * e);
    } else {
        // eval: rotating buffers (block input / a1 / a2 / y / yd / output)
        const size_t r0 = a.take(maxact), r1 = a.take(maxact), r2 = a.take(maxact), r3 = a.take(maxact),
                     r4 = a.take(maxact), r5 = a.take(maxact);
        p.y0 = r3; p.a0 = r1; p.p0 = r0; p.idx0 = 0;
        size_t xin = r0, xout = r5;
        for (size_t i = 0; i < net->blocks.size(); ++i) {
            BlockBufs& bb = p.blk[i];
            bb.y1 = r3; bb.a1 = r1; bb.y2 = r3; bb.a2 = r2; bb.y3 = r3; bb.yd = r4; bb.out = xout; bb.bits = kNoBuf;
            const size_t t = xin; xin = xout; xout = t;
        }
        for (int i = 0; i < 5; ++i) p.gbuf[i] = 0;
        p.aside = kNoBuf;
        p.wg_partial = p.wt = 0;
        p.wt_all = kNoBuf;
        p.wg_partial_bytes = 0;
    }
    p.total = a.top;
    return p;
}

// ---- eval mode: BatchNorm folded into the convolutions -----------------------------------------------
// y = gamma * (conv(x, w) - mean) / sqrt(var + eps) + beta = conv(x, w * s) + (beta - mean * s), s = gamma / sqrt(var+eps):
// one launch rewrites every filter (scaled per output channel, in the activation storage type) and every bias, then
// each convolution applies bias (+ residual) (+ ReLU) in its epilogue -- an inference forward has no BatchNorm pass.
struct FoldDesc {
    long w_off, g_off, b_off, run_off, bias_off;
    int rows, rowlen;
};
struct FoldTable {
    int n;
    FoldDesc d[56];
};
template <typename T>
__global__ __launch_bounds__(256) void fold_bn_kernel(FoldTable t, const float* __restrict__ params,
                                                     const float* __restrict__ running, float eps,
                                                     T* __restrict__ wout, float* __restrict__ bias) {
    const FoldDesc d = t.d[blockIdx.y];
    for (int row = blockIdx.x; row < d.rows; row += gridDim.x) {
        const float s = params[d.g_off + row] / sqrtf(running[d.run_off + d.rows + row] + eps);
        if (threadIdx.x == 0) bias[d.bias_off + row] = params[d.b_off + row] - running[d.run_off + row] * s;
        const float* src = params + d.w_off + (size_t)row * d.rowlen;
        T* dst = wout + d.w_off + (size_t)row * d.rowlen;
        for (int k = threadIdx.x; k < d.rowlen; k += 256) {
            if constexpr (sizeof(T) == 2) dst[k] = io_f2bf(src[k] * s);
            else dst[k] = src[k] * s;
        }
    }
}

#define IO_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_) return rc_;    \
    } while (0)

struct Tables {
    float *mean, *rstd, *scale, *shift;
};

struct Ctx {
    const io_net* net;
    const float* params;
    float* running;
    float* grads;
    char* ws;
    Plan plan;
    int N, S, G;
    int SW;                 // input width (== S except for an eval forward on a non-square input)
    bool training;
    hipStream_t st;
    std::vector<long> chan_prefix;

    float* buf(size_t off) const { return reinterpret_cast<float*>(ws + off); }   // fp32 scratch / tables
    void* act(size_t off) const { return ws + off; }                               // activation-typed tensors
    int dt() const { return net->dtype; }
    // GEMM operand view of a filter: the fp32 master itself, or its bf16 copy made at the start of the pass
    const void* wop(long w_off) const {
        return net->dtype == IO_BF16 ? (const void*)(reinterpret_cast<const bf16_t*>(ws + plan.wop) + w_off)
                                     : (const void*)(params + w_off);
    }
    const void* wfold(long w_off) const {      // eval: folded filter in the activation storage type
        return net->dtype == IO_BF16 ? (const void*)(reinterpret_cast<const bf16_t*>(ws + plan.wfold) + w_off)
                                     : (const void*)(reinterpret_cast<const float*>(ws + plan.wfold) + w_off);
    }
    float* fbias(const BnL& b) const { return buf(plan.fbias) + chan_prefix[b.index]; }
    Tables tables(const BnL& b) const {
        float* base = buf(plan.tables) + (size_t)4 * kMaxGroups * chan_prefix[b.index];
        const size_t gs = (size_t)kMaxGroups * b.C;
        return Tables{base, base + gs, base + 2 * gs, base + 3 * gs};
    }
};

// fp32 stem: the reduction runs over the real input channels only (exact-K mode of the conv kernels); the filter is
// re-packed per pass (64 x 256 floats)
bool stem_exact(const Ctx& c, const ConvL& L) { return L.cin_store == 8 && c.net->dtype == IO_F32; }

// xf: the BatchNorm whose (training) scale / shift tables + ReLU are applied to x while it is staged
// xr: x is the raw conv3 output of the PREVIOUS block and xr its bn3: the operand is that block's output relu(bn3(x) +
// xr_id), evaluated while it is staged and written to xr_out (IoBwStats::xb_res)
int conv_fwd(const Ctx& c, const ConvL& L, const void* x, void* y, int H, bool stats = false, const BnL* xf = nullptr,
             int Mout = 0, const BnL* xr = nullptr, const void* xr_id = nullptr, void* xr_out = nullptr,
             bool xr_two = false, uint32_t* xr_bits = nullptr) {
    IoConvGeom g = io_geom_fwd(c.N, H, H, L.cin_store, L.cout, L.k, L.k, L.stride, L.pad);
    const bool stem = L.cin_store == 8;      // the packed input x8 has the net's storage type too
    const void* w = c.wop(L.w_off);
    if (stem_exact(c, L)) {
        g.cr = L.cin;
        IO_TRY(io_stem_pack_filter(c.params + L.w_off, c.buf(c.plan.stem_wp), L.cout, L.k * L.k, L.cin, c.st));
        w = c.buf(c.plan.stem_wp);
    }
    IoBwStats ep{};
    if (xf) {
        Tables t = c.tables(*xf);
        ep.in_mean = t.mean;
        ep.in_scale = t.scale;
        ep.in_shift = t.shift;
        ep.in_Mg = Mout / c.G;
    }
    // 3x3 stride-1 convolutions in fp32: scratch for the Winograd row form (the launcher falls back where it does not apply)
    const bool wino = c.plan.wino_u != kNoBuf && L.k == 3 && L.stride == 1 && c.dt() == IO_F32;
    if (wino) ep.wino_u = c.buf(c.plan.wino_u);
    if (xr) {
        Tables t = c.tables(*xr);
        ep.xb_y = xr_id;
        ep.xb_a = t.scale;
        ep.xb_b = t.mean;
        ep.xb_c = t.shift;
        ep.xb_out = xr_out;
        ep.xb_Mg = Mout / c.G;       // (a 1x1 stride-1 convolution: operand rows = output rows)
        ep.xb_res = 1;
        ep.xb_bits = xr_bits;
        if (xr_two) {                // previous block with a downsample branch: the folded tables wait in plan.coef
            const size_t gs = (size_t)c.G * xr->C;
            ep.xb_a = c.buf(c.plan.coef);
            ep.xb_b = c.buf(c.plan.coef) + gs;
            ep.xb_c = c.buf(c.plan.coef) + 2 * gs;
            ep.xb_res = 2;
        }
    }
    return io_launch_conv_nt(g, x, w, y, nullptr, nullptr, stem, c.st,
                             stats ? c.buf(c.plan.tile_mean) : nullptr, stats ? c.buf(c.plan.tile_m2) : nullptr,
                             (xf || xr || wino) ? &ep : nullptr, c.dt(), c.dt());
}

// BN statistics (training) or table preparation (eval) for y[M][C]
int bn_prepare(const Ctx& c, const BnL& b, const void* y, int M, bool from_tiles = false) {
    Tables t = c.tables(b);
    if (c.training && from_tiles)
        return io_bn_finalize_tiles(c.buf(c.plan.tile_mean), c.buf(c.plan.tile_m2), M, b.C, c.G,
                                    c.params + b.g_off, c.params + b.b_off, c.running + b.run_off,
                                    c.running + b.run_off + b.C, kBnMomentum, kBnEps, t.mean, t.rstd, t.scale,
                                    t.shift, c.st);
    if (c.training)
        return io_bn_stats_finalize_t(y, M, b.C, c.G, c.params + b.g_off, c.params + b.b_off, c.running + b.run_off,
                                      c.running + b.run_off + b.C, kBnMomentum, kBnEps, t.mean, t.rstd, t.scale,
                                      t.shift, c.buf(c.plan.bn_partial), c.plan.bn_partial_floats, c.st, c.dt());
    return io_bn_eval_prepare(b.C, c.params + b.g_off, c.params + b.b_off, c.running + b.run_off,
                              c.running + b.run_off + b.C, kBnEps, t.mean, t.scale, t.shift, c.st);
}

// conv followed by the statistics of its output; the statistics ride in the conv epilogue whenever a
// 128-row tile never straddles two BN groups
int conv_bn(const Ctx& c, const ConvL& L, const BnL& b, const void* x, void* y, int Hin, int Mout,
            const BnL* xf = nullptr, const BnL* xr = nullptr, const void* xr_id = nullptr, void* xr_out = nullptr,
            bool xr_two = false, uint32_t* xr_bits = nullptr) {
    const bool fuse = c.training && (Mout / c.G) % kIoStatTileRows == 0;
    IO_TRY(conv_fwd(c, L, x, y, Hin, fuse, xf, Mout, xr, xr_id, xr_out, xr_two, xr_bits));
    return bn_prepare(c, b, y, Mout, fuse);
}

// Does conv2 / conv3 of a block with Mout output rows read its input through the producer's BatchNorm + ReLU instead of
// from a stored activation?  fp32 with whole 128-row tiles per BatchNorm group; the forward then never writes
// relu(bn(y)), and the backward gets it back for the one filter gradient that needs it as a side output of the
// data-gradient launch that recomputes the ReLU mask from y anyway (IoBwStats::a_out).  (Transforming the operand of the
// filter-gradient kernel the same way was measured: +20..35 % on the 128 x 128 tiles, which have no registers to spare.)
// Is the output of a block with Mout rows -- out = relu(bn3(y3) + identity) -- built by the NEXT block's conv1 while it
// stages its operand (conv_fwd's xr) instead of by a pass of its own?  fp32 (there the transform hides under the MFMAs),
// whole 128-row tiles per group; the caller also needs an identity that is a plain tensor (no downsample branch) and a
// next block.  Like the backward form: a 12 B / element pass becomes 8 B / element inside a GEMM.
#ifndef IO_XB_BF16_MAXP
#define IO_XB_BF16_MAXP 64
#endif
#ifndef IO_XB_BF16_L1
#define IO_XB_BF16_L1 1   // bf16: the operand forms on the layer-1 (64-plane) blocks only
#endif
// Who builds the output of block i -- out = relu(bn3(y3) + identity)?  0: a BatchNorm pass of its own (bn_act); 1: the NEXT
// block's conv1 on conv_nt_kernel's staging registers (fp32; bf16 only on the 64-plane blocks of layer 1 -- the form is VALU-bound
// next to bf16 MFMAs, but those launches are so HBM-bound that dropping the pass still wins, as for the backward form, run_backward);
// 2 (bf16, round 6): the next block's conv1 on the 256-row kernel, the transform applied IN LDS to every A k-tile after its DMA
// has landed (conv_p256.hip XOP) -- that launch also writes the one-bit mask of the output.
int out_rows(const Ctx& c, size_t i) {
    int H = c.S / 4;
    for (size_t k = 0; k <= i; ++k) H /= c.net->blocks[k].stride;
    return c.N * H * H;
}
int xr_route(const Ctx& c, size_t i) {
    if (!c.training || i + 1 >= c.net->blocks.size()) return 0;
    const Block& b = c.net->blocks[i];
    const int Mout = out_rows(c, i);
    if (Mout % c.G != 0 || (Mout / c.G) % kIoStatTileRows != 0) return 0;
    if (c.net->dtype == IO_F32) return 1;
    if (IO_XB_BF16_L1 && b.planes <= IO_XB_BF16_MAXP) return 1;
    return io_conv_p256_takes_xop(Mout, b.planes * 4, c.net->blocks[i + 1].planes, Mout / c.G) ? 2 : 0;
}

// the one-bit ReLU mask of block i's output (training plans; nullptr where the plan has none)
uint32_t* bits_of(const Ctx& c, size_t i) {
    const size_t off = c.plan.blk[i].bits;
    return (c.training && off != kNoBuf) ? reinterpret_cast<uint32_t*>(c.ws + off) : nullptr;
}

// ... and whether the forward wrote it: block outputs built by the next block's conv1 on conv_nt_kernel have no launch that does
bool bits_written(const Ctx& c, size_t i) {
    return c.plan.blk[i].bits != kNoBuf && xr_route(c, i) != 1;
}

bool fuse_in(const Ctx& c, int Mout) {
    return c.training && c.net->dtype == IO_F32 && Mout % c.G == 0 && (Mout / c.G) % kIoStatTileRows == 0;
}

int bn_act(const Ctx& c, const BnL& b, const void* y, int M, const void* idt, const BnL* b2, int relu,
           void* out, uint32_t* bits = nullptr) {
    Tables t = c.tables(b);
    const float *m2 = nullptr, *s2 = nullptr, *h2 = nullptr;
    if (b2) {
        Tables t2 = c.tables(*b2);
        m2 = t2.mean;
        s2 = t2.scale;
        h2 = t2.shift;
    }
    // tables are laid out with a group stride of C (training) -- eval uses one shared row
    return io_bn_apply_t(y, M, b.C, c.training ? c.G : 1, c.training ? 1 : 0, t.mean, t.scale, t.shift, idt, m2, s2,
                         h2, relu, out, c.st, c.dt(), bits);
}

// conv + folded BatchNorm (+ residual) (+ ReLU) of an inference forward
int conv_folded(const Ctx& c, const ConvL& L, const BnL& b, const void* x, void* y, int H, int W, const void* add,
                int relu) {
    IoConvGeom g = io_geom_fwd(c.N, H, W, L.cin_store, L.cout, L.k, L.k, L.stride, L.pad);
    IoBwStats ep{};
    ep.bias = c.fbias(b);
    ep.relu = relu;
    if (c.plan.wino_u != kNoBuf && L.k == 3 && L.stride == 1 && c.dt() == IO_F32 && !add)
        ep.wino_u = c.buf(c.plan.wino_u);      // (square or H x W inputs alike; odd widths fall back in the launcher)
    const void* w = c.wfold(L.w_off);
    if (stem_exact(c, L)) {
        g.cr = L.cin;
        IO_TRY(io_stem_pack_filter((const float*)w, c.buf(c.plan.stem_wp), L.cout, L.k * L.k, L.cin, c.st));
        w = c.buf(c.plan.stem_wp);
    }
    return io_launch_conv_nt(g, x, w, y, add, nullptr, L.cin_store == 8, c.st, nullptr, nullptr, &ep,
                             c.dt(), c.dt());
}

int run_forward_eval(Ctx& c, const void* x8, float* logits) {
    const io_net* net = c.net;
    const Plan& p = c.plan;
    const int H0 = c.S / 2, H1 = c.S / 4;
    {
        FoldTable t;
        t.n = 0;
        auto add = [&](const ConvL& L, const BnL& b) {
            FoldDesc& d = t.d[t.n++];
            d.w_off = L.w_off; d.g_off = b.g_off; d.b_off = b.b_off; d.run_off = b.run_off;
            d.bias_off = c.chan_prefix[b.index];
            d.rows = L.cout; d.rowlen = L.k * L.k * L.cin_store;
        };
        add(net->stem, net->bn1);
        for (const Block& b : net->blocks) {
            add(b.c1, b.b1); add(b.c2, b.b2); add(b.c3, b.b3);
            if (b.down) add(b.cd, b.bd);
        }
        IO_REQUIRE(t.n <= 56, IO_ERR_STATE, "fold table overflow");
        IoProfScope prof(IO_PROF_TRANSPOSE, 0.0, (4.0 + io_dtype_bytes(c.dt())) * (double)net->param_floats, c.st);
        if (c.dt() == IO_BF16)
            hipLaunchKernelGGL(fold_bn_kernel<bf16_t>, dim3(64, t.n), dim3(256), 0, c.st, t, c.params, c.running, kBnEps,
                               reinterpret_cast<bf16_t*>(c.ws + p.wfold), c.buf(p.fbias));
        else
            hipLaunchKernelGGL(fold_bn_kernel<float>, dim3(64, t.n), dim3(256), 0, c.st, t, c.params, c.running, kBnEps,
                               reinterpret_cast<float*>(c.ws + p.wfold), c.buf(p.fbias));
        IO_TRY(io_check_launch("fold_bn"));
    }
    const int W0 = c.SW / 2;
    IO_TRY(conv_folded(c, net->stem, net->bn1, x8, c.act(p.a0), c.S, c.SW, nullptr, 1));
    IO_TRY(io_maxpool_fwd_t(c.act(p.a0), c.N, H0, W0, 64, c.act(p.p0), nullptr, c.st, c.dt()));
    const void* x = c.act(p.p0);
    int H = H1, W = c.SW / 4;
    for (size_t i = 0; i < net->blocks.size(); ++i) {
        const Block& b = net->blocks[i];
        const BlockBufs& bb = p.blk[i];
        const int Ho = H / b.stride, Wo = W / b.stride;
        IO_TRY(conv_folded(c, b.c1, b.b1, x, c.act(bb.a1), H, W, nullptr, 1));
        IO_TRY(conv_folded(c, b.c2, b.b2, c.act(bb.a1), c.act(bb.a2), H, W, nullptr, 1));
        const void* identity = x;
        if (b.down) {
            IO_TRY(conv_folded(c, b.cd, b.bd, x, c.act(bb.yd), H, W, nullptr, 0));
            identity = c.act(bb.yd);
        }
        IO_TRY(conv_folded(c, b.c3, b.b3, c.act(bb.a2), c.act(bb.out), Ho, Wo, identity, 1));
        x = c.act(bb.out);
        H = Ho;
        W = Wo;
    }
    const float* w1 = net->n_heads > 1 ? c.params + net->fcw_off[1] : nullptr;
    const float* b1 = net->n_heads > 1 ? c.params + net->fcb_off[1] : nullptr;
    return io_avgpool_fc_fwd_t(x, c.N, H * W, 2048, c.params + net->fcw_off[0], c.params + net->fcb_off[0],
                               net->head_dims[0], w1, b1, net->n_heads > 1 ? net->head_dims[1] : 0, c.buf(p.pooled),
                               logits, c.st, c.dt());
}

int run_forward(Ctx& c, const void* x8, float* logits) {
    if (!c.training) return run_forward_eval(c, x8, logits);
    const io_net* net = c.net;
    const Plan& p = c.plan;
    const int H0 = c.S / 2, H1 = c.S / 4;
    if (c.dt() == IO_BF16)      // bf16 GEMM operands: one cast of the whole flat fp32 parameter buffer
        IO_TRY(io_filter_prepare_t(c.params, 1, 1, (int)net->param_floats, c.act(p.wop), 0, c.st, IO_BF16));
    // stem
    IO_TRY(conv_bn(c, net->stem, net->bn1, x8, c.act(p.y0), c.S, c.N * H0 * H0));
    {
        // relu(bn1(.)) inside the pooling kernel: the 2.1 GB activation of the bench batch is neither written nor re-read
        Tables t = c.tables(net->bn1);
        IO_TRY(io_maxpool_fwd_t(c.act(p.y0), c.N, H0, H0, 64, c.act(p.p0), reinterpret_cast<uint32_t*>(c.ws + p.idx0),
                                c.st, c.dt(), t.scale, t.shift, c.G, t.mean));
    }
    const void* x = c.act(p.p0);
    int H = H1;
    // a block output whose construction was left to the next block's conv1 (xr_route): bn3, y3, identity of that block
    const BnL* pend_bn = nullptr;
    const void *pend_y3 = nullptr, *pend_id = nullptr;
    bool pend_two = false;        // ... of a block with a downsample branch: identity = bnd(yd), tables folded into plan.coef
    for (size_t i = 0; i < net->blocks.size(); ++i) {
        const Block& b = net->blocks[i];
        const BlockBufs& bb = p.blk[i];
        const int Ho = H / b.stride;
        const int Min = c.N * H * H, Mout = c.N * Ho * Ho;
        if (pend_bn) {
            // x = the previous block's output does not exist yet: conv1 evaluates it from (y3, identity) on its operand
            // and writes it out (first output-channel tile); everything below that reads x comes after this launch
            IO_TRY(conv_bn(c, b.c1, b.b1, pend_y3, c.act(bb.y1), H, Min, nullptr, pend_bn, pend_id,
                           c.act(p.blk[i - 1].out), pend_two, xr_route(c, i - 1) == 2 ? bits_of(c, i - 1) : nullptr));
            pend_bn = nullptr;
        } else {
            IO_TRY(conv_bn(c, b.c1, b.b1, x, c.act(bb.y1), H, Min));
        }
        // conv2 reads relu(bn1(y1)), conv3 reads relu(bn2(y2)): through the input transform straight from y1 / y2, or
        // from a stored activation (the strided conv2 of a stage's first block keeps a1: its data gradient runs as
        // parity classes and cannot rebuild it)
        const bool f3 = fuse_in(c, Mout), f2 = f3 && b.stride == 1;
        IO_REQUIRE((f2 || bb.a1 != kNoBuf) && (f3 || bb.a2 != kNoBuf), IO_ERR_SHAPE,
                   "io_net_forward: G=%d does not divide this batch into whole 128-row tiles (use G in 1, 2, 4, 8)", c.G);
        if (f2) {
            IO_TRY(conv_bn(c, b.c2, b.b2, c.act(bb.y1), c.act(bb.y2), H, Mout, &b.b1));
        } else {
            IO_TRY(bn_act(c, b.b1, c.act(bb.y1), Min, nullptr, nullptr, 1, c.act(bb.a1)));
            IO_TRY(conv_bn(c, b.c2, b.b2, c.act(bb.a1), c.act(bb.y2), H, Mout));
        }
        if (f3) {
            IO_TRY(conv_bn(c, b.c3, b.b3, c.act(bb.y2), c.act(bb.y3), Ho, Mout, &b.b2));
        } else {
            IO_TRY(bn_act(c, b.b2, c.act(bb.y2), Mout, nullptr, nullptr, 1, c.act(bb.a2)));
            IO_TRY(conv_bn(c, b.c3, b.b3, c.act(bb.a2), c.act(bb.y3), Ho, Mout));
        }
        if (b.down) {
            IO_TRY(conv_bn(c, b.cd, b.bd, x, c.act(bb.yd), H, Mout));
            if (xr_route(c, i)) {
                // relu(bn3(y3) + bnd(yd)) = relu(a * y3 + b * yd + c): one table set, then as below
                Tables t3 = c.tables(b.b3), td = c.tables(b.bd);
                IO_TRY(io_bn_resid2_tables(t3.mean, t3.scale, t3.shift, td.mean, td.scale, td.shift, c.G, b.b3.C,
                                           c.buf(c.plan.coef), c.st));
                pend_bn = &b.b3;
                pend_y3 = c.act(bb.y3);
                pend_id = c.act(bb.yd);
                pend_two = true;
            } else {
                IO_TRY(bn_act(c, b.b3, c.act(bb.y3), Mout, c.act(bb.yd), &b.bd, 1, c.act(bb.out), bits_of(c, i)));
            }
        } else if (xr_route(c, i)) {
            pend_bn = &b.b3;                 // built by the next block's conv1
            pend_y3 = c.act(bb.y3);
            pend_id = x;
            pend_two = false;
        } else {
            IO_TRY(bn_act(c, b.b3, c.act(bb.y3), Mout, x, nullptr, 1, c.act(bb.out), bits_of(c, i)));
        }
        x = c.act(bb.out);
        H = Ho;
    }
    const float* w1 = net->n_heads > 1 ? c.params + net->fcw_off[1] : nullptr;
    const float* b1 = net->n_heads > 1 ? c.params + net->fcb_off[1] : nullptr;
    IO_TRY(io_avgpool_fc_fwd_t(x, c.N, H * H, 2048, c.params + net->fcw_off[0], c.params + net->fcb_off[0],
                               net->head_dims[0], w1, b1, net->n_heads > 1 ? net->head_dims[1] : 0, c.buf(p.pooled),
                               logits, c.st, c.dt()));
    return IO_OK;
}

// mask: 0 = no ReLU behind this BN, 1 = recompute relu(bn(y)) > 0 from y, 2 = read the stored activation
int bn_back(const Ctx& c, const BnL& b, const void* dout, int mask, const void* act, const void* y, int M,
            void* dy, void* dz_out) {
    Tables t = c.tables(b);
    return io_bn_bwd_t(dout, mask == 2 ? act : nullptr, mask == 1 ? t.scale : nullptr, mask == 1 ? t.shift : nullptr,
                       y, M, b.C, c.G, c.params + b.g_off, t.mean, t.rstd, c.grads + b.g_off, c.grads + b.b_off, dy,
                       dz_out, c.buf(c.plan.bn_partial), c.plan.bn_partial_floats, c.buf(c.plan.coef), c.st, c.dt());
}

int conv_wgrad(const Ctx& c, const ConvL& L, const void* x, const void* dy, int H) {
    IoConvGeom g = io_geom_fwd(c.N, H, H, L.cin_store, L.cout, L.k, L.k, L.stride, L.pad);
    const bool stem = L.cin_store == 8;
    if (stem_exact(c, L)) {
        g.cr = L.cin;
        IO_TRY(io_launch_conv_wgrad(g, x, dy, c.buf(c.plan.stem_dwp), c.buf(c.plan.wg_partial), c.plan.wg_partial_bytes,
                                    1, c.st, c.dt(), c.dt()));
        return io_stem_unpack_grad(c.buf(c.plan.stem_dwp), c.grads + L.w_off, L.cout, L.k * L.k, L.cin, c.st);
    }
    return io_launch_conv_wgrad(g, x, dy, c.grads + L.w_off, c.buf(c.plan.wg_partial), c.plan.wg_partial_bytes, stem,
                                c.st, c.dt(), c.dt());
}

// the transposes of ALL filters (io_net_backward's first stage; the staged calls of one pass share the workspace)
int transpose_filters(const Ctx& c) {
    if (c.plan.wt_all == kNoBuf) return IO_OK;
    IoFilterTable tab;
    tab.n = 0;
    tab.start[0] = 0;
    for (const Block& b : c.net->blocks) {
        const ConvL* cs[4] = {&b.c1, &b.c2, &b.c3, b.down ? &b.cd : nullptr};
        for (int j = 0; j < 4; ++j)
            if (cs[j]) IO_TRY(io_filter_table_add(tab, cs[j]->w_off, cs[j]->cout, cs[j]->k * cs[j]->k, cs[j]->cin));
    }
    return io_filter_transpose_all(tab, c.params, c.act(c.plan.wt_all), c.st, c.dt());
}

int conv_dgrad(const Ctx& c, const ConvL& L, const void* dy, void* dx, const void* add, const void* mask,
               int H, const IoBwStats* bw = nullptr) {
    void* wt = c.act(c.plan.wt);
    if (c.plan.wt_all != kNoBuf)
        wt = (char*)c.act(c.plan.wt_all) + (size_t)L.w_off * io_dtype_bytes(c.dt());
    else
        IO_TRY(io_filter_prepare_t(c.params + L.w_off, L.cout, L.k * L.k, L.cin, wt, 1, c.st, c.dt()));
    return io_run_dgrad(dy, wt, dx, add, mask, c.N, H, H, L.cin, L.cout, L.k, L.k, L.stride, L.pad, c.st, bw, c.dt());
}

bool tiles_ok(const Ctx& c, int M) { return M % c.G == 0 && (M / c.G) % kIoStatTileRows == 0; }

// Does the BatchNorm backward of a layer with M rows leave its apply pass to the data-gradient kernel that consumes dy
// (IoBwStats::xb_a: dy = a * dz + b * y + c evaluated on the staged operand, written out once for the filter gradient)?
// fp32 only: there the transform hides under the MFMAs (as the forward one does); whole 128-row tiles per group.
// Applied: bn3 -> conv3's, bn1 -> conv1's (also of the three blocks whose conv2 is strided), the downsample BatchNorm's -> the
// downsample convolution's data gradient, the stem's bn1 in the staging of the stem's filter gradient (stem.hip).  NOT bn2 ->
// conv2's 3x3 data gradient (measured: a loss, see run_backward).
bool xb_ok(const Ctx& c, int M) { return c.net->dtype == IO_F32 && tiles_ok(c, M); }

IoBwStats bw_for(const Ctx& c, const BnL& b, const void* y, int M, bool mask_from_y) {
    Tables t = c.tables(b);
    IoBwStats bw{};
    bw.y = y;
    bw.mean = t.mean;
    bw.rstd = t.rstd;
    bw.mscale = mask_from_y ? t.scale : nullptr;
    bw.mshift = mask_from_y ? t.shift : nullptr;
    bw.p1 = c.buf(c.plan.tile_mean);     // the forward's tile-partial scratch is free during the backward
    bw.p2 = c.buf(c.plan.tile_m2);
    bw.Mg = M / c.G;
    return bw;
}

// BatchNorm backward WITHOUT the apply pass: dgamma / dbeta and the coefficient tables of the operand transform, from the
// tile partials the epilogue of the launch that completed dz left behind, or (have_tiles false) from a reduction pass
// over (dz, y); dz already carries its ReLU mask.
int bn_back_coefs(const Ctx& c, const BnL& b, const void* dz, const void* y, int M, bool have_tiles) {
    Tables t = c.tables(b);
    if (have_tiles)
        return io_bn_bwd_coefs_from_tiles(c.buf(c.plan.tile_mean), c.buf(c.plan.tile_m2), M, b.C, c.G,
                                          c.params + b.g_off, t.mean, t.rstd, c.grads + b.g_off, c.grads + b.b_off,
                                          c.buf(c.plan.coef), c.st);
    return io_bn_bwd_coefs_t(dz, y, M, b.C, c.G, c.params + b.g_off, t.mean, t.rstd, c.grads + b.g_off,
                             c.grads + b.b_off, c.buf(c.plan.coef), c.buf(c.plan.bn_partial), c.plan.bn_partial_floats,
                             c.st, c.dt());
}

// the operand-transform part of a data-gradient launch whose A operand is that BatchNorm's input gradient
void xb_fill(const Ctx& c, IoBwStats& bw, const BnL& b, const void* y, int M, void* dy_out) {
    const size_t gs = (size_t)c.G * b.C;
    bw.xb_y = y;
    bw.xb_a = c.buf(c.plan.coef);
    bw.xb_b = c.buf(c.plan.coef) + gs;
    bw.xb_c = c.buf(c.plan.coef) + 2 * gs;
    bw.xb_out = dy_out;
    bw.xb_Mg = M / c.G;
}

int bn_back_tiles(const Ctx& c, const BnL& b, const void* dz, const void* y, int M, void* dy) {
    Tables t = c.tables(b);
    return io_bn_bwd_from_tiles(c.buf(c.plan.tile_mean), c.buf(c.plan.tile_m2), dz, y, M, b.C, c.G,
                                c.params + b.g_off, t.mean, t.rstd, c.grads + b.g_off, c.grads + b.b_off, dy,
                                c.buf(c.plan.coef), c.st, c.dt());
}

// data gradient of conv L (-> dx, the gradient of relu(bn(y))), then the backward of that BN (-> dyb).
// With a stride-1 conv the BN-backward reductions ride in the conv epilogue (which also applies the ReLU
// mask recomputed from y), and only the apply pass remains.
// a_out (optional, fused path only): relu(bn(y)) rebuilt next to dx
// xb (optional, fused path only): `dy` is NOT the gradient of the conv output but the masked gradient dz of the BatchNorm
// behind it, whose coefficients are in plan.coef: the launch evaluates dy on its operand and writes it to xb_out
// dyb == nullptr: stop after the data gradient -- the tile partials of b are left for bn_back_coefs
int dgrad_then_bn(const Ctx& c, const ConvL& L, const void* dy, void* dx, int H, const BnL& b, const void* y,
                  int M, void* dyb, void* a_out = nullptr, const BnL* xb = nullptr, const void* xb_y = nullptr,
                  int xb_M = 0, void* xb_out = nullptr) {
    if (L.stride == 1 && tiles_ok(c, M)) {
        IoBwStats bw = bw_for(c, b, y, M, true);
        bw.a_out = a_out;
        if (xb) xb_fill(c, bw, *xb, xb_y, xb_M, xb_out);
        if (c.plan.wino_u != kNoBuf && L.k == 3 && c.dt() == IO_F32 && !xb) bw.wino_u = c.buf(c.plan.wino_u);
        IO_TRY(conv_dgrad(c, L, dy, dx, nullptr, nullptr, H, &bw));
        if (!dyb) return IO_OK;
        return bn_back_tiles(c, b, dx, y, M, dyb);
    }
    IO_REQUIRE(!xb && dyb, IO_ERR_STATE, "dgrad_then_bn: the operand transform needs the fused path");
    IO_REQUIRE(!a_out, IO_ERR_STATE, "dgrad_then_bn: no fused epilogue to rebuild the activation in");
    IO_TRY(conv_dgrad(c, L, dy, dx, nullptr, nullptr, H));
    return bn_back(c, b, dx, 1, nullptr, y, M, dyb, nullptr);
}

// Backward stages, in execution order (gradients of a stage are final when its last launch has run, so a data-parallel
// caller can start exchanging that slice of the flat gradient buffer while the next stage computes):
//   0 = heads + layer4, 1 = layer3, 2 = layer2, 3 = layer1 + stem.   Runs the stages [stage_lo, stage_hi).
// Nothing is carried between calls but the workspace: which scratch buffer holds d(block output) and whether the tile
// partials of a block's bn3 are waiting follow from the block index alone.
constexpr int kBwdStages = 4;
int run_backward(Ctx& c, const float* dlogits, const void* x8, int stage_lo = 0, int stage_hi = kBwdStages) {
    const io_net* net = c.net;
    const Plan& p = c.plan;
    void* Gd = c.act(p.gbuf[0]);
    void* Ge = c.act(p.gbuf[1]);
    void* Ga = c.act(p.gbuf[2]);
    void* Gb = c.act(p.gbuf[3]);
    void* Gc = c.act(p.gbuf[4]);
    const int H0 = c.S / 2, H1 = c.S / 4;
    // spatial size of the last stage
    int Hlast = H1;
    for (const Block& b : net->blocks) Hlast /= b.stride;
    const size_t nb = net->blocks.size();
    const float* w1 = net->n_heads > 1 ? c.params + net->fcw_off[1] : nullptr;
    // Gradients of block outputs are kept ALREADY MASKED by that output's ReLU: whoever writes the last
    // contribution to d(out) applies [out > 0] in its epilogue (here: the pooling backward; below: the
    // data-gradient kernels), so the BN backward of the block needs neither the activation nor a mask pass.
    if (stage_lo == 0) IO_TRY(transpose_filters(c));
    if (stage_lo == 0)
    IO_TRY(io_avgpool_fc_bwd_t(dlogits, c.buf(p.pooled), c.N, Hlast * Hlast, 2048, c.params + net->fcw_off[0],
                               net->head_dims[0], w1, net->n_heads > 1 ? net->head_dims[1] : 0,
                               c.act(p.blk[nb - 1].out), Gd, c.grads + net->fcw_off[0], c.grads + net->fcb_off[0],
                               net->n_heads > 1 ? c.grads + net->fcw_off[1] : nullptr,
                               net->n_heads > 1 ? c.grads + net->fcb_off[1] : nullptr, c.st, c.dt()));
    // per-block input resolution
    std::vector<int> Hin(nb);
    {
        int H = H1;
        for (size_t i = 0; i < nb; ++i) { Hin[i] = H; H /= net->blocks[i].stride; }
    }
    // stage of a block: the blocks of one resolution, last resolution first
    std::vector<int> stage_of(nb);
    {
        int st = kBwdStages, res = -1;
        for (size_t i = 0; i < nb; ++i) {                    // forward order: a stride-2 block opens a new stage ...
            if (i == 0 || net->blocks[i].stride != 1) res += 1;
            stage_of[i] = res;
        }
        for (size_t i = 0; i < nb; ++i) stage_of[i] = res - stage_of[i];     // ... numbered from the back
        st = res + 1;
        IO_REQUIRE(st == kBwdStages, IO_ERR_STATE, "run_backward: %d resolution stages, expected %d", st, kBwdStages);
    }
    bool have_tiles = false;    // BN-backward partial sums of the current block's bn3 already produced?
    for (size_t ii = nb; ii-- > 0;) {
        const Block& b = net->blocks[ii];
        const BlockBufs& bb = p.blk[ii];
        const int H = Hin[ii], Ho = H / b.stride;
        const int Min = c.N * H * H, Mout = c.N * Ho * Ho;
        if (stage_of[ii] < stage_lo || stage_of[ii] >= stage_hi) {
            // not ours: only keep the bookkeeping in step (buffer roles alternate per block; the carry rule of below)
            have_tiles = ii > 0 && tiles_ok(c, Min);
            void* t = Gd; Gd = Ge; Ge = t;
            continue;
        }
        const void* xin = ii == 0 ? c.act(p.p0) : c.act(p.blk[ii - 1].out);
        // the block input is the previous block's post-ReLU output (for block 0 it is the max-pool output,
        // whose gradient is not masked here)
        const void* xmask = ii == 0 ? nullptr : xin;
        // bn3: Gd holds dz = d(out) * [out > 0]; dy3 -> Ga
        // Where xb_ok, the apply pass of a BatchNorm backward (read dz, read y, write dy) does not exist: the reductions
        // become three coefficient tables and the data-gradient launch that consumes dy evaluates it on its operand --
        // dz and y are loaded side by side, dy = a * dz + b * y + c in registers -- and writes it out once (first
        // output-channel tile only) for the filter gradient, which therefore runs AFTER that launch.
        // Measured per shape at the bench batch (tools/xb_bench.py, profiles/r03_xb_microbench_fp32.txt): the 1x1 data
        // gradients pay +0.00..0.17 ms for the doubled operand load (+0.54 on the HBM-bound 256 -> 64 layer) and save an
        // apply pass of 0.04..1.29 ms -- a gain on every conv3 and conv1; a 3x3 data gradient stages every chunk nine
        // times (once per tap), pays +0.21..0.61 ms and saves 0.04..0.32: bn2 keeps its apply pass.
        // (bf16: the transform is VALU-bound next to bf16 MFMAs and loses everywhere except on layer 1's 256 -> 64 data
        // gradient, which is so HBM-bound that the saved pass still wins: -0.33 ms per launch, r03_xb_microbench_bf16.txt)
        // (bf16, round 6: on layers 2-4 the same form runs on the 256-row kernel as an in-LDS transform of the A k-tiles
        // (conv_p256.hip XOP) wherever that kernel takes the launch -- conv3's data gradient only: x3p)
        const bool x3p = c.net->dtype == IO_BF16 && b.planes > IO_XB_BF16_MAXP && tiles_ok(c, Mout) &&
                         io_conv_p256_takes_xop(Mout, b.planes * 4, b.planes, Mout / c.G);
        const bool x3 = xb_ok(c, Mout) || x3p ||                 // bn3 -> conv3's data gradient
                        (IO_XB_BF16_L1 && c.net->dtype == IO_BF16 && b.planes <= IO_XB_BF16_MAXP && tiles_ok(c, Mout));
        const bool f3 = fuse_in(c, Mout), f2 = f3 && b.stride == 1;
        // bn1 -> conv1's; needs bn1's tile partials from the epilogue of conv2's dense data gradient
        const bool x1d = b.stride == 1 && xb_ok(c, Min) && tiles_ok(c, Mout);
        // a strided conv2 runs its data gradient as parity classes, which carry no BatchNorm epilogue: there the mask comes
        // from the stored activation a1 (kept for exactly these blocks) in that launch's epilogue, bn1's sums from a
        // reduction pass over (dz1, y1), and the apply pass again rides in conv1's operand load
        const bool x1s = b.stride != 1 && xb_ok(c, Min) && bb.a1 != kNoBuf;
        const bool x1 = x1d || x1s;
        if (x3)
            IO_TRY(bn_back_coefs(c, b.b3, Gd, c.act(bb.y3), Mout, have_tiles));
        else if (have_tiles)
            IO_TRY(bn_back_tiles(c, b.b3, Gd, c.act(bb.y3), Mout, Ga));
        else
            IO_TRY(bn_back(c, b.b3, Gd, 0, nullptr, c.act(bb.y3), Mout, Ga, nullptr));
        have_tiles = false;
        // Where the forward read relu(bn(y)) through the input transform (fuse_in) that activation was never stored: the
        // data-gradient launch, which recomputes the ReLU mask from y anyway, rebuilds it into the one `aside` buffer, and
        // the filter gradient that needs it runs right after (instead of right before) that launch.
        void* As = f3 ? c.act(p.aside) : nullptr;
        if (!f3 && !x3) IO_TRY(conv_wgrad(c, b.c3, c.act(bb.a2), Ga, Ho));
        // conv3: (dy3 | dz3 + y3) -> Gb = dz2 (+ bn2's partials), Gc = dy2
        IO_TRY(dgrad_then_bn(c, b.c3, x3 ? Gd : Ga, Gb, Ho, b.b2, c.act(bb.y2), Mout, Gc, f3 ? As : nullptr,
                             x3 ? &b.b3 : nullptr, c.act(bb.y3), Mout, Ga));
        if (f3 || x3) IO_TRY(conv_wgrad(c, b.c3, f3 ? As : c.act(bb.a2), Ga, Ho));
        if (!f2) IO_TRY(conv_wgrad(c, b.c2, c.act(bb.a1), Gc, H));
        // conv2: (dy2 | dz2 + y2) -> Ga = dz1 (+ bn1's partials); without x1 also Gb = dy1
        if (x1s)
            IO_TRY(conv_dgrad(c, b.c2, Gc, Ga, nullptr, c.act(bb.a1), H));
        else
            IO_TRY(dgrad_then_bn(c, b.c2, Gc, Ga, H, b.b1, c.act(bb.y1), Min, x1 ? nullptr : Gb, f2 ? As : nullptr));
        if (f2) IO_TRY(conv_wgrad(c, b.c2, As, Gc, H));
        if (!x1) IO_TRY(conv_wgrad(c, b.c1, xin, Gb, H));
        // d(x_in) = dgrad(conv1) + identity path, masked by the ReLU of x_in (= previous block's output).
        // Without a downsample branch this launch completes d(x_in), so it can also carry the reductions of
        // the previous block's bn3.  With one, the mask is idempotent and is applied by both kernels (the
        // strided one only touches its own output lattice).
        // With a downsample branch its (strided) data gradient goes FIRST into d(x_in) -- the lattice classes it
        // does not reach are written as zeros -- so that conv1's dense stride-1 data gradient is again the launch
        // that completes d(x_in) and can carry the previous block's bn3 reductions there too.
        const void* partial = Gd;          // what conv1's data gradient accumulates onto: the identity path ...
        if (b.down && x3 && !x3p) {
            // the downsample BatchNorm the same way: reductions over (dz, yd) -> tables, dy evaluated on the operand of the
            // (strided) 1x1 data gradient -- only the lattice class that has a tap stages anything -- and written to Gc for
            // the filter gradient.  (Gc: dy2 has been consumed by conv2's data and filter gradients; Ga may hold dz1.)
            IO_TRY(bn_back_coefs(c, b.bd, Gd, c.act(bb.yd), Mout, false));
            IoBwStats bw{};
            xb_fill(c, bw, b.bd, c.act(bb.yd), Mout, Gc);
            IO_TRY(conv_dgrad(c, b.cd, Gd, Ge, nullptr, nullptr, H, &bw));
            IO_TRY(conv_wgrad(c, b.cd, xin, Gc, H));
            partial = Ge;
        } else if (b.down) {
            IO_TRY(bn_back(c, b.bd, Gd, 0, nullptr, c.act(bb.yd), Mout, Gc, nullptr));
            IO_TRY(conv_wgrad(c, b.cd, xin, Gc, H));
            IO_TRY(conv_dgrad(c, b.cd, Gc, Ge, nullptr, nullptr, H));
            partial = Ge;                  // ... or the downsample path
        }
        // conv1: (dy1 | dz1 + y1) -> Ge = d(x_in).  (bn1's tables only now: the downsample BatchNorm's backward above
        // uses the same coefficient scratch; bn1's tile partials are untouched by it)
        if (x1) IO_TRY(bn_back_coefs(c, b.b1, Ga, c.act(bb.y1), Min, x1d));
        {
            IoBwStats bw{};
            const bool carry = ii > 0 && tiles_ok(c, Min);
            if (carry) bw = bw_for(c, net->blocks[ii - 1].b3, c.act(p.blk[ii - 1].y3), Min, false);
            if (x1) xb_fill(c, bw, b.b1, c.act(bb.y1), Min, Gb);
            // [x_in > 0] also as one bit per element where the forward left it: the 256-row kernel reads that instead of the
            // tensor (4 of the 17 bytes per element this launch moves in bf16); conv_nt_kernel ignores it
            const uint32_t* mbits = (ii > 0 && xmask && bits_written(c, ii - 1)) ? bits_of(c, ii - 1) : nullptr;
            if (mbits) bw.maskbits = mbits;
            IO_TRY(conv_dgrad(c, b.c1, x1 ? Ga : Gb, Ge, partial, xmask, H, (carry || x1 || mbits) ? &bw : nullptr));
            have_tiles = carry;
        }
        if (x1) IO_TRY(conv_wgrad(c, b.c1, xin, Gb, H));
        void* t = Gd; Gd = Ge; Ge = t;
    }
    if (stage_hi < kBwdStages) return IO_OK;
    // Gd = d(maxpool output)
    IO_TRY(io_maxpool_bwd_t(Gd, reinterpret_cast<const uint32_t*>(c.ws + p.idx0), c.N, H0, H0, 64, Ge, c.st, c.dt()));
    // stem filter gradient only: the network input needs no data gradient -- so bn1's backward has ONE reader, and on
    // the row-persistent kernel it is evaluated while that reader stages its rows (no apply pass: reduction + tables only)
    {
        IoConvGeom g = io_geom_fwd(c.N, c.S, c.S, 8, 64, 7, 7, 2, 3);
        g.cr = net->stem.cin;
        if (stem_exact(c, net->stem) && io_stem_rows_ok(g) && c.N % c.G == 0) {
            const BnL& b = net->bn1;
            Tables t = c.tables(b);
            IO_TRY(io_bn_bwd_coefs_t(Ge, c.act(p.y0), c.N * H0 * H0, b.C, c.G, c.params + b.g_off, t.mean, t.rstd,
                                     c.grads + b.g_off, c.grads + b.b_off, c.buf(c.plan.coef), c.buf(c.plan.bn_partial),
                                     c.plan.bn_partial_floats, c.st, c.dt(), t.scale, t.shift));
            const size_t gs = (size_t)c.G * b.C;
            IoStemXb xb{};
            xb.y = (const float*)c.act(p.y0);
            xb.a = c.buf(c.plan.coef);
            xb.b = c.buf(c.plan.coef) + gs;
            xb.c = c.buf(c.plan.coef) + 2 * gs;
            xb.mean = t.mean;
            xb.scale = t.scale;
            xb.shift = t.shift;
            xb.G = c.G;
            IO_TRY(io_launch_stem_wgrad_rows(g, (const float*)x8, (const float*)Ge, c.buf(c.plan.stem_dwp),
                                             c.buf(c.plan.wg_partial), c.plan.wg_partial_bytes, c.st, &xb));
            return io_stem_unpack_grad(c.buf(c.plan.stem_dwp), c.grads + net->stem.w_off, net->stem.cout, 49,
                                       net->stem.cin, c.st);
        }
    }
    // bf16: the same fusion on stem_wgrad_halo_kernel (conv_halo3.hip) where the stem has 128-wide output rows
    if (c.net->dtype == IO_BF16) {
        IoConvGeom g = io_geom_fwd(c.N, c.S, c.S, 8, 64, 7, 7, 2, 3);
        if (io_stem_wgrad_halo_ok(g, c.plan.wg_partial_bytes, c.G)) {
            const BnL& b = net->bn1;
            Tables t = c.tables(b);
            IO_TRY(io_bn_bwd_coefs_t(Ge, c.act(p.y0), c.N * H0 * H0, b.C, c.G, c.params + b.g_off, t.mean, t.rstd,
                                     c.grads + b.g_off, c.grads + b.b_off, c.buf(c.plan.coef), c.buf(c.plan.bn_partial),
                                     c.plan.bn_partial_floats, c.st, c.dt(), t.scale, t.shift));
            const size_t gs = (size_t)c.G * b.C;
            IoStemXb xb{};
            xb.y = (const float*)c.act(p.y0);        // (bf16 storage)
            xb.a = c.buf(c.plan.coef);
            xb.b = c.buf(c.plan.coef) + gs;
            xb.c = c.buf(c.plan.coef) + 2 * gs;
            xb.mean = t.mean;
            xb.scale = t.scale;
            xb.shift = t.shift;
            xb.G = c.G;
            return io_launch_stem_wgrad_halo(g, x8, Ge, c.grads + net->stem.w_off, c.buf(c.plan.wg_partial),
                                             c.plan.wg_partial_bytes, c.st, &xb);
        }
    }
    IO_TRY(bn_back(c, net->bn1, Ge, 1, nullptr, c.act(p.y0), c.N * H0 * H0, Ga, nullptr));
    IO_TRY(conv_wgrad(c, net->stem, x8, Ga, c.S));
    return IO_OK;
}

}  // namespace

extern "C" io_net* io_net_create(int in_channels, int n_heads, const int* head_dims) {
    if (in_channels < 1 || in_channels > 5 || n_heads < 1 || n_heads > 2) {
        io_set_error("io_net_create: in_channels=%d (1..5), n_heads=%d (1..2)", in_channels, n_heads);
        return nullptr;
    }
    for (int i = 0; i < n_heads; ++i)
        if (head_dims[i] < 1 || head_dims[i] > 4 || (n_heads == 2 && i == 0 && head_dims[i] != 2)) {
            io_set_error("io_net_create: unsupported head dims");
            return nullptr;
        }
    io_net* net = new io_net();
    net->in_ch = in_channels;
    net->n_heads = n_heads;
    net->head_dims[0] = head_dims[0];
    net->head_dims[1] = n_heads > 1 ? head_dims[1] : 0;
    net->param_floats = 0;
    net->running_floats = 0;
    net->n_bn = 0;
    net->bn_channels = 0;
    net->stem = add_conv(net, "conv1", 64, in_channels, 7, 2, 3);
    net->stem.cin_store = 8;
    net->bn1 = add_bn(net, "bn1", 64);
    const int layers[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512};
    int inC = 64;
    for (int li = 0; li < 4; ++li)
        for (int bi = 0; bi < layers[li]; ++bi) {
            Block b;
            char nm[64];
            b.planes = planes[li];
            b.inC = inC;
            b.stride = (bi == 0 && li > 0) ? 2 : 1;
            b.down = bi == 0;
            snprintf(nm, sizeof(nm), "layer%d.%d.conv1", li + 1, bi);
            b.c1 = add_conv(net, nm, b.planes, inC, 1, 1, 0);
            snprintf(nm, sizeof(nm), "layer%d.%d.bn1", li + 1, bi);
            b.b1 = add_bn(net, nm, b.planes);
            snprintf(nm, sizeof(nm), "layer%d.%d.conv2", li + 1, bi);
            b.c2 = add_conv(net, nm, b.planes, b.planes, 3, b.stride, 1);
            snprintf(nm, sizeof(nm), "layer%d.%d.bn2", li + 1, bi);
            b.b2 = add_bn(net, nm, b.planes);
            snprintf(nm, sizeof(nm), "layer%d.%d.conv3", li + 1, bi);
            b.c3 = add_conv(net, nm, b.planes * 4, b.planes, 1, 1, 0);
            snprintf(nm, sizeof(nm), "layer%d.%d.bn3", li + 1, bi);
            b.b3 = add_bn(net, nm, b.planes * 4);
            if (b.down) {
                snprintf(nm, sizeof(nm), "layer%d.%d.downsample.0", li + 1, bi);
                b.cd = add_conv(net, nm, b.planes * 4, inC, 1, b.stride, 0);
                snprintf(nm, sizeof(nm), "layer%d.%d.downsample.1", li + 1, bi);
                b.bd = add_bn(net, nm, b.planes * 4);
            }
            inC = b.planes * 4;
            net->blocks.push_back(b);
        }
    const char* hn1[1] = {"fc"};
    const char* hn2[2] = {"fc_occ", "fc_depth"};
    for (int i = 0; i < n_heads; ++i) {
        char nm[64];
        const char* base = n_heads == 1 ? hn1[0] : hn2[i];
        const long ws[2] = {head_dims[i], 2048};
        snprintf(nm, sizeof(nm), "%s.weight", base);
        add_tensor(net, nm, 3, 2, ws, (long)head_dims[i] * 2048, 2048, &net->fcw_off[i]);
        const long bs[1] = {head_dims[i]};
        snprintf(nm, sizeof(nm), "%s.bias", base);
        add_tensor(net, nm, 4, 1, bs, head_dims[i], head_dims[i], &net->fcb_off[i]);
    }
    return net;
}

extern "C" void io_net_destroy(io_net* net) { delete net; }
extern "C" int io_net_set_dtype(io_net* net, int dtype) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "io_net_set_dtype: unknown dtype %d", dtype);
    net->dtype = dtype;
    return IO_OK;
}
extern "C" int io_net_get_dtype(const io_net* net) { return net->dtype; }
extern "C" long io_net_param_floats(const io_net* net) { return net->param_floats; }
extern "C" long io_net_running_floats(const io_net* net) { return net->running_floats; }
extern "C" int io_net_num_tensors(const io_net* net) { return (int)net->tensors.size(); }
extern "C" int io_net_num_logits(const io_net* net) { return net->head_dims[0] + net->head_dims[1]; }

extern "C" int io_net_tensor_info(const io_net* net, int i, io_tensor_info* out) {
    IO_REQUIRE(i >= 0 && i < (int)net->tensors.size(), IO_ERR_SHAPE, "tensor index %d out of range", i);
    *out = net->tensors[i];
    return IO_OK;
}

static int check_shape(int N, int S, int G) {
    IO_REQUIRE(S >= 32 && S % 32 == 0, IO_ERR_SHAPE, "input size S=%d must be a multiple of 32", S);
    IO_REQUIRE(N >= 1 && G >= 1 && G <= kMaxGroups && N % G == 0, IO_ERR_SHAPE, "N=%d G=%d (G<=%d, G | N)", N, G,
               kMaxGroups);
    // tensors may exceed 4 GiB (the kernels address them tile by tile); row and pixel counts must fit an int
    IO_REQUIRE((double)N * S * S < 2.0e9, IO_ERR_SHAPE, "N=%d x S=%d: more than 2^31 input pixels", N, S);
    return IO_OK;
}

extern "C" size_t io_net_workspace_bytes(const io_net* net, int N, int S, int training) {
    if (check_shape(N, S, 1)) return 0;
    return make_plan(net, N, S, training != 0).total;
}

extern "C" long io_net_activation_offset(const io_net* net, int N, int S, int which) {
    if (check_shape(N, S, 1)) return -1;
    const Plan p = make_plan(net, N, S, true);
    const int nb = (int)net->blocks.size();
    if (which == 0) return (long)p.y0;
    if (which == 1) {
        io_set_error("io_net_activation_offset: relu(bn1(.)) is not stored (it is evaluated inside the pooling kernel)");
        return -1;
    }
    if (which == 2) return (long)p.p0;
    if (which >= 3 && which < 3 + nb) return (long)p.blk[which - 3].out;
    io_set_error("io_net_activation_offset: which=%d (0..%d)", which, 2 + nb);
    return -1;
}

static void fill_ctx(Ctx& c, io_net* net, const float* params, float* running, float* grads, int N, int S, int G,
                     bool training, void* ws, hipStream_t st, int SW = 0) {
    c.net = net;
    c.params = params;
    c.running = running;
    c.grads = grads;
    c.ws = (char*)ws;
    c.N = N; c.S = S; c.G = G;
    c.SW = SW > 0 ? SW : S;
    c.training = training;
    c.st = st;
    c.plan = make_plan(net, N, S, training, c.SW);
    c.chan_prefix.assign(net->n_bn, 0);
    // BN index -> channel prefix (tables are packed in BN creation order)
    std::vector<int> Cs(net->n_bn, 0);
    Cs[net->bn1.index] = net->bn1.C;
    for (const Block& b : net->blocks) {
        Cs[b.b1.index] = b.b1.C;
        Cs[b.b2.index] = b.b2.C;
        Cs[b.b3.index] = b.b3.C;
        if (b.down) Cs[b.bd.index] = b.bd.C;
    }
    long acc = 0;
    for (int i = 0; i < net->n_bn; ++i) { c.chan_prefix[i] = acc; acc += Cs[i]; }
}

extern "C" int io_net_forward(io_net* net, const float* params, float* running, const void* x8, int N, int S,
                              int G, int training, void* ws, size_t ws_bytes, float* logits, hipStream_t st) {
    IO_TRY(check_shape(N, S, G));
    IO_REQUIRE(training || G == 1, IO_ERR_SHAPE, "eval forward uses running statistics: G must be 1");
    Ctx c;
    fill_ctx(c, net, params, running, nullptr, N, S, G, training != 0, ws, st);
    IO_REQUIRE(ws_bytes >= c.plan.total, IO_ERR_WORKSPACE, "io_net_forward: workspace %zu < %zu bytes", ws_bytes,
               c.plan.total);
    return run_forward(c, x8, logits);
}

// Inference on H x W inputs (the reference's 'orig' mode: the whole image at its own aspect ratio, sides rounded to
// multiples of 32; the network is fully convolutional up to the global average pool)
static int check_shape_hw(int N, int H, int W) {
    IO_REQUIRE(H >= 32 && H % 32 == 0 && W >= 32 && W % 32 == 0, IO_ERR_SHAPE,
               "input %d x %d: both sides must be multiples of 32 (five stride-2 stages)", H, W);
    IO_REQUIRE(N >= 1 && (double)N * H * W < 2.0e9, IO_ERR_SHAPE, "N=%d x %d x %d: more than 2^31 input pixels", N, H, W);
    return IO_OK;
}
extern "C" size_t io_net_workspace_bytes_hw(const io_net* net, int N, int H, int W) {
    if (check_shape_hw(N, H, W)) return 0;
    return make_plan(net, N, H, false, W).total;
}
extern "C" int io_net_forward_eval_hw(io_net* net, const float* params, float* running, const void* x8, int N, int H,
                                      int W, void* ws, size_t ws_bytes, float* logits, hipStream_t st) {
    IO_TRY(check_shape_hw(N, H, W));
    Ctx c;
    fill_ctx(c, net, params, running, nullptr, N, H, 1, false, ws, st, W);
    IO_REQUIRE(ws_bytes >= c.plan.total, IO_ERR_WORKSPACE, "io_net_forward_eval_hw: workspace %zu < %zu bytes", ws_bytes,
               c.plan.total);
    return run_forward_eval(c, x8, logits);
}

extern "C" int io_net_backward(io_net* net, const float* params, float* grads, const void* x8,
                               const float* dlogits, int N, int S, int G, void* ws, size_t ws_bytes,
                               hipStream_t st) {
    IO_TRY(check_shape(N, S, G));
    Ctx c;
    fill_ctx(c, net, params, nullptr, grads, N, S, G, true, ws, st);
    IO_REQUIRE(ws_bytes >= c.plan.total, IO_ERR_WORKSPACE, "io_net_backward: workspace %zu < %zu bytes", ws_bytes,
               c.plan.total);
    return run_backward(c, dlogits, x8);
}

extern "C" int io_net_backward_stages(io_net* net, const float* params, float* grads, const void* x8,
                                      const float* dlogits, int N, int S, int G, void* ws, size_t ws_bytes,
                                      int stage_lo, int stage_hi, hipStream_t st) {
    IO_TRY(check_shape(N, S, G));
    IO_REQUIRE(stage_lo >= 0 && stage_lo < stage_hi && stage_hi <= kBwdStages, IO_ERR_SHAPE,
               "io_net_backward_stages: stages [%d, %d) outside [0, %d)", stage_lo, stage_hi, kBwdStages);
    Ctx c;
    fill_ctx(c, net, params, nullptr, grads, N, S, G, true, ws, st);
    IO_REQUIRE(ws_bytes >= c.plan.total, IO_ERR_WORKSPACE, "io_net_backward_stages: workspace %zu < %zu bytes", ws_bytes,
               c.plan.total);
    return run_backward(c, dlogits, x8, stage_lo, stage_hi);
}

extern "C" int io_net_backward_num_stages(const io_net*) { return kBwdStages; }
