// C ABI glue: error plumbing, device probe, convolution entry points (geometry builders).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "io_common.h"

static thread_local char g_err[512] = "";

void io_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int io_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        io_set_error("%s: %s", what, hipGetErrorString(e));
        return IO_ERR_LAUNCH;
    }
    return IO_OK;
}

extern "C" int io_abi_version(void) { return 1; }

// Winograd row forms on / off (header: io_set_winograd).  -1 = not read from the environment yet.
static std::atomic<int> g_wino{-1};
bool io_wino_on() {
    int v = g_wino.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("IO_WINOGRAD");
        v = (e && e[0] == '0') ? 0 : 1;
        g_wino.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" int io_get_winograd(void) { return io_wino_on() ? 1 : 0; }
extern "C" int io_set_winograd(int on) {
    const int prev = io_wino_on() ? 1 : 0;
    g_wino.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}
extern "C" const char* io_last_error_string(void) { return g_err; }

extern "C" int io_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

// ---- geometry builders ---------------------------------------------------------------------
IoConvGeom io_geom_fwd(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad) {
    IoConvGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.Hi = H; g.Wi = W; g.Ci = Cin;
    g.Ho = (H + 2 * pad - R) / stride + 1;
    g.Wo = (W + 2 * pad - S) / stride + 1;
    g.Co = Cout;
    g.outH = g.Ho; g.outW = g.Wo;
    g.os = 1; g.ooh = 0; g.oow = 0;
    g.is = stride;
    g.Th = R; g.Tw = S;
    g.dh0 = -pad; g.dhs = 1; g.dw0 = -pad; g.dws = 1;
    g.r0 = 0; g.rs = 1; g.s0 = 0; g.ss = 1;
    g.S = S; g.wT = R * S;
    io_geom_finish(g);
    return g;
}

// data gradient of the conv above for the output-pixel parity class (ph, pw) of dx[N,H,W,Cin]:
// dx[h] = sum_r dy[(h + pad - r)/stride] W[r] over r == (h + pad) mod stride.
IoConvGeom io_geom_dgrad(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int ph,
                         int pw) {
    IoConvGeom g;
    memset(&g, 0, sizeof(g));
    const int Hy = (H + 2 * pad - R) / stride + 1, Wy = (W + 2 * pad - S) / stride + 1;
    g.N = N; g.Hi = Hy; g.Wi = Wy; g.Ci = Cout;
    g.Ho = (H - ph + stride - 1) / stride;
    g.Wo = (W - pw + stride - 1) / stride;
    g.Co = Cin;
    g.outH = H; g.outW = W;
    g.os = stride; g.ooh = ph; g.oow = pw;
    g.is = 1;
    const int rf = (ph + pad) % stride, sf = (pw + pad) % stride;
    g.Th = rf < R ? (R - rf + stride - 1) / stride : 0;
    g.Tw = sf < S ? (S - sf + stride - 1) / stride : 0;
    g.dh0 = (ph + pad - rf) / stride; g.dhs = -1;
    g.dw0 = (pw + pad - sf) / stride; g.dws = -1;
    g.r0 = rf; g.rs = stride; g.s0 = sf; g.ss = stride;
    g.S = S; g.wT = R * S;
    io_geom_finish(g);
    return g;
}

int io_run_dgrad(const void* dy, const void* wt, void* dx, const void* add, const void* mask, int N, int H,
                 int W, int Cin, int Cout, int R, int S, int stride, int pad, hipStream_t st, const IoBwStats* bw,
                 int dt, int gw) {
    IO_REQUIRE(!bw || !bw->y || stride == 1, IO_ERR_SHAPE, "dgrad: fused BN-backward reductions need a stride-1 convolution");
    IO_REQUIRE(!bw || !bw->xb_a || stride == 1 || (R == 1 && S == 1 && pad == 0), IO_ERR_SHAPE,
               "dgrad: the operand transform needs a stride-1 convolution or a strided 1x1 one");
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            IoConvGeom g = io_geom_dgrad(N, H, W, Cin, Cout, R, S, stride, pad, ph, pw);
            g.gw = gw;
            if (g.Ho <= 0 || g.Wo <= 0) continue;
            // a lattice class no tap reaches keeps its values when accumulating in place -- unless a ReLU
            // mask has to be applied to them
            if ((g.Th == 0 || g.Tw == 0) && add == dx && !mask) continue;
            int rc = io_launch_conv_nt(g, dy, wt, dx, add, mask, 0, st, nullptr, nullptr, bw, dt, dt);
            if (rc) return rc;
        }
    return IO_OK;
}

extern "C" int io_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout,
                             int R, int S, int stride, int pad, hipStream_t st) {
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    return io_launch_conv_nt(g, x, w, y, nullptr, nullptr, Cin == 8, st);
}

extern "C" size_t io_conv2d_bnstats_workspace_floats(int N, int H, int W, int Cout, int R, int S, int stride, int pad,
                                                     int G) {
    const long Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    const long tiles = (N * Ho * Wo + kIoStatTileRows - 1) / kIoStatTileRows;
    return (size_t)2 * (size_t)((tiles + tiles / 64 + G + 2) * Cout);
}

extern "C" int io_conv2d_fwd_bnstats(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout,
                                     int R, int S, int stride, int pad, int G, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps,
                                     float* mean, float* rstd, float* scale, float* shift, float* workspace,
                                     size_t workspace_floats, hipStream_t st) {
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    const int M = N * g.Ho * g.Wo;
    IO_REQUIRE(G >= 1 && M % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "conv2d_fwd_bnstats: rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    const size_t need = io_conv2d_bnstats_workspace_floats(N, H, W, Cout, R, S, stride, pad, G);
    IO_REQUIRE(workspace_floats >= need, IO_ERR_WORKSPACE, "conv2d_fwd_bnstats: workspace %zu < %zu floats",
               workspace_floats, need);
    float* tmean = workspace;
    float* tm2 = workspace + need / 2;
    int rc = io_launch_conv_nt(g, x, w, y, nullptr, nullptr, Cin == 8, st, tmean, tm2);
    if (rc) return rc;
    return io_bn_finalize_tiles(tmean, tm2, M, Cout, G, gamma, beta, running_mean, running_var, momentum, eps, mean,
                                rstd, scale, shift, st);
}

extern "C" int io_conv2d_dgrad(const float* dy, const float* wt, float* dx, const float* add, const float* relu_mask,
                               int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                               hipStream_t st) {
    IO_REQUIRE(Cin % 64 == 0, IO_ERR_SHAPE, "conv2d_dgrad: Cin=%d must be a multiple of 64", Cin);
    return io_run_dgrad(dy, wt, dx, add, relu_mask, N, H, W, Cin, Cout, R, S, stride, pad, st);
}

extern "C" int io_conv2d_dgrad_bnbwd(const float* dy, const float* wt, float* dz, int N, int H, int W, int Cin, int Cout,
                                     int R, int S, int pad, const float* y, int G, const float* gamma,
                                     const float* mean, const float* rstd, const float* scale, const float* shift,
                                     float* dgamma, float* dbeta, float* dyb, float* workspace,
                                     size_t workspace_floats, hipStream_t st) {
    IO_REQUIRE(Cin % 64 == 0, IO_ERR_SHAPE, "conv2d_dgrad_bnbwd: Cin=%d must be a multiple of 64", Cin);
    const int M = N * H * W;
    IO_REQUIRE(G >= 1 && M % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "conv2d_dgrad_bnbwd: rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    const size_t tiles = (size_t)M / kIoStatTileRows;
    const size_t per = (tiles + tiles / 64 + G + 2) * (size_t)Cin;
    IO_REQUIRE(workspace_floats >= 2 * per + 2 * (size_t)G * Cin, IO_ERR_WORKSPACE,
               "conv2d_dgrad_bnbwd: workspace %zu < %zu floats", workspace_floats, 2 * per + 2 * (size_t)G * Cin);
    IoBwStats bw{};
    bw.y = y; bw.mean = mean; bw.rstd = rstd; bw.mscale = scale; bw.mshift = shift;
    bw.p1 = workspace; bw.p2 = workspace + per; bw.Mg = M / G;
    int rc = io_run_dgrad(dy, wt, dz, nullptr, nullptr, N, H, W, Cin, Cout, R, S, 1, pad, st, &bw);
    if (rc) return rc;
    return io_bn_bwd_from_tiles(bw.p1, bw.p2, dz, y, M, Cin, G, gamma, mean, rstd, dgamma, dbeta, dyb,
                                workspace + 2 * per, st);
}

extern "C" size_t io_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int R, int S, int stride,
                                                  int pad) {
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    return io_conv_wgrad_partial_bytes(g, Cin == 8);
}

extern "C" int io_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                               int R, int S, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    return io_launch_conv_wgrad(g, x, dy, dw, (float*)ws, ws_bytes, Cin == 8, st);
}

// ---- profiling: start/stop events around every launch group of a class, summed at io_prof_end ------
#include <mutex>
#include <vector>
namespace {
// e0 of a scope that directly follows a closed scope on the same stream IS that scope's e1 (own0 = false): one event
// record per launch group instead of two -- every record is a barrier packet the GPU has to chew through, ~1 us each,
// 1600 of them per step.  A class's time then runs from the end of the previous profiled launch to the end of its own.
struct ProfRec { int cls; double flops, bytes, fexec; hipEvent_t e0, e1; bool own0, closed; hipStream_t st; };
std::mutex g_prof_mu;
bool g_prof_on = false;
bool g_prof_share = true;       // consecutive scopes share one event (see IoProfScope); off when foreign launches interleave
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
const char* kProfNames[IO_PROF_NCLASS] = {"conv_nt_kernel<128,false>", "conv_nt_kernel<64,false>",
    "conv_nt_kernel<64,true>", "conv_wgrad_kernel", "conv_wgrad_kernel<64,64,true>", "bn_stats_finalize",
    "bn_apply", "bn_bwd", "pool_head", "filter_transpose", "pack_planes", "order_loss", "sgd_momentum",
    "conv_nt_kernel<wino>", "conv_wgrad_kernel<wino>"};
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

IoProfScope::IoProfScope(int cls, double flops, double bytes, hipStream_t stream, double flops_exec) : idx(-1), st(stream) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return;
    ProfRec r;
    r.cls = cls; r.flops = flops; r.bytes = bytes; r.fexec = flops_exec < 0.0 ? flops : flops_exec;
    r.st = st; r.closed = false;
    if (g_prof_share && !g_prof_recs.empty() && g_prof_recs.back().closed && g_prof_recs.back().st == st) {
        r.e0 = g_prof_recs.back().e1;
        r.own0 = false;
    } else {
        r.e0 = prof_event();
        r.own0 = true;
        (void)hipEventRecord(r.e0, st);
    }
    r.e1 = prof_event();
    idx = (int)g_prof_recs.size();
    g_prof_recs.push_back(r);
}
IoProfScope::~IoProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx < (int)g_prof_recs.size()) {
        (void)hipEventRecord(g_prof_recs[idx].e1, st);
        g_prof_recs[idx].closed = true;
    }
}

extern "C" int io_prof_begin(void);
/* share_events = 0: every launch group gets its own start event -- for callers that put OTHER work (torch kernels of an
 * op-by-op graph, host gaps) between the library's launches; with shared events that time would be charged to the next
 * class and the classes would always sum to wall time. */
extern "C" int io_prof_begin_ex(int share_events) {
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_share = share_events != 0;
    }
    return io_prof_begin();
}

extern "C" int io_prof_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfRec& r : g_prof_recs) { if (r.own0) g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
    g_prof_recs.clear();
    g_prof_on = true;
    return IO_OK;
}

/* per-launch records of the running profile (launch order), WITHOUT ending it: each entry = one launch group */
extern "C" int io_prof_launches(io_prof_entry* out, int max_entries) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (ProfRec& r : g_prof_recs) {
        if (n >= max_entries) break;
        float ms = 0.f;
        if (!r.closed || hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess)
            continue;
        io_prof_entry& e = out[n++];
        snprintf(e.name, sizeof(e.name), "%s", kProfNames[r.cls]);
        e.launches = 1;
        e.total_ms = ms;
        e.flops = r.flops;
        e.bytes = r.bytes;
        e.flops_executed = r.fexec;
    }
    return n;
}

extern "C" int io_prof_end(io_prof_entry* out, int max_entries) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    io_prof_entry acc[IO_PROF_NCLASS];
    memset(acc, 0, sizeof(acc));
    for (int c = 0; c < IO_PROF_NCLASS; ++c) snprintf(acc[c].name, sizeof(acc[c].name), "%s", kProfNames[c]);
    for (ProfRec& r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            acc[r.cls].launches += 1;
            acc[r.cls].total_ms += ms;
            acc[r.cls].flops += r.flops;
            acc[r.cls].bytes += r.bytes;
            acc[r.cls].flops_executed += r.fexec;
        }
        if (r.own0) g_prof_pool.push_back(r.e0);
        g_prof_pool.push_back(r.e1);
    }
    g_prof_recs.clear();
    int n = 0;
    for (int c = 0; c < IO_PROF_NCLASS && n < max_entries; ++c)
        if (acc[c].launches > 0) out[n++] = acc[c];
    return n;
}

// ---- storage-typed variants (dtype: 0 = fp32, 1 = bf16 activations / operands; see IoDType) ------------------
extern "C" int io_pack_planes_nhwc8_dt(const float* const* planes, const long* sample_strides, int nplanes, int N,
                                       int H, int W, void* out, int dtype, hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "pack: unknown dtype %d", dtype);
    return io_pack_planes_t(planes, sample_strides, nplanes, N, H, W, out, st, dtype);
}

extern "C" int io_conv2d_fwd_dt(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R,
                                int S, int stride, int pad, int dtype_in, int dtype_out, hipStream_t st) {
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    return io_launch_conv_nt(g, x, w, y, nullptr, nullptr, Cin == 8, st, nullptr, nullptr, nullptr, dtype_in,
                             dtype_out);
}

extern "C" int io_conv2d_dgrad_dt(const void* dy, const void* wt, void* dx, const void* add, const void* relu_mask,
                                  int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  int dtype, hipStream_t st) {
    IO_REQUIRE(Cin % 64 == 0, IO_ERR_SHAPE, "conv2d_dgrad: Cin=%d must be a multiple of 64", Cin);
    return io_run_dgrad(dy, wt, dx, add, relu_mask, N, H, W, Cin, Cout, R, S, stride, pad, st, nullptr, dtype);
}

extern "C" int io_conv2d_wgrad_dt(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                                  int R, int S, int stride, int pad, void* ws, size_t ws_bytes, int dtype_in,
                                  int dtype_dy, hipStream_t st) {
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    return io_launch_conv_wgrad(g, x, dy, dw, (float*)ws, ws_bytes, Cin == 8, st, dtype_in, dtype_dy);
}

extern "C" int io_filter_prepare(const float* w, int Cout, int taps, int Cin, void* dst, int transpose, int dtype,
                                 hipStream_t st) {
    return io_filter_prepare_t(w, Cout, taps, Cin, dst, transpose, st, dtype);
}

extern "C" int io_bn_stats_finalize_dt(const void* y, int M, int C, int G, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps,
                                       float* mean, float* rstd, float* scale, float* shift, float* partial,
                                       size_t partial_floats, int dtype, hipStream_t st) {
    return io_bn_stats_finalize_t(y, M, C, G, gamma, beta, running_mean, running_var, momentum, eps, mean, rstd, scale,
                                  shift, partial, partial_floats, st, dtype);
}

extern "C" int io_bn_apply_dt(const void* y, int M, int C, int G, int per_group_tables, const float* mean,
                              const float* scale, const float* shift, const void* identity, const float* mean2,
                              const float* scale2, const float* shift2, int relu, void* out, int dtype,
                              hipStream_t st) {
    return io_bn_apply_t(y, M, C, G, per_group_tables, mean, scale, shift, identity, mean2, scale2, shift2, relu, out,
                         st, dtype);
}

extern "C" int io_bn_apply_bits_dt(const void* y, int M, int C, int G, int per_group_tables, const float* mean,
                                   const float* scale, const float* shift, const void* identity, const float* mean2,
                                   const float* scale2, const float* shift2, void* out, uint32_t* bits, int dtype,
                                   hipStream_t st) {
    IO_REQUIRE(bits, IO_ERR_SHAPE, "bn_apply_bits: no bit mask given");
    return io_bn_apply_t(y, M, C, G, per_group_tables, mean, scale, shift, identity, mean2, scale2, shift2, 1, out, st,
                         dtype, bits);
}

extern "C" int io_bn_bwd_dt(const void* dout, const void* act, const float* mask_scale, const float* mask_shift,
                            const void* y, int M, int C, int G, const float* gamma, const float* mean,
                            const float* rstd, float* dgamma, float* dbeta, void* dy, void* dz_out, float* partial,
                            size_t partial_floats, float* coef, int dtype, hipStream_t st) {
    return io_bn_bwd_t(dout, act, mask_scale, mask_shift, y, M, C, G, gamma, mean, rstd, dgamma, dbeta, dy, dz_out,
                       partial, partial_floats, coef, st, dtype);
}


extern "C" int io_maxpool_fwd_dt(const void* x, int N, int H, int W, int C, void* out, uint32_t* idx, int dtype,
                                 hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "maxpool: unknown dtype %d", dtype);
    return io_maxpool_fwd_t(x, N, H, W, C, out, idx, st, dtype);
}

extern "C" int io_maxpool_bwd_dt(const void* dy, const uint32_t* idx, int N, int H, int W, int C, void* dx, int dtype,
                                 hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "maxpool: unknown dtype %d", dtype);
    return io_maxpool_bwd_t(dy, idx, N, H, W, C, dx, st, dtype);
}

extern "C" int io_avgpool_fc_fwd_dt(const void* x, int N, int HW, int C, const float* w0, const float* b0, int K0,
                                    const float* w1, const float* b1, int K1, float* pooled, float* logits, int dtype,
                                    hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "avgpool_fc: unknown dtype %d", dtype);
    return io_avgpool_fc_fwd_t(x, N, HW, C, w0, b0, K0, w1, b1, K1, pooled, logits, st, dtype);
}

extern "C" int io_avgpool_fc_bwd_dt(const float* dlogits, const float* pooled, int N, int HW, int C, const float* w0,
                                    int K0, const float* w1, int K1, const void* relu_mask, void* dx, float* dw0,
                                    float* db0, float* dw1, float* db1, int dtype, hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "avgpool_fc: unknown dtype %d", dtype);
    return io_avgpool_fc_bwd_t(dlogits, pooled, N, HW, C, w0, K0, w1, K1, relu_mask, dx, dw0, db0, dw1, db1, st, dtype);
}


/* conv + training BatchNorm statistics in one pass, storage-typed, dense (gw = 0) or grouped-window (gw = 64) */
extern "C" int io_conv2d_fwd_bnstats_dt(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout,
                                        int R, int S, int stride, int pad, int G, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                        float* rstd, float* scale, float* shift, float* workspace,
                                        size_t workspace_floats, int dtype, int gw, hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "conv2d_fwd_bnstats: unknown dtype %d", dtype);
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    g.gw = gw;
    const int M = N * g.Ho * g.Wo;
    IO_REQUIRE(G >= 1 && M % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "conv2d_fwd_bnstats: rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    const size_t need = io_conv2d_bnstats_workspace_floats(N, H, W, Cout, R, S, stride, pad, G);
    IO_REQUIRE(workspace_floats >= need, IO_ERR_WORKSPACE, "conv2d_fwd_bnstats: workspace %zu < %zu floats",
               workspace_floats, need);
    float* tmean = workspace;
    float* tm2 = workspace + need / 2;
    int rc = io_launch_conv_nt(g, x, w, y, nullptr, nullptr, Cin == 8, st, tmean, tm2, nullptr, dtype, dtype);
    if (rc) return rc;
    return io_bn_finalize_tiles(tmean, tm2, M, Cout, G, gamma, beta, running_mean, running_var, momentum, eps, mean,
                                rstd, scale, shift, st);
}


/* io_conv2d_dgrad_bnbwd for either storage type and for the grouped window form (gw = 64: wt = wtc of io_gconv_pack, Cin
 * == Cout) */
extern "C" int io_conv2d_dgrad_bnbwd_dt(const void* dy, const void* wt, void* dz, int N, int H, int W, int Cin, int Cout,
                                        int R, int S, int pad, const void* y, int G, const float* gamma, const float* mean,
                                        const float* rstd, const float* scale, const float* shift, float* dgamma,
                                        float* dbeta, void* dyb, float* workspace, size_t workspace_floats, int dtype,
                                        int gw, hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "conv2d_dgrad_bnbwd: unknown dtype %d", dtype);
    IO_REQUIRE(Cin % 64 == 0, IO_ERR_SHAPE, "conv2d_dgrad_bnbwd: Cin=%d must be a multiple of 64", Cin);
    const int M = N * H * W;
    IO_REQUIRE(G >= 1 && M % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "conv2d_dgrad_bnbwd: rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    const size_t tiles = (size_t)M / kIoStatTileRows;
    const size_t per = (tiles + tiles / 64 + G + 2) * (size_t)Cin;
    IO_REQUIRE(workspace_floats >= 2 * per + 2 * (size_t)G * Cin, IO_ERR_WORKSPACE,
               "conv2d_dgrad_bnbwd: workspace %zu < %zu floats", workspace_floats, 2 * per + 2 * (size_t)G * Cin);
    IoBwStats bw{};
    bw.y = y; bw.mean = mean; bw.rstd = rstd; bw.mscale = scale; bw.mshift = shift;
    bw.p1 = workspace; bw.p2 = workspace + per; bw.Mg = M / G;
    int rc = io_run_dgrad(dy, wt, dz, nullptr, nullptr, N, H, W, Cin, Cout, R, S, 1, pad, st, &bw, dtype, gw);
    if (rc) return rc;
    return io_bn_bwd_from_tiles(bw.p1, bw.p2, dz, y, M, Cin, G, gamma, mean, rstd, dgamma, dbeta, dyb,
                                workspace + 2 * per, st, dtype);
}

/* the stem in exact-K mode (fp32): what net.hip launches for the fp32 step */
extern "C" size_t io_stem_packed_floats(int real_channels) { return (size_t)64 * io_stem_kp(49, real_channels); }
static IoConvGeom stem_geom_exact(int N, int H, int W, int cr) {
    IoConvGeom g = io_geom_fwd(N, H, W, 8, 64, 7, 7, 2, 3);
    g.cr = cr;
    return g;
}
extern "C" size_t io_stem_wgrad_exact_workspace_bytes(int N, int H, int W, int real_channels) {
    return io_conv_wgrad_partial_bytes(stem_geom_exact(N, H, W, real_channels), 1);
}
extern "C" int io_stem_fwd_bnstats_exact(const float* x8, const float* w, float* y, int N, int H, int W, int real_channels,
                                         int G, const float* gamma, const float* beta, float* running_mean,
                                         float* running_var, float momentum, float eps, float* mean, float* rstd,
                                         float* scale, float* shift, float* workspace, size_t workspace_floats,
                                         float* packed, hipStream_t st) {
    IO_REQUIRE(real_channels >= 1 && real_channels <= 8, IO_ERR_SHAPE, "stem_exact: real_channels=%d (1..8)", real_channels);
    IoConvGeom g = stem_geom_exact(N, H, W, real_channels);
    const int M = N * g.Ho * g.Wo;
    IO_REQUIRE(G >= 1 && M % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "stem_exact: rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    const size_t need = io_conv2d_bnstats_workspace_floats(N, H, W, 64, 7, 7, 2, 3, G);
    IO_REQUIRE(workspace_floats >= need && packed, IO_ERR_WORKSPACE, "stem_exact: workspace %zu < %zu floats",
               workspace_floats, need);
    int rc = io_stem_pack_filter(w, packed, 64, 49, real_channels, st);
    if (rc) return rc;
    float* tmean = workspace;
    float* tm2 = workspace + need / 2;
    rc = io_launch_conv_nt(g, x8, packed, y, nullptr, nullptr, 1, st, tmean, tm2);
    if (rc) return rc;
    return io_bn_finalize_tiles(tmean, tm2, M, 64, G, gamma, beta, running_mean, running_var, momentum, eps, mean, rstd,
                                scale, shift, st);
}
extern "C" int io_stem_wgrad_exact(const float* x8, const float* dy, float* dw, int N, int H, int W, int real_channels,
                                   void* ws, size_t ws_bytes, float* packed, hipStream_t st) {
    IO_REQUIRE(real_channels >= 1 && real_channels <= 8 && packed, IO_ERR_SHAPE, "stem_wgrad_exact: real_channels=%d (1..8)",
               real_channels);
    IoConvGeom g = stem_geom_exact(N, H, W, real_channels);
    int rc = io_launch_conv_wgrad(g, x8, dy, packed, (float*)ws, ws_bytes, 1, st);
    if (rc) return rc;
    return io_stem_unpack_grad(packed, dw, 64, 49, real_channels, st);
}

/* ... with bn1's backward evaluated while the gradient kernel stages its rows (what the fp32 step runs where 256 | H, W):
 * da = gradient of relu(bn1(y)) (the pooling backward's output), y = the raw stem output */
extern "C" int io_stem_wgrad_exact_bn(const float* x8, const float* da, const float* y, float* dw, int N, int H, int W,
                                      int real_channels, int G, const float* gamma, const float* mean, const float* rstd,
                                      const float* scale, const float* shift, float* dgamma, float* dbeta, float* coef,
                                      float* bn_partial, size_t bn_partial_floats, void* ws, size_t ws_bytes, float* packed,
                                      hipStream_t st) {
    IO_REQUIRE(real_channels >= 1 && real_channels <= 8 && packed && coef, IO_ERR_SHAPE,
               "stem_wgrad_exact_bn: real_channels=%d (1..8)", real_channels);
    IoConvGeom g = stem_geom_exact(N, H, W, real_channels);
    IO_REQUIRE(io_stem_rows_ok(g) && G >= 1 && N % G == 0, IO_ERR_SHAPE,
               "stem_wgrad_exact_bn: needs 5 real channels, 256 | H, W and G | N (else: io_bn_bwd + io_stem_wgrad_exact)");
    const int M = N * g.Ho * g.Wo;
    int rc = io_bn_bwd_coefs_t(da, y, M, 64, G, gamma, mean, rstd, dgamma, dbeta, coef, bn_partial, bn_partial_floats, st, IO_F32,
                               scale, shift);
    if (rc) return rc;
    IoStemXb xb{};
    xb.y = y;
    xb.a = coef;
    xb.b = coef + (size_t)G * 64;
    xb.c = coef + (size_t)2 * G * 64;
    xb.mean = mean;
    xb.scale = scale;
    xb.shift = shift;
    xb.G = G;
    rc = io_launch_stem_wgrad_rows(g, x8, da, packed, (float*)ws, ws_bytes, st, &xb);
    if (rc) return rc;
    return io_stem_unpack_grad(packed, dw, 64, 49, real_channels, st);
}

extern "C" size_t io_stem_wgrad_bf16_workspace_bytes(void) { return io_stem_wgrad_halo_partial_bytes(); }
extern "C" int io_stem_wgrad_bn_bf16(const void* x8, const void* da, const void* y, float* dw, int N, int H, int W, int G,
                                     const float* gamma, const float* mean, const float* rstd, const float* scale,
                                     const float* shift, float* dgamma, float* dbeta, float* coef, float* bn_partial,
                                     size_t bn_partial_floats, void* ws, size_t ws_bytes, hipStream_t st) {
    IO_REQUIRE(x8 && da && y && dw && coef && ws, IO_ERR_SHAPE, "stem_wgrad_bn_bf16: null argument");
    IoConvGeom g = io_geom_fwd(N, H, W, 8, 64, 7, 7, 2, 3);
    IO_REQUIRE(io_stem_wgrad_halo_ok(g, ws_bytes, G), IO_ERR_SHAPE,
               "stem_wgrad_bn_bf16: needs 256 x 256 inputs (128-wide output rows), G | N, io_stem_wgrad_bf16_workspace_bytes() of "
               "workspace and enough rows for the persistent blocks (else: io_bn_bwd_dt + io_conv2d_wgrad_dt)");
    const int M = N * g.Ho * g.Wo;
    int rc = io_bn_bwd_coefs_t(da, y, M, 64, G, gamma, mean, rstd, dgamma, dbeta, coef, bn_partial, bn_partial_floats, st, IO_BF16,
                               scale, shift);
    if (rc) return rc;
    IoStemXb xb{};
    xb.y = (const float*)y;
    xb.a = coef;
    xb.b = coef + (size_t)G * 64;
    xb.c = coef + (size_t)2 * G * 64;
    xb.mean = mean;
    xb.scale = scale;
    xb.shift = shift;
    xb.G = G;
    return io_launch_stem_wgrad_halo(g, x8, da, dw, (float*)ws, ws_bytes, st, &xb);
}

extern "C" size_t io_bn_tile_partial_floats(int M, int C, int G) {
    if (M <= 0 || C <= 0 || G <= 0) return 0;
    const size_t tiles = (size_t)(M + kIoStatTileRows - 1) / kIoStatTileRows;
    return (tiles + tiles / 64 + (size_t)G + 2) * (size_t)C;
}

extern "C" int io_bn_bwd_coefs_dt(const void* dz, const void* y, int M, int C, int G, const float* gamma, const float* mean,
                                  const float* rstd, float* dgamma, float* dbeta, float* coef, float* partial,
                                  size_t partial_floats, int dtype, hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "bn_bwd_coefs: unknown dtype %d", dtype);
    return io_bn_bwd_coefs_t(dz, y, M, C, G, gamma, mean, rstd, dgamma, dbeta, coef, partial, partial_floats, st, dtype);
}

extern "C" int io_bn_bwd_coefs_from_tile_partials(float* p1, float* p2, int M, int C, int G, const float* gamma,
                                                  const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                                  float* coef, hipStream_t st) {
    return io_bn_bwd_coefs_from_tiles(p1, p2, M, C, G, gamma, mean, rstd, dgamma, dbeta, coef, st);
}

/* the data-gradient launch of the training step with everything it can carry (include/instaorder_hip.h) */
extern "C" int io_conv2d_dgrad_fused_dt(const void* dy, const void* wt, void* dx, int N, int H, int W, int Cin, int Cout,
                                        int R, int S, int pad, int G, const io_dgrad_fused* f, int dtype, hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "conv2d_dgrad_fused: unknown dtype %d", dtype);
    IO_REQUIRE(f != nullptr, IO_ERR_SHAPE, "conv2d_dgrad_fused: the option struct is required");
    IO_REQUIRE(Cin % 64 == 0, IO_ERR_SHAPE, "conv2d_dgrad_fused: Cin=%d must be a multiple of 64", Cin);
    const int M = N * H * W;
    const bool grouped = f->xb_coef || f->ep_y;
    IO_REQUIRE(!grouped || (G >= 1 && M % G == 0 && (M / G) % kIoStatTileRows == 0), IO_ERR_SHAPE,
               "conv2d_dgrad_fused: rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    IO_REQUIRE(!f->xb_coef || f->xb_y, IO_ERR_SHAPE, "conv2d_dgrad_fused: xb_coef needs xb_y");
    IO_REQUIRE(!f->ep_y || (f->ep_mean && f->ep_rstd && f->ep_p1 && f->ep_p2 && (f->ep_scale == nullptr) == (f->ep_shift == nullptr)),
               IO_ERR_SHAPE, "conv2d_dgrad_fused: ep_y needs ep_mean / ep_rstd / ep_p1 / ep_p2 (and scale / shift as a pair)");
    IO_REQUIRE(!f->ep_act_out || (f->ep_y && f->ep_scale), IO_ERR_SHAPE,
               "conv2d_dgrad_fused: ep_act_out needs ep_y with ep_scale / ep_shift");
    IoBwStats bw{};
    if (f->ep_y) {
        bw.y = f->ep_y; bw.mean = f->ep_mean; bw.rstd = f->ep_rstd; bw.mscale = f->ep_scale; bw.mshift = f->ep_shift;
        bw.p1 = f->ep_p1; bw.p2 = f->ep_p2; bw.Mg = M / G; bw.a_out = f->ep_act_out;
    }
    if (f->xb_coef) {
        const size_t gs = (size_t)G * Cout;
        bw.xb_y = f->xb_y; bw.xb_a = f->xb_coef; bw.xb_b = f->xb_coef + gs; bw.xb_c = f->xb_coef + 2 * gs;
        bw.xb_out = f->xb_dy_out; bw.xb_Mg = M / G;
    }
    if (f->wino_scratch) {
        IO_REQUIRE(f->wino_scratch_floats >= io_conv2d_wino_scratch_floats(Cin, Cout), IO_ERR_WORKSPACE,
                   "conv2d_dgrad_fused: wino_scratch %zu < %zu floats", f->wino_scratch_floats,
                   io_conv2d_wino_scratch_floats(Cin, Cout));
        bw.wino_u = f->wino_scratch;
    }
    if (f->relu_maskbits) {
        IO_REQUIRE(f->relu_mask && Cin % 32 == 0, IO_ERR_SHAPE,
                   "conv2d_dgrad_fused: relu_maskbits goes WITH relu_mask (the same mask, one bit per element), 32 | Cin");
        bw.maskbits = f->relu_maskbits;
    }
    return io_run_dgrad(dy, wt, dx, f->add, f->relu_mask, N, H, W, Cin, Cout, R, S, 1, pad, st,
                        (grouped || f->wino_scratch || f->relu_maskbits) ? &bw : nullptr, dtype);
}

/* Forward convolution whose INPUT goes through the BatchNorm + ReLU of the layer that produced it, applied while the
 * operand is staged (the normalised activation never exists in memory): y = conv(relu((x - in_mean[g]) * in_scale[g] +
 * in_shift[g]), w) -- bn_apply's expression, in_mean optional (NULL = 0) --
 * zero padding applied AFTER the transform, as nn.Conv2d pads relu(bn(x)) (resnet_cls.py:99-111: bn -> relu -> next
 * conv).  in_scale / in_shift: [G][Cin]; group g = the samples [g N/G, (g+1) N/G).  With workspace != NULL the training
 * statistics of y are produced as by io_conv2d_fwd_bnstats_dt (same G). */
extern "C" size_t io_conv2d_wino_scratch_floats(int Cin, int Cout) { return (size_t)18 * Cin * Cout; }   // F(4,3): 3 x 6 planes

static int conv2d_fwd_xf_impl(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R,
                              int S, int stride, int pad, int G, const float* in_mean, const float* in_scale,
                              const float* in_shift, const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* mean, float* rstd, float* scale, float* shift,
                              float* workspace, size_t workspace_floats, int dtype, hipStream_t st, float* wino_scratch,
                              size_t wino_floats);

extern "C" int io_conv2d_fwd_xf_dt(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R,
                                   int S, int stride, int pad, int G, const float* in_mean, const float* in_scale,
                                   const float* in_shift, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, float* mean, float* rstd, float* scale, float* shift,
                                   float* workspace, size_t workspace_floats, int dtype, hipStream_t st) {
    IO_REQUIRE(in_scale && in_shift, IO_ERR_SHAPE, "conv2d_fwd_xf: in_scale / in_shift are required");
    return conv2d_fwd_xf_impl(x, w, y, N, H, W, Cin, Cout, R, S, stride, pad, G, in_mean, in_scale, in_shift, gamma, beta,
                              running_mean, running_var, momentum, eps, mean, rstd, scale, shift, workspace,
                              workspace_floats, dtype, st, nullptr, 0);
}

/* The 3x3 stride-1 pad-1 fp32 convolution of io_conv2d_fwd_xf_dt (in_scale NULL: no input transform) in the Winograd
 * F(2, 3) row form; wino_scratch: io_conv2d_wino_scratch_floats(Cin, Cout) floats of device scratch for the transformed
 * filters.  Shapes the form does not cover (odd W, rows not in whole 128-row tiles) run the direct kernel. */
extern "C" int io_conv2d_fwd_wino(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int G,
                                  const float* in_mean, const float* in_scale, const float* in_shift, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                  float* mean, float* rstd, float* scale, float* shift, float* workspace,
                                  size_t workspace_floats, float* wino_scratch, size_t wino_floats, hipStream_t st) {
    IO_REQUIRE(wino_scratch && wino_floats >= io_conv2d_wino_scratch_floats(Cin, Cout), IO_ERR_WORKSPACE,
               "conv2d_fwd_wino: wino_scratch %zu < %zu floats", wino_floats, io_conv2d_wino_scratch_floats(Cin, Cout));
    return conv2d_fwd_xf_impl(x, w, y, N, H, W, Cin, Cout, 3, 3, 1, 1, G, in_mean, in_scale, in_shift, gamma, beta,
                              running_mean, running_var, momentum, eps, mean, rstd, scale, shift, workspace,
                              workspace_floats, IO_F32, st, wino_scratch, wino_floats);
}

static int conv2d_fwd_xf_impl(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R,
                              int S, int stride, int pad, int G, const float* in_mean, const float* in_scale,
                              const float* in_shift, const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* mean, float* rstd, float* scale, float* shift,
                              float* workspace, size_t workspace_floats, int dtype, hipStream_t st, float* wino_scratch,
                              size_t wino_floats) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "conv2d_fwd_xf: unknown dtype %d", dtype);
    IO_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), IO_ERR_SHAPE, "conv2d_fwd_xf: in_scale / in_shift come as a pair");
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    const int M = N * g.Ho * g.Wo;
    IO_REQUIRE(G >= 1 && N % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "conv2d_fwd_xf: output rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    IoBwStats ep{};
    ep.in_mean = in_mean;
    ep.in_scale = in_scale;
    ep.in_shift = in_shift;
    ep.in_Mg = M / G;
    ep.wino_u = wino_scratch;
    if (!workspace) return io_launch_conv_nt(g, x, w, y, nullptr, nullptr, 0, st, nullptr, nullptr, &ep, dtype, dtype);
    const size_t need = io_conv2d_bnstats_workspace_floats(N, H, W, Cout, R, S, stride, pad, G);
    IO_REQUIRE(workspace_floats >= need, IO_ERR_WORKSPACE, "conv2d_fwd_xf: workspace %zu < %zu floats", workspace_floats,
               need);
    float* tmean = workspace;
    float* tm2 = workspace + need / 2;
    int rc = io_launch_conv_nt(g, x, w, y, nullptr, nullptr, 0, st, tmean, tm2, &ep, dtype, dtype);
    if (rc) return rc;
    return io_bn_finalize_tiles(tmean, tm2, M, Cout, G, gamma, beta, running_mean, running_var, momentum, eps, mean,
                                rstd, scale, shift, st);
}

/* 1x1 convolution whose INPUT is a Bottleneck output that does not exist yet: conv(relu(bn3(y3) + identity), w), with
 * that tensor written to `out` on the way (resnet_cls.py:108-114 out = relu(bn3(conv3(.)) + identity); then the next block's
 * conv1, :99).  fp32, H*W*N/G a multiple of 128.  Statistics of y as in io_conv2d_fwd_xf_dt. */
extern "C" int io_conv2d_fwd_resid(const float* y3, const float* identity, const float* w, float* y, float* out, int N,
                                   int H, int W, int Cin, int Cout, int G, const float* in_mean, const float* in_scale,
                                   const float* in_shift, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float momentum, float eps, float* mean, float* rstd, float* scale,
                                   float* shift, float* workspace, size_t workspace_floats, hipStream_t st) {
    IO_REQUIRE(in_mean && in_scale && in_shift && identity, IO_ERR_SHAPE, "conv2d_fwd_resid: tables and identity are required");
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, 1, 1, 1, 0);
    const int M = N * H * W;
    IO_REQUIRE(G >= 1 && N % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "conv2d_fwd_resid: rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    IoBwStats ep{};
    ep.xb_y = identity; ep.xb_a = in_scale; ep.xb_b = in_mean; ep.xb_c = in_shift; ep.xb_out = out; ep.xb_Mg = M / G;
    ep.xb_res = 1;
    if (!workspace) return io_launch_conv_nt(g, y3, w, y, nullptr, nullptr, 0, st, nullptr, nullptr, &ep);
    const size_t need = io_conv2d_bnstats_workspace_floats(N, H, W, Cout, 1, 1, 1, 0, G);
    IO_REQUIRE(workspace_floats >= need, IO_ERR_WORKSPACE, "conv2d_fwd_resid: workspace %zu < %zu floats", workspace_floats,
               need);
    float* tmean = workspace;
    float* tm2 = workspace + need / 2;
    int rc = io_launch_conv_nt(g, y3, w, y, nullptr, nullptr, 0, st, tmean, tm2, &ep);
    if (rc) return rc;
    return io_bn_finalize_tiles(tmean, tm2, M, Cout, G, gamma, beta, running_mean, running_var, momentum, eps, mean,
                                rstd, scale, shift, st);
}

/* io_conv2d_fwd_resid for either storage type, with both table forms and the one-bit mask of `out` (header) */
extern "C" int io_conv2d_fwd_resid_dt(const void* y3, const void* second, const void* w, void* y, void* out, uint32_t* out_bits,
                                      int N, int H, int W, int Cin, int Cout, int G, int two, const float* ta, const float* tb,
                                      const float* tc, float* tile_mean, float* tile_m2, int dtype, hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "conv2d_fwd_resid_dt: unknown dtype %d", dtype);
    IO_REQUIRE(ta && tb && tc && second, IO_ERR_SHAPE, "conv2d_fwd_resid_dt: tables and the second operand are required");
    IO_REQUIRE((tile_mean == nullptr) == (tile_m2 == nullptr), IO_ERR_SHAPE, "conv2d_fwd_resid_dt: statistics outputs come in pairs");
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, 1, 1, 1, 0);
    const int M = N * H * W;
    IO_REQUIRE(G >= 1 && N % G == 0 && (M / G) % kIoStatTileRows == 0, IO_ERR_SHAPE,
               "conv2d_fwd_resid_dt: rows per BN group (%d) must be a multiple of %d", G ? M / G : 0, kIoStatTileRows);
    IoBwStats ep{};
    ep.xb_y = second; ep.xb_a = ta; ep.xb_b = tb; ep.xb_c = tc; ep.xb_out = out; ep.xb_Mg = M / G;
    ep.xb_res = two ? 2 : 1;
    ep.xb_bits = out_bits;
    return io_launch_conv_nt(g, y3, w, y, nullptr, nullptr, 0, st, tile_mean, tile_m2, &ep, dtype, dtype);
}

/* convolution with an inference epilogue: y = [relu](conv(x, w) + bias[o] (+ add)) -- a BatchNorm in eval mode folded
 * into pre-scaled filters, or a biased nn.Conv2d; dense (gw = 0) or grouped-window (gw = 64) */
extern "C" int io_conv2d_fwd_bias_dt(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R,
                                     int S, int stride, int pad, const float* bias, const void* add, int relu, int dtype,
                                     int gw, hipStream_t st) {
    IO_REQUIRE(dtype == IO_F32 || dtype == IO_BF16, IO_ERR_SHAPE, "conv2d_fwd_bias: unknown dtype %d", dtype);
    IO_REQUIRE(bias != nullptr, IO_ERR_SHAPE, "conv2d_fwd_bias: bias is required");
    IoConvGeom g = io_geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad);
    g.gw = gw;
    IoBwStats ep{};
    ep.bias = bias;
    ep.relu = relu;
    return io_launch_conv_nt(g, x, w, y, add, nullptr, Cin == 8, st, nullptr, nullptr, &ep, dtype, dtype);
}
