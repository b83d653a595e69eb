/* instaorder_hip.h -- C ABI of libinstaorder_hip.so (gfx950 / MI355X only).
 *
 * Drop-in boundary for the pairwise order-prediction hot path of POSTECH-CVLab/InstaOrder.
 * The reference has no FFI of its own (100 % Python on torch.nn); what it binds for this path
 * are torch.nn modules and functions, so each entry point below names the reference call site
 * it replaces (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers unless stated otherwise; NHWC activations, fp32 (`float*`) or, in the
 *     `_dt` / `dtype` entry points, of the stated storage type (`void*`: float or bfloat16); filters as
 *     [Cout][R*S][Cin] ("KRSC"); the 5-channel network input is padded to 8 channels; tensors may exceed 4 GiB;
 *   - return 0 on success, a negative IO_ERR_* code on failure; never throw; the message of the
 *     last failure on the calling thread is returned by io_last_error_string();
 *   - never allocate device memory, never synchronise: work is enqueued on `stream`
 *     (hipStream_t passed as void*; NULL = the default stream); scratch comes from the caller;
 *   - no global mutable state: safe to call from several host threads on different streams.
 */
#ifndef INSTAORDER_HIP_H
#define INSTAORDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IO_OK 0
#define IO_ERR_SHAPE (-1)      /* unsupported / inconsistent shape argument            */
#define IO_ERR_WORKSPACE (-2)  /* caller-provided scratch too small                    */
#define IO_ERR_LAUNCH (-3)     /* HIP reported an error for a launch                   */
#define IO_ERR_STATE (-4)      /* call sequence violated (e.g. backward before forward) */
#define IO_ERR_NODEVICE (-5)   /* no gfx950 device visible                             */

/* hipStream_t comes from the HIP runtime header when the caller has included it; otherwise it is an opaque pointer */
#if !defined(__HIP__) && !defined(HIP_INCLUDE_HIP_HIP_RUNTIME_API_H)
typedef void* hipStream_t;
#endif

int io_abi_version(void);
const char* io_last_error_string(void);
/* number of visible HIP devices whose arch is gfx950; <= 0 means the library cannot run */
int io_device_count(void);
/* Product form of the fp32 3x3 stride-1 convolutions / data gradients / filter gradients: 1 (default) = Winograd's minimal
 * row forms F(4,3) / F(2,3) where the shape allows (half / two thirds of the matrix products, re-associated sums), 0 = the
 * direct implicit GEMM everywhere -- the arithmetic of the reference's nn.Conv2d up to summation order (resnet_cls.py:23-26).
 * Process-wide, read by every later launch; the initial value comes from the environment variable IO_WINOGRAD (unset = 1).
 * Workspace sizes do not depend on it.  Returns the previous value. */
int io_set_winograd(int on);
int io_get_winograd(void);
/* bf16 forward convolutions / data gradients on whole 256-row tiles with Cin % 64 == 0, Cout % 128 == 0 and a dense output:
 * 1 (default) = the persistent LDS-DMA kernels -- csrc/conv_p256.hip, and csrc/conv_halo3.hip for the 3x3 stride-1 launches
 * with 64 / 128 channels on 64- / 32-wide maps (input staged once per tile as a halo image); 2 = conv_p256 only; 3 = both, also for
 * launches with too few tiles to fill a round of persistent blocks (where 1 prefers the 128-row kernel: tests); 0 = the
 * 128-row kernel every other launch runs (same arithmetic, fp32 accumulation; sums may associate differently).
 * Process-wide; initial value from the environment variable IO_P256 (unset = 1).  Returns the previous value. */
int io_set_bf16_p256(int on);
/* Further environment switches of the launchers (read once per process; tuning / A-B runs, results identical up to summation order):
 *   IO_NT_SMALL_TILES=n   conv_nt_kernel: a forward / data-gradient launch with at most n 128-wide tiles runs 64-wide tiles (default
 *                         256 = one tile per CU: +3 % at 32 pairs per GPU; 0 = always 128 wide where Cout allows)
 *   IO_HALO3_BREG=0       conv_halo3_kernel: filters streamed through LDS instead of held in registers (64 -> 64 channels)
 *   IO_P256_XOP=0         see io_set_bf16_p256_xop below
 *   IO_DEPTH_STREAMS=0 | force, IO_DEPTH_FORK=0, IO_DEPTH_DIRECT_GRADS=0, IO_COMM_OVERLAP=0 | force, IO_NO_GRAPH=1
 *                         host-side switches of the Python wrappers (midas_net.py, ops.py, supervised_order.py) */
int io_get_bf16_p256(void);
/* Test hook: the kernel family the most recent forward / data-gradient launch of this process went to -- 0 = the 128-row
 * kernel, 1 = conv_p256, 2 = conv_halo3, 3 = stem_halo (so a parity test can assert that the kernel it means to check is the one that ran). */
int io_debug_last_nt_route(void);

/* ---- convolutions (implicit GEMM on v_mfma_f32_32x32x2_f32) --------------------------------
 * nn.Conv2d(bias=False) forward as used by conv1x1 / conv3x3 / the 7x7 stem
 * (models/backbone/resnet_cls.py:23-31, :140).  x[N,H,W,Cin], w[Cout][R*S][Cin],
 * y[N,Ho,Wo,Cout], Ho = (H + 2*pad - R)/stride + 1.  Cin % 32 == 0 (or Cin == 8 for the stem,
 * R = S = 7), Cout % 64 == 0. */
int io_conv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int R,
                  int S, int stride, int pad, hipStream_t stream);
/* the same convolution with the training-mode BatchNorm statistics of its output (see
 * io_bn_stats_finalize) accumulated in the GEMM epilogue instead of a second pass over y: what
 * `out = self.bnK(self.convK(x))` needs before the normalisation (resnet_cls.py:100-110).  N*Ho*Wo/G must be
 * a multiple of 128 (a row tile never straddles two BN groups). */
size_t io_conv2d_bnstats_workspace_floats(int N, int H, int W, int Cout, int R, int S, int stride, int pad, int G);
int io_conv2d_fwd_bnstats(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int R,
                          int S, int stride, int pad, int G, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* mean,
                          float* rstd, float* scale, float* shift, float* workspace, size_t workspace_floats,
                          hipStream_t stream);
/* gradient w.r.t. the input (autograd of the above, models/supervised_order.py:545 loss.backward()).
 * wt is the transposed filter [Cin][R*S][Cout] produced by io_filter_transpose; when `add` is not
 * NULL it is summed into the result (residual / accumulation; may alias dx); when `relu_mask` is not
 * NULL (a tensor shaped like dx: the post-ReLU activation this gradient belongs to) the result is zeroed
 * where relu_mask <= 0, i.e. the ReLU backward of Bottleneck.forward's final nn.ReLU (resnet_cls.py:114)
 * is applied in the epilogue. */
int io_conv2d_dgrad(const float* dy, const float* wt, float* dx, const float* add, const float* relu_mask, int N,
                    int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, hipStream_t stream);
/* Backward through `conv(relu(bn(y)))` for a stride-1 conv, fused: the data gradient of the conv, the ReLU
 * mask recomputed from y (scale/shift = the forward's tables), and the BatchNorm backward whose two
 * reductions ride in the conv epilogue.  dz[N,H,W,Cin] gets the masked gradient of bn's output, dyb the
 * gradient of y, dgamma/dbeta [Cin].  N*H*W/G must be a multiple of 128.  workspace: at least
 * 2*((tiles + tiles/64 + G + 2)*Cin) + 2*G*Cin floats with tiles = N*H*W/128. */
int io_conv2d_dgrad_bnbwd(const float* dy, const float* wt, float* dz, int N, int H, int W, int Cin, int Cout, int R,
                          int S, int pad, const float* y, int G, const float* gamma, const float* mean,
                          const float* rstd, const float* scale, const float* shift, float* dgamma, float* dbeta,
                          float* dyb, float* workspace, size_t workspace_floats, hipStream_t stream);
/* gradient w.r.t. the filter; workspace from io_conv2d_wgrad_workspace_bytes (split-K partials). */
size_t io_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad);
int io_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout, int R,
                    int S, int stride, int pad, void* workspace, size_t workspace_bytes, hipStream_t stream);
int io_filter_transpose(const float* w, int Cout, int taps, int Cin, float* wt, hipStream_t stream);

/* ---- BatchNorm2d (resnet_cls.py:142, :87-92, :189), G independent statistic groups ----------- */
size_t io_bn_partial_floats(int M, int C, int G);
/* training statistics of y[M][C] (M = N*H*W rows, G consecutive equal groups): writes per-group
 * mean, rstd, scale = gamma*rstd, shift = beta ([G][C] each) and advances the running
 * estimates once per group, in group order (momentum 0.1 / unbiased variance in the reference). */
int io_bn_stats_finalize(const float* y, int M, int C, int G, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, float* mean,
                         float* rstd, float* scale, float* shift, float* partial, size_t partial_floats,
                         hipStream_t stream);
/* eval mode: mean/scale/shift [C] from the running estimates */
int io_bn_eval_prepare(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* mean, float* scale, float* shift,
                       hipStream_t stream);
/* out = [relu]( (y-mean)*scale + shift  [+ identity | + (identity-mean2)*scale2 + shift2] ): BN
 * (+ residual add of Bottleneck.forward, resnet_cls.py:96-116, with or without the downsample BN)
 * (+ nn.ReLU). */
int io_bn_apply(const float* y, int M, int C, int G, int per_group_tables, const float* mean, const float* scale,
                const float* shift, const float* identity, const float* mean2, const float* scale2,
                const float* shift2, int relu, float* out, hipStream_t stream);
/* backward of [ReLU o] BN: dz = dout*[a>0] where the post-ReLU value a is either read from `act`, or
 * recomputed from y with the forward's own scale/shift tables (mask_scale/mask_shift: saves re-reading the
 * activation when no residual is added before the ReLU), or absent (all three NULL: no ReLU behind this BN).
 * dgamma/dbeta [C] summed over groups, dy = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)); dz_out
 * (optional, may alias dout) gets dz.  coef: 2*G*C floats of scratch. */
int io_bn_bwd(const float* dout, const float* act, const float* mask_scale, const float* mask_shift, const float* y,
              int M, int C, int G, const float* gamma, const float* mean, const float* rstd, float* dgamma,
              float* dbeta, float* dy, float* dz_out, float* partial, size_t partial_floats, float* coef,
              hipStream_t stream);

/* ---- pooling / heads ------------------------------------------------------------------------
 * nn.MaxPool2d(3, 2, 1) (resnet_cls.py:144); idx: one byte per output element (packed x4). */
int io_maxpool_fwd(const float* x, int N, int H, int W, int C, float* out, uint32_t* idx, hipStream_t stream);
int io_maxpool_bwd(const float* dy, const uint32_t* idx, int N, int H, int W, int C, float* dx,
                   hipStream_t stream);
/* AdaptiveAvgPool2d(1) + flatten + fc | (fc_occ, fc_depth) (resnet_cls.py:152-160, 214-222);
 * logits[N][K0+K1] (head 0 first). */
int io_avgpool_fc_fwd(const float* x, int N, int HW, int C, const float* w0, const float* b0, int K0,
                      const float* w1, const float* b1, int K1, float* pooled, float* logits, hipStream_t stream);
/* relu_mask (optional, shaped like x): dx is zeroed where relu_mask <= 0 (ReLU backward of the pooled tensor) */
int io_avgpool_fc_bwd(const float* dlogits, const float* pooled, int N, int HW, int C, const float* w0, int K0,
                      const float* w1, int K1, const float* relu_mask, float* dx, float* dw0, float* db0, float* dw1,
                      float* db1, hipStream_t stream);

/* ---- input packing: torch.cat([modal_a, modal_b, rgb], 1) (supervised_order.py:537-538) fused with
 * NCHW -> NHWC and channel padding 5 -> 8.  planes/sample_strides are HOST arrays of nplanes entries. */
int io_pack_planes_nhwc8(const float* const* planes, const long* sample_strides, int nplanes, int N, int H, int W,
                         float* out, hipStream_t stream);

/* ---- losses: sigmoid+BCELoss and softmax+CrossEntropyLoss over both mask orders, /world_size
 * (supervised_order.py:59-95, 413-438, 481-493, 535-548).  N = G*B rows (direction-major),
 * logits[N][Kocc+Kdep]; occ_target[N][2] fp32; depth_target[N] int64; is_overlap[B] int64 or NULL
 * (NULL: plain mean).  losses[3] = {total/world, occlusion, depth}; dlogits may be NULL. */
int io_order_loss(const float* logits, int N, int B, int Kocc, int Kdep, const float* occ_target,
                  const long* depth_target, const long* is_overlap, float overlap_weight, float distinct_weight,
                  float inv_world, float* losses, float* dlogits, hipStream_t stream);

/* ---- torch.optim.SGD(momentum=0.9, weight_decay) step over a flat buffer (single_stage_model.py:35-38) */
int io_sgd_momentum(float* params, const float* grads, float* momentum_buf, size_t n, float lr, float momentum,
                    float weight_decay, hipStream_t stream);

/* ---- whole-network executor: resnet50_cls(in_channels=5, num_classes=K | [K0,K1])
 * (resnet_cls.py:259-268) forward / backward over caller-owned flat buffers. ----------------- */
typedef struct io_net io_net;

typedef struct io_tensor_info {
    char name[64];      /* state_dict key without the "module." prefix              */
    int kind;           /* 0 conv filter, 1 bn weight, 2 bn bias, 3 fc weight, 4 fc bias */
    int ndim;
    long shape[4];      /* logical (torch) shape: OIHW for filters                   */
    long offset;        /* float offset into the flat parameter / gradient buffer    */
    long numel_storage; /* floats occupied (stem filter is stored with 8 channels)   */
    int cin_storage;    /* innermost storage channels (8 for the stem, else Cin)     */
    int bn_index;       /* for bn tensors: which BN (running stats offset = index)   */
    long running_offset; /* float offset of this BN's running_mean in the running buffer; var follows at +C */
} io_tensor_info;

io_net* io_net_create(int in_channels, int n_heads, const int* head_dims);
void io_net_destroy(io_net* net);
long io_net_param_floats(const io_net* net);    /* size of the flat parameter buffer        */
long io_net_running_floats(const io_net* net);  /* size of the flat running-statistics buffer */
int io_net_num_tensors(const io_net* net);
int io_net_tensor_info(const io_net* net, int i, io_tensor_info* out);
int io_net_num_logits(const io_net* net);
/* bytes of workspace needed for a forward(/backward) of N samples of size S x S */
size_t io_net_workspace_bytes(const io_net* net, int N, int S, int training);
/* Byte offset, inside a TRAINING workspace of that shape, of an activation the forward pass keeps (for layer-wise
 * parity checks against resnet_cls.py:203-222): which = 0 conv1 output [N,S/2,S/2,64], (1 = relu(bn1(.)) is not
 * stored any more: -1), 2 max-pool
 * output [N,S/4,S/4,64], 3+i output of bottleneck i (0..15), NHWC, element type = the net's dtype.  -1 on error. */
long io_net_activation_offset(const io_net* net, int N, int S, int which);
/* x8[N,S,S,8] (element type = the net's dtype: float, or bf16 after io_net_set_dtype(net, 1)) -> logits[N][K].
 * training != 0: batch statistics in G groups, running stats updated, activations kept in `workspace` for
 * io_net_backward. */
int io_net_forward(io_net* net, const float* params, float* running, const void* x8, int N, int S, int G,
                   int training, void* workspace, size_t workspace_bytes, float* logits, hipStream_t stream);
/* Inference forward on H x W inputs (both multiples of 32): the eval path of io_net_forward (folded BatchNorm, running
 * statistics) for inputs that are not square -- resnet_cls.py:199-222 is fully convolutional up to its AdaptiveAvgPool, and
 * the reference's 'orig' inference mode (inference.py:401-407) feeds it whole images at their own aspect ratio.
 * x8[N,H,W,8] in the net's storage type; workspace of io_net_workspace_bytes_hw(net, N, H, W) bytes. */
size_t io_net_workspace_bytes_hw(const io_net* net, int N, int H, int W);
int io_net_forward_eval_hw(io_net* net, const float* params, float* running, const void* x8, int N, int H, int W,
                           void* workspace, size_t workspace_bytes, float* logits, hipStream_t stream);
/* gradient of every parameter into grads (same layout as params; fully overwritten).  Must follow a
 * training io_net_forward with the same x8 / N / S / G / workspace. */

/* The same pass cut into io_net_backward_num_stages() stages in execution order -- 0: heads + layer4, 1: layer3,
 * 2: layer2, 3: layer1 + stem -- running the stages [stage_lo, stage_hi).  The parameter gradients of a stage are final
 * when its call returns (stream order), so a data-parallel caller can start the exchange of that contiguous slice of
 * `grads` (utils/distributed_utils.py:27-31 average_gradients) while the next stage computes.  Calling every stage once, in order,
 * with the same arguments IS io_net_backward: nothing is carried between the calls but `workspace`.
 * CONTRACT: the stages of one pass must be enqueued IN ORDER (0, 1, 2, 3; a call may cover several consecutive ones) on
 * ONE workspace, after the training io_net_forward they belong to and with nothing else writing that workspace in
 * between: a later stage reads what the earlier ones left there (the transposed filters made by stage 0, the gradient
 * of the stage's input, BatchNorm tile partials).  The library cannot check this -- the state lives in device memory and
 * no entry point synchronises -- so a call with stage_lo > 0 on a fresh or reused workspace returns IO_OK and garbage. */
int io_net_backward_num_stages(const io_net* net);
int io_net_backward_stages(io_net* net, const float* params, float* grads, const void* x8, const float* dlogits, int N,
                           int S, int G, void* workspace, size_t workspace_bytes, int stage_lo, int stage_hi,
                           hipStream_t stream);
int io_net_backward(io_net* net, const float* params, float* grads, const void* x8, const float* dlogits, int N,
                    int S, int G, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* ---- bf16 configuration (BASELINE configs[2], [3]) ------------------------------------------------------
 * dtype 0 = fp32 (everything above), 1 = bf16: activations, activation gradients and GEMM operands are bf16
 * (v_mfma_f32_32x32x16_bf16, fp32 accumulate); parameters and their gradients, BatchNorm statistics, losses and
 * the optimiser stay fp32; the packed network input x8 is bf16 as well (masks are exact, the normalised image
 * rounds to 8 bits of mantissa).  The `_dt` entry points are the storage-typed forms of the functions of the same
 * name; `void*` tensors have the element type `dtype` says. */
#define IO_DTYPE_F32 0
#define IO_DTYPE_BF16 1
int io_net_set_dtype(io_net* net, int dtype);
int io_net_get_dtype(const io_net* net);
int io_pack_planes_nhwc8_dt(const float* const* planes, const long* sample_strides, int nplanes, int N, int H, int W,
                            void* out, int dtype, hipStream_t stream);
int io_conv2d_fwd_dt(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                     int stride, int pad, int dtype_in, int dtype_out, hipStream_t stream);
int io_conv2d_dgrad_dt(const void* dy, const void* wt, void* dx, const void* add, const void* relu_mask, int N, int H,
                       int W, int Cin, int Cout, int R, int S, int stride, int pad, int dtype, hipStream_t stream);
int io_conv2d_wgrad_dt(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int Cout, int R, int S,
                       int stride, int pad, void* workspace, size_t workspace_bytes, int dtype_in, int dtype_dy,
                       hipStream_t stream);
/* fp32 master filter [Cout][taps][Cin] -> GEMM operand of `dtype`: a plain cast (transpose = 0, bf16 only) or the
 * transposed filter [Cin][taps][Cout] the data gradient reads (transpose = 1). */
int io_filter_prepare(const float* w, int Cout, int taps, int Cin, void* dst, int transpose, int dtype,
                      hipStream_t stream);
int io_bn_stats_finalize_dt(const void* y, int M, int C, int G, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* mean,
                            float* rstd, float* scale, float* shift, float* partial, size_t partial_floats, int dtype,
                            hipStream_t stream);
/* io_bn_apply_dt with relu != 0 that also writes the sign mask of its output, one bit per element: bits[(m * C + c) / 32] bit
 * c % 32 = (out[m][c] > 0); C % 32 == 0, M * C / 32 words.  What the executor keeps of a block output for the ReLU backward
 * (resnet_cls.py:114 `out = self.relu(out)`) instead of re-reading the tensor in the data gradient that completes d(out). */
int io_bn_apply_bits_dt(const void* y, int M, int C, int G, int per_group_tables, const float* mean, const float* scale,
                        const float* shift, const void* identity, const float* mean2, const float* scale2,
                        const float* shift2, void* out, uint32_t* bits, int dtype, hipStream_t stream);
int io_bn_apply_dt(const void* y, int M, int C, int G, int per_group_tables, const float* mean, const float* scale,
                   const float* shift, const void* identity, const float* mean2, const float* scale2,
                   const float* shift2, int relu, void* out, int dtype, hipStream_t stream);
int io_bn_bwd_dt(const void* dout, const void* act, const float* mask_scale, const float* mask_shift, const void* y,
                 int M, int C, int G, const float* gamma, const float* mean, const float* rstd, float* dgamma,
                 float* dbeta, void* dy, void* dz_out, float* partial, size_t partial_floats, float* coef, int dtype,
                 hipStream_t stream);

/* ---- operators of the MiDaS branch (InstaDepthNet_od / _d: midas/midas_net.py:116-212, midas/blocks.py:71-195) ----
 * NHWC; `dtype` (IO_DTYPE_F32 | IO_DTYPE_BF16) is the element type of every `void*` tensor; parameters, their
 * gradients, biases and reduction results stay float.
 * Grouped 3x3 convolution of the ResNeXt-101 32x8d encoder (resnet_cls.py:309-320, `groups=32`): the filter
 * w[C][cg][R*S] (OIHW, cg = C/groups input channels per group, cg | 64) is expanded by io_gconv_pack to two
 * block-diagonal operands over 64-channel windows -- wc[C][R*S][64] for the forward, wtc[C][R*S][64] for the data
 * gradient -- and the filter gradient comes back (float) in the wc layout (io_gconv_unpack_grad extracts
 * [C][cg][R*S]). */
int io_gconv_pack(const float* w, int C, int cg, int taps, void* wc, void* wtc, int dtype, hipStream_t stream);
int io_gconv_unpack_grad(const float* dwc, int C, int cg, int taps, float* dw, hipStream_t stream);
int io_gconv2d_fwd(const void* x, const void* wc, void* y, int N, int H, int W, int C, int R, int S, int stride, int pad,
                   int dtype, hipStream_t stream);
int io_gconv2d_dgrad(const void* dy, const void* wtc, void* dx, int N, int H, int W, int C, int R, int S, int stride,
                     int pad, int dtype, hipStream_t stream);
size_t io_gconv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int R, int S, int stride, int pad);
int io_gconv2d_wgrad(const void* x, const void* dy, float* dwc, int N, int H, int W, int C, int R, int S, int stride,
                     int pad, void* workspace, size_t workspace_bytes, int dtype, hipStream_t stream);
/* nn.functional.interpolate(scale_factor=2, mode='bilinear', align_corners=...) (midas/blocks.py:111-113, 186-188):
 * x[N,H,W,C] -> out[N,2H,2W,C]; bwd is its exact adjoint dy[N,2H,2W,C] -> dx[N,H,W,C]. */
int io_upsample2x_bilinear_fwd(const void* x, int N, int H, int W, int C, int align_corners, void* out, int dtype,
                               hipStream_t stream);
int io_upsample2x_bilinear_bwd(const void* dy, int N, int H, int W, int C, int align_corners, void* dx, int dtype,
                               hipStream_t stream);
/* out[M][C] = [relu](x + bias) (bias may be NULL; out may alias x): conv bias / nn.ReLU of midas/blocks.py:121-160 */
int io_bias_act(const void* x, const float* bias, int M, int C, int relu, void* out, int dtype, hipStream_t stream);
/* dx = dy * [act > 0] (dx may alias dy) */
int io_relu_bwd(const void* dy, const void* act, size_t n, void* dx, int dtype, hipStream_t stream);
int io_add(const void* a, const void* b, size_t n, void* out, int dtype, hipStream_t stream);
/* out[c] = sum_m x[m][c] (bias gradient); C must divide 256; partial: io_colsum_partial_floats(M, C) floats */
size_t io_colsum_partial_floats(int M, int C);
int io_colsum(const void* x, int M, int C, float* out, float* partial, size_t partial_floats, int dtype,
              hipStream_t stream);
/* nn.Conv2d(C, 1, 1) [+ nn.ReLU] (midas_net.py:139-140): out[m] (float) = act(b + sum_c x[m*pitch + c] * w[c]); the
 * input may carry padding channels (pitch >= C, <= 64).  bwd: dx[M][pitch] (zero in the padding channels), dw[C],
 * db[1]; partial: io_colsum_partial_floats(M, C) floats. */
int io_head1_fwd(const void* x, int M, int pitch, int C, const float* w, const float* b, int relu, float* out, int dtype,
                 hipStream_t stream);
int io_head1_bwd(const float* dy, const float* out, const void* x, int M, int pitch, int C, const float* w, int relu,
                 void* dx, float* dw, float* db, float* partial, size_t partial_floats, int dtype, hipStream_t stream);
/* ---- the two MiDaS-branch losses that are not order-head losses ---------------------------------------------------------
 * Edge-aware smoothness (models/supervised_order.py:214-235 get_smooth_loss): disp [B,H,W] fp32 is min-max normalised
 * ((d - min) / (max + 1e-7)), divided by its mean (+1e-7), and loss = mean(|dx n| exp(-mean_c |dx img|)) + the same in y;
 * img [B,3,H,W] NCHW fp32.  loss[0] = out_scale * that; g [B,H,W] and workspace keep what the backward needs.
 * io_smooth_loss_bwd: ddisp (+)= grad_out[0] * scale * dloss/ddisp (grad_out on the device: no host round trip), through
 * both normalisations incl. the min / max elements torch's chained min(2).min(3) selects. */
size_t io_smooth_loss_workspace_floats(int B, int H, int W);
int io_smooth_loss_fwd(const float* disp, const float* img, int B, int H, int W, float out_scale, float* loss, float* g,
                       float* workspace, size_t workspace_floats, hipStream_t stream);
int io_smooth_loss_bwd(const float* g, const float* workspace, const float* grad_out, float scale, int B, int H, int W,
                       int accumulate, float* ddisp, hipStream_t stream);
/* Disparity-order count (supervised_order.py:152-173; a pure count, no gradient): per pair with is_overlap == 0 and
 * depth_order1 in {0, 1}, over the 3x3-cross erosions (scipy.ndimage.binary_erosion, border 0) e1, e2 of the two masks:
 * #{d[e1] <= max d[e2]} + #{min d[e1] <= d[e2]} on disp1 (with >= when depth_order1 != le_order) and the opposite
 * comparison on disp2; pairs whose eroded masks are empty are skipped.  out[0] = out_scale * total / (H * W);
 * workspace: io_disp_order_workspace_floats(B, H, W) floats of scratch (per-block extrema and counts: a sample is spread
 * over several blocks, three launches). */
size_t io_disp_order_workspace_floats(int B, int H, int W);
int io_disp_order_count(const float* disp1, const float* disp2, const float* modal1, const float* modal2,
                        const long* depth_order1, const long* is_overlap, int B, int H, int W, int le_order,
                        float out_scale, float* out, float* workspace, size_t workspace_floats, hipStream_t stream);
/* io_conv2d_fwd_bnstats for either storage type and for the grouped window form (gw = 64; w = wc of io_gconv_pack,
 * Cin == Cout); gw = 0 is the dense convolution */
int io_conv2d_fwd_bnstats_dt(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                             int stride, int pad, int G, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, float* mean, float* rstd, float* scale,
                             float* shift, float* workspace, size_t workspace_floats, int dtype, int gw,
                             hipStream_t stream);
/* io_conv2d_dgrad_bnbwd for either storage type (dy, wt, dz, y, dyb of `dtype`) and for the grouped window form (gw = 64:
 * wt = wtc of io_gconv_pack, Cin == Cout); gw = 0 is the dense convolution */
int io_conv2d_dgrad_bnbwd_dt(const void* dy, const void* wt, void* dz, int N, int H, int W, int Cin, int Cout, int R, int S,
                             int pad, const void* y, int G, const float* gamma, const float* mean, const float* rstd,
                             const float* scale, const float* shift, float* dgamma, float* dbeta, void* dyb,
                             float* workspace, size_t workspace_floats, int dtype, int gw, hipStream_t stream);
/* The 7x7 / 2 stem (resnet_cls.py:140) in EXACT-K mode, fp32 -- what the executor launches for the fp32 step: the packed
 * x8 input carries `real_channels` (1..8, here 5) real channels, and the reduction runs over k = tap * real_channels +
 * channel only (245 instead of 392 products per output).  w / dw: [64][49][8] as for io_conv2d_fwd / io_conv2d_wgrad
 * (pad channels of dw are written as zeros); packed: scratch of io_stem_packed_floats() floats. */
size_t io_stem_packed_floats(int real_channels);
size_t io_stem_wgrad_exact_workspace_bytes(int N, int H, int W, int real_channels);
int io_stem_fwd_bnstats_exact(const float* x8, const float* w, float* y, int N, int H, int W, int real_channels, int G,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* mean, float* rstd, float* scale, float* shift,
                              float* workspace, size_t workspace_floats, float* packed, hipStream_t stream);
int io_stem_wgrad_exact(const float* x8, const float* dy, float* dw, int N, int H, int W, int real_channels,
                        void* workspace, size_t workspace_bytes, float* packed, hipStream_t stream);
/* The same gradient with bn1's backward (resnet_cls.py:157: the BatchNorm behind the stem) folded into it: da[N,H/2,W/2,64]
 * is the gradient of relu(bn1(y)) -- what the max-pool backward leaves -- and y the raw stem output.  One reduction pass
 * produces dgamma / dbeta and the coefficient tables coef[3][G][64] (a, b, c of dy = a * dz + b * y + c, dz = da where
 * scale * (y - mean) + shift > 0); the gradient kernel applies them while it stages its rows, so dy is never written.
 * Needs 5 real channels, 256 | H, W and G | N (IO_ERR_SHAPE otherwise: use io_bn_bwd + io_stem_wgrad_exact).
 * bn_partial: io_bn_partial_floats(N * H/2 * W/2, 64, G) floats; workspace as io_stem_wgrad_exact. */
int io_stem_wgrad_exact_bn(const float* x8, const float* da, const float* y, float* dw, int N, int H, int W,
                           int real_channels, int G, const float* gamma, const float* mean, const float* rstd,
                           const float* scale, const float* shift, float* dgamma, float* dbeta, float* coef,
                           float* bn_partial, size_t bn_partial_floats, void* workspace, size_t workspace_bytes,
                           float* packed, hipStream_t stream);

/* ---- BatchNorm backward without an apply pass: the training step's fused data-gradient launch ---------------------------
 * autograd of `out = conv(relu(bn(y)))` chains (models/backbone/resnet_cls.py:99-113, loss.backward() at
 * models/supervised_order.py:545).  The BatchNorm input gradient dy = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)) is
 * an affine function of (dz, y) per channel: dy = a[g][c]*dz + b[g][c]*y + c[g][c].  io_bn_bwd_coefs* produce dgamma /
 * dbeta and the three [G][C] tables (coef: 3*G*C floats, a | b | c) -- from (dz, y) by a reduction pass, or from the
 * per-tile partial sums p1 / p2 that the epilogue of io_conv2d_dgrad_fused_dt left behind (each array
 * io_bn_tile_partial_floats(M, C, G) floats; M/G a multiple of 128) -- and io_conv2d_dgrad_fused_dt consumes them: its A
 * operand is evaluated from dz and y while it is staged (two loads per chunk instead of one), and the blocks of the
 * first output-channel tile write dy out once for the filter gradient.  dz must already carry its ReLU mask. */
size_t io_bn_tile_partial_floats(int M, int C, int G);
int io_bn_bwd_coefs_dt(const void* dz, const void* y, int M, int C, int G, const float* gamma, const float* mean,
                       const float* rstd, float* dgamma, float* dbeta, float* coef, float* partial,
                       size_t partial_floats, int dtype, hipStream_t stream);
int io_bn_bwd_coefs_from_tile_partials(float* p1, float* p2, int M, int C, int G, const float* gamma, const float* mean,
                                       const float* rstd, float* dgamma, float* dbeta, float* coef, hipStream_t stream);
/* Everything one data-gradient launch of the training step can carry (all optional; zero-initialise the struct):
 *  operand side   xb_y / xb_coef / xb_dy_out: `dy` is the masked gradient dz of the BatchNorm output behind the convolution,
 *                 xb_y that BatchNorm's input, xb_coef its tables (above); xb_dy_out (optional) receives dy [N,Ho,Wo,Cout].
 *                 Needs a stride-1 same-size convolution (1x1, or 3x3 pad 1) and M/G a multiple of 128.
 *  epilogue       add (may alias dx) / relu_mask as io_conv2d_dgrad;
 *                 ep_y: the BatchNorm whose OUTPUT gradient dx is (y = its input [N,H,W,Cin], ep_mean / ep_rstd [G][Cin]):
 *                 per-tile sums of dx and dx*xhat go to ep_p1 / ep_p2 for io_bn_bwd_coefs_from_tile_partials; with
 *                 ep_scale / ep_shift dx is first masked by [relu(bn(y)) > 0] recomputed from y, and ep_act_out
 *                 (optional) receives relu(bn(y)).  Needs M/G a multiple of 128. */
typedef struct io_dgrad_fused {
    const void* xb_y;
    const float* xb_coef;
    void* xb_dy_out;
    const void* add;
    const void* relu_mask;
    const void* ep_y;
    const float *ep_mean, *ep_rstd, *ep_scale, *ep_shift;
    void* ep_act_out;
    float *ep_p1, *ep_p2;
    /* optional: io_conv2d_wino_scratch_floats(Cin, Cout) floats of device scratch -- a 3x3 pad-1 fp32 data gradient without
     * add / relu_mask / xb_* on whole 128-row tiles with even W then runs in the Winograd F(2, 3) row form (2/3 of the
     * MFMAs; results differ from the direct form at rounding level) */
    float* wino_scratch;
    size_t wino_scratch_floats;
    /* optional, NEXT TO relu_mask (same information): the mask as ONE BIT per element -- word (m * Cin + c) / 32, bit c % 32
     * set where the masking activation was > 0 (io_bn_apply_bits_dt writes it beside the activation).  The bf16 256-row
     * kernel reads it instead of the tensor (1/16 of the bytes); every other launch reads relu_mask.  Cin % 32 == 0. */
    const uint32_t* relu_maskbits;
} io_dgrad_fused;
int io_conv2d_dgrad_fused_dt(const void* dy, const void* wt, void* dx, int N, int H, int W, int Cin, int Cout, int R,
                             int S, int pad, int G, const io_dgrad_fused* f, int dtype, hipStream_t stream);
/* conv(relu((x - in_mean[g][c]) * in_scale[g][c] + in_shift[g][c]), w): the BatchNorm + ReLU between two convolutions of a Bottleneck
 * (models/backbone/resnet_cls.py:99-111: out = relu(bn1(conv1(x))); out = conv2(out)) applied to the operand of the
 * SECOND convolution while it is staged, so relu(bn1(.)) is never written to memory.  Padding is zero AFTER the
 * transform.  in_mean / in_scale / in_shift [G][Cin] are the mean / scale / shift tables io_bn_stats_finalize /
 * io_conv2d_fwd_bnstats produce for the first BatchNorm (the expression is io_bn_apply's; in_mean may be NULL = 0).  workspace != NULL: also the training statistics of y, exactly as
 * io_conv2d_fwd_bnstats_dt (gamma .. shift then describe the BatchNorm AFTER this convolution); workspace == NULL: they
 * are ignored.  Output rows per group must be a multiple of 128; Cin a multiple of 32 (fp32) / 64 (bf16). */
/* Winograd F(2, 3) row form of the 3x3 stride-1 pad-1 fp32 convolution (resnet_cls.py:23-26 conv3x3): the arguments of
 * io_conv2d_fwd_xf_dt (in_scale / in_shift NULL: no input transform; workspace NULL: no statistics) + device scratch of
 * io_conv2d_wino_scratch_floats(Cin, Cout) floats for the transformed filters.  A pair of horizontally adjacent outputs
 * costs 4 products per filter row instead of 6; the result differs from the direct form at fp32 rounding level only.
 * Needs even W and N*H*W/G a multiple of 128 (other shapes silently run the direct kernel). */
size_t io_conv2d_wino_scratch_floats(int Cin, int Cout);
int io_conv2d_fwd_wino(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int G,
                       const float* in_mean, const float* in_scale, const float* in_shift, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                       float* rstd, float* scale, float* shift, float* workspace, size_t workspace_floats,
                       float* wino_scratch, size_t wino_floats, hipStream_t stream);
int io_conv2d_fwd_xf_dt(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                        int stride, int pad, int G, const float* in_mean, const float* in_scale, const float* in_shift,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                        float* mean, float* rstd, float* scale, float* shift, float* workspace, size_t workspace_floats,
                        int dtype, hipStream_t stream);
/* y = conv1x1(out, w) with out = relu((y3 - in_mean[g]) * in_scale[g] + in_shift[g] + identity) evaluated on the staged
 * operand and written to `out` (may be NULL) by the blocks of the first output-channel tile: the residual BatchNorm +
 * ReLU that ends a Bottleneck (resnet_cls.py:108-114) fused into the first convolution of the next one (:99) -- io_bn_apply's
 * expression, bit for bit.  fp32; y3 / identity / out [N,H,W,Cin]; N*H*W/G a multiple of 128.  workspace != NULL: also the
 * training statistics of y (as io_conv2d_fwd_xf_dt). */
int io_conv2d_fwd_resid(const float* y3, const float* identity, const float* w, float* y, float* out, int N, int H, int W,
                        int Cin, int Cout, int G, const float* in_mean, const float* in_scale, const float* in_shift,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                        float eps, float* mean, float* rstd, float* scale, float* shift, float* workspace,
                        size_t workspace_floats, hipStream_t stream);
/* The same fusion for either storage type, with the second table form and the one-bit mask (round 6): operand =
 *   two == 0:  relu((y3 - tb[g][c]) * ta[g][c] + tc[g][c] + second)   ta / tb / tc = scale / mean / shift of bn3, second = identity
 *   two != 0:  relu(ta[g][c] * y3 + tb[g][c] * second + tc[g][c])     a block with a downsample branch: second = the downsample
 *              convolution's raw output, the two BatchNorms folded into one table set (io_bn_resid2_tables)
 * written to `out` (optional) and, as [out > 0] one bit per element (word (m * Cin + c) / 32, bit c % 32; optional, bf16
 * 256-row kernel only), to out_bits.  tile_mean / tile_m2 (a pair, optional): per-(128-row tile, channel) mean / M2 of y,
 * io_bn_tile_partial_floats(M, Cout, G) floats each, for io_bn_finalize_tiles.  bf16: launches of whole 256-row tiles with
 * 64 | Cin, 128 | Cout and 256 | rows per group run on conv_p256_kernel, the transform applied IN LDS to each A k-tile
 * after its DMA has landed (IO_P256_XOP=0 / io_set_bf16_p256_xop(0): on conv_nt_kernel's staging registers instead). */
int io_conv2d_fwd_resid_dt(const void* y3, const void* second, const void* w, void* y, void* out, uint32_t* out_bits, int N,
                           int H, int W, int Cin, int Cout, int G, int two, const float* ta, const float* tb, const float* tc,
                           float* tile_mean, float* tile_m2, int dtype, hipStream_t stream);
/* Run-time switch of the in-LDS operand forms of the bf16 256-row kernel (process-wide; initial value from IO_P256_XOP,
 * unset = 1): with 0 the network executor keeps the stand-alone BatchNorm passes on layers 2-4 (the round-5 step) and a
 * launch that asks for an operand form runs on conv_nt_kernel.  Returns the previous value.  Like io_set_bf16_p256 it decides
 * which launch builds a block output and its one-bit mask: change either switch BETWEEN training steps, not between
 * io_net_forward and the io_net_backward* calls of one step. */
int io_set_bf16_p256_xop(int on);
int io_get_bf16_p256_xop(void);
/* All filters of a module tree in one launch (the op-by-op graphs of instaorder_amd.ops; midas/midas_net.py's ~200 dense
 * convolutions): `table` is a DEVICE array of n io_weight_desc.  io_weights_prepare writes, for every entry, the
 * [Cop][T][Cip] operand (element type dtype, zero where o >= Co or c >= Ci) of the OIHW fp32 master at params + src to
 * ops + dst_op and, if dst_t >= 0, its transpose [Cip][T][Cop] (the data-gradient operand) to ops + dst_t;
 * io_weights_unpack_grads copies the filter gradients [Cop][T][Cip] at gk + dst_g back to OIHW at grads + src. */
typedef struct io_weight_desc {
    long src, dst_op, dst_t, dst_g;
    int Co, Ci, T, Cop, Cip;
} io_weight_desc;
int io_weights_prepare(const void* table, int n, const float* params, void* ops, int dtype, hipStream_t stream);
int io_weights_unpack_grads(const void* table, int n, const float* gk, float* grads, hipStream_t stream);
/* bf16 counterpart (round 5; csrc/conv_halo3.hip: stem_wgrad_halo_kernel): x8 / da / y bf16, dw fp32 [64][49][8].  256 x 256 inputs
 * (128-wide output rows), G | N; workspace of io_stem_wgrad_bf16_workspace_bytes(); IO_ERR_SHAPE where the kernel does not apply
 * (other sizes, too few rows for its persistent blocks, io_set_bf16_p256(0)): use io_bn_bwd_dt + io_conv2d_wgrad_dt there.
 * The plain filter gradient of the bf16 stem (io_conv2d_wgrad_dt with Cin = 8) takes the same kernel without the fold when its
 * workspace (io_conv2d_wgrad_workspace_bytes) allows; io_debug_last_wgrad_route() = 1 tells a test that it did. */
size_t io_stem_wgrad_bf16_workspace_bytes(void);
int io_stem_wgrad_bn_bf16(const void* x8, const void* da, const void* y, float* dw, int N, int H, int W, int G,
                          const float* gamma, const float* mean, const float* rstd, const float* scale, const float* shift,
                          float* dgamma, float* dbeta, float* coef, float* bn_partial, size_t bn_partial_floats,
                          void* workspace, size_t workspace_bytes, hipStream_t stream);
int io_debug_last_wgrad_route(void);
/* The stem's pooling over a transformed input (see io_conv2d_fwd_xf_dt; tables [G][C], group = sample / (N / G)):
 * nn.MaxPool2d(3, 2, 1) over relu(bn1(x)) (resnet_cls.py:205-208) with the arg-max indices io_maxpool_bwd consumes. */
int io_maxpool_fwd_xf_dt(const void* x, int N, int H, int W, int C, void* out, uint32_t* idx, int G,
                         const float* in_mean, const float* in_scale, const float* in_shift, int dtype,
                         hipStream_t stream);
/* y = [relu](conv(x, w) + bias[o] (+ add)): a biased nn.Conv2d, or conv + eval-mode BatchNorm (+ residual) (+ ReLU) with
 * the BatchNorm folded into pre-scaled filters (w[o] * gamma[o]/sqrt(var[o]+eps), bias = beta - mean * that) --
 * resnet_cls.py:99-114 in model.eval().  add (optional) has the shape and type of y.  gw as above. */
int io_conv2d_fwd_bias_dt(const void* x, const void* w, void* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                          int stride, int pad, const float* bias, const void* add, int relu, int dtype, int gw,
                          hipStream_t stream);
/* storage-typed forms of io_maxpool_* / io_avgpool_fc_* (pooled / logits / parameter gradients stay float) */
int io_maxpool_fwd_dt(const void* x, int N, int H, int W, int C, void* out, uint32_t* idx, int dtype, hipStream_t stream);
int io_maxpool_bwd_dt(const void* dy, const uint32_t* idx, int N, int H, int W, int C, void* dx, int dtype,
                      hipStream_t stream);
int io_avgpool_fc_fwd_dt(const void* x, int N, int HW, int C, const float* w0, const float* b0, int K0, const float* w1,
                         const float* b1, int K1, float* pooled, float* logits, int dtype, hipStream_t stream);
int io_avgpool_fc_bwd_dt(const float* dlogits, const float* pooled, int N, int HW, int C, const float* w0, int K0,
                         const float* w1, int K1, const void* relu_mask, void* dx, float* dw0, float* db0, float* dw1,
                         float* db1, int dtype, hipStream_t stream);

/* ---- input pipeline on the device: crop with zero padding + resize + flip + normalise of one uint8
 * image and two uint8 instance masks per pair, into the fp32 tensors set_input() takes -- the per-item work of
 * SupOcclusionOrderDataset._get_pair / _get_pair_image / _get_pair_resize (datasets/occ_order_dataset.py:81-180,
 * utils/data_utils.py:105-124: crop_padding) and of the per-pair pre-processing of inference.py:449-482.
 * Masks: INTER_NEAREST; image: OpenCV's 8-bit fixed-point INTER_LINEAR (interp 1) or INTER_CUBIC (interp 2), then
 * x / 255, (x - mean) / std in fp32 (mean / std rounded to fp32 first, as transforms.Normalize does); interp 3:
 * INTER_CUBIC on the float64 image x / 255., normalised in float64 and rounded once -- MiDaS' Resize / NormalizeImage /
 * PrepareForNet chain of the 'resize' inference mode (utils/data_utils.py:37-53, midas/transforms.py:163-173, 206-222).  `arena` is one device buffer holding the decoded images (HxWx3) and masks (HxW), all
 * uint8, addressed by byte offsets; desc_dev / desc_host are the same P descriptors in device and host memory (the
 * host copy is validated against arena_bytes).  rgb[P][3][S][S] (NULL: masks only), modal1 / modal2 [P][S][S]. */
typedef struct io_pair_desc {
    int64_t image_off;            /* byte offset of the image in the arena                               */
    int64_t mask1_off, mask2_off; /* byte offsets of the two instance masks                              */
    int32_t H, W;                 /* image (and mask) size                                               */
    int32_t x, y, w, h;           /* crop rectangle in image coordinates; parts outside the image read 0 */
    int32_t flip;                 /* != 0: mirror the outputs horizontally                               */
    int32_t interp;               /* image interpolation: 1 linear, 2 cubic, 3 cubic in float64           */
} io_pair_desc;
int io_pair_planes_u8(const uint8_t* arena, size_t arena_bytes, const io_pair_desc* desc_dev,
                      const io_pair_desc* desc_host, int P, int S, const double* mean3, const double* std3, float* rgb,
                      float* modal1, float* modal2, hipStream_t stream);
/* ... onto SH x SW planes (rgb[P][3][SH][SW], modal1 / modal2 [P][SH][SW]): the 'orig' inference mode renders the whole
 * image at its own aspect ratio, sides rounded to multiples of 32 (inference.py:401-407, 490-496, 569-575). */
int io_pair_planes_u8_hw(const uint8_t* arena, size_t arena_bytes, const io_pair_desc* desc_dev,
                         const io_pair_desc* desc_host, int P, int SH, int SW, const double* mean3, const double* std3,
                         float* rgb, float* modal1, float* modal2, hipStream_t stream);

/* ---- measurement aid (bench.py): HIP-event timing of every launch, per kernel class, on the launch
 * stream.  Process-global; io_prof_end synchronises on the recorded events and returns the number of
 * classes written.  flops / bytes are the ALGORITHMIC figures of the timed launches (the direct convolution's count). */
typedef struct io_prof_entry {
    char name[48];
    long launches;
    double total_ms;
    double flops;
    double bytes;
    double flops_executed;   /* what the matrix pipes actually multiplied: = flops for the direct kernels, 1/2 (F(4,3)) or 2/3
                              * (F(2,3)) of it for the Winograd row forms of the 3x3 stride-1 convolutions */
} io_prof_entry;
int io_prof_begin(void);
/* share_events != 0 (what io_prof_begin uses): a launch group's start event is the previous group's end event -- half the
 * event records, right for the network executor, where the library's launches follow each other on the stream.
 * share_events == 0: every group gets its own start event -- for callers that put other work (kernels of their own, host
 * gaps) between the library's launches, which would otherwise be charged to the next class. */
int io_prof_begin_ex(int share_events);
int io_prof_end(io_prof_entry* out, int max_entries);
/* the launch groups recorded so far, one entry each (launches = 1), in launch order; the profile keeps running */
int io_prof_launches(io_prof_entry* out, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* INSTAORDER_HIP_H */
