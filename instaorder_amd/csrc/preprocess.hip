// Input pipeline in front of the order networks, on the device: for every instance pair of a batch, crop (with zero
// padding) + resize + flip + normalise of the uint8 image and of the two uint8 instance masks, straight into the
// fp32 tensors the wrappers' set_input() takes -- what SupOcclusionOrderDataset._get_pair / _get_pair_image /
// _get_pair_resize do per item with cv2 on the host (datasets/occ_order_dataset.py:81-180, utils/data_utils.py:105-124)
// and what inference.py:449-482 repeats per pair.  The host then ships the decoded uint8 image and masks once (a
// fraction of the 335 MB of fp32 planes per 256-pair batch) and no resized intermediate ever exists in memory.
//
// Resize arithmetic is OpenCV's 8-bit fixed-point one (resize.cpp: 11-bit coefficients, horizontal pass to int, vertical
// pass with the (x >> 4) * b >> 16 form for INTER_LINEAR and a 22-bit rounding shift for INTER_CUBIC, A = -0.75,
// taps clamped to the edge of the CROPPED image), evaluated per output pixel; it is bit-exact against
// oracle/preprocess_oracle.py, whose header states what that restatement is and is not pinned against.
// HBM-bound by construction (one thread per output pixel, 8 or 48 byte gathers from L2-resident crops, 20 bytes
// written); no LDS.
#include "io_common.h"

#pragma clang fp contract(off)   // coefficient arithmetic must round like the scalar C / numpy evaluation

namespace {

constexpr int kCoefBits = 11;

struct Taps {
    int idx[4];
    int coef[4];
    int n;
};

// destination index d of `dst` -> source taps over a `src`-wide axis (resizeGeneric_'s float / floor split)
__device__ __forceinline__ Taps make_taps(int d, int src, int dst, int interp) {
    Taps t;
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (interp == 1) {
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
        t.n = 2;
        t.idx[0] = s;
        t.idx[1] = min(s + 1, src - 1);
        t.coef[0] = __float2int_rn((1.f - f) * 2048.f);
        t.coef[1] = __float2int_rn(f * 2048.f);
        t.idx[2] = t.idx[3] = 0;
        t.coef[2] = t.coef[3] = 0;
    } else {
        const float A = -0.75f;
        const float x1 = f + 1.f;
        const float c0 = ((A * x1 - 5.f * A) * x1 + 8.f * A) * x1 - 4.f * A;
        const float c1 = ((A + 2.f) * f - (A + 3.f)) * f * f + 1.f;
        const float xr = 1.f - f;
        const float c2 = ((A + 2.f) * xr - (A + 3.f)) * xr * xr + 1.f;
        const float c3 = 1.f - c0 - c1 - c2;
        t.n = 4;
        t.coef[0] = __float2int_rn(c0 * 2048.f);
        t.coef[1] = __float2int_rn(c1 * 2048.f);
        t.coef[2] = __float2int_rn(c2 * 2048.f);
        t.coef[3] = __float2int_rn(c3 * 2048.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) t.idx[k] = min(max(s - 1 + k, 0), src - 1);
    }
    return t;
}

__device__ __forceinline__ int nearest_index(int d, int src, int dst) {
    const double scale = 1.0 / ((double)dst / (double)src);
    return min((int)floor((double)d * scale), src - 1);
}

// interp 3: INTER_CUBIC on the float64 image x / 255. (MiDaS' Resize transform, midas/transforms.py:163-173 via
// utils/data_utils.py:37-53): OpenCV's non-fixed-point path -- float coefficients, double accumulation left to right
// (HResizeCubic<double,double,float>, VResizeCubic<double,double,float>), then (x - mean) / std in double and one
// rounding to fp32 (PrepareForNet).
__device__ __forceinline__ void cubic_coefs_f(int d, int src, int dst, int (&idx)[4], float (&c)[4]) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    const float A = -0.75f;
    const float x1 = f + 1.f;
    c[0] = ((A * x1 - 5.f * A) * x1 + 8.f * A) * x1 - 4.f * A;
    c[1] = ((A + 2.f) * f - (A + 3.f)) * f * f + 1.f;
    const float xr = 1.f - f;
    c[2] = ((A + 2.f) * xr - (A + 3.f)) * xr * xr + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[k] = min(max(s - 1 + k, 0), src - 1);
}

__global__ __launch_bounds__(256) void pair_planes_kernel(const uint8_t* __restrict__ arena,
                                                         const io_pair_desc* __restrict__ desc, int SH, int S,
                                                         double m0, double m1, double m2, double s0, double s1, double s2,
                                                         float* __restrict__ rgb, float* __restrict__ modal1,
                                                         float* __restrict__ modal2) {
    const int p = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    // (S = output width, SH = output height; the square outputs of the training / 'patch' / 'image' / 'resize' paths
    // have SH == S, the 'orig' inference mode renders the whole image at its own aspect ratio)
    if (pix >= SH * S) return;
    const io_pair_desc d = desc[p];
    const int oy = pix / S, ox = pix - oy * S;
    const int dx = d.flip ? S - 1 - ox : ox;
    const size_t plane = (size_t)SH * S;

    // masks: nearest sample of the zero-padded crop
    {
        const int cx = nearest_index(dx, d.w, S), cy = nearest_index(oy, d.h, SH);
        const int ix = d.x + cx, iy = d.y + cy;
        const bool in = (unsigned)ix < (unsigned)d.W && (unsigned)iy < (unsigned)d.H;
        const size_t o = in ? (size_t)iy * d.W + ix : 0;
        modal1[p * plane + pix] = in ? (float)arena[d.mask1_off + o] : 0.f;
        modal2[p * plane + pix] = in ? (float)arena[d.mask2_off + o] : 0.f;
    }
    if (!rgb) return;

    const uint8_t* img = arena + d.image_off;
    const double mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    if (d.interp == 3) {
        int xi[4], yi[4];
        float xa[4], ya[4];
        cubic_coefs_f(dx, d.w, S, xi, xa);
        cubic_coefs_f(oy, d.h, SH, yi, ya);
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = d.y + yi[r];
            double hrow[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ix = d.x + xi[k];
                const bool in = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                const uint8_t* px = img + (in ? ((size_t)iy * d.W + ix) * 3 : 0);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double v = in ? (double)px[c] / 255.0 : 0.0;
                    hrow[c] = k == 0 ? v * (double)xa[0] : hrow[c] + v * (double)xa[k];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = r == 0 ? hrow[c] * (double)ya[0] : acc[c] + hrow[c] * (double)ya[r];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[(p * 3 + c) * plane + pix] = (float)((acc[c] - mean[c]) / sd[c]);
        return;
    }
    const Taps tx = make_taps(dx, d.w, S, d.interp), ty = make_taps(oy, d.h, SH, d.interp);
    int h[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h[r][0] = h[r][1] = h[r][2] = 0;
        if (r < ty.n) {
            const int iy = d.y + ty.idx[r];
            if ((unsigned)iy < (unsigned)d.H) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ix = d.x + tx.idx[k];
                    if (k < tx.n && (unsigned)ix < (unsigned)d.W) {
                        const uint8_t* px = img + ((size_t)iy * d.W + ix) * 3;
                        h[r][0] += (int)px[0] * tx.coef[k];
                        h[r][1] += (int)px[1] * tx.coef[k];
                        h[r][2] += (int)px[2] * tx.coef[k];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v;
        if (d.interp == 1) {
            v = (((ty.coef[0] * (h[0][c] >> 4)) >> 16) + ((ty.coef[1] * (h[1][c] >> 4)) >> 16) + 2) >> 2;
        } else {
            v = (ty.coef[0] * h[0][c] + ty.coef[1] * h[1][c] + ty.coef[2] * h[2][c] + ty.coef[3] * h[3][c] +
                 (1 << (2 * kCoefBits - 1))) >> (2 * kCoefBits);
        }
        v = min(max(v, 0), 255);
        const float x = __fdiv_rn((float)v, 255.f);
        rgb[(p * 3 + c) * plane + pix] = __fdiv_rn(__fsub_rn(x, (float)mean[c]), (float)sd[c]);
    }
}

}  // namespace

extern "C" int io_pair_planes_u8_hw(const uint8_t* arena, size_t arena_bytes, const io_pair_desc* desc_dev,
                                    const io_pair_desc* desc_host, int P, int SH, int SW, const double* mean3,
                                    const double* std3, float* rgb, float* modal1, float* modal2, hipStream_t st);
extern "C" int io_pair_planes_u8(const uint8_t* arena, size_t arena_bytes, const io_pair_desc* desc_dev,
                                 const io_pair_desc* desc_host, int P, int S, const double* mean3, const double* std3,
                                 float* rgb, float* modal1, float* modal2, hipStream_t st) {
    return io_pair_planes_u8_hw(arena, arena_bytes, desc_dev, desc_host, P, S, S, mean3, std3, rgb, modal1, modal2, st);
}

extern "C" int io_pair_planes_u8_hw(const uint8_t* arena, size_t arena_bytes, const io_pair_desc* desc_dev,
                                    const io_pair_desc* desc_host, int P, int SH, int SW, const double* mean3,
                                    const double* std3, float* rgb, float* modal1, float* modal2, hipStream_t st) {
    const int S = SW;
    IO_REQUIRE(P > 0 && SH > 0 && SW > 0 && arena && desc_dev && desc_host && modal1 && modal2, IO_ERR_SHAPE,
               "pair_planes: empty batch or null pointer (P=%d, %d x %d)", P, SH, SW);
    IO_REQUIRE(P <= 65535 && (long)SH * SW < (1L << 31), IO_ERR_SHAPE, "pair_planes: P=%d %d x %d out of range", P, SH, SW);
    IO_REQUIRE(!rgb || (mean3 && std3), IO_ERR_SHAPE, "pair_planes: rgb output needs mean / std");
    // the descriptors are validated on the host copy: every byte the kernel may touch lies inside the arena
    for (int p = 0; p < P; ++p) {
        const io_pair_desc& d = desc_host[p];
        const size_t hw = (size_t)(d.H > 0 ? d.H : 0) * (size_t)(d.W > 0 ? d.W : 0);
        IO_REQUIRE(d.H > 0 && d.W > 0 && d.w > 0 && d.h > 0, IO_ERR_SHAPE,
                   "pair_planes: pair %d has an empty image or crop (%dx%d, crop %dx%d)", p, d.H, d.W, d.w, d.h);
        IO_REQUIRE(d.interp >= 1 && d.interp <= 3, IO_ERR_SHAPE,
                   "pair_planes: pair %d interp=%d (1 linear, 2 cubic, 3 cubic in float64)", p, d.interp);
        IO_REQUIRE(d.mask1_off >= 0 && d.mask2_off >= 0 && (size_t)d.mask1_off + hw <= arena_bytes &&
                       (size_t)d.mask2_off + hw <= arena_bytes,
                   IO_ERR_SHAPE, "pair_planes: pair %d masks outside the arena", p);
        IO_REQUIRE(!rgb || (d.image_off >= 0 && (size_t)d.image_off + 3 * hw <= arena_bytes), IO_ERR_SHAPE,
                   "pair_planes: pair %d image outside the arena", p);
    }
    double m[3] = {0.0, 0.0, 0.0}, s[3] = {1.0, 1.0, 1.0};
    if (rgb)
        for (int c = 0; c < 3; ++c) {
            m[c] = mean3[c];
            s[c] = std3[c];
        }
    IoProfScope prof(IO_PROF_PACK, 0.0, (double)P * SH * S * (rgb ? 20.0 : 8.0), st);
    hipLaunchKernelGGL(pair_planes_kernel, dim3(io_cdiv((long)SH * S, 256), P), dim3(256), 0, st, arena, desc_dev, SH, S,
                       m[0], m[1], m[2], s[0], s[1], s[2], rgb, modal1, modal2);
    return io_check_launch("pair_planes");
}
