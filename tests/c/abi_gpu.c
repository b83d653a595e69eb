/* libinstaorder_hip.so driven from C with nothing but the HIP runtime (no torch, no Python): a 3x3 convolution forward,
 * its data gradient and its filter gradient on device buffers from hipMalloc, checked against loops on the host.
 * Built with gcc (-lamdhip64) and run by tests/test_gpu_ops.py::test_c_abi_from_plain_c_program (needs an MI355X). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "instaorder_hip.h"

#define N 2
#define H 9
#define W 7
#define CI 64
#define CO 64
#define R 3
#define PAD 1

static float frand(unsigned* s) {
    *s = *s * 1664525u + 1013904223u;
    return ((float)((*s >> 8) & 0xffff) / 65536.0f) - 0.5f;
}

#define CK(x)                                                                 \
    do {                                                                      \
        if ((x) != hipSuccess) { fprintf(stderr, "HIP error line %d\n", __LINE__); return 10; } \
    } while (0)
#define IO(x)                                                                                   \
    do {                                                                                        \
        if ((x) != 0) { fprintf(stderr, "line %d: %s\n", __LINE__, io_last_error_string()); return 11; } \
    } while (0)

int main(void) {
    const size_t nx = (size_t)N * H * W * CI, nw = (size_t)CO * R * R * CI, ny = (size_t)N * H * W * CO;
    float *x = malloc(nx * 4), *w = malloc(nw * 4), *dy = malloc(ny * 4), *y = malloc(ny * 4), *dx = malloc(nx * 4),
          *dw = malloc(nw * 4), *ry = calloc(ny, 4), *rdx = calloc(nx, 4), *rdw = calloc(nw, 4);
    float *d_x, *d_w, *d_wt, *d_y, *d_dy, *d_dx, *d_dw;
    void* d_ws;
    size_t ws_bytes, i;
    unsigned seed = 7;
    double e_y = 0, n_y = 0, e_dx = 0, n_dx = 0, e_dw = 0, n_dw = 0;
    int n, h, ww, o, c, r, s;

    if (io_device_count() < 1) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    for (i = 0; i < nx; ++i) x[i] = frand(&seed);
    for (i = 0; i < nw; ++i) w[i] = frand(&seed) * 0.2f;
    for (i = 0; i < ny; ++i) dy[i] = frand(&seed);
    /* host reference: y[n,h,w,o] = sum x[n,h+r-1,w+s-1,c] w[o,r,s,c]; dx, dw its adjoints */
    for (n = 0; n < N; ++n) for (h = 0; h < H; ++h) for (ww = 0; ww < W; ++ww) for (o = 0; o < CO; ++o)
        for (r = 0; r < R; ++r) for (s = 0; s < R; ++s) {
            const int hi = h + r - PAD, wi = ww + s - PAD;
            if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
            for (c = 0; c < CI; ++c) {
                const size_t ix = (((size_t)n * H + hi) * W + wi) * CI + c, iw = (((size_t)o * R + r) * R + s) * CI + c,
                             iy = (((size_t)n * H + h) * W + ww) * CO + o;
                ry[iy] += x[ix] * w[iw];
                rdx[ix] += dy[iy] * w[iw];
                rdw[iw] += dy[iy] * x[ix];
            }
        }
    CK(hipMalloc((void**)&d_x, nx * 4)); CK(hipMalloc((void**)&d_w, nw * 4)); CK(hipMalloc((void**)&d_wt, nw * 4));
    CK(hipMalloc((void**)&d_y, ny * 4)); CK(hipMalloc((void**)&d_dy, ny * 4)); CK(hipMalloc((void**)&d_dx, nx * 4));
    CK(hipMalloc((void**)&d_dw, nw * 4));
    ws_bytes = io_conv2d_wgrad_workspace_bytes(N, H, W, CI, CO, R, R, 1, PAD);
    CK(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 16));
    CK(hipMemcpy(d_x, x, nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, w, nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dy, dy, ny * 4, hipMemcpyHostToDevice));
    IO(io_conv2d_fwd(d_x, d_w, d_y, N, H, W, CI, CO, R, R, 1, PAD, NULL));
    IO(io_filter_transpose(d_w, CO, R * R, CI, d_wt, NULL));
    IO(io_conv2d_dgrad(d_dy, d_wt, d_dx, NULL, NULL, N, H, W, CI, CO, R, R, 1, PAD, NULL));
    IO(io_conv2d_wgrad(d_x, d_dy, d_dw, N, H, W, CI, CO, R, R, 1, PAD, d_ws, ws_bytes, NULL));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y, d_y, ny * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dx, d_dx, nx * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dw, d_dw, nw * 4, hipMemcpyDeviceToHost));
    for (i = 0; i < ny; ++i) { e_y += (y[i] - ry[i]) * (double)(y[i] - ry[i]); n_y += ry[i] * (double)ry[i]; }
    for (i = 0; i < nx; ++i) { e_dx += (dx[i] - rdx[i]) * (double)(dx[i] - rdx[i]); n_dx += rdx[i] * (double)rdx[i]; }
    for (i = 0; i < nw; ++i) { e_dw += (dw[i] - rdw[i]) * (double)(dw[i] - rdw[i]); n_dw += rdw[i] * (double)rdw[i]; }
    printf("rel err: y %.2e dx %.2e dw %.2e\n", sqrt(e_y / n_y), sqrt(e_dx / n_dx), sqrt(e_dw / n_dw));
    return (sqrt(e_y / n_y) < 1e-5 && sqrt(e_dx / n_dx) < 1e-5 && sqrt(e_dw / n_dw) < 1e-5) ? 0 : 1;
}
