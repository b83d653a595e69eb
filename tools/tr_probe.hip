// Hardware probe for the two gfx950 features the LDS-DMA / transpose-read filter-gradient kernel relies on:
//   1. buffer_load_dwordx4 ... lds (16 B per lane straight into LDS): lane i lands at M0 base + 16 i; a lane whose
//      offset is out of the descriptor's range writes ZEROS (not "nothing");  soffset takes part in the address.
//   2. ds_read_b64_tr_b16: in each 16-lane group, lane i = 4 j + q supplies the address of 4 consecutive bf16 of row j
//      (columns 4q..4q+3 of the group's 16); the result in lane c is column c of rows 0..3.
// and times the fragment-read pattern of the kernel's two LDS images (rotated 4-row chunks for 128-wide operands,
// half-swapped 8-row chunks for 64-wide ones) against an unswizzled image (bank conflicts).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/tr_probe tools/tr_probe.hip ; run on the GPU box: tools/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x27000);
}
__device__ __forceinline__ s16x4 tr_read(const char* lds, unsigned byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)((__attribute__((address_space(3))) char*)lds + byte_off));
}

// ---- 1. DMA semantics: one wave, src[i] = i (u16); lane l fetches 16 B at element offset perm(l)*8, lane 5 invalid ----
__global__ void dma_probe(const unsigned short* src, unsigned src_bytes, unsigned soff, unsigned short* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) ((unsigned short*)lds)[i] = 0xdead;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = rsrc(src, src_bytes);
    unsigned voff = (unsigned)((lane * 7) & 63) * 16u;       // a permutation of the 64 16-byte pieces
    if (lane == 5) voff = 0xFFFFFFFFu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + 1024), 16, voff, soff, 0, 0);
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = ((unsigned short*)lds)[i];
}

// ---- 2. transpose-read semantics: lds[i] = i; lane address = row (l&15)>>2 of a [4][pitch] block + 4*(l&3) + group*16 ----
__global__ void tr_probe(int pitch_elems, s16x4* out) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, j = i >> 2, q = i & 3;
    const unsigned off = (unsigned)(j * pitch_elems + g * 16 + 4 * q) * 2u;
    out[l] = tr_read((const char*)lds, off);
}

// ---- 3. timing of the kernel's fragment reads: MODE 0 unswizzled 128-wide, 1 rotated 128-wide, 2 unswizzled 64-wide,
//         3 half-swapped 64-wide.  4 waves like the kernel (wm = wave >> 1), 8 reads per k-step, 4 k-steps, REP times.
template <int MODE>
__global__ __launch_bounds__(256) void tr_time(int rep, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1;
    const int g = l >> 4, ii = l & 15, j = ii >> 2, q = ii & 3;
    unsigned base[2];
    for (int t = 0; t < 2; ++t) {
        if (MODE < 2) {                       // 128-wide: chunk = 4 rows of 256 B
            const int c = wm * 64 + t * 32 + (g & 1) * 16 + 4 * q, cs = c >> 3;
            const int slot = MODE == 1 ? ((cs + 4 * j) & 15) : cs;
            base[t] = (unsigned)((g >> 1) * 2 * 1024 + j * 256 + slot * 16 + (c & 7) * 2);
        } else {                              // 64-wide: chunk = 8 rows of 128 B; in-chunk row = 4 h + j
            const int c = (wm * 32 + (g & 1) * 16 + 4 * q + t * 0) & 63, hs = c >> 5;
            const int P = MODE == 3 ? 2 * j + (hs ^ ((j >> 1) & 1)) : 2 * j + hs;
            base[t] = (unsigned)((g >> 1) * 1024 + P * 64 + ((c >> 3) & 3) * 16 + (c & 7) * 2);
        }
    }
    s16x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned o = MODE < 2 ? kk * 4096 + h * 1024 : kk * 2048 + h * 512;
                    s16x4 a = tr_read(lds, base[t] + o);               // the dY image
                    s16x4 b = tr_read(lds, base[t] + o + 16384u);      // the X image
                    asm volatile("" : "+v"(a), "+v"(b));
                    acc += a + b;
                }
    }
    if (acc[0] == 12345 && acc[1] == 54321) sink[0] = 1.f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    // 1. DMA
    std::vector<unsigned short> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)i;
    unsigned short *dsrc, *dout;
    CK(hipMalloc(&dsrc, 8192)); CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(dsrc, h.data(), 8192, hipMemcpyHostToDevice));
    int bad = 0;
    for (unsigned soff : {0u, 2048u}) {
        hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 4096, 0, dsrc, 4096u /* range: first 2048 elements */, soff, dout);
        std::vector<unsigned short> o(2048);
        CK(hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost));
        int nb = 0;
        for (int i = 0; i < 512; ++i) nb += o[i] != 0xdead;
        for (int i = 1024; i < 2048; ++i) nb += o[i] != 0xdead;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                const unsigned want = l == 5 ? 0 : (unsigned)(((l * 7) & 63) * 8 + e + soff / 2);
                if (o[512 + l * 8 + e] != want) {
                    if (nb < 8) printf("  dma soff=%u lane %d e %d: got %u want %u\n", soff, l, e, o[512 + l * 8 + e], want);
                    ++nb;
                }
            }
        printf("DMA probe soffset=%u: %s (%d mismatches)\n", soff, nb ? "FAIL" : "ok: lane i -> M0 + 16 i, invalid lane -> zeros, soffset added, not range-checked", nb);
        bad += nb;
    }
    // 2. tr semantics
    s16x4* dtr;
    CK(hipMalloc(&dtr, 64 * 8));
    for (int pitch : {16, 64, 128, 160}) {
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, pitch, dtr);
        short o[64][4];
        CK(hipMemcpy(o, dtr, sizeof(o), hipMemcpyDeviceToHost));
        int nb = 0;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const int want = e * pitch + (l >> 4) * 16 + (l & 15);     // row e, column (l & 15) of group l >> 4
                if (o[l][e] != want) {
                    if (nb < 8) printf("  tr pitch %d lane %d e %d: got %d want %d\n", pitch, l, e, o[l][e], want);
                    ++nb;
                }
            }
        printf("tr probe pitch %d: %s (%d mismatches)\n", pitch, nb ? "FAIL" : "ok: lane c gets column c of rows 0..3", nb);
        bad += nb;
    }
    // 3. timing
    float* sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rep = 2000, blocks = 512;
    auto run = [&](int mode) {
        float best = 1e9f;
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(tr_time<0>, dim3(blocks), dim3(256), 32768, 0, rep, sink);
            if (mode == 1) hipLaunchKernelGGL(tr_time<1>, dim3(blocks), dim3(256), 32768, 0, rep, sink);
            if (mode == 2) hipLaunchKernelGGL(tr_time<2>, dim3(blocks), dim3(256), 32768, 0, rep, sink);
            if (mode == 3) hipLaunchKernelGGL(tr_time<3>, dim3(blocks), dim3(256), 32768, 0, rep, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double reads = (double)blocks * 4 * rep * 32;        // wave-instructions
        printf("tr timing mode %d: %.3f ms, %.2f ns per wave-instruction per CU-slot (%.1f TB/s aggregate)\n", mode, best,
               best * 1e6 / (reads / 256.0), reads * 512.0 / (best * 1e-3) / 1e12);
    };
    for (int m = 0; m < 4; ++m) run(m);
    printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad != 0;
}
