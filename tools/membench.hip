// membench.hip -- how fast can a CU pull halo tiles of a planar fp32 tensor?  (design probe, not product code)
// Each workgroup walks tiles of TH x TW output pixels; per step it reads a (TH+2) x (TW+2) halo of KH channels with
// dword loads, lanes = consecutive halo pixels (the conv kernels' staging pattern), NSET register sets in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int TW, int TH, int NWAVE, int NSET, int VEC>
__global__ void probe(const float* __restrict__ in, float* out, int C, int H, int W, int tiles_x, int ntiles, int nchunks) {
    constexpr int RS = TW + 2, PL = (TH + 2) * RS, KH = 16;
    constexpr int NIT = (KH * PL + NWAVE * 64 * VEC - 1) / (NWAVE * 64 * VEC);   // loads per lane per step
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W;
    float acc = 0.f;
    float regs[NSET][NIT][VEC];
    int e_c[NIT], e_y[NIT], e_x[NIT];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int e = (tid + k * NWAVE * 64) * VEC;
        const int c = e / PL, pix = e - c * PL;
        e_c[k] = c < KH ? c : KH - 1; e_y[k] = pix / RS; e_x[k] = pix - e_y[k] * RS;
    }
    int step_tile = blockIdx.x, step_chunk = 0;
    const int my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my * nchunks;
    auto fetch = [&](int set) {
        const int ty0 = (step_tile / tiles_x) * TH, tx0 = (step_tile % tiles_x) * TW;
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            const int c = e_c[k];
            int y = ty0 + e_y[k] - 1, x = tx0 + e_x[k] - 1;
            y = y < 0 ? 0 : (y >= H ? H - 1 : y); x = x < 0 ? 0 : (x > W - VEC ? W - VEC : x);
            int cg = step_chunk * KH + c; cg = cg < C ? cg : C - 1;
            const float* p = in + (size_t)cg * plane + (size_t)y * W + x;
#pragma unroll
            for (int v = 0; v < VEC; v++) regs[set][k][v] = p[v];
        }
        if (++step_chunk == nchunks) { step_chunk = 0; step_tile += gridDim.x; }
    };
    auto use = [&](int set) {
#pragma unroll
        for (int k = 0; k < NIT; k++)
#pragma unroll
            for (int v = 0; v < VEC; v++) acc += regs[set][k][v];
    };
    for (int s = 0; s < NSET && s < S; s++) fetch(s);
    for (int s = 0; s < S; s += NSET) {
#pragma unroll
        for (int j = 0; j < NSET; j++)
            if (s + j < S) { use(j); if (s + j + NSET < S) fetch(j); }
    }
    if (acc == 1234.5f) out[tid] = acc;
}

template <int TW, int TH, int NWAVE, int NSET, int VEC>
int run(const char* name, const float* d_in, float* d_out, int C, int H, int W, int grid) {
    const int tiles_x = W / TW, ntiles = tiles_x * (H / TH), nchunks = C / 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<TW, TH, NWAVE, NSET, VEC>), dim3(grid), dim3(NWAVE * 64), 0, 0, d_in, d_out, C, H, W, tiles_x, ntiles, nchunks);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it == 2) {
            const double useful = (double)C * H * W * 4, halo = useful * (TH + 2) * (TW + 2) / (double)(TH * TW);
            printf("%-34s grid %4d  %7.1f us  useful %.2f TB/s  with-halo %.2f TB/s\n", name, grid, ms * 1e3, useful / ms / 1e9, halo / ms / 1e9);
        }
    }
    return 0;
}

// channel-quad interleaved layout [C/4][H][W][4]: one 16-byte load per (quad, halo pixel), lanes = consecutive pixels
template <int TW, int TH, int NWAVE, int NSET>
__global__ void probe_c4(const float4* __restrict__ in, float* out, int C, int H, int W, int tiles_x, int ntiles, int nchunks) {
    constexpr int RS = TW + 2, PL = (TH + 2) * RS, KH = 16;
    constexpr int NIT = (KH / 4 * PL + NWAVE * 64 - 1) / (NWAVE * 64);
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W;
    float acc = 0.f;
    float4 regs[NSET][NIT];
    int e_q[NIT], e_y[NIT], e_x[NIT];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int e = tid + k * NWAVE * 64;
        const int q = e / PL, pix = e - q * PL;
        e_q[k] = q < KH / 4 ? q : KH / 4 - 1; e_y[k] = pix / RS; e_x[k] = pix - e_y[k] * RS;
    }
    int step_tile = blockIdx.x, step_chunk = 0;
    const int my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int S = my * nchunks;
    auto fetch = [&](int set) {
        const int ty0 = (step_tile / tiles_x) * TH, tx0 = (step_tile % tiles_x) * TW;
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            int y = ty0 + e_y[k] - 1, x = tx0 + e_x[k] - 1;
            y = y < 0 ? 0 : (y >= H ? H - 1 : y); x = x < 0 ? 0 : (x >= W ? W - 1 : x);
            int qg = step_chunk * (KH / 4) + e_q[k]; qg = qg < C / 4 ? qg : C / 4 - 1;
            regs[set][k] = in[(size_t)qg * plane + (size_t)y * W + x];
        }
        if (++step_chunk == nchunks) { step_chunk = 0; step_tile += gridDim.x; }
    };
    auto use = [&](int set) {
#pragma unroll
        for (int k = 0; k < NIT; k++) acc += regs[set][k].x + regs[set][k].y + regs[set][k].z + regs[set][k].w;
    };
    for (int s = 0; s < NSET && s < S; s++) fetch(s);
    for (int s = 0; s < S; s += NSET) {
#pragma unroll
        for (int j = 0; j < NSET; j++)
            if (s + j < S) { use(j); if (s + j + NSET < S) fetch(j); }
    }
    if (acc == 1234.5f) out[tid] = acc;
}
template <int TW, int TH, int NWAVE, int NSET>
int run_c4(const char* name, const float* d_in, float* d_out, int C, int H, int W, int grid) {
    const int tiles_x = W / TW, ntiles = tiles_x * (H / TH), nchunks = C / 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe_c4<TW, TH, NWAVE, NSET>), dim3(grid), dim3(NWAVE * 64), 0, 0, (const float4*)d_in, d_out, C, H, W, tiles_x, ntiles, nchunks);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it == 2) {
            const double useful = (double)C * H * W * 4, halo = useful * (TH + 2) * (TW + 2) / (double)(TH * TW);
            printf("%-34s grid %4d  %7.1f us  useful %.2f TB/s  with-halo %.2f TB/s\n", name, grid, ms * 1e3, useful / ms / 1e9, halo / ms / 1e9);
        }
    }
    return 0;
}

int main() {
    const int C = 64, H = 736, W = 1280;
    float *d_in, *d_out;
    CK(hipMalloc(&d_in, (size_t)C * H * W * 4)); CK(hipMalloc(&d_out, 1 << 20));
    CK(hipMemset(d_in, 0, (size_t)C * H * W * 4));
    run<32, 8, 4, 1, 1>("8x32 tile 4 waves 1 set", d_in, d_out, C, H, W, 256);
    run<32, 8, 4, 2, 1>("8x32 tile 4 waves 2 sets", d_in, d_out, C, H, W, 256);
    run<32, 8, 4, 2, 1>("8x32 tile 4 waves 2 sets x2 blocks", d_in, d_out, C, H, W, 512);
    run<32, 8, 4, 2, 1>("8x32 tile 4 waves 2 sets x4 blocks", d_in, d_out, C, H, W, 1024);
    run<32, 8, 8, 2, 1>("8x32 tile 8 waves 2 sets", d_in, d_out, C, H, W, 256);
    run<32, 8, 8, 2, 1>("8x32 tile 8 waves 2 sets x2 blocks", d_in, d_out, C, H, W, 512);
    run<64, 4, 4, 2, 1>("4x64 tile 4 waves 2 sets", d_in, d_out, C, H, W, 256);
    run<128, 2, 4, 2, 1>("2x128 tile 4 waves 2 sets", d_in, d_out, C, H, W, 256);
    run<64, 8, 4, 2, 1>("8x64 tile 4 waves 2 sets", d_in, d_out, C, H, W, 256);
    run<64, 8, 8, 2, 1>("8x64 tile 8 waves 2 sets", d_in, d_out, C, H, W, 256);
    run<64, 8, 8, 2, 1>("8x64 tile 8 waves 2 sets x2 blocks", d_in, d_out, C, H, W, 512);
    run<32, 8, 4, 2, 2>("8x32 tile 4 waves 2 sets dwordx2", d_in, d_out, C, H, W, 256);
    run<64, 8, 4, 2, 2>("8x64 tile 4 waves 2 sets dwordx2", d_in, d_out, C, H, W, 256);
    run_c4<32, 8, 4, 1>("C4 8x32 tile 4 waves 1 set", d_in, d_out, C, H, W, 256);
    run_c4<32, 8, 4, 2>("C4 8x32 tile 4 waves 2 sets", d_in, d_out, C, H, W, 256);
    run_c4<32, 8, 4, 2>("C4 8x32 tile 4 waves 2 sets x2", d_in, d_out, C, H, W, 512);
    run_c4<32, 8, 8, 2>("C4 8x32 tile 8 waves 2 sets", d_in, d_out, C, H, W, 256);
    run_c4<32, 8, 8, 2>("C4 8x32 tile 8 waves 2 sets x2", d_in, d_out, C, H, W, 512);
    run_c4<32, 8, 4, 4>("C4 8x32 tile 4 waves 4 sets x2", d_in, d_out, C, H, W, 512);
    run_c4<64, 8, 8, 2>("C4 8x64 tile 8 waves 2 sets", d_in, d_out, C, H, W, 256);
    return 0;
}
