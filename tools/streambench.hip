// streambench.hip -- what HBM actually delivers on this box for the conv's traffic mix: float4 streams with R:W = 1:0,
// 1:1, 2:1 (the split-fp16 conv on enc1.l2a reads 241 MB and writes 120 MB), grid-stride, 4 loads in flight per thread.
// hipcc --offload-arch=gfx950 -O3 tools/streambench.hip -o tools/streambench && ./tools/streambench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NR, int NW>
__global__ __launch_bounds__(256) void stream(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    // each step: NR float4 reads from NR disjoint regions, NW float4 writes
    const size_t stride = (size_t)gridDim.x * 256;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float4 v[NR > 0 ? NR : 1];
#pragma unroll
        for (int r = 0; r < NR; r++) v[r] = in[(size_t)r * n + i];
#pragma unroll
        for (int r = 0; r < NR; r++) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
#pragma unroll
        for (int w = 0; w < NW; w++) out[(size_t)w * n + i] = acc;
    }
    if (NW == 0 && acc.x == 12345.678f) out[0] = acc;
}

template <int NR, int NW>
int run(const float4* in, float4* out, size_t n, hipStream_t st, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {2048, 8192}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL((stream<NR, NW>), dim3(grid), dim3(256), 0, st, in, out, n);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = (double)(NR + NW) * n * 16;
        printf("%-12s grid %5d: %7.1f us  %.2f TB/s (%.0f MB)\n", name, grid, best * 1e3, bytes / best / 1e9, bytes / 1e6);
    }
    return 0;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t n = (size_t)120 << 16;                       // 7.5 M float4 = 126 MB per region
    float4 *in, *out;
    CK(hipMalloc((void**)&in, n * 16 * 4)); CK(hipMalloc((void**)&out, n * 16 * 2));
    CK(hipMemsetAsync(in, 0, n * 16 * 4, st)); CK(hipMemsetAsync(out, 0, n * 16 * 2, st));
    if (run<2, 0>(in, out, n, st, "read 2:0")) return 1;
    if (run<4, 0>(in, out, n, st, "read 4:0")) return 1;
    if (run<1, 1>(in, out, n, st, "copy 1:1")) return 1;
    if (run<2, 1>(in, out, n, st, "mix 2:1")) return 1;
    if (run<0, 1>(in, out, n, st, "write 0:1")) return 1;
    return 0;
}
