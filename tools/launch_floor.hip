// launch_floor.hip -- how long does a dependent chain of tiny kernels take per kernel: plain stream launches vs one
// hipGraph launch.  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
__global__ void big(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* d; const size_t N = (size_t)64 << 20; CK(hipMalloc((void**)&d, N * 4)); CK(hipMemsetAsync(d, 0, N * 4, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 200;
    for (int blocks : {1, 32, 256}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < K; k++) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, d, blocks * 256);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("stream  tiny x%d (%3d blocks): %.2f us per kernel\n", K, blocks, ms * 1e3 / K);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; k++) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, d, blocks * 256);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("graph   tiny x%d (%3d blocks): %.2f us per kernel\n", K, blocks, ms * 1e3 / K);
        }
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    // a big kernel followed by a tiny one: does the boundary cost depend on how much the big one dirtied?
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < 20; k++) { hipLaunchKernelGGL(big, dim3(2048), dim3(256), 0, st, d, N); hipLaunchKernelGGL(tiny, dim3(32), dim3(256), 0, st, d, 8192); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        float ms2;
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < 20; k++) hipLaunchKernelGGL(big, dim3(2048), dim3(256), 0, st, d, N);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms2, e0, e1));
        if (rep) printf("big(256 MB rw) + tiny: %.1f us per pair; big alone %.1f us -> tiny costs %.1f us\n", ms * 1e3 / 20, ms2 * 1e3 / 20, (ms - ms2) * 1e3 / 20);
    }
    return 0;
}
