// coresident.hip -- can a small workgroup START on a CU while a persistent workgroup of another stream occupies it?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/coresident.hip -o tools/ubench/coresident && tools/ubench/coresident
// Kernel A ("big"): 256 workgroups (one per CU) of WA waves, VA VGPRs per lane (forced by a live register array), LA bytes of LDS, spinning
// for ~200 us.  Kernel B ("small"): NB workgroups of 4 waves, ~140 VGPRs, 49 KB LDS, launched on ANOTHER stream 30 us after A; every B
// workgroup stamps s_memrealtime at its start.  Reported: when B's workgroups started relative to A's start and end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NREG, int WAVES, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void big(float* out, unsigned long long* stamp, int spin_us) {
    extern __shared__ float lds[];
    float r[NREG];
#pragma unroll
    for (int k = 0; k < NREG; k++) r[k] = threadIdx.x * 0.001f + k;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamp[blockIdx.x * 2] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) {
#pragma unroll
        for (int k = 0; k < NREG; k++) r[k] = __builtin_fmaf(r[k], 1.0001f, 0.5f);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < NREG; k++) s += r[k];
    lds[threadIdx.x] = s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[threadIdx.x];
    if (threadIdx.x == 0) stamp[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
}
template <int NREG>
__global__ __launch_bounds__(256, 1) void small(float* out, unsigned long long* stamp, int spin_us) {
    extern __shared__ float lds[];
    float r[NREG];
#pragma unroll
    for (int k = 0; k < NREG; k++) r[k] = threadIdx.x * 0.002f + k;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamp[blockIdx.x * 2] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) {
#pragma unroll
        for (int k = 0; k < NREG; k++) r[k] = __builtin_fmaf(r[k], 1.0001f, 0.25f);
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < NREG; k++) s += r[k];
    lds[threadIdx.x] = s;
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
    if (threadIdx.x == 0) stamp[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
}

template <int NREG, int WAVES, int MINW>
static int run(const char* name, size_t lds_big, int nb) {
    float *oa, *ob; unsigned long long *sa, *sb;
    CK(hipMalloc(&oa, 4 << 20)); CK(hipMalloc(&ob, 4 << 20)); CK(hipMalloc(&sa, 16 * 1024)); CK(hipMalloc(&sb, 16 * 1024));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)big<NREG, WAVES, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
    CK(hipFuncSetAttribute((const void*)small<120>, hipFuncAttributeMaxDynamicSharedMemorySize, 49 * 1024));
    hipFuncAttributes fa, fb;
    CK(hipFuncGetAttributes(&fa, (const void*)big<NREG, WAVES, MINW>)); CK(hipFuncGetAttributes(&fb, (const void*)small<120>));
    for (int warm = 0; warm < 2; warm++) {
        hipLaunchKernelGGL((big<NREG, WAVES, MINW>), dim3(256), dim3(WAVES * 64), lds_big, s1, oa, sa, 200);
        hipLaunchKernelGGL((small<120>), dim3(8), dim3(256), 49 * 1024, s2, ob, sb, 1);   // (first submissions set the queues up)
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL((big<NREG, WAVES, MINW>), dim3(256), dim3(WAVES * 64), lds_big, s1, oa, sa, 200);
        // ~30 us later
        hipLaunchKernelGGL((small<120>), dim3(1), dim3(256), 49 * 1024, s2, ob, sb, 25);
        hipLaunchKernelGGL((small<120>), dim3(nb), dim3(256), 49 * 1024, s2, ob, sb, 15);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> ha(512), hb(2 * nb);
    CK(hipMemcpy(ha.data(), sa, 4096, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), sb, 16 * nb, hipMemcpyDeviceToHost));
    unsigned long long a0 = ~0ull, a1 = 0;
    for (int k = 0; k < 256; k++) { a0 = std::min(a0, ha[2 * k]); a1 = std::max(a1, ha[2 * k + 1]); }
    int during = 0; double first = 1e30, last = 0;
    for (int k = 0; k < nb; k++) {
        const double st = ((double)hb[2 * k] - (double)a0) / 100.0;
        during += hb[2 * k] < a1 - 2000;                    // started at least 20 us before A ended
        first = std::min(first, st); last = std::max(last, st);
    }
    printf("%-34s big: %d VGPRs, %zu B LDS, lasted %.0f us | small: %d VGPRs; %d of %d small workgroups started while big ran "
           "(first %.0f us, last %.0f us after big's start)\n", name, fa.numRegs, lds_big, (a1 - a0) / 100.0, fb.numRegs, during, nb, first, last);
    return 0;
}
int main() {
    if (run<130, 12, 3>("12 waves x ~168 VGPRs, 72 KB", 72 * 1024, 128)) return 1;
    if (run<130, 8, 3>("8 waves x ~168 VGPRs, 72 KB", 72 * 1024, 128)) return 1;
    if (run<130, 8, 3>("8 waves x ~168 VGPRs, 108 KB", 108 * 1024, 128)) return 1;
    if (run<130, 8, 3>("8 waves x ~168 VGPRs, 72 KB, 345 small", 72 * 1024, 345)) return 1;
    if (run<60, 8, 3>("8 waves x ~100 VGPRs, 72 KB", 72 * 1024, 128)) return 1;
    return 0;
}
