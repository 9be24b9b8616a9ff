// atomicbench.hip -- cost of accumulating per-channel BN sums with fp64 atomics instead of a finalize launch:
// 3680 workgroups x 64 atomicAdd(double) (32 channels x {sum, sum of squares}) into `slots` replicas, compact or one
// 128-byte line per accumulator.  Each workgroup also streams 96 KB so the atomics compete with real traffic.
// hipcc --offload-arch=gfx950 -O3 tools/atomicbench.hip -o tools/atomicbench && ./tools/atomicbench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void k(const float4* in, float4* out, double* acc, int slots, int stride, int do_atomics) {
    // do_atomics == 2: the replicas of a channel share one 128-byte line ([channel][slot][2])
    const size_t base = (size_t)blockIdx.x * 6144;              // 6144 float4 = 96 KB per workgroup
    float4 s = make_float4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < 6144; i += 512) { const float4 v = in[base + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    for (int i = threadIdx.x; i < 2048; i += 512) out[(size_t)blockIdx.x * 2048 + i] = s;
    if (do_atomics && threadIdx.x < 32) {
        const int slot = blockIdx.x % slots;
        double* a = do_atomics == 2 ? acc + ((size_t)threadIdx.x * slots + slot) * 2 : acc + ((size_t)slot * 32 + threadIdx.x) * 2 * stride;
        atomicAdd(a, (double)s.x);
        atomicAdd(a + (do_atomics == 2 ? 1 : stride), (double)s.y);
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int nb = 3680;
    float4 *in, *out; double* acc;
    CK(hipMalloc((void**)&in, (size_t)nb * 6144 * 16)); CK(hipMalloc((void**)&out, (size_t)nb * 2048 * 16));
    CK(hipMalloc((void**)&acc, (size_t)64 * 32 * 2 * 16 * 8));
    CK(hipMemsetAsync(in, 0, (size_t)nb * 6144 * 16, st)); CK(hipMemsetAsync(acc, 0, (size_t)64 * 32 * 2 * 16 * 8, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { int atom, slots, stride; const char* name; } cfg[] = {
        {0, 1, 1, "no atomics"}, {1, 1, 1, "1 slot compact"}, {1, 8, 1, "8 slots compact"}, {1, 32, 1, "32 slots compact"},
        {1, 8, 16, "8 slots, line each"}, {1, 1, 16, "1 slot, line each"}, {2, 8, 1, "8 slots in the channel's line"}, {2, 4, 1, "4 slots in the channel's line"}, {1, 4, 1, "4 slots compact"}, {1, 2, 1, "2 slots compact"}};
    for (auto& c : cfg) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, st, in, out, acc, c.slots, c.stride, c.atom);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-22s %7.1f us\n", c.name, best * 1e3);
    }
    return 0;
}
