// Which CUs does a hipExtStreamCreateWithCUMask stream run on, and how are a launch's workgroups dealt to the XCDs?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/cumask.hip -o tools/ubench/cumask && tools/ubench/cumask
// For masks "bits [lo, hi)" prints, per XCD, the number of distinct CUs (HW_ID: cu_id, sh_id, se_id) that ran a workgroup of a
// 2048-workgroup launch whose workgroups hold their CU for a while, and whether workgroup b still runs on XCD b & 7 (what
// conv3x3_f16x3r's row-band mapping assumes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void who(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, nwg = 2048;
    uint32_t* d; CK(hipMalloc(&d, nwg * 8));
    std::vector<uint32_t> h(2 * nwg);
    const int ranges[][2] = {{0, ncu}, {0, 128}, {128, 256}, {0, 96}, {96, 256}, {0, 80}, {80, 256}, {0, 64}, {64, 256}, {0, 112}, {112, 256}, {0, 8}, {0, 16}, {8, 16}, {0, 1}, {1, 2}};
    for (auto& rg : ranges) {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int c = rg[0]; c < rg[1] && c < ncu; c++) mask[c / 32] |= 1u << (c % 32);
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
        CK(hipMemsetAsync(d, 0xff, nwg * 8, st));
        hipLaunchKernelGGL(who, dim3(nwg), dim3(768), 0, st, d, 1000);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
        std::set<uint32_t> cus[8]; bool rr = true;
        for (int b = 0; b < nwg; b++) { const int x = h[2 * b] & 7; rr = rr && x == (b & 7); cus[x].insert((h[2 * b + 1] >> 8) & 0xffu); }
        printf("mask bits [%3d,%3d): CUs per XCD", rg[0], rg[1]);
        for (int x = 0; x < 8; x++) printf(" %2zu", cus[x].size());
        printf("   workgroup b on XCD b & 7: %s\n", rr ? "yes" : "NO");
        CK(hipStreamDestroy(st));
    }
    return 0;
}
