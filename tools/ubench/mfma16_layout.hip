// mfma16_layout.hip -- operand / result layout and issue rate of v_mfma_f32_16x16x32_f16 on gfx950 (used by conv3x3_f16x3q)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma16_layout.hip -o tools/ubench/mfma16_layout && tools/ubench/mfma16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// D = A (16 x 32) * B (32 x 16) with the ASSUMED layouts: lane l holds A[l % 16][8 (l / 16) + e], B[8 (l / 16) + e][l % 16], e = 0..7,
// and receives D[4 (l / 16) + r][l % 16], r = 0..3
__global__ void layout(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)A[(l % 16) * 32 + 8 * (l / 16) + e]; b[e] = (_Float16)B[(8 * (l / 16) + e) * 16 + l % 16]; }
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * (l / 16) + r) * 16 + l % 16] = d[r];
}
template <int SHAPE>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int loops) {
    f16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.02f * (threadIdx.x - e)); }
    f32x4 d4[4] = {};
    f32x16 d16[2] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < loops; it++) {
        if (SHAPE == 16) {
#pragma unroll
            for (int i = 0; i < 32; i++) d4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d4[i & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) d16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d16[i & 1], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 4; r++) s += d4[i][r];
    for (int i = 0; i < 2; i++) for (int r = 0; r < 16; r++) s += d16[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    std::vector<float> A(16 * 32), B(32 * 16), D(256), R(256, 0.f);
    srand(1);
    for (auto& v : A) v = (float)(rand() % 17 - 8) / 8.0f;
    for (auto& v : B) v = (float)(rand() % 17 - 8) / 8.0f;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 32; k++) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
    float *dA, *dB, *dD; unsigned long long* dc;
    CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dD, 1 << 20)); CK(hipMalloc(&dc, 8 * 256));
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += D[i] != R[i];
    printf("16x16x32 f16 layout (A[l%%16][8(l/16)+e], B[8(l/16)+e][l%%16], D[4(l/16)+r][l%%16]): %d of 256 mismatches\n", bad);
    for (int G : {1, 256}) {
        unsigned long long h;
        hipLaunchKernelGGL(rate<16>, dim3(G), dim3(256), 0, 0, dD, dc, 4000); CK(hipDeviceSynchronize());
        CK(hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost));
        printf("G=%3d 16x16x32: %.2f cycles per MFMA per SIMD (one wave per SIMD, 4 chains)\n", G, (double)h / (4000.0 * 32));
        hipLaunchKernelGGL(rate<32>, dim3(G), dim3(256), 0, 0, dD, dc, 4000); CK(hipDeviceSynchronize());
        CK(hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost));
        printf("G=%3d 32x32x16: %.2f cycles per MFMA per SIMD (one wave per SIMD, 2 chains)\n", G, (double)h / (4000.0 * 16));
    }
    return 0;
}
