// valu_mfma.hip -- what do VALU instructions cost on gfx950, and do they overlap the MFMAs of other waves of the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_mfma.hip -o /tmp/valu_mfma && /tmp/valu_mfma
// One workgroup per CU, W waves per SIMD (blockDim = 256 W).  Each wave runs LOOPS iterations of a body chosen by `mode`:
//   0: 27 dependent-free v_mfma_f32_32x32x16_f16 (three accumulators)          1: NV VALU ops of kind `kind`
//   2: both in the same wave (27 MFMAs then NV VALU ops, independent)         3: even waves MFMA, odd waves VALU (same SIMD when W = 2)
// Prints cycles per body (s_memtime of wave 0) so that "sum" and "max" behaviour can be told apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND, int NV>
__device__ __forceinline__ void valu_body(float (&v)[8], unsigned (&u)[4], float s) {
#pragma unroll
    for (int i = 0; i < NV / 8; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) v[k] = __builtin_fmaf(v[k], s, 1.0f);                                   // v_fma_f32
            if (KIND == 1) { u[k & 3] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u[k & 3], 0x138, 0xF, 0xF, true); }   // wave_shr:1
            if (KIND == 2) { u[k & 3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[k], v[(k + 1) & 7])); v[k] += (float)u[k & 3]; }   // cvt_pkrtz + cvt + add
            if (KIND == 3) v[k] = v[k] > s ? v[k] : v[(k + 1) & 7];                                // v_cmp + v_cndmask
            if (KIND == 4) { u[k & 3] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u[k & 3], 0xB1, 0xF, 0xF, true); }    // quad_perm
            if (KIND == 5) v[k] = __builtin_fmaxf(v[k], v[k] * s);                                 // mul + max
        }
    }
}

template <int KIND, int NV>
__global__ __launch_bounds__(1024) void bench(float* out, unsigned long long* cyc, int mode, int loops, float s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[r][k] = 0.f;
    float v[8]; unsigned u[4];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = 0.5f + 0.01f * (lane + k);
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = lane * 7 + k;
    const bool do_mfma = mode == 0 || mode == 2 || (mode == 3 && !(wave & 4));   // waves w and w + 4 share a SIMD (round-robin placement)
    const bool do_valu = mode == 1 || mode == 2 || (mode == 3 && (wave & 4));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < loops; it++) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 9; i++)
#pragma unroll
                for (int r = 0; r < 3; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r], 0, 0, 0);
        }
        if (do_valu) valu_body<KIND, NV>(v, u, s);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) r += acc[0][k] + acc[1][k] + acc[2][k];
#pragma unroll
    for (int k = 0; k < 8; k++) r += v[k];
#pragma unroll
    for (int k = 0; k < 4; k++) r += (float)u[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int KIND, int NV>
static int run(const char* name, float* d_out, unsigned long long* d_cyc) {
    for (int W : {1, 2, 3}) {
        for (int mode = 0; mode < 4; mode++) {
            if (mode == 3 && W != 2) continue;
            const int loops = 2000;
            hipLaunchKernelGGL((bench<KIND, NV>), dim3(256), dim3(256 * W), 0, 0, d_out, d_cyc, mode, loops, 0.999f);
            CK(hipDeviceSynchronize());
            unsigned long long h[16];
            CK(hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost));
            // s_memtime counts at 100 MHz on gfx950: convert with the measured kernel time instead
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((bench<KIND, NV>), dim3(256), dim3(256 * W), 0, 0, d_out, d_cyc, mode, loops, 0.999f);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-22s NV=%3d W=%d mode=%d: %8.1f ns per body (%7.0f cycles at 2.4 GHz)\n", name, NV, W, mode, ms * 1e6 / loops, ms * 1e6 / loops * 2.4);
        }
    }
    return 0;
}

int main() {
    float* d_out; unsigned long long* d_cyc;
    CK(hipMalloc((void**)&d_out, 256 * 1024 * 4)); CK(hipMalloc((void**)&d_cyc, 256 * 16 * 8));
    printf("modes: 0 = 27 MFMA 32x32x16 f16, 1 = NV VALU ops, 2 = both in every wave, 3 = W=2: one wave of a SIMD MFMA, the other VALU\n");
    if (run<0, 104>("v_fma_f32", d_out, d_cyc)) return 1;
    if (run<1, 104>("dpp wave_shr:1", d_out, d_cyc)) return 1;
    if (run<4, 104>("dpp quad_perm", d_out, d_cyc)) return 1;
    if (run<2, 104>("cvt_pkrtz+cvt+add", d_out, d_cyc)) return 1;
    if (run<3, 104>("cmp+cndmask", d_out, d_cyc)) return 1;
    if (run<5, 104>("mul+max", d_out, d_cyc)) return 1;
    return 0;
}
