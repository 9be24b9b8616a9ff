// pk_f32_mfma_erratum.hip -- gfx950 (MI355X, ROCm 7.2): wrong VALU results in lanes 48..63 beside another wave's fp16 MFMAs.
//
// A loop of packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 with scalar, constant and op_sel operands: what hipcc's SLP
// vectoriser makes of any two adjacent fp32 operations) returns wrong values in the LAST QUARTER of the wave (lanes 48..63) when
//   (1) another wave on the same SIMD issues v_mfma_f32_32x32x16_f16 with gaps between its MFMAs (one dependent accumulator
//       chain per wave; four independent chains, which keep the matrix pipe full, never disturb anything), AND
//   (2) the loop's code sits at a particular 4-byte phase of the 32-byte instruction-fetch window (here: PAD = 7 and 15 of 0..15
//       s_nop paddings in front of the byte-identical loop; the other fourteen phases are always right).
// The same arithmetic written with one fp32 operation per instruction (PK = 0) is right at every phase.  Nothing is shared between
// the two kernels but the CU.  Found as "a path-trace bounce kernel beside the split-fp16 conv kernel computes wrong values for
// runs of lanes ending at a 16-lane boundary" (DESIGN.md); fixed there by building the bounce kernels without packed fp32
// (-fno-slp-vectorize).      hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pk_f32_mfma_erratum.hip -o pk_erratum && ./pk_erratum [runs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash32(unsigned a) {
    a = (a + 0x7ed55d16u) + (a << 12); a = (a ^ 0xc761c23cu) ^ (a >> 19); a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9); a = (a + 0xfd7046c5u) + (a << 3); a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
__device__ __forceinline__ float unit(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f) + 0.25f; }
template <int PK, int PAD>
__global__ __launch_bounds__(256) void victim(int tmask, float* out, int n, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    asm volatile(".rept %0\n s_nop 0\n .endr" :: "i"(PAD));             // shifts the loop below by 4 * PAD bytes
    unsigned h = hash32((unsigned)t * 2654435761u + 12345u);
    float x = unit(h), y = unit(hash32(h)), z = unit(hash32(h + 1u)), acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        const float c0 = (float)tmask * (1.0f / 65536.0f), c1 = (float)n * (1.0f / 8192.0f), c2 = (float)iters * (1.0f / 512.0f);   // wave-uniform
        if (PK) {
            f32x2 a = {x, y}, b = {z, x};
            a = a * (f32x2){c0, c1} + (f32x2){-0.5f, -0.5f};
            b = b * (f32x2){a[1], a[1]} - a * (f32x2){c2, c2};
            a = (f32x2){b[1], b[0]} * a;
            acc += a[0] - a[1];
            x = 0.25f + 0.5f * (a[0] - (float)(int)a[0]); y = 0.25f + 0.5f * (b[1] - (float)(int)b[1]); z = 0.25f + 0.5f * (b[0] - (float)(int)b[0]);
        } else {                                                          // the same operations, one per instruction
            float a0 = x * c0 - 0.5f, a1 = y * c1 - 0.5f;
            asm volatile("" : "+v"(a0), "+v"(a1));
            float b0 = z * a1 - a0 * c2, b1 = x * a1 - a1 * c2;
            asm volatile("" : "+v"(b0), "+v"(b1));
            a0 = b1 * a0; asm volatile("" : "+v"(a0)); a1 = b0 * a1;
            acc += a0 - a1;
            x = 0.25f + 0.5f * (a0 - (float)(int)a0); y = 0.25f + 0.5f * (b1 - (float)(int)b1); z = 0.25f + 0.5f * (b0 - (float)(int)b0);
        }
    }
    out[t] = acc + x;
}
// aggressors: fp16 / bf16 / fp32-input MFMAs, ONE dependent accumulator chain per wave (DEP) or four independent ones
template <int TYPE, bool DEP>
__global__ __launch_bounds__(512, 4) void mfma(float* sink, int iters) {
    f16x8 a, b;
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.5f - 0.01f * k); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 d0 = {};
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    for (int it = 0; it < iters; it++) {
        if (TYPE == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        if (TYPE == 1) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
        if (TYPE == 2) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], d0, 0, 0, 0);
        if (!DEP) { c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c3, 0, 0, 0); }
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] + d0[0] == 12345.678f) sink[0] = 1.0f;
}
typedef void (*vlaunch)(hipStream_t, float*, int);
template <int PK, int PAD> static void go(hipStream_t st, float* out, int n) { hipLaunchKernelGGL((victim<PK, PAD>), dim3(n / 256), dim3(256), 0, st, 65535, out, n, 400); }
#define ROW(PK) {go<PK, 0>, go<PK, 1>, go<PK, 2>, go<PK, 3>, go<PK, 4>, go<PK, 5>, go<PK, 6>, go<PK, 7>, go<PK, 8>, go<PK, 9>, go<PK, 10>, go<PK, 11>, go<PK, 12>, go<PK, 13>, go<PK, 14>, go<PK, 15>}
int main(int argc, char** argv) {
    const int runs = argc > 1 ? atoi(argv[1]) : 200, n = 6144;
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    float *d_out, *d_sink; CK(hipMalloc((void**)&d_out, 4 * n)); CK(hipMalloc((void**)&d_sink, 64));
    std::vector<float> ref(n), got(n);
    static const vlaunch V[2][16] = {ROW(0), ROW(1)};
    const char* agg[5] = {"nothing", "4 independent f16 MFMA chains", "1 dependent f16 MFMA chain", "1 dependent bf16 MFMA chain", "1 dependent f32 MFMA chain"};
    for (int pk = 1; pk >= 0; pk--)
        for (int mode = 0; mode < 5; mode++) {
            printf("%-22s beside %-30s: bad runs of %d at code phase 0..15:", pk ? "packed fp32 (v_pk_*)" : "one op per instruction", agg[mode], runs);
            unsigned long long lanes = 0;
            for (int pad = 0; pad < 16; pad++) {
                CK(hipDeviceSynchronize());
                V[pk][pad](sv, d_out, n); CK(hipStreamSynchronize(sv));
                CK(hipMemcpy(ref.data(), d_out, 4 * n, hipMemcpyDeviceToHost));           // solo run = reference
                int bad_runs = 0;
                for (int r = 0; r < runs; r++) {
                    for (int k = 0; k < 3; k++) {
                        if (mode == 1) hipLaunchKernelGGL((mfma<0, false>), dim3(2048), dim3(512), 0, sa, d_sink, 64);
                        if (mode == 2) hipLaunchKernelGGL((mfma<0, true>), dim3(2048), dim3(512), 0, sa, d_sink, 256);
                        if (mode == 3) hipLaunchKernelGGL((mfma<1, true>), dim3(2048), dim3(512), 0, sa, d_sink, 256);
                        if (mode == 4) hipLaunchKernelGGL((mfma<2, true>), dim3(2048), dim3(512), 0, sa, d_sink, 256);
                    }
                    V[pk][pad](sv, d_out, n);
                    CK(hipMemcpyAsync(got.data(), d_out, 4 * n, hipMemcpyDeviceToHost, sv)); CK(hipStreamSynchronize(sv));
                    int nb = 0;
                    for (int i = 0; i < n; i++) if (got[i] != ref[i] && !(got[i] != got[i] && ref[i] != ref[i])) { nb++; lanes |= 1ull << (i & 63); }
                    bad_runs += nb != 0;
                    if ((r & 7) == 7) CK(hipStreamSynchronize(sa));
                }
                printf(" %d", bad_runs);
            }
            CK(hipDeviceSynchronize());
            printf("   wrong-lane mask %016llx\n", lanes);
            fflush(stdout);
        }
    return 0;
}
