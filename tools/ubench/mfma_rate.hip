// mfma_rate.hip -- what is the dense fp16 MFMA ceiling of THIS chip under load?  (DESIGN said 19 ns per
// v_mfma_f32_32x32x16_f16 per SIMD = 1.7 GHz = 1.7 PFLOP/s; MI355X_MICROARCH.md quotes 2 495 TF measured = 13.4 ns.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o tools/ubench/mfma_rate && tools/ubench/mfma_rate
// Per configuration: G workgroups (one per CU up to 256) x W waves per SIMD x NCH independent accumulator chains per wave, LOOPS x 32
// MFMAs per wave, operands random / zero.  Wave 0 of every workgroup stamps s_memtime (shader clock: it FOLLOWS the DVFS state) and
// s_memrealtime (constant 100 MHz) around its loop, the host brackets the launch with HIP events:
//   cycles per MFMA per SIMD  = d(s_memtime) / (W * LOOPS * 32)         (32 = the pipe is full; > 32 = issue gaps / dependency stalls)
//   effective shader clock    = d(s_memtime) / d(s_memrealtime) * 100 MHz
//   TFLOP/s                   = G * 4 * W * LOOPS * 32 * 32768 flop / wall
// `long SECONDS` runs the full-chip random-data configuration back to back so that rocm-smi can sample the clock beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NCH>
__global__ __launch_bounds__(1024) void bench(const _Float16* in, float* out, unsigned long long* stamp, int loops) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = in[(lane * 8 + i) & 4095]; b[i] = in[(2048 + wave * 64 + lane * 8 + i) & 4095]; }
    f32x16 acc[NCH];
#pragma unroll
    for (int r = 0; r < NCH; r++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[r][k] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < loops; it++) {
#pragma unroll
        for (int i = 0; i < 32 / NCH; i++)
#pragma unroll
            for (int r = 0; r < NCH; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < NCH; r++)
#pragma unroll
        for (int k = 0; k < 16; k++) s += acc[r][k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamp[blockIdx.x * 2] = t1 - t0; stamp[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int NCH>
static int run(int G, int W, int loops, const _Float16* d_in, float* d_out, unsigned long long* d_st, const char* data, bool print = true) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<NCH>, dim3(G), dim3(256 * W), 0, 0, d_in, d_out, d_st, loops / 10 + 1);      // warm
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench<NCH>, dim3(G), dim3(256 * W), 0, 0, d_in, d_out, d_st, loops);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * G);
    CK(hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 2 * G, hipMemcpyDeviceToHost));
    std::vector<double> cyc(G), ghz(G);
    const double per_simd = (double)W * loops * 32;
    for (int g = 0; g < G; g++) { cyc[g] = h[2 * g] / per_simd; ghz[g] = (double)h[2 * g] / (double)h[2 * g + 1] * 0.1; }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double tf = (double)G * 4 * per_simd * 32768.0 / (ms * 1e-3) / 1e12;
    if (print)
        printf("G=%3d W=%d chains=%d %-6s: %6.2f cyc/MFMA/SIMD (median; min %.2f max %.2f)  shader clock %.3f GHz (min %.3f max %.3f)  "
               "%6.2f ns/MFMA/SIMD  wall %.3f ms  %7.1f TFLOP/s\n", G, W, NCH, data, cyc[G / 2], cyc[0], cyc[G - 1], ghz[G / 2], ghz[0], ghz[G - 1],
               cyc[G / 2] / ghz[G / 2], ms, tf);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv) {
    _Float16* d_in[2]; float* d_out; unsigned long long* d_st;
    std::vector<_Float16> h(4096);
    srand(565);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    CK(hipMalloc(&d_in[0], 8192)); CK(hipMalloc(&d_in[1], 8192));
    CK(hipMemcpy(d_in[0], h.data(), 8192, hipMemcpyHostToDevice));
    CK(hipMemset(d_in[1], 0, 8192));
    CK(hipMalloc(&d_out, sizeof(float) * 1024 * 1024)); CK(hipMalloc(&d_st, sizeof(unsigned long long) * 2048));
    if (argc > 2 && !strcmp(argv[1], "long")) {
        const double secs = atof(argv[2]);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        double el = 0;
        int n = 0;
        while (el < secs * 1e3) {
            for (int k = 0; k < 20; k++) hipLaunchKernelGGL(bench<4>, dim3(256), dim3(768), 0, 0, d_in[0], d_out, d_st, 20000);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); el = ms; n += 20;
        }
        printf("long: %d launches of G=256 W=3 chains=4 random in %.1f ms: %.1f TFLOP/s sustained\n", n, el,
               (double)n * 256 * 4 * 3 * 20000 * 32 * 32768.0 / (el * 1e-3) / 1e12);
        return run<4>(256, 3, 20000, d_in[0], d_out, d_st, "random");
    }
    const int loops = 4000;
    for (int z = 0; z < 2; z++) {
        const char* data = z ? "zero" : "random";
        for (int G : {1, 8, 64, 256}) {
            for (int W : {1, 2, 3, 4}) {
                if (run<1>(G, W, loops, d_in[z], d_out, d_st, data)) return 1;
                if (run<2>(G, W, loops, d_in[z], d_out, d_st, data)) return 1;
                if (run<4>(G, W, loops, d_in[z], d_out, d_st, data)) return 1;
            }
        }
    }
    return 0;
}
