// mfma_fill.hip -- how many of conv3x3_f16x3r's non-MFMA instructions fit in the shadow of its MFMAs?  (VERDICT r5 item 1, step A.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fill.hip -o tools/ubench/mfma_fill && tools/ubench/mfma_fill
// tools/ubench/valu_mfma.hip only ever ran "27 MFMAs THEN n VALU instructions" in one wave (additive by construction for an in-order
// wave) and compared nanoseconds of bodies that run at different clocks.  This one places K fillers BETWEEN consecutive
// v_mfma_f32_32x32x16_f16 of ONE wave and reports CYCLES (s_memtime = shader clock) per MFMA per SIMD, plus the sampled clock
// (s_memtime against the 100 MHz s_memrealtime).  One workgroup per CU on all 256 CUs, W waves per SIMD, random operands.
//   PAT 0  interleaved: MFMA, K fillers, MFMA, K fillers ...                 (what a software-pipelined chunk body would issue)
//   PAT 1  bursts:      12 MFMAs back to back, then 12 K fillers               (what hipcc emits for the kernel today, per wave)
//   NACC   accumulators the 12 MFMAs of a body rotate over (1 = every MFMA depends on the one before it, as the kernel's hi.hi /
//          lo.hi / hi.lo triple does; 3 = the three output rows a halo row feeds)
//   MIX 0  v_fma_f32 only      MIX 1  the transform's VALU mix (fma, mul, max, cvt_pkrtz, fma_mixlo/hi: 9 per 4 channel pairs)
//   MIX 2  MIX 1 + DPP row shifts (9 : 4)      MIX 3  MIX 2 + ds_read_b128 of a weight fragment (9 : 4 : 4, the kernel's ratio)
// Time per SIMD = (last wave's end) - (first wave's start) over the W waves that share it (all waves stamp); floor = 32 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Regs { float v[8]; unsigned u[4]; f32x4 l[2]; };

// filler number n of the stream (compile-time): which instruction it is depends on MIX
template <int MIX, int N>
__device__ __forceinline__ void filler(Regs& r, float s, unsigned ldsaddr) {
    constexpr int PERIOD = MIX == 0 ? 1 : MIX == 1 ? 9 : MIX == 2 ? 13 : 17;
    constexpr int p = N % PERIOD;
    constexpr int a = N % 8, b = (N + 3) % 8, c = N % 4;
    // the order inside a period follows the kernel's row: per channel pair {fma, fma, mul, mul, max, max, cvt_pkrtz, mixlo, mixhi}, DPP
    // shifts and fragment reads spread between the pairs
    constexpr int kind = MIX == 0 ? 0
                       : MIX == 1 ? (p < 2 ? 0 : p < 4 ? 1 : p < 6 ? 2 : p == 6 ? 3 : p == 7 ? 4 : 5)
                       : MIX == 2 ? (p < 2 ? 0 : p == 2 ? 6 : p < 5 ? 1 : p == 5 ? 6 : p < 8 ? 2 : p == 8 ? 6 : p == 9 ? 3 : p == 10 ? 4 : p == 11 ? 6 : 5)
                       : (p < 2 ? 0 : p == 2 ? 6 : p == 3 ? 7 : p < 6 ? 1 : p == 6 ? 6 : p == 7 ? 7 : p < 10 ? 2 : p == 10 ? 6 : p == 11 ? 7 : p == 12 ? 3 : p == 13 ? 4 : p == 14 ? 6 : p == 15 ? 7 : 5);
    if (kind == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r.v[a]) : "v"(r.v[b]), "v"(s));
    if (kind == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r.v[a]) : "v"(r.v[b]), "v"(s));
    if (kind == 2) asm volatile("v_max_f32 %0, %1, %2" : "=v"(r.v[a]) : "v"(r.v[b]), "v"(r.v[(N + 5) % 8]));
    if (kind == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r.u[c]) : "v"(r.v[a]), "v"(r.v[b]));
    if (kind == 4) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r.u[c]) : "v"(r.u[(N + 1) % 4]), "v"(r.v[a]));
    if (kind == 5) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r.u[c]) : "v"(r.u[(N + 1) % 4]), "v"(r.v[a]));
    if (kind == 6) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r.u[c]) : "v"(r.u[(N + 2) % 4]));
    if (kind == 7) asm volatile("s_waitcnt lgkmcnt(2)\n\tds_read_b128 %0, %1" : "=v"(r.l[N & 1]) : "v"(ldsaddr));
}
template <int MIX, int N0, int K, int I = 0>
__device__ __forceinline__ void fillers(Regs& r, float s, unsigned ldsaddr) {
    if constexpr (I < K) { filler<MIX, N0 + I>(r, s, ldsaddr); fillers<MIX, N0, K, I + 1>(r, s, ldsaddr); }
}

template <int K, int NACC, int PAT, int MIX, int I = 0>
__device__ __forceinline__ void body(f32x16 (&acc)[3], const f16x8& a, const f16x8& b, Regs& r, float s, unsigned ldsaddr) {
    if constexpr (I < 12) {
        acc[I % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[I % NACC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (PAT == 0) { fillers<MIX, I * K, K>(r, s, ldsaddr); __builtin_amdgcn_sched_barrier(0); }
        body<K, NACC, PAT, MIX, I + 1>(acc, a, b, r, s, ldsaddr);
    } else if constexpr (PAT == 1) {
        fillers<MIX, 0, 12 * K>(r, s, ldsaddr);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int K, int NACC, int PAT, int MIX>
__global__ __launch_bounds__(768) void bench(const _Float16* in, float* out, unsigned long long* stamp, int loops) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)in[i];
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = in[(lane * 8 + i) & 4095]; b[i] = in[(2048 + wave * 64 + lane * 8 + i) & 4095]; }
    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[q][k] = 0.f;
    Regs r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = 0.5f + 0.001f * (lane + k);
#pragma unroll
    for (int k = 0; k < 4; k++) r.u[k] = 0x3c003c00u + lane + k;
    r.l[0] = r.l[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned ldsaddr = (unsigned)(lane * 16);
    const float s = 0.9990234375f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < loops; it++) body<K, NACC, PAT, MIX>(acc, a, b, r, s, ldsaddr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int k = 0; k < 16; k++) t += acc[q][k];
#pragma unroll
    for (int k = 0; k < 8; k++) t += r.v[k];
#pragma unroll
    for (int k = 0; k < 4; k++) t += (float)r.u[k];
    t += r.l[0][0] + r.l[1][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (lane == 0) {
        unsigned long long* st = stamp + ((size_t)blockIdx.x * 12 + wave) * 4;
        st[0] = t0; st[1] = t1; st[2] = r0; st[3] = r1;
    }
}

struct Res { double cyc, ghz, ns; };
template <int K, int NACC, int PAT, int MIX>
static int run(int W, const _Float16* d_in, float* d_out, unsigned long long* d_st, Res* res) {
    const int G = 256, loops = 3000;
    hipLaunchKernelGGL((bench<K, NACC, PAT, MIX>), dim3(G), dim3(256 * W), 0, 0, d_in, d_out, d_st, loops / 10 + 1);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((bench<K, NACC, PAT, MIX>), dim3(G), dim3(256 * W), 0, 0, d_in, d_out, d_st, loops);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)G * 12 * 4);
    CK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, ghz;
    for (int g = 0; g < G; g++) {
        unsigned long long lo = ~0ull, hi = 0, rlo = ~0ull, rhi = 0;
        for (int w = 0; w < 4 * W; w++) {
            const unsigned long long* st = &h[((size_t)g * 12 + w) * 4];
            lo = std::min(lo, st[0]); hi = std::max(hi, st[1]); rlo = std::min(rlo, st[2]); rhi = std::max(rhi, st[3]);
        }
        cyc.push_back((double)(hi - lo) / ((double)W * loops * 12));
        ghz.push_back((double)(hi - lo) / (double)(rhi - rlo) * 0.1);
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    res->cyc = cyc[G / 2]; res->ghz = ghz[G / 2]; res->ns = ms * 1e6 / ((double)W * loops * 12);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

template <int NACC, int PAT, int MIX>
static int sweep(const char* name, const _Float16* d_in, float* d_out, unsigned long long* d_st) {
    for (int W : {1, 3}) {
        Res r[9];
        if (run<0, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[0])) return 1;
        if (run<1, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[1])) return 1;
        if (run<2, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[2])) return 1;
        if (run<3, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[3])) return 1;
        if (run<4, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[4])) return 1;
        if (run<5, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[5])) return 1;
        if (run<6, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[6])) return 1;
        if (run<7, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[7])) return 1;
        if (run<8, NACC, PAT, MIX>(W, d_in, d_out, d_st, &r[8])) return 1;
        printf("%-28s acc=%d %-11s W=%d  cycles/MFMA/SIMD:", name, NACC, PAT ? "bursts" : "interleaved", W);
        for (int k = 0; k <= 8; k++) printf(" %6.1f", r[k].cyc);
        printf("   | GHz:");
        for (int k = 0; k <= 8; k++) printf(" %.2f", r[k].ghz);
        printf("   | ns:");
        for (int k = 0; k <= 8; k++) printf(" %.1f", r[k].ns);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

int main() {
    _Float16* d_in; float* d_out; unsigned long long* d_st;
    std::vector<_Float16> h(4096);
    srand(565);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    CK(hipMalloc(&d_in, 8192)); CK(hipMemcpy(d_in, h.data(), 8192, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, sizeof(float) * 256 * 768)); CK(hipMalloc(&d_st, sizeof(unsigned long long) * 256 * 12 * 4));
    printf("columns: K = 0 .. 8 fillers per MFMA; 256 CUs x W waves per SIMD, 12 MFMAs per body, random operands\n");
    if (sweep<3, 0, 0>("v_fma_f32", d_in, d_out, d_st)) return 1;
    if (sweep<1, 0, 0>("v_fma_f32", d_in, d_out, d_st)) return 1;
    if (sweep<3, 1, 0>("v_fma_f32", d_in, d_out, d_st)) return 1;
    if (sweep<3, 0, 1>("transform VALU mix", d_in, d_out, d_st)) return 1;
    if (sweep<1, 0, 1>("transform VALU mix", d_in, d_out, d_st)) return 1;
    if (sweep<3, 1, 1>("transform VALU mix", d_in, d_out, d_st)) return 1;
    if (sweep<3, 0, 2>("VALU mix + DPP", d_in, d_out, d_st)) return 1;
    if (sweep<3, 1, 2>("VALU mix + DPP", d_in, d_out, d_st)) return 1;
    if (sweep<3, 0, 3>("VALU mix + DPP + ds_read_b128", d_in, d_out, d_st)) return 1;
    if (sweep<1, 0, 3>("VALU mix + DPP + ds_read_b128", d_in, d_out, d_st)) return 1;
    if (sweep<3, 1, 3>("VALU mix + DPP + ds_read_b128", d_in, d_out, d_st)) return 1;
    return 0;
}
