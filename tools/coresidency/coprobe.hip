// coprobe.hip -- bisecting the co-residency hazard of DESIGN.md "Known issue" (a bounce kernel beside the split-fp16 conv kernel).
//
// Small self-checking VICTIM kernels (one class of operations each) run again and again on one stream while an AGGRESSOR kernel
// (one hardware feature of conv3x3_f16x3 each) loops on another; a victim run whose output differs from its own solo run is a
// hit.  Built both as an executable (synthetic x synthetic matrix) and as a shared library whose C entry points let
// tools/coresidency/run_matrix.py pair the synthetic kernels with the REAL kernels of libaiptd.so and with torch kernels.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off coprobe.hip -o coprobe
//   hipcc ... -shared -fPIC coprobe.hip -o libcoprobe.so
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #e, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------------------------- victims
enum { V_DIVSQRT, V_FMA, V_CROSSLANE, V_LDS, V_GATHER, V_SCALAR, V_TRANS, V_BRANCHY, V_PKF32, V_CAMRAY, V_PKSEL, V_PKSEL_KARG, V_PKSEL_NOLDS, V_COUNT };
static const char* V_NAME[V_COUNT] = {"divsqrt", "fma", "crosslane", "lds", "gather", "scalar", "trans", "branchy", "pkf32", "camray", "pksel", "pksel_karg", "pksel_nolds"};

__device__ __forceinline__ unsigned hash32(unsigned a) {
    a = (a + 0x7ed55d16u) + (a << 12); a = (a ^ 0xc761c23cu) ^ (a >> 19); a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9); a = (a + 0xfd7046c5u) + (a << 3); a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
__device__ __forceinline__ float unit(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f) + 0.25f; }   // [0.25, 1.25)

// every victim: thread t computes a value from hash(t) with `iters` rounds of ONE class of operations and stores it
template <int KIND>
__global__ __launch_bounds__(256) void victim(const float4* __restrict__ table, int tmask, float* out, int n, int iters) {
    __shared__ float s_buf[(KIND == V_PKSEL_NOLDS || KIND == V_PKSEL_KARG) ? 1 : 256 * 4];
    const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    if (t >= n) return;
    unsigned h = hash32((unsigned)t * 2654435761u + 12345u);
    float x = unit(h), y = unit(hash32(h)), z = unit(hash32(h + 1u)), acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        if (KIND == V_DIVSQRT) {                    // correctly rounded divide / sqrt: v_div_scale, v_rcp, v_div_fmas, v_div_fixup, v_sqrt + fixups
            const float q = x / y, r = sqrtf(q + z), s = (r - 0.5f) / (q + 1.0f);
            acc += s; x = y; y = z; z = 0.25f + fabsf(s - (float)(int)s);
        } else if (KIND == V_FMA) {
            acc = __builtin_fmaf(x, y, acc); x = __builtin_fmaf(y, z, 0.125f) * 0.5f + 0.25f; y = __builtin_fmaf(z, x, 0.0625f) * 0.5f + 0.25f; z = acc * 1e-3f + 0.3f;
        } else if (KIND == V_CROSSLANE) {           // ballot / popcount / readlane-style shuffles / DPP
            const unsigned long long m = __ballot(x > y);
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            const float o = __shfl_xor(x, 1 + (it & 31)), p = __shfl(y, rank & 63);
            const float d = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, z), 0xB1, 0xF, 0xF, false));
            acc += o * 0.5f + p * 0.25f + d * 0.125f + (float)rank;
            x = y; y = z; z = 0.25f + 0.5f * (o - (float)(int)o);
        } else if (KIND == V_LDS) {                 // per-lane LDS addresses, written and read back through a rotation
            s_buf[threadIdx.x * 4 + (it & 3)] = x + (float)it;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const float v = s_buf[((threadIdx.x & ~63) + ((lane + 1 + it) & 63)) * 4 + (it & 3)];
            acc += v; x = y; y = z; z = 0.25f + 0.5f * (v - (float)(int)v);
        } else if (KIND == V_GATHER) {              // dependent 16-byte gathers (the BVH walk's access pattern)
            const float4 v = table[h & (unsigned)tmask];
            acc += v.x + v.y * 0.5f; h = hash32(h ^ __builtin_bit_cast(unsigned, v.z)); x = v.w;
        } else if (KIND == V_SCALAR) {              // wave-uniform loop over a table (scalar loads), per-lane compares: the broad phase
            float best = 1e30f; int bi = -1;
            for (int g = 0; g < 8; g++) {
                const float4 b = table[(it * 8 + g) & tmask];
                const float tn = (b.x - x) * y + (b.y - z) * x;
                if (tn < best) { best = tn; bi = g; }
            }
            acc += best + (float)bi; x = y; y = z; z = 0.25f + 0.5f * (best - (float)(int)best);
        } else if (KIND == V_TRANS) {               // the transcendental unit alone: v_rcp_f32, v_rsq_f32, v_sqrt_f32 approximations
            const float a = __builtin_amdgcn_rcpf(x), b = __builtin_amdgcn_rsqf(y), c = __builtin_amdgcn_sqrtf(z);
            acc += a + b + c; x = y; y = z; z = 0.25f + 0.5f * (a * b - (float)(int)(a * b));
        } else if (KIND == V_PKF32) {               // packed fp32: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on register pairs
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 a = {x, y}, b = {y, z}, c = {z, x};
            a = a * b + c;                                                   // (contraction is off: v_pk_mul_f32 + v_pk_add_f32)
            b = b * (f32x2){-0.5f, 0.75f} + a;
            c = __builtin_elementwise_fma(a, b, c);                          // v_pk_fma_f32
            acc += c[0] + c[1];
            x = 0.25f + 0.5f * (a[0] - (float)(int)a[0]); y = 0.25f + 0.5f * (b[1] - (float)(int)b[1]); z = 0.25f + 0.5f * (c[1] - (float)(int)c[1]);
        } else if (KIND == V_CAMRAY) {              // trace_bounce's generateRayFromCamera arithmetic (uniform camera in SGPRs, per-lane pixel)
            const float4 c0 = table[0], c1 = table[1], c2 = table[2];          // wave-uniform: scalar loads (view | up | right + pixel lengths)
            const int pix = t + it, W = 96;
            const int px = pix % (W + (tmask & 1) - 1), py = pix / (W + (tmask & 1) - 1);      // run-time divisor, as p.W
            const float sx = (float)px - 96.0f * 0.5f, sy = (float)py - 64.0f * 0.5f;
            const float dx = (c0.x - c2.x * c0.w * sx) - c1.x * c1.w * sy;
            const float dy = (c0.y - c2.y * c0.w * sx) - c1.y * c1.w * sy;
            const float dz = (c0.z - c2.z * c0.w * sx) - c1.z * c1.w * sy;
            const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
            acc += dx * inv + (dy * inv) * 2.0f + (dz * inv) * 3.0f;
        } else if (KIND == V_PKSEL || KIND == V_PKSEL_KARG || KIND == V_PKSEL_NOLDS) {   // packed fp32 with scalar / broadcast operands (op_sel forms) and negation modifiers
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            // wave-uniform -> SGPR operands: from a scalar load of global memory, or (KARG) from kernel arguments
            const float4 c0 = KIND == V_PKSEL_KARG ? make_float4((float)tmask * (1.0f / 65536.0f), (float)n * (1.0f / 8192.0f), (float)iters * (1.0f / 512.0f), 0.0f) : table[0];
            f32x2 a = {x, y}, b = {z, x};
            a = a * (f32x2){c0.x, c0.y} + (f32x2){-0.5f, -0.5f};             // SGPR pair operand, inline-constant broadcast
            b = b * (f32x2){a[1], a[1]} - a * (f32x2){c0.z, c0.z};           // op_sel broadcast of a high half, SGPR broadcast, negated product
            a = (f32x2){b[1], b[0]} * a;                                      // swapped halves
            acc += a[0] - a[1];
            x = 0.25f + 0.5f * (a[0] - (float)(int)a[0]); y = 0.25f + 0.5f * (b[1] - (float)(int)b[1]); z = 0.25f + 0.5f * (b[0] - (float)(int)b[0]);
        } else {                                    // divergent control flow with per-lane trip counts and EXEC masks
            int trips = 1 + (int)(h & 7u);
            float w = x;
            while (trips--) { if (w > y) w = w * 0.75f + z; else w = w + y * 0.5f; }
            acc += w; h = hash32(h + (unsigned)it); x = unit(h); y = unit(hash32(h)); z = 0.25f + 0.5f * (w - (float)(int)w);
        }
    }
    out[t] = acc + x;
}

// The suspect: a struct array passed BY VALUE as a kernel argument and indexed with a per-lane (VGPR) index compiles to VECTOR
// loads from the kernel-argument segment (global_load from kernarg base + fr * stride), as trace_bounce's p.cams[fr] did.  The
// host rewrites the argument block for every launch with new values.
struct KArgs { float cam[16][21]; int n; int pad; float* out; };
__global__ __launch_bounds__(256) void victim_kernarg(const KArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int fr = t % a.n;                          // a.n == 1 at run time: always 0, but a VGPR to the compiler
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 21; k++) s += a.cam[fr][k] * (float)(k + 1);
    a.out[t] = s;
}
// the same values through scalar loads (wave-uniform index)
__global__ __launch_bounds__(256) void victim_kernarg_scalar(const KArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 21; k++) s += a.cam[0][k] * (float)(k + 1);
    a.out[t] = s;
}

// ------------------------------------------------------------------------------------------------------------------ aggressors
enum { A_NONE, A_MFMA_F16, A_MFMA_F32, A_MFMA_BF16, A_CVT_DPP, A_LDS_BARRIER, A_VALU, A_STREAM_NT, A_MFMA_F16_LDS, A_PKF32, A_MFMA_F16_PK, A_CONVLIKE, A_MFMA_F16_1ACC, A_COUNT };
static const char* A_NAME[A_COUNT] = {"none", "mfma_f16", "mfma_f32", "mfma_bf16", "cvt_dpp", "lds62k_barrier", "valu", "stream_nt", "mfma_f16_lds", "pkf32", "mfma_f16_pk", "convlike", "mfma_f16_dep"};

template <int KIND>
__global__ __launch_bounds__(512, 4) void aggressor(float* sink, const float4* __restrict__ src, size_t nsrc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(KIND == A_LDS_BARRIER || KIND == A_MFMA_F16_LDS || KIND == A_CONVLIKE) ? 62080 : 16];
    const int tid = threadIdx.x, lane = tid & 63;
    float seed = (float)(tid + 1) * 1e-3f;
    float res = 0.0f;
    if (KIND == A_MFMA_F16 || KIND == A_MFMA_F16_LDS) {
        f16x8 a, b;
        for (int k = 0; k < 8; k++) { a[k] = (_Float16)(seed + k * 0.01f); b[k] = (_Float16)(0.5f - k * 0.01f); }
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; it++) {
            if (KIND == A_MFMA_F16_LDS) {
                *reinterpret_cast<f16x8*>(smem + ((tid * 48) % 61440)) = a;
                __syncthreads();
                b = *reinterpret_cast<const f16x8*>(smem + (((tid + 37) * 48) % 61440));
                __syncthreads();
            }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c3, 0, 0, 0);
        }
        res = c0[0] + c1[5] + c2[9] + c3[15];
    } else if (KIND == A_MFMA_BF16) {
        s16x8 a, b;
        for (int k = 0; k < 8; k++) { a[k] = (short)(0x3f80 + k + lane); b[k] = (short)(0x3f00 + k); }
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; it++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a), c3, 0, 0, 0);
        }
        res = c0[0] + c1[5] + c2[9] + c3[15];
    } else if (KIND == A_MFMA_F32) {
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        const float a = seed, b = 0.5f - seed;
        for (int it = 0; it < iters * 4; it++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, c3, 0, 0, 0);
        }
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (KIND == A_CVT_DPP) {
        float v0 = seed, v1 = seed * 2.0f, v2 = 0.3f, v3 = 0.7f;
        for (int it = 0; it < iters * 16; it++) {
            const auto h01 = __builtin_amdgcn_cvt_pkrtz(v0, v1);
            const auto h23 = __builtin_amdgcn_cvt_pkrtz(v2, v3);
            const float f0 = (float)h01[0], f1 = (float)h01[1], f2 = (float)h23[0], f3 = (float)h23[1];
            v0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f1), 0xB1, 0xF, 0xF, false)) + 0.001f;
            v1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f0), 0x4E, 0xF, 0xF, false)) * 0.999f;
            v2 = (v2 - f2) * 2048.0f + 0.3f; v3 = (v3 - f3) * 2048.0f + 0.7f;
        }
        res = v0 + v1 + v2 + v3;
    } else if (KIND == A_LDS_BARRIER) {
        for (int it = 0; it < iters * 2; it++) {
            *reinterpret_cast<float2*>(smem + ((tid * 48 + (it & 1) * 8) % 61440)) = make_float2(seed + it, seed - it);
            __syncthreads();
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + (((tid * 7 + it) * 48) % 61440));
            res += v[0] + v[3];
            __syncthreads();
        }
    } else if (KIND == A_VALU) {
        float a = seed, b = 0.5f, c = 0.25f;
        for (int it = 0; it < iters * 32; it++) { a = __builtin_fmaf(a, b, c); b = __builtin_fmaf(b, c, a) * 0.5f; c = __builtin_fmaf(c, a, b) * 0.5f; }
        res = a + b + c;
    } else if (KIND == A_PKF32) {                   // packed fp32 VALU only
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 a = {seed, 0.5f}, b = {0.25f, seed}, c = {0.125f, 0.0625f};
        for (int it = 0; it < iters * 32; it++) { a = __builtin_elementwise_fma(a, b, c); b = __builtin_elementwise_fma(b, c, a) * (f32x2){0.5f, 0.5f}; c = __builtin_elementwise_fma(c, a, b) * (f32x2){0.5f, 0.5f}; }
        res = a[0] + b[1] + c[0];
    } else if (KIND == A_MFMA_F16_PK) {             // f16 MFMAs with packed-fp32 VALU work between them, in the same wave
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f16x8 a, b;
        for (int k = 0; k < 8; k++) { a[k] = (_Float16)(seed + k * 0.01f); b[k] = (_Float16)(0.5f - k * 0.01f); }
        f32x16 c0 = {}, c1 = {};
        f32x2 u = {seed, 0.5f}, v = {0.25f, seed}, w = {0.125f, 0.0625f};
        for (int it = 0; it < iters * 2; it++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            u = __builtin_elementwise_fma(u, v, w); v = __builtin_elementwise_fma(v, w, u) * (f32x2){0.5f, 0.5f};
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
            w = __builtin_elementwise_fma(w, u, v) * (f32x2){0.5f, 0.5f};
        }
        res = c0[0] + c1[5] + u[0] + v[1] + w[0];
    } else if (KIND == A_MFMA_F16_1ACC) {           // dependent f16 MFMAs (one accumulator chain per wave, as the conv's per-row chains)
        f16x8 a, b;
        for (int k = 0; k < 8; k++) { a[k] = (_Float16)(seed + k * 0.01f); b[k] = (_Float16)(0.5f - k * 0.01f); }
        f32x16 c0 = {};
        for (int it = 0; it < iters * 4; it++) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        res = c0[0] + c0[7];
    } else if (KIND == A_CONVLIKE) {                // the conv's chunk loop in miniature: loads -> pk math + cvt_pkrtz -> ds_write -> barrier -> ds_read_b128 + MFMAs -> barrier
        f32x16 c0 = {}, c1 = {};
        const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
        size_t gi = ((size_t)blockIdx.x * 512 + tid) % (nsrc - 8192);
        for (int it = 0; it < iters; it++) {
            const f32x4 x = s4[gi], y = s4[gi + 4096];
            gi = (gi + 512 * 97) % (nsrc - 8192);
            const f32x4 v = x * seed + y, vs = v * 0.1f;
            f32x4 m;
            for (int t = 0; t < 4; t++) m[t] = fmaxf(v[t], vs[t]);
            const auto h01 = __builtin_amdgcn_cvt_pkrtz(m[0], m[1]);
            const auto h23 = __builtin_amdgcn_cvt_pkrtz(m[2], m[3]);
            const f32x4 d = (m - (f32x4){(float)h01[0], (float)h01[1], (float)h23[0], (float)h23[1]}) * 2048.0f;
            const auto l01 = __builtin_amdgcn_cvt_pkrtz(d[0], d[1]);
            const auto l23 = __builtin_amdgcn_cvt_pkrtz(d[2], d[3]);
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<f16x4*>(smem + ((tid * 48) % 30720)) = f16x4{(_Float16)h01[0], (_Float16)h01[1], (_Float16)h23[0], (_Float16)h23[1]};
            *reinterpret_cast<f16x4*>(smem + 30720 + ((tid * 48) % 30720)) = f16x4{(_Float16)l01[0], (_Float16)l01[1], (_Float16)l23[0], (_Float16)l23[1]};
            __syncthreads();
            for (int k = 0; k < 4; k++) {
                const f16x8 a = *reinterpret_cast<const f16x8*>(smem + (((tid + k * 5) * 48) % 30720));
                const f16x8 b = *reinterpret_cast<const f16x8*>(smem + 30720 + (((tid + k * 7) * 48) % 30720));
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c1, 0, 0, 0);
            }
            __syncthreads();
        }
        res = c0[0] + c1[5];
    } else if (KIND == A_STREAM_NT) {
        const size_t stride = (size_t)gridDim.x * 512;
        float4 s = make_float4(0, 0, 0, 0);
        for (size_t i = (size_t)blockIdx.x * 512 + tid; i < nsrc; i += stride) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + i);
            s.x += v[0]; s.y += v[1]; s.z += v[2]; s.w += v[3];
        }
        res = s.x + s.y + s.z + s.w;
    }
    if (res == 123456.789f) sink[0] = res;          // keeps the work alive
}

// ------------------------------------------------------------------------------------------------------------------ host side
struct Probe {
    hipStream_t sv = nullptr, sa = nullptr;
    float4* table = nullptr; int tmask = 0;
    float *out = nullptr, *ref = nullptr, *sink = nullptr;
    float4* big = nullptr; size_t nbig = 0;
    unsigned* d_bad = nullptr;
    int n = 0;
};
static Probe g;

__global__ void compare_words(const unsigned* a, const unsigned* b, int n, unsigned* bad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(bad, 1u);
}

static void launch_victim(int kind, hipStream_t st, float* out, int n, int iters) {
    const dim3 grid((n + 255) / 256), blk(256);
    switch (kind) {
#define VCASE(K) case K: hipLaunchKernelGGL((victim<K>), grid, blk, 0, st, g.table, g.tmask, out, n, iters); break;
        VCASE(V_DIVSQRT) VCASE(V_FMA) VCASE(V_CROSSLANE) VCASE(V_LDS) VCASE(V_GATHER) VCASE(V_SCALAR) VCASE(V_TRANS) VCASE(V_BRANCHY) VCASE(V_PKF32) VCASE(V_CAMRAY) VCASE(V_PKSEL) VCASE(V_PKSEL_KARG) VCASE(V_PKSEL_NOLDS)
#undef VCASE
    }
}
static void launch_aggressor(int kind, hipStream_t st, int blocks, int iters) {
    const dim3 grid(blocks), blk(512);
    switch (kind) {
#define ACASE(K) case K: hipLaunchKernelGGL((aggressor<K>), grid, blk, 0, st, g.sink, g.big, g.nbig, iters); break;
        ACASE(A_MFMA_F16) ACASE(A_MFMA_F32) ACASE(A_MFMA_BF16) ACASE(A_CVT_DPP) ACASE(A_LDS_BARRIER) ACASE(A_VALU) ACASE(A_STREAM_NT) ACASE(A_MFMA_F16_LDS) ACASE(A_PKF32) ACASE(A_MFMA_F16_PK) ACASE(A_CONVLIKE) ACASE(A_MFMA_F16_1ACC)
#undef ACASE
        default: break;
    }
}

extern "C" {

// victim stream: `stream` (NULL: an own stream); n threads per victim launch
int coprobe_init(int n) {
    CK(hipStreamCreateWithFlags(&g.sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&g.sa, hipStreamNonBlocking));
    g.n = n;
    g.tmask = (1 << 16) - 1;
    std::vector<float4> t((size_t)g.tmask + 1);
    unsigned s = 777u;
    for (auto& v : t) {
        float f[4];
        for (float& x : f) { s = s * 1664525u + 1013904223u; x = (float)(s >> 8) / 16777216.0f + 0.25f; }
        v = make_float4(f[0], f[1], f[2], f[3]);
    }
    CK(hipMalloc((void**)&g.table, t.size() * sizeof(float4)));
    CK(hipMemcpy(g.table, t.data(), t.size() * sizeof(float4), hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&g.out, sizeof(float) * n));
    CK(hipMalloc((void**)&g.ref, sizeof(float) * n));
    CK(hipMalloc((void**)&g.sink, 64));
    CK(hipMalloc((void**)&g.d_bad, 4));
    g.nbig = (size_t)64 << 20;                       // 1 GiB of float4 for the streaming aggressor
    CK(hipMalloc((void**)&g.big, g.nbig * sizeof(float4)));
    CK(hipMemset(g.big, 0, g.nbig * sizeof(float4)));
    CK(hipDeviceSynchronize());
    return 0;
}
int coprobe_victim_count() { return V_COUNT; }
int coprobe_aggressor_count() { return A_COUNT; }
const char* coprobe_victim_name(int k) { return V_NAME[k]; }
const char* coprobe_aggressor_name(int k) { return A_NAME[k]; }
// solo reference of a victim (nothing else running)
int coprobe_victim_reference(int kind, int iters) {
    CK(hipDeviceSynchronize());
    launch_victim(kind, g.sv, g.ref, g.n, iters);
    CK(hipStreamSynchronize(g.sv));
    return 0;
}
// one victim run on the probe's victim stream (or `stream` when not NULL); returns the number of words that differ from the reference
int coprobe_victim_run(int kind, int iters, void* stream) {
    hipStream_t st = stream ? (hipStream_t)stream : g.sv;
    CK(hipMemsetAsync(g.d_bad, 0, 4, st));
    launch_victim(kind, st, g.out, g.n, iters);
    hipLaunchKernelGGL(compare_words, dim3((g.n + 255) / 256), dim3(256), 0, st, (const unsigned*)g.out, (const unsigned*)g.ref, g.n, g.d_bad);
    unsigned bad = 0;
    CK(hipMemcpyAsync(&bad, g.d_bad, 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    return (int)bad;
}
// queue `count` aggressor launches on the aggressor stream (or `stream`)
int coprobe_aggressor_launch(int kind, int blocks, int iters, int count, void* stream) {
    hipStream_t st = stream ? (hipStream_t)stream : g.sa;
    for (int k = 0; k < count; k++) launch_aggressor(kind, st, blocks, iters);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
// one launch of the kernel-argument victim with argument values derived from `seq` (different every launch, as a camera pan);
// returns the number of threads whose result is not the one THIS launch's arguments give; first[0..1] = first and last bad thread
int coprobe_kernarg_run(int seq, int scalar, void* stream, int* first_last) {
    hipStream_t st = stream ? (hipStream_t)stream : g.sv;
    KArgs a;
    double want = 0.0;
    for (int f = 0; f < 16; f++)
        for (int k = 0; k < 21; k++) a.cam[f][k] = (float)((seq * 7 + f * 3 + k) % 1021) * 0.125f;
    float w = 0.0f;
    for (int k = 0; k < 21; k++) w += a.cam[0][k] * (float)(k + 1);      // exact in fp32: small multiples of 1/8
    (void)want;
    a.n = 1; a.pad = 0; a.out = g.out;
    if (scalar) hipLaunchKernelGGL(victim_kernarg_scalar, dim3((g.n + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(victim_kernarg, dim3((g.n + 255) / 256), dim3(256), 0, st, a);
    std::vector<float> h(g.n);
    CK(hipMemcpyAsync(h.data(), g.out, 4 * (size_t)g.n, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    int bad = 0, lo = -1, hi = -1;
    for (int i = 0; i < g.n; i++) if (h[i] != w) { if (lo < 0) lo = i; hi = i; bad++; }
    if (first_last) { first_last[0] = lo; first_last[1] = hi; }
    return bad;
}
int coprobe_aggressor_sync(void* stream) { CK(hipStreamSynchronize(stream ? (hipStream_t)stream : g.sa)); return 0; }
// first differing words of the last run, for the lane pattern: idx[k], got[k], want[k]
int coprobe_last_diff(int* idx, unsigned* got, unsigned* want, int max) {
    std::vector<unsigned> a(g.n), b(g.n);
    CK(hipMemcpy(a.data(), g.out, 4 * (size_t)g.n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), g.ref, 4 * (size_t)g.n, hipMemcpyDeviceToHost));
    int k = 0;
    for (int i = 0; i < g.n && k < max; i++) if (a[i] != b[i]) { idx[k] = i; got[k] = a[i]; want[k] = b[i]; k++; }
    return k;
}

}  // extern "C"

#ifndef COPROBE_NO_MAIN
// coprobe [runs per pair] [victim threads] [victim iters] : the synthetic x synthetic matrix
int main(int argc, char** argv) {
    const int runs = argc > 1 ? atoi(argv[1]) : 1000, n = argc > 2 ? atoi(argv[2]) : 6144, viters = argc > 3 ? atoi(argv[3]) : 400;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz, memory clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.memoryClockRate);
    coprobe_init(n);
    for (int v = 0; v < V_COUNT; v++) {
        coprobe_victim_reference(v, viters);
        for (int a = 0; a < A_COUNT; a++) {
            int bad_runs = 0, bad_words = 0;
            for (int r = 0; r < runs; r++) {
                if (a != A_NONE) coprobe_aggressor_launch(a, 2048, 64, 3, nullptr);
                const int bad = coprobe_victim_run(v, viters, nullptr);
                if (bad) {
                    if (!bad_runs) {
                        int idx[24]; unsigned got[24], want[24];
                        const int k = coprobe_last_diff(idx, got, want, 24);
                        printf("   first hit, victim %s beside %s: %d words; threads", V_NAME[v], A_NAME[a], bad);
                        for (int j = 0; j < k; j++) printf(" %d", idx[j]);
                        printf("\n");
                    }
                    bad_runs++; bad_words += bad;
                }
                if ((r & 7) == 7) coprobe_aggressor_sync(nullptr);
            }
            coprobe_aggressor_sync(nullptr);
            printf("victim %-10s aggressor %-15s runs %d bad %d (words %d)\n", V_NAME[v], A_NAME[a], runs, bad_runs, bad_words);
            fflush(stdout);
        }
    }
    for (int scalar = 0; scalar < 2; scalar++)
        for (int a = 0; a < A_COUNT; a++) {
            int bad_runs = 0, bad_threads = 0;
            for (int r = 0; r < runs; r++) {
                if (a != A_NONE) coprobe_aggressor_launch(a, 2048, 64, 3, nullptr);
                int fl[2];
                const int bad = coprobe_kernarg_run(r, scalar, nullptr, fl);
                if (bad) { if (!bad_runs) printf("   first hit: %d threads, %d..%d (lanes %d..%d)\n", bad, fl[0], fl[1], fl[0] & 63, fl[1] & 63); bad_runs++; bad_threads += bad; }
                if ((r & 7) == 7) coprobe_aggressor_sync(nullptr);
            }
            coprobe_aggressor_sync(nullptr);
            printf("victim %-10s aggressor %-15s runs %d bad %d (threads %d)\n", scalar ? "karg_sload" : "karg_vload", A_NAME[a], runs, bad_runs, bad_threads);
            fflush(stdout);
        }
    return 0;
}
#endif
