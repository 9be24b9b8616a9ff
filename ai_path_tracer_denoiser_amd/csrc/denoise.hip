// denoise.hip -- recurrent denoising auto-encoder forward pass on CDNA4 (gfx950), hand-written HIP.
//
// Replaces torch::jit::load + module.forward (reference Inference/src/main.cpp:104-111); the arithmetic is the model of
// training/recurrent_autoencoder_model.py:8-142 (28 x conv3x3+bias, 28 x BatchNorm2d, LeakyReLU(0.1), 5 x MaxPool2d(2),
// 5 x nearest Upsample x2, skip concats, 6 recurrent hidden states).
//
// Data layout in HBM: every activation is fp32 in the channel-quad interleaved "C4" layout [C/4][h][w][4] (see ConvSrc);
// the planar G-buffer input is read as it is by the first conv.
// Each conv writes its RAW output (conv + bias, optionally LeakyReLU for the encoder's conv->LReLU->BN order) once and adds its
// per-channel sum / sum of squares to the layer's statistics table (two-word fixed-point atomics, 8 replicas); the CONSUMER turns
// them into the per-channel affine (a, b) = (gamma/sqrt(var+eps), beta - mean*a) in its prologue (BnRef / bn_ab) and applies
// x -> lrelu(a*x+b) while it stages its input, so BatchNorm, LeakyReLU, channel concat (two source pointers) and nearest upsample
// (source indexed at (y>>1, x>>1)) never touch HBM as separate passes and there is no launch between two convs.  Zero padding is
// applied in the normalised domain (out-of-image taps load 0, not f(0)).  MaxPool is the 2x2 max (min where gamma < 0) of the raw
// output in the producing conv's epilogue; the network's last BatchNorm + crop is the second pass of the output layer.
//
// Kernels: conv3x3_f16x3r -- the two big levels (>= 368 x 640, the planar-input first conv and the depth-to-space form of dec1.c1
// included): persistent, register-staged implicit GEMM on v_mfma_f32_32x32x16_f16 with fp32 operands split into fp16 hi/lo pairs,
// weights resident in LDS, no barriers; conv3x3_f16x3 -- the same arithmetic LDS-tiled and barrier-phased (the levels below; 4-row
// tiles with three waves per row where all of them fit the chip at once); conv3x3_mfma -- the same GEMM on v_mfma_f32_16x16x4_f32
// (exact fp32 products and accumulation: AIPT_DN_IMPL_MFMA, and levels beyond the fp16 operand range); conv3x3_quad -- the 3 -> 3
// output layer as two streaming passes.  Which kernel runs a level: run_conv + aipt_denoise_set_option (include/aiptd.h); no
// environment variable takes part.  Rooflines and measurements: DESIGN.md.
#include "internal.h"

#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>
#include <algorithm>

namespace aipt {

constexpr int KC = 8;            // input channels per LDS chunk (2 MFMA k-steps)
constexpr int NLAYERS = 28;
constexpr float BN_EPS = 1e-5f;
constexpr float SLOPE = 0.1f;

static const int ENC_CH[5] = {32, 43, 57, 76, 101};
static const int DEC_CH[6] = {0, 3, 32, 43, 57, 76};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Activation layout in HBM: channel-QUAD interleaved, float[ceil(C/4)][h][w][4] ("C4"): one 16-byte load gives a lane the
// four channels of a pixel -- the unit the conv kernels stage (measured 6.1 TB/s of halo fetch vs 2.9 TB/s with 4-byte
// loads from planar tensors, tools/membench.hip).  Pad channels of the last quad hold 0.  Only the network input (the
// G-buffer contract is planar, pathtrace.cu:81-94) and the API-facing outputs are planar.
// How a consumer obtains the per-channel affine (a, b) of a tensor (y = lrelu(a*x + b)).  Batch-statistics BatchNorm: the
// producing conv left per-channel sums in `stat` (64-bit atomics into NSLOT replicas, slot = workgroup % NSLOT: 3680
// workgroups x 64 atomics cost 1.6 us that way, tools/atomicbench.hip) and every consumer workgroup turns them into
// (a, b) itself in its prologue -- there is no finalize launch between two convs (28 launches x 4.1 us per frame).
// The sums are FIXED-POINT integers, two 64-bit words per sum (integer part + 40 fractional bits): integer addition is
// associative, so the statistics -- and with them every output bit -- do not depend on the order in which the workgroups'
// atomics land.  (fp64 sums of fp32 partials are exact only while all partials fit one 53-bit window; a channel with a wide
// spread of partial magnitudes broke run-to-run equality about once in 15 runs.)  Range: |sum| < 9.2e18, whatever the scale
// of the tensor (rounds 2-3 kept ONE word with 24 / 20 fractional bits: a sum of squares past 8.8e12 -- an rms of 3 000 at
// 720p, which a G-buffer in centimetres reaches after the first conv -- wrapped silently).  A partial is rounded to 2^-40 once.
// Running statistics / identity: `ab` (or nothing).
constexpr int NSLOT = 8;
constexpr int BN_WORDS = 4;                                   // per channel: {sum: integer part, fraction * 2^40, sum of squares: the same}
constexpr double BN_FRAC = 1099511627776.0, BN_FRAC_INV = 1.0 / 1099511627776.0;   // 2^40
struct BnRef {
    const float2* ab;      // explicit affine; with stat == nullptr and ab == nullptr: identity
    const long long* stat; // [NSLOT][sc][BN_WORDS]
    const float* gamma;
    const float* beta;
    int sc;                // channel stride of stat
    double inv_n;          // 1 / (pixels the sums run over)
};
__device__ __forceinline__ float2 bn_ab(const BnRef& r, int c) {
    if (r.stat) {
        // all replicas are loaded before the first add: written as one accumulate loop, hipcc waits for each 16-byte load
        // before issuing the next (8 serial L2 round trips, 7 k cycles per consumer workgroup)
        typedef long long l2 __attribute__((ext_vector_type(2)));
        const l2* st = reinterpret_cast<const l2*>(r.stat) + (size_t)c * 2;
        l2 v[NSLOT][2];
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            v[k][0] = __builtin_nontemporal_load(st + (size_t)k * r.sc * 2);
            v[k][1] = __builtin_nontemporal_load(st + (size_t)k * r.sc * 2 + 1);
        }
        long long xi = 0, xf = 0, qi = 0, qf = 0;
#pragma unroll
        for (int k = 0; k < NSLOT; k++) { xi += v[k][0][0]; xf += v[k][0][1]; qi += v[k][1][0]; qf += v[k][1][1]; }
        const double sx = (double)xi + (double)xf * BN_FRAC_INV, sxx = (double)qi + (double)qf * BN_FRAC_INV;
        const double mean = sx * r.inv_n;
        double var = sxx * r.inv_n - mean * mean;          // biased variance, as torch normalises with
        if (var < 0) var = 0;
        // the cancellation-prone part (mean, variance) is fp64; scale and shift are fp32 like the tensors they multiply
        const float sc = r.gamma[c] * rsqrtf((float)(var + 1e-5));
        return make_float2(sc, fmaf(-(float)mean, sc, r.beta[c]));
    }
    return r.ab ? r.ab[c] : make_float2(1.0f, 0.0f);
}
// a (workgroup's) sum -> its two fixed-point words; clamped so that NaN / inf stay defined
struct BnFix { long long i, f; };
__device__ __forceinline__ BnFix bn_fix(double v) {
    v = fmin(fmax(v, -9.0e18), 9.0e18);
    if (!(v == v)) v = 0.0;
    const double fl = floor(v);
    BnFix r;
    r.i = __double2ll_rn(fl);
    r.f = __double2ll_rn((v - fl) * BN_FRAC);              // [0, 2^40]
    return r;
}
// a workgroup's BN sums of channel c (already reduced over the workgroup) -> the producer's stat table
__device__ __forceinline__ void bn_accumulate_slot(long long* stat, int sc, int slot, int c, double sum, double sumsq) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(stat + ((size_t)slot * sc + c) * BN_WORDS);
    const BnFix x = bn_fix(sum), q = bn_fix(sumsq);
    atomicAdd(a, (unsigned long long)x.i);
    atomicAdd(a + 1, (unsigned long long)x.f);
    atomicAdd(a + 2, (unsigned long long)q.i);
    atomicAdd(a + 3, (unsigned long long)q.f);
}
__device__ __forceinline__ void bn_accumulate(long long* stat, int sc, int c, float sum, float sumsq) {
    bn_accumulate_slot(stat, sc, blockIdx.x % NSLOT, c, (double)sum, (double)sumsq);
}

struct ConvSrc {
    const float* p;      // C4: [ceil(C/4)][sh][sw][4]   (planar != 0: [C][sh][sw])
    BnRef bn;            // per-channel affine
    int C;
    int up;              // 1: stored at half resolution, nearest-upsampled on load
    float slope;         // LeakyReLU slope applied after the affine (1 = none)
    int planar;
};
__host__ __device__ __forceinline__ int pad4(int c) { return (c + 3) & ~3; }
// Channel concat (a's channels first) is indexed in the PADDED-CONCAT space pc: [0, pad4(a.C)) is a, then b; weights are laid
// out over pc with zero rows for the pad channels, so a channel quad never straddles the two sources.

struct ConvArgs {
    ConvSrc a, b;        // channel concat: a's channels first
    int H, W;            // conv (output) resolution
    const float* w;      // [nchunks][9][KC][NP]
    const float* w_raw;  // [cout][cin][3][3] (VALU cross-check kernel)
    const float* bias;   // [NP]
    int cin, cout, NP, nchunks;
    float* out;          // [cout][H][W] raw
    int out_lrelu;
    long long* stat;     // BN sums of the output (nullptr: not wanted), [NSLOT][sc][BN_WORDS]
    int sc;
    int d2s;             // depth-to-space store: virtual channel v = (2a+b)*d2s + j goes to out[j][2y+a][2x+b]
    int tiles_x, tiles_y, groups;   // pixel tiles and output-channel groups of the launch (1-D XCD-aware grid)
    // conv3x3_quad<.., false>: the layer's own normalisation and the planar, cropped network output
    BnRef out_bn;
    float out_slope;
    float* out_planar;   // [cout][oh][ow]
    int oh, ow;
};

// 1-D grid -> (pixel tile, output-channel group), XCD-aware.  Workgroup b runs on XCD b % 8 (observed dispatch order;
// used for speed only).  Each XCD gets a CONTIGUOUS range of pixel tiles and walks it with the channel group as the
// fastest index, so (a) neighbouring tiles' halos and (b) the same tile's input for the next channel group are served
// from that XCD's L2 instead of being fetched again (the per-XCD L2s are not shared).
struct TileId { int tx, ty, gz, lin; bool valid; };
__device__ __forceinline__ TileId tile_of_block(int tiles_x, int tiles_y, int groups) {
    TileId t;
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int ntiles = tiles_x * tiles_y, per = (ntiles + 7) >> 3;
    t.gz = slot % groups;
    const int tl = slot / groups;
    t.lin = xcd * per + tl;
    t.valid = tl < per && t.lin < ntiles;
    t.ty = t.lin / tiles_x;
    t.tx = t.lin - t.ty * tiles_x;
    return t;
}
static inline unsigned grid_1d(int tiles_x, int tiles_y, int groups) {
    return 8u * (unsigned)((tiles_x * tiles_y + 7) / 8) * (unsigned)groups;
}

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.0f ? v : v * slope; }

__device__ __forceinline__ float load_src(const ConvSrc& s, int ch, int y, int x, int H, int W) {
    const int sh = s.up ? (H >> 1) : H, sw = s.up ? (W >> 1) : W;
    const int sy = s.up ? (y >> 1) : y, sx = s.up ? (x >> 1) : x;
    float v = s.planar ? s.p[((size_t)ch * sh + sy) * sw + sx]
                       : s.p[(((size_t)(ch >> 2) * sh + sy) * sw + sx) * 4 + (ch & 3)];
    const float2 ab = bn_ab(s.bn, ch);
    v = fmaf(ab.x, v, ab.y);
    return lrelu(v, s.slope);
}

// per padded-concat channel (a, b) table in LDS; pad channels and channels past the end get (0, 0) so they stage as 0
__device__ __forceinline__ void fill_abs_tab(float2* abs_tab, const ConvSrc& a, const ConvSrc& b, int entries, int tid,
                                             int nthreads) {
    const int PA = pad4(a.C);
    for (int pc = tid; pc < entries; pc += nthreads) {
        float2 t = make_float2(0.0f, 0.0f);
        if (pc < PA) { if (pc < a.C) t = bn_ab(a.bn, pc); }
        else if (pc - PA < b.C) t = bn_ab(b.bn, pc - PA);
        abs_tab[pc] = t;
    }
}
// The four values of a (channel quad, pixel) unit.  C4 sources: ONE 16-byte load; the source of the quad (a or b of the
// concat) is chosen with selects on the pointer/offset -- never a branch around the load (hipcc would wait vmcnt(0) per
// unit).  Offsets fit 32 bits (quads x plane < 2^31).  (The planar network input is converted to C4 by planar_to_c4 first.)
__device__ __forceinline__ float4 load_quad_c4(const float4* a4, const float4* b4, int pq, int PAq, unsigned plane, int goff) {
    const bool fa = pq < PAq;
    const unsigned off = (unsigned)(fa ? pq : pq - PAq) * plane + (unsigned)goff;
    const float4* base = fa ? a4 : b4;
    return base[off];
}
// 4x4 transpose across a quad of lanes: in: lane t holds r[i] = value(pixel i, channel t); out: r[i] = value(pixel t, channel i)
__device__ __forceinline__ void quad_transpose(float (&r)[4], int lane) {
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const float send = (lane & 1) ? r[i] : r[i + 1];
        const float recv = __shfl_xor(send, 1);
        if (lane & 1) r[i] = recv; else r[i + 1] = recv;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float send = (lane & 2) ? r[i] : r[i + 2];
        const float recv = __shfl_xor(send, 2);
        if (lane & 2) r[i] = recv; else r[i + 2] = recv;
    }
}

// -------------------------------------------------------------------------------------------------- MFMA conv
// Block = 256 threads = 4 waves.  Tile = (4*RW) rows x (16*MBX) cols of output pixels x (16*NBB) output channels
// (blockIdx.z selects the channel group).  Wave w owns rows [w*RW, w*RW+RW) of the tile.
//
// Software pipeline over 8-channel chunks, LDS double-buffered, ONE barrier per chunk:
//     issue global loads of chunk c+1 (halo tile + weight slab) into registers
//     9 taps x 2 k-steps of MFMA on chunk c from LDS[cur]
//     transform (BN affine + LReLU) and write chunk c+1 into LDS[cur^1];  barrier
// so HBM/L2 latency hides under the MFMAs of the same workgroup.  The (channel, row, col) -> address decomposition of a
// thread's staging elements does not depend on the chunk and is done once before the loop.
template <int RW, int MBX, int NBB>
struct ConvCfg {
    static constexpr int TH = 4 * RW, TW = 16 * MBX;
    static constexpr int RS = TW + 2;                               // LDS row stride of the halo tile
    static constexpr int PL = (TH + 2) * RS;                        // halo pixels per channel
    static constexpr int CS = PL + ((16 - (PL % 32)) + 32) % 32;    // channel stride == 16 (mod 32): k and k+1 hit disjoint banks
    static constexpr int NPB0 = 16 * NBB;
    static constexpr int NPB = NPB0 + ((16 - (NPB0 % 32)) + 32) % 32;
    static constexpr int A_FLOATS = KC * CS;
    static constexpr int B_FLOATS = 9 * KC * NPB;
    static constexpr int STAGE = A_FLOATS + B_FLOATS;               // one pipeline stage
    static constexpr int NU = (KC / 4 * PL + 255) / 256;            // (channel quad, halo pixel) units per thread
    static constexpr int NW = (9 * KC * NBB * 4 + 255) / 256;       // weight float4 per thread
    static constexpr int MAXC = 208;                                // >= padded-concat channels (104 + 104), multiple of KC
};

template <int RW, int MBX, int NBB>
__global__ __launch_bounds__(256, (RW * MBX * NBB >= 12) ? 2 : (RW * MBX * NBB >= 4) ? 3 : 4) void conv3x3_mfma(const ConvArgs g) {
    using Cfg = ConvCfg<RW, MBX, NBB>;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, RS = Cfg::RS, PL = Cfg::PL, CS = Cfg::CS, NPB = Cfg::NPB;
    constexpr int NU = Cfg::NU, NW = Cfg::NW;
    __shared__ __attribute__((aligned(16))) float smem[2 * Cfg::STAGE + 2 * Cfg::MAXC];
    float2* abs_tab = reinterpret_cast<float2*>(smem + 2 * Cfg::STAGE);   // per concat channel (a, b)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TileId tile = tile_of_block(g.tiles_x, g.tiles_y, g.groups);
    if (!tile.valid) return;
    const int tx0 = tile.tx * TW, ty0 = tile.ty * TH;
    const int n0 = tile.gz * NBB * 16;         // first output channel of this block
    const int H = g.H, W = g.W;
    const int lk = lane >> 4, li = lane & 15;
    const int PA = pad4(g.a.C), pcin = PA + pad4(g.b.C);   // padded-concat channel space
    const int up = g.a.up;                     // both sources share the resampling mode (host checks)
    const int sw = up ? (W >> 1) : W;
    const size_t plane = (size_t)(up ? (H >> 1) : H) * sw;
    const float4* a4 = reinterpret_cast<const float4*>(g.a.p);
    const float4* b4 = reinterpret_cast<const float4*>(g.b.p ? g.b.p : g.a.p);

    fill_abs_tab(abs_tab, g.a, g.b, g.nchunks * KC, tid, 256);

    // ---- chunk-invariant part of this thread's staging units: unit u -> (channel quad q of the chunk, halo pixel).
    // Loads are issued UNCONDITIONALLY from clamped addresses and masked afterwards (abs_tab holds (0,0) for pad
    // channels): a branch around a load makes hipcc wait vmcnt(0) per element.
    static_assert(CS > PL + 3, "the padding floats of a channel serve as the dump slot of out-of-range staging units");
    int u_goff[NU], u_lds[NU], u_q[NU];
    unsigned in_mask = 0;
#pragma unroll
    for (int j = 0; j < NU; j++) {
        const int u = tid + j * 256;
        const int q = u / PL;
        const int rem = u - q * PL;
        const int yy = rem / RS, xx = rem - yy * RS;
        const int y = ty0 + yy - 1, x = tx0 + xx - 1;
        const bool valid = u < KC / 4 * PL;
        const bool in = valid && y >= 0 && y < H && x >= 0 && x < W;
        u_q[j] = valid ? q : 0;
        u_lds[j] = valid ? q * 4 * CS + yy * RS + xx : PL;     // channel t of the quad at + t*CS
        u_goff[j] = in ? (up ? (y >> 1) * sw + (x >> 1) : y * sw + x) : 0;
        in_mask |= in ? (1u << j) : 0u;
    }

    f32x4 acc[RW][MBX][NBB];
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
        for (int m = 0; m < MBX; m++)
#pragma unroll
            for (int n = 0; n < NBB; n++) acc[r][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 pa[NU];
    f32x4 pw[NW];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NU; j++) {
            int pq = chunk * (KC / 4) + u_q[j];                 // quad index in the padded-concat space
            pq = pq * 4 < pcin ? pq : pcin / 4 - 1;
            pa[j] = load_quad_c4(a4, b4, pq, PA / 4, (unsigned)plane, u_goff[j]);
        }
        const float* wsrc = g.w + (size_t)chunk * 9 * KC * g.NP + n0;
#pragma unroll
        for (int j = 0; j < NW; j++) {
            int e = tid + j * 256;
            e = e < 9 * KC * NBB * 4 ? e : 9 * KC * NBB * 4 - 1;
            const int row = e / (NBB * 4), q = e - row * (NBB * 4);
            pw[j] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)row * g.NP + q * 4);
        }
    };
    auto stash = [&](int chunk, float* As, float* Bs) {
#pragma unroll
        for (int j = 0; j < NU; j++) {
            const int pc = (chunk * (KC / 4) + u_q[j]) * 4;
            const bool ok = (in_mask >> j) & 1u;
            const float slope = pc < PA ? g.a.slope : g.b.slope;
            const float raw[4] = {pa[j].x, pa[j].y, pa[j].z, pa[j].w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float2 ab = abs_tab[pc + t];
                const float v = lrelu(fmaf(ab.x, raw[t], ab.y), slope);
                As[u_lds[j] + t * CS] = ok ? v : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int e = tid + j * 256;
            const int row = e / (NBB * 4), q = e - row * (NBB * 4);
            if (e < 9 * KC * NBB * 4) *reinterpret_cast<f32x4*>(Bs + row * NPB + q * 4) = pw[j];
        }
    };

    fetch(0);
    __syncthreads();                           // abs_tab visible
    stash(0, smem, smem + Cfg::A_FLOATS);
    __syncthreads();

    for (int chunk = 0; chunk < g.nchunks; chunk++) {
        const float* As = smem + (chunk & 1) * Cfg::STAGE;
        const float* Bs = As + Cfg::A_FLOATS;
        const bool more = chunk + 1 < g.nchunks;
        if (more) fetch(chunk + 1);
        // ---- 9 taps x 2 k-steps of v_mfma_f32_16x16x4_f32
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int ks = 0; ks < KC / 4; ks++) {
                float bf[NBB];
#pragma unroll
                for (int n = 0; n < NBB; n++) bf[n] = Bs[(tap * KC + ks * 4 + lk) * NPB + n * 16 + li];
#pragma unroll
                for (int r = 0; r < RW; r++) {
                    const float* arow = As + (ks * 4 + lk) * CS + (wave * RW + r + ky) * RS + li + kx;
#pragma unroll
                    for (int m = 0; m < MBX; m++) {
                        const float af = arow[m * 16];
#pragma unroll
                        for (int n = 0; n < NBB; n++)
                            acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[n], acc[r][m][n], 0, 0, 0);
                    }
                }
            }
        }
        if (more) {
            float* An = smem + ((chunk + 1) & 1) * Cfg::STAGE;
            stash(chunk + 1, An, An + Cfg::A_FLOATS);
        }
        __syncthreads();
    }

    // ---- epilogue: bias (+LReLU), store raw output (C4 layout), per-channel sum / sum-of-squares partials
    // D fragment: register q of lane l holds pixel 4*(l>>4)+q of the 16-pixel block, output channel l&15.  Lanes 4k..4k+3
    // hold the four channels of quad k for the same four pixels: a 4x4 lane-quad transpose gives every lane the four
    // channels of ONE pixel = one 16-byte store.
    float s1[NBB], s2[NBB];
#pragma unroll
    for (int n = 0; n < NBB; n++) { s1[n] = 0.f; s2[n] = 0.f; }
#pragma unroll
    for (int n = 0; n < NBB; n++) {
        const int j = n0 + n * 16 + li;
        const bool jok = j < g.cout;
        const float bj = g.bias[n0 + n * 16 + li];
        const bool quad_ok = (j & ~3) < g.cout;                 // the quad holds at least one real channel
#pragma unroll
        for (int r = 0; r < RW; r++) {
            const int y = ty0 + wave * RW + r;
#pragma unroll
            for (int m = 0; m < MBX; m++) {
                const int xb = tx0 + m * 16 + lk * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float t = acc[r][m][n][q] + bj;
                    if (g.out_lrelu) t = lrelu(t, SLOPE);
                    v[q] = jok ? t : 0.0f;                      // pad channels of the last quad are stored as 0
                    if (jok && y < H && xb + q < W) { s1[n] += t; s2[n] += t * t; }
                }
                if (g.d2s) {
                    if (jok && y < H) {
                        const int par = j / g.d2s, real = j - par * g.d2s;
                        float* o = g.out + (((size_t)(2 * y + (par >> 1)) * (2 * W)) + (par & 1)) * 4 + real;
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (xb + q < W) o[(size_t)2 * (xb + q) * 4] = v[q];
                    }
                } else {
                    quad_transpose(v, lane);                    // now v[i] = channel (j&~3)+i of pixel xb + (lane&3)
                    const int x = xb + (lane & 3);
                    if (quad_ok && y < H && x < W)
                        *reinterpret_cast<float4*>(g.out + (((size_t)(j >> 2) * H + y) * W + x) * 4) =
                            make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    if (g.stat) {
        float2* red = reinterpret_cast<float2*>(smem);     // [4 waves][NBB*16]; the last loop barrier already passed
#pragma unroll
        for (int n = 0; n < NBB; n++) {
            float a = s1[n], b = s2[n];
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if (lk == 0) red[wave * (NBB * 16) + n * 16 + li] = make_float2(a, b);
        }
        __syncthreads();
        if (tid < NBB * 16) {
            const int j = n0 + tid;
            if (j < g.cout) {
                float2 t = red[tid];
                for (int w = 1; w < 4; w++) { t.x += red[w * (NBB * 16) + tid].x; t.y += red[w * (NBB * 16) + tid].y; }
                bn_accumulate(g.stat, g.sc, g.d2s ? j % g.d2s : j, t.x, t.y);   // d2s: the 4 parities of a channel share its sums
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------- split-fp16 MFMA conv
// Same implicit GEMM, but on v_mfma_f32_32x32x16_f16 with every fp32 operand split as x = hi + lo * 2^-11
// (hi = fp16(x), lo = fp16((x - hi) * 2^11)); three MFMAs per product keep hi*hi, hi*lo and lo*hi, accumulated in fp32
// (the dropped lo*lo term is 2^-22 relative).  Measured accuracy equals the fp32 FMA chain, at 16/3 of its
// MFMA rate.  M = 32 consecutive pixels of a row, N = 32 output channels, K = 16 input channels of one tap.
// Block = NWV waves, tile = (NWV*RW) rows x 32 cols x 32 output channels (one channel group of 32 per workgroup).
//
// The kernel is VALU-bound before it is MFMA- or HBM-bound (PMC: 15 VALU instructions per MFMA in the first version), so
// everything around the MFMAs is arranged to cost no vector ALU work:
//   * K16-aligned concat space: source a owns chunks [0, ca16), source b chunks [ca16, ca16+cb16) -- the source of a chunk
//     is wave-uniform, its base pointer lives in SGPRs and advances by a scalar per chunk; a thread's vector offset
//     (pixel, quad-of-the-chunk) is computed once per tile;
//   * staging unit = (halo pixel, channel quad): the quad of a thread is fixed (tid / (NT/4)), so its four (a,b) BN
//     coefficients are two LDS reads per chunk, and consecutive lanes read consecutive 16-byte C4 pixels;
//   * the LDS halo image is zeroed once per tile; out-of-image units never write (EXEC mask, no selects);
//   * weights are pre-split and pre-tiled on the host as [group][chunk][hi|lo][tap][32 cout][16] halfs, so a chunk's slab
//     is a linear 18 KB copy;
//   * bias is the initial accumulator, pad output channels have zero weights and bias (no masking), and the BN partial
//     sums skip the image-bounds test on interior tiles.
// LDS holds activations channel-last as [pixel][16 hi halfs | pad] and [pixel][16 lo halfs | pad] (48-byte pixel stride:
// ds_read_b128 / ds_write_b64 conflict-free) and the weight slab [tap][cout][16 | pad] hi and lo.
// Single LDS stage + register prefetch: the next chunk's global loads are in flight during the MFMAs.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Two accumulators: acc0 takes hi * hi, acc1 the two cross terms with the low halves scaled by 2^11 (normal fp16 numbers down to
// |x| = 2^-25: no absolute floor that matters); result = acc0 + acc1 * 2^-11.
// PLANAR network input (identity transform, unknown units -- HDR radiance, first-hit distance in scene units): staged times
// XSP = 2^-4 and scaled back in the epilogue, so that the operand split holds |x| up to 2^20 (1.0e6) instead of 65 504.
constexpr float LO_SCALE = 2048.0f;
constexpr float XSP = 0.0625f;
constexpr int KH = 16;             // input channels per chunk of the fp16 kernel
constexpr int PXB = 48;            // bytes per pixel / per weight row in LDS (16 halfs + 8 halfs padding)
constexpr int WSLAB = 2 * 9 * 32 * KH * 2;   // bytes of one (group, chunk) weight slab: hi and lo

static inline int pad16(int c) { return (c + 15) & ~15; }

struct ConvArgsH {
    ConvSrc a, b;
    int H, W;
    const unsigned char* wsplit;   // [group][chunk][hi|lo][9][32][16] halfs
    const float* bias;             // [coutp], zero for pad channels
    int cout, coutp;
    int nchunks, ca16;             // chunks to run; chunks of source a
    int wchunks;                   // chunks per group in wsplit
    float* out;
    int out_lrelu;
    long long* stat;               // BN sums of the output (nullptr: not wanted), [NSLOT][sc][BN_WORDS]
    int sc;
    int d2s;                       // depth-to-space store: virtual channel v = (2a+b)*d2s + j goes to pixel (2y+a, 2x+b), channel j
    int tiles_x, tiles_y, groups;
    // fused 2x2 max pool of the consumers' view of this output (encoder blocks): the consumers see lrelu(a*x + b) with
    // a = gamma * rstd, which is monotone in x with the sign of gamma -- known at launch -- so the pooled tensor is the RAW
    // output's 2x2 max (gamma >= 0) or min (gamma < 0), normalised by its consumers like any raw tensor: the same value,
    // bit for bit, as pooling the normalised tensor, without a pass over it (pool2_norm: 5 launches, 57 us per frame)
    float* pool_out;               // [ceil(cout/4)][H/2][W/2][4] or nullptr
    const float* pool_gamma;       // [cout]
    int ablate;                    // -DAIPT_CONV_ABLATE builds only (tools/conv_ablate.sh): bit mask of the parts to leave out
};

template <int RW, int NWV, int KYS = 1>
struct ConvCfgH {
    static constexpr int TH = NWV * RW, TW = 32, NT = NWV * KYS * 64;
    static constexpr int RS = TW + 2;
    static constexpr int PL = (TH + 2) * RS;
    static constexpr int A_BYTES = PL * PXB;                 // one of hi / lo
    static constexpr int B_BYTES = 9 * 32 * PXB;             // one of hi / lo
    static constexpr int TPQ = NT / 4;                       // threads per channel quad
    static constexpr int NU = (PL + TPQ - 1) / TPQ;          // halo pixels per thread
    static constexpr int NWP = (WSLAB / 16 + NT - 1) / NT;   // 16-byte weight pieces per thread
    static constexpr int MAXC = 224;                         // K16 concat channels (2 x pad16(101))
};

// quad-lane exchange on the VALU (DPP quad_perm), no LDS round trip
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
}
// 4x4 transpose across a quad of lanes: in: lane t holds r[i] = value(pixel i, channel t); out: r[i] = value(pixel t, channel i)
__device__ __forceinline__ void quad_transpose_dpp(float (&r)[4], int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const float p0 = dpp_xor1(r[i]), p1 = dpp_xor1(r[i + 1]);
        r[i] = o1 ? p1 : r[i];
        r[i + 1] = o1 ? r[i + 1] : p0;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float p0 = dpp_xor2(r[i]), p1 = dpp_xor2(r[i + 2]);
        r[i] = o2 ? p1 : r[i];
        r[i + 2] = o2 ? r[i + 2] : p0;
    }
}

// -DAIPT_CONV_PHASES: cycle stamps of wave 0 of every workgroup of the launches whose (W, nchunks, cout) match g_conv_key,
// summed per phase (tools/conv_phases.py); compiled out of the product build.
#ifdef AIPT_CONV_PHASES
__device__ unsigned long long g_conv_phase[16];
__device__ unsigned int g_conv_key;
#define CPH_INIT() const bool cph_on = (threadIdx.x == 0) && g_conv_key == (((unsigned)g.W << 16) | ((unsigned)g.nchunks << 8) | (unsigned)g.cout); \
    unsigned long long cph_t = __builtin_amdgcn_s_memtime(), cph_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long cph_t0 = cph_t
#define CPH(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); cph_acc[k] += now_ - cph_t; cph_t = now_; } while (0)
#define CPH_WAITVM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define CPH_END() do { if (cph_on) { for (int k_ = 0; k_ < 14; k_++) atomicAdd(&g_conv_phase[k_], cph_acc[k_]); \
    atomicAdd(&g_conv_phase[14], __builtin_amdgcn_s_memtime() - cph_t0); atomicAdd(&g_conv_phase[15], 1ull); } } while (0)
#else
#define CPH_INIT() do {} while (0)
#define CPH(k) do {} while (0)
#define CPH_WAITVM() do {} while (0)
#define CPH_END() do {} while (0)
#endif

// PLANAR: source a is a planar tensor [C][h][w] (the network input = the G-buffer contract): four 4-byte loads per staging
// unit instead of one 16-byte load, no separate layout pass over the input.
// W16: fp16 conv weights (BASELINE configs[4], AIPT_DN_IMPL_MFMA_F16W): the weights are the fp16 roundings the hi half
// of the slab already holds, so the lo half is neither staged nor multiplied (2 MFMAs per product, half the weight traffic).
// KYS = 3 (4-row tiles of the small levels): THREE waves per tile row, one per tap row ky -- twelve waves share a chunk's staging
// (a third of the transform work each) and run 9 MFMAs per chunk instead of 27; the three partial accumulators of a row are
// summed through LDS once, in a fixed order, by the ky = 0 wave, which also runs the epilogue.  The launches of the small levels
// have fewer workgroups than the chip has CUs and last as long as ONE workgroup's chain of chunk steps: this
// shortens the step.
template <int RW, int NWV, bool PLANAR = false, bool W16 = false, int KYS = 1>
__global__ __launch_bounds__(NWV * KYS * 64, KYS == 3 ? 3 : NWV == 8 ? 4 : 2) void conv3x3_f16x3(const ConvArgsH g) {
    static_assert(KYS == 1 || (KYS == 3 && RW == 1), "tap-row split: one row per wave");
    using Cfg = ConvCfgH<RW, NWV, KYS>;
    constexpr int NT = Cfg::NT, TPQ = Cfg::TPQ;
    constexpr int WP = W16 ? WSLAB / 32 : WSLAB / 16;          // 16-byte weight pieces of a chunk that are staged
    constexpr int TH = Cfg::TH, RS = Cfg::RS, PL = Cfg::PL, NU = Cfg::NU, NWP = Cfg::NWP;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * Cfg::A_BYTES + 2 * Cfg::B_BYTES + 8 * Cfg::MAXC];
    unsigned char* Ahi = smem;
    unsigned char* Alo = smem + Cfg::A_BYTES;
    unsigned char* Bhi = smem + 2 * Cfg::A_BYTES;
    float* tab_a = reinterpret_cast<float*>(smem + 2 * Cfg::A_BYTES + 2 * Cfg::B_BYTES);   // BN scale per K16 channel
    float* tab_b = tab_a + Cfg::MAXC;                                                       // BN shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = KYS == 1 ? tid >> 6 : (tid >> 6) % NWV;      // the tile row of this wave
    const int kyg = KYS == 1 ? 0 : (tid >> 6) / NWV;              // KYS = 3: its tap row
    const bool primary = kyg == 0;                                // runs the epilogue
    const TileId tile = tile_of_block(g.tiles_x, g.tiles_y, g.groups);
    if (!tile.valid) return;
    CPH_INIT();
#ifdef AIPT_CONV_ABLATE
    if (g.ablate & 512) return;    // 256: no BN table of the input, 512: an empty workgroup
    const int abl = g.ablate;      // 1: no LDS reads + MFMAs, 2: no activation loads, 4: no weight loads, 8: no transform + LDS writes,
#else                              // 16: no output stores, 32: no BN sums, 64: LDS reads but no MFMAs, 128: no barriers in the chunk loop
    constexpr int abl = 0;
#endif
    const int tx0 = tile.tx * 32, ty0 = tile.ty * TH;
    const int n0 = tile.gz * 32;
    const int H = g.H, W = g.W;
    const int li = lane & 31, lg = lane >> 5;
    const int up = g.a.up;
    const int sw = up ? (W >> 1) : W;
    const unsigned plane16 = (unsigned)((up ? (H >> 1) : H) * sw) * 16u;      // bytes of one channel quad
    const int ca16 = g.ca16;

    // chunk-invariant staging units: this thread stages channel quad q of halo pixels slot, slot+TPQ, ...
    const int q = tid / TPQ, slot = tid - q * TPQ;
    unsigned u_off[NU];          // byte offset of the pixel inside one channel-quad plane
    int u_lds[NU];
    bool u_in[NU];
#pragma unroll
    for (int j = 0; j < NU; j++) {
        const int pix = slot + j * TPQ;
        const int yy = pix / RS, xx = pix - yy * RS;
        const int y = ty0 + yy - 1, x = tx0 + xx - 1;
        const bool in = pix < PL && y >= 0 && y < H && x >= 0 && x < W;
        u_in[j] = in;
        u_lds[j] = pix * PXB + q * 8;
        u_off[j] = in ? (unsigned)(up ? (y >> 1) * sw + (x >> 1) : y * sw + x) * 16u : 0u;
    }
    const unsigned q_off = (unsigned)q * plane16;
    // weight pieces: piece p of the slab -> LDS [hi|lo][tap*32+cout][16 B half]
    int w_lds[NWP];
#pragma unroll
    for (int j = 0; j < NWP; j++) {
        const int p = tid + j * NT;
        const int hl = p >= 9 * 32 * 2;
        const int pp = hl ? p - 9 * 32 * 2 : p;
        w_lds[j] = hl * Cfg::B_BYTES + (pp >> 1) * PXB + (pp & 1) * 16;
    }
    const unsigned char* wslab = g.wsplit + (size_t)tile.gz * g.wchunks * WSLAB + tid * 16;

    const float bj = g.bias[n0 + li];
    f32x16 acc0[RW], acc1[RW];
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
        for (int k = 0; k < 16; k++) { acc0[r][k] = !primary ? 0.f : PLANAR ? bj * XSP : bj; acc1[r][k] = 0.f; }

    // register sets of prefetched chunks: ONE chunk ahead.  (Two ahead with KYS = 3 -- a set is 16 registers there -- measured no
    // faster, 17.0 vs 16.3 us on enc5.l1: the chunk step is not waiting for memory.)
    constexpr int PFD = 1;
    f32x4 pa_sets[PFD][NU];
    u32x4 pw_sets[PFD][NWP];
#ifdef AIPT_CONV_ABLATE
#pragma unroll
    for (int u = 0; u < PFD; u++) {
#pragma unroll
        for (int j = 0; j < NU; j++) pa_sets[u][j] = f32x4{0.5f, 0.25f, 0.125f, 1.0f};
#pragma unroll
        for (int j = 0; j < NWP; j++) pw_sets[u][j] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    }
#endif
    auto fetch = [&](int chunk, f32x4 (&pa)[NU], u32x4 (&pw)[NWP]) {
        const bool fa = chunk < ca16;
        const int cl = fa ? chunk : chunk - ca16;                         // chunk inside its source
        const ConvSrc& s = fa ? g.a : g.b;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(s.p) + (size_t)cl * 4 * plane16;
        const int nq = (pad4(s.C) >> 2) - cl * 4;                          // real quads in this chunk (>= 1)
        const unsigned qo = q < nq ? q_off : 0u;                           // pad quads re-read quad 0 (their a,b are 0)
        if (PLANAR) {
            const float* pl = g.a.p;
            const size_t plane = plane16 >> 4;                              // elements of one channel plane
#pragma unroll
            for (int j = 0; j < NU; j++) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    int ch = chunk * KH + q * 4 + t;
                    ch = ch < g.a.C ? ch : g.a.C - 1;                       // pad channels: any finite value (their a, b are 0)
                    pa[j][t] = pl[(size_t)ch * plane + (u_off[j] >> 4)];
                }
            }
        } else if (!(abl & 2)) {
#pragma unroll
            for (int j = 0; j < NU; j++) pa[j] = *reinterpret_cast<const f32x4*>(base + (u_off[j] + qo));
        }
        const unsigned char* wsrc = wslab + (size_t)chunk * WSLAB;
        if (!(abl & 4)) {
#pragma unroll
        for (int j = 0; j < NWP; j++)
            if ((j + 1) * NT <= WP || tid + j * NT < WP)      // wave-uniform (WP is a multiple of 64)
                pw[j] = *reinterpret_cast<const u32x4*>(wsrc + j * NT * 16);
        }
    };
    auto stash = [&](int chunk, const f32x4 (&pa)[NU], const u32x4 (&pw)[NWP]) {
        if (abl & 8) return;
        const float slope = chunk < ca16 ? g.a.slope : g.b.slope;
        const f32x4 ca = *reinterpret_cast<const f32x4*>(tab_a + chunk * KH + q * 4);
        const f32x4 cb = *reinterpret_cast<const f32x4*>(tab_b + chunk * KH + q * 4);
#pragma unroll
        for (int j = 0; j < NU; j++) {
            const f32x4 x = ca * pa[j] + cb;
            const f32x4 xs = x * slope;
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] = fmaxf(x[t], xs[t]);        // LeakyReLU for 0 < slope <= 1
            if (PLANAR) {
                // the network input in the caller's units, times 2^-4: saturates at +-65 504 (|x| = 1 048 064) exactly like the
                // register-staged kernel's planar conv, so that a frame means the same whichever kernel its size selects
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = __builtin_amdgcn_fmed3f(v[t], -65504.0f, 65504.0f);
            }
            // hi = fp16(v) rounded toward zero (any fp16 near v works: lo carries the exact remainder, scaled by 2^11)
            const f16x2 h01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[0], v[1]));
            const f16x2 h23 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[2], v[3]));
            const f32x4 hf = {(float)h01[0], (float)h01[1], (float)h23[0], (float)h23[1]};
            const f32x4 d = (v - hf) * LO_SCALE;
            const f16x2 l01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(d[0], d[1]));
            const f16x2 l23 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(d[2], d[3]));
            if (u_in[j]) {
                *reinterpret_cast<f16x4*>(Ahi + u_lds[j]) = f16x4{h01[0], h01[1], h23[0], h23[1]};
                *reinterpret_cast<f16x4*>(Alo + u_lds[j]) = f16x4{l01[0], l01[1], l23[0], l23[1]};
            }
        }
#pragma unroll
        for (int j = 0; j < NWP; j++)
            if ((j + 1) * NT <= WP || tid + j * NT < WP)
                *reinterpret_cast<u32x4*>(Bhi + w_lds[j]) = pw[j];
    };

    CPH(0);
    fetch(0, pa_sets[0], pw_sets[0]);
    if (PFD == 2 && g.nchunks > 1) fetch(1, pa_sets[PFD - 1], pw_sets[PFD - 1]);
    CPH(1);
    CPH_WAITVM();
    CPH(10);
    // BN coefficient table over the K16 concat space ((0,0) for pad channels), behind the first chunk's loads: its own
    // round trip to the statistics (10 k cycles of a 55 k-cycle workgroup when it ran first) overlaps theirs
    for (int kc = tid; kc < g.nchunks * KH; kc += NT) {
        const bool fa = kc < ca16 * KH;
        const int c = fa ? kc : kc - ca16 * KH;
        const ConvSrc& s = fa ? g.a : g.b;
        float2 t = make_float2(0.0f, 0.0f);
        if (c < s.C && !(abl & 256)) t = bn_ab(s.bn, c);
        tab_a[kc] = PLANAR ? t.x * XSP : t.x;                  // (LeakyReLU is positively homogeneous: the scale commutes)
        tab_b[kc] = PLANAR ? t.y * XSP : t.y;
    }
    CPH(11);
    // zeroed halo image: only tiles whose halo leaves the image need it (out-of-image units never write); an interior tile
    // overwrites every byte the MFMA phase reads, every chunk
    const bool halo_inside = ty0 >= 1 && ty0 + TH + 1 <= H && tx0 >= 1 && tx0 + 33 <= W;
    if (!halo_inside)
        for (int i = tid; i < 2 * Cfg::A_BYTES / 16; i += NT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
    CPH(12);
    __syncthreads();                                           // tables and the zeroed halo visible
    CPH(2);
    for (int chunk0 = 0; chunk0 < g.nchunks; chunk0 += PFD) {
#pragma unroll
    for (int u = 0; u < PFD; u++) {
        const int chunk = chunk0 + u;
        if (chunk >= g.nchunks) break;                             // (workgroup-uniform)
        CPH_WAITVM();
        CPH(3);
        stash(chunk, pa_sets[u], pw_sets[u]);
        CPH(4);
        if (!(abl & 128)) __syncthreads();
        CPH(5);
        if (chunk + PFD < g.nchunks) fetch(chunk + PFD, pa_sets[u], pw_sets[u]);
        CPH(6);
        if (!(abl & 1)) {
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            f16x8 fah[RW + 2], fal[RW + 2];
#pragma unroll
            for (int hr = 0; hr < RW + 2; hr++) {
                if (KYS == 3 && hr != 0) continue;                   // (KYS = 3: slot 0 holds this wave's one halo row, wave + kyg)
                const int off = ((wave * RW + hr + (KYS == 3 ? kyg : 0)) * RS + li + kx) * PXB + lg * 16;
                fah[hr] = *reinterpret_cast<const f16x8*>(Ahi + off);
                fal[hr] = *reinterpret_cast<const f16x8*>(Alo + off);
            }
#pragma unroll
            for (int kyi = 0; kyi < (KYS == 3 ? 1 : 3); kyi++) {
                const int ky = KYS == 3 ? kyg : kyi;
                const int boff = ((ky * 3 + kx) * 32 + li) * PXB + lg * 16;
                const f16x8 fbh = *reinterpret_cast<const f16x8*>(Bhi + boff);
                const f16x8 fbl = *reinterpret_cast<const f16x8*>(Bhi + Cfg::B_BYTES + boff);
#ifdef AIPT_CONV_ABLATE
                if (abl & 64) {                                  // fragments read (and kept alive), nothing multiplied
#pragma unroll
                    for (int r = 0; r < RW; r++) asm volatile("" :: "v"(fah[KYS == 3 ? 0 : r + kyi]), "v"(fal[KYS == 3 ? 0 : r + kyi]), "v"(fbh), "v"(fbl));
                    continue;
                }
#endif
#pragma unroll
                for (int r = 0; r < RW; r++) {
                    const int fr = KYS == 3 ? 0 : r + kyi;
                    acc0[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[fr], fbh, acc0[r], 0, 0, 0);
                    if (!W16) acc1[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[fr], fbl, acc1[r], 0, 0, 0);
                    acc1[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[fr], fbh, acc1[r], 0, 0, 0);
                }
            }
        }
        }
        CPH(7);
        if (!(abl & 128)) __syncthreads();
        CPH(8);
    }
    }

    // ---- epilogue.  D fragment (32x32): register k of lane l = pixel (k&3) + 8*(k>>2) + 4*(l>>5), channel l&31.
    // Lanes 4c..4c+3 hold the four channels of quad c for the same four pixels; a lane-quad transpose turns that into
    // one 16-byte C4 store per lane.  Pad channels (j >= cout) come out as exact zeros (zero weights and bias).
    const int j = n0 + li;
    const bool quad_ok = (j & ~3) < g.cout;
    const bool interior = ty0 + TH <= H && tx0 + 32 <= W;     // block-uniform
    float s1 = 0.f, s2 = 0.f;
    // KYS = 3: the tap rows' partial sums meet in LDS (behind the slots the epilogue uses; the loop's last barrier has passed):
    // [ky - 1][row][register][lane], added by the ky = 0 wave in the order ky = 0, 1, 2 whatever the waves' timing was
    float* kred = reinterpret_cast<float*>(smem) + 3072;
    static_assert(KYS == 1 || 3072 * 4 + (KYS - 1) * NWV * 1024 * 4 <= (int)sizeof(smem), "tap-row partial sums must fit the staging area");
    if (KYS == 3) {
        if (!primary) {
            const f32x16 t = acc0[0] + acc1[0] * (1.0f / 2048.0f);
#pragma unroll
            for (int k = 0; k < 16; k++) kred[(((kyg - 1) * NWV + wave) * 16 + k) * 64 + lane] = t[k];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int y = ty0 + wave * RW + r;
        f32x16 t = acc0[r] + acc1[r] * (1.0f / 2048.0f);
        if (KYS == 3 && primary) {
#pragma unroll
            for (int kk = 1; kk < KYS; kk++)
#pragma unroll
                for (int k = 0; k < 16; k++) t[k] += kred[(((kk - 1) * NWV + wave) * 16 + k) * 64 + lane];
        }
        if (PLANAR) t = t * (1.0f / XSP);
        if (g.out_lrelu) {
#pragma unroll
            for (int k = 0; k < 16; k++) t[k] = fmaxf(t[k], t[k] * SLOPE);
        }
        if (!primary) {
            // (tap-row helpers: nothing to sum or store; they still meet the barriers below)
        } else if (interior) {
#pragma unroll
            for (int k = 0; k < 16; k++) { s1 += t[k]; s2 = fmaf(t[k], t[k], s2); }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int x = tx0 + 8 * (k >> 2) + 4 * lg + (k & 3);
                const float m = (y < H && x < W) ? t[k] : 0.0f;
                s1 += m; s2 = fmaf(m, m, s2);
            }
        }
        if (RW == 1 && g.pool_out) {
            // 2x2 pool of the raw output: horizontal pairs are register pairs (pixels 2m, 2m+1 of this lane's 16), vertical
            // pairs are the rows of waves 2v and 2v+1 -- through LDS (behind the BN reduction slots; the loop's last barrier
            // has passed); the even wave stores 16 pooled pixels x 32 channels as C4 quads
            const bool pos = j >= g.cout || !(g.pool_gamma[j] < 0.0f);
            float hv[8];
#pragma unroll
            for (int m = 0; m < 8; m++) hv[m] = pos ? fmaxf(t[2 * m], t[2 * m + 1]) : fminf(t[2 * m], t[2 * m + 1]);
            float* pbuf = reinterpret_cast<float*>(smem) + 1024 + wave * 8 * 64;
            if ((wave & 1) && primary) {
#pragma unroll
                for (int m = 0; m < 8; m++) pbuf[m * 64 + lane] = hv[m];
            }
            __syncthreads();
            if (!(wave & 1) && primary) {
                const float* qbuf = pbuf + 8 * 64;                  // the odd wave below
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const float o = qbuf[m * 64 + lane];
                    hv[m] = pos ? fmaxf(hv[m], o) : fminf(hv[m], o);
                }
                // pooled pixel of hv[m]: x' = 4*(m>>1) + 2*lg + (m&1); lanes 4c..4c+3 hold channels 4c..4c+3
                const int hh = H >> 1, hw = W >> 1;
                float* prow = g.pool_out + (((size_t)(j >> 2) * hh + (y >> 1)) * hw + (tx0 >> 1)) * 4;
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    float v[4] = {hv[half * 4], hv[half * 4 + 1], hv[half * 4 + 2], hv[half * 4 + 3]};
                    quad_transpose_dpp(v, lane);
                    const int mi = half * 4 + (lane & 3);
                    const int xp = 4 * (mi >> 1) + 2 * lg + (mi & 1);
                    if (quad_ok && y < H && tx0 + 2 * xp < W && !(abl & 16))
                        *reinterpret_cast<f32x4*>(prow + xp * 4) = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
        }
        if (!primary) continue;
        if (g.d2s) {
            // upsample+conv as a half-resolution conv: the 4 * d2s (<= 16) virtual channels of a pixel are the d2s real
            // channels of its four full-resolution children.  Through LDS (this wave's own 2 KB; the loop's last barrier
            // has passed): [pixel][16 channels], then one 16-byte C4 store per (pixel, child).
            float* buf = reinterpret_cast<float*>(smem) + 1024 + (wave * RW + r) * 32 * 16;    // behind the BN reduction slots
            if (li < 16) {
#pragma unroll
                for (int k = 0; k < 16; k++) buf[(8 * (k >> 2) + 4 * lg + (k & 3)) * 16 + li] = t[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int o = lane + 64 * h, px = o >> 2, par = o & 3;
                const float* c = buf + px * 16 + par * g.d2s;
                const int x = tx0 + px;
                if (y < H && x < W)
                    *reinterpret_cast<f32x4*>(g.out + (((size_t)(2 * y + (par >> 1)) * (2 * W)) + 2 * x + (par & 1)) * 4) =
                        f32x4{c[0], g.d2s > 1 ? c[1] : 0.0f, g.d2s > 2 ? c[2] : 0.0f, g.d2s > 3 ? c[3] : 0.0f};
            }
            continue;
        }
        float* orow = g.out + (((size_t)(j >> 2) * H + y) * W + tx0 + 4 * lg + (lane & 3)) * 4;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            float v[4] = {t[qq * 4], t[qq * 4 + 1], t[qq * 4 + 2], t[qq * 4 + 3]};
            quad_transpose_dpp(v, lane);
            const int x = tx0 + 8 * qq + 4 * lg + (lane & 3);
            if (quad_ok && y < H && x < W && !(abl & 16)) *reinterpret_cast<f32x4*>(orow + 32 * qq) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
    if (g.stat && !(abl & 32)) {
        float2* red = reinterpret_cast<float2*>(smem);         // [NWV waves][32]; the last loop barrier already passed
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (lg == 0 && primary) red[wave * 32 + li] = make_float2(s1, s2);
        __syncthreads();
        if (tid < 32) {
            const int jj = n0 + tid;
            if (jj < g.cout) {
                float2 t = red[tid];
                for (int w = 1; w < NWV; w++) { t.x += red[w * 32 + tid].x; t.y += red[w * 32 + tid].y; }
                bn_accumulate(g.stat, g.sc, g.d2s ? jj % g.d2s : jj, t.x, t.y);   // d2s: a channel's four children share its sums
            }
        }
    }
    CPH(9);
    CPH_END();
}

// -------------------------------------------------------------------------------------------------- split-fp16 conv, register-staged
// conv3x3_f16x3r: the big levels' kernel.  Same split-fp16 arithmetic as conv3x3_f16x3, a different machine mapping, chosen from
// the round-3 ablation of that kernel (tools/conv_ablate.sh, profiles/r03_conv_ablate.txt): its parts -- input loads, transform +
// LDS writes, LDS reads + MFMAs, stores, and a per-workgroup start-up of ~8 k cycles of serial latencies -- ADD UP (128 us on
// enc1.l2a = 31 floor + 39 loads + 13 staging + 36 MFMA + 10 stores) instead of overlapping: two barrier-phased workgroups per CU
// march in lockstep, and every tile pays the start-up again.  Here nothing is phased and nothing is paid per tile:
//   * PERSISTENT, ONE WORKGROUP PER CU, 12 INDEPENDENT WAVES: a wave owns whole items (4 output rows x 30 pixels x 32 output
//     channels) and never meets a barrier after the prologue; twelve waves per CU at different points of their items overlap loads,
//     vector work, LDS reads and MFMAs statistically -- what the hardware scheduler is for;
//   * WEIGHTS RESIDENT IN LDS: the group's hi/lo slabs of ALL chunks (<= 8 x 18 KB) are copied once per workgroup, XOR-swizzled
//     (conflict-free ds_read_b128); the per-tile 18 KB-per-chunk weight traffic through L1 + LDS writes is gone;
//   * ACTIVATIONS NEVER TOUCH LDS: in the C4 layout one 16-byte load gives a lane the 4 channels of a pixel, which IS the MFMA
//     operand layout (lane = pixel, k-group = lane >> 5: 8 channels = two loads); the lane transforms (BN affine, LeakyReLU),
//     splits into fp16 hi/lo and holds the fragment.  The +-1 column taps are the same registers shifted by one lane (DPP
//     wave_shr / wave_shl: 8 moves per shift): a wave loads pixels X..X+31 of a halo row and produces outputs X+1..X+30 (the two
//     edge rows of the 32-row MFMA operand are discarded: 6 % of the MFMA work instead of LDS staging);
//   * halo row h of an item feeds output rows h-2..h: up to 27 MFMAs (3 column taps x 3 rows x 3 split products) per transformed
//     row, weights re-read from LDS per row (0.67 ds_read_b128 per MFMA instead of 1.33);
//   * loads run THREE halo rows ahead of their use through a register ring, across chunk and item boundaries;
//   * ONE fp32 accumulator per output row (activations x 2^4, weights x 2^7, low halves unscaled: the round-2 -DAIPT_ONE_ACC
//     arithmetic): 4 rows = 64 VGPRs, three waves per SIMD;
//   * weights are the MFMA's A operand and activations its B operand, so a lane's accumulators are 16 output channels of ONE
//     pixel, four consecutive channels per register quad: the C4 store is four plain 16-byte stores (no lane transposes), the
//     2x2 pool is a register max between rows and one lane shift, and the BN sums are a 5-step butterfly per item kept in a
//     64-bit fixed-point register pair per lane until the workgroup ends (32 x 2 atomics per workgroup, order-independent).
constexpr int RR_PX = 30;                                     // valid output pixels of an item's row (32 loaded)
constexpr float XS1 = 16.0f, WS1 = 128.0f;
constexpr int RR_MAXCH = 8;                                   // chunks whose weights fit LDS (fp16-weight mode: twice as many)
#ifdef AIPT_R_CLOCK
// debug builds (tools/build_variant.sh ... -DAIPT_R_CLOCK): per layer {first wave start, last wave start, first wave end, last wave
// end, after the prologue barrier (last), main loop end (last)} on the chip-wide 100 MHz clock, one wave's shader-clock count and its
// 100 MHz count: the timeline of a forward pass without a profiler (run_conv prints it after the last layer)
__device__ unsigned long long g_rclk[32][8];
__device__ unsigned long long g_rclk_w[32][4096][4];       // per wave: start, after the prologue barrier, main loop end, (unused)
#endif
#ifndef AIPT_R_WAVES
#define AIPT_R_WAVES 12                                       // waves per workgroup of the C4-input instantiations (A/B builds: 8)
#endif
constexpr int R_NWV = AIPT_R_WAVES, R_PF = 3;
static inline size_t convr_lds_bytes(int nchunks, bool w16) {
    return (size_t)nchunks * (w16 ? WSLAB / 2 : WSLAB) + (size_t)nchunks * KH * 8 + 32 * 4 + 32 * BN_WORDS * 8 + 32;
}

// lo half of the split: the fp16 roundings of v0 - hi.lo and v1 - hi.hi, packed, in two mixed-precision FMAs (instead of two
// conversions, two subtractions and a pack: VALU instructions add to the MFMA time of a SIMD, tools/ubench/valu_mfma.hip)
__device__ __forceinline__ unsigned split_lo_mix(unsigned hi, float v0, float v1) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(v1));
    return d;
}
// the same for the two-accumulator arithmetic of the planar input: the fp16 roundings of (v - hi) * 2^11, from vs = v * 2^11 and
// nscale = -2^11 (exact in fp32: hi is v rounded to 11 bits -- to nearest in the planar stash, so the remainder may be negative --
// and the remainder has at most 13)
__device__ __forceinline__ unsigned split_lo_mix_scaled(unsigned hi, float v0s, float v1s, float nscale) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %3, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hi), "v"(v0s), "v"(nscale));
    asm("v_fma_mixhi_f16 %0, %1, %3, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(v1s), "v"(nscale));
    return d;
}
__device__ __forceinline__ unsigned dpp_wave_shr1(unsigned v) {   // lane i <- lane i - 1
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned dpp_wave_shl1(unsigned v) {   // lane i <- lane i + 1
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
}
// sum over the 32 lanes of a half-wave of 16 per-lane values: a halving butterfly (31 exchanges instead of 80); lane m ends with the
// total of value k = 8 (m & 1) + 4 (m >> 1 & 1) + 2 (m >> 2 & 1) + (m >> 3 & 1)
__device__ __forceinline__ float halfwave_sum16(const float (&s)[16], int m) {
    const bool o1 = m & 1, o2 = m & 2, o4 = m & 4, o8 = m & 8;
    float a8[8], a4[4], a2[2];
#pragma unroll
    for (int j = 0; j < 8; j++) a8[j] = (o1 ? s[8 + j] : s[j]) + dpp_xor1(o1 ? s[j] : s[8 + j]);
#pragma unroll
    for (int j = 0; j < 4; j++) a4[j] = (o2 ? a8[4 + j] : a8[j]) + dpp_xor2(o2 ? a8[j] : a8[4 + j]);
#pragma unroll
    for (int j = 0; j < 2; j++) a2[j] = (o4 ? a4[2 + j] : a4[j]) + __shfl_xor(o4 ? a4[j] : a4[2 + j], 4);
    float a1 = (o8 ? a2[1] : a2[0]) + __shfl_xor(o8 ? a2[0] : a2[1], 8);
    a1 += __shfl_xor(a1, 16);
    return a1;
}

// RR_ROWS = output rows of an item: 4 on the levels with many items; 2 on the small levels (twice the items, half the serial
// chain of steps per item: those launches last as long as one wave's item).
// PLANAR: source a is the planar network input [C][h][w], C <= 16 (one chunk): eight 4-byte loads per lane and halo row instead of
// two 16-byte ones.  Its values are in the caller's units (identity transform, no LeakyReLU), so this instantiation keeps the
// wide-range arithmetic of conv3x3_f16x3: input times XSP = 2^-4, TWO accumulators, low halves scaled by 2^11, weights from the
// hi + 2^11 lo slabs -- |x| up to 1.0e6 and no absolute floor (the single-accumulator form holds |x| < 4 094 and loses bits
// below 2^-29, fine for normalised activations, not for a first-hit distance in centimetres).  One chunk: the layer is bound by
// its 158 MB, the second accumulator is free (8 waves per CU).
template <bool W16, int NWV, int PF, bool WC, int RR_ROWS = 4, bool PLANAR = false>
__global__ __launch_bounds__(NWV * 64, (PLANAR || NWV <= 8) ? 2 : 3) void conv3x3_f16x3r(const ConvArgsH g) {
    constexpr int RR_NT = NWV * 64, RR_WAVES = NWV;
    static_assert((RR_ROWS + 2) % PF == 0, "the ring slot of a halo row must be static");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WB = W16 ? WSLAB / 2 : WSLAB;                // LDS bytes of a chunk's weights
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 31, gq = lane >> 5;   // (wave: an SGPR, so is everything derived from it)
    const int nch = g.nchunks, ca16 = g.ca16, H = g.H, W = g.W, up = g.a.up;
    float* tab_a = reinterpret_cast<float*>(smem + nch * WB);
    float* tab_b = tab_a + nch * KH;
    float* bias_s = tab_b + nch * KH;                          // [32], times the operand scaling (2^4 2^7 = 2^11, or 2^-4 with PLANAR): the accumulators start from it
    long long* bnacc = reinterpret_cast<long long*>(bias_s + 32);   // [32][BN_WORDS]
    const float* zeros = reinterpret_cast<const float*>(bnacc + 32 * BN_WORDS);  // [8]: the BN coefficients of out-of-image pixels

#ifdef AIPT_R_CLOCK
    // debug builds: the shader clock this launch ran at (s_memtime follows DVFS, s_memrealtime is 100 MHz), printed by one wave
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long* const clk_w = g_rclk_w[g.ablate & 31][(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095];
    if ((threadIdx.x & 63) == 0) clk_w[0] = clk_r0;
#endif
    // ---- workgroup -> (XCD, output-channel group); XCD b & 7 owns a contiguous band of item rows
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, groups = g.groups;
    const int wpg = (int)(gridDim.x >> 3) / groups;            // workgroups per (XCD, group)
    const int gz = slot % groups, wi = slot / groups;
    if (wi >= wpg) return;
    const int n0 = gz * 32;

    const int tiles_x = g.tiles_x;
    const int per = (g.tiles_y + 7) >> 3, rb0 = xcd * per, rb1 = min(g.tiles_y, rb0 + per);
    const int nitems = max(0, rb1 - rb0) * tiles_x, stride = wpg * RR_WAVES;
    int it = wave * wpg + wi;                                  // consecutive items go to different workgroups: even load per CU
    const int sw = up ? (W >> 1) : W;
    const unsigned plane16 = (unsigned)((up ? (H >> 1) : H) * sw) * 16u;
    const int w_rd = m * 32 + (((gq ^ (m >> 3)) & 1) << 4);
    // pool: bit k set = register k's channel pools with max (gamma >= 0), else min
    unsigned posmask = 0xFFFFu;
    if (g.pool_out) {
        posmask = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int c = n0 + (k & 3) + 8 * (k >> 2) + 4 * gq;
            posmask |= (c >= g.cout || !(g.pool_gamma[c] < 0.0f)) ? (1u << k) : 0u;
        }
    }

    // ---- prefetch cursor: (item, chunk, halo row) PF rows ahead of the consumer; the first rows are requested before the prologue
    int pf_it = it, pf_c = 0, pf_y0 = 0;
    unsigned pf_v0 = 0, pf_v1 = 0;                             // this lane's byte offsets inside a chunk's row: pixel + its two channel quads
    const unsigned char* pf_base = nullptr;                    // the cursor's chunk: first of its four channel-quad planes (scalar)
    auto pf_chunk = [&]() {
        const bool fa = pf_c < ca16;
        pf_base = reinterpret_cast<const unsigned char*>(fa ? g.a.p : g.b.p) + (size_t)(fa ? pf_c : pf_c - ca16) * 4 * plane16;
    };
    auto pf_item = [&](int item) {                             // geometry of the cursor's item
        const int rb = item / tiles_x, tx = item - rb * tiles_x;
        pf_y0 = (rb0 + rb) * RR_ROWS;
        const int x = min(max(tx * RR_PX - 1 + m, 0), W - 1);
        pf_v0 = PLANAR ? (unsigned)x * 4u : (unsigned)(up ? (x >> 1) : x) * 16u + (unsigned)(2 * gq) * plane16;
        pf_v1 = pf_v0 + plane16;
    };
    f32x4 raw[PF][2] = {};
    auto issue = [&](int slot, int hp) {                       // the two channel quads of this lane's pixel in halo row hp
#ifdef AIPT_R_NOLOADS
        return;                                                // (ablation builds: results are wrong by design)
#endif
        const int y = min(max(pf_y0 - 1 + hp, 0), H - 1);
        const unsigned roff = (unsigned)((up ? (y >> 1) : y) * sw) * 16u;
        if (PLANAR) {
            // channel 8 gq + t of this lane's pixel (past the last channel: the last one again; its BN coefficients are (0, 0))
            const unsigned planeb = plane16 >> 2;                                  // bytes of one channel plane
            const unsigned char* rowp = reinterpret_cast<const unsigned char*>(g.a.p) + (roff >> 2);
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const unsigned c0 = (unsigned)min(t, g.a.C - 1) * planeb, c1 = (unsigned)min(8 + t, g.a.C - 1) * planeb;
                raw[slot][t >> 2][t & 3] = *reinterpret_cast<const float*>(rowp + (pf_v0 + (gq ? c1 : c0)));
            }
            return;
        }
        // (tensors are allocated in whole 16-channel chunks: no clamping of pad quads; scalar row base + per-lane offset)
        const unsigned char* rowp = pf_base + roff;
        raw[slot][0] = *reinterpret_cast<const f32x4*>(rowp + pf_v0);
        raw[slot][1] = *reinterpret_cast<const f32x4*>(rowp + pf_v1);
    };
    if (it < nitems) {
        pf_item(it);
        pf_chunk();
#pragma unroll
        for (int h = 0; h < PF; h++) issue(h, h);
    }

    // ---- prologue: weights of all chunks, BN coefficient table, bias -> LDS
    {
        const unsigned char* src = g.wsplit + (size_t)gz * g.wchunks * WSLAB;
        constexpr int PPC = WB / 16, WBATCH = 6;
        // [r6] six 16-byte pieces per thread in flight (as one piece per trip, hipcc waited for each load before the next: six to
        // twelve serial L2 round trips, 6-9 us of every launch before its first MFMA)
        for (int p0 = tid; p0 < nch * PPC; p0 += RR_NT * WBATCH) {
            u32x4 v[WBATCH];
#pragma unroll
            for (int j = 0; j < WBATCH; j++) {
                const int p = min(p0 + j * RR_NT, nch * PPC - 1), c = p / PPC, pp = p - c * PPC;
                v[j] = *reinterpret_cast<const u32x4*>(src + (size_t)c * WSLAB + pp * 16);
            }
#pragma unroll
            for (int j = 0; j < WBATCH; j++) {
                const int p = p0 + j * RR_NT, c = p / PPC, pp = p - c * PPC, row = pp >> 1;
                if (p < nch * PPC) *reinterpret_cast<u32x4*>(smem + c * WB + row * 32 + ((((pp & 1) ^ (row >> 3)) & 1) << 4)) = v[j];
            }
        }
        for (int kc = tid; kc < nch * KH; kc += RR_NT) {
            const bool fa = kc < ca16 * KH;
            const int c = fa ? kc : kc - ca16 * KH;
            const ConvSrc& sr = fa ? g.a : g.b;
            float2 t = make_float2(0.0f, 0.0f);
            if (c < sr.C) t = bn_ab(sr.bn, c);
            tab_a[kc] = t.x * (PLANAR ? XSP : XS1);
            tab_b[kc] = t.y * (PLANAR ? XSP : XS1);
        }
        if (tid < 32) bias_s[tid] = g.bias[n0 + tid] * (PLANAR ? XSP : XS1 * WS1);
        if (tid < 32 * BN_WORDS) bnacc[tid] = 0;
        if (tid < 8) const_cast<float*>(zeros)[tid] = 0.0f;
    }
    __syncthreads();

#ifdef AIPT_R_CLOCK
    if (lane == 0) clk_w[1] = __builtin_amdgcn_s_memrealtime();
#endif
    const int hh = H >> 1, hw = W >> 1;
    // Output through buffer descriptors: a store's address is (descriptor, per-lane byte offset, SCALAR byte offset) -- the row /
    // channel-quad part is scalar arithmetic and the lanes that must not store carry an offset past the descriptor's size (the
    // hardware drops the store): no 64-bit vector address arithmetic and no exec-mask branches per store (150 VALU instructions
    // and 50 branches per item in the pointer form).  Tensors are allocated in whole 16-channel chunks, so a group's pad quads
    // inside the allocation are stored too (exact zeros: zero weights and bias), whole quad pairs past it skipped by a scalar test.
    const unsigned aq = (unsigned)((g.d2s ? 4 : g.cout) + 15) / 16u * 4u;                // allocated channel quads of the output
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)(g.d2s ? 4u * (unsigned)(H * W) * 16u : aq * (unsigned)(H * W) * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t pool_rs = __builtin_amdgcn_make_buffer_rsrc(g.pool_out ? g.pool_out : g.out, 0, (int)(aq * (unsigned)(hh * hw) * 16u), 0x00020000);
    constexpr unsigned NO_STORE = 0x80000000u;                                           // + any scalar offset: past every descriptor (< 2 GiB, host-checked)
#ifdef AIPT_R_ABL
    const bool abl_nostore = (AIPT_R_ABL & 16) != 0;           // (every store gets the out-of-range offset: the instruction stream stays, the traffic goes)
#else
    constexpr bool abl_nostore = false;
#endif
    const int q0 = n0 >> 2;
    typedef unsigned u4s __attribute__((ext_vector_type(4)));
    f32x16 acc[RR_ROWS], acc1[PLANAR ? RR_ROWS : 1];
    auto acc_init = [&]() {
        if (PLANAR) {
#pragma unroll
            for (int r = 0; r < RR_ROWS; r++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc1[r][k] = 0.f;
        }
        {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_s + 8 * j + 4 * gq);
#pragma unroll
                for (int r = 0; r < RR_ROWS; r++) { acc[r][4 * j] = bq[0]; acc[r][4 * j + 1] = bq[1]; acc[r][4 * j + 2] = bq[2]; acc[r][4 * j + 3] = bq[3]; }
            }
        }
    };
    auto epilogue = [&](const int y0, const int x) {
        // ---- epilogue.  D (32 x 32): register k of lane l = channel (k & 3) + 8 (k >> 2) + 4 (l >> 5), pixel l & 31: registers
        // 4 j .. 4 j + 3 are channel quad 2 j + (l >> 5) of this lane's pixel
        const bool lane_ok = m >= 1 && m <= RR_PX && x < W;
        float s1[16], s2[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
        for (int r = 0; r < RR_ROWS; r++) {
            if (PLANAR) acc[r] = (acc[r] + acc1[PLANAR ? r : 0] * (1.0f / LO_SCALE)) * (1.0f / XSP);
            else acc[r] = acc[r] * (1.0f / (XS1 * WS1));
            if (g.out_lrelu) {
#pragma unroll
                for (int k = 0; k < 16; k++) acc[r][k] = fmaxf(acc[r][k], acc[r][k] * SLOPE);
            }
            const int y = y0 + r;
            if (y < H) {                                       // wave-uniform
#pragma unroll
                for (int k = 0; k < 16; k++) { s1[k] += acc[r][k]; s2[k] = fmaf(acc[r][k], acc[r][k], s2[k]); }
                if (g.d2s) {
                    // upsample + conv as a half-resolution conv (see conv3x3_f16x3): virtual channel 4 p + c is channel c of the child
                    // (2 y + (p >> 1), 2 x + (p & 1)) of this lane's pixel; register quad j of half gq is parity p = 2 j + gq, its fourth
                    // value an exact zero (zero weights and bias): one C4 store per parity.  Lane part: child column 2 x + gq;
                    // scalar part: child row 2 y + j
                    const unsigned voff = lane_ok && !abl_nostore ? (unsigned)(2 * x + gq) * 16u : NO_STORE;
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4s, f32x4{acc[r][4 * j], acc[r][4 * j + 1], acc[r][4 * j + 2], acc[r][4 * j + 3]}),
                                                               out_rs, voff, (unsigned)((2 * y + j) * (2 * W)) * 16u, 0);
                } else {
                    const unsigned voff = lane_ok && !abl_nostore ? (unsigned)x * 16u + (gq ? (unsigned)(H * W) * 16u : 0u) : NO_STORE;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if ((unsigned)(q0 + 2 * j) >= aq) continue;      // (scalar) the whole quad pair lies past the allocation
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4s, f32x4{acc[r][4 * j], acc[r][4 * j + 1], acc[r][4 * j + 2], acc[r][4 * j + 3]}),
                                                               out_rs, voff, (unsigned)((q0 + 2 * j) * H + y) * (unsigned)W * 16u, 0);
                    }
                }
            }
        }
        if (g.pool_out) {
            // 2x2 pool of the raw output (see conv3x3_f16x3): rows 2 rp, 2 rp + 1 are registers of this lane, columns x (even: odd
            // m) and x + 1 are this lane and the next
#pragma unroll
            for (int rp = 0; rp < RR_ROWS / 2; rp++) {
                const int y = y0 + 2 * rp;
                // max where gamma >= 0, min where gamma < 0, as ONE instruction per pair: med3(a, b, +inf) = max(a, b),
                // med3(a, b, -inf) = min(a, b) (selecting between a computed max and a computed min cost three)
                float pv[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float lim = ((posmask >> k) & 1u) ? INFINITY : -INFINITY;
                    const float v = __builtin_amdgcn_fmed3f(acc[2 * rp][k], acc[2 * rp + 1][k], lim);
                    const float o = __builtin_bit_cast(float, dpp_wave_shl1(__builtin_bit_cast(unsigned, v)));
                    pv[k] = __builtin_amdgcn_fmed3f(v, o, lim);
                }
                if (y < H) {
                    const unsigned voff = ((m & 1) && m < RR_PX && x < W && !abl_nostore) ? (unsigned)(x >> 1) * 16u + (gq ? (unsigned)(hh * hw) * 16u : 0u) : NO_STORE;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if ((unsigned)(q0 + 2 * j) >= aq) continue;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4s, f32x4{pv[4 * j], pv[4 * j + 1], pv[4 * j + 2], pv[4 * j + 3]}),
                                                               pool_rs, voff, (unsigned)((q0 + 2 * j) * hh + (y >> 1)) * (unsigned)hw * 16u, 0);
                    }
                }
            }
        }
        if (g.stat) {
#pragma unroll
            for (int k = 0; k < 16; k++) { s1[k] = lane_ok ? s1[k] : 0.f; s2[k] = lane_ok ? s2[k] : 0.f; }
            // [r5] committed item by item to the workgroup's fixed-point accumulators (integer adds): the sums do not depend on
            // which wave of which workgroup ran which item -- the same bits on any number of CUs (aipt_frame_prefetch runs this
            // kernel on a CU-masked stream with fewer workgroups); measured free against one fp32 running sum per wave
            const float h1 = halfwave_sum16(s1, m), h2 = halfwave_sum16(s2, m);
            if (m < 16) {                                      // lanes m < 16 of both halves hold channel (k & 3) + 8 (k >> 2) + 4 gq, k as in halfwave_sum16
                const int k = 8 * (m & 1) + 4 * ((m >> 1) & 1) + 2 * ((m >> 2) & 1) + ((m >> 3) & 1);
                int cl = (k & 3) + 8 * (k >> 2) + 4 * gq;
                if (g.d2s) cl = cl < 16 ? (cl & 3) : 31;       // a channel's four children share its sums (slot 31: the unused virtual channels, all zero)
                const BnFix fx = bn_fix((double)h1), fq = bn_fix((double)h2);
                unsigned long long* acc4 = reinterpret_cast<unsigned long long*>(bnacc + cl * BN_WORDS);
                atomicAdd(acc4, (unsigned long long)fx.i); atomicAdd(acc4 + 1, (unsigned long long)fx.f);
                atomicAdd(acc4 + 2, (unsigned long long)fq.i); atomicAdd(acc4 + 3, (unsigned long long)fq.f);
            }
        }
    };
    for (; it < nitems; it += stride) {
        const int rb = it / tiles_x, tx = it - rb * tiles_x;
        const int y0 = (rb0 + rb) * RR_ROWS, X = tx * RR_PX - 1;
        const int x = X + m;
        const bool xin = x >= 0 && x < W;
        acc_init();
        for (int c = 0; c < nch; c++) {
            const float slope = c < ca16 ? g.a.slope : g.b.slope;
#ifdef AIPT_R_ABL
            const f16x8 abl_w = *reinterpret_cast<const f16x8*>(smem + c * WB + w_rd);
#endif
            // the chunk's 18 weight fragments and this lane's 16 BN coefficients stay in registers for its six halo rows
            const unsigned char* wl = smem + c * WB + w_rd;
            // (WC; without it they are re-read from LDS where they are used: fewer registers, three waves per SIMD)
            f16x8 wfh[WC ? 9 : 1], wfl[WC && !W16 ? 9 : 1];
            const float* ta = tab_a + c * KH + gq * 8;
            const float* tb = tab_b + c * KH + gq * 8;
            f32x4 a0, a1, b0, b1;
            if (WC) {
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    wfh[t] = *reinterpret_cast<const f16x8*>(wl + t * 1024);
                    if (!W16) wfl[t] = *reinterpret_cast<const f16x8*>(wl + (9 + t) * 1024);
                }
                a0 = *reinterpret_cast<const f32x4*>(ta); a1 = *reinterpret_cast<const f32x4*>(ta + 4);
                b0 = *reinterpret_cast<const f32x4*>(tb); b1 = *reinterpret_cast<const f32x4*>(tb + 4);
            }
#pragma unroll
            for (int h = 0; h < RR_ROWS + 2; h++) {
                // ---- transform + split this lane's 8 channels of halo row h.  (Nothing of this row may be scheduled above this
                // point: hipcc otherwise hoists the rows' first FMAs -- and with them the wait for loads issued one row ago.)
                __builtin_amdgcn_sched_barrier(0);
                // zero padding in the normalised domain: an out-of-image pixel takes its coefficients from the zero block (WC: masked below)
                const bool ok = xin && (unsigned)(y0 - 1 + h) < (unsigned)H;
                if (!WC) {
                    const float* tal = ok ? ta : zeros;
                    const float* tbl = ok ? tb : zeros;
                    a0 = *reinterpret_cast<const f32x4*>(tal); a1 = *reinterpret_cast<const f32x4*>(tal + 4);
                    b0 = *reinterpret_cast<const f32x4*>(tbl); b1 = *reinterpret_cast<const f32x4*>(tbl + 4);
                }
                unsigned xh[4], xl[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const f32x4& rw = raw[h % PF][p >> 1];
                    const f32x4& aa = (p >> 1) ? a1 : a0;
                    const f32x4& bb = (p >> 1) ? b1 : b0;
                    const int e = (p & 1) * 2;
                    float v0 = fmaf(aa[e], rw[e], bb[e]), v1 = fmaf(aa[e + 1], rw[e + 1], bb[e + 1]);
                    if (!PLANAR) { v0 = fmaxf(v0, v0 * slope); v1 = fmaxf(v1, v1 * slope); }   // (the network input has no LeakyReLU: the host checks)
                    else {
                        // the caller's units: |x| 2^-4 beyond fp16's 65 504 saturates HERE, so that the low half stays finite
                        // ((v - hi) 2^11 of an unclamped v is inf, and inf x 0 of a pad weight is NaN for the whole frame)
                        v0 = __builtin_amdgcn_fmed3f(v0, -65504.0f, 65504.0f); v1 = __builtin_amdgcn_fmed3f(v1, -65504.0f, 65504.0f);
                    }
                    if (PLANAR) {
                        // hi rounded to NEAREST (v_cvt_pk_f16_f32): |v - hi| <= ulp / 2 <= 16, so the low half (v - hi) 2^11 stays
                        // <= 32 768.  With the round-toward-zero pack the remainder reaches a whole ulp -- 32 above 32 768, i.e. for
                        // |x| >= 2^19 -- and (v - hi) 2^11 >= 65 520 rounds to fp16 inf: one such pixel and the frame is NaN
                        // (found in round 5 by feeding |x| up to 1.0e6, the range include/aiptd.h promises)
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
                        xh[p] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0, v1}, f16x2v));
                        xl[p] = split_lo_mix_scaled(xh[p], v0 * LO_SCALE, v1 * LO_SCALE, -LO_SCALE);
                    } else {
                        xh[p] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
                        xl[p] = split_lo_mix(xh[p], v0, v1);
                    }
                    if (WC) { xh[p] = ok ? xh[p] : 0u; xl[p] = ok ? xl[p] : 0u; }
#ifdef AIPT_R_ABL
                    if (AIPT_R_ABL & 8) { xh[p] = __builtin_bit_cast(unsigned, rw[e]); xl[p] = __builtin_bit_cast(unsigned, rw[e + 1]); }
#endif
                }
                // ---- the ring slot is free: fetch PF rows ahead (into the next chunk / the next item when that wraps)
                if ((h + PF) % (RR_ROWS + 2) == 0) {
                    if (++pf_c == nch) {
                        pf_c = 0;
                        if (pf_it + stride < nitems) { pf_it += stride; pf_item(pf_it); }   // (past the end: harmless re-reads)
                    }
                    pf_chunk();
                }
                issue(h % PF, (h + PF) % (RR_ROWS + 2));
                // ---- MFMAs of every (output row, tap row) pair this halo row serves
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    unsigned sh[4], sl[4];
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        sh[p] = kx == 1 ? xh[p] : kx == 0 ? dpp_wave_shr1(xh[p]) : dpp_wave_shl1(xh[p]);
                        sl[p] = kx == 1 ? xl[p] : kx == 0 ? dpp_wave_shr1(xl[p]) : dpp_wave_shl1(xl[p]);
                    }
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    const f16x8 fxh = __builtin_bit_cast(f16x8, (u4){sh[0], sh[1], sh[2], sh[3]});
                    const f16x8 fxl = __builtin_bit_cast(f16x8, (u4){sl[0], sl[1], sl[2], sl[3]});
#pragma unroll
                    for (int ky = 0; ky < 3; ky++) {
                        const int r = h - ky;
                        if (r < 0 || r >= RR_ROWS) continue;
#ifdef AIPT_R_ABL   // ablation builds (results wrong by design): 1 no weight-fragment reads, 2 no epilogue, 4 no MFMAs, 8 no transform
                        const f16x8 fwh = (AIPT_R_ABL & 1) ? abl_w : WC ? wfh[ky * 3 + kx] : *reinterpret_cast<const f16x8*>(wl + (ky * 3 + kx) * 1024);
#else
                        const f16x8 fwh = WC ? wfh[ky * 3 + kx] : *reinterpret_cast<const f16x8*>(wl + (ky * 3 + kx) * 1024);
#endif
                        f32x16& lo_acc = PLANAR ? acc1[PLANAR ? r : 0] : acc[r];
#ifdef AIPT_R_ABL
                        if (AIPT_R_ABL & 4) { acc[r][(ky * 3 + kx) & 15] += __builtin_bit_cast(float, sh[0] ^ sl[1]) + (float)fwh[1]; continue; }
#endif
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh, fxh, acc[r], 0, 0, 0);
                        if (!W16) {
#ifdef AIPT_R_ABL
                            const f16x8 fwl = (AIPT_R_ABL & 1) ? abl_w : WC ? wfl[ky * 3 + kx] : *reinterpret_cast<const f16x8*>(wl + (9 + ky * 3 + kx) * 1024);
#else
                            const f16x8 fwl = WC ? wfl[ky * 3 + kx] : *reinterpret_cast<const f16x8*>(wl + (9 + ky * 3 + kx) * 1024);
#endif
                            lo_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl, fxh, lo_acc, 0, 0, 0);
                        }
                        lo_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh, fxl, lo_acc, 0, 0, 0);
                    }
                }
            }
        }
#ifdef AIPT_R_ABL
        if (AIPT_R_ABL & 2) {                                  // one store so that the accumulators stay live
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RR_ROWS; r++)
#pragma unroll
                for (int k = 0; k < 16; k++) t += acc[r][k];
            if (t == 123.456f) g.out[lane] = t;
            continue;
        }
#endif
        epilogue(y0, x);
    }

#ifdef AIPT_R_CLOCK
    if (lane == 0) {
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        clk_w[2] = r1;
        if (blockIdx.x == 8 && tid == 0) { g_rclk[g.ablate & 31][6] = __builtin_amdgcn_s_memtime() - clk_t0; g_rclk[g.ablate & 31][7] = r1 - clk_r0; }
    }
#endif
    // ---- BN sums of the workgroup -> the layer's table
    if (g.stat) {
        __syncthreads();
        if (tid < 32 && n0 + tid < (g.d2s ? g.d2s : g.cout)) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(g.stat + ((size_t)(blockIdx.x % NSLOT) * g.sc + n0 + tid) * BN_WORDS);
#pragma unroll
            for (int k = 0; k < BN_WORDS; k++) atomicAdd(dst + k, (unsigned long long)bnacc[tid * BN_WORDS + k]);
        }
    }
}

// -------------------------------------------------------------------------------------------------- single-quad conv
// dec1.c2 (3 -> 3) + the network's last BatchNorm / LeakyReLU + crop: one C4 quad in, three planes out, no concat, no resampling.
// 76 MFLOP on 15 MB, so the layer is two streaming passes instead of conv -> raw tensor -> normalisation pass:
//   STATS = true   the conv's per-channel sums only (batch statistics; nothing is stored)
//   STATS = false  the conv again, normalised with those sums (or the running statistics) and written as the planar, cropped
//                  network output: 15 + 15 MB read and 11 MB written instead of 15 r + 15 w, 15 r + 11 w and a third launch.
// Both passes run the same instruction sequence on the same inputs: the values pass two normalises are the values pass one summed.
// Mapping: lane = pixel.  A wave slides down a strip of 64 columns x QROWS rows: ONE coalesced 16-byte load per lane and input row
// (the next row is in flight while this one is multiplied), the left / right neighbours are the same registers shifted by one
// lane (DPP), so a wave produces the 62 inner columns.  (Round 3's form -- a thread per 4-pixel strip with its 3 x 6 neighbourhood
// from eighteen 16-byte loads at a 64-byte lane stride -- was bound by L1 accesses: 18 us per pass for 15 MB.)
constexpr int QROWS = 8, QCOLS = 62;
__device__ __forceinline__ float dpp_shr1f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_shl1f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true)); }
template <int CIN, int COUT, bool STATS>
__global__ __launch_bounds__(256) void conv3x3_quad(const ConvArgs g) {
    __shared__ float2 red[4][COUT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = g.H, W = g.W;
    const int nsx = (W + QCOLS - 1) / QCOLS;
    const int strip = blockIdx.x * 4 + wave;                      // wave-uniform
    const int sy = strip / nsx, sx = strip - sy * nsx;
    const int y0 = sy * QROWS, x = sx * QCOLS - 1 + lane;
    const bool xin = x >= 0 && x < W;
    const float4* col = reinterpret_cast<const float4*>(g.a.p) + min(max(x, 0), W - 1);
    const bool live = y0 < H;                                      // wave-uniform (strips past the end: the last block's spare waves)
    // all QROWS + 2 input rows of the strip are requested before anything else -- before the BatchNorm coefficients, whose own
    // chain (statistics -> fp64 -> rsqrt) is a memory round trip: with one row in flight per wave a pass was a chain of
    // QROWS + 1 dependent round trips (14 us for 15 MB)
    float4 raw[QROWS + 2];
    if (live) {
#pragma unroll
        for (int r = 0; r < QROWS + 2; r++) raw[r] = col[(size_t)min(max(y0 - 1 + r, 0), H - 1) * W];
    }
    float ca[CIN], cb[CIN];
#pragma unroll
    for (int c = 0; c < CIN; c++) {
        const float2 ab = bn_ab(g.a.bn, c);                       // wave-uniform: scalar loads
        ca[c] = ab.x; cb[c] = ab.y;
    }
    const float slope = g.a.slope;
    float2 oab[COUT];
    if (!STATS) {
#pragma unroll
        for (int j = 0; j < COUT; j++) oab[j] = bn_ab(g.out_bn, j);
    }
    // win[ky][kx][c]: the normalised inputs of the rows y-1, y, y+1 at columns x-1, x, x+1 (zero padding in the normalised domain)
    float win[3][3][CIN];
    auto put_row = [&](int slot, int yy, float4 raw) {
        const bool in = xin && yy >= 0 && yy < H;
        const float rr[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int c = 0; c < CIN; c++) {
            const float t = fmaf(ca[c], rr[c], cb[c]);
            const float v = in ? fmaxf(t, t * slope) : 0.0f;
            win[slot][1][c] = v;
            win[slot][0][c] = dpp_shr1f(v);                        // column x - 1 (lane 0: unused)
            win[slot][2][c] = dpp_shl1f(v);                        // column x + 1 (lane 63: unused)
        }
    };
    float s1[COUT], s2[COUT];
#pragma unroll
    for (int j = 0; j < COUT; j++) { s1[j] = 0.f; s2[j] = 0.f; }
    if (live) {
        put_row(0, y0 - 1, raw[0]);
        put_row(1, y0, raw[1]);
        const bool col_ok = lane >= 1 && lane <= QCOLS && x < W;
#pragma unroll
        for (int r = 0; r < QROWS; r++) {
            const int y = y0 + r;
            // window slots rotate with r: row y-1 in slot r % 3, y in (r + 1) % 3, y + 1 in (r + 2) % 3
            put_row((r + 2) % 3, y + 1, raw[r + 2]);
            float acc[COUT];
#pragma unroll
            for (int j = 0; j < COUT; j++) acc[j] = g.bias[j];
#pragma unroll
            for (int j = 0; j < COUT; j++)
#pragma unroll
                for (int c = 0; c < CIN; c++)
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++)
                            acc[j] = fmaf(win[(r + ky) % 3][kx][c], g.w_raw[((size_t)j * CIN + c) * 9 + ky * 3 + kx], acc[j]);   // scalar loads
            if (g.out_lrelu) {
#pragma unroll
                for (int j = 0; j < COUT; j++) acc[j] = lrelu(acc[j], SLOPE);
            }
            const bool ok = col_ok && y < H;
            if (STATS) {
#pragma unroll
                for (int j = 0; j < COUT; j++) { const float t = ok ? acc[j] : 0.0f; s1[j] += t; s2[j] = fmaf(t, t, s2[j]); }
            } else if (ok && y < g.oh && x < g.ow) {
#pragma unroll
                for (int j = 0; j < COUT; j++)
                    g.out_planar[((size_t)j * g.oh + y) * g.ow + x] = lrelu(fmaf(oab[j].x, acc[j], oab[j].y), g.out_slope);
            }
        }
    }
    if (STATS) {
#pragma unroll
        for (int j = 0; j < COUT; j++) {
            float a = s1[j], b = s2[j];
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            if (lane == 0) red[wave][j] = make_float2(a, b);
        }
        __syncthreads();
        if (tid < COUT) {
            float2 t = red[0][tid];
            for (int w = 1; w < 4; w++) { t.x += red[w][tid].x; t.y += red[w][tid].y; }
            bn_accumulate(g.stat, g.sc, tid, t.x, t.y);
        }
    }
}

// -------------------------------------------------------------------------------------------------- VALU conv
// One thread per output element; reads its 9*cin taps straight from HBM.  Only for on-GPU cross-checks.
__global__ __launch_bounds__(64) void conv3x3_valu(const ConvArgs g) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y, j = blockIdx.z;
    if (x >= g.W) return;
    float acc = g.bias[j];
    const int ctot = g.a.C + g.b.C;
    for (int c = 0; c < ctot; c++) {
        const float* wk = g.w_raw + ((size_t)j * g.cin + c) * 9;
        for (int ky = 0; ky < 3; ky++)
            for (int kx = 0; kx < 3; kx++) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) continue;
                const float v = c < g.a.C ? load_src(g.a, c, yy, xx, g.H, g.W) : load_src(g.b, c - g.a.C, yy, xx, g.H, g.W);
                acc = fmaf(v, wk[ky * 3 + kx], acc);
            }
    }
    if (g.out_lrelu) acc = lrelu(acc, SLOPE);
    g.out[(((size_t)(j >> 2) * g.H + y) * g.W + x) * 4 + (j & 3)] = acc;
}

// per-channel sum / sum-of-squares of a stored tensor (VALU path only): one block per channel
__global__ __launch_bounds__(256) void channel_stats(const float* t, size_t hw, long long* stat) {
    const int c = blockIdx.x;
    const float* p = t + (size_t)(c >> 2) * hw * 4 + (c & 3);      // C4 layout
    double a = 0, b = 0;
    for (size_t i = threadIdx.x; i < hw; i += 256) { const double v = p[i * 4]; a += v; b += v * v; }
    __shared__ double sa[256], sb[256];
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {                                    // slot 0 of a zeroed table, in the table's fixed-point format
        const BnFix fx = bn_fix(sa[0]), fq = bn_fix(sb[0]);
        long long* d = stat + (size_t)blockIdx.x * BN_WORDS;
        d[0] = fx.i; d[1] = fx.f; d[2] = fq.i; d[3] = fq.f;
    }
}

// -------------------------------------------------------------------------------------------------- elementwise
// out = MaxPool2d(2) of the normalised tensor lrelu(a*raw+b); C4 in, C4 out: blockIdx.y = channel quad (its four (a, b)
// are wave-uniform: scalar loads), one thread per pooled pixel
__global__ __launch_bounds__(256) void pool2_norm(const float* raw, const BnRef bn, float slope, int C, int H, int W,
                                                  float* out) {
    const int h = H >> 1, w = W >> 1, q = blockIdx.y;
    const size_t n = (size_t)h * w;
    const float4* in4 = reinterpret_cast<const float4*>(raw) + (size_t)q * H * W;
    float4* out4 = reinterpret_cast<float4*>(out) + (size_t)q * n;
    float2 f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) f[k] = q * 4 + k < C ? bn_ab(bn, q * 4 + k) : make_float2(0.0f, 0.0f);   // pad channels stay 0
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)(i / w);
        const float4* p = in4 + (size_t)(2 * y) * W + 2 * x;
        const float4 r[4] = {p[0], p[1], p[W], p[W + 1]};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float e0 = k == 0 ? r[0].x : k == 1 ? r[0].y : k == 2 ? r[0].z : r[0].w;
            const float e1 = k == 0 ? r[1].x : k == 1 ? r[1].y : k == 2 ? r[1].z : r[1].w;
            const float e2 = k == 0 ? r[2].x : k == 1 ? r[2].y : k == 2 ? r[2].z : r[2].w;
            const float e3 = k == 0 ? r[3].x : k == 1 ? r[3].y : k == 2 ? r[3].z : r[3].w;
            o[k] = fmaxf(fmaxf(lrelu(fmaf(f[k].x, e0, f[k].y), slope), lrelu(fmaf(f[k].x, e1, f[k].y), slope)),
                         fmaxf(lrelu(fmaf(f[k].x, e2, f[k].y), slope), lrelu(fmaf(f[k].x, e3, f[k].y), slope)));
        }
        out4[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// planar out[c][y][x] = lrelu(a*raw+b) of a C4 tensor [.][H][W][4], cropped to oh x ow (network output: the frame's
// padding is dropped here; hidden-state export: oh = H, ow = W).  blockIdx.y = channel quad: one 16-byte load per pixel,
// up to four coalesced plane stores.
__global__ __launch_bounds__(256) void apply_norm(const float* raw, const BnRef bn, float slope, int C, int H, int W,
                                                  float* out, int oh, int ow) {
    const int q = blockIdx.y;
    float2 f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) f[k] = q * 4 + k < C ? bn_ab(bn, q * 4 + k) : make_float2(0.0f, 0.0f);
    const float4* src = reinterpret_cast<const float4*>(raw) + (size_t)q * H * W;
    const size_t n = (size_t)oh * ow;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / ow), x = (int)(i - (size_t)y * ow);
        const float4 r = src[(size_t)y * W + x];
        const float e[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (q * 4 + k < C) out[(size_t)(q * 4 + k) * n + i] = lrelu(fmaf(f[k].x, e[k], f[k].y), slope);
    }
}

// planar [C][hw] -> C4 (hidden-state import); pad channels are written as 0
__global__ __launch_bounds__(256) void planar_to_c4(const float* in, int C, size_t hw, float* out) {
    const size_t n = (size_t)pad4(C) * hw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i & 3);
        const size_t t = i >> 2;
        const size_t px = t % hw;
        const int c = (int)(t / hw) * 4 + k;
        out[i] = c < C ? in[(size_t)c * hw + px] : 0.0f;
    }
}

// -------------------------------------------------------------------------------------------------- host side
struct LayerW {
    int cin = 0, cout = 0, NB = 0, NP = 0, nchunks = 0;
    int ca = 0, pcin = 0;   // channels of the first concat source; padded-concat channel count (pad4(ca) + pad4(cin - ca))
    float *d_w = nullptr, *d_w_raw = nullptr, *d_bias = nullptr, *d_gamma = nullptr, *d_beta = nullptr;
    float* d_w_raw16 = nullptr;              // the weights rounded to fp16 (AIPT_DN_IMPL_MFMA_F16W), for the direct-conv kernels
    unsigned char* d_wsplit_d2s16 = nullptr; // depth-to-space weights built from the fp16-rounded taps
    float2* d_ab_running = nullptr;
    float *d_w_d2s = nullptr, *d_bias_d2s = nullptr;   // upsample+conv as a half-resolution conv with 4*cout virtual channels
    // split-fp16 copy of the weights for conv3x3_f16x3: [coutp32/32][nchunks16][hi | lo*2^11][9][32][16] halfs over the
    // K16-aligned concat space (source a in chunks [0, ca16), source b after it)
    int coutp32 = 0, nchunks16 = 0, ca16 = 0;
    unsigned char* d_wsplit = nullptr;
    unsigned char* d_wsplit1 = nullptr;      // the same tiling in the single-accumulator scaling of conv3x3_f16x3r: hi = fp16(2^7 w), lo unscaled
    unsigned char* d_wsplit1_16 = nullptr;   // ... of the fp16-ROUNDED weights (AIPT_DN_IMPL_MFMA_F16W): hi = 2^7 fp16(w) exactly, lo = 0
    float* d_bias32 = nullptr;
    unsigned char* d_wsplit_d2s = nullptr;   // the same layout for the depth-to-space form of dec1.c1 (12 virtual outputs in one group)
    float* d_bias32_d2s = nullptr;
    // ... and for conv3x3_f16x3r: single-accumulator scaling, virtual channel 4 * parity + j (a parity's three channels + a zero one
    // = one register quad of a lane = one C4 store); [0]: fp32 taps, [1]: taps rounded to fp16 first (AIPT_DN_IMPL_MFMA_F16W)
    unsigned char* d_wsplit1_d2s[2] = {nullptr, nullptr};
    float* d_bias32_d2s4 = nullptr;
};

struct Tensor {
    float* p = nullptr;     // C4 layout unless planar
    BnRef bn = {nullptr, nullptr, nullptr, nullptr, 0, 0.0};   // identity until a conv produces the tensor
    int C = 0;
    float slope = SLOPE;
    int planar = 0;
};

struct DenoiseState {
    bool have_weights = false;
    LayerW L[NLAYERS];
    int H = 0, W = 0;
    // NSET activation sets, rotating by frame: frame n works in A[n % NSET] and reads the hidden states frame n-1 left in
    // the set before it.  That is what lets aipt_frames run consecutive frames of ONE recurrent sequence on NSET streams.
    struct ActSet {
        Tensor In;                     // C4 copy of the planar network input
        Tensor T1[6], T2[6], Hid[6];   // per level 0..5 (5 = bottleneck)
        Tensor P[5];                   // pooled, normalised encoder outputs (identity transform)
        Tensor D1[6], D2[6];           // decoder k = 1..5
    } A[AIPT_DN_PIPE];
    static constexpr int NSET = AIPT_DN_PIPE;
    int aset = 0;                      // set of the last frame (its Hid[] are the carried hidden states)
    // BN sums of every conv layer, one set per frame in a ring of 2 x NSET: the hidden states written in frame n are read in
    // frame n+1, whose own sums go to the next set (one memset per frame instead of a finalize launch per conv); the set
    // zeroed for frame n belonged to frame n - 2 NSET (same stream: retired) and was last read, as hidden-state statistics,
    // by frame n - 2 NSET + 1, whose bottleneck the frames in between have waited for.
    static constexpr int STAT_SC = 128;                                    // channel stride (>= every cout)
    static constexpr size_t STAT_LAYER = (size_t)NSLOT * STAT_SC * BN_WORDS;   // 64-bit words per layer
    static constexpr int STAT_SETS = 2 * AIPT_DN_PIPE;
    long long* stat[STAT_SETS] = {};
    int sset = 0;
    // Frame pipelining (aipt_frames): frame n+1 may enter encoder level L as soon as frame n has left it (its hidden state of
    // that level is complete), so the launches of NSET frames interleave on NSET streams and the many small launches of the
    // deep levels and the decoder of one frame (a few dozen workgroups on 256 CUs) run beside the full-size layers of another.
    hipStream_t cur = nullptr;         // stream of the forward pass being enqueued
    hipEvent_t ev_level[AIPT_DN_PIPE][6] = {};    // [set][level]: the hidden state of that level is written
    hipEvent_t ev_fence[AIPT_DN_PIPE] = {};       // around a timed forward pass: [k < NSET-1] drain of the other streams, [NSET-1] its end
    bool ev_level_valid[AIPT_DN_PIPE][6] = {};
    bool hidden_valid = false;
    int impl = AIPT_DN_IMPL_MFMA_F16X3;
    int num_cus = 256;
    int run_cus = 256;                 // CUs the forward pass being enqueued may use (fewer than num_cus on aipt_frame's CU-masked denoiser stream)
    // kernel selection of the split-fp16 implementations (aipt_denoise_set_option; level sizes in pixels)
    long long opt_r_minpix = 200000;   // >= : conv3x3_f16x3r (register-staged, persistent)
    long long opt_f16_minpix = 14000;  // >= : conv3x3_f16x3 with 8-row tiles, below: 4-row tiles
    long long opt_small_minpix = 0;    // <  : the f32-MFMA kernels
    int opt_fused_pool = 1;            // 2x2 pool of an encoder block's output in the conv's epilogue (0: pool2_norm launches)
    int opt_ky_split = 1;              // 4-row tiles of conv3x3_f16x3 with three waves per row (one per tap row)
    // largest |gamma| / |beta| of the loaded BatchNorms: with batch statistics |BN(x)| <= |gamma| sqrt(pixels) + |beta|, which
    // decides whether a level may run on kernels that hold normalised activations in fp16 pairs (run_conv)
    float bn_gmax = 0.0f, bn_bmax = 0.0f;
    std::vector<void*> allocs;
    // per-layer HIP-event profiling (aipt_denoise_profile_*)
    uint32_t prof_mask = 0;
    int prof_max = 0, prof_calls = 0;
    int prof_every = 1, prof_seen = 0;   // sample every prof_every-th forward (aipt_denoise_profile_stride)
    std::vector<hipEvent_t> prof_ev;     // [call][layer][2]
    char kname[NLAYERS][56] = {};        // kernel that ran each layer in the last forward
};

static inline bool w16_mode(const DenoiseState* s) { return s->impl == AIPT_DN_IMPL_MFMA_F16W; }
static inline bool impl_is_f16(int impl) { return impl == AIPT_DN_IMPL_MFMA_F16X3 || impl == AIPT_DN_IMPL_MFMA_F16W; }

static void build_table(int* cin, int* cout) {
    int n = 0, c_in = 10;
    for (int i = 0; i < 5; i++) {
        const int c = ENC_CH[i];
        cin[n] = c_in;  cout[n++] = c;
        cin[n] = 2 * c; cout[n++] = c;
        cin[n] = c;     cout[n++] = c;
        c_in = c;
    }
    cin[n] = 101; cout[n++] = 101;
    cin[n] = 202; cout[n++] = 101;
    cin[n] = 101; cout[n++] = 101;
    int prev = 101;
    for (int k = 5; k >= 1; k--) {
        cin[n] = prev + ENC_CH[k - 1]; cout[n++] = DEC_CH[k];
        cin[n] = DEC_CH[k];            cout[n++] = DEC_CH[k];
        prev = DEC_CH[k];
    }
}

static void free_weights(DenoiseState* s) {
    for (auto& l : s->L) {
        hipFree(l.d_w_raw16); hipFree(l.d_wsplit_d2s16);
        hipFree(l.d_w); hipFree(l.d_w_raw); hipFree(l.d_bias); hipFree(l.d_gamma); hipFree(l.d_beta);
        hipFree(l.d_ab_running); hipFree(l.d_wsplit); hipFree(l.d_wsplit1); hipFree(l.d_wsplit1_16); hipFree(l.d_bias32);
        hipFree(l.d_w_d2s); hipFree(l.d_bias_d2s); hipFree(l.d_wsplit_d2s); hipFree(l.d_bias32_d2s);
        hipFree(l.d_wsplit1_d2s[0]); hipFree(l.d_wsplit1_d2s[1]); hipFree(l.d_bias32_d2s4);
        l = LayerW();
    }
    s->have_weights = false;
}

static void free_profile(DenoiseState* s) {
    for (hipEvent_t e : s->prof_ev) if (e) hipEventDestroy(e);
    s->prof_ev.clear();
    s->prof_mask = 0; s->prof_max = 0; s->prof_calls = 0;
}

static void free_activations(DenoiseState* s) {
    for (void* p : s->allocs) hipFree(p);
    s->allocs.clear();
    for (int k = 0; k < DenoiseState::STAT_SETS; k++) s->stat[k] = nullptr;
    for (int a = 0; a < DenoiseState::NSET; a++)
        for (int l = 0; l < 6; l++) {
            if (s->ev_level[a][l]) hipEventDestroy(s->ev_level[a][l]);
            s->ev_level[a][l] = nullptr;
            s->ev_level_valid[a][l] = false;
        }
    for (int k = 0; k < DenoiseState::NSET; k++) {
        if (s->ev_fence[k]) hipEventDestroy(s->ev_fence[k]);
        s->ev_fence[k] = nullptr;
    }
    s->H = s->W = 0;
}

void denoise_destroy(aipt_ctx* ctx) {
    if (!ctx->dn) return;
    free_weights(ctx->dn);
    free_activations(ctx->dn);
    free_profile(ctx->dn);
    delete ctx->dn;
    ctx->dn = nullptr;
}

static DenoiseState* state(aipt_ctx* ctx) {
    if (!ctx->dn) {
        ctx->dn = new DenoiseState();
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->dn->num_cus = prop.multiProcessorCount;
#if defined(AIPT_DEBUG_HOOKS) || defined(AIPT_CONV_ABLATE)
        // debug / ablation builds only (tools/conv_ablate.sh): initial values of the selection options from the environment
        if (getenv("AIPT_F16R_MINPIX")) ctx->dn->opt_r_minpix = atoll(getenv("AIPT_F16R_MINPIX"));
        if (getenv("AIPT_F16_MINPIX")) ctx->dn->opt_f16_minpix = atoll(getenv("AIPT_F16_MINPIX"));
        if (getenv("AIPT_F16_SMALL_MINPIX")) ctx->dn->opt_small_minpix = atoll(getenv("AIPT_F16_SMALL_MINPIX"));
#endif
        // conv3x3_f16x3r declares its LDS at launch: all weight chunks of a channel group + tables
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_f16x3r<false, 12, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_f16x3r<true, 12, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_f16x3r<false, 8, 3, false, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_f16x3r<true, 8, 3, false, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    return ctx->dn;
}

// The name rocprofv3 reports for a conv3x3_f16x3<1, rows, planar, w16> instantiation, without blanks: bench.py and
// tools/pmc_summarize.py match kernel rows of profiles/ on it, so every launch site takes it from here.
static void f16x3_name(char* dst, size_t n, int rows, bool planar, bool w16, int kys) {
    snprintf(dst, n, "conv3x3_f16x3<1,%d,%s,%s,%d>", rows, planar ? "true" : "false", w16 ? "true" : "false", kys);
}
#define F16X3_NAME_8ROW "conv3x3_f16x3<1,8,false,false,1>"

template <int RW, int MBX, int NBB>
static void launch_mfma(ConvArgs a, dim3 grid, hipStream_t st) {
    a.tiles_x = grid.x; a.tiles_y = grid.y; a.groups = grid.z;
    hipLaunchKernelGGL((conv3x3_mfma<RW, MBX, NBB>), dim3(grid_1d(grid.x, grid.y, grid.z)), dim3(256), 0, st, a);
}

struct TileChoice { int rw, mbx, nbb; };

// Pick the tile so that the launch has enough workgroups for 256 CUs (MI355X): big tiles for the full-resolution
// levels, narrow channel groups and small tiles for the deep, low-resolution levels.
static TileChoice choose_tile(int H, int W, int NB) {
    auto blocks = [&](int rw, int mbx, int nbb) {
        return (long)((W + 16 * mbx - 1) / (16 * mbx)) * ((H + 4 * rw - 1) / (4 * rw)) * (NB / nbb);
    };
    if (NB <= 3 && blocks(2, 2, NB) >= 512) return {2, 2, NB};
    const int cand[][3] = {{1, 2, 3}, {1, 2, 2}, {1, 2, 1}, {1, 1, 1}};
    TileChoice best = {1, 1, 1};
    long bestb = -1;
    for (auto& c : cand) {
        if (NB % c[2]) continue;
        const long b = blocks(c[0], c[1], c[2]);
        if (b >= 400) return {c[0], c[1], c[2]};
        if (b > bestb) { bestb = b; best = {c[0], c[1], c[2]}; }
    }
    return best;
}

static int conv_nblk(const TileChoice& t, int H, int W) {
    return ((W + 16 * t.mbx - 1) / (16 * t.mbx)) * ((H + 4 * t.rw - 1) / (4 * t.rw));
}

// May a conv hold its NORMALISED input activations y = lrelu(a x + b) as fp16 pairs with |y| * scale <= 65 504?
// With batch statistics |y| <= |gamma| sqrt(n - 1) + |beta| over the n pixels THE STATISTICS RAN OVER (one outlier carrying all
// of the variance) -- the producer's pixel count, not the consumer's: the first conv of encoder level i >= 1 and the
// depth-to-space dec1.c1 read a fused-pool tensor whose statistics ran over four times their own pixels (rounds 3-4 used the
// consumer's H x W there and under-estimated the bound by 2x: ADVICE r4).  So the answer follows from the loaded weights and the
// sources' BnRef: limit = 4 000 for conv3x3_f16x3r (activations times 2^4: beyond 4 094 the hi half saturates, beyond 8 190 the
// lo half becomes inf and inf x 0 = NaN spreads over the frame), 65 000 for conv3x3_f16x3.  A level that fails a limit runs on
// the next kernel down (conv3x3_f16x3, then the exact f32 MFMA kernel, which has the range of fp32).  Running statistics bound
// nothing: there the caller keeps |y| < 4 094 (include/aiptd.h) or selects AIPT_DN_IMPL_MFMA.
static bool f16_range_ok(const DenoiseState* s, bool batch, double stat_pixels, double limit) {
    if (!batch) return true;
    const double bound = (double)s->bn_gmax * sqrt(stat_pixels) + (double)s->bn_bmax;
    return bound < limit;       // (false for NaN)
}
// the largest pixel count the batch statistics of a conv's sources ran over (0: no source is normalised by batch statistics)
static double conv_stat_pixels(const ConvSrc& a, const ConvSrc& b) {
    double n = 0.0;
    if (a.bn.stat && a.bn.inv_n > 0.0) n = 1.0 / a.bn.inv_n;
    if (b.C && b.bn.stat && b.bn.inv_n > 0.0 && 1.0 / b.bn.inv_n > n) n = 1.0 / b.bn.inv_n;
    return n;
}
// true when the LAST conv of an encoder block at an h x w level (its input: the block's previous conv, statistics over h x w
// pixels) runs on a split-fp16 kernel: those fuse the 2x2 pool of the block's output into their epilogue (ConvArgsH::pool_out)
static bool conv_fuses_pool(const DenoiseState* s, bool batch, int H, int W) {
    return s->opt_fused_pool && impl_is_f16(s->impl) && (long long)H * W >= s->opt_small_minpix &&
           f16_range_ok(s, batch, (double)H * (double)W, 65000.0);
}

// One conv layer: picks the kernel for the level (implementation, size, operand range), launches it and leaves in `dst` the raw
// output together with how its consumers normalise it (batch statistics accumulated by the launch, or the running-statistics affine).
static int run_conv(aipt_ctx* ctx, DenoiseState* s, int li, const Tensor& A, int upA, const Tensor* B, int upB,
                    int H, int W, int out_lrelu, Tensor& dst, bool batch, bool use_b, const Tensor* pool_dst = nullptr,
                    float* final_out = nullptr, int final_h = 0, int final_w = 0) {
    const LayerW& L = s->L[li];
    ConvArgs g;
    g.a = ConvSrc{A.p, A.bn, A.C, upA, A.slope, A.planar};
    if (B && use_b) g.b = ConvSrc{B->p, B->bn, B->C, upB, B->slope, B->planar};
    else g.b = ConvSrc{nullptr, BnRef{nullptr, nullptr, nullptr, nullptr, 0, 0.0}, 0, 0, 1.0f, 0};
    g.H = H; g.W = W;
    g.w = L.d_w; g.w_raw = s->impl == AIPT_DN_IMPL_MFMA_F16W ? L.d_w_raw16 : L.d_w_raw; g.bias = L.d_bias;
    g.cin = L.cin; g.cout = L.cout; g.NP = L.NP;
    g.nchunks = (pad4(g.a.C) + pad4(g.b.C) + KC - 1) / KC;      // padded-concat channel space
    g.out = dst.p; g.out_lrelu = out_lrelu;
    g.d2s = 0;
    if (B && use_b && upA != upB) return fail(ctx, AIPT_E_STATE, "layer %d: concat sources must share the resampling mode", li);
    const int expect = A.C + (B ? B->C : 0);
    if (expect != L.cin || A.C != L.ca)
        return fail(ctx, AIPT_E_STATE, "layer %d: %d+%d input channels wired, %d+%d expected", li, A.C, expect - A.C, L.ca, L.cin - L.ca);
    long long* const stat = batch ? s->stat[s->sset] + (size_t)li * DenoiseState::STAT_LAYER : nullptr;
    g.stat = stat; g.sc = DenoiseState::STAT_SC;
    const double nstat = conv_stat_pixels(g.a, g.b);       // what bounds the normalised inputs (f16_range_ok)
    const bool prof = ((s->prof_mask >> li) & 1u) && s->prof_calls < s->prof_max && s->prof_seen % s->prof_every == 0;
    hipEvent_t* pev = prof ? &s->prof_ev[((size_t)s->prof_calls * NLAYERS + li) * 2] : nullptr;
#ifdef AIPT_DEBUG_HOOKS
    // AIPT_DEBUG_LAYER_MASK (tools/coresidency, debug builds only): launch only the conv layers whose bit is set -- the outputs
    // are then garbage; for bisecting which launch of a forward pass disturbs a kernel running beside it
    static const unsigned long debug_mask = getenv("AIPT_DEBUG_LAYER_MASK") ? strtoul(getenv("AIPT_DEBUG_LAYER_MASK"), nullptr, 0) : ~0ul;
    if (!((debug_mask >> li) & 1ul)) {
        if (batch) dst.bn = BnRef{nullptr, stat, L.d_gamma, L.d_beta, DenoiseState::STAT_SC, 1.0 / ((double)H * (double)W)};
        else dst.bn = BnRef{L.d_ab_running, nullptr, nullptr, nullptr, 0, 0.0};
        return AIPT_OK;
    }
#endif
    if (prof) AIPT_HIP(ctx, hipEventRecord(pev[0], s->cur));
    if (s->impl == AIPT_DN_IMPL_VALU) {
        snprintf(s->kname[li], sizeof(s->kname[li]), "conv3x3_valu");
        hipLaunchKernelGGL(conv3x3_valu, dim3((W + 63) / 64, H, L.cout), dim3(64), 0, s->cur, g);
        if (batch)
            hipLaunchKernelGGL(channel_stats, dim3(L.cout), dim3(256), 0, s->cur, dst.p, (size_t)H * W, stat);
    } else if (L.d_wsplit_d2s && upA && g.b.C && impl_is_f16(s->impl) && 4 * L.cout <= 16 && f16_range_ok(s, batch, nstat, 65000.0)) {
        // upsample + conv with 3 outputs on the split-fp16 kernel: half-resolution conv, 12 virtual channels, depth-to-space store
        ConvArgsH gh;
        gh.a = g.a; gh.b = g.b; gh.a.up = 0; gh.b.up = 0;
        gh.H = H / 2; gh.W = W / 2;
        // fp16-weight mode: the virtual-channel weights are sums of fp16-ROUNDED taps and keep their hi/lo split (rounding the
        // sums again would not be "the conv with fp16 weights")
        gh.wsplit = s->impl == AIPT_DN_IMPL_MFMA_F16W ? L.d_wsplit_d2s16 : L.d_wsplit_d2s; gh.bias = L.d_bias32_d2s;
        gh.cout = 4 * L.cout; gh.coutp = 32;
        gh.nchunks = L.nchunks16; gh.wchunks = L.nchunks16; gh.ca16 = L.ca16;
        gh.out = dst.p; gh.out_lrelu = out_lrelu;
        gh.stat = stat; gh.sc = DenoiseState::STAT_SC;
        gh.d2s = L.cout;
        gh.pool_out = nullptr; gh.pool_gamma = nullptr; gh.ablate = 0;
#ifdef AIPT_R_CLOCK
        gh.ablate = li;
#endif
        const int r_wpg = s->run_cus / 8;
        if ((long long)gh.H * gh.W >= s->opt_r_minpix && gh.nchunks <= RR_MAXCH && r_wpg >= 1 && !(gh.H & 1) && !(gh.W & 1) && L.d_wsplit1_d2s[0] &&
            (long long)H * W * 16 < (1ll << 31) &&
            f16_range_ok(s, batch, nstat, 4000.0)) {
            // the register-staged kernel: 16 virtual channels (4 x parity + channel) in its one group of 32
            gh.wsplit = L.d_wsplit1_d2s[w16_mode(s) ? 1 : 0]; gh.bias = L.d_bias32_d2s4;
            gh.cout = 16;
            gh.tiles_x = (gh.W + RR_PX - 1) / RR_PX; gh.tiles_y = (gh.H + 3) / 4; gh.groups = 1;
            snprintf(s->kname[li], sizeof(s->kname[li]), "conv3x3_f16x3r<false,%d,%d,false,4,false>", R_NWV, R_PF);
            hipLaunchKernelGGL((conv3x3_f16x3r<false, R_NWV, R_PF, false>), dim3(8u * (unsigned)r_wpg), dim3(R_NWV * 64), convr_lds_bytes(gh.nchunks, false), s->cur, gh);
        } else {
            const dim3 grid((gh.W + 31) / 32, (gh.H + 7) / 8, 1);
            gh.tiles_x = grid.x; gh.tiles_y = grid.y; gh.groups = 1;
            snprintf(s->kname[li], sizeof(s->kname[li]), "%s", F16X3_NAME_8ROW);   // the <1,8,false,false> instantiation in both weight modes
            hipLaunchKernelGGL((conv3x3_f16x3<1, 8>), dim3(grid_1d(grid.x, grid.y, 1)), dim3(512), 0, s->cur, gh);
        }
    } else if (L.d_w_d2s && upA) {
        // upsample + conv with 3 outputs -> half-resolution conv with 12 virtual channels + depth-to-space store
        g.a.up = 0; g.b.up = 0;
        g.H = H / 2; g.W = W / 2;
        g.w = L.d_w_d2s; g.bias = L.d_bias_d2s;
        g.cout = 4 * L.cout; g.NP = 16; g.d2s = L.cout;
        const TileChoice t = choose_tile(g.H, g.W, 1);
        const dim3 grid((g.W + 16 * t.mbx - 1) / (16 * t.mbx), (g.H + 4 * t.rw - 1) / (4 * t.rw), 1);
        snprintf(s->kname[li], sizeof(s->kname[li]), "conv3x3_mfma<%d,%d,1>", t.rw, t.mbx);
        if (t.rw == 2 && t.mbx == 2) launch_mfma<2, 2, 1>(g, grid, s->cur);
        else if (t.rw == 1 && t.mbx == 2) launch_mfma<1, 2, 1>(g, grid, s->cur);
        else launch_mfma<1, 1, 1>(g, grid, s->cur);
    } else if (L.cout == 3 && L.cin == 3 && !B && !upA && !g.a.planar) {
        // the output layer: statistics pass (batch statistics only), then conv + normalisation + crop straight into the caller's planes
        if (!final_out) return fail(ctx, AIPT_E_STATE, "layer %d: the single-quad kernel writes the network output", li);
        const int strips = ((W + QCOLS - 1) / QCOLS) * ((H + QROWS - 1) / QROWS);      // one wave per 62 x 8 pixel strip
        const int nblk = (strips + 3) / 4;
        g.out_planar = final_out; g.oh = final_h; g.ow = final_w; g.out_slope = SLOPE;
        g.out_bn = batch ? BnRef{nullptr, stat, L.d_gamma, L.d_beta, DenoiseState::STAT_SC, 1.0 / ((double)H * (double)W)}
                         : BnRef{L.d_ab_running, nullptr, nullptr, nullptr, 0, 0.0};
        snprintf(s->kname[li], sizeof(s->kname[li]), "conv3x3_quad<3,3,false>");
        if (batch) hipLaunchKernelGGL((conv3x3_quad<3, 3, true>), dim3(nblk), dim3(256), 0, s->cur, g);
        hipLaunchKernelGGL((conv3x3_quad<3, 3, false>), dim3(nblk), dim3(256), 0, s->cur, g);
    } else if (impl_is_f16(s->impl) && (long long)H * W >= s->opt_small_minpix && f16_range_ok(s, batch, nstat, 65000.0)) {
        // split-fp16 MFMA
        ConvArgsH gh;
        gh.a = g.a; gh.b = g.b; gh.H = H; gh.W = W;
        gh.wsplit = L.d_wsplit; gh.bias = L.d_bias32;
        gh.cout = L.cout; gh.coutp = L.coutp32;
        gh.nchunks = g.b.C ? L.nchunks16 : L.ca16;                 // an all-zero second source (hidden reset) is skipped
        gh.wchunks = L.nchunks16; gh.ca16 = L.ca16;
        gh.out = dst.p; gh.out_lrelu = out_lrelu;
        gh.stat = stat; gh.sc = DenoiseState::STAT_SC;
        gh.d2s = 0;
        gh.pool_out = pool_dst ? pool_dst->p : nullptr; gh.pool_gamma = L.d_gamma;
        gh.ablate = 0;
#ifdef AIPT_R_CLOCK
        gh.ablate = li;
#endif
#ifdef AIPT_CONV_ABLATE
        static const int ablate_env = getenv("AIPT_CONV_ABLATE") ? (int)strtol(getenv("AIPT_CONV_ABLATE"), nullptr, 0) : 0;
        gh.ablate = ablate_env;
#endif
        const bool w16 = w16_mode(s);
        if (gh.a.planar && (gh.b.C || gh.a.up || gh.a.slope != 1.0f || gh.a.bn.stat || gh.a.bn.ab || gh.a.C > KH))
            return fail(ctx, AIPT_E_STATE, "layer %d: a planar conv input must be the untransformed network input", li);
        // the levels of >= opt_r_minpix pixels: persistent register-staged kernel (conv3x3_f16x3r), when the group's weights fit
        // LDS and the normalised activations provably fit its operand range (f16_range_ok)
        const int r_groups = L.coutp32 / 32, r_wpg = (s->run_cus / 8) / r_groups;
        if ((long long)H * W >= s->opt_r_minpix && gh.nchunks <= (w16 ? 2 * RR_MAXCH : RR_MAXCH) && r_wpg >= 1 && !(H & 1) && !(W & 1) &&
            (long long)pad16(L.cout) * H * W * 4 < (1ll << 31) &&      // its output descriptors address the tensor with 31-bit offsets
            (gh.a.planar || f16_range_ok(s, batch, nstat, 4000.0))) {
            // planar input: the wide-range two-accumulator arithmetic on the hi + 2^11 lo slabs (see the kernel)
            gh.wsplit = gh.a.planar ? L.d_wsplit : w16 ? L.d_wsplit1_16 : L.d_wsplit1;
            gh.tiles_x = (W + RR_PX - 1) / RR_PX; gh.tiles_y = (H + 3) / 4; gh.groups = r_groups;
            const unsigned pgrid = 8u * (unsigned)r_wpg * (unsigned)r_groups;
            const size_t lds = convr_lds_bytes(gh.nchunks, w16);
            // the name rocprofv3 reports for the instantiation, without blanks (bench.py matches profiles/ on it)
            snprintf(s->kname[li], sizeof(s->kname[li]), "conv3x3_f16x3r<%s,%d,%d,false,4,%s>", w16 ? "true" : "false",
                     gh.a.planar ? 8 : R_NWV, gh.a.planar ? 3 : R_PF, gh.a.planar ? "true" : "false");
            if (gh.a.planar && w16) hipLaunchKernelGGL((conv3x3_f16x3r<true, 8, 3, false, 4, true>), dim3(pgrid), dim3(512), lds, s->cur, gh);
            else if (gh.a.planar) hipLaunchKernelGGL((conv3x3_f16x3r<false, 8, 3, false, 4, true>), dim3(pgrid), dim3(512), lds, s->cur, gh);
            else if (w16) hipLaunchKernelGGL((conv3x3_f16x3r<true, R_NWV, R_PF, false>), dim3(pgrid), dim3(R_NWV * 64), lds, s->cur, gh);
            else hipLaunchKernelGGL((conv3x3_f16x3r<false, R_NWV, R_PF, false>), dim3(pgrid), dim3(R_NWV * 64), lds, s->cur, gh);
        } else {
            // LDS-tiled kernel.  Tile rows = waves per workgroup: 8, or 4 on the levels below opt_f16_minpix (too few 8 x 32 tiles
            // for 256 CUs); the planar input always on 8-row tiles
            // ... and, with the tap-row split (12 waves per 4-row workgroup, one workgroup per CU), on any level whose 4-row tiles
            // all fit the chip at once: the launch then lasts one workgroup's chain of chunk steps, which the split shortens
            // (dec4.c1 28 -> 23 us; enc4.* with 345 workgroups would need two rounds: 8-row tiles stay)
            const long long wg4 = (long long)((W + 31) / 32) * ((H + 3) / 4) * (L.coutp32 / 32);
            const bool one_round4 = s->opt_ky_split && wg4 <= s->num_cus;       // (the device's CUs, not a masked stream's: the selection must not change the bits)
            const int rows = gh.a.planar ? 8 : ((long long)H * W >= s->opt_f16_minpix && !one_round4) ? 8 : 4;
            const dim3 grid((W + 31) / 32, (H + rows - 1) / rows, L.coutp32 / 32);
            gh.tiles_x = grid.x; gh.tiles_y = grid.y; gh.groups = grid.z;
            const unsigned nb1 = grid_1d(grid.x, grid.y, grid.z);
            const int kys = rows == 4 && s->opt_ky_split ? 3 : 1;      // 4-row tiles: three waves per row, one per tap row
            f16x3_name(s->kname[li], sizeof(s->kname[li]), rows, gh.a.planar != 0, w16, kys);
            if (gh.a.planar && w16) hipLaunchKernelGGL((conv3x3_f16x3<1, 8, true, true>), dim3(nb1), dim3(512), 0, s->cur, gh);
            else if (gh.a.planar) hipLaunchKernelGGL((conv3x3_f16x3<1, 8, true>), dim3(nb1), dim3(512), 0, s->cur, gh);
            else if (w16 && rows == 8) hipLaunchKernelGGL((conv3x3_f16x3<1, 8, false, true>), dim3(nb1), dim3(512), 0, s->cur, gh);
            else if (w16 && kys == 3) hipLaunchKernelGGL((conv3x3_f16x3<1, 4, false, true, 3>), dim3(nb1), dim3(768), 0, s->cur, gh);
            else if (w16) hipLaunchKernelGGL((conv3x3_f16x3<1, 4, false, true>), dim3(nb1), dim3(256), 0, s->cur, gh);
            else if (rows == 8) hipLaunchKernelGGL((conv3x3_f16x3<1, 8>), dim3(nb1), dim3(512), 0, s->cur, gh);
            else if (kys == 3) hipLaunchKernelGGL((conv3x3_f16x3<1, 4, false, false, 3>), dim3(nb1), dim3(768), 0, s->cur, gh);
            else hipLaunchKernelGGL((conv3x3_f16x3<1, 4>), dim3(nb1), dim3(256), 0, s->cur, gh);
        }
    } else {
        const TileChoice t = choose_tile(H, W, L.NB);
        const dim3 grid((W + 16 * t.mbx - 1) / (16 * t.mbx), (H + 4 * t.rw - 1) / (4 * t.rw), L.NB / t.nbb);
        const int key = t.rw * 100 + t.mbx * 10 + t.nbb;
        snprintf(s->kname[li], sizeof(s->kname[li]), "conv3x3_mfma<%d,%d,%d>", t.rw, t.mbx, t.nbb);
        switch (key) {
            case 221: launch_mfma<2, 2, 1>(g, grid, s->cur); break;
            case 222: launch_mfma<2, 2, 2>(g, grid, s->cur); break;
            case 223: launch_mfma<2, 2, 3>(g, grid, s->cur); break;
            case 121: launch_mfma<1, 2, 1>(g, grid, s->cur); break;
            case 122: launch_mfma<1, 2, 2>(g, grid, s->cur); break;
            case 123: launch_mfma<1, 2, 3>(g, grid, s->cur); break;
            case 111: launch_mfma<1, 1, 1>(g, grid, s->cur); break;
            default: return fail(ctx, AIPT_E_STATE, "no conv instantiation for tile %d", key);
        }
    }
    if (prof) AIPT_HIP(ctx, hipEventRecord(pev[1], s->cur));
#ifdef AIPT_R_CLOCK
    {
        static unsigned long long h[32][8];
        static std::vector<unsigned long long> hw(32 * 4096 * 4);
        static int calls = 0;
        auto reset = [&]() {
            for (auto& r : h) { r[0] = ~0ull; r[1] = 0; r[2] = ~0ull; r[3] = 0; r[4] = 0; r[5] = 0; r[6] = 0; r[7] = 1; }
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rclk), h, sizeof(h));
            std::fill(hw.begin(), hw.end(), 0ull);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rclk_w), hw.data(), hw.size() * 8);
        };
        if (calls == 0) { (void)hipDeviceSynchronize(); reset(); }
        if (li == NLAYERS - 1 && ++calls % 16 == 12) {
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rclk), sizeof(h));
            (void)hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(g_rclk_w), hw.size() * 8);
            unsigned long long prev_end = 0;
            for (int l = 0; l < NLAYERS; l++) {
                for (int w = 0; w < 4096; w++) {
                    const unsigned long long* q = &hw[((size_t)l * 4096 + w) * 4];
                    if (!q[0] || !q[2]) continue;
                    h[l][0] = std::min(h[l][0], q[0]); h[l][1] = std::max(h[l][1], q[0]);
                    h[l][2] = std::min(h[l][2], q[2]); h[l][3] = std::max(h[l][3], q[2]);
                    h[l][4] = std::max(h[l][4], q[1]);
                }
                if (h[l][1] == 0) continue;
                fprintf(stderr, "rclk layer %2d %-44s gap %6.2f  ramp %5.2f  prologue %5.2f  first-end %6.2f  last-end %6.2f us | wave: %.3f GHz\n", l, s->kname[l],
                        prev_end ? ((double)h[l][0] - (double)prev_end) * 0.01 : 0.0, (double)(h[l][1] - h[l][0]) * 0.01, (double)(h[l][4] - h[l][0]) * 0.01,
                        (double)(h[l][2] - h[l][0]) * 0.01, (double)(h[l][3] - h[l][0]) * 0.01, (double)h[l][6] / (double)h[l][7] * 0.1);
                prev_end = h[l][3];
            }
            reset();
        } else if (li == NLAYERS - 1) { (void)hipDeviceSynchronize(); reset(); }
    }
#endif
    // how consumers normalise dst: batch statistics from the sums this launch accumulates, or the running-statistics affine
    if (batch) dst.bn = BnRef{nullptr, stat, L.d_gamma, L.d_beta, DenoiseState::STAT_SC, 1.0 / ((double)H * (double)W)};
    else dst.bn = BnRef{L.d_ab_running, nullptr, nullptr, nullptr, 0, 0.0};
    AIPT_HIP(ctx, hipGetLastError());
    return AIPT_OK;
}

}  // namespace aipt

using namespace aipt;

extern "C" {

int aipt_denoise_load_weights(aipt_ctx* ctx, const void* blob, size_t bytes) {
    AIPT_CHECK_CTX(ctx);
    if (!blob) return fail(ctx, AIPT_E_INVALID, "aipt_denoise_load_weights: blob is NULL");
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    const uint8_t* p = (const uint8_t*)blob;
    if (bytes < 16 || memcmp(p, "AIPTDW01", 8)) return fail(ctx, AIPT_E_FORMAT, "weight blob: bad magic");
    uint32_t n; memcpy(&n, p + 8, 4);
    if (n != NLAYERS) return fail(ctx, AIPT_E_FORMAT, "weight blob: %u layers, expected %d", n, NLAYERS);
    int cin[NLAYERS], cout[NLAYERS];
    build_table(cin, cout);
    size_t off = 16, nfl = 0;
    if (bytes < off + 8 * NLAYERS) return fail(ctx, AIPT_E_FORMAT, "weight blob: truncated header");
    for (int i = 0; i < NLAYERS; i++) {
        uint32_t ci, co; memcpy(&ci, p + off, 4); memcpy(&co, p + off + 4, 4); off += 8;
        if ((int)ci != cin[i] || (int)co != cout[i])
            return fail(ctx, AIPT_E_FORMAT, "weight blob: layer %d is %ux%u, expected %dx%d", i, ci, co, cin[i], cout[i]);
        nfl += (size_t)9 * ci * co + 5 * (size_t)co;
    }
    if (off + 4 * nfl != bytes) return fail(ctx, AIPT_E_FORMAT, "weight blob: %zu bytes, expected %zu", bytes, off + 4 * nfl);
    DenoiseState* s = state(ctx);
    free_weights(s);
    s->bn_gmax = 0.0f; s->bn_bmax = 0.0f;
    // the carried hidden states reference the old layers' gamma/beta/statistics: a reload resets the recurrent state
    s->hidden_valid = false;
    for (DenoiseState::ActSet& X : s->A)
        for (Tensor& t : X.Hid) t.bn = BnRef{nullptr, nullptr, nullptr, nullptr, 0, 0.0};
    std::vector<float> f(nfl);
    memcpy(f.data(), p + off, 4 * nfl);
    const float* q = f.data();
    for (int i = 0; i < NLAYERS; i++) {
        LayerW& L = s->L[i];
        L.cin = cin[i]; L.cout = cout[i];
        L.NB = (L.cout + 15) / 16; L.NP = L.NB * 16;
        // channels of the first concat source: l2a layers cat(out1, hidden), decoder c1 layers cat(prev, skip)
        static const int dec_prev[5] = {101, 76, 57, 43, 32};
        L.ca = L.cin;
        if (i < 18 && i % 3 == 1) L.ca = L.cin / 2;
        else if (i >= 18 && (i - 18) % 2 == 0) L.ca = dec_prev[(i - 18) / 2];
        const int PA = pad4(L.ca);
        L.pcin = PA + pad4(L.cin - L.ca);                       // padded-concat channel count
        auto pc_of = [&](int c) { return c < L.ca ? c : PA + (c - L.ca); };
        L.nchunks = (L.pcin + KC - 1) / KC;
        const float* w = q;          q += (size_t)9 * L.cin * L.cout;
        const float* b = q;          q += L.cout;
        const float* gamma = q;      q += L.cout;
        const float* beta = q;       q += L.cout;
        const float* mean = q;       q += L.cout;
        const float* var = q;        q += L.cout;
        // implicit-GEMM layout: [chunk][tap][kc][NP], zero padded in both cin and cout
        std::vector<float> wg((size_t)L.nchunks * 9 * KC * L.NP, 0.0f);
        for (int j = 0; j < L.cout; j++)
            for (int c = 0; c < L.cin; c++)
                for (int t = 0; t < 9; t++)
                    wg[(((size_t)(pc_of(c) / KC) * 9 + t) * KC + (pc_of(c) % KC)) * L.NP + j] = w[((size_t)j * L.cin + c) * 9 + t];
        std::vector<float> bp(L.NP, 0.0f);
        memcpy(bp.data(), b, 4 * L.cout);
        std::vector<float2> abr(L.cout);
        for (int j = 0; j < L.cout; j++) {
            const double sc = (double)gamma[j] / sqrt((double)var[j] + (double)BN_EPS);
            abr[j] = make_float2((float)sc, (float)((double)beta[j] - (double)mean[j] * sc));
        }
        if (i == 26) {
            // dec1.c1 = conv3x3(upsample2(x)) with 3 outputs: as a plain conv on the half-resolution x with 12 virtual
            // channels v = (2a+b)*3 + j, one per output parity (a,b); W'[v][c][dy][dx] = sum of the taps (ky,kx) whose
            // upsampled source row/col floor((a+ky-1)/2), floor((b+kx-1)/2) is dy-1, dx-1.  Zero padding is equivalent.
            const int vco = 4 * L.cout, vnp = (vco + 15) / 16 * 16;
            std::vector<float> wv((size_t)L.nchunks * 9 * KC * vnp, 0.0f), bv(vnp, 0.0f);
            for (int par = 0; par < 4; par++) {
                const int a = par >> 1, bb = par & 1;
                for (int j = 0; j < L.cout; j++) {
                    const int v = par * L.cout + j;
                    bv[v] = b[j];
                    for (int c = 0; c < L.cin; c++)
                        for (int ky = 0; ky < 3; ky++)
                            for (int kx = 0; kx < 3; kx++) {
                                const int dy = (a + ky + 1) / 2 - 1, dx = (bb + kx + 1) / 2 - 1;   // floor((a+ky-1)/2)
                                wv[(((size_t)(pc_of(c) / KC) * 9 + (dy + 1) * 3 + (dx + 1)) * KC + (pc_of(c) % KC)) * vnp + v] +=
                                    w[((size_t)j * L.cin + c) * 9 + ky * 3 + kx];
                            }
                }
            }
            AIPT_HIP(ctx, hipMalloc((void**)&L.d_w_d2s, wv.size() * 4));
            AIPT_HIP(ctx, hipMalloc((void**)&L.d_bias_d2s, vnp * 4));
            AIPT_HIP(ctx, hipMemcpy(L.d_w_d2s, wv.data(), wv.size() * 4, hipMemcpyHostToDevice));
            AIPT_HIP(ctx, hipMemcpy(L.d_bias_d2s, bv.data(), vnp * 4, hipMemcpyHostToDevice));
            // the same virtual-channel weights, split and tiled for conv3x3_f16x3 (one group of 32, K16 concat space)
            const int cbv = L.cin - L.ca, nch = pad16(L.ca) / KH + pad16(cbv) / KH;
            for (int rounded = 0; rounded < 2; rounded++) {       // 1: taps rounded to fp16 first (AIPT_DN_IMPL_MFMA_F16W)
                std::vector<float> wf((size_t)vco * (nch * KH) * 9, 0.0f);                 // [v][kc][tap]
                for (int par = 0; par < 4; par++) {
                    const int a = par >> 1, bb = par & 1;
                    for (int j = 0; j < L.cout; j++)
                        for (int c = 0; c < L.cin; c++) {
                            const int kc = c < L.ca ? c : pad16(L.ca) + (c - L.ca);
                            for (int ky = 0; ky < 3; ky++)
                                for (int kx = 0; kx < 3; kx++) {
                                    const int dy = (a + ky + 1) / 2 - 1, dx = (bb + kx + 1) / 2 - 1;
                                    wf[((size_t)(par * L.cout + j) * (nch * KH) + kc) * 9 + (dy + 1) * 3 + (dx + 1)] +=
                                        (rounded ? (float)(_Float16)w[((size_t)j * L.cin + c) * 9 + ky * 3 + kx] : w[((size_t)j * L.cin + c) * 9 + ky * 3 + kx]);
                                }
                        }
                }
                std::vector<_Float16> ws((size_t)nch * (WSLAB / 2), (_Float16)0.0f);
                for (int v = 0; v < vco; v++)
                    for (int kc = 0; kc < nch * KH; kc++)
                        for (int t = 0; t < 9; t++) {
                            const float x = wf[((size_t)v * (nch * KH) + kc) * 9 + t];
                            const _Float16 h = (_Float16)x;
                            const size_t o = (size_t)(kc / KH) * (WSLAB / 2) + ((size_t)t * 32 + v) * KH + (kc % KH);
                            ws[o] = h;
                            ws[o + 9 * 32 * KH] = (_Float16)((x - (float)h) * LO_SCALE);
                        }
                unsigned char*& dst_w = rounded ? L.d_wsplit_d2s16 : L.d_wsplit_d2s;
                AIPT_HIP(ctx, hipMalloc((void**)&dst_w, ws.size() * 2));
                AIPT_HIP(ctx, hipMemcpy(dst_w, ws.data(), ws.size() * 2, hipMemcpyHostToDevice));
                if (L.cout <= 3) {                                // conv3x3_f16x3r's copy (see LayerW)
                    std::vector<_Float16> w1((size_t)nch * (WSLAB / 2), (_Float16)0.0f);
                    for (int v = 0; v < vco; v++)
                        for (int kc = 0; kc < nch * KH; kc++)
                            for (int t = 0; t < 9; t++) {
                                const float x = wf[((size_t)v * (nch * KH) + kc) * 9 + t] * WS1;
                                const _Float16 h = (_Float16)x;
                                const int v4 = 4 * (v / L.cout) + v % L.cout;
                                const size_t o = (size_t)(kc / KH) * (WSLAB / 2) + ((size_t)t * 32 + v4) * KH + (kc % KH);
                                w1[o] = h;
                                w1[o + 9 * 32 * KH] = (_Float16)(x - (float)h);
                            }
                    AIPT_HIP(ctx, hipMalloc((void**)&L.d_wsplit1_d2s[rounded], w1.size() * 2));
                    AIPT_HIP(ctx, hipMemcpy(L.d_wsplit1_d2s[rounded], w1.data(), w1.size() * 2, hipMemcpyHostToDevice));
                }
            }
            if (L.cout <= 3) {
                std::vector<float> b4(32, 0.0f);
                for (int v = 0; v < vco; v++) b4[4 * (v / L.cout) + v % L.cout] = bv[v];
                AIPT_HIP(ctx, hipMalloc((void**)&L.d_bias32_d2s4, 32 * 4));
                AIPT_HIP(ctx, hipMemcpy(L.d_bias32_d2s4, b4.data(), 32 * 4, hipMemcpyHostToDevice));
            }
            std::vector<float> b32v(32, 0.0f);
            for (int v = 0; v < vco; v++) b32v[v] = bv[v];
            AIPT_HIP(ctx, hipMalloc((void**)&L.d_bias32_d2s, 32 * 4));
            AIPT_HIP(ctx, hipMemcpy(L.d_bias32_d2s, b32v.data(), 32 * 4, hipMemcpyHostToDevice));
        }
        {   // split-fp16 weights, pre-tiled per (output-channel group, chunk) slab
            const int cb = L.cin - L.ca;
            L.coutp32 = (L.cout + 31) / 32 * 32;
            L.ca16 = pad16(L.ca) / KH;
            L.nchunks16 = L.ca16 + pad16(cb) / KH;
            const size_t nh = (size_t)(L.coutp32 / 32) * L.nchunks16 * (WSLAB / 2);
            std::vector<_Float16> ws(nh, (_Float16)0.0f), ws1(nh, (_Float16)0.0f), ws1r(nh, (_Float16)0.0f);
            for (int j = 0; j < L.cout; j++)
                for (int c = 0; c < L.cin; c++) {
                    const int kc = c < L.ca ? c : pad16(L.ca) + (c - L.ca);
                    for (int t = 0; t < 9; t++) {
                        const float v = w[((size_t)j * L.cin + c) * 9 + t];
                        const _Float16 h = (_Float16)v;
                        const size_t slab = ((size_t)(j / 32) * L.nchunks16 + kc / KH) * (WSLAB / 2);
                        const size_t o = slab + ((size_t)t * 32 + (j % 32)) * KH + (kc % KH);
                        ws[o] = h;
                        ws[o + 9 * 32 * KH] = (_Float16)((v - (float)h) * LO_SCALE);
                        const float v1 = w[((size_t)j * L.cin + c) * 9 + t] * WS1;
                        const _Float16 h1 = (_Float16)v1;
                        ws1[o] = h1;
                        ws1[o + 9 * 32 * KH] = (_Float16)(v1 - (float)h1);
                        // fp16-weight mode: "the model with its conv weights rounded to fp16" on every level -- 2^7 fp16(w), which is
                        // not fp16(2^7 w) when w is an fp16 subnormal
                        ws1r[o] = (_Float16)((float)(_Float16)w[((size_t)j * L.cin + c) * 9 + t] * WS1);
                    }
                }
            AIPT_HIP(ctx, hipMalloc((void**)&L.d_wsplit1, nh * 2));
            AIPT_HIP(ctx, hipMemcpy(L.d_wsplit1, ws1.data(), nh * 2, hipMemcpyHostToDevice));
            AIPT_HIP(ctx, hipMalloc((void**)&L.d_wsplit1_16, nh * 2));
            AIPT_HIP(ctx, hipMemcpy(L.d_wsplit1_16, ws1r.data(), nh * 2, hipMemcpyHostToDevice));
            std::vector<float> b32(L.coutp32, 0.0f);
            memcpy(b32.data(), b, 4 * L.cout);
            AIPT_HIP(ctx, hipMalloc((void**)&L.d_wsplit, nh * 2));
            AIPT_HIP(ctx, hipMalloc((void**)&L.d_bias32, L.coutp32 * 4));
            AIPT_HIP(ctx, hipMemcpy(L.d_wsplit, ws.data(), nh * 2, hipMemcpyHostToDevice));
            AIPT_HIP(ctx, hipMemcpy(L.d_bias32, b32.data(), L.coutp32 * 4, hipMemcpyHostToDevice));
        }
        AIPT_HIP(ctx, hipMalloc((void**)&L.d_w, wg.size() * 4));
        AIPT_HIP(ctx, hipMalloc((void**)&L.d_w_raw, (size_t)9 * L.cin * L.cout * 4));
        AIPT_HIP(ctx, hipMalloc((void**)&L.d_bias, L.NP * 4));
        AIPT_HIP(ctx, hipMalloc((void**)&L.d_gamma, L.cout * 4));
        AIPT_HIP(ctx, hipMalloc((void**)&L.d_beta, L.cout * 4));
        AIPT_HIP(ctx, hipMalloc((void**)&L.d_ab_running, L.cout * sizeof(float2)));
        AIPT_HIP(ctx, hipMemcpy(L.d_w, wg.data(), wg.size() * 4, hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMemcpy(L.d_w_raw, w, (size_t)9 * L.cin * L.cout * 4, hipMemcpyHostToDevice));
        {
            std::vector<float> w16((size_t)9 * L.cin * L.cout);
            for (size_t k = 0; k < w16.size(); k++) w16[k] = (float)(_Float16)w[k];
            AIPT_HIP(ctx, hipMalloc((void**)&L.d_w_raw16, w16.size() * 4));
            AIPT_HIP(ctx, hipMemcpy(L.d_w_raw16, w16.data(), w16.size() * 4, hipMemcpyHostToDevice));
        }
        AIPT_HIP(ctx, hipMemcpy(L.d_bias, bp.data(), L.NP * 4, hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMemcpy(L.d_gamma, gamma, L.cout * 4, hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMemcpy(L.d_beta, beta, L.cout * 4, hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMemcpy(L.d_ab_running, abr.data(), L.cout * sizeof(float2), hipMemcpyHostToDevice));
        for (int j = 0; j < L.cout; j++) {
            if (!(fabsf(gamma[j]) <= s->bn_gmax)) s->bn_gmax = fabsf(gamma[j]);      // (NaN-proof: a NaN gamma makes the bound NaN -> exact kernels)
            if (!(fabsf(beta[j]) <= s->bn_bmax)) s->bn_bmax = fabsf(beta[j]);
        }
    }
    s->have_weights = true;
    return AIPT_OK;
}

int aipt_denoise_configure(aipt_ctx* ctx, int height, int width) {
    AIPT_CHECK_CTX(ctx);
    if (height <= 0 || width <= 0 || (height % 32) || (width % 32))
        return fail(ctx, AIPT_E_INVALID, "aipt_denoise_configure: %dx%d is not a positive multiple of 32 "
                    "(5 x MaxPool2d(2) + skip concat, recurrent_autoencoder_model.py:98-107,136-140)", height, width);
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    DenoiseState* s = state(ctx);
    if (s->H == height && s->W == width) return AIPT_OK;
    free_activations(s);
    auto alloc = [&](size_t bytes, void** out) -> int {
        AIPT_HIP(ctx, hipMalloc(out, bytes));
        s->allocs.push_back(*out);
        return AIPT_OK;
    };
    auto mk = [&](Tensor& t, int C, int lvl) -> int {
        t.C = C; t.slope = SLOPE; t.planar = 0;
        // C4 layout, allocated up to a multiple of 16 channels: conv3x3_f16x3r fetches whole 16-channel chunks without clamping the
        // quad index.  Consumers give the extra planes BN coefficients (0, 0); the LDS-tiled kernels never write them, the
        // register-staged kernel stores its pad quads there -- zeros, because pad weights and pad bias are exactly 0 and every
        // operand is finite (the range checks of f16_range_ok and the clamp of the planar input see to that)
        const size_t bytes = sizeof(float) * (size_t)pad16(C) * (height >> lvl) * (width >> lvl);
        int rc = alloc(bytes, (void**)&t.p);
        if (rc) return rc;
        AIPT_HIP(ctx, hipMemsetAsync(t.p, 0, bytes, ctx->stream));   // pad channels and pad planes stay 0
        t.bn = BnRef{nullptr, nullptr, nullptr, nullptr, 0, 0.0};
        return AIPT_OK;
    };
    int rc = 0;
    for (int a = 0; a < DenoiseState::NSET && !rc; a++) {
        DenoiseState::ActSet& X = s->A[a];
        rc = mk(X.In, 10, 0);
        for (int i = 0; i < 6 && !rc; i++) {
            const int c = i < 5 ? ENC_CH[i] : 101;
            rc = mk(X.T1[i], c, i);
            if (!rc) rc = mk(X.T2[i], c, i);
            if (!rc) rc = mk(X.Hid[i], c, i);
            if (!rc && i < 5) { rc = mk(X.P[i], c, i + 1); X.P[i].slope = 1.0f; }
        }
        for (int k = 1; k <= 5 && !rc; k++) {
            rc = mk(X.D1[k], DEC_CH[k], k - 1);
            if (!rc) rc = mk(X.D2[k], DEC_CH[k], k - 1);
        }
    }
    if (rc) { free_activations(s); return rc; }
    for (int k = 0; k < DenoiseState::STAT_SETS && !rc; k++)
        rc = alloc(sizeof(long long) * NLAYERS * DenoiseState::STAT_LAYER, (void**)&s->stat[k]);
    if (rc) { free_activations(s); return rc; }
    for (int a = 0; a < DenoiseState::NSET; a++)
        for (int l = 0; l < 6; l++) AIPT_HIP(ctx, hipEventCreateWithFlags(&s->ev_level[a][l], hipEventDisableTiming));
    for (int k = 0; k < DenoiseState::NSET; k++) AIPT_HIP(ctx, hipEventCreateWithFlags(&s->ev_fence[k], hipEventDisableTiming));
    s->sset = 0; s->aset = 0;
    s->H = height; s->W = width;
    s->hidden_valid = false;
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    return AIPT_OK;
}

int aipt_denoise_set_impl(aipt_ctx* ctx, int impl) {
    AIPT_CHECK_CTX(ctx);
    if (impl != AIPT_DN_IMPL_MFMA && impl != AIPT_DN_IMPL_VALU && impl != AIPT_DN_IMPL_MFMA_F16X3 && impl != AIPT_DN_IMPL_MFMA_F16W) return fail(ctx, AIPT_E_INVALID, "unknown impl %d", impl);
    state(ctx)->impl = impl;
    return AIPT_OK;
}

int aipt_denoise_set_option(aipt_ctx* ctx, int option, long long value) {
    AIPT_CHECK_CTX(ctx);
    DenoiseState* s = state(ctx);
    if (value < 0) return fail(ctx, AIPT_E_INVALID, "aipt_denoise_set_option: value %lld", value);
    switch (option) {
        case AIPT_DN_OPT_R_MINPIX: s->opt_r_minpix = value; break;
        case AIPT_DN_OPT_F16_MINPIX: s->opt_f16_minpix = value; break;
        case AIPT_DN_OPT_SMALL_MINPIX: s->opt_small_minpix = value; break;
        case AIPT_DN_OPT_FUSED_POOL: s->opt_fused_pool = value != 0; break;
        case AIPT_DN_OPT_KY_SPLIT: s->opt_ky_split = value != 0; break;
        default: return fail(ctx, AIPT_E_INVALID, "aipt_denoise_set_option: unknown option %d", option);
    }
    return AIPT_OK;
}

int aipt_denoise_reset_hidden(aipt_ctx* ctx) {
    AIPT_CHECK_CTX(ctx);
    state(ctx)->hidden_valid = false;
    return AIPT_OK;
}

int aipt_denoise(aipt_ctx* ctx, const float* d_in10, float* d_out3, uint32_t flags) {
    AIPT_CHECK_CTX(ctx);
    DenoiseState* s = state(ctx);
    return aipt::denoise_run(ctx, d_in10, d_out3, flags, s->H, s->W);
}

}  // extern "C"

// forward pass; the planar output is cropped to out_h x out_w (aipt_frame drops its padding here).  pipelined: the frame runs
// on the stream of its activation set (set 0: the context's stream, set k: ctx->pipe[k-1]) and waits, level by level, for the
// hidden states of the frame before it; the caller forks / joins the streams around a run of such frames.
int aipt::denoise_run(aipt_ctx* ctx, const float* d_in10, float* d_out3, uint32_t flags, int out_h, int out_w, bool pipelined,
                      hipStream_t on, int on_cus) {
    DenoiseState* s = state(ctx);
    if (!s->have_weights) return fail(ctx, AIPT_E_STATE, "aipt_denoise: no weights loaded");
    if (!s->H) return fail(ctx, AIPT_E_STATE, "aipt_denoise: call aipt_denoise_configure first");
    if (!d_in10 || !d_out3) return fail(ctx, AIPT_E_INVALID, "aipt_denoise: NULL buffer");
    if (out_h < 1 || out_h > s->H || out_w < 1 || out_w > s->W) return fail(ctx, AIPT_E_INVALID, "aipt_denoise: output %dx%d", out_w, out_h);
    const bool batch = (flags & AIPT_DN_BN_BATCH) != 0;
    const bool carry = (flags & AIPT_DN_HIDDEN_CARRY) != 0 && s->hidden_valid;
    const int H = s->H, W = s->W;
    int li = 0, rc = 0;
    constexpr int NSET = DenoiseState::NSET;
    const int a = (s->aset + 1) % NSET, ap = s->aset;          // this frame's set; the previous frame's
    DenoiseState::ActSet& X = s->A[a];
    DenoiseState::ActSet& prevX = s->A[ap];
    pipelined = pipelined && ctx->ev_fork && !on;
    auto stream_of = [&](int set) { return set == 0 ? ctx->stream : ctx->pipe[set - 1]; };
    hipStream_t st = on ? on : pipelined ? stream_of(a) : ctx->stream;     // on: the CU-masked denoiser stream of aipt_frame
    // the pass reads the G-buffer the last trace wrote: unless it runs on the CU-masked denoiser stream of aipt_frame_prefetch
    // (which orders itself behind ITS trace), it starts after the last trace, whichever stream that ran on
    if (ctx->last_trace_stream && ctx->last_trace_stream != st && st != ctx->st_dn)
        AIPT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_traced, 0));
    s->cur = st;
    s->aset = a;
    // a CU mask enables the same number of CUs on every XCD and workgroup b still starts on XCD b & 7 (tools/ubench/cumask.hip): the
    // persistent kernel takes one workgroup per ENABLED CU (sized for all CUs it would run two rounds of statically dealt items)
    s->run_cus = on && on_cus >= 8 && on_cus <= s->num_cus ? on_cus & ~7 : s->num_cus;
    // a forward pass whose launches are being timed (aipt_denoise_profile_*) runs alone: the other streams drain before it
    // and resume after it, so that an event pair brackets one kernel and not its overlap with another frame's
    const bool timed = pipelined && s->prof_mask && s->prof_calls < s->prof_max && s->prof_seen % s->prof_every == 0;
    if (timed) {
        for (int k = 1; k < NSET; k++) {
            AIPT_HIP(ctx, hipEventRecord(s->ev_fence[k - 1], stream_of((a + k) % NSET)));
            AIPT_HIP(ctx, hipStreamWaitEvent(st, s->ev_fence[k - 1], 0));
        }
    }
    if (batch) {   // this frame's BN sums go to the next set of the ring: the carried hidden states still point into the last one
        s->sset = (s->sset + 1) % DenoiseState::STAT_SETS;
        AIPT_HIP(ctx, hipMemsetAsync(s->stat[s->sset], 0, sizeof(long long) * NLAYERS * DenoiseState::STAT_LAYER, st));
    }
    // level l of this frame reads the hidden state the previous frame wrote at level l (on the other stream when pipelined)
    auto wait_hidden = [&](int l) -> hipError_t {
        if (pipelined && carry && s->ev_level_valid[ap][l]) return hipStreamWaitEvent(st, s->ev_level[ap][l], 0);
        return hipSuccess;
    };
    auto hidden_written = [&](int l) -> hipError_t {
        s->ev_level_valid[a][l] = pipelined;
        return pipelined ? hipEventRecord(s->ev_level[a][l], st) : hipSuccess;
    };
    Tensor in;
    // the G-buffer contract is planar [10][H][W] (pathtrace.cu:81-94).  The split-fp16 conv reads it as it is; the other
    // implementations get a C4 copy first.
    in = X.In; in.C = 10; in.slope = 1.0f; in.planar = 0;         // bn: identity
    const bool direct = impl_is_f16(s->impl) && (long long)H * W >= s->opt_f16_minpix && (long long)H * W >= s->opt_small_minpix && f16_range_ok(s, batch, 0.0, 65000.0);   // an untransformed input: no statistics
    if (direct) {
        in.p = const_cast<float*>(d_in10); in.planar = 1;
    } else {
        const size_t hw = (size_t)H * W, n = (size_t)pad4(10) * hw;
        const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
        hipLaunchKernelGGL(planar_to_c4, dim3(grid), dim3(256), 0, st, d_in10, 10, hw, X.In.p);
    }
    const Tensor* x = &in;
    // encoders: out1 = LReLU(BN(conv(X))); out2 = LReLU(BN(conv(BN(LReLU(conv(cat(out1, hidden))))))); then MaxPool
    for (int i = 0; i < 5; i++) {
        const int h = H >> i, w = W >> i;
        if ((rc = run_conv(ctx, s, li++, *x, 0, nullptr, 0, h, w, 0, X.T1[i], batch, false))) return rc;
        AIPT_HIP(ctx, wait_hidden(i));
        if ((rc = run_conv(ctx, s, li++, X.T1[i], 0, &prevX.Hid[i], 0, h, w, 1, X.T2[i], batch, carry))) return rc;
        Tensor t2 = X.T2[i]; t2.slope = 1.0f;        // LReLU already applied by the producer (conv -> LReLU -> BN)
        const bool fused_pool = conv_fuses_pool(s, batch, h, w);
        if ((rc = run_conv(ctx, s, li++, t2, 0, nullptr, 0, h, w, 0, X.Hid[i], batch, false, fused_pool ? &X.P[i] : nullptr))) return rc;
        X.Hid[i].slope = SLOPE;
        AIPT_HIP(ctx, hidden_written(i));
        if (fused_pool) {
            // the conv left the pooled RAW output (max or min by the sign of gamma): its consumers normalise it like Hid[i]
            X.P[i].bn = X.Hid[i].bn; X.P[i].slope = SLOPE;
        } else {
            const size_t n = (size_t)(h / 2) * (w / 2);                     // one thread per pooled pixel of a channel quad
            const int quads = pad4(ENC_CH[i]) / 4;
            int grid = (int)((n + 255) / 256);
            if (grid * quads > 4096) grid = (4096 + quads - 1) / quads;
            hipLaunchKernelGGL(pool2_norm, dim3(grid, quads), dim3(256), 0, st, X.Hid[i].p, X.Hid[i].bn, SLOPE,
                               ENC_CH[i], h, w, X.P[i].p);
            X.P[i].bn = BnRef{nullptr, nullptr, nullptr, nullptr, 0, 0.0}; X.P[i].slope = 1.0f;
        }
        x = &X.P[i];
    }
    {   // bottleneck: conv-BN-LReLU three times
        const int h = H >> 5, w = W >> 5;
        if ((rc = run_conv(ctx, s, li++, *x, 0, nullptr, 0, h, w, 0, X.T1[5], batch, false))) return rc;
        AIPT_HIP(ctx, wait_hidden(5));
        if ((rc = run_conv(ctx, s, li++, X.T1[5], 0, &prevX.Hid[5], 0, h, w, 0, X.T2[5], batch, carry))) return rc;
        if ((rc = run_conv(ctx, s, li++, X.T2[5], 0, nullptr, 0, h, w, 0, X.Hid[5], batch, false))) return rc;
        X.Hid[5].slope = SLOPE;
        AIPT_HIP(ctx, hidden_written(5));
    }
    // decoders: cat(prev, skip) -> Upsample x2 -> conv BN LReLU conv BN LReLU
    const Tensor* prev = &X.Hid[5];
    for (int k = 5; k >= 1; k--) {
        const int h = H >> (k - 1), w = W >> (k - 1);
        if ((rc = run_conv(ctx, s, li++, *prev, 1, &X.P[k - 1], 1, h, w, 0, X.D1[k], batch, true))) return rc;
        // dec1.c2 writes the normalised, cropped network output itself (conv3x3_quad); the VALU cross-check path keeps the raw
        // tensor + apply_norm
        const bool last = k == 1 && s->impl != AIPT_DN_IMPL_VALU;
        if ((rc = run_conv(ctx, s, li++, X.D1[k], 0, nullptr, 0, h, w, 0, X.D2[k], batch, false, nullptr, last ? d_out3 : nullptr, out_h, out_w))) return rc;
        prev = &X.D2[k];
    }
    if (s->impl == AIPT_DN_IMPL_VALU) {
        const size_t n = (size_t)out_h * out_w;
        const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
        hipLaunchKernelGGL(apply_norm, dim3(grid, 1), dim3(256), 0, st, X.D2[1].p, X.D2[1].bn, SLOPE, 3, H, W,
                           d_out3, out_h, out_w);
    }
    AIPT_HIP(ctx, hipGetLastError());
    if (timed) {
        AIPT_HIP(ctx, hipEventRecord(s->ev_fence[NSET - 1], st));
        for (int k = 1; k < NSET; k++) AIPT_HIP(ctx, hipStreamWaitEvent(stream_of((a + k) % NSET), s->ev_fence[NSET - 1], 0));
    }
    s->hidden_valid = true;
    if (s->prof_mask) {
        if (s->prof_calls < s->prof_max && s->prof_seen % s->prof_every == 0) s->prof_calls++;
        s->prof_seen++;
    }
    return li == NLAYERS ? AIPT_OK : fail(ctx, AIPT_E_STATE, "aipt_denoise: ran %d layers", li);
}

extern "C" {

int aipt_denoise_profile_begin(aipt_ctx* ctx, uint32_t layer_mask, int max_calls) {
    AIPT_CHECK_CTX(ctx);
    DenoiseState* s = state(ctx);
    if (max_calls < 1 || max_calls > 4096) return fail(ctx, AIPT_E_INVALID, "aipt_denoise_profile_begin: max_calls %d", max_calls);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    free_profile(s);
    s->prof_ev.resize((size_t)max_calls * NLAYERS * 2, nullptr);
    for (int c = 0; c < max_calls; c++)
        for (int l = 0; l < NLAYERS; l++)
            if ((layer_mask >> l) & 1u)
                for (int k = 0; k < 2; k++) AIPT_HIP(ctx, hipEventCreate(&s->prof_ev[((size_t)c * NLAYERS + l) * 2 + k]));
    s->prof_mask = layer_mask & ((1u << NLAYERS) - 1u);
    s->prof_max = max_calls;
    s->prof_calls = 0;
    s->prof_seen = 0;
    return AIPT_OK;
}

int aipt_denoise_profile_stride(aipt_ctx* ctx, int every) {
    AIPT_CHECK_CTX(ctx);
    if (every < 1) return fail(ctx, AIPT_E_INVALID, "aipt_denoise_profile_stride: every %d", every);
    state(ctx)->prof_every = every;
    return AIPT_OK;
}

int aipt_denoise_profile_end(aipt_ctx* ctx, double* sum_ms28, int* calls) {
    AIPT_CHECK_CTX(ctx);
    DenoiseState* s = state(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    if (sum_ms28) {
        for (int l = 0; l < NLAYERS; l++) sum_ms28[l] = 0.0;
        for (int c = 0; c < s->prof_calls; c++)
            for (int l = 0; l < NLAYERS; l++)
                if ((s->prof_mask >> l) & 1u) {
                    float ms = 0;
                    AIPT_HIP(ctx, hipEventElapsedTime(&ms, s->prof_ev[((size_t)c * NLAYERS + l) * 2],
                                                      s->prof_ev[((size_t)c * NLAYERS + l) * 2 + 1]));
                    sum_ms28[l] += ms;
                }
    }
    if (calls) *calls = s->prof_calls;
    std::vector<hipEvent_t> tmp;
    tmp.swap(s->prof_ev);
    for (hipEvent_t e : tmp) if (e) hipEventDestroy(e);
    s->prof_mask = 0; s->prof_max = 0; s->prof_calls = 0;
    return AIPT_OK;
}

int aipt_denoise_layer_info(aipt_ctx* ctx, int layer, char* kernel, size_t kernel_len, int* cin, int* cout,
                            int* height, int* width, double* flops) {
    AIPT_CHECK_CTX(ctx);
    DenoiseState* s = state(ctx);
    if (layer < 0 || layer >= NLAYERS) return fail(ctx, AIPT_E_INVALID, "aipt_denoise_layer_info: layer %d", layer);
    if (!s->have_weights || !s->H) return fail(ctx, AIPT_E_STATE, "aipt_denoise_layer_info: weights + configure first");
    int lvl;
    if (layer < 15) lvl = layer / 3;                      // enc1..5
    else if (layer < 18) lvl = 5;                         // bottleneck
    else lvl = 4 - (layer - 18) / 2;                      // dec5..dec1 run at levels 4..0
    const int h = s->H >> lvl, w = s->W >> lvl;
    if (kernel && kernel_len) { strncpy(kernel, s->kname[layer], kernel_len - 1); kernel[kernel_len - 1] = 0; }
    if (cin) *cin = s->L[layer].cin;
    if (cout) *cout = s->L[layer].cout;
    if (height) *height = h;
    if (width) *width = w;
    if (flops) *flops = 2.0 * 9.0 * s->L[layer].cin * s->L[layer].cout * (double)h * (double)w;
    return AIPT_OK;
}

int aipt_denoise_get_hidden(aipt_ctx* ctx, int level, float* d_dst) {
    AIPT_CHECK_CTX(ctx);
    DenoiseState* s = state(ctx);
    if (!s->H) return fail(ctx, AIPT_E_STATE, "aipt_denoise_get_hidden: not configured");
    if (level < 0 || level > 5 || !d_dst) return fail(ctx, AIPT_E_INVALID, "aipt_denoise_get_hidden: level %d", level);
    const Tensor& t = s->A[s->aset].Hid[level];
    const size_t hw = (size_t)(s->H >> level) * (s->W >> level);
    if (!s->hidden_valid) {
        AIPT_HIP(ctx, hipMemsetAsync(d_dst, 0, sizeof(float) * t.C * hw, ctx->stream));
        return AIPT_OK;
    }
    const int h = s->H >> level, w = s->W >> level;
    const int grid = (int)((hw + 255) / 256 < 64 ? (hw + 255) / 256 : 64);
    hipLaunchKernelGGL(apply_norm, dim3(grid, pad4(t.C) / 4), dim3(256), 0, ctx->stream, t.p, t.bn, t.slope, t.C, h, w, d_dst, h, w);
    AIPT_HIP(ctx, hipGetLastError());
    return AIPT_OK;
}

int aipt_denoise_set_hidden(aipt_ctx* ctx, int level, const float* d_src) {
    AIPT_CHECK_CTX(ctx);
    DenoiseState* s = state(ctx);
    if (!s->H) return fail(ctx, AIPT_E_STATE, "aipt_denoise_set_hidden: not configured");
    if (level < 0 || level > 5 || !d_src) return fail(ctx, AIPT_E_INVALID, "aipt_denoise_set_hidden: level %d", level);
    Tensor& t = s->A[s->aset].Hid[level];
    const size_t hw = (size_t)(s->H >> level) * (s->W >> level);
    if (!s->hidden_valid) {
        // the other levels must read as zeros: raw 0 with identity transform
        for (int l = 0; l < 6; l++) {
            Tensor& o = s->A[s->aset].Hid[l];
            AIPT_HIP(ctx, hipMemsetAsync(o.p, 0, sizeof(float) * pad4(o.C) * (size_t)(s->H >> l) * (s->W >> l), ctx->stream));
            o.bn = BnRef{nullptr, nullptr, nullptr, nullptr, 0, 0.0};
            o.slope = 1.0f;
        }
        s->hidden_valid = true;
    }
    {   // planar [C][h][w] from the caller -> C4
        const size_t n = (size_t)pad4(t.C) * hw;
        const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(planar_to_c4, dim3(grid), dim3(256), 0, ctx->stream, d_src, t.C, hw, t.p);
    }
    t.bn = BnRef{nullptr, nullptr, nullptr, nullptr, 0, 0.0};
    t.slope = 1.0f;      // already normalised: no LReLU on load
    AIPT_HIP(ctx, hipGetLastError());
    return AIPT_OK;
}

}  // extern "C"

#ifdef AIPT_CONV_PHASES
// debug build only (not in include/aiptd.h): key != 0 selects the launches to stamp and zeroes the sums; key == 0 reads them
extern "C" int aipt_debug_conv_phases(aipt_ctx* ctx, unsigned long long* out16, unsigned key) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    if (key) {
        unsigned long long z[16] = {0};
        AIPT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_conv_phase), z, sizeof z));
        AIPT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_conv_key), &key, sizeof key));
    } else if (out16) {
        AIPT_HIP(ctx, hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_conv_phase), 128));
    }
    return AIPT_OK;
}
#endif
