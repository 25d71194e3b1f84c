// a3/a4, round 4: the masked UP-sampling 3x3 modulated conv (model.py:287-300, 385-400) with all FOUR output parities of a position in one workgroup.
// In the parity-composed form (modconv_mx.hip) the outputs (2y + pa, 2x + pb) of input position (y, x) are four 3x3 convolutions of the same
// nine input pixels; what differs is the composed weight — and, in general, the modulation, because it is the OUTPUT pixel's region that modulates.  Where the
// four outputs of every position of a tile carry one region (every map whose region borders run along even output coordinates: the benchmark's cells, the inside
// of any face-sized region) the modulated / f16-split / fp6-converted activation operand of a tap is the same for the four parities.  This kernel prepares it ONCE
// and feeds it to 4 parities x 64 output channels (8 accumulator blocks per wave instead of 4): a quarter of the per-tap VALU work, of the activation reads and of
// the two 32-value fp6 conversions per kernel row, and half the barriers per MFMA (the loop's time is the SUM of those parts and its MFMAs — DESIGN.md section 8,
// "what the masked f16 + fp6 loop's time is made of").  The tile's outputs leave as 8-byte stores of (pb = 0, pb = 1) pairs.
// A tile with a position of mixed regions is computed, in the same launch, as the composed kernel computes it (its tile is a device function, modconv_mx_tile.h): every
// workgroup decides for its tile (quad_uniform_tile, modconv_sb.h) and takes one of the two roles — see the kernel at the end of the file.  Same products in the same
// order either way: the layer's output is bit-identical to e4s_region_modconv3x3_mx's on every map, and a map without a single qualifying tile costs what it cost before.
// f16 + 2 x MX-fp6 arithmetic only (the split-bf16 re-run of an overflowed pass uses the composed kernel).
// Weights (e4s_modconv_prep_weights_mx4): one ROW SLOT per (chunk, 64-co tile, kernel row) = the four parities' [w1 f16 [tap 3][half 2][co 64] x 16 B |
// fp6 codes first 16 B [term 2][half 2][co 64] | last 8 B [term][half][co] | E8M0 scales [half 2][co 64] x 4 B, padded to 1 KB] = 4 x 13 312 B, DMA'd as 52 pieces of
// 1 KB into a ring of TWO slots (the request for row g + 1 goes out behind row g's first MFMAs, into the slot row g - 1 left at the last barrier).
#include "modconv_mx_tile.h"

namespace {

constexpr int Q_TN = 64;                                       // output channels per workgroup
constexpr int Q_W1B = 3 * 2 * Q_TN * 16;                       // 6 144
constexpr int Q_F6LO = 2 * 2 * Q_TN * 16;                      // 4 096
constexpr int Q_F6HI = 2 * 2 * Q_TN * 8;                       // 2 048
constexpr int Q_SCB = 1024;                                    // 2 * 64 * 4 = 512 B of scales, padded to a DMA piece
constexpr int Q_PARB = Q_W1B + Q_F6LO + Q_F6HI + Q_SCB;        // 13 312 per parity
constexpr int Q_ROWB = 4 * Q_PARB;                             // 53 248
constexpr int Q_NPIECE = Q_ROWB / 1024;                        // 52
using CQ = SbCfg<2, 1, 1, 8, 5>;                               // 64 co x (32 x 8) positions, 512 threads; wave w = tile row w
constexpr int Q_PSTRIDE = 352;
constexpr int Q_PATCHB = 4 * Q_PSTRIDE * 16;                   // 22 528: fp32 [16-B slot 4][pixel 352], as modconv_mx.hip
constexpr int Q_SSB = E4S_MAX_REGIONS * CKS * 4;               // 1 024
constexpr int Q_PATCH0 = 2 * Q_ROWB, Q_SS0 = Q_PATCH0 + 2 * Q_PATCHB, Q_LDS = Q_SS0 + 2 * Q_SSB;      // 153 600
static_assert(Q_LDS + 16 <= 160 * 1024 && (E4S_MAX_REGIONS + 1) * Q_TN * 4 <= Q_ROWB, "LDS plan");

// "nothing moves across": the scheduling fence alone does not keep instruction selection from hoisting later LDS reads (a row's 16 fp6 operand tuples at once: 100
// registers, accumulators spilled); the empty asm with a memory clobber does
#define Q_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// ============================================================================ weight preparation
// One thread per (chunk, co tile, row, par, half, co): the lane's 24 values of that kernel row — value for value e4s_modconv_prep_weights_mx's f16 + fp6 form.
__global__ __launch_bounds__(256) void prep_weights_mx4_kernel(unsigned char* __restrict__ dst, const float* __restrict__ weight, const float* __restrict__ blur,
                                                               int cout, int cin, float scale) {
    const int nchunk = (cin + CKS - 1) / CKS;
    const int ntile = (cout + Q_TN - 1) / Q_TN;
    const int64_t total = (int64_t)nchunk * ntile * 3 * 4 * 2 * Q_TN;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int n = (int)(r % Q_TN); r /= Q_TN;
        const int half = (int)(r & 1); r >>= 1;
        const int par = (int)(r & 3); r >>= 2;
        const int row = (int)(r % 3); r /= 3;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int co = tile * Q_TN + n;
        unsigned char* slot = dst + (((size_t)chunk * ntile + tile) * 3 + row) * Q_ROWB + (size_t)par * Q_PARB;
        float v[24];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ci = chunk * CKS + half * 8 + e;
                float x = 0.f;
                if (ci < cin && co < cout) { x = sb_weff(weight, blur, cin, co, ci, row * 3 + t, par, 1); x *= scale; }
                v[t * 8 + e] = x;
            }
        uint4* w1p = reinterpret_cast<uint4*>(slot);
        uint4* f6lo = reinterpret_cast<uint4*>(slot + Q_W1B);
        uint2* f6hi = reinterpret_cast<uint2*>(slot + Q_W1B + Q_F6LO);
        unsigned* scp = reinterpret_cast<unsigned*>(slot + Q_W1B + Q_F6LO + Q_F6HI);
        u32x16 q1, q2;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float a = v[2 * j], b = v[2 * j + 1];
            const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
            const float ra = (a - (float)h[0]) * 4096.f, rb = (b - (float)h[1]) * 4096.f;
            q1[j] = __builtin_bit_cast(unsigned, h);
            q2[j] = pack_f16_rne(ra, rb);
            m1 = fmaxf(m1, fmaxf(fabsf((float)h[0]), fabsf((float)h[1])));
            m2 = fmaxf(m2, fmaxf(fabsf(ra), fabsf(rb)));
        }
#pragma unroll
        for (int j = 12; j < 16; ++j) { q1[j] = 0u; q2[j] = 0u; }
#pragma unroll
        for (int t = 0; t < 3; ++t) w1p[(t * 2 + half) * Q_TN + n] = make_uint4(q1[4 * t], q1[4 * t + 1], q1[4 * t + 2], q1[4 * t + 3]);
        auto expo = [](float m) { const unsigned ex = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu; return ex > 3u ? ex - 2u : 1u; };
        const unsigned e1 = expo(m1), e2 = expo(m2);
        const u32x6 c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
        const u32x6 c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
        f6lo[(0 * 2 + half) * Q_TN + n] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
        f6hi[(0 * 2 + half) * Q_TN + n] = make_uint2(c1[4], c1[5]);
        f6lo[(1 * 2 + half) * Q_TN + n] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
        f6hi[(1 * 2 + half) * Q_TN + n] = make_uint2(c2[4], c2[5]);
        const unsigned e2s = e2 > 12u ? e2 - 12u : 0u;
        scp[half * Q_TN + n] = e1 | (e2s << 8);
        if (half == 0) scp[2 * Q_TN + n] = 0u;                 // (the padding of the scale piece: never read, written so that the buffer is fully defined)
        if (half == 1) scp[3 * Q_TN + n] = 0u;
    }
}

// ============================================================================ the four-parity tile
// One workgroup tile: the four parities x 64 output channels (64 cotile .. + 63) of the 32 x 8 positions of `tile`, image b; `c_own` = the raw label of this lane's position
// (quad_uniform_tile: all four of its outputs carry it).
__device__ __forceinline__ void q_tile_body(const SbParams& p, unsigned char* lds_raw, const int tile, const int cotile, const int b, const int c_own) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l5 = lane & 31, khalf = lane >> 5;
    const int y0 = (tile / p.tiles_x) * CQ::TH, x0 = (tile % p.tiles_x) * CQ::TW;
    const int co0 = cotile * Q_TN;
    const int hw = p.h * p.w;
    const int ho = 2 * p.h, wo = 2 * p.w;
    const int nchunk = (p.cin + CKS - 1) / CKS;
    const int ncot = (p.cout + Q_TN - 1) / Q_TN;
    const int cls = c_own < p.nreg ? c_own : -1;
    const int xoff = wave * CQ::PW + l5;

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;

    const float* xb = p.x + (size_t)b * p.cin * hw;
    const float* sb = p.s + (size_t)b * p.nreg * p.cin;
    // The next chunk's activations travel in two halves of 8 channels (8 registers instead of 16: this kernel holds 128 accumulators): half 0 is requested in row 0
    // and written to the other patch buffer at the start of row 1, half 1 requested there and written at the start of row 2 (every row ends with vmcnt(0)).
    float xr[8];
    float sr = 0.f;
    const int ppy = tid / CQ::PW, ppx = tid - ppy * CQ::PW;
    const int pgy = y0 - 1 + ppy, pgx = x0 - 1 + ppx;
    const bool p_in = tid < CQ::PATCH && pgy >= 0 && pgy < p.h && pgx >= 0 && pgx < p.w;
    const int goffs = p_in ? pgy * p.w + pgx : 0;
    const int s_r = tid / CKS < p.nreg ? tid / CKS : p.nreg - 1, s_c = tid % CKS;
    auto load_x = [&](int chunk, int hf) __attribute__((always_inline)) {       // (unconditional loads from clamped addresses: see modconv_mx.hip)
        const int ci0 = chunk * CKS;
        const int cmax = p.cin - 1 - ci0;
        if (wave < (CQ::PATCH + 63) / 64) {
#pragma unroll
            for (int c = 0; c < 8; ++c) xr[c] = xb[(size_t)(ci0 + (8 * hf + c < cmax ? 8 * hf + c : cmax)) * hw + goffs];
        }
        if (hf == 0) sr = sb[(size_t)s_r * p.cin + ci0 + (s_c < cmax ? s_c : cmax)];
    };
    auto store_x = [&](int buf, int chunk, int hf) __attribute__((always_inline)) {
        float4* xf4 = reinterpret_cast<float4*>(lds_raw + Q_PATCH0 + buf * Q_PATCHB);
        if (tid < CQ::PATCH) {
#pragma unroll
            for (int c = 0; c < 8; ++c) xr[c] = p_in ? xr[c] : 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) xf4[(2 * hf + k) * Q_PSTRIDE + tid] = make_float4(xr[4 * k], xr[4 * k + 1], xr[4 * k + 2], xr[4 * k + 3]);
        }
        if (hf == 0 && tid < E4S_MAX_REGIONS * CKS)
            reinterpret_cast<float*>(lds_raw + Q_SS0 + buf * Q_SSB)[tid] = (tid / CKS < p.nreg && chunk * CKS + s_c < p.cin) ? sr : 0.f;
    };
    auto dma_row = [&](int chunk, int row, int slot) __attribute__((always_inline)) {
        const unsigned char* src = p.wmx + ((size_t)(chunk * ncot + cotile) * 3 + row) * Q_ROWB;
#pragma unroll
        for (int k = 0; k < (Q_NPIECE + 7) / 8; ++k) {
            const int piece = wave + 8 * k;
            if (piece < Q_NPIECE) dma16_asm(src + piece * 1024, (unsigned)(lane * 16), (unsigned)(slot * Q_ROWB + piece * 1024));
        }
    };

    dma_row(0, 0, 0);
    load_x(0, 0);
    store_x(0, 0, 0);
    load_x(0, 1);
    store_x(0, 0, 1);
    E4S_WAIT_VM(0);
    E4S_LDS_BARRIER();

    bool ovf = false;
#pragma unroll 1
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunk;
        if (more) load_x(chunk + 1, 0);
        const float4* xf4 = reinterpret_cast<const float4*>(lds_raw + Q_PATCH0 + cur * Q_PATCHB);
        const float* ss = reinterpret_cast<const float*>(lds_raw + Q_SS0 + cur * Q_SSB);
        float sv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[e] = cls >= 0 ? ss[cls * CKS + khalf * 8 + e] : 0.f;

#pragma unroll
        for (int row = 0; row < 3; ++row) {
            if (row == 1 && more) { store_x(cur ^ 1, chunk + 1, 0); load_x(chunk + 1, 1); }
            if (row == 2 && more) store_x(cur ^ 1, chunk + 1, 1);
            const int slot = __builtin_amdgcn_readfirstlane((chunk + row) & 1);            // global row 3 chunk + row, two slots
            const unsigned char* slotp = lds_raw + slot * Q_ROWB;
            const uint4* w1half = reinterpret_cast<const uint4*>(slotp) + khalf * Q_TN + l5;        // + par * (Q_PARB / 16) + t * 2 * Q_TN + i * 32
            u32x16 v1, v2;
            float amax = 0.f;
            // Register budget: 128 accumulators of the 256 a wave may hold.  So the row runs in three passes that do not overlap inside the wave (the SIMD's other wave
            // fills the gaps): (A) the three taps' operand preparation — nothing but the patch fragments, the modulation values and the two 16-register tuples live;
            // (B) the 24 f16 MFMAs, weight fragments four at a time; (C) the two conversions and the 16 fp6 MFMAs, operands read right in front of each.
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int e = xoff + row * CQ::PW + t;
                const float4 xa = xf4[(2 * khalf) * Q_PSTRIDE + e], xq = xf4[(2 * khalf + 1) * Q_PSTRIDE + e];
                const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xq.x, xq.y, xq.z, xq.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = xv[2 * j] * sv[2 * j], bq = xv[2 * j + 1] * sv[2 * j + 1];
                    const f16x2 a1 = __builtin_convertvector((f32x2){a, bq}, f16x2);
                    v1[t * 4 + j] = __builtin_bit_cast(unsigned, a1);
                    v2[t * 4 + j] = resid_pair_f16(xv[2 * j], sv[2 * j], xv[2 * j + 1], sv[2 * j + 1], v1[t * 4 + j]);
                    amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(bq)));
                }
            }
            Q_FENCE();
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const uint4 b1 = make_uint4(v1[t * 4], v1[t * 4 + 1], v1[t * 4 + 2], v1[t * 4 + 3]);
#pragma unroll
                for (int ap = 0; ap < 2; ++ap) {
                    uint4 wv[2][2];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int i = 0; i < 2; ++i) wv[a][i] = w1half[(2 * ap + a) * (Q_PARB / 16) + t * 2 * Q_TN + i * 32];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[2 * ap + a][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv[a][i]), __builtin_bit_cast(f16x8, b1), acc[2 * ap + a][i], 0, 0, 0);
                    if (t == 0 && ap == 0) {                    // the next row's weights, into the slot the previous row left at the last barrier
                        if (row < 2) dma_row(chunk, row + 1, slot ^ 1);
                        else if (more) dma_row(chunk + 1, 0, slot ^ 1);
                    }
                    Q_FENCE();
                }
            }
            const unsigned ex = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
            ovf |= amax >= 65520.f;
            const unsigned e1 = ex > 3u ? ex - 2u : 1u, e2 = ex > 14u ? ex - 13u : 1u;
            const u32x6 p1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, v1), __builtin_bit_cast(float, e1 << 23));
            const u32x6 p2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, v2), __builtin_bit_cast(float, e2 << 23));
            const i32x8 bx1 = mx_op6(p1), bx2 = mx_op6(p2);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint4* f6lo = reinterpret_cast<const uint4*>(slotp + a * Q_PARB + Q_W1B) + khalf * Q_TN + l5;                 // + term * 2 * TN + i * 32
                const uint2* f6hi = reinterpret_cast<const uint2*>(slotp + a * Q_PARB + Q_W1B + Q_F6LO) + khalf * Q_TN + l5;
                const unsigned* wsc = reinterpret_cast<const unsigned*>(slotp + a * Q_PARB + Q_W1B + Q_F6LO + Q_F6HI) + khalf * Q_TN + l5;
                int sc[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) sc[i] = (int)wsc[i * 32];
#pragma unroll
                for (int i = 0; i < 2; ++i) {      // fp6(w - w1) x fp6(a1)     (operands read right in front of their MFMA, as modconv_mx.hip does: 6 registers each)
                    const uint4 lo = f6lo[2 * Q_TN + i * 32];
                    const uint2 hi = f6hi[2 * Q_TN + i * 32];
                    acc[a][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(lo, hi), bx1, acc[a][i], 2, 2, 1, sc[i], 0, (int)e1);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {      // fp6(w1) x fp6(a - a1)
                    const uint4 lo = f6lo[i * 32];
                    const uint2 hi = f6hi[i * 32];
                    acc[a][i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(lo, hi), bx2, acc[a][i], 2, 2, 0, sc[i], 0, (int)e2);
                }
                Q_FENCE();
            }
            // (the accumulators are pinned here: LLVM otherwise sinks a row's fp6 MFMAs — register-only instructions — behind the barrier into the next row and
            // carries their 16 operand tuples across it in scratch)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(acc[a][i]));
            E4S_WAIT_VM(0);
            E4S_LDS_BARRIER();
        }
    }
    if (p.flags && __builtin_amdgcn_ballot_w64(ovf) != 0 && lane == 0) { atomicOr(p.flags, 1); atomicAdd(p.flags + 1, 1); }

    // ---- epilogue: demodulation of the tile's positions' region, noise, bias, leaky ReLU; (pb = 0, pb = 1) pairs as 8-byte stores.  Every global load before the first store.
    float* dt = reinterpret_cast<float*>(lds_raw);          // [MAX_REG][64] over the ring (every wave is past the loop's last barrier)
    float* bt = dt + E4S_MAX_REGIONS * Q_TN;                // [64]
    for (int v = tid; v < E4S_MAX_REGIONS * Q_TN; v += 512) {
        const int r = v / Q_TN, n = v % Q_TN;
        dt[v] = (r < p.nreg && co0 + n < p.cout) ? (p.d ? p.d[((size_t)b * p.nreg + r) * p.cout + co0 + n] : 1.f) : 0.f;
    }
    if (tid < Q_TN) bt[tid] = (p.act_bias && co0 + tid < p.cout) ? p.act_bias[co0 + tid] : 0.f;
    const int y = y0 + wave, x = x0 + l5;
    const bool pix_ok = y < p.h && x < p.w;
    const float nw = p.noise ? p.noise_weight[0] : 0.f;
    float nz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (p.noise && pix_ok) {
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const float2 q = *reinterpret_cast<const float2*>(p.noise + (size_t)b * p.noise_bstride + (size_t)(2 * y + pa) * wo + 2 * x);
            nz[pa][0] = nw * q.x; nz[pa][1] = nw * q.y;
        }
    }
    __syncthreads();
    const float* drow = dt + (cls >= 0 ? cls : 0) * Q_TN;
    const float dz = cls >= 0 ? 1.f : 0.f;
    if (!pix_ok) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const int co = co0 + n;
            if (co >= p.cout) continue;
            float* oc = p.out + ((size_t)b * p.cout + co) * ho * wo + (size_t)(2 * y) * wo + 2 * x;
#pragma unroll
            for (int pa = 0; pa < 2; ++pa) {
                float v0 = acc[2 * pa][i][r] * drow[n] * dz + nz[pa][0] + bt[n];
                float v1_ = acc[2 * pa + 1][i][r] * drow[n] * dz + nz[pa][1] + bt[n];
                if (p.act) {
                    v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * 1.41421356237309515f;
                    v1_ = (v1_ > 0.f ? v1_ : v1_ * 0.2f) * 1.41421356237309515f;
                }
                *reinterpret_cast<float2*>(oc + (size_t)pa * wo) = make_float2(v0, v1_);
            }
        }
}

// ============================================================================ the kernel
// gridDim.x = 2 * PA workgroups, PA = (tiles x images) x G with G = cout / 64 (cout % 128 == 0).  Workgroup lin belongs to phase lin / PA and, inside the phase, to
// (item, c) — item = a tile of an image, c = 0 .. G - 1.  Where the item's positions all have four outputs of one region, phase-0 workgroup (item, c) computes its four
// parities for output channels 64 c .. + 63 and the phase-1 workgroup exits; otherwise each of the two runs one tile of the composed kernel (modconv_mx_tile.h): output
// channels 128 (c / 2) .. + 127 at parity 2 phase + (c & 1) — the eight (parity, 128-channel) tiles of the item between them.  A map of aligned cells is then ONE round of
// working workgroups followed by workgroups that leave after reading four labels; a map with no such tile is the composed kernel's launch, workgroup for workgroup.
// XCD affinity (workgroup lin runs on XCD lin % 8, each with its own L2): for G in {1, 2, 4, 8} the 8 / G XCDs of a group share c — one 64-channel weight set in the
// four-parity role, (a half of) one 128-channel set in the composed role — as modconv_mx.hip's launcher arranges it for its own grid.
__global__ __launch_bounds__(512, 2) void region_upconv_mx4_kernel(const SbParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const unsigned PA = gridDim.x >> 1;
    const unsigned phase = blockIdx.x >= PA ? 1u : 0u;
    const unsigned l = blockIdx.x - phase * PA;
    const unsigned G = (unsigned)(p.cout / Q_TN);
    unsigned c, item;
    if ((G == 1u || G == 2u || G == 4u || G == 8u) && (PA & 7u) == 0u) {
        const unsigned per = 8u / G, xcd = l & 7u, q = l >> 3;
        c = xcd / per;
        item = q * per + xcd % per;
    } else {
        c = l % G;
        item = l / G;
    }
    const int ntile = p.tiles_x * p.tiles_y;
    const int tile = (int)(item % (unsigned)ntile), b = (int)(item / (unsigned)ntile);
    const int y0 = (tile / p.tiles_x) * CQ::TH, x0 = (tile % p.tiles_x) * CQ::TW;
    int c_own;
    const bool uni = quad_uniform_tile(p, b, y0, x0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 31), c_own, reinterpret_cast<volatile int*>(lds_raw + Q_LDS));
    if (uni) {
        if (phase) return;
        q_tile_body(p, lds_raw, tile, (int)c, b, c_own);
    } else {
        SbParams pc = p;                                    // (the composed kernel's weights travel in the unused split-bf16 slab pointer)
        pc.wmx = reinterpret_cast<const unsigned char*>(p.whi);
        mx_tile_body<1, false, false, false>(pc, lds_raw, tile, (int)(2u * phase + (c & 1u)), 0, (int)(c >> 1), b);
    }
}

}  // namespace

extern "C" int e4s_modconv_mx4_weight_bytes(int cout, int cin, int64_t* bytes) {
    E4S_REQUIRE(bytes && cout >= 1 && cin >= 1, "modconv_mx4_weight_bytes: bad arguments");
    *bytes = (int64_t)cdiv(cin, CKS) * cdiv(cout, Q_TN) * 3 * Q_ROWB;
    return 0;
}

extern "C" int e4s_modconv_prep_weights_mx4(void* dst, const float* weight, const float* blur, int cout, int cin, void* stream) {
    E4S_REQUIRE(dst && weight && blur, "modconv_prep_weights_mx4: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= 1, "modconv_prep_weights_mx4: bad size");
    E4S_REQUIRE(((uintptr_t)dst & 15) == 0, "modconv_prep_weights_mx4: the destination must be 16-byte aligned");
    const int64_t total = (int64_t)cdiv(cin, CKS) * cdiv(cout, Q_TN) * 3 * 4 * 2 * Q_TN;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(prep_weights_mx4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst, weight, blur, cout, cin, 1.0f / sqrtf((float)cin * 9.f));
    return check_launch("modconv_prep_weights_mx4");
}

extern "C" int e4s_region_upconv_mx4(float* out, const float* x, const void* wmx4, const void* wmx, int* flags, const float* s, const float* d, const uint8_t* labels,
                                     int lh, int lw, const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act, int bs, int cin,
                                     int cout, int h, int w, int nreg, void* stream) {
    E4S_REQUIRE(out && x && wmx4 && wmx && s && labels, "region_upconv_mx4: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= CKS && cin % CKS == 0 && cout >= 128 && cout % 128 == 0 && h >= 1 && w >= 32,
                "region_upconv_mx4: bad size (cin %% 16 == 0, cout %% 128 == 0, width >= 32)");
    E4S_REQUIRE(nreg >= 1 && nreg <= E4S_MAX_REGIONS && lh >= 1 && lw >= 1, "region_upconv_mx4: bad region map");
    E4S_REQUIRE(!noise || (noise_weight && (noise_bs == 1 || noise_bs == bs)), "region_upconv_mx4: noise needs its weight and batch 1 or bs");
    E4S_REQUIRE((((uintptr_t)wmx4 | (uintptr_t)wmx | (uintptr_t)out) & 15) == 0 && (!noise || ((uintptr_t)noise & 7) == 0),
                "region_upconv_mx4: weights / output must be 16-byte aligned, noise 8-byte");
    if (bs == 0) return 0;
    SbParams p;
    memset(&p, 0, sizeof(p));
    p.out = out; p.x = x; p.wmx = reinterpret_cast<const unsigned char*>(wmx4); p.whi = reinterpret_cast<const uint4*>(wmx); p.flags = flags; p.s = s; p.d = d;
    p.labels = labels; p.lh = lh; p.lw = lw; p.noise = noise; p.noise_weight = noise_weight; p.act_bias = act_bias; p.act = act;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.nreg = nreg; p.up = 1;
    p.lscale_y = (float)lh / (float)(2 * h);
    p.lscale_x = (float)lw / (float)(2 * w);
    p.noise_bstride = (noise && noise_bs > 1) ? 4 * h * w : 0;
    p.tiles_x = cdiv(w, CQ::TW);
    p.tiles_y = cdiv(h, CQ::TH);
    p.ksplit = 1;
    p.chunks_per = cdiv(cin, CKS);
    const int64_t pa = (int64_t)p.tiles_x * p.tiles_y * bs * (cout / Q_TN);
    E4S_REQUIRE(2 * pa <= 0x7fffffff, "region_upconv_mx4: launch too large");
    constexpr int lds = Q_LDS + 16;                        // (+ the tile predicate's word; the composed tile's plan is smaller)
    static_assert(MxLds<1>::BYTES <= Q_LDS, "the composed tile runs inside this kernel's LDS allocation");
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&region_upconv_mx4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return fail((int)attr, "region_upconv_mx4: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL(region_upconv_mx4_kernel, dim3((unsigned)(2 * pa)), dim3(512), lds, (hipStream_t)stream, p);
    return check_launch("region_upconv_mx4");
}
