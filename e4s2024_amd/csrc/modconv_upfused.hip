// a3/a4, single-region (unmasked) up layer in ONE launch: stride-2 transposed 3x3 conv on split-bf16 MFMA at 1x its algorithmic
// MACs -> pre-blur tile in LDS -> 4x4 blur + demodulation + noise + bias + leaky-relu, written once.
// Reference: ModulatedConv2d.forward upsample branch (models/stylegan2/model.py:287-301) + NoiseInjection/FusedLeakyReLU of
// StyledConv.forward (:417-421).  Same arithmetic as e4s_modconv_tconv_sb + e4s_blur_epilogue (modconv_sb.hip) without the
// [bs,cout,2h+1,2w+1] round trip through HBM (2 x 537 MB per step at 1024^2, batch 4).
//
// Geometry.  Lanes sit on a 16 x 16 tile of POSITIONS (a,b), a in [P0y, P0y+16): tap (ky,kx) adds W[ky][kx] * x[a-(ky>>1)][b-(kx>>1)]
// to the pre-blur pixel z[2a+(ky&1)][2b+(kx&1)], so a tile owns z rows [2*P0y, 2*P0y+32) in four parity accumulators.  Output row
// oy needs z rows oy-1..oy+2 (upfirdn2d pad (1,1)); with P0y = 14*t - 1 the tile yields the 28 output rows [28t, 28t+28)
// from its own z rows only, so neighbouring tiles overlap by 2 positions (1.31x the MACs, nothing re-read from HBM but x's halo).
// Positions outside [0,h] read zero x and give z = 0, which is exactly the blur's zero padding.
#include <stdlib.h>

#include "common.h"
#include "sb_common.h"

using namespace e4s;

namespace {

struct UpFusedParams {
    float* out;
    const float* x;
    const uint4* whi;
    const uint4* wlo;
    const float* s;
    const float* d;
    const float* blur;
    const float* noise;
    const float* noise_weight;
    const float* act_bias;
    int noise_bstride;
    int act;
    int bs, cin, cout, h, w;
    int tiles_x, tiles_y;
    const float* s_next;     // [bs][cout] (OSP): modulation of the NEXT layer, applied before the bf16 split of the output
    int64_t plane_in, plane_out;   // uint4 per split plane (XSP / OSP)
    const float* zeros;      // >= 64 zero bytes (DMA source of an absent noise / bias operand)
    int exp;                 // tuning experiments of the -DE4S_PHASE_PROF build (E4S_UF_EXP): 1 = no output stores, 2 = no blur, 4 = no z-tile writes
};
// channel-blocked activations ([bs, c/8, h, w, 8]) are compile-time variants: XN = input, ON = output (E4S_X_NHWC / E4S_OUT_NHWC in `act`);
// XSP / OSP = split planes (modconv_chain.hip: [hi|lo][bs][c/8][h][w][8 x bf16], pre-modulated and pre-split by the producer): an XSP input
// is copied into LDS as it is (no multiply, no split), an OSP output is multiplied by the next layer's modulation and split before it is written

constexpr int UF_T = 16;                    // positions per tile side
constexpr int UF_STEP = UF_T - 2;           // 14 new positions per tile
constexpr int UF_OUT = 2 * UF_STEP;         // 28 output rows / columns per tile
constexpr int UF_PW = UF_T + 1;             // x patch side (positions read x[a-1], x[a])
constexpr int UF_PATCH = UF_PW * UF_PW;     // 289
constexpr int UF_NT = 512;                  // 8 waves: wave v owns position rows 2v, 2v+1
constexpr int UF_ZS = 34;                   // row stride of the pre-blur tile in LDS (32 + 2: float2-aligned, spreads banks)
constexpr int UF_ZCO = 8;                   // output channels blurred per LDS pass
constexpr int UF_ZCS_NHWC = 32 * UF_ZS + 8; // channel stride of the pre-blur tile for channels-last output: +8 floats so that 8 channels x 8 columns hit 64 banks

E4S_PROF_DECL(g_prof_up)

template <int CB>
struct UfCfg {
    static constexpr int TN = CB * 32;
    static constexpr int W4 = 2 * 9 * 2 * TN;                 // uint4 per chunk: [hi/lo][tap][half][TN]
    static constexpr int WPT = (W4 + UF_NT - 1) / UF_NT;
    static constexpr int MAIN_BYTES = W4 * 16 + UF_PATCH * 64;   // weights + x hi/lo planes
    static constexpr int ZT_BYTES = UF_ZCO * UF_ZCS_NHWC * 4;
    static constexpr int BODY = MAIN_BYTES > ZT_BYTES ? MAIN_BYTES : ZT_BYTES;
    static constexpr int EP_FLOATS = 3 * TN + UF_OUT * UF_OUT;   // epilogue operands fetched at kernel start: d, bias, s_next, noise_weight * noise tile
    static constexpr int LDS_BYTES = BODY + EP_FLOATS * 4;
};

template <int CB, int MINW, bool XN = false, bool ON = false, bool XSP = false, bool OSP = false>
__global__ __launch_bounds__(UF_NT, MINW) void up_fused_sb_kernel(const UpFusedParams p) {
    using C = UfCfg<CB>;
    static_assert(!(XSP && XN) && !(OSP && !ON), "split-plane input excludes the fp32 blocked one; split-plane output uses the blocked item mapping");
    constexpr int UF_ZCS = ON ? UF_ZCS_NHWC : 32 * UF_ZS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* wsm = reinterpret_cast<uint4*>(lds_raw);                       // [2][9][2][TN]
    uint4* xh4 = reinterpret_cast<uint4*>(lds_raw + C::W4 * 16);          // [PATCH][2] uint4 = 16 bf16 (hi), halves swizzled
    uint4* xl4 = xh4 + 2 * UF_PATCH;                                      // lo plane
    float* ep_d = reinterpret_cast<float*>(lds_raw + C::BODY);            // [TN] demodulation
    float* ep_b = ep_d + C::TN;                                           // [TN] activation bias
    float* ep_s = ep_b + C::TN;                                           // [TN] next layer's modulation (OSP)
    float* ep_n = ep_s + C::TN;                                           // [28][28] noise_weight * noise of this tile's outputs

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    E4S_PROF_MARK(g_prof_up, 0);

    const int tyt = blockIdx.x / p.tiles_x, txt = blockIdx.x - tyt * p.tiles_x;
    const int p0y = tyt * UF_STEP - 1, p0x = txt * UF_STEP - 1;
    const int co0 = blockIdx.y * C::TN;
    const int b = blockIdx.z;
    const int hw = p.h * p.w;
    const int nchunk = (p.cin + CKS - 1) / CKS;

    // staging element of this thread (one patch pixel, 16 channels per chunk)
    const int se_y = tid / UF_PW, se_x = tid - se_y * UF_PW;
    const int sgy = p0y - 1 + se_y, sgx = p0x - 1 + se_x;
    const bool s_in = tid < UF_PATCH && sgy >= 0 && sgy < p.h && sgx >= 0 && sgx < p.w;
    const int sgoff = s_in ? sgy * p.w + sgx : 0;
    const float* xb = p.x + (size_t)b * p.cin * hw;
    const float* sb = p.s + (size_t)b * p.cin;

    const int pty = 2 * wave + (l5 >> 4), ptx = l5 & 15;   // this lane's position inside the tile
    const int xoff = pty * UF_PW + ptx;

    f32x16 accs[4][CB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[a][i][r] = 0.f;

    float xr[CKS];            // (XSP: the 16 registers hold the chunk's four 16-byte fragments hi/half0, hi/half1, lo/half0, lo/half1 as bits)
    unsigned wr[C::WPT][4];   // scalar components (a uint4 array would be placed in scratch)
    const ptrdiff_t wdelta = p.wlo - p.whi;
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const int ci0 = chunk * CKS;
        const int cmax = p.cin - 1 - ci0;
        if constexpr (XSP) {  // split planes: the chunk's fragments of this thread's patch pixel, 4 x 16 bytes, kept as bits
            const uint4* xq = reinterpret_cast<const uint4*>(p.x);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const uint4 v = xq[(size_t)(f >> 1) * p.plane_in + ((size_t)(b * (p.cin >> 3) + (ci0 >> 3) + (f & 1)) * hw + sgoff)];
                xr[4 * f] = __builtin_bit_cast(float, v.x); xr[4 * f + 1] = __builtin_bit_cast(float, v.y);
                xr[4 * f + 2] = __builtin_bit_cast(float, v.z); xr[4 * f + 3] = __builtin_bit_cast(float, v.w);
            }
        } else if constexpr (XN) {   // channel-blocked input [cin/8][h][w][8]: this thread's patch pixel, the chunk's two 8-channel blocks, 4 x 16 bytes
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)(ci0 / 8 + blk) * hw + sgoff) * 8 + 4 * hf);
                    xr[8 * blk + 4 * hf] = v.x; xr[8 * blk + 4 * hf + 1] = v.y; xr[8 * blk + 4 * hf + 2] = v.z; xr[8 * blk + 4 * hf + 3] = v.w;
                }
        } else {
#pragma unroll
            for (int c = 0; c < CKS; ++c) xr[c] = xb[(size_t)(ci0 + (c < cmax ? c : cmax)) * hw + sgoff];
        }
        const size_t wbase = (size_t)chunk * 18 * p.cout;   // uint4 units: [tap][half][cout]
#pragma unroll
        for (int v = 0; v < C::WPT; ++v) {
            int idx = tid + v * UF_NT;
            idx = idx < C::W4 ? idx : C::W4 - 1;
            const int hl = idx / (18 * C::TN);
            const int rem = idx - hl * 18 * C::TN;
            const int th = rem / C::TN, n = rem - th * C::TN;
            const int co = (co0 + n < p.cout) ? co0 + n : p.cout - 1;
            const uint4 t4 = p.whi[(ptrdiff_t)hl * wdelta + (ptrdiff_t)(wbase + (size_t)th * p.cout + co)];
            wr[v][0] = t4.x; wr[v][1] = t4.y; wr[v][2] = t4.z; wr[v][3] = t4.w;
        }
    };
    auto store_chunk = [&](int chunk) __attribute__((always_inline)) {
        if (tid < UF_PATCH) {
            unsigned hi[8], lo[8];
            if constexpr (XSP) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    hi[c] = s_in ? __builtin_bit_cast(unsigned, xr[c]) : 0u;
                    lo[c] = s_in ? __builtin_bit_cast(unsigned, xr[8 + c]) : 0u;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int c0 = chunk * CKS + 2 * c;
                    const float s0 = c0 < p.cin ? sb[c0] : 0.f, s1 = c0 + 1 < p.cin ? sb[c0 + 1] : 0.f;   // wave-uniform
                    split2(s_in ? xr[2 * c] * s0 : 0.f, s_in ? xr[2 * c + 1] * s1 : 0.f, hi[c], lo[c]);
                }
            }
            const int sw = (tid >> 3) & 1;
            xh4[tid * 2 + (0 ^ sw)] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            xh4[tid * 2 + (1 ^ sw)] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
            xl4[tid * 2 + (0 ^ sw)] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            xl4[tid * 2 + (1 ^ sw)] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
        }
#pragma unroll
        for (int v = 0; v < C::WPT; ++v) {
            const int idx = tid + v * UF_NT;
            if (idx < C::W4) wsm[idx] = make_uint4(wr[v][0], wr[v][1], wr[v][2], wr[v][3]);
        }
    };

    // Epilogue operands: fetched now, next to the first chunk's loads, and parked in LDS when that chunk is staged — the epilogue then
    // has no global loads at all (they used to cost ~10 us per workgroup there, and a load behind a store drains the store first).
    const int ho = 2 * p.h, wo = 2 * p.w;
    float ep_r[4] = {1.f, 0.f, 0.f, 0.f};
    float ep_sn = 0.f;
    {
        const int co = co0 + tid;
        if (tid < C::TN && co < p.cout) {
            if (p.d) ep_r[0] = p.d[(size_t)b * p.cout + co];
            if (p.act_bias) ep_r[1] = p.act_bias[co];
            if constexpr (OSP) ep_sn = p.s_next[(size_t)b * p.cout + co];
        }
        if (p.noise) {
            const float nw0 = p.noise_weight[0];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = tid + k * UF_NT;
                const int ny = tyt * UF_OUT + e / UF_OUT, nx = txt * UF_OUT + e % UF_OUT;
                if (e < UF_OUT * UF_OUT && ny < ho && nx < wo) ep_r[2 + k] = nw0 * p.noise[(size_t)b * p.noise_bstride + (size_t)ny * wo + nx];
            }
        }
    }
    load_chunk(0);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();
        store_chunk(chunk);
        if (chunk == 0) {
            if (tid < C::TN) { ep_d[tid] = ep_r[0]; ep_b[tid] = ep_r[1]; if constexpr (OSP) ep_s[tid] = ep_sn; }
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (tid + k * UF_NT < UF_OUT * UF_OUT) ep_n[tid + k * UF_NT] = ep_r[2 + k];
        }
        __syncthreads();
        if (chunk == 0) E4S_PROF_MARK(g_prof_up, 1);
        if (chunk + 1 < nchunk) load_chunk(chunk + 1);
        const uint4* whalf = wsm + khalf * C::TN + l5;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int e = xoff + (1 - (ky >> 1)) * UF_PW + (1 - (kx >> 1));
            const int ai = 2 * (ky & 1) + (kx & 1);
            const int slot = e * 2 + (khalf ^ ((e >> 3) & 1));
            const uint4 bh = xh4[slot], bl = xl4[slot];
            uint4 ah[CB], al[CB];
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                ah[i] = whalf[tap * 2 * C::TN + i * 32];
                al[i] = whalf[18 * C::TN + tap * 2 * C::TN + i * 32];
            }
#pragma unroll
            for (int i = 0; i < CB; ++i)
                accs[ai][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh), accs[ai][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CB; ++i)
                accs[ai][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl), accs[ai][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CB; ++i)
                accs[ai][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh), accs[ai][i], 0, 0, 0);
        }
    }

    // ---- epilogue: 8 output channels at a time through LDS.  Blur item of this thread: one output column, 14 rows, one channel.
    E4S_PROF_MARK(g_prof_up, 2);
    __syncthreads();
    float* zt = reinterpret_cast<float*>(lds_raw);   // [8][32][ZS]
    float kf[16];                                    // kf[ty*4+tx] = blur[3-ty][3-tx]  (uniform loads -> scalar registers)
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = p.blur[15 - t];
    constexpr int NITEM = UF_ZCO * 2 * UF_OUT;       // 448
    // channels-first output: consecutive lanes = consecutive pixels of one channel; channels-last: consecutive lanes = the 8 channels of a pixel
    const int it_co = ON ? (tid & 7) : tid / (2 * UF_OUT);
    const int it_rem = ON ? (tid >> 3) : tid - it_co * 2 * UF_OUT;
    const int it_rg = it_rem / UF_OUT, it_x = it_rem - it_rg * UF_OUT;
    const int oy0 = tyt * UF_OUT + it_rg * UF_STEP, ox = txt * UF_OUT + it_x;
    const bool it_ok = tid < NITEM && ox < wo && oy0 < ho;
    const int nrow = it_ok ? (ho - oy0 < UF_STEP ? ho - oy0 : UF_STEP) : 0;   // valid output rows of this item
    float* ob = p.out + (size_t)b * p.cout * ho * wo;
    if constexpr (OSP) {   // the zero element behind the output planes (padding source of the consumers' LDS-DMA)
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid < 4) reinterpret_cast<unsigned*>(p.out)[(size_t)p.plane_out * 8 + tid] = 0u;
    }
    const unsigned pix0 = (unsigned)(oy0 * wo + ox);
    const float* zc = ON ? zt + it_co * UF_ZCS + (it_rg * UF_STEP + 1) * UF_ZS + it_x + 1
                         : zt + (it_co * 32 + it_rg * UF_STEP + 1) * UF_ZS + it_x + 1;   // z row (local) of output row r, tap t: r + 1 + t
    // (the noise of this item's 14 outputs is read from LDS where it is used: 14 registers less across the whole epilogue — the kernel
    //  sits at the 128-register limit of two workgroups per CU and used to spill 80 bytes per lane here)
    const float* nzp = ep_n + (tid < NITEM ? it_rg * UF_STEP * UF_OUT + it_x : 0);

    E4S_PROF_MARK(g_prof_up, 3);
#pragma unroll
    for (int i = 0; i < CB; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int col = 4 * khalf + rr;   // channel (within the group of 8) held by register 4g+rr of this half-wave
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
                    *reinterpret_cast<float2*>(&zt[ON ? col * UF_ZCS + (2 * pty + ci) * UF_ZS + 2 * ptx : (col * 32 + 2 * pty + ci) * UF_ZS + 2 * ptx]) =
                        make_float2(accs[2 * ci][i][4 * g + rr], accs[2 * ci + 1][i][4 * g + rr]);
            }
            __syncthreads();
            const int co = co0 + i * 32 + 8 * g + it_co;
#ifdef E4S_PHASE_PROF
            if (p.exp & 2) { __syncthreads(); continue; }
#endif
            if (nrow > 0 && co < p.cout) {
                const float dd = ep_d[i * 32 + 8 * g + it_co], bi = ep_b[i * 32 + 8 * g + it_co];
                const float sn = OSP ? ep_s[i * 32 + 8 * g + it_co] : 0.f;
                // (OSP) dword 0 of this lane's column in its plane: even lanes write the hi plane, odd lanes the lo plane; rows are 32-bit offsets
                unsigned* osp_row = reinterpret_cast<unsigned*>(p.out) + ((tid & 1) ? (size_t)p.plane_out * 4 : (size_t)0)
                                    + (((size_t)b * (p.cout >> 3) + (size_t)(co >> 3)) * ho * wo + pix0) * 4 + ((co & 7) >> 1);
                const float neg = p.act ? 0.2f : 1.f, gain = p.act ? 1.41421356237309515f : 1.f;
                // channel-blocked output [cout/8][ho][wo][8]: the 8 channels of this pass are one block, lanes (channel, x) write contiguous bytes
                const unsigned o0 = ON ? ((unsigned)(co >> 3) * (unsigned)(ho * wo) + pix0) * 8u + (unsigned)(co & 7) : (unsigned)co * (unsigned)(ho * wo) + pix0;
                // the 14 rows in two halves of 7: seven accumulators instead of fourteen (three z rows are read twice) — with fourteen the
                // kernel spilled into scratch inside this loop, and a scratch access costs a global-memory round trip
                constexpr int HR = UF_STEP / 2;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float a[HR];
#pragma unroll
                    for (int r = 0; r < HR; ++r) a[r] = 0.f;
#pragma unroll
                    for (int zr = 0; zr < HR + 3; ++zr) {
                        const float* zp = zc + (hf * HR + zr) * UF_ZS;
                        const float z0 = zp[0], z1 = zp[1], z2 = zp[2], z3 = zp[3];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int r = zr - t;
                            if (r >= 0 && r < HR) {
                                a[r] = __builtin_fmaf(z0, kf[t * 4], a[r]);
                                a[r] = __builtin_fmaf(z1, kf[t * 4 + 1], a[r]);
                                a[r] = __builtin_fmaf(z2, kf[t * 4 + 2], a[r]);
                                a[r] = __builtin_fmaf(z3, kf[t * 4 + 3], a[r]);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < HR; ++r) {
                        const int ro = hf * HR + r;
                        if (ro < nrow) {
                            float v = __builtin_fmaf(a[r], dd, bi) + nzp[ro * UF_OUT];
                            v = fmaxf(v, v * neg) * gain;     // leaky relu 0.2 (max picks v for v >= 0, 0.2 v otherwise)
#ifdef E4S_PHASE_PROF
                            if ((p.exp & 1) && v != 12345.678f) continue;
#endif
                            if constexpr (OSP) {
                                // lanes (2m, 2m+1) hold channels (2m, 2m+1) of one pixel: both form the pair's split; the even one writes the hi
                                // dword, the odd one the lo dword — one store per lane and row, as for the fp32 output
                                const float u = __fmul_rn(v, sn);
                                const float other = __shfl_xor(u, 1, 64);
                                unsigned h2, l2;
                                split2((tid & 1) ? other : u, (tid & 1) ? u : other, h2, l2);
                                osp_row[(unsigned)(ro * wo * 4)] = (tid & 1) ? l2 : h2;
                            } else if constexpr (ON) ob[o0 + (unsigned)(ro * wo * 8)] = v;
                            else ob[o0 + (unsigned)(ro * wo)] = v;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    E4S_PROF_MARK(g_prof_up, 4);
    E4S_PROF_DRAIN();
    E4S_PROF_MARK(g_prof_up, 5);
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// The chain's up layer (split planes in, split planes out) with LDS-DMA staging.  Phase timestamps of the register-staged kernel above:
// first chunk 7 us, K loop 17.5 us for ~3 us of MFMA (every chunk waits a memory round trip for loads issued one short MFMA phase earlier),
// epilogue 21 us.  Here the chunks land by global_load_lds_dwordx4 into a ring of two stages (two chunks in flight, no registers, no VALU,
// no ds_write), issued by all eight waves; the workgroup shape (two per CU, so one's blur epilogue overlaps the other's K loop), the tile
// geometry and the epilogue are those of the kernel above.
constexpr int UD_XS4 = 4 * UF_PATCH, UD_W4 = 36 * 32, UD_STAGE4 = UD_XS4 + UD_W4;      // uint4 per stage: [hi|lo][half][289] + [hi|lo][tap][half][32]
constexpr int UD_BODY = 2 * UD_STAGE4 * 16;                                               // 73 856 bytes (the pre-blur tile overlays it afterwards)
constexpr int UD_EP_D = 0, UD_EP_B = 64, UD_EP_S = 128, UD_EP_N = 192, UD_EP_FLOATS = UD_EP_N + 13 * 64;
constexpr int UD_LDS_BYTES = UD_BODY + UD_EP_FLOATS * 4;                                  // 77 952: two workgroups per CU
constexpr int UD_G = 6;                                                                   // DMA instructions per chunk and wave (3 + 3, surplus ones repeat a piece)
static_assert(UD_LDS_BYTES <= 80 * 1024 && UF_ZCO * UF_ZCS_NHWC * 4 <= UD_BODY, "two workgroups per CU; the pre-blur tile fits the stages");

__global__ __launch_bounds__(UF_NT, 4) void up_fused_dma_kernel(const UpFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* lds4 = reinterpret_cast<uint4*>(lds_raw);
    float* epw = reinterpret_cast<float*>(lds_raw + UD_BODY);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l5 = lane & 31, khalf = lane >> 5;
    E4S_PROF_MARK(g_prof_up, 0);
    const int tyt = blockIdx.x / p.tiles_x, txt = blockIdx.x - tyt * p.tiles_x;
    const int p0y = tyt * UF_STEP - 1, p0x = txt * UF_STEP - 1;
    const int cot = blockIdx.y, co0 = cot * 32;
    const int b = blockIdx.z;
    const int hw = p.h * p.w, ho = 2 * p.h, wo = 2 * p.w;
    const int nchunk = p.cin >> 4, cb8 = p.cin >> 3;
    const unsigned zero_off = (unsigned)(2 * p.plane_in * 16);             // the 16 zero bytes behind the two input planes

    // ---- this wave's share of a chunk: activation units u = wave, wave + 8, wave + 16 of 20 (5 pieces x 4 (plane, half)); weight pieces
    // wave, wave + 8, wave + 16 of 18
    unsigned xoffs[3], xdst[3];
    bool xin[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int u = wave + 8 * k;
        u = u < 20 ? u : 19;
        const int j = u % 5, combo = u / 5;
        const int e = j * 64 + lane;
        const int py = e / UF_PW, px = e - py * UF_PW;
        const int gy = p0y - 1 + py, gx = p0x - 1 + px;
        xin[k] = e < UF_PATCH && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        // uint4 index of this lane's pixel in chunk 0 of its (plane, half); a chunk further is 2 * hw on
        xoffs[k] = (unsigned)((combo >> 1) * p.plane_in) + (unsigned)((b * cb8 + (combo & 1)) * hw + gy * p.w + gx);
        xdst[k] = (unsigned)((combo * UF_PATCH + j * 64) * 16);
    }
    // Every request goes out from inline asm (dma16_asm / dma4_asm, sb_common.h): issued through the builtin, hipcc put `s_waitcnt vmcnt(0)` in front of the
    // loop's first LDS read — chunk c + 1 had to land before chunk c's MFMAs could start, one exposed memory round trip per chunk (rounds 2-3: K loop
    // 10 us per workgroup for ~3 us of MFMA).  The kernel's own counted wait (E4S_WAIT_VM(UD_G)) is now the only one.
    auto issue = [&](int c) __attribute__((always_inline)) {
        const unsigned st = (unsigned)(c & 1) * (unsigned)(UD_STAGE4 * 16);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e = ((wave + 8 * k < 20 ? wave + 8 * k : 19) % 5) * 64 + lane;
            if (e < UF_PATCH) dma16_asm(p.x, xin[k] ? (xoffs[k] + (unsigned)(2 * c * hw)) * 16u : zero_off, st + xdst[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int piece = wave + 8 * k;
            piece = piece < 18 ? piece : 17;
            const int hl = piece / 9;                                            // 18 x 32 uint4 = 9 pieces per slab
            const int rem = piece * 64 - hl * 576 + lane;                        // [tap][half][32] index
            dma16_asm(hl ? p.wlo : p.whi, (unsigned)((((c * 18 + (rem >> 5)) * p.cout) + co0 + (rem & 31)) * 16), st + (unsigned)((UD_XS4 + piece * 64) * 16));
        }
    };
    // ---- epilogue operands by DMA as well: noise tile (13 pieces: 2 per wave), d / s_next / bias of the 32 channels (every wave, same bytes)
    {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int piece = wave + 8 * k;
            piece = piece < 13 ? piece : 12;
            const int e = piece * 64 + lane;
            const int ry = e / UF_OUT, rx = e - ry * UF_OUT;
            const int ny = tyt * UF_OUT + ry, nx = txt * UF_OUT + rx;
            const bool ok = p.noise && e < UF_OUT * UF_OUT && ny < ho && nx < wo;
            // (two exec-masked requests, together they fill the piece; a wave may skip one entirely — that only lowers the number of OLDER requests in flight,
            //  which a counted vmcnt wait tolerates)
            if (ok) dma4_asm(p.noise, (unsigned)((b * p.noise_bstride + ny * wo + nx) * 4), (unsigned)(UD_BODY + (UD_EP_N + piece * 64) * 4));
            else dma4_asm(p.zeros, 0u, (unsigned)(UD_BODY + (UD_EP_N + piece * 64) * 4));
        }
        const unsigned co4 = (unsigned)((co0 + l5) * 4);
        dma4_asm(p.d, (unsigned)(b * p.cout * 4) + co4, (unsigned)(UD_BODY + UD_EP_D * 4));
        dma4_asm(p.s_next, (unsigned)(b * p.cout * 4) + co4, (unsigned)(UD_BODY + UD_EP_S * 4));
        dma4_asm(p.act_bias ? p.act_bias : p.zeros, p.act_bias ? co4 : 0u, (unsigned)(UD_BODY + UD_EP_B * 4));
    }
    issue(0);
    if (nchunk > 1) issue(1);

    const int pty = 2 * wave + (l5 >> 4), ptx = l5 & 15;   // this lane's position inside the tile
    const int xoff = pty * UF_PW + ptx;
    f32x16 accs[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[a][r] = 0.f;
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        // chunk c has landed for this wave (only chunk c + 1's instructions may still be in flight), then for all of them
        if (c + 1 < nchunk) E4S_WAIT_VM(UD_G); else E4S_WAIT_VM(0);
        E4S_LDS_BARRIER();
        if (c == 0) E4S_PROF_MARK(g_prof_up, 1);
        unsigned xb_i = (unsigned)((c & 1) * UD_STAGE4 + khalf * UF_PATCH + xoff);
        unsigned wb_i = (unsigned)((c & 1) * UD_STAGE4 + UD_XS4 + khalf * 32 + l5);
        asm volatile("" : "+v"(xb_i), "+v"(wb_i));
        const uint4* xs = lds4 + xb_i;
        const uint4* whalf = lds4 + wb_i;
        uint4 bh[2], bl[2], ah[2], al[2];
        auto fetch = [&](int tap, int slot) __attribute__((always_inline)) {
            const int ky = tap / 3, kx = tap % 3;
            const int eo = (1 - (ky >> 1)) * UF_PW + (1 - (kx >> 1));
            bh[slot] = xs[eo];
            bl[slot] = xs[2 * UF_PATCH + eo];
            ah[slot] = whalf[tap * 64];
            al[slot] = whalf[18 * 32 + tap * 64];
        };
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cs = tap & 1;
            if (tap + 1 < 9) fetch(tap + 1, cs ^ 1);
            const int ai = 2 * ((tap / 3) & 1) + ((tap % 3) & 1);
            accs[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[cs]), __builtin_bit_cast(bf16x8, bh[cs]), accs[ai], 0, 0, 0);
            accs[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[cs]), __builtin_bit_cast(bf16x8, bl[cs]), accs[ai], 0, 0, 0);
            accs[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[cs]), __builtin_bit_cast(bf16x8, bh[cs]), accs[ai], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        E4S_LDS_BARRIER();                                   // everyone is done with this stage: refill it with chunk c + 2
        if (c + 2 < nchunk) issue(c + 2);
    }

    // ---- epilogue (as up_fused_sb_kernel's split-plane variant): 8 output channels per pass through the pre-blur tile, which overlays the stages
    E4S_PROF_MARK(g_prof_up, 2);
    // Channel PAIRS through the blur: the pre-blur tile is kept as [4 pairs][32 rows][34 columns][2 channels], so one ds_read_b64 feeds
    // a packed FMA (v_pk_fma_f32: both channels of a pair per instruction) and the thread that finishes a pixel holds exactly the two
    // values of one bf16 dword of the split planes — no lane exchange, no select.  Thread = (column, pair, group of 7 rows): a wave is
    // 16 columns x 4 pairs (the four dwords of a pixel's 16 bytes leave in one store instruction), waves = 2 column halves x 4 row groups.
    // Per output the same FMAs in the same order as the single-channel form (bit-identical results).
    float2* zt2 = reinterpret_cast<float2*>(lds_raw);
    const float* ep_d = epw + UD_EP_D;
    const float* ep_b = epw + UD_EP_B;
    const float* ep_s = epw + UD_EP_S;
    const float* ep_n = epw + UD_EP_N;
    // the 16 taps as 8 aligned register pairs: a packed FMA broadcasts either half of a pair to both channels (op_sel), so no tap is ever copied
    f32x2 kp[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) kp[t] = (f32x2){p.blur[15 - 2 * t], p.blur[14 - 2 * t]};
    const float nw = p.noise ? p.noise_weight[0] : 0.f;
    constexpr int ZP_ROW = UF_ZS;                       // float2 per row (34)
    constexpr int ZP_PAIR = 32 * UF_ZS + 16;            // float2 per channel pair: +16 so that the four pairs of a wave fall on the two bank halves
    static_assert(4 * ZP_PAIR * 8 <= UD_BODY, "the pair-interleaved pre-blur tile fits the stages");
    constexpr int RG = 7;                               // output rows per thread
    const int bx = (lane & 15) + 16 * (wave & 1), bcp = (lane >> 4), brg = wave >> 1;
    const int oy0 = tyt * UF_OUT + brg * RG, ox = txt * UF_OUT + bx;
    const bool col_ok = bx < UF_OUT && ox < wo;
    int nrow = ho - oy0;
    nrow = nrow < 0 ? 0 : (nrow > RG ? RG : nrow);     // wave-uniform
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid < 4) reinterpret_cast<unsigned*>(p.out)[(size_t)p.plane_out * 8 + tid] = 0u;   // zero tail
    const float2* zc0 = zt2 + bcp * ZP_PAIR + (brg * RG + 1) * ZP_ROW + (bx < UF_OUT ? bx : 0) + 1;
    const float* nzp = ep_n + brg * RG * UF_OUT + (bx < UF_OUT ? bx : 0);
    const float neg = p.act ? 0.2f : 1.f, gain = p.act ? 1.41421356237309515f : 1.f;
    E4S_PROF_MARK(g_prof_up, 3);
    static_assert(2 * 4 * ZP_PAIR * 8 <= UD_BODY, "two passes of the pre-blur tile fit the stages");
    // Two passes (16 channels) of pre-blur values go to LDS at a time: half the barriers, and half of the accumulators are dead before
    // the first blur starts — the blur's own registers then fit under 128 without spilling (a spill reload would queue behind the output
    // stores: one in-order vmcnt).
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
#pragma unroll
        for (int gs = 0; gs < 2; ++gs) {
            const int g = 2 * gg + gs;
            // this lane's four channels of the pass (4 khalf + 0..3 = pairs 2 khalf, 2 khalf + 1) at its 2 x 2 pre-blur values
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    float2* zd = &zt2[(4 * gs + 2 * khalf + q) * ZP_PAIR + (2 * pty + ci) * ZP_ROW + 2 * ptx];
                    zd[0] = make_float2(accs[2 * ci][4 * g + 2 * q], accs[2 * ci][4 * g + 2 * q + 1]);            // adjacent registers: 8-byte stores, no copies
                    zd[1] = make_float2(accs[2 * ci + 1][4 * g + 2 * q], accs[2 * ci + 1][4 * g + 2 * q + 1]);
                }
        }
        __syncthreads();
#pragma unroll 1
        for (int gs = 0; gs < 2; ++gs) {
        const int g = 2 * gg + gs;
        const float2* zc = zc0 + gs * 4 * ZP_PAIR;
#ifdef E4S_PHASE_PROF
        if (p.exp & 2) continue;                                         // experiment: no blur
#endif
        const int cl = 8 * g + 2 * bcp;                 // the pair's first channel inside this workgroup's 32
        const int co = co0 + cl;
        if (nrow > 0 && co < p.cout) {
            const float2 dd = make_float2(ep_d[cl], ep_d[cl + 1]), bi = make_float2(ep_b[cl], ep_b[cl + 1]), sn = make_float2(ep_s[cl], ep_s[cl + 1]);
            unsigned* orow = reinterpret_cast<unsigned*>(p.out) + (((size_t)b * (p.cout >> 3) + (size_t)(co >> 3)) * ho * wo + (size_t)oy0 * wo + ox) * 4 + ((co & 7) >> 1);
            // Four output rows in flight: pre-blur row zr gives its last term (tap row 3) to output row zr - 3, which is finished and stored
            // at once, then the window moves on.  A rolled loop with a fixed body: 4 accumulator pairs + 2 rows of reads, whatever RG is.
            f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3;
            f32x2 z[4], zn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float2 t2 = zc[u]; z[u] = (f32x2){t2.x, t2.y}; }
#pragma unroll 2          // (two rows per trip: the z <- zn hand-over and the loop control halve; -2 % on the two launches, 0.597 -> 0.585 ms per step; five per trip spills)
            for (int zr = 0; zr < RG + 3; ++zr) {
                if (zr + 1 < RG + 3) {
                    const float2* zp = zc + (zr + 1) * ZP_ROW;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const float2 t2 = zp[u]; zn[u] = (f32x2){t2.x, t2.y}; }
                }
                a3 = (f32x2){0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x2 zz = z[u];
#define E4S_KB(i) (((i) & 1) ? __builtin_shufflevector(kp[(i) >> 1], kp[(i) >> 1], 1, 1) : __builtin_shufflevector(kp[(i) >> 1], kp[(i) >> 1], 0, 0))
                    a0 = __builtin_elementwise_fma(zz, E4S_KB(12 + u), a0);
                    a1 = __builtin_elementwise_fma(zz, E4S_KB(8 + u), a1);
                    a2 = __builtin_elementwise_fma(zz, E4S_KB(4 + u), a2);
                    a3 = __builtin_elementwise_fma(zz, E4S_KB(u), a3);
#undef E4S_KB
                }
                const int r = zr - 3;
                if (r >= 0 && r < nrow && col_ok) {
                    const float nz = __fmul_rn(nw, nzp[r * UF_OUT]);
                    float v0 = __builtin_fmaf(a0[0], dd.x, bi.x) + nz, v1 = __builtin_fmaf(a0[1], dd.y, bi.y) + nz;
                    v0 = fmaxf(v0, v0 * neg) * gain;
                    v1 = fmaxf(v1, v1 * neg) * gain;
                    unsigned h2, l2;
                    split2(__fmul_rn(v0, sn.x), __fmul_rn(v1, sn.y), h2, l2);
#ifdef E4S_PHASE_PROF
                    if ((p.exp & 1) && v0 != 12345.678f) continue;      // experiment: no output stores
#endif
                    orow[(unsigned)(r * wo * 4)] = h2;
                    orow[(size_t)p.plane_out * 4 + (unsigned)(r * wo * 4)] = l2;
                }
                a0 = a1; a1 = a2; a2 = a3;
#pragma unroll
                for (int u = 0; u < 4; ++u) z[u] = zn[u];
            }
        }
        }
        __syncthreads();
    }
    E4S_PROF_MARK(g_prof_up, 4);
    E4S_PROF_DRAIN();
    E4S_PROF_MARK(g_prof_up, 5);
}

__device__ uint4 g_uf_zero[4];   // 64 zero bytes (UpFusedParams::zeros)

template <int CB, int MINW, bool XN = false, bool ON = false, bool XSP = false, bool OSP = false>
int launch_up_fused(UpFusedParams& p, hipStream_t st) {
    using C = UfCfg<CB>;
    dim3 grid(p.tiles_x * p.tiles_y, cdiv(p.cout, C::TN), p.bs);
    if (C::LDS_BYTES > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&up_fused_sb_kernel<CB, MINW, XN, ON, XSP, OSP>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (attr != hipSuccess) return fail((int)attr, "modconv_up_fused_sb: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    }
    hipLaunchKernelGGL((up_fused_sb_kernel<CB, MINW, XN, ON, XSP, OSP>), grid, dim3(UF_NT), C::LDS_BYTES, st, p);
    return check_launch("modconv_up_fused_sb");
}

}  // namespace

#ifdef E4S_PHASE_PROF
extern "C" E4S_API int e4s_prof_read_up(long long* host, int64_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof_up), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
extern "C" E4S_API int e4s_prof_clear_up() {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_prof_up));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(ptr, 0, sizeof(long long) * (size_t)E4S_PROF_BLOCKS * E4S_PROF_SLOTS);
}
#endif

extern "C" int e4s_modconv_up_fused_sb(float* out, const float* x, const uint16_t* whi, const uint16_t* wlo, const float* s, const float* d,
                                       const float* blur, const float* noise, int noise_bs, const float* noise_weight, const float* act_bias,
                                       int act, int bs, int cin, int cout, int h, int w, const float* s_next, void* stream) {
    E4S_REQUIRE(out && x && whi && wlo && s && blur, "modconv_up_fused_sb: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= 1 && cout >= 1 && h >= 1 && w >= 1, "modconv_up_fused_sb: bad size");
    E4S_REQUIRE((int64_t)cout * 4 * h * w < ((int64_t)1 << 31) && (int64_t)cin * h * w < ((int64_t)1 << 31), "modconv_up_fused_sb: one sample must stay below 2^31 elements");
    E4S_REQUIRE((((uintptr_t)whi | (uintptr_t)wlo) & 15) == 0, "modconv_up_fused_sb: weight slabs must be 16-byte aligned");
    E4S_REQUIRE(!noise || (noise_weight && (noise_bs == 1 || noise_bs == bs)), "modconv_up_fused_sb: noise needs its weight and batch 1 or bs");
    const bool x_nhwc = (act & E4S_X_NHWC) != 0, out_nhwc = (act & E4S_OUT_NHWC) != 0, x_sp = (act & E4S_X_SP) != 0, out_sp = (act & E4S_OUT_SP) != 0;
    act &= 1;
    E4S_REQUIRE(!(x_sp && x_nhwc) && !(out_sp && out_nhwc), "modconv_up_fused_sb: one layout per side");
    E4S_REQUIRE(!x_sp || (cin % 16 == 0 && ((uintptr_t)x & 15) == 0), "modconv_up_fused_sb: split-plane input needs cin %% 16 == 0 and a 16-byte aligned tensor");
    E4S_REQUIRE(!out_sp || (s_next && cout % 8 == 0 && ((uintptr_t)out & 15) == 0), "modconv_up_fused_sb: split-plane output needs s_next, cout %% 8 == 0 and a 16-byte aligned tensor");
    E4S_REQUIRE(!x_nhwc || (cin % 16 == 0 && ((uintptr_t)x & 15) == 0), "modconv_up_fused_sb: channels-last input needs cin %% 16 == 0 and a 16-byte aligned tensor");
    E4S_REQUIRE(!out_nhwc || cout % 8 == 0, "modconv_up_fused_sb: channels-last output needs cout %% 8 == 0");
    if (bs == 0) return 0;
    UpFusedParams p;
    p.out = out; p.x = x; p.whi = reinterpret_cast<const uint4*>(whi); p.wlo = reinterpret_cast<const uint4*>(wlo); p.s = s; p.d = d;
    p.blur = blur; p.noise = noise; p.noise_weight = noise_weight; p.act_bias = act_bias;
    p.noise_bstride = (noise && noise_bs > 1) ? 4 * h * w : 0;
    p.act = act; p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w;
    p.s_next = s_next;
    p.plane_in = (int64_t)bs * (cin / 8) * h * w;
    p.plane_out = (int64_t)bs * (cout / 8) * 4 * h * w;
    p.tiles_x = cdiv(2 * w, UF_OUT);
    p.tiles_y = cdiv(2 * h, UF_OUT);
#ifdef E4S_PHASE_PROF
    { const char* e = getenv("E4S_UF_EXP"); p.exp = e ? atoi(e) : 0; }      // (tuning build only)
#else
    p.exp = 0;
#endif
    hipStream_t st = (hipStream_t)stream;
    if (x_sp && out_sp) {
        constexpr int use_dma = 1;
        if (use_dma && cout % 32 == 0 && cin % 16 == 0 && d) {
            static const float* zeros = [] {
                void* ptr = nullptr;
                return hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_uf_zero)) == hipSuccess ? static_cast<const float*>(ptr) : nullptr;
            }();
            constexpr int lds_bytes = UD_LDS_BYTES;
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&up_fused_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (zeros && attr == hipSuccess) {
                p.zeros = zeros;
                hipLaunchKernelGGL(up_fused_dma_kernel, dim3(p.tiles_x * p.tiles_y, cout / 32, bs), dim3(UF_NT), lds_bytes, st, p);
                return check_launch("modconv_up_fused_sb");
            }
        }
        return launch_up_fused<1, 4, false, true, true, true>(p, st);
    }
    if (x_sp || out_sp) return fail(E4S_ERR_ARG, "modconv_up_fused_sb: split planes are built for both sides together (the chain's up layers)");
    if (x_nhwc && out_nhwc) return launch_up_fused<1, 4, true, true>(p, st);
    if (x_nhwc) return launch_up_fused<1, 4, true, false>(p, st);
    if (out_nhwc) return launch_up_fused<1, 4, false, true>(p, st);
    return launch_up_fused<1, 4>(p, st);                          // 32 co per workgroup, 2 workgroups per CU
}
