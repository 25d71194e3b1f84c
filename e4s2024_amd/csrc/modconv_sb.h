// Shared by modconv_sb.hip (register-staged kernels) and modconv_mx.hip (DMA-fed masked kernel): launch parameters, tile geometry and the StyledConv
// epilogue (demodulation, noise, bias, leaky ReLU, optional fused single-region ToRGB and split-plane hand-over).
#pragma once
#include "common.h"
#include "sb_common.h"

namespace e4s {

struct SbParams {
    float* out;
    const float* x;
    const uint4* whi;  // bf16 x 8 per uint4
    const uint4* wlo;
    const float* s;
    const float* d;
    const uint8_t* labels;
    const float* noise;
    const float* noise_weight;
    const float* act_bias;
    int lh, lw;
    float lscale_y, lscale_x;
    int noise_bstride;
    int act;
    int bs, cin, cout, h, w, nreg, up;
    int x_nhwc, out_nhwc;    // channel-blocked activations [bs, c/8, h, w, 8] (cin % 16 == 0 / cout % 8 == 0): a pixel's 8 channels are 32
                             // contiguous bytes and consecutive pixels follow — 16-byte loads / stores that fill whole cache lines on both sides
    int tiles_x, tiles_y;
    int ksplit, chunks_per;  // split-K: block ks handles chunks [ks*chunks_per, (ks+1)*chunks_per)
    float* partial;
    // optional fused single-region ToRGB (model.py:439-479) on this layer's output; needs the whole Cout in one workgroup
    float* rgb_out;          // [bs,3,ho,wo]
    const float* rgb_wt;     // [cout][3]  (e4s_modconv_prep_weights, k = 1)
    const float* rgb_s;      // [bs][cout]
    const float* rgb_bias;   // [3]
    const float* rgb_skip;   // [bs,3,ho/2,wo/2] or NULL
    const float* rgb_upk;    // [4,4]
    // optional split-plane output (modconv_chain.hip): out = [hi|lo][bs][cout/8][ho][wo][8 x bf16] of act * s_next[b][co], for a single-region consumer
    const float* s_next;     // [bs][cout]
    int64_t plane_out;       // uint4 per plane
    // optional (masked up layer): [bs][ho/16][wo/16] map of region-uniform 16 x 16 output blocks (modconv_upblock.hip computes those: value <
    // nreg); this kernel then skips them — whole workgroups where all their blocks are uniform, single pixels otherwise
    const uint8_t* uni_blocks;
    const int* uni_ctrl;     // uni_ctrl[2] == 0: the map is not in use for this layer (too few blocks qualify)
    const unsigned char* wmx;  // (modconv_mx.hip) the weights as DMA-ready row slots: e4s_modconv_prep_weights_mx
    const float* in_mean;      // (modconv_mx.hip, plain-convolution mode) [bs][cin] instance-norm statistics applied to the input on load, or NULL
    const float* in_rstd;
    const float* slope;        // (plain-convolution mode) PReLU slopes [cout] or NULL
    int xcd_remap;             // (modconv_mx.hip) 1: workgroups are re-indexed so that each XCD (own L2) works on one co tile's weights
    int* flags;                // (modconv_mx.hip, f16 arithmetic) flags[0] |= 1 when a modulated activation leaves the f16 range
    unsigned perm_mul;       // (with uni_blocks) workgroup i works on tile slot (i * perm_mul) % gridDim.x: consecutive workgroups go to the 8 XCDs
                             // round-robin, so a skip pattern with a period of 2 / 4 / 8 tiles would idle whole XCDs; a golden-ratio stride
                             // coprime with the grid spreads any spatially coherent skip set evenly
};

template <int CB, int PB, int WC, int WP, int LOG_TW>
struct SbCfg {
    static constexpr int TN = WC * CB * 32;
    static constexpr int NPB = WP * PB;
    static constexpr int TW = 1 << LOG_TW;
    static constexpr int RPB = 32 >> LOG_TW;                // tile rows per 32-pixel MFMA block (tiles up to 32 wide)
    static constexpr int BPR = TW > 32 ? TW / 32 : 1;       // 32-pixel blocks per tile row (64-wide tiles: a row segment of 66 pixels costs
                                                            // 3 cache lines per channel for 264 B instead of 3 for 136 B)
    static constexpr int TH = TW > 32 ? NPB / BPR : NPB * RPB;
    static_assert(TW <= 32 || NPB % BPR == 0, "whole rows only");
    // tile coordinates of lane l5 (0..31) of pixel block pbk
    __host__ __device__ static constexpr int blk_y(int pbk, int l5) { return TW > 32 ? pbk / BPR : pbk * RPB + (l5 >> LOG_TW); }
    __host__ __device__ static constexpr int blk_x(int pbk, int l5) { return TW > 32 ? (pbk % BPR) * 32 + l5 : (l5 & (TW - 1)); }
    static constexpr int PW = TW + 2, PH = TH + 2;
    static constexpr int PATCH = PH * PW;
    static constexpr int NT = 64 * WC * WP;                 // threads per workgroup (256 or 512)
    static constexpr int EPT = (PATCH + NT - 1) / NT;
    static constexpr int XS_FLOATS = CKS * PATCH;
    static constexpr int W4 = 2 * 9 * 2 * TN;               // uint4 per chunk: [hi/lo][tap][half][TN]
    static constexpr int WPT = (W4 + NT - 1) / NT;          // uint4 per thread
    static constexpr int SS_FLOATS = E4S_MAX_REGIONS * CKS;
    static constexpr int LDS_BYTES = XS_FLOATS * 4 + W4 * 16 + SS_FLOATS * 4;
    static constexpr int LDS_BYTES_UNI = XS_FLOATS * 4 + W4 * 16;   // single-region kernels keep no per-region style table: 32 co x 256 px
                                                                    // is then 40 192 B, four workgroups per CU instead of three
    static_assert(WC * WP == 4 || WC * WP == 8, "256- or 512-thread workgroups");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per CU");
    static_assert((E4S_MAX_REGIONS + 5) * TN * 4 + 64 <= W4 * 16, "demod + bias + ToRGB + next-modulation tables overlay the weight stage");
};

// Element [co][ci][tap] of the kernel a 3x3 modulated conv multiplies with (before the equalised-lr scale): the weight itself, or — `up` — the stride-2
// transposed conv composed with the 4x4 blur for output parity `par` (model.py:287-300).
__device__ __forceinline__ float sb_weff(const float* __restrict__ weight, const float* __restrict__ blur, int cin, int co, int ci, int tap, int par, int up) {
    const float* w = weight + ((size_t)co * cin + ci) * 9;
    if (!up) return w[tap];
    const int a = par >> 1, b = par & 1;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    float v = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = ky + 2 * dy + 1 - a;
        if (ty < 0 || ty > 3) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = kx + 2 * dx + 1 - b;
            if (tx < 0 || tx > 3) continue;
            v += blur[(3 - ty) * 4 + (3 - tx)] * w[ky * 3 + kx];
        }
    }
    return v;
}

// (masked up layers) Do the four outputs (2y + pa, 2x + pb) of EVERY position of the 32 x 8 tile at (y0, x0) carry one region?  Workgroup-uniform (one barrier); every
// thread passes its position (tile row ty = its wave, column tx = lane & 31: the half-waves hold the same positions); `c_own` = the raw label of the position's outputs
// (E4S_LABEL_NONE outside the map).  modconv_mx4.hip runs its four-parity tile where this holds and the composed kernel's tile (modconv_mx_tile.h) where it does not.
// `flagw` = one LDS word of the caller's own dynamic allocation that nothing else uses.  (NOT __syncthreads_and: the device library's workgroup reduction brings 256 bytes
// of STATIC LDS with it, the dynamic region then starts at 256, and these kernels hand absolute LDS addresses to their LDS-DMA.)
__device__ __forceinline__ bool quad_uniform_tile(const SbParams& p, int b, int y0, int x0, int ty, int tx, int& c_own, volatile int* flagw) {
    const int y = y0 + ty, x = x0 + tx;
    bool uni = true;
    c_own = E4S_LABEL_NONE;
    if (y < p.h && x < p.w) {
        const uint8_t* lb = p.labels + (size_t)b * p.lh * p.lw;
        const int r0 = nearest_src(2 * y, p.lscale_y, p.lh) * p.lw, r1 = nearest_src(2 * y + 1, p.lscale_y, p.lh) * p.lw;
        const int q0 = nearest_src(2 * x, p.lscale_x, p.lw), q1 = nearest_src(2 * x + 1, p.lscale_x, p.lw);
        const int c00 = lb[r0 + q0], c01 = lb[r0 + q1], c10 = lb[r1 + q0], c11 = lb[r1 + q1];
        uni = c00 == c01 && c00 == c10 && c00 == c11;
        c_own = c00;
    }
    if (threadIdx.x == 0) *flagw = 1;
    __syncthreads();
    if (!uni) *flagw = 0;
    __syncthreads();
    return *flagw != 0;
}

// The one-pass epilogue of a workgroup tile.  `acc[i][q]` = raw sums of output-channel block i (32 channels) x pixel block q (32 pixels) of this wave;
// `cls[q]` = region of the lane's pixel (-1: none); the tables overlay the first (MAX_REG + 5) * TN * 4 + 64 bytes of `lds_raw`, which the caller
// must be done with (RGB with two wave groups per pixel: 3 * NPB * 32 more floats behind them).  Every global LOAD happens before the first store (one in-order vmcnt for both on gfx9).
template <class C, int CB, int PB, int WP, bool RGB, bool OSP>
__device__ __forceinline__ void sb_epilogue(const SbParams& p, unsigned char* lds_raw, f32x16 (&acc)[CB][PB], const int (&cls)[PB], int co0, int b, int y0, int x0,
                                            int pa, int pb_, int ho, int wo, unsigned ub_skip) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    const int wc = wave / WP, wp = wave % WP;
    constexpr int WC_ = C::TN / (CB * 32);          // wave groups along the output channels
    static_assert(!RGB || WC_ <= 2, "fused ToRGB: one or two wave groups per pixel");
    float rgbp[PB][3];                              // (RGB, WC_ == 2) this wave's partial ToRGB sums
    (void)rgbp;
    __syncthreads();
    float* dt = reinterpret_cast<float*>(lds_raw);  // [MAX_REG][TN] over the weight stage
    for (int v = tid; v < E4S_MAX_REGIONS * C::TN; v += C::NT) {
        const int r = v / C::TN, n = v % C::TN;
        float val = 0.f;
        if (r < p.nreg && co0 + n < p.cout) val = p.d ? p.d[((size_t)b * p.nreg + r) * p.cout + co0 + n] : 1.f;
        dt[v] = val;
    }
    float* bt = dt + E4S_MAX_REGIONS * C::TN;    // [TN] activation bias
    for (int v = tid; v < C::TN; v += C::NT) bt[v] = (p.act_bias && co0 + v < p.cout) ? p.act_bias[co0 + v] : 0.f;
    float* wsr = bt + C::TN;                     // [TN][3]: ToRGB weight x its (single-region) modulation
    float* kfr = wsr + 3 * C::TN;                // [16] flipped skip-upsample taps
    float* snt = kfr + 16;                       // [TN] next layer's modulation (OSP)
    if constexpr (OSP) {
        for (int v = tid; v < C::TN; v += C::NT) snt[v] = (co0 + v < p.cout) ? p.s_next[(size_t)b * p.cout + co0 + v] : 0.f;
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid < 4) reinterpret_cast<unsigned*>(p.out)[(size_t)p.plane_out * 8 + tid] = 0u;   // zero tail
    }
    if constexpr (RGB) {
        for (int v = tid; v < 3 * C::TN; v += C::NT) {
            const int n = v / 3;
            wsr[v] = (co0 + n < p.cout) ? p.rgb_wt[(size_t)(co0 + n) * 3 + (v - n * 3)] * p.rgb_s[(size_t)b * p.cout + co0 + n] : 0.f;
        }
        if (tid < 16) kfr[tid] = p.rgb_upk ? p.rgb_upk[15 - tid] : 0.f;
    }
    __syncthreads();
    // Every global LOAD of the epilogue happens before the first store: gfx9 counts loads and stores in one in-order counter
    // (vmcnt), so a load issued after a store cannot be waited for without also waiting for that store to reach memory.
    const float nw = p.noise ? p.noise_weight[0] : 0.f;
    float nzq[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int y = y0 + C::blk_y(pbk, l5), x = x0 + C::blk_x(pbk, l5);
        const int oy = p.up ? 2 * y + pa : y, ox = p.up ? 2 * x + pb_ : x;
        nzq[q] = (p.noise && y < p.h && x < p.w) ? nw * p.noise[(size_t)b * p.noise_bstride + (size_t)oy * wo + ox] : 0.f;
    }
    float rgbadd[PB][3];   // fused ToRGB: bias + FIR-upsampled skip of this lane's pixels, also fetched before any store
    if constexpr (RGB) {
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int pbk = wp * PB + q;
            const int oy = y0 + C::blk_y(pbk, l5), ox = x0 + C::blk_x(pbk, l5);   // fused ToRGB runs on same-resolution layers only
            const bool ok = oy < p.h && ox < p.w;
            const int hs = ho >> 1, wsk = wo >> 1;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float u = p.rgb_bias[o];
                if (p.rgb_skip && ok) {  // upfirdn2d(skip, up=2, pad=(2,1)) at (oy, ox): see region_torgb_kernel
                    const int iy0 = (oy - 1) >> 1, ix0 = (ox - 1) >> 1;
                    const int ky0 = 2 * iy0 + 2 - oy, kx0 = 2 * ix0 + 2 - ox;
                    const float* sp = p.rgb_skip + ((size_t)b * 3 + o) * hs * wsk;
#pragma unroll
                    for (int ty = 0; ty < 2; ++ty) {
                        const int iy = iy0 + ty;
                        if (iy < 0 || iy >= hs) continue;
#pragma unroll
                        for (int tx = 0; tx < 2; ++tx) {
                            const int ix = ix0 + tx;
                            if (ix < 0 || ix >= wsk) continue;
                            u += sp[(size_t)iy * wsk + ix] * kfr[(ky0 + 2 * ty) * 4 + kx0 + 2 * tx];
                        }
                    }
                }
                rgbadd[q][o] = u;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int y = y0 + C::blk_y(pbk, l5), x = x0 + C::blk_x(pbk, l5);
        const bool pix_ok = y < p.h && x < p.w && !((ub_skip >> ((x - x0) >> 3)) & 1u);
        const int oy = p.up ? 2 * y + pa : y, ox = p.up ? 2 * x + pb_ : x;
        const size_t opix = (size_t)oy * wo + ox;
        const float nz = nzq[q];
        const float* drow = dt + (cls[q] >= 0 ? cls[q] : 0) * C::TN;
        const float dz = cls[q] >= 0 ? 1.f : 0.f;
        float rgb0 = 0.f, rgb1 = 0.f, rgb2 = 0.f;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * r4 + e;
                    const int n = (wc * CB + i) * 32 + e + 8 * r4 + 4 * khalf;
                    const int co = co0 + n;
                    float v = acc[i][q][r] * drow[n] * dz + nz + bt[n];
                    if (p.act) v = (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
                    v4[e] = v;
                    if (co < p.cout && pix_ok) {
                        if ((!RGB || p.out) && !p.out_nhwc && !OSP) p.out[((size_t)b * p.cout + co) * ho * wo + opix] = v;   // out == NULL: only the fused RGB is wanted
                        if constexpr (RGB) {
                            rgb0 += v * wsr[n * 3 + 0];
                            rgb1 += v * wsr[n * 3 + 1];
                            rgb2 += v * wsr[n * 3 + 2];
                        }
                    }
                }
                const int co4 = co0 + (wc * CB + i) * 32 + 8 * r4 + 4 * khalf;      // four consecutive output channels of this pixel
                if constexpr (OSP) {
                    if (pix_ok && co4 < p.cout) {
                        const int n4 = co4 - co0;
                        unsigned h0, l0, h1, l1;
                        split2(__fmul_rn(v4[0], snt[n4]), __fmul_rn(v4[1], snt[n4 + 1]), h0, l0);
                        split2(__fmul_rn(v4[2], snt[n4 + 2]), __fmul_rn(v4[3], snt[n4 + 3]), h1, l1);
                        uint2* osp = reinterpret_cast<uint2*>(p.out);
                        const size_t o8 = (((size_t)b * (p.cout >> 3) + (co4 >> 3)) * ho * wo + opix) * 2 + khalf;   // this half-wave's 8 of the block's 16 bytes
                        osp[o8] = make_uint2(h0, h1);
                        osp[(size_t)p.plane_out * 2 + o8] = make_uint2(l0, l1);
                    }
                } else if (p.out_nhwc && (!RGB || p.out) && pix_ok && co4 < p.cout)
                    *reinterpret_cast<float4*>(p.out + (((size_t)b * (p.cout / 8) + co4 / 8) * ho * wo + opix) * 8 + (co4 & 7)) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            }
        }
        if constexpr (RGB) {  // WC == 1: this wave holds every output channel of its pixels, split over the two half-waves
            rgb0 += __shfl_xor(rgb0, 32, 64);
            rgb1 += __shfl_xor(rgb1, 32, 64);
            rgb2 += __shfl_xor(rgb2, 32, 64);
            if constexpr (WC_ == 1) {
                if (khalf == 0 && pix_ok) {
                    p.rgb_out[((size_t)b * 3 + 0) * ho * wo + opix] = rgb0 + rgbadd[q][0];
                    p.rgb_out[((size_t)b * 3 + 1) * ho * wo + opix] = rgb1 + rgbadd[q][1];
                    p.rgb_out[((size_t)b * 3 + 2) * ho * wo + opix] = rgb2 + rgbadd[q][2];
                }
            } else {
                rgbp[q][0] = rgb0; rgbp[q][1] = rgb1; rgbp[q][2] = rgb2;
            }
        }
    }
    if constexpr (RGB && WC_ == 2) {
        // WC == 2 (2 x 64 channels per pixel in waves wc = 0 / 1): the upper channel half's partial sums cross through LDS behind the tables
        float* xch = snt + C::TN;                    // [3][NPB * 32]
        if (wc == 1 && khalf == 0) {
#pragma unroll
            for (int q = 0; q < PB; ++q)
#pragma unroll
                for (int o = 0; o < 3; ++o) xch[o * (C::NPB * 32) + (wp * PB + q) * 32 + l5] = rgbp[q][o];
        }
        __syncthreads();
        if (wc == 0 && khalf == 0) {
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int pbk = wp * PB + q;
                const int y = y0 + C::blk_y(pbk, l5), x = x0 + C::blk_x(pbk, l5);
                if (y < p.h && x < p.w) {
                    const size_t opix = (size_t)y * wo + x;        // (fused ToRGB: same-resolution layers only)
#pragma unroll
                    for (int o = 0; o < 3; ++o)
                        p.rgb_out[((size_t)b * 3 + o) * ho * wo + opix] = rgbp[q][o] + xch[o * (C::NPB * 32) + pbk * 32 + l5] + rgbadd[q][o];
                }
            }
        }
    }
}

// modconv_mx.hip: the DMA-fed 128 co x 256 px masked kernel (p.wmx set).  Returns E4S_OK or an error code.
int launch_modconv_mx(SbParams& p, int arith, hipStream_t st, float* workspace, int64_t workspace_floats, bool plain_conv = false);

}  // namespace e4s
