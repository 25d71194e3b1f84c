// a3/a4, the single-region chain's same-resolution layers (past remaining_layer_idx: 512 conv, 1024 conv; the up layers: modconv_uphc.hip): persistent
// workgroups fed by LDS-DMA from PRE-MODULATED, PRE-SPLIT activation planes.
//
// Why.  The single-region layers are short-K (32-128 input channels) and their old kernels (modconv_sb.hip UNI path, modconv_upfused.hip)
// spend a workgroup's life waiting: loads go to registers first (so only one 16-channel chunk can be in flight), every staged value is
// multiplied by the layer's modulation and split into bf16 hi/lo on its way into LDS, and each workgroup pays its own prologue / epilogue
// latency chain (profiles/r01: 0.16-0.25 of either roof).  For a single-region layer the modulation s[b][ci] is a property of the INPUT
// channel alone, and every layer's tables are known before the first layer runs (ops.style_demod_plan) — so the PRODUCER of an
// activation can apply the consumer's modulation and the bf16 split in its epilogue and write the tensor as
//
//      "split planes"  xsp[plane hi/lo][b][c/8][y][x][8 x bf16]           (same bytes as the fp32 tensor)
//
// whose 16-byte element (8 channels of one pixel) IS the MFMA B fragment of one lane.  The consumer then stages with
// global_load_lds_dwordx4 only: no registers, no VALU, no ds_write; as many chunks in flight as LDS holds; and a persistent workgroup
// walks over tiles so that the next tile's chunks land while the current tile's epilogue runs.  Arithmetic is unchanged: fl(v * s) then
// the RNE split is exactly what the old consumers computed while staging, MFMA order and epilogue formulas are the same.
//
// Reference semantics: ModulatedConv2d.forward fused branch (models/stylegan2/model.py:276-320), StyledConv noise / bias / activation
// (:417-421), ToRGB (:439-479) — the single-region case (mask_op False) of each.
#include <stdlib.h>

#include "common.h"
#include "sb_common.h"

using namespace e4s;

namespace {

__device__ uint4 g_zero16[4];   // 64 zero bytes (zero-initialised device global): ChainParams::zeros
#define CH_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// s_waitcnt vmcnt(n) for an n that is a constant only after unrolling: the switch folds to the one case (vmcnt is a 6-bit field)
__device__ __forceinline__ void wait_vm(int n) {
    n = n < 0 ? 0 : (n > 63 ? 63 : n);
    switch (n) {
#define CH_CASE(v) case v: CH_WAIT_VM(v); break;
#define CH_CASE8(v) CH_CASE(v) CH_CASE(v + 1) CH_CASE(v + 2) CH_CASE(v + 3) CH_CASE(v + 4) CH_CASE(v + 5) CH_CASE(v + 6) CH_CASE(v + 7)
        CH_CASE8(0) CH_CASE8(8) CH_CASE8(16) CH_CASE8(24) CH_CASE8(32) CH_CASE8(40) CH_CASE8(48) CH_CASE8(56)
#undef CH_CASE8
#undef CH_CASE
    }
}
// LDS-DMA stays in flight across this barrier (a __syncthreads() would drain it: its fence waits vmcnt(0) while a DMA is pending)
#define CH_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int CT_TW = 32, CT_TH = 16;                 // conv output tile (pixels): 8 compute waves x 2 rows of 32
constexpr int CT_PW = CT_TW + 2, CT_PH = CT_TH + 2;
constexpr int CT_PATCH = CT_PW * CT_PH;               // 612 patch pixels
constexpr int CT_NPIECE = (CT_PATCH + 63) / 64;       // 10 DMA pieces of 64 pixels per (plane, half)
constexpr int CT_NCW = 8;                             // compute waves; wave CT_NCW is the loader
constexpr int CT_NT = 64 * (CT_NCW + 1);              // 576 threads
constexpr int CT_SKW = CT_TW / 2 + 2, CT_SKH = CT_TH / 2 + 2;   // skip patch of the fused ToRGB: 18 x 10 per colour
constexpr int CT_SKIP = 3 * CT_SKW * CT_SKH;          // 540 floats

struct ChainParams {
    const uint4* xsp;        // [2][bs][cin/8][h][w] uint4
    int64_t plane_in;        // uint4 per plane
    const uint4* whi;
    const uint4* wlo;
    const float* d;          // [bs][cout]
    const float* noise;      // [noise_bs][ho*wo] or NULL
    const float* noise_weight;
    const float* act_bias;   // [cout] or NULL
    const float* blur;       // [4,4] (up layers)
    int noise_bstride, act;
    uint32_t* out_sp;        // [2][bs][cout/8][ho][wo] uint4, addressed in dwords
    int64_t plane_out;       // uint4 per plane
    const float* s_next;     // [bs][cout]
    float* rgb_out;
    const float* rgb_wt;     // [cout][3]
    const float* rgb_s;      // [bs][cout]
    const float* rgb_bias;   // [3]
    const float* rgb_skip;   // [bs,3,h/2,w/2] or NULL
    const float* rgb_upk;    // [4,4]
    int bs, cin, cout, h, w;
    int tiles_x, tiles_y, ntile;
    const float* zeros;      // >= 64 zero bytes (source of absent noise / skip / bias elements)
    int exp;                 // tuning experiments (E4S_CHAIN_EXP): 1 = no epilogue, 2 = no MFMAs, 4 = no activation DMA
    int walk;                // (conv kernel) bit 0: XCD-aware start offsets, bits 8..: band height of the tile enumeration (sb_common.h)
};

// ------------------------------------------------------------------------------------------------------------------------------------
// Roles.  A workgroup is one LOADER wave and eight COMPUTE waves, one workgroup per CU, walking over tiles blockIdx.x, + gridDim.x, ...
// The loader issues every LDS-DMA (activation chunks [+ weight chunks], epilogue operands), waits for them (vmcnt is per wave: it sees
// nothing but its own DMAs) and publishes them at the chunk barrier; the compute waves never touch vmcnt: barrier, 9 taps of MFMAs from
// LDS, barrier, ..., epilogue (LDS operands in, stores out).  Stage ring: chunk k lives in stage k % NSTAGE; after barrier B_k every
// compute wave has finished chunk k - 1, so the loader refills that stage with chunk k + NSTAGE - 1.  All waves run the same barrier
// sequence.  Ghost chunks past a workgroup's last tile re-load its last tile so that every count stays uniform.
// ------------------------------------------------------------------------------------------------------------------------------------

template <int CB, int NCH, bool WRES, int NSTAGE>
struct ChainCfg {
    static constexpr int TN = CB * 32;
    static constexpr int W4 = 36 * TN;                              // uint4 per weight chunk: [hi/lo][tap][half][TN]
    static constexpr int XS4 = 4 * CT_PATCH;                        // uint4 per activation chunk: [hi/lo][half][patch]
    static constexpr int STAGE4 = XS4 + (WRES ? 0 : W4);
    static constexpr int WRES4 = WRES ? NCH * W4 : 0;
    static constexpr int EP_OFF = (WRES4 + NSTAGE * STAGE4) * 16;   // bytes
    // epilogue operands (floats): noise tile | skip patch | d | s_next | bias | rgb_wt | wsr | flipped up kernel + rgb bias | tile descriptor
    static constexpr int EP_NOISE = 0, EP_SKIP = EP_NOISE + CT_TW * CT_TH, EP_D = EP_SKIP + 576, EP_SN = EP_D + TN, EP_BIAS = EP_SN + TN,
                         EP_RW = EP_BIAS + TN, EP_WSR = EP_RW + 3 * TN, EP_KF = EP_WSR + 3 * TN, EP_DESC = EP_KF + 20, EP_FLOATS = EP_DESC + 4;
    static constexpr int LDS_BYTES = EP_OFF + EP_FLOATS * 4;
    static constexpr int NWPIECE = W4 / 64;                          // DMA pieces of one weight chunk (18 or 36)
    static constexpr int GL = 4 * CT_NPIECE + (WRES ? 0 : NWPIECE);  // DMA instructions of one chunk group
    static_assert(W4 % 64 == 0, "whole pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per CU");
    static_assert(NSTAGE >= 2 && NSTAGE - 1 <= NCH, "prefetch distance");
    static_assert((NSTAGE - 2) * GL <= 63, "vmcnt is a 6-bit counter");
};

// Same-resolution 3x3 single-region StyledConv on split planes.  RGB: the following single-region ToRGB rides in the epilogue;
// OUT_SP: the activation is written as split planes pre-modulated for the next layer (otherwise it is not written at all — the last layer).
template <int CB, int NCH, bool WRES, int NSTAGE, bool RGB, bool OUT_SP>
__global__ __launch_bounds__(CT_NT) void chain_conv_kernel(const ChainParams p) {
    using C = ChainCfg<CB, NCH, WRES, NSTAGE>;
    constexpr int D = NSTAGE - 1;                                          // chunks in flight ahead of the one being computed
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];   // the ONE shared object (a second one makes hipcc drain DMA)
    uint4* lds4 = reinterpret_cast<uint4*>(lds_raw);
    float* epw = reinterpret_cast<float*>(lds_raw + C::EP_OFF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: roles branch on it, DMA destinations (M0) stay in SGPRs
    const int hw = p.h * p.w;

    // ---- layer constants into LDS (plain loads: nothing is in flight yet)
    for (int v = tid; v < C::TN; v += CT_NT) epw[C::EP_BIAS + v] = (p.act_bias && v < p.cout) ? p.act_bias[v] : 0.f;
    if constexpr (RGB) {
        for (int v = tid; v < 3 * C::TN; v += CT_NT) epw[C::EP_RW + v] = (v < 3 * p.cout) ? p.rgb_wt[v] : 0.f;
        if (tid < 16) epw[C::EP_KF + tid] = p.rgb_upk ? p.rgb_upk[15 - tid] : 0.f;
        if (tid < 3) epw[C::EP_KF + 16 + tid] = p.rgb_bias[tid];
    }
    const int stride = gridDim.x, first = walk_offset(blockIdx.x, stride, p.walk & 1);
    const int my_tiles = (p.ntile - first + stride - 1) / stride;
    const int per_img = p.tiles_x * p.tiles_y;

    if (wave == CT_NCW) {
        // =================================================================================== loader
        const int cb8 = p.cin >> 3;
        auto tile_coords = [&](int t, int& b, int& y0, int& x0) {
            b = t / per_img;
            int ty, tx;
            walk_tile_xy(t - b * per_img, p.tiles_x, p.tiles_y, p.walk >> 8, ty, tx);
            y0 = ty * CT_TH;
            x0 = tx * CT_TW;
        };
        auto tile_of = [&](int k) {
            int i = k / NCH;
            i = i < my_tiles ? i : my_tiles - 1;
            return first + i * stride;
        };
        lds_byte* const lds_b = (lds_byte*)lds_raw;                            // LDS as LDS (32-bit addresses)
        constexpr unsigned K_WRES = C::WRES4 * 16, K_STAGE = C::STAGE4 * 16, K_XS = C::XS4 * 16, K_W = C::W4 * 16, K_EP = C::EP_OFF;
        constexpr int K_NWPIECE = C::NWPIECE, K_EP_NOISE = C::EP_NOISE, K_EP_SKIP = C::EP_SKIP, K_EP_DESC = C::EP_DESC, K_EP_D = C::EP_D,
                      K_EP_SN = C::EP_SN, K_EP_WSR = C::EP_WSR, K_EP_RW = C::EP_RW, K_TN = C::TN;
        const unsigned zero_off = (unsigned)(2 * p.plane_in * 16);             // the 16 zero bytes behind the two input planes
        auto issue_weights = [&](int c, unsigned dst_off) __attribute__((always_inline)) {
            // chunk c of both slabs: LDS image [hl][tap][half][TN] = global [hl-slab][chunk][tap][half][cout] rows (cout == TN here)
#pragma unroll
            for (int piece = 0; piece < K_NWPIECE; ++piece) {
                const int hl = piece * 64 / (18 * K_TN);                       // (a piece never straddles the two slabs: 18 TN is a multiple of 64)
                const unsigned voff = (unsigned)(((c * 18 * K_TN) + piece * 64 - hl * 18 * K_TN + lane) * 16);
                dma16(hl ? p.wlo : p.whi, voff, lds_b + dst_off + piece * 1024);
            }
        };
        auto issue_group = [&](int k) __attribute__((always_inline)) {
            const unsigned st_off = K_WRES + (unsigned)(k % NSTAGE) * K_STAGE;
            const int c = k % NCH;
            int b, y0, x0;
            tile_coords(tile_of(k), b, y0, x0);
            if (!(p.exp & 4)) {
                // (plane, half) = combo: its block of this chunk starts cbase(combo) uint4 into the tensor — scalar; a lane adds its pixel
                const unsigned cb0 = (unsigned)((b * cb8 + 2 * c) * hw);
#pragma unroll
                for (int j = 0; j < CT_NPIECE; ++j) {                          // this lane's patch pixel of piece j: the same for the 4 (plane, half)
                    const int e = j * 64 + lane;
                    const int py = e / CT_PW, px = e - py * CT_PW;
                    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                    const bool inb = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
                    const unsigned pix = (unsigned)(gy * p.w + gx);
                    if (e < CT_PATCH) {
#pragma unroll
                        for (int combo = 0; combo < 4; ++combo) {
                            const unsigned cbase = (unsigned)((combo >> 1) * p.plane_in) + cb0 + (unsigned)((combo & 1) * hw);
                            const unsigned voff = inb ? (cbase + pix) * 16u : zero_off;
                            dma16(p.xsp, voff, lds_b + st_off + (combo * CT_PATCH + j * 64) * 16);
                        }
                    }
                }
            }
            if constexpr (!WRES) issue_weights(c, st_off + K_XS);
        };
        int cur_b = -1;
        auto tile_setup = [&](int t, bool with_ep) __attribute__((always_inline)) {
            int b, y0, x0;
            tile_coords(t, b, y0, x0);
            if (with_ep) {                                                     // noise tile (8 pieces) + skip patch (9 pieces)
#pragma unroll
                for (int piece = 0; piece < 8; ++piece) {
                    const int e = piece * 64 + lane;
                    const unsigned voff = (unsigned)((b * p.noise_bstride + (y0 + (e >> 5)) * p.w + x0 + (e & 31)) * 4);
                    dma4(p.noise ? p.noise : p.zeros, p.noise ? voff : 0u, lds_b + K_EP + (K_EP_NOISE + piece * 64) * 4);
                }
                if constexpr (RGB) {
                    const int hs = p.h >> 1, wsk = p.w >> 1;
#pragma unroll
                    for (int piece = 0; piece < 9; ++piece) {
                        const int e = piece * 64 + lane;
                        const int o = e / (CT_SKW * CT_SKH);
                        const int r = e - o * (CT_SKW * CT_SKH);
                        const int sy = r / CT_SKW, sx = r - sy * CT_SKW;
                        const int iy = (y0 >> 1) - 1 + sy, ix = (x0 >> 1) - 1 + sx;
                        const bool ok = p.rgb_skip && e < CT_SKIP && iy >= 0 && iy < hs && ix >= 0 && ix < wsk;
                        // out-of-image / absent elements come from the zero block: two exec-masked DMAs, together they fill the piece
                        if (ok) dma4(p.rgb_skip, (unsigned)((((b * 3 + o) * hs + iy) * wsk + ix) * 4), lds_b + K_EP + (K_EP_SKIP + piece * 64) * 4);
                        else dma4(p.zeros, 0u, lds_b + K_EP + (K_EP_SKIP + piece * 64) * 4);
                    }
                }
            }
            if (lane == 0) {
                int* desc = reinterpret_cast<int*>(epw + K_EP_DESC);
                desc[0] = b; desc[1] = y0; desc[2] = x0;
            }
            if (b != cur_b) {                 // per-sample tables (plain loads: the compiler drains the loader's DMAs here — at most bs times)
                cur_b = b;
                for (int v = lane; v < K_TN; v += 64) {
                    const bool ok = v < p.cout;
                    epw[K_EP_D + v] = ok ? p.d[(size_t)b * p.cout + v] : 0.f;
                    if constexpr (OUT_SP) epw[K_EP_SN + v] = ok ? p.s_next[(size_t)b * p.cout + v] : 0.f;
                    if constexpr (RGB) {
                        const float rs = ok ? p.rgb_s[(size_t)b * p.cout + v] : 0.f;
#pragma unroll
                        for (int o = 0; o < 3; ++o) epw[K_EP_WSR + v * 3 + o] = epw[K_EP_RW + v * 3 + o] * rs;
                    }
                }
            }
        };
        CH_BARRIER();                                                          // the layer constants above are in LDS (read by tile_setup)
        if constexpr (WRES) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) issue_weights(c, (unsigned)c * K_W);
        }
        tile_setup(first, true);
#pragma unroll
        for (int k = 0; k < D; ++k) issue_group(k);
        const int total = my_tiles * NCH;
#pragma unroll 1
        for (int k = 0; k < total; ++k) {
            // chunk k has landed: everything this wave issued except the D - 1 younger groups (epilogue operands are older than the
            // NCH - 1 >= D - 1 groups issued since, so they have landed by a tile's last chunk as well)
            wait_vm((D - 1) * C::GL);
            CH_BARRIER();
            if (k > 0 && k % NCH == 0) tile_setup(tile_of(k), true);
            issue_group(k + D);
        }
        CH_WAIT_VM(0);   // ghost chunks must land before the workgroup's LDS is released
        return;
    }

    // ======================================================================================= compute waves
    const int l5 = lane & 31, khalf = lane >> 5;
    const float nw = p.noise ? p.noise_weight[0] : 0.f;
    const int xoff0 = (2 * wave) * CT_PW + l5;
    if constexpr (OUT_SP) {                                                    // the zero element behind the output planes (the consumer's padding source)
        if (blockIdx.x == 0 && tid < 4) p.out_sp[(size_t)p.plane_out * 8 + tid] = 0u;
    }
    CH_BARRIER();                                                              // (pairs with the loader's first barrier)
#pragma unroll 1
    for (int ti = 0; ti < my_tiles; ++ti) {
        f32x16 acc[CB][2];
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            const int k = ti * NCH + c;
            CH_BARRIER();
            // One opaque base register per operand and chunk, everything else as instruction immediates: LDS is 160 KB here, absolute
            // addresses past 64 KB do not fit a ds_read offset field, and left alone hipcc keeps a register per distinct absolute address.
            unsigned xb_i = (unsigned)(C::WRES4 + (k % NSTAGE) * C::STAGE4 + khalf * CT_PATCH + xoff0);
            unsigned wb_i = (WRES ? (unsigned)(c * C::W4) : (unsigned)(C::WRES4 + (k % NSTAGE) * C::STAGE4 + C::XS4)) + (unsigned)(khalf * C::TN + l5);
            asm volatile("" : "+v"(xb_i), "+v"(wb_i));
            const uint4* xs = lds4 + xb_i;           // this lane's patch origin in the hi plane of its half; + 2 * PATCH = lo plane
            const uint4* whalf = lds4 + wb_i;
            // one tap of fragments is fetched ahead of the MFMAs that use it; the scheduling barriers keep hipcc from hoisting all nine
            // taps' LDS reads to the top of the chunk
            uint4 bh[2][2], bl[2][2], ah[2][CB], al[2][CB];
            auto fetch = [&](int tap, int slot) __attribute__((always_inline)) {
                const int toff = (tap / 3) * CT_PW + (tap % 3);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    bh[slot][q] = xs[q * CT_PW + toff];
                    bl[slot][q] = xs[2 * CT_PATCH + q * CT_PW + toff];
                }
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    ah[slot][i] = whalf[tap * 2 * C::TN + i * 32];
                    al[slot][i] = whalf[18 * C::TN + tap * 2 * C::TN + i * 32];
                }
            };
            if (p.exp & 2) continue;
            fetch(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cs = tap & 1;
                if (tap + 1 < 9) fetch(tap + 1, cs ^ 1);
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[cs][i]), __builtin_bit_cast(bf16x8, bh[cs][q]), acc[i][q], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[cs][i]), __builtin_bit_cast(bf16x8, bl[cs][q]), acc[i][q], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[cs][i]), __builtin_bit_cast(bf16x8, bh[cs][q]), acc[i][q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue: every operand is in LDS; only stores go to memory.  Same formulas, in the same order, as modconv_sb.hip's.
        if (p.exp & 1) continue;
        int b, y0, x0;
        {
            const int* desc = reinterpret_cast<const int*>(epw + C::EP_DESC);
            b = __builtin_amdgcn_readfirstlane(desc[0]); y0 = __builtin_amdgcn_readfirstlane(desc[1]); x0 = __builtin_amdgcn_readfirstlane(desc[2]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // opaque base, immediates after it — and opaque PER pixel block: the per-channel tables are the same for both blocks, and given
            // the chance hipcc keeps all of them in registers across the two
            unsigned ep_i = (unsigned)(C::EP_OFF / 4);
            asm volatile("" : "+v"(ep_i));
            const float* ep = reinterpret_cast<const float*>(lds_raw) + ep_i;
            const int row = 2 * wave + q;
            const int oy = y0 + row, ox = x0 + l5;
            const size_t opix = (size_t)oy * p.w + ox;
            const float nz = __fmul_rn(nw, ep[C::EP_NOISE + row * CT_TW + l5]);
            float rgbadd[3] = {0.f, 0.f, 0.f};
            if constexpr (RGB) {
                const int iy0 = (oy - 1) >> 1, ix0 = (ox - 1) >> 1;
                const int ky0 = 2 * iy0 + 2 - oy, kx0 = 2 * ix0 + 2 - ox;
                const int sy0 = iy0 - ((y0 >> 1) - 1), sx0 = ix0 - ((x0 >> 1) - 1);      // skip patch coordinates
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    float u = ep[C::EP_KF + 16 + o];
                    if (p.rgb_skip) {
#pragma unroll
                        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                            for (int tx = 0; tx < 2; ++tx)      // out-of-image skip pixels were staged as zeros: fma with 0 leaves u unchanged
                                u = __builtin_fmaf(ep[C::EP_SKIP + (o * CT_SKH + sy0 + ty) * CT_SKW + sx0 + tx], ep[C::EP_KF + (ky0 + 2 * ty) * 4 + kx0 + 2 * tx], u);
                    }
                    rgbadd[o] = u;
                }
            }
            float rgb0 = 0.f, rgb1 = 0.f, rgb2 = 0.f;
#pragma unroll
            for (int i = 0; i < CB; ++i) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int n0 = i * 32 + 8 * r4 + 4 * khalf;
                    const float4 d4 = *reinterpret_cast<const float4*>(ep + C::EP_D + n0);
                    const float4 b4 = *reinterpret_cast<const float4*>(ep + C::EP_BIAS + n0);
                    const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
                    float v4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = __fadd_rn(__fadd_rn(__fmul_rn(acc[i][q][4 * r4 + e], dd[e]), nz), bb[e]);
                        if (p.act) v = __fmul_rn(v > 0.f ? v : __fmul_rn(v, 0.2f), 1.41421356237309515f);
                        v4[e] = v;
                        if constexpr (RGB) {
                            const float* wq = ep + C::EP_WSR + (n0 + e) * 3;
                            rgb0 = __builtin_fmaf(v, wq[0], rgb0);
                            rgb1 = __builtin_fmaf(v, wq[1], rgb1);
                            rgb2 = __builtin_fmaf(v, wq[2], rgb2);
                        }
                    }
                    if constexpr (OUT_SP) {
                        const float4 s4 = *reinterpret_cast<const float4*>(ep + C::EP_SN + n0);
                        unsigned h0, l0, h1, l1;
                        split2(__fmul_rn(v4[0], s4.x), __fmul_rn(v4[1], s4.y), h0, l0);
                        split2(__fmul_rn(v4[2], s4.z), __fmul_rn(v4[3], s4.w), h1, l1);
                        // 8-channel block n0 / 8 of this pixel: this half-wave owns bytes [8 khalf, 8 khalf + 8) of its 16
                        const size_t o8 = (((size_t)b * (p.cout >> 3) + (n0 >> 3)) * hw + opix) * 2 + khalf;
                        uint2* osp = reinterpret_cast<uint2*>(p.out_sp);
                        osp[o8] = make_uint2(h0, h1);
                        osp[(size_t)p.plane_out * 2 + o8] = make_uint2(l0, l1);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one group of four channels at a time
                }
            }
            if constexpr (RGB) {
                rgb0 += __shfl_xor(rgb0, 32, 64);
                rgb1 += __shfl_xor(rgb1, 32, 64);
                rgb2 += __shfl_xor(rgb2, 32, 64);
                if (khalf == 0) {
                    float* ro = p.rgb_out + (size_t)b * 3 * hw + opix;
                    ro[0] = rgb0 + rgbadd[0];
                    ro[(size_t)hw] = rgb1 + rgbadd[1];
                    ro[2 * (size_t)hw] = rgb2 + rgbadd[2];
                }
            }
        }
    }
}

static int chain_grid(int ntile) {
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return ntile < ncu ? ntile : ncu;       // one persistent workgroup per CU (LDS-bound)
}

template <int CB, int NCH, bool WRES, int NSTAGE, bool RGB, bool OUT_SP>
int launch_chain_conv(ChainParams& p, hipStream_t st) {
    using C = ChainCfg<CB, NCH, WRES, NSTAGE>;
    auto kern = &chain_conv_kernel<CB, NCH, WRES, NSTAGE, RGB, OUT_SP>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (attr != hipSuccess) return fail((int)attr, "chain_conv3x3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL(kern, dim3(chain_grid(p.ntile)), dim3(CT_NT), C::LDS_BYTES, st, p);
    return check_launch("chain_conv3x3");
}


// fp32 activation -> split planes, modulated by s[b][c]: out[hl][b][c/8][p][c%8] = split(x[b][c][p] * s[b][c])
__global__ __launch_bounds__(256) void to_split_planes_kernel(uint4* __restrict__ out, int64_t plane, const float* __restrict__ x, const float* __restrict__ s, int bs,
                                                              int c, int hw, int x_nhwc) {
    const int64_t total = (int64_t)bs * (c >> 3) * hw;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[2 * plane] = make_uint4(0u, 0u, 0u, 0u);      // the zero element behind the planes (consumers' padding source)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int pix = (int)(i % hw);
        const int64_t r = i / hw;
        const int blk = (int)(r % (c >> 3)), b = (int)(r / (c >> 3));
        float v[8];
        if (x_nhwc) {
            const float4 a = *reinterpret_cast<const float4*>(x + i * 8), bq = *reinterpret_cast<const float4*>(x + i * 8 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = x[((size_t)b * c + blk * 8 + e) * hw + pix];
        }
        const float* sb = s + (size_t)b * c + blk * 8;
        unsigned h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split2(__fmul_rn(v[2 * e], sb[2 * e]), __fmul_rn(v[2 * e + 1], sb[2 * e + 1]), h[e], l[e]);
        out[i] = make_uint4(h[0], h[1], h[2], h[3]);
        out[plane + i] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

}  // namespace

extern "C" int e4s_to_split_planes(uint16_t* out_sp, const float* x, const float* s, int bs, int c, int h, int w, int x_nhwc, void* stream) {
    E4S_REQUIRE(out_sp && x && s, "to_split_planes: null tensor");
    E4S_REQUIRE(bs >= 0 && c >= 8 && c % 8 == 0 && h >= 1 && w >= 1, "to_split_planes: bad size (channels in blocks of 8)");
    E4S_REQUIRE((((uintptr_t)out_sp | (uintptr_t)x) & 15) == 0, "to_split_planes: tensors must be 16-byte aligned");
    if (bs == 0) return 0;
    const int64_t total = (int64_t)bs * (c / 8) * h * w;
    const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
    hipLaunchKernelGGL(to_split_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint4*>(out_sp), total, x, s, bs, c, h * w, x_nhwc);
    return check_launch("to_split_planes");
}


static int fill_chain_params(ChainParams& p, const E4sChainLayer* L, int ho, int wo, const char* who) {
    memset(&p, 0, sizeof(p));
    p.xsp = reinterpret_cast<const uint4*>(L->x_sp);
    p.plane_in = (int64_t)L->bs * (L->cin / 8) * L->h * L->w;
    p.whi = reinterpret_cast<const uint4*>(L->whi); p.wlo = reinterpret_cast<const uint4*>(L->wlo);
    p.d = L->d; p.noise = L->noise; p.noise_weight = L->noise_weight; p.act_bias = L->act_bias; p.act = L->act;
    p.noise_bstride = (L->noise && L->noise_bs > 1) ? ho * wo : 0;
    p.out_sp = reinterpret_cast<uint32_t*>(L->out_sp);
    p.plane_out = (int64_t)L->bs * (L->cout / 8) * ho * wo;
    p.s_next = L->s_next;
    p.rgb_out = L->rgb_out; p.rgb_wt = L->rgb_wt; p.rgb_s = L->rgb_s; p.rgb_bias = L->rgb_bias; p.rgb_skip = L->rgb_skip; p.rgb_upk = L->rgb_up_kernel;
    p.bs = L->bs; p.cin = L->cin; p.cout = L->cout; p.h = L->h; p.w = L->w;
#ifdef E4S_PHASE_PROF
    { const char* e = getenv("E4S_CHAIN_EXP"); p.exp = e ? atoi(e) : 0; }   // (tuning build only)
#else
    p.exp = 0;
#endif
    static const float* zeros = [] {
        void* ptr = nullptr;
        return hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_zero16)) == hipSuccess ? static_cast<const float*>(ptr) : nullptr;
    }();
    if (!zeros) return fail(E4S_ERR_ARG, "%s: cannot resolve the zero block", who);
    p.zeros = zeros;
    return 0;
}

extern "C" int e4s_chain_conv3x3(const E4sChainLayer* L, void* stream) {
    E4S_REQUIRE(L, "chain_conv3x3: null layer");
    E4S_REQUIRE(L->x_sp && L->whi && L->wlo && L->d, "chain_conv3x3: null tensor");
    E4S_REQUIRE(L->out_sp || L->rgb_out, "chain_conv3x3: nothing to produce (neither split-plane output nor fused ToRGB)");
    E4S_REQUIRE(!L->out_sp || L->s_next, "chain_conv3x3: split-plane output needs the next layer's modulation");
    E4S_REQUIRE(!L->rgb_out || (L->rgb_wt && L->rgb_s && L->rgb_bias && (!L->rgb_skip || L->rgb_up_kernel)), "chain_conv3x3: incomplete fused-ToRGB arguments");
    E4S_REQUIRE(L->bs >= 0 && L->bs <= 32768 && L->h % CT_TH == 0 && L->w % CT_TW == 0 && L->h >= CT_TH && L->w >= CT_TW,
                "chain_conv3x3: %d x %d is not a multiple of the %d x %d tile", L->h, L->w, CT_TH, CT_TW);
    E4S_REQUIRE(!L->noise || (L->noise_weight && (L->noise_bs == 1 || L->noise_bs == L->bs)), "chain_conv3x3: noise needs its weight and batch 1 or bs");
    E4S_REQUIRE((((uintptr_t)L->x_sp | (uintptr_t)L->whi | (uintptr_t)L->wlo | (uintptr_t)L->out_sp) & 15) == 0, "chain_conv3x3: tensors must be 16-byte aligned");
    if (L->bs == 0) return 0;
    ChainParams p;
    if (int rc = fill_chain_params(p, L, L->h, L->w, "chain_conv3x3")) return rc;
    p.tiles_x = L->w / CT_TW; p.tiles_y = L->h / CT_TH;
    p.ntile = p.tiles_x * p.tiles_y * L->bs;
    static const int walk = [] { const char* e = getenv("E4S_WALK"); return e ? atoi(e) : (1 | (8 << 8)); }();     // (E4S_WALK=0: every eighth tile, row-major)
    p.walk = walk;
    hipStream_t st = (hipStream_t)stream;
    const bool rgb = L->rgb_out != nullptr, osp = L->out_sp != nullptr;
    // the shapes of the chain (Generator(1024): 64 -> 64 at 512 x 512, 32 -> 32 at 1024 x 1024); anything else stays on e4s_region_modconv3x3_sb
    if (L->cin == 32 && L->cout == 32 && rgb && !osp) return launch_chain_conv<1, 2, true, 3, true, false>(p, st);    // last layer: image only
    if (L->cin == 64 && L->cout == 64 && rgb && osp) return launch_chain_conv<2, 4, false, 2, true, true>(p, st);
    return fail(E4S_ERR_ARG, "chain_conv3x3: no kernel for %d -> %d channels, rgb %d, out %d (built: 32 -> 32 + ToRGB, 64 -> 64 + ToRGB + split-plane output)",
                L->cin, L->cout, (int)rgb, (int)osp);
}
