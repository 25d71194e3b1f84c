// a3/a4, the chain's single-region up layers (256 -> 512, 512 -> 1024) in the HALF-COMPOSED form (round 4).
//
// Reference: ModulatedConv2d.forward upsample branch (models/stylegan2/model.py:287-301: conv_transpose2d stride 2, then Blur = upfirdn2d with the
// 4 x 4 kernel outer([1,3,3,1]) / 16, pad (1,1)) + NoiseInjection / FusedLeakyReLU of StyledConv.forward (:417-421), single-region case.
//
// Why.  The fused kernel of modconv_upfused.hip evaluates the transposed conv at 1x its MACs and then pays for the blur in its epilogue: the
// pre-blur tile goes through LDS in passes of 8-16 channels (barriers in between), every output costs 16 FMAs and the stores leave 4 bytes per
// lane — 54 % of a workgroup's life (phase marks, round 3) for a layer whose MFMAs are a fifth of its time.  The blur kernel is an outer
// product, blur[r][c] = kv[r] * kh[c], so the two directions can be treated differently:
//   * VERTICALLY the blur is composed into the weights (as DESIGN.md section 2 does in both directions for the masked layers): for output row parity
//     a, Wv[a][dy][kx] = sum_ky kv'[ky + 2 dy + 1 - a] W[ky][kx], dy in {-1, 0, 1} — 2 x 9 taps over input rows m-1, m, m+1 instead of 9.  The MFMAs then
//     produce V[2m + a][2b + pb], the vertically blurred pre-blur value: 2x the MACs of the bare transposed conv, and NO vertical tile overlap;
//   * HORIZONTALLY a lane (= one position b of a row of 16 positions = one DPP row of 16 lanes) holds V at columns 2b, 2b + 1; the four-tap
//     filter for its two output columns 2b, 2b + 1 needs columns 2b - 1 .. 2b + 3 = its own two values, the left neighbour's second and the right
//     neighbour's two: three DPP row shifts per register.  8 FMAs per two outputs instead of 32, no LDS, no barrier.
// After that every lane holds FINISHED outputs (2 rows x 2 columns x 16 channels): demodulation, noise, bias, leaky ReLU, the next layer's
// modulation and the bf16 hi / lo split happen in registers, the two half-waves exchange halves (v_permlane32_swap) so that each lane owns all 8
// channels of one pixel, and the tile leaves in 16-byte stores.  The epilogue has no LDS traffic and no barrier at all.
// MACs: 2 x (16 / 14) = 2.29x the algorithmic ones (fused kernel: 1.31x) — on a layer that was nowhere near MFMA-bound.
//
// Staging: LDS-DMA of split planes (modconv_chain.hip) issued from inline asm by all eight waves; ring = two activation chunks (16 channels each) +
// two weight UNITS (unit = (chunk, row parity): [hi|lo][9 taps][half][32 co] = 18 KB), 76 KB per workgroup -> two workgroups per CU, so that one's
// epilogue (VALU + stores) runs under the other's K loop (MFMA).
#include <stdlib.h>

#include "common.h"
#include "sb_common.h"

using namespace e4s;

namespace {

struct UpHcParams {
    uint4* out;               // split planes [2][bs][cout/8][2h][2w] uint4 (+ 16 zero bytes behind)
    const uint4* x;           // split planes [2][bs][cin/8][h][w] uint4, already carrying this layer's modulation
    const uint4* whi;         // [2 row parities][cin/16][9 taps = (dy + 1) * 3 + kx][2 halves][cout] uint4 (e4s_modconv_prep_weights_hc)
    const uint4* wlo;
    const float* d;           // [bs][cout]
    const float* blur;        // [4][4], rank 1 (checked by the caller)
    const float* noise;       // [noise_bs][2h * 2w] or NULL
    const float* noise_weight;
    const float* act_bias;    // [cout] or NULL
    const float* s_next;      // [bs][cout]
    const float* zeros;       // >= 64 zero bytes
    int noise_bstride, act;
    int bs, cin, cout, h, w;
    int tiles_x, tiles_y;
    int ntile, ncot;                 // (persistent form) tiles_x * tiles_y * bs, cout / 32
    int exp;                         // tuning experiments of the -DE4S_PHASE_PROF build (E4S_HC_EXP): 1 = no epilogue, 2 = no MFMAs, 4 = no output stores
    int rev;                         // (persistent form) walk the tiles last to first: see e4s_modconv_up_hc
    int walk;                        // (persistent form) bit 0: XCD-aware start offsets, bits 8..: band height of the tile enumeration (sb_common.h)
    int64_t plane_in, plane_out;     // uint4 per plane
};

constexpr int HC_T = 16;                         // positions per tile side (16 rows x 16 columns; wave v owns rows 2v, 2v + 1)
constexpr int HC_STEP = HC_T - 2;                // 14 new positions per tile in x (the outer two only feed their neighbours' filters)
constexpr int HC_PW = HC_T + 1, HC_PH = HC_T + 2;
constexpr int HC_PATCH = HC_PW * HC_PH;          // 306 patch pixels: rows m0 - 1 .. m0 + 16, columns p0x - 1 .. p0x + 15
constexpr int HC_NT = 512;
constexpr int HC_XB4 = 4 * HC_PATCH;             // uint4 per activation buffer: [hi|lo][half][306]
constexpr int HC_W4 = 36 * 32;                   // uint4 per weight unit: [hi|lo][tap 9][half][32]
constexpr int HC_BODY = (2 * HC_XB4 + 2 * HC_W4) * 16;      // 76 032
constexpr int HC_EP_D = 0, HC_EP_B = 64, HC_EP_S = 128, HC_EP_FLOATS = 192;   // (a table DMA writes 64 lanes x 4 bytes: the tables sit 64 floats apart)
constexpr int HC_LDS = HC_BODY + HC_EP_FLOATS * 4;          // 76 800: two workgroups per CU
constexpr int HC_G = 3;                          // DMA requests per wave for one activation chunk, and for one weight unit
static_assert(HC_LDS <= 80 * 1024, "two workgroups per CU");

// v_mov_b32_dpp with a row shift: lane i of each row of 16 receives lane i -/+ 1 (0 at the row's ends)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_ROW_SHL1 = 0x101;              // lane i <- lane i + 1
constexpr int DPP_ROW_SHR1 = 0x111;              // lane i <- lane i - 1

// One unit (16-channel chunk x row parity PAR) of the K loop: nine taps (dyi, kx) x three split-bf16 MFMAs into accs[2 PAR + (kx & 1)].  `xs` = this lane's
// patch element (its position's row m - 1, column b - 1) in the hi plane of its K half (+ 2 * HC_PATCH = lo plane), `whalf` = its weight fragment of tap 0.
// LDS traffic is what these kernels are short of (tuning build: DMA, MFMA and epilogue times ADD — the compute waves' fragment reads and the DMA writes share one
// LDS port), so the activation fragment of input row dyi is read ONCE: taps kx = 0 and kx = 1 both multiply x[m - 1 + dyi][b], and tap kx = 2 multiplies
// x[..][b - 1] = the LEFT neighbour lane's fragment, a DPP row shift (lane ptx = 0 receives zeros there: its kx = 2 term only enters column parity 0 of a
// position whose outputs are discarded, and its right neighbour's filter reads its parity-1 value only) — 24 LDS reads per unit instead of 36.
__device__ __forceinline__ uint4 dpp_row_shr1(uint4 v) {
    uint4 r;
    r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, DPP_ROW_SHR1, 0xf, 0xf, true);
    r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, DPP_ROW_SHR1, 0xf, 0xf, true);
    r.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, DPP_ROW_SHR1, 0xf, 0xf, true);
    r.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, DPP_ROW_SHR1, 0xf, 0xf, true);
    return r;
}
template <int PAR>
__device__ __forceinline__ void hc_unit(f32x16 (&accs)[4], const uint4* xs, const uint4* whalf) {
    uint4 bh[2], bl[2], ah[2], al[2];
    auto fetch_b = [&](int dyi, int slot) __attribute__((always_inline)) {
        bh[slot] = xs[dyi * HC_PW + 1];
        bl[slot] = xs[2 * HC_PATCH + dyi * HC_PW + 1];
    };
    auto fetch_a = [&](int tap, int slot) __attribute__((always_inline)) {
        ah[slot] = whalf[tap * 64];
        al[slot] = whalf[18 * 32 + tap * 64];
    };
    fetch_b(0, 0);
    fetch_a(0, 0);
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi) {
        const int bs = dyi & 1;
        if (dyi + 1 < 3) fetch_b(dyi + 1, bs ^ 1);
        const uint4 sh = dpp_row_shr1(bh[bs]), sl = dpp_row_shr1(bl[bs]);          // column b - 1
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tap = dyi * 3 + kx, cs = tap & 1;
            if (tap + 1 < 9) fetch_a(tap + 1, cs ^ 1);
            const uint4 xh = kx == 2 ? sh : bh[bs], xl = kx == 2 ? sl : bl[bs];
            const int ai = 2 * PAR + (kx & 1);
            accs[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[cs]), __builtin_bit_cast(bf16x8, xh), accs[ai], 0, 0, 0);
            accs[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[cs]), __builtin_bit_cast(bf16x8, xl), accs[ai], 0, 0, 0);
            accs[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[cs]), __builtin_bit_cast(bf16x8, xh), accs[ai], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

E4S_PROF_DECL(g_prof_hc)

// The register epilogue shared by both kernels: horizontal four-tap filter across the lanes of a DPP row, demodulation, noise, bias, leaky ReLU, the next
// layer's modulation, bf16 hi / lo split, half exchange, 16-byte stores.  `accs[2 a + pb]` = V[2 pm + a][2 pbx + pb] of this lane's position for the 16 channels
// 8 g + 4 khalf + j of its registers 4 g + j; the d / bias / s_next tables of the workgroup's 32 channels sit at LDS byte `ep_off` (64 floats apart).
__device__ __forceinline__ void hc_epilogue(const UpHcParams& p, const f32x16 (&accs)[4], const unsigned char* lds_raw, int ep_off, const float2 (&nz)[2], bool lane_ok,
                                            int b, int co0, int pm, int pbx, int khalf) {
    const int ho = 2 * p.h, wo = 2 * p.w;
    // kh'[u] = flipped horizontal factor: out[ox] = sum_u kh'[u] V[ox - 1 + u]
    float khf[4];
    {
        float cs[4], S = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { cs[cc] = p.blur[cc] + p.blur[4 + cc] + p.blur[8 + cc] + p.blur[12 + cc]; S += cs[cc]; }
        const float rs = 1.f / sqrtf(S);
#pragma unroll
        for (int u = 0; u < 4; ++u) khf[u] = cs[3 - u] * rs;
    }
    const float nw = p.noise ? p.noise_weight[0] : 0.f;
    const float neg = p.act ? 0.2f : 1.f, gain = p.act ? 1.41421356237309515f : 1.f;
    const size_t opl = (size_t)ho * wo;
    // uint4 index of this lane's pixel after the half exchange (lower half-wave: column 2b, upper: 2b + 1) in 8-channel block 0 of this workgroup, row parity 0
    const size_t o_base = ((size_t)b * (p.cout >> 3) + (size_t)(co0 >> 3)) * opl + (size_t)(2 * pm) * wo + (size_t)(2 * pbx + khalf);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // registers 4g .. 4g + 3 = channels 8g + 4 khalf + (0..3)
        unsigned ep_i = (unsigned)(ep_off / 4 + 8 * g + 4 * khalf);
        asm volatile("" : "+v"(ep_i));
        const float* ep = reinterpret_cast<const float*>(lds_raw) + ep_i;
        const float4 d4 = *reinterpret_cast<const float4*>(ep + HC_EP_D);
        const float4 b4 = *reinterpret_cast<const float4*>(ep + HC_EP_B);
        const float4 s4 = *reinterpret_cast<const float4*>(ep + HC_EP_S);
        const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const float nze = __fmul_rn(nw, nz[pa].x), nzo = __fmul_rn(nw, nz[pa].y);
            float ue[4], uo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z0 = accs[2 * pa][4 * g + j], z1 = accs[2 * pa + 1][4 * g + j];
                const float l1 = dpp_row<DPP_ROW_SHR1>(z1);                // column 2b - 1
                const float r0 = dpp_row<DPP_ROW_SHL1>(z0);                // column 2b + 2
                const float r1 = dpp_row<DPP_ROW_SHL1>(z1);                // column 2b + 3
                float oe = __fmul_rn(khf[0], l1);
                oe = __builtin_fmaf(khf[1], z0, oe);
                oe = __builtin_fmaf(khf[2], z1, oe);
                oe = __builtin_fmaf(khf[3], r0, oe);
                float oo = __fmul_rn(khf[0], z0);
                oo = __builtin_fmaf(khf[1], z1, oo);
                oo = __builtin_fmaf(khf[2], r0, oo);
                oo = __builtin_fmaf(khf[3], r1, oo);
                float ve = __builtin_fmaf(oe, dd[j], bb[j]) + nze, vo = __builtin_fmaf(oo, dd[j], bb[j]) + nzo;
                ve = fmaxf(ve, ve * neg) * gain;     // leaky relu 0.2 (max picks v for v >= 0, 0.2 v otherwise)
                vo = fmaxf(vo, vo * neg) * gain;
                ue[j] = __fmul_rn(ve, sn[j]);
                uo[j] = __fmul_rn(vo, sn[j]);
            }
            unsigned he[2], le[2], ho_[2], lo_[2];
            split2(ue[0], ue[1], he[0], le[0]);
            split2(ue[2], ue[3], he[1], le[1]);
            split2(uo[0], uo[1], ho_[0], lo_[0]);
            split2(uo[2], uo[3], ho_[1], lo_[1]);
            // Half exchange (guide T21): the lower half-wave keeps its column-2b values and receives the upper half-wave's (channels + 4 .. + 7 of the same pixel);
            // the upper half-wave receives the lower one's column-(2b + 1) values and keeps its own: every lane then holds the 8 channels = 16 bytes of ONE pixel.
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                auto r = __builtin_amdgcn_permlane32_swap(he[q], ho_[q], false, false);
                he[q] = r[0]; ho_[q] = r[1];
                auto r2 = __builtin_amdgcn_permlane32_swap(le[q], lo_[q], false, false);
                le[q] = r2[0]; lo_[q] = r2[1];
            }
#ifdef E4S_PHASE_PROF
            if ((p.exp & 4) && he[0] != 0x12345678u) continue;        // experiment: everything but the stores
#endif
            if (lane_ok) {
                const size_t o4 = o_base + (size_t)g * opl + (size_t)pa * wo;
                p.out[o4] = make_uint4(he[0], he[1], ho_[0], ho_[1]);
                p.out[(size_t)p.plane_out + o4] = make_uint4(le[0], le[1], lo_[0], lo_[1]);
            }
        }
    }
}


__global__ __launch_bounds__(HC_NT, 4) void up_hc_kernel(const UpHcParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint4* lds4 = reinterpret_cast<const uint4*>(lds_raw);
    const float* epw = reinterpret_cast<const float*>(lds_raw + HC_BODY);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l5 = lane & 31, khalf = lane >> 5;
    E4S_PROF_MARK(g_prof_hc, 0);
    const int tyt = blockIdx.x / p.tiles_x, txt = blockIdx.x - tyt * p.tiles_x;
    const int m0 = tyt * HC_T, p0x = txt * HC_STEP - 1;
    const int co0 = blockIdx.y * 32;
    const int b = blockIdx.z;
    const int hw = p.h * p.w, ho = 2 * p.h, wo = 2 * p.w;
    const int nchunk = p.cin >> 4, cb8 = p.cin >> 3;
    const unsigned zero_off = (unsigned)(2 * p.plane_in * 16);             // the 16 zero bytes behind the two input planes

    // this lane's position and the noise of its 2 x 2 outputs (plain loads, the oldest requests of the wave: the compiler waits for them at their
    // first use, in the epilogue, when every DMA has long landed)
    const int pty = 2 * wave + (l5 >> 4), ptx = l5 & 15;
    const int pm = m0 + pty, pbx = p0x + ptx;
    const bool lane_ok = ptx >= 1 && ptx <= HC_STEP && pbx < p.w && pm < p.h;
    float2 nz[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    if (p.noise && lane_ok) {
        const float* np = p.noise + (size_t)b * p.noise_bstride + (size_t)(2 * pm) * wo + 2 * pbx;
        nz[0] = *reinterpret_cast<const float2*>(np);
        nz[1] = *reinterpret_cast<const float2*>(np + wo);
    }

    // ---- this wave's share of an activation chunk: pieces u = wave, wave + 8, wave + 16 of 20 (5 pieces of 64 pixels x 4 (plane, half))
    unsigned xoffs[3], xdst[3];
    bool xin[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int u = wave + 8 * k;
        u = u < 20 ? u : 19;
        const int j = u % 5, combo = u / 5;
        const int e = j * 64 + lane;
        const int py = e / HC_PW, px = e - py * HC_PW;
        const int gy = m0 - 1 + py, gx = p0x - 1 + px;
        xin[k] = e < HC_PATCH && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        xoffs[k] = (unsigned)((combo >> 1) * p.plane_in) + (unsigned)((b * cb8 + (combo & 1)) * hw + gy * p.w + gx);   // uint4 index in chunk 0; a chunk further is 2 hw on
        xdst[k] = (unsigned)((combo * HC_PATCH + j * 64) * 16);
    }
    auto issue_x = [&](int c) __attribute__((always_inline)) {
        const unsigned st = (unsigned)(c & 1) * (unsigned)(HC_XB4 * 16);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e = ((wave + 8 * k < 20 ? wave + 8 * k : 19) % 5) * 64 + lane;
            if (e < HC_PATCH) dma16_asm(p.x, xin[k] ? (xoffs[k] + (unsigned)(2 * c * hw)) * 16u : zero_off, st + xdst[k]);
        }
    };
    auto issue_w = [&](int u) __attribute__((always_inline)) {              // unit u = 2 chunk + row parity -> weight slot u & 1
        const unsigned st = (unsigned)((2 * HC_XB4 + (u & 1) * HC_W4) * 16);
        const int c = u >> 1, par = u & 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int piece = wave + 8 * k;
            piece = piece < 18 ? piece : 17;
            const int hl = piece / 9;                                        // 18 x 32 uint4 = 9 pieces per slab
            const int rem = piece * 64 - hl * 576 + lane;                    // [tap][half][32] index
            dma16_asm(hl ? p.wlo : p.whi, (unsigned)(((((par * nchunk + c) * 18 + (rem >> 5)) * p.cout) + co0 + (rem & 31)) * 16), st + (unsigned)(piece * 1024));
        }
    };
    {   // epilogue tables of the 32 channels (every wave, same bytes)
        const unsigned co4 = (unsigned)((co0 + l5) * 4);
        dma4_asm(p.d, (unsigned)(b * p.cout * 4) + co4, (unsigned)(HC_BODY + HC_EP_D * 4));
        dma4_asm(p.s_next, (unsigned)(b * p.cout * 4) + co4, (unsigned)(HC_BODY + HC_EP_S * 4));
        dma4_asm(p.act_bias ? p.act_bias : p.zeros, p.act_bias ? co4 : 0u, (unsigned)(HC_BODY + HC_EP_B * 4));
    }
    // request order: X(0) W(0) W(1) X(1) | after unit u: W(u + 2), and behind an odd unit X(chunk + 2).  In front of unit u the requests younger than
    // the ones it needs are then exactly one weight unit + one activation chunk (6 per wave) — 3 in front of the last but one, none in front of the last.
    issue_x(0);
    issue_w(0);
    issue_w(1);
    if (nchunk > 1) issue_x(1);

    const int xoff = pty * HC_PW + ptx;              // patch element (pty, ptx) = image (pm - 1, pbx - 1)
    f32x16 accs[4];                                  // [2 row parity + column parity]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[a][r] = 0.f;

#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        const bool last = c + 1 >= nchunk;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            if (!last) E4S_WAIT_VM(2 * HC_G);
            else if (par == 0) E4S_WAIT_VM(HC_G);
            else E4S_WAIT_VM(0);
            E4S_LDS_BARRIER();
            if (c == 0 && par == 0) E4S_PROF_MARK(g_prof_hc, 1);
            unsigned xb_i = (unsigned)((c & 1) * HC_XB4 + khalf * HC_PATCH + xoff);
            unsigned wb_i = (unsigned)(2 * HC_XB4 + par * HC_W4 + khalf * 32 + l5);
            asm volatile("" : "+v"(xb_i), "+v"(wb_i));
            const uint4* xs = lds4 + xb_i;
            const uint4* whalf = lds4 + wb_i;
            if (par == 0) hc_unit<0>(accs, xs, whalf); else hc_unit<1>(accs, xs, whalf);
            E4S_LDS_BARRIER();                                   // everyone is done with this unit's weight slot (and, behind par 1, with the chunk's buffer)
            if (!last) {
                issue_w(2 * c + par + 2);
                if (par == 1 && c + 2 < nchunk) issue_x(c + 2);
            }
        }
    }
    E4S_PROF_MARK(g_prof_hc, 2);

    E4S_PROF_MARK(g_prof_hc, 3);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid < 1) p.out[(size_t)p.plane_out * 2] = make_uint4(0u, 0u, 0u, 0u);   // zero tail
    hc_epilogue(p, accs, lds_raw, HC_BODY, nz, lane_ok, b, co0, pm, pbx, khalf);
    E4S_PROF_MARK(g_prof_hc, 4);
    E4S_PROF_DRAIN();
    E4S_PROF_MARK(g_prof_hc, 5);
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// The persistent form (default).  Measured on the kernel above: 0.315 ms for the 64 -> 32 layer at batch 4 where its MFMAs need 0.12 — a workgroup lives
// 35 us for 3.4 us of matrix work because its ring is ONE unit deep (two workgroups of 76 KB per CU leave no room for more): every unit waits a memory round
// trip for a request issued 0.4 us earlier, and every workgroup pays the launch + first-chunk latency again.  Here ONE workgroup per CU walks over its
// tiles (modconv_chain.hip's roles): eight compute waves + TWO loader waves that issue every LDS-DMA ahead of its use, across tile boundaries — the next
// tile's first chunks land under this tile's epilogue.  Compute waves never touch vmcnt except for their own noise loads.
// Why two loaders.  Tuning build, 64 -> 32 layer at batch 4 (E4S_HC_EXP): the full kernel 0.311 ms; without its epilogue 0.208; without its MFMAs 0.214;
// with neither 0.098 — the three parts ADD.  Every persistent kernel of the chain sits at ~70 - 115 cycles per 1 KB request that enters or leaves a CU
// (this kernel: 224 KB of DMA + 114 KB of stores per tile), whatever its arithmetic: a wave's vmcnt is a 6-bit counter, so ONE loader wave keeps at most 63
// requests = 63 KB in flight per CU, and what is in flight / latency is the CU's ingest rate (Little's law: 63 KB / ~3 us under LDS contention = 21 GB/s per
// CU = 5.4 TB/s for the chip).  Each loader wave has its own counter: wave 8 owns the weight ring (four units, three in flight: 54 requests), wave 9 the
// activation ring (four chunks, three in flight: 60 requests) — 114 KB in flight per CU instead of 56.
// A tile's co tiles (cout / 32) are consecutive items of the SAME workgroup, so the second one's activation patch comes out of this XCD's L2.
constexpr int HP_NCW = 8;                                   // compute waves; wave 8 loads weights (+ the epilogue tables), wave 9 activations
constexpr int HP_NT = 64 * (HP_NCW + 2);                    // 640 threads
constexpr int HP_NX = 4, HP_NW = 4;                         // activation buffers (chunks), weight slots (units)
constexpr int HP_W0 = HP_NX * HC_XB4;                       // uint4 offset of the weight ring
constexpr int HP_BODY = (HP_NX * HC_XB4 + HP_NW * HC_W4) * 16;     // 152 064
constexpr int HP_LDS = HP_BODY + HC_EP_FLOATS * 4;          // 152 832
constexpr int HP_GX = 20, HP_GW = 18;                       // requests of one activation chunk / one weight unit
static_assert(HP_LDS <= 160 * 1024 && (HP_NX - 1) * HP_GX <= 63 && (HP_NW - 1) * HP_GW + 3 <= 63, "LDS per CU; vmcnt is a 6-bit counter");

__global__ __launch_bounds__(HP_NT) void up_hcp_kernel(const UpHcParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint4* lds4 = reinterpret_cast<const uint4*>(lds_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = p.h * p.w, wo = 2 * p.w;
    const int nchunk = p.cin >> 4, NU = 2 * nchunk;
    const int stride = gridDim.x, first = walk_offset(blockIdx.x, stride, p.walk & 1);
    const int my_tiles = (p.ntile - first + stride - 1) / stride;
    const int my_items = my_tiles * p.ncot;
    const int per_img = p.tiles_x * p.tiles_y;
    const int total = my_items * NU;                                       // units of this workgroup's walk = barriers every wave passes
    // item i of this workgroup: tile first + (i / ncot) * stride, co tile i % ncot
    auto item_coords = [&](int i, int& b, int& cot, int& tyt, int& txt) __attribute__((always_inline)) {
        i = i < my_items ? i : my_items - 1;                               // (ghost items past the end repeat the last one)
        int t = first + (i / p.ncot) * stride;
        t = p.rev ? p.ntile - 1 - t : t;
        cot = i % p.ncot;
        b = t / per_img;
        walk_tile_xy(t - b * per_img, p.tiles_x, p.tiles_y, p.walk >> 8, tyt, txt);
    };

    if (wave == HP_NCW) {
        // =================================================================================== weight loader: unit k lives in slot k % 4; after barrier B_k
        // every compute wave has finished unit k - 1, whose slot takes unit k + 3
        lds_byte* const lds_b = (lds_byte*)lds_raw;
        auto issue_w = [&](int k) __attribute__((always_inline)) {           // unit k of the walk: item k / NU, chunk (k % NU) / 2, row parity k & 1
            const int u = k % NU, c = u >> 1, par = u & 1;
            int b, cot, tyt, txt;
            item_coords(k / NU, b, cot, tyt, txt);
            const unsigned wst = (unsigned)((HP_W0 + (k % HP_NW) * HC_W4) * 16);
#pragma unroll
            for (int piece = 0; piece < HP_GW; ++piece) {
                const int hl = piece / 9;                                     // 18 x 32 uint4 = 9 pieces per slab
                const int rem = piece * 64 - hl * 576 + lane;                 // [tap][half][32] index
                dma16(hl ? p.wlo : p.whi, (unsigned)(((((par * nchunk + c) * 18 + (rem >> 5)) * p.cout) + cot * 32 + (rem & 31)) * 16), lds_b + wst + piece * 1024);
            }
        };
        auto item_setup = [&](int i) __attribute__((always_inline)) {         // d / s_next / bias of the item's 32 channels: read in its epilogue, NU - 1 >= 3 units later
            int b, cot, tyt, txt;
            item_coords(i, b, cot, tyt, txt);
            const unsigned co4 = (unsigned)((cot * 32 + (lane & 31)) * 4);
            dma4(p.d, (unsigned)(b * p.cout * 4) + co4, lds_b + HP_BODY + HC_EP_D * 4);
            dma4(p.s_next, (unsigned)(b * p.cout * 4) + co4, lds_b + HP_BODY + HC_EP_S * 4);
            dma4(p.act_bias ? p.act_bias : p.zeros, p.act_bias ? co4 : 0u, lds_b + HP_BODY + HC_EP_B * 4);
        };
        item_setup(0);
#pragma unroll
        for (int k = 0; k < HP_NW - 1; ++k) issue_w(k);
#pragma unroll 1
        for (int k = 0; k < total; ++k) {
            // unit k has landed: everything this wave requested except (at most) the two youngest units, 36 requests; an item's three table requests in
            // between only make the wait a little earlier than necessary
            E4S_WAIT_VM(2 * HP_GW);
            E4S_LDS_BARRIER();
            if (k > 0 && k % NU == 0) item_setup(k / NU);
            issue_w(k + HP_NW - 1);
        }
        E4S_WAIT_VM(0);   // ghost units must land before the workgroup's LDS is released
        return;
    }
    if (wave == HP_NCW + 1) {
        // =================================================================================== activation loader: chunk q (= unit 2q, 2q + 1) lives in buffer
        // q % 4; after barrier B_2q every compute wave has finished chunk q - 1, whose buffer takes chunk q + 3
        const int cb8 = p.cin >> 3;
        lds_byte* const lds_b = (lds_byte*)lds_raw;
        const unsigned zero_off = (unsigned)(2 * p.plane_in * 16);
        auto issue_x = [&](int q) __attribute__((always_inline)) {           // chunk q of the walk: item q / nchunk, its chunk q % nchunk
            const int c = q % nchunk;
            int b, cot, tyt, txt;
            item_coords(q / nchunk, b, cot, tyt, txt);
            const unsigned xst = (unsigned)((q % HP_NX) * (HC_XB4 * 16));
            const int m0 = tyt * HC_T, p0x = txt * HC_STEP - 1;
            const unsigned cb0 = (unsigned)((b * cb8 + 2 * c) * hw);
#pragma unroll
            for (int j = 0; j < 5; ++j) {                                     // this lane's patch pixel of piece j: the same for the 4 (plane, half)
                const int e = j * 64 + lane;
                const int py = e / HC_PW, px = e - py * HC_PW;
                const int gy = m0 - 1 + py, gx = p0x - 1 + px;
                const bool inb = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
                const unsigned pix = (unsigned)(gy * p.w + gx);
                if (e < HC_PATCH) {
#pragma unroll
                    for (int combo = 0; combo < 4; ++combo) {
                        const unsigned cbase = (unsigned)((combo >> 1) * p.plane_in) + cb0 + (unsigned)((combo & 1) * hw);
                        dma16(p.x, inb ? (cbase + pix) * 16u : zero_off, lds_b + xst + (combo * HC_PATCH + j * 64) * 16);
                    }
                }
            }
        };
#pragma unroll
        for (int q = 0; q < HP_NX - 1; ++q) issue_x(q);
#pragma unroll 1
        for (int k = 0; k < total; ++k) {
            if (k & 1) { E4S_LDS_BARRIER(); continue; }                       // (a chunk's second unit: nothing to publish, nothing to refill)
            E4S_WAIT_VM(2 * HP_GX);                                           // chunk k / 2 has landed: all but the two youngest chunks' requests
            E4S_LDS_BARRIER();
            issue_x((k >> 1) + HP_NX - 1);
        }
        E4S_WAIT_VM(0);
        return;
    }

    // ======================================================================================= compute waves
    const int l5 = lane & 31, khalf = lane >> 5;
    const int pty = 2 * wave + (l5 >> 4), ptx = l5 & 15;
    const int xoff = pty * HC_PW + ptx;
    if (blockIdx.x == 0 && tid < 1) p.out[(size_t)p.plane_out * 2] = make_uint4(0u, 0u, 0u, 0u);   // the zero element behind the output planes
    int xbuf = 0;                                        // (chunk of the walk) % HP_NX, carried along
#pragma unroll 1
    for (int ti = 0; ti < my_items; ++ti) {
        int b, cot, tyt, txt;
        item_coords(ti, b, cot, tyt, txt);
        const int pm = tyt * HC_T + pty, pbx = txt * HC_STEP - 1 + ptx;
        const bool lane_ok = ptx >= 1 && ptx <= HC_STEP && pbx < p.w && pm < p.h;
        float2 nz[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
        if (p.noise && lane_ok) {                        // requested now, used after the K loop
            const float* np = p.noise + (size_t)b * p.noise_bstride + (size_t)(2 * pm) * wo + 2 * pbx;
            nz[0] = *reinterpret_cast<const float2*>(np);
            nz[1] = *reinterpret_cast<const float2*>(np + wo);
        }
        f32x16 accs[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[a][r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int k = ti * NU + 2 * c + par;
                E4S_LDS_BARRIER();
                unsigned xb_i = (unsigned)(xbuf * HC_XB4 + khalf * HC_PATCH + xoff);
                unsigned wb_i = (unsigned)(HP_W0 + (k % HP_NW) * HC_W4 + khalf * 32 + l5);
                asm volatile("" : "+v"(xb_i), "+v"(wb_i));
                const uint4* xs = lds4 + xb_i;
                const uint4* whalf = lds4 + wb_i;
#ifdef E4S_PHASE_PROF
                if (p.exp & 2) continue;
#endif
                if (par == 0) hc_unit<0>(accs, xs, whalf); else hc_unit<1>(accs, xs, whalf);
            }
            xbuf = xbuf + 1 < HP_NX ? xbuf + 1 : 0;
        }
#ifdef E4S_PHASE_PROF
        if (p.exp & 1) { asm volatile("" :: "v"(accs[0][0]), "v"(accs[1][0]), "v"(accs[2][0]), "v"(accs[3][0])); continue; }
#endif
        hc_epilogue(p, accs, lds_raw, HP_BODY, nz, lane_ok, b, cot * 32, pm, pbx, khalf);
    }
}

// Half-composed weights: out[par][chunk][tap = (dy + 1) * 3 + kx][half][co][e] = scale * sum_ky kv'[ky + 2 dy + 1 - par] * W[co][ci][ky][kx], split into bf16 hi / lo;
// kv'[t] = kv[3 - t], kv[r] = (sum_c blur[r][c]) / sqrt(sum blur) — the vertical factor of the rank-1 blur kernel.
__global__ __launch_bounds__(256) void prep_weights_hc_kernel(uint16_t* __restrict__ whi, uint16_t* __restrict__ wlo, const float* __restrict__ weight,
                                                              const float* __restrict__ blur, int cout, int cin, float scale) {
    const int nchunk = (cin + CKS - 1) / CKS;
    const int64_t total = (int64_t)2 * nchunk * 9 * 2 * cout * 8;
    float kvf[4];
    {
        float rsum[4], S = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { rsum[r] = blur[4 * r] + blur[4 * r + 1] + blur[4 * r + 2] + blur[4 * r + 3]; S += rsum[r]; }
        const float rs = 1.f / sqrtf(S);
#pragma unroll
        for (int t = 0; t < 4; ++t) kvf[t] = rsum[3 - t] * rs;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i & 7);
        int64_t r = i >> 3;
        const int co = (int)(r % cout); r /= cout;
        const int half = (int)(r & 1); r >>= 1;
        const int tap = (int)(r % 9); r /= 9;
        const int chunk = (int)(r % nchunk);
        const int par = (int)(r / nchunk);
        const int ci = chunk * CKS + half * 8 + e;
        float v = 0.f;
        if (ci < cin) {
            const float* w = weight + ((size_t)co * cin + ci) * 9;
            const int dy = tap / 3 - 1, kx = tap % 3;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ty = ky + 2 * dy + 1 - par;
                if (ty >= 0 && ty <= 3) v += kvf[ty] * w[ky * 3 + kx];
            }
            v *= scale;
        }
        const unsigned hp = pack_bf16_rne(v, 0.f) & 0xffffu;
        const float hf = __builtin_bit_cast(float, hp << 16);
        whi[i] = (uint16_t)hp;
        wlo[i] = (uint16_t)(pack_bf16_rne(v - hf, 0.f) & 0xffffu);
    }
}

__device__ uint4 g_hc_zero[4];   // 64 zero bytes (UpHcParams::zeros)

}  // namespace

#ifdef E4S_PHASE_PROF
extern "C" E4S_API int e4s_prof_read_hc(long long* host, int64_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof_hc), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
extern "C" E4S_API int e4s_prof_clear_hc() {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_prof_hc));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(ptr, 0, sizeof(long long) * (size_t)E4S_PROF_BLOCKS * E4S_PROF_SLOTS);
}
#endif

extern "C" int e4s_modconv_prep_weights_hc(uint16_t* whi, uint16_t* wlo, const float* weight, const float* blur, int cout, int cin, void* stream) {
    E4S_REQUIRE(whi && wlo && weight && blur, "modconv_prep_weights_hc: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= 1, "modconv_prep_weights_hc: bad channel counts");
    const float scale = 1.0f / sqrtf((float)cin * 9.f);
    const int64_t total = (int64_t)2 * cdiv(cin, CKS) * 9 * 2 * cout * 8;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(prep_weights_hc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, whi, wlo, weight, blur, cout, cin, scale);
    return check_launch("modconv_prep_weights_hc");
}

extern "C" int e4s_modconv_up_hc(uint16_t* out_sp, const uint16_t* x_sp, const uint16_t* whi, const uint16_t* wlo, const float* d, const float* blur,
                                 const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act,
                                 int bs, int cin, int cout, int h, int w, const float* s_next, void* stream) {
    E4S_REQUIRE(out_sp && x_sp && whi && wlo && d && blur && s_next, "modconv_up_hc: null tensor (d and s_next are required)");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && h >= 1 && w >= 1 && cin >= 16 && cin % 16 == 0 && cout >= 32 && cout % 32 == 0, "modconv_up_hc: bad size (cin %% 16, cout %% 32)");
    E4S_REQUIRE((int64_t)bs * cout * 4 * h * w < ((int64_t)1 << 31) && (int64_t)bs * cin * h * w < ((int64_t)1 << 31), "modconv_up_hc: a tensor must stay below 2^31 elements");
    E4S_REQUIRE((((uintptr_t)out_sp | (uintptr_t)x_sp | (uintptr_t)whi | (uintptr_t)wlo) & 15) == 0, "modconv_up_hc: tensors must be 16-byte aligned");
    E4S_REQUIRE(!noise || (noise_weight && (noise_bs == 1 || noise_bs == bs) && ((uintptr_t)noise & 7) == 0), "modconv_up_hc: noise needs its weight, batch 1 or bs and 8-byte alignment");
    if (bs == 0) return 0;
    static const float* zeros = [] {
        void* ptr = nullptr;
        return hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_hc_zero)) == hipSuccess ? static_cast<const float*>(ptr) : nullptr;
    }();
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&up_hc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HC_LDS);
    static const hipError_t attr_p = hipFuncSetAttribute(reinterpret_cast<const void*>(&up_hcp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HP_LDS);
    if (!zeros || attr != hipSuccess || attr_p != hipSuccess) return fail(E4S_ERR_ARG, "modconv_up_hc: cannot set up the kernel (zero block / dynamic LDS limit)");
    UpHcParams p;
    p.out = reinterpret_cast<uint4*>(out_sp); p.x = reinterpret_cast<const uint4*>(x_sp);
    p.whi = reinterpret_cast<const uint4*>(whi); p.wlo = reinterpret_cast<const uint4*>(wlo);
    p.d = d; p.blur = blur; p.noise = noise; p.noise_weight = noise_weight; p.act_bias = act_bias; p.s_next = s_next; p.zeros = zeros;
    p.noise_bstride = (noise && noise_bs > 1) ? 4 * h * w : 0;
    p.act = act & 1; p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w;
    p.tiles_x = cdiv(2 * w, 2 * HC_STEP); p.tiles_y = cdiv(h, HC_T);
    p.ntile = p.tiles_x * p.tiles_y * bs; p.ncot = cout / 32;
    p.plane_in = (int64_t)bs * (cin / 8) * h * w;
    p.plane_out = (int64_t)bs * (cout / 8) * 4 * h * w;
    // Tile order.  The chain's hand-overs at 512^2 / 1024^2 (268 / 537 MB per batch of 4) are larger than the 256 MB Infinity Cache: a consumer that walks the tensor in its
    // producer's order reads what was written longest ago.  The up layers walk their tiles LAST TO FIRST, the same-resolution layers first to last: every layer of the
    // chain starts with the bytes its producer wrote last (E4S_HC_REV=0: the old order; tools/time_chain.py).
    static const int rev = [] { const char* e = getenv("E4S_HC_REV"); return e ? atoi(e) : 1; }();
    static const int walk = [] { const char* e = getenv("E4S_WALK"); return e ? atoi(e) : (1 | (8 << 8)); }();     // (E4S_WALK=0: every eighth tile, row-major)
    p.rev = rev; p.walk = walk;
#ifdef E4S_PHASE_PROF
    { const char* e = getenv("E4S_HC_EXP"); p.exp = e ? atoi(e) : 0; }      // (tuning build only)
#else
    p.exp = 0;
#endif
    // (cin = 16 has too few units per item for the persistent loader's table hand-over: the two-workgroups-per-CU form serves it; at cin >= 32 the two forms
    //  tie — 0.315 against 0.320 ms on the 64 -> 32 layer, 0.265 against 0.255 on 128 -> 64 — and the persistent one is the one the f16 + fp6 arithmetic needs next)
    if (cin >= 32) {
        static const int ncu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return n > 0 ? n : 256;
        }();
        hipLaunchKernelGGL(up_hcp_kernel, dim3(p.ntile < ncu ? p.ntile : ncu), dim3(HP_NT), HP_LDS, (hipStream_t)stream, p);
        return check_launch("modconv_up_hc");
    }
    hipLaunchKernelGGL(up_hc_kernel, dim3(p.tiles_x * p.tiles_y, cout / 32, bs), dim3(HC_NT), HC_LDS, (hipStream_t)stream, p);
    return check_launch("modconv_up_hc");
}
