// a3/a4, round 3: the masked 3x3 modulated conv (128 co x 256 px workgroup tile, the seven 32^2 ... 256^2 launches of a synthesis step) with
//   (1) a DMA-fed pipeline — the weights of a K chunk live in a ring of three ROW slots (the three taps of one kernel row, 24-25 KB) that
//       global_load_lds refills as soon as every wave is done with a row, the activation patch is double-buffered, so the K loop has no store phase:
//       three barriers per chunk with nothing but waits between them (modconv_sb.hip's kernel: stage-in-registers, two barriers around a 95 KB
//       LDS store phase during which the matrix pipe idles: 42-49 % busy in its loop against 94 % for the same loop on registers alone);
//   (2) a choice of split arithmetic (template ARITH):
//       0  bf16 x 3 — a_hi w_hi + a_hi w_lo + a_lo w_hi, bit-identical to modconv_sb.hip (same products, same accumulation order);
//       1  f16 + 2 x MX fp6 — a1 w1 on v_mfma_f32_32x32x16_f16 (a1 = f16(a), w1 = f16(w)), and the two cross terms fp6(a) * fp6(w - w1) and
//          fp6(a - a1) * fp6(w1) on v_mfma_scale_f32_32x32x64_f8f6f4 (e2m3, one power-of-two scale per lane = per 24 values: the three taps of a
//          kernel row x 8 channels; K slots 24..31 meet zero weights).  The residuals are ~2^-12 of their operands, so 4 significant bits of
//          them leave ~2^-16 per product: measured end to end on the 1024^2 generator 1.9e-4 max-abs against the fp32 oracle (bf16 x 3: 8e-5;
//          bar 1e-3; tests/experiments/emulate_split_variants.py — the two-term f16 forms (a1 + a2) w1 / a1 (w1 + w2) miss the bar at 2-5e-3).
//          Matrix-pipe time per 16-deep K step of one kernel row, measured (tools/probes/mx_probe.hip, random operands, power-limited clocks):
//          3 x 4 f16 (20.2 ns each) + 2 x 4 fp6 K=64 (23.8 ns each) = 433 ns against 36 bf16 (18.4 ns each) = 663 ns.
//          f16 range: |x * s| must stay below 65520; a wave that sees a larger value raises flags[0] bit 0 and bumps the counter flags[1].  The kernel itself does NOT
//          fall back: ops.MxGuard (e4s2024_amd/ops.py) snapshots the counter around a forward pass and the entry points (Generator.forward, FSEncoder_PSP.forward,
//          pipeline.swap_batch, runner) re-run a pass that moved it with ARITH 0.
// Measured and not kept: s_setprio(1) around the MFMA bursts (+3 % time), the residual through fma (hipcc keeps cvt + sub), the DMA through the
// builtin (vmcnt(0) in front of every row's first LDS read: no gain over the register-staged kernel).
// Operand layouts, prepared by e4s_modconv_prep_weights_mx, one ROW SLOT = what a workgroup DMAs for (parity, chunk, co tile, kernel row):
//   ARITH 0: [hi | lo][tap 3][k half 2][co 128] x 16 B (8 bf16 = channels 8 half .. 8 half + 7 of the chunk)                          24 576 B
//   ARITH 1: w1 f16 [tap 3][half 2][co 128] x 16 B | fp6 codes, first 16 B [term 2][half 2][co 128] | last 8 B [term 2][half 2][co 128] |
//            E8M0 scales [half 2][co 128] x 4 B (byte 0: term 0 = fp6(w1), byte 1: term 1 = fp6(w - w1))                              25 600 B
//            a lane's 32 fp6 positions: tap t (of the row) channel e -> position 8 t + e, positions 24..31 zero.
#include "modconv_mx_tile.h"

namespace {

// ============================================================================ weight preparation
// One thread per (par, chunk, co tile, row, half, co): the lane's 24 values of that kernel row.
__global__ __launch_bounds__(256) void prep_weights_mx_kernel(unsigned char* __restrict__ dst, const float* __restrict__ weight, const float* __restrict__ blur,
                                                              int cout, int cin, int up, int arith, float scale) {
    const int npar = up ? 4 : 1;
    const int nchunk = (cin + CKS - 1) / CKS;
    const int ntile = (cout + MX_TN - 1) / MX_TN;
    const int64_t total = (int64_t)npar * nchunk * ntile * 3 * 2 * MX_TN;
    const int rowb = mx_rowb(arith);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int n = (int)(r % MX_TN); r /= MX_TN;
        const int half = (int)(r & 1); r >>= 1;
        const int row = (int)(r % 3); r /= 3;
        const int tile = (int)(r % ntile); r /= ntile;
        const int chunk = (int)(r % nchunk);
        const int par = (int)(r / nchunk);
        const int co = tile * MX_TN + n;
        unsigned char* slot = dst + ((((size_t)par * nchunk + chunk) * ntile + tile) * 3 + row) * rowb;
        float v[24];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ci = chunk * CKS + half * 8 + e;
                float x = 0.f;
                if (ci < cin && co < cout) { x = sb_weff(weight, blur, cin, co, ci, row * 3 + t, par, up); x *= scale; }
                v[t * 8 + e] = x;
            }
        if (arith == 0) {
            uint4* hi = reinterpret_cast<uint4*>(slot);
            uint4* lo = reinterpret_cast<uint4*>(slot + 3 * 2 * MX_TN * 16);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {          // exactly prep_weights_sb_kernel's split: per element, RNE both times
                    const unsigned h0 = pack_bf16_rne(v[t * 8 + 2 * j], 0.f) & 0xffffu, h1 = pack_bf16_rne(v[t * 8 + 2 * j + 1], 0.f) & 0xffffu;
                    const unsigned l0 = pack_bf16_rne(v[t * 8 + 2 * j] - __builtin_bit_cast(float, h0 << 16), 0.f) & 0xffffu;
                    const unsigned l1 = pack_bf16_rne(v[t * 8 + 2 * j + 1] - __builtin_bit_cast(float, h1 << 16), 0.f) & 0xffffu;
                    h[j] = h0 | (h1 << 16);
                    l[j] = l0 | (l1 << 16);
                }
                hi[(t * 2 + half) * MX_TN + n] = make_uint4(h[0], h[1], h[2], h[3]);
                lo[(t * 2 + half) * MX_TN + n] = make_uint4(l[0], l[1], l[2], l[3]);
            }
        } else {
            uint4* w1p = reinterpret_cast<uint4*>(slot);
            uint4* f6lo = reinterpret_cast<uint4*>(slot + MX_W1B);
            uint2* f6hi = reinterpret_cast<uint2*>(slot + MX_W1B + MX_F6LO);
            unsigned* scp = reinterpret_cast<unsigned*>(slot + MX_W1B + MX_F6LO + MX_F6HI);
            // w1 = f16(w); the residual w - w1 is at most 2^-12 |w|: it goes through f16 scaled by 2^12 (so that it stays a normal f16 whatever
            // the layer's weight scale) and the factor comes back out through its block scale
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
            for (int t = 0; t < 3; ++t) w1p[(t * 2 + half) * MX_TN + n] = make_uint4(q1[4 * t], q1[4 * t + 1], q1[4 * t + 2], q1[4 * t + 3]);
            // block scale 2^(E - 2) with 2^E <= max < 2^(E + 1): the largest value lands in [4, 8) of e2m3's [0, 7.5] (a maximum above 7.5 * 2^(E - 2)
            // saturates: 6 % on that one element of a term that is 2^-12 of the product)
            auto expo = [](float m) { const unsigned ex = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu; return ex > 3u ? ex - 2u : 1u; };
            const unsigned e1 = expo(m1), e2 = expo(m2);
            const u32x6 c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
            const u32x6 c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
            f6lo[(0 * 2 + half) * MX_TN + n] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
            f6hi[(0 * 2 + half) * MX_TN + n] = make_uint2(c1[4], c1[5]);
            f6lo[(1 * 2 + half) * MX_TN + n] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
            f6hi[(1 * 2 + half) * MX_TN + n] = make_uint2(c2[4], c2[5]);
            const unsigned e2s = e2 > 12u ? e2 - 12u : 0u;       // the 2^12 of the residual's f16 detour
            scp[half * MX_TN + n] = e1 | (e2s << 8);
        }
    }
}

// ============================================================================ the conv kernel (its tile: modconv_mx_tile.h)
template <int ARITH, bool RGB, bool OSP, bool ENC = false>
__global__ __launch_bounds__(512, 2) void region_modconv_mx_kernel(const SbParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int ntile = p.tiles_x * p.tiles_y;
    const int npar = p.up ? 4 : 1;
    // XCD affinity (p.xcd_remap, set by the launcher when the grid allows it): workgroup `lin` runs on XCD lin % 8 (observed dispatch order; a different
    // placement would only cost speed), and each XCD has its own 4 MB L2.  A co tile's weights for all chunks are 1.2-2.5 MB and every workgroup of that co
    // tile streams them once — so the XCDs are divided among the co tiles (8 / ncot XCDs each) and an XCD's L2 holds ONE co tile's weights instead of all of
    // them (9.8 MB for a 512 -> 512 layer: each launch re-fetched its weights from beyond the L2 at the LDS-DMA path's limit, tools: -DMX_ABL).
    unsigned bx_g = blockIdx.x, cot_g = blockIdx.y, b_g = blockIdx.z;
    if (p.xcd_remap) {
        const unsigned nx = gridDim.x, ncg = gridDim.y;
        const unsigned lin = blockIdx.x + nx * (blockIdx.y + ncg * blockIdx.z);
        const unsigned per = 8u / ncg;                       // XCDs per co tile (ncg in {1, 2, 4, 8})
        const unsigned xcd = lin & 7u, q = lin >> 3;
        cot_g = xcd / per;
        const unsigned r = q * per + (xcd % per);            // 0 .. nx * nb - 1
        bx_g = r % nx;
        b_g = r / nx;
    }
    unsigned bxp = p.perm_mul ? (unsigned)(((unsigned long long)bx_g * p.perm_mul) % gridDim.x) : bx_g;
    if (p.perm_mul && (gridDim.x & 7u) == 0) bxp = (bxp & ~7u) | ((bxp + (bxp >> 3)) & 7u);       // (see modconv_sb.hip)
    const int ks = bxp / (ntile * npar);
    const int bx = bxp - ks * ntile * npar;
    const int tile = bx % ntile;
    const int par = bx / ntile;
    mx_tile_body<ARITH, RGB, OSP, ENC>(p, lds_raw, tile, par, ks, (int)cot_g, (int)b_g);
}

template <int ARITH, bool RGB, bool OSP, bool ENC = false>
int launch_mx_variant(const SbParams& p, dim3 grid, hipStream_t st) {
    constexpr int lds = MxLds<ARITH>::BYTES + (ENC ? MX_NORM_BYTES : 0);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&region_modconv_mx_kernel<ARITH, RGB, OSP, ENC>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return fail((int)attr, "region_modconv3x3_mx: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL((region_modconv_mx_kernel<ARITH, RGB, OSP, ENC>), grid, dim3(512), lds, st, p);
    return check_launch(ENC ? "conv3x3_mx" : "region_modconv3x3_mx");
}

}  // namespace

extern "C" int e4s_modconv_mx_weight_bytes(int cout, int cin, int up, int arith, int64_t* bytes) {
    E4S_REQUIRE(bytes && cout >= 1 && cin >= 1 && (arith == 0 || arith == 1), "modconv_mx_weight_bytes: bad arguments");
    *bytes = (int64_t)(up ? 4 : 1) * cdiv(cin, CKS) * cdiv(cout, MX_TN) * 3 * mx_rowb(arith);
    return 0;
}

static int prep_weights_mx(void* dst, const float* weight, const float* blur, int cout, int cin, int up, int arith, float scale, void* stream);

extern "C" int e4s_modconv_prep_weights_mx(void* dst, const float* weight, const float* blur, int cout, int cin, int up, int arith, void* stream) {
    return prep_weights_mx(dst, weight, blur, cout, cin, up, arith, 1.0f / sqrtf((float)cin * 9.f), stream);      // the equalised-lr scale of ModulatedConv2d
}

// A plain convolution weight [cout, cin, 3, 3] (no scale) in the same row-slot layout, for e4s_conv3x3_mx.
extern "C" int e4s_conv_prep_weights_mx(void* dst, const float* weight, int cout, int cin, int arith, void* stream) {
    return prep_weights_mx(dst, weight, nullptr, cout, cin, 0, arith, 1.0f, stream);
}

static int prep_weights_mx(void* dst, const float* weight, const float* blur, int cout, int cin, int up, int arith, float scale, void* stream) {
    E4S_REQUIRE(dst && weight, "modconv_prep_weights_mx: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= 1 && (arith == 0 || arith == 1), "modconv_prep_weights_mx: bad arguments");
    E4S_REQUIRE(!up || blur, "modconv_prep_weights_mx: up-conv needs the 4x4 blur kernel");
    E4S_REQUIRE(((uintptr_t)dst & 15) == 0, "modconv_prep_weights_mx: the destination must be 16-byte aligned");
    const int64_t total = (int64_t)(up ? 4 : 1) * cdiv(cin, CKS) * cdiv(cout, MX_TN) * 3 * 2 * MX_TN;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(prep_weights_mx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst, weight, blur, cout, cin, up, arith, scale);
    return check_launch("modconv_prep_weights_mx");
}

// out = PReLU(conv3x3(norm(x), W)), stride 1, pad 1: the regional-style encoder's stride-1 3x3 convolutions (models/encoders/helpers.py:128-139) on the
// DMA-fed kernel.  in_mean / in_rstd [bs][cin] (optional, together): instance norm of the input on load; prelu_slope [cout] optional.
extern "C" int e4s_conv3x3_mx(float* out, const float* x, const void* wmx, int arith, int* flags, const float* in_mean, const float* in_rstd,
                              const float* prelu_slope, int bs, int cin, int cout, int h, int w, void* stream) {
    E4S_REQUIRE(out && x && wmx, "conv3x3_mx: null tensor");
    E4S_REQUIRE((in_mean == nullptr) == (in_rstd == nullptr), "conv3x3_mx: in_mean and in_rstd go together");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= 16 && cin % CKS == 0 && cin <= MX_NORM_MAX_CIN && cout >= 1 && h >= 1 && w >= 1 && (arith == 0 || arith == 1), "conv3x3_mx: bad size (cin %% 16 == 0, cin <= 1024)");
    E4S_REQUIRE(((uintptr_t)wmx & 15) == 0, "conv3x3_mx: the weights must be 16-byte aligned");
    if (bs == 0) return 0;
    SbParams p;
    memset(&p, 0, sizeof(p));
    p.out = out; p.x = x; p.wmx = reinterpret_cast<const unsigned char*>(wmx); p.flags = flags;
    p.in_mean = in_mean; p.in_rstd = in_rstd; p.slope = prelu_slope;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.nreg = 1;
    return launch_modconv_mx(p, arith, (hipStream_t)stream, nullptr, 0, true);
}

int e4s::launch_modconv_mx(SbParams& p, int arith, hipStream_t st, float* workspace, int64_t workspace_floats, bool plain_conv) {
    p.tiles_x = cdiv(p.w, C::TW);
    p.tiles_y = cdiv(p.h, C::TH);
    const int npar = p.up ? 4 : 1;
    const int64_t base = (int64_t)p.tiles_x * p.tiles_y * npar * cdiv(p.cout, MX_TN) * p.bs;
    const int ho = p.up ? 2 * p.h : p.h, wo = p.up ? 2 * p.w : p.w;
    const int64_t out_floats = (int64_t)p.bs * p.cout * ho * wo;
    const int nchunk = cdiv(p.cin, CKS);
    int ksplit = 1;
    if (workspace && base < 384)         // one workgroup per CU: split K until one round of the chip is full (as modconv_sb.hip's 96 KB tile)
        while (ksplit < 16 && base * ksplit * 2 <= 256 && ksplit * 2 <= nchunk && (int64_t)(ksplit * 2) * out_floats <= workspace_floats) ksplit *= 2;
    if (p.rgb_out) {
        if (p.cout > MX_TN || p.up) return fail(E4S_ERR_ARG, "region_modconv3x3_mx: fused ToRGB needs all %d output channels in one workgroup tile", p.cout);
        ksplit = 1;
    }
    if (p.uni_blocks) ksplit = 1;
    p.ksplit = ksplit;
    p.chunks_per = cdiv(nchunk, ksplit);
    p.partial = workspace;
    dim3 grid(p.tiles_x * p.tiles_y * npar * ksplit, cdiv(p.cout, MX_TN), p.bs);
    p.perm_mul = p.uni_blocks ? coprime_stride(grid.x) : 0u;
    constexpr int xcd_on = 1;
    auto remap_ok = [&](dim3 g) { const unsigned long long t = (unsigned long long)g.x * g.y * g.z; return xcd_on && (g.y == 1 || g.y == 2 || g.y == 4 || g.y == 8) && t % 8 == 0 && g.y > 1; };
    p.xcd_remap = remap_ok(grid) ? 1 : 0;
    if (plain_conv) {
        p.ksplit = 1;
        p.chunks_per = nchunk;
        grid = dim3(p.tiles_x * p.tiles_y, cdiv(p.cout, MX_TN), p.bs);
        p.xcd_remap = remap_ok(grid) ? 1 : 0;
        return arith == 0 ? launch_mx_variant<0, false, false, true>(p, grid, st) : launch_mx_variant<1, false, false, true>(p, grid, st);
    }
    const bool rgb = p.rgb_out != nullptr, osp = p.s_next != nullptr;
    if (osp && !rgb) return fail(E4S_ERR_ARG, "region_modconv3x3_mx: split-plane output is built for the masked fused-ToRGB layer");
    int rc;
    if (arith == 0) rc = osp ? launch_mx_variant<0, true, true>(p, grid, st) : rgb ? launch_mx_variant<0, true, false>(p, grid, st) : launch_mx_variant<0, false, false>(p, grid, st);
    else            rc = osp ? launch_mx_variant<1, true, true>(p, grid, st) : rgb ? launch_mx_variant<1, true, false>(p, grid, st) : launch_mx_variant<1, false, false>(p, grid, st);
    return rc;
}
