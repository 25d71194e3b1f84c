// a3/a4, fast path: region-aware 3x3 modulated conv on BF16 MFMA with SPLIT operands ("split-bf16").
//
// fp32 MFMA on gfx950 runs at 1/16 of the bf16 MFMA rate.  Every fp32 operand is therefore split into two bf16 numbers,
//     a = a_hi + a_lo,   a_hi = bf16_rne(a),  a_lo = bf16_rne(a - a_hi)          (|a - a_hi - a_lo| <= 2^-17 |a|)
// and a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation: three v_mfma_f32_32x32x16_bf16 per
// 16-deep K step instead of eight v_mfma_f32_32x32x2_f32 — 5.3x less matrix-pipe time.  The dropped a_lo*b_lo term and the
// representation error are ~2^-17 relative per product; measured end-to-end on the 1024x1024 generator (17 stacked layers,
// tests/experiments/emulate_split_bf16.py): max-abs pixel error 8.2e-5 against the fp32 path, 12x inside the 1e-3 parity bar.
//
//   weights  : split once at preparation time into two K-major bf16 slabs  [par][Cin/16][tap][half][Cout][8]
//   activation operand  B[k][pix] = x[ci][pix+tap] * s[region(pix)][ci] : multiplied in fp32 and split on the fly between
//              the LDS read and the MFMA (4 VALU ops per element, hidden behind the 3 MFMAs it feeds)
//   everything else (tiling, region lookup, split-K for small maps, epilogue) as in modconv.hip.
//
// Software pipeline: the global loads of chunk t+1 (x patch + both weight slabs) are issued into registers before chunk t
// is computed and written to LDS after it, so HBM/L2 latency hides behind a whole chunk of MFMAs.
#include <stdlib.h>

#include "common.h"
#include "sb_common.h"
#include "modconv_sb.h"

using namespace e4s;

// ============================================================================ weight preparation (split + re-layout)
// whi/wlo[(((par*nchunk + chunk)*9 + tap)*2 + half)*cout + co][e]  <-  Weff[par][co][ci = chunk*16 + half*8 + e][tap] * scale
__global__ __launch_bounds__(256) void prep_weights_sb_kernel(uint16_t* __restrict__ whi, uint16_t* __restrict__ wlo,
                                                              const float* __restrict__ weight, const float* __restrict__ blur, int cout,
                                                              int cin, int up, float scale) {
    const int npar = up ? 4 : 1;
    const int nchunk = (cin + CKS - 1) / CKS;
    const int64_t total = (int64_t)npar * nchunk * 9 * 2 * cout * 8;
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
            if (!up) {
                v = w[tap];
            } else {  // transposed conv (stride 2) composed with the 4x4 blur: see modconv.hip / DESIGN.md §2
                const int a = par >> 1, b = par & 1;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                for (int ky = 0; ky < 3; ++ky) {
                    const int ty = ky + 2 * dy + 1 - a;
                    if (ty < 0 || ty > 3) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int tx = kx + 2 * dx + 1 - b;
                        if (tx < 0 || tx > 3) continue;
                        v += blur[(3 - ty) * 4 + (3 - tx)] * w[ky * 3 + kx];
                    }
                }
            }
            v *= scale;
        }
        const unsigned hp = pack_bf16_rne(v, 0.f) & 0xffffu;
        const float hf = __builtin_bit_cast(float, hp << 16);
        whi[i] = (uint16_t)hp;
        wlo[i] = (uint16_t)(pack_bf16_rne(v - hf, 0.f) & 0xffffu);
    }
}

__global__ __launch_bounds__(256) void wsq_sb_kernel(float* __restrict__ wsq, const float* __restrict__ weight, int cout, int cin, float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cin * cout) return;
    const int co = i % cout, ci = i / cout;
    const float* w = weight + ((size_t)co * cin + ci) * 9;
    float a = 0.f;
    for (int t = 0; t < 9; ++t) {
        const float v = w[t] * scale;
        a += v * v;
    }
    wsq[i] = a;
}

extern "C" int e4s_modconv_prep_weights_sb(uint16_t* whi, uint16_t* wlo, float* wsq, const float* weight, const float* blur, int cout, int cin,
                                           int up, void* stream) {
    E4S_REQUIRE(whi && wlo && weight, "modconv_prep_weights_sb: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= 1, "modconv_prep_weights_sb: bad channel counts");
    E4S_REQUIRE(!up || blur, "modconv_prep_weights_sb: up-conv needs the 4x4 blur kernel");
    const float scale = 1.0f / sqrtf((float)cin * 9.f);
    const int64_t total = (int64_t)(up ? 4 : 1) * cdiv(cin, CKS) * 9 * 2 * cout * 8;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prep_weights_sb_kernel, dim3(grid), dim3(256), 0, st, whi, wlo, weight, blur, cout, cin, up, scale);
    if (wsq) hipLaunchKernelGGL(wsq_sb_kernel, dim3(cdiv(cin * cout, 256)), dim3(256), 0, st, wsq, weight, cout, cin, scale);
    return check_launch("modconv_prep_weights_sb");
}

// ============================================================================ the conv kernel
E4S_PROF_DECL(g_prof_sb)
#ifdef E4S_PHASE_PROF
extern "C" E4S_API int e4s_prof_read_sb(long long* host, int64_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof_sb), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
extern "C" E4S_API int e4s_prof_clear_sb() {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_prof_sb));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(ptr, 0, sizeof(long long) * (size_t)E4S_PROF_BLOCKS * E4S_PROF_SLOTS);
}
#endif

// UNI = every output pixel of the launch has the same region (unmasked layers): x*s is then a property of the INPUT pixel, so it is
// multiplied and split once while staging (each staged value feeds 9 taps) and kept in LDS as two bf16 planes [pixel][16 ch]; the
// MFMA B operand is a single 16-byte LDS read per plane — no VALU in the main loop.  The two 16-byte halves of a pixel are swapped
// on every other group of 8 pixels so that the ds_read_b128 of 16 consecutive pixels touches all 64 banks once.
//
// TCONV (single-region up layers only) = the stride-2 TRANSPOSED conv alone, at 1x its algorithmic MACs: the block's lanes sit on
// an (h+1) x (w+1) grid of positions (a,b); tap (ky,kx) of the 3x3 kernel contributes W[ky][kx] * x[a-(ky>>1)][b-(kx>>1)] to the
// pre-blur pixel z[2a+(ky&1)][2b+(kx&1)], so the 9 taps feed four accumulator sets (one per output parity) and the raw sums
// are written to z [bs,cout,2h+1,2w+1]; e4s_blur_epilogue then applies blur, demodulation, noise, bias and activation.
template <int CB, int PB, int WC, int WP, int LOG_TW, int MINW, bool UNI, bool TCONV, bool RGB = false, bool XN = false, bool OSP = false>
__global__ __launch_bounds__(64 * WC * WP, MINW) void region_modconv_sb_kernel(const SbParams p) {
    // OSP = the activation leaves as split planes for the single-region chain (multiplied by the consumer's modulation, split into bf16 hi / lo)
    // XN = the input activation is channels-last (compile-time: the two staging paths must not share a register allocation)
    static_assert(!TCONV || UNI, "the transposed-conv split is only built for single-region layers");
    constexpr int NACC = TCONV ? 4 : 1;
    using C = SbCfg<CB, PB, WC, WP, LOG_TW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];           // C::LDS_BYTES, set at launch
    uint4* wsm = reinterpret_cast<uint4*>(lds_raw);                                   // [2][9][2][TN] uint4
    float* xs = reinterpret_cast<float*>(lds_raw + C::W4 * 16);                       // (!UNI) [PATCH][16 ch] fp32, 16-B slots swizzled
    float4* xf4 = reinterpret_cast<float4*>(lds_raw + C::W4 * 16);                    //        slot k (channels 4k..4k+3) of pixel e at e*4 + (k ^ ((e>>2)&3))
    uint4* xh4 = reinterpret_cast<uint4*>(lds_raw + C::W4 * 16);                      // [PATCH][2] uint4 = 16 bf16   (UNI) hi plane
    uint4* xl4 = xh4 + 2 * C::PATCH;                                                  //                                   lo plane
    float* ss = xs + C::XS_FLOATS;                                                    // [MAX_REG][CKS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    const int wc = wave / WP, wp = wave % WP;
    E4S_PROF_MARK(g_prof_sb, 0);

    const int ntile = p.tiles_x * p.tiles_y;
    const int npar = p.up ? 4 : 1;
    unsigned bxp = p.perm_mul ? (unsigned)(((unsigned long long)blockIdx.x * p.perm_mul) % gridDim.x) : blockIdx.x;
    // A multiplication keeps residues: workgroups i = j (mod 8) — one XCD — would all get tile slots of one residue mod 8, i.e. the same one or two
    // tile COLUMNS of a map that is 4 or 8 tiles wide, and a skip set made of columns (face in the middle, background left and right) idles half the
    // XCDs.  Rotating the slot inside its group of eight by the group's index gives every XCD every column.
    if (p.perm_mul && (gridDim.x & 7u) == 0) bxp = (bxp & ~7u) | ((bxp + (bxp >> 3)) & 7u);
    const int ks = bxp / (ntile * npar);
    const int bx = bxp - ks * ntile * npar;
    const int tile = bx % ntile;
    const int par = bx / ntile;
    const int pa = par >> 1, pb_ = par & 1;
    const int y0 = (tile / p.tiles_x) * C::TH, x0 = (tile % p.tiles_x) * C::TW;
    const int co0 = blockIdx.y * C::TN;
    const int b = blockIdx.z;
    const int hw = p.h * p.w;
    const int ho = p.up ? 2 * p.h : p.h, wo = p.up ? 2 * p.w : p.w;
    const int nchunk = (p.cin + CKS - 1) / CKS;
    // the uniform 16 x 16 output blocks under this tile (up layer: outputs (2y + pa, 2x + pb) of an 8 x 32 input tile = one block row, four blocks)
    unsigned ub_skip = 0;       // bit j: block j of this tile is computed by the block kernel
    if constexpr (C::TH == 8 && C::TW == 32 && !TCONV && !UNI) {
        if (p.up && p.uni_blocks && p.uni_ctrl[2] != 0) {
            const int nbx = wo >> 4, nby = ho >> 4;
            const int by = (2 * y0) >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bxk = ((2 * x0) >> 4) + j;
                const bool uni = by < nby && bxk < nbx && p.uni_blocks[((size_t)b * nby + by) * nbx + bxk] != 255;   // one region, or four uniform sub-blocks
                const bool outside = by >= nby || bxk >= nbx;
                if (uni || outside) ub_skip |= 1u << j;
            }
            if (ub_skip == 0xfu) return;
        }
    }

    int goff[C::EPT];
    bool ginb[C::EPT];
#pragma unroll
    for (int j = 0; j < C::EPT; ++j) {
        const int e = tid + j * C::NT;
        const int py = e / C::PW, px = e - py * C::PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        ginb[j] = (e < C::PATCH) && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        goff[j] = gy * p.w + gx;
    }
    const float* xb = p.x + (size_t)b * p.cin * hw;
    const float* sb = p.s + (size_t)b * p.nreg * p.cin;

    int xoff[PB], cls[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int ty = C::blk_y(pbk, l5), tx = C::blk_x(pbk, l5);
        xoff[q] = ty * C::PW + tx;
        const int y = y0 + ty, x = x0 + tx;
        int c = 0;
        if (p.labels) {
            c = E4S_LABEL_NONE;
            if (y < p.h && x < p.w) {
                const int oy = p.up ? 2 * y + pa : y, ox = p.up ? 2 * x + pb_ : x;
                c = p.labels[((size_t)b * p.lh + nearest_src(oy, p.lscale_y, p.lh)) * p.lw + nearest_src(ox, p.lscale_x, p.lw)];
            }
        }
        cls[q] = (c < p.nreg) ? c : -1;
    }

    f32x16 accs[NACC][CB][PB];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int q = 0; q < PB; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[a][i][q][r] = 0.f;
    auto& acc = accs[0];

    // register stage of the NEXT chunk
    float xr[CKS][C::EPT];
    unsigned wr[C::WPT][4];   // scalar components: a uint4 array here ends up in scratch
    float sr = 0.f;

    // Unconditional loads (no per-element branches, so the compiler issues them here and they really are a prefetch): out-of-image
    // elements read a valid clamped address and are zeroed when the chunk is written to LDS; channels >= cin meet zero-padded
    // weights; output-channel rows >= cout are computed but never stored.  The lambdas must be inlined or xr/wr live in scratch.
    int goffs[C::EPT];
#pragma unroll
    for (int j = 0; j < C::EPT; ++j) goffs[j] = ginb[j] ? goff[j] : 0;
    const ptrdiff_t wdelta = p.wlo - p.whi;
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const int ci0 = chunk * CKS;
        const int cmax = p.cin - 1 - ci0;
        if constexpr (XN) {   // channel-blocked input [cin/8][h][w][8]: each patch pixel's two 8-channel blocks, four 16-byte loads per pixel
#pragma unroll
            for (int j = 0; j < C::EPT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)(ci0 / 8 + (q >> 1)) * hw + goffs[j]) * 8 + 4 * (q & 1));
                    xr[4 * q][j] = v.x; xr[4 * q + 1][j] = v.y; xr[4 * q + 2][j] = v.z; xr[4 * q + 3][j] = v.w;
                }
        } else {
#pragma unroll
            for (int c = 0; c < CKS; ++c) {
                const float* xc = xb + (size_t)(ci0 + (c < cmax ? c : cmax)) * hw;
#pragma unroll
                for (int j = 0; j < C::EPT; ++j) xr[c][j] = xc[goffs[j]];
            }
        }
        const size_t wbase = ((size_t)par * nchunk + chunk) * 18 * p.cout;  // uint4 units: [tap][half][cout]
#pragma unroll
        for (int v = 0; v < C::WPT; ++v) {
            int idx = tid + v * C::NT;
            idx = idx < C::W4 ? idx : C::W4 - 1;
            const int hl = idx / (18 * C::TN);
            const int rem = idx - hl * 18 * C::TN;
            const int th = rem / C::TN, n = rem - th * C::TN;
            const int co = (co0 + n < p.cout) ? co0 + n : p.cout - 1;
            const uint4 t4 = p.whi[(ptrdiff_t)hl * wdelta + (ptrdiff_t)(wbase + (size_t)th * p.cout + co)];   // lo slab = hi slab + wdelta
            wr[v][0] = t4.x; wr[v][1] = t4.y; wr[v][2] = t4.z; wr[v][3] = t4.w;
        }
        if (tid < E4S_MAX_REGIONS * CKS) {
            const int r = tid / CKS, c = tid % CKS;
            sr = (r < p.nreg && ci0 + c < p.cin) ? sb[(size_t)r * p.cin + ci0 + c] : 0.f;
        }
    };
    auto store_chunk = [&](int chunk) __attribute__((always_inline)) {
        if constexpr (UNI) {
            // s of the single region for the 16 channels of this chunk (wave-uniform loads)
            float sc[CKS];
#pragma unroll
            for (int c = 0; c < CKS; ++c) sc[c] = (chunk * CKS + c < p.cin) ? sb[chunk * CKS + c] : 0.f;
#pragma unroll
            for (int j = 0; j < C::EPT; ++j) {
                const int e = tid + j * C::NT;
                if (e < C::PATCH) {
                    unsigned h[8], l[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        split2(ginb[j] ? xr[2 * c][j] * sc[2 * c] : 0.f, ginb[j] ? xr[2 * c + 1][j] * sc[2 * c + 1] : 0.f, h[c], l[c]);
                    const int sw = (e >> 3) & 1;
                    xh4[e * 2 + (0 ^ sw)] = make_uint4(h[0], h[1], h[2], h[3]);
                    xh4[e * 2 + (1 ^ sw)] = make_uint4(h[4], h[5], h[6], h[7]);
                    xl4[e * 2 + (0 ^ sw)] = make_uint4(l[0], l[1], l[2], l[3]);
                    xl4[e * 2 + (1 ^ sw)] = make_uint4(l[4], l[5], l[6], l[7]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::EPT; ++j) {
                const int e = tid + j * C::NT;
                if (e < C::PATCH) {
                    const int g = (e >> 2) & 3;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        xf4[e * 4 + (k ^ g)] = ginb[j] ? make_float4(xr[4 * k][j], xr[4 * k + 1][j], xr[4 * k + 2][j], xr[4 * k + 3][j])
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < C::WPT; ++v) {
            const int idx = tid + v * C::NT;
            if (idx < C::W4) wsm[idx] = make_uint4(wr[v][0], wr[v][1], wr[v][2], wr[v][3]);
        }
        if (!UNI && tid < E4S_MAX_REGIONS * CKS) ss[tid] = sr;
    };

    const int ch_begin = ks * p.chunks_per;
    const int ch_end = (ch_begin + p.chunks_per < nchunk) ? ch_begin + p.chunks_per : nchunk;
    if (ch_begin < ch_end) load_chunk(ch_begin);
    for (int chunk = ch_begin; chunk < ch_end; ++chunk) {
        __syncthreads();
        store_chunk(chunk);
        __syncthreads();
        if (chunk == ch_begin) E4S_PROF_MARK(g_prof_sb, 1);
        if (chunk + 1 < ch_end) load_chunk(chunk + 1);

        // (!UNI) modulation of this lane's pixels for its 8 channels of the chunk: ci = 8*khalf + e
        float sv[PB][8];
        if constexpr (!UNI) {
#pragma unroll
            for (int q = 0; q < PB; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) sv[q][e] = cls[q] >= 0 ? ss[cls[q] * CKS + khalf * 8 + e] : 0.f;
        }
        const uint4* whalf = wsm + khalf * C::TN + wc * CB * 32 + l5;   // + tap*2*TN, + 18*TN for the lo slab

        {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int toff = TCONV ? (1 - (ky >> 1)) * C::PW + (1 - (kx >> 1)) : ky * C::PW + kx;
            const int ai = TCONV ? 2 * (ky & 1) + (kx & 1) : 0;
            uint4 bh[PB], bl[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                if constexpr (UNI) {
                    const int e = xoff[q] + toff;
                    const int slot = e * 2 + (khalf ^ ((e >> 3) & 1));
                    bh[q] = xh4[slot];
                    bl[q] = xl4[slot];
                } else {
                    const int e = xoff[q] + toff;
                    const int g = (e >> 2) & 3;
                    const float4 x0 = xf4[e * 4 + ((2 * khalf) ^ g)], x1 = xf4[e * 4 + ((2 * khalf + 1) ^ g)];
                    split2(x0.x * sv[q][0], x0.y * sv[q][1], bh[q].x, bl[q].x);
                    split2(x0.z * sv[q][2], x0.w * sv[q][3], bh[q].y, bl[q].y);
                    split2(x1.x * sv[q][4], x1.y * sv[q][5], bh[q].z, bl[q].z);
                    split2(x1.z * sv[q][6], x1.w * sv[q][7], bh[q].w, bl[q].w);
                }
            }
            uint4 ah[CB], al[CB];
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                ah[i] = whalf[tap * 2 * C::TN + i * 32];
                al[i] = whalf[18 * C::TN + tap * 2 * C::TN + i * 32];
            }
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int q = 0; q < PB; ++q)
                    accs[ai][i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[q]), accs[ai][i][q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int q = 0; q < PB; ++q)
                    accs[ai][i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[q]), accs[ai][i][q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int q = 0; q < PB; ++q)
                    accs[ai][i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[q]), accs[ai][i][q], 0, 0, 0);
        }
        }
    }

    E4S_PROF_MARK(g_prof_sb, 2);
    if constexpr (TCONV) {  // raw pre-blur sums: z[b][co][2a+i][2b+j], z is (2h+1) x (2w+1)
        const int zh = 2 * p.h + 1, zw = 2 * p.w + 1;
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int pbk = wp * PB + q;
            const int a = y0 + C::blk_y(pbk, l5), bb = x0 + C::blk_x(pbk, l5);
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int zy = 2 * a + (cl >> 1), zx = 2 * bb + (cl & 1);
                if (zy >= zh || zx >= zw) continue;
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                        if (co < p.cout) p.out[(((size_t)b * p.cout + co) * zh + zy) * zw + zx] = accs[cl][i][q][r];
                    }
            }
        }
        return;
    }

    if (p.ksplit > 1) {
        float* part = p.partial + ((size_t)ks * p.bs + b) * p.cout * ho * wo;
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int pbk = wp * PB + q;
            const int y = y0 + C::blk_y(pbk, l5), x = x0 + C::blk_x(pbk, l5);
            if (y >= p.h || x >= p.w) continue;
            const size_t opix = (size_t)(p.up ? 2 * y + pa : y) * wo + (p.up ? 2 * x + pb_ : x);
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    if (co < p.cout) part[(size_t)co * ho * wo + opix] = acc[i][q][r];
                }
        }
        return;
    }

    E4S_PROF_MARK(g_prof_sb, 3);
    sb_epilogue<C, CB, PB, WP, RGB, OSP>(p, lds_raw, acc, cls, co0, b, y0, x0, pa, pb_, ho, wo, ub_skip);
    E4S_PROF_MARK(g_prof_sb, 4);
    E4S_PROF_DRAIN();
    E4S_PROF_MARK(g_prof_sb, 5);
}

// Sum the K-slices in a fixed order and apply the StyledConv epilogue.
__global__ __launch_bounds__(256) void modconv_sb_finalize_kernel(const SbParams p, int ho, int wo) {
    const size_t ohw = (size_t)ho * wo;
    const size_t per_b = (size_t)p.cout * ohw;
    const size_t total = (size_t)p.bs * per_b;
    const float nw = p.noise ? p.noise_weight[0] : 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / per_b);
        const size_t r = i - (size_t)b * per_b;
        const int co = (int)(r / ohw);
        const size_t opix = r - (size_t)co * ohw;
        const int oy = (int)(opix / wo), ox = (int)(opix - (size_t)oy * wo);
        float a = 0.f;
        for (int k = 0; k < p.ksplit; ++k) a += p.partial[(size_t)k * total + i];
        int c = 0;
        if (p.labels) c = p.labels[((size_t)b * p.lh + nearest_src(oy, p.lscale_y, p.lh)) * p.lw + nearest_src(ox, p.lscale_x, p.lw)];
        float v = 0.f;
        if (c < p.nreg) v = a * (p.d ? p.d[((size_t)b * p.nreg + c) * p.cout + co] : 1.f);
        if (p.noise) v += nw * p.noise[(size_t)b * p.noise_bstride + opix];
        if (p.act_bias) v += p.act_bias[co];
        if (p.act) v = (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
        p.out[i] = v;
    }
}

template <int CB, int PB, int WC, int WP, int LOG_TW>
static int launch_sb_tconv(SbParams& p, hipStream_t st) {
    using C = SbCfg<CB, PB, WC, WP, LOG_TW>;
    p.tiles_x = cdiv(p.w + 1, C::TW);   // positions (a,b) run over (h+1) x (w+1)
    p.tiles_y = cdiv(p.h + 1, C::TH);
    p.up = 0;                            // one pass, no parity replication
    p.ksplit = 1;
    p.chunks_per = cdiv(p.cin, CKS);
    p.partial = nullptr;
    dim3 grid(p.tiles_x * p.tiles_y, cdiv(p.cout, C::TN), p.bs);
    hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, 2, true, true>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
    return check_launch("modconv_tconv_sb");
}

template <int CB, int PB, int WC, int WP, int LOG_TW>
static int launch_sb(SbParams& p, hipStream_t st, float* workspace, int64_t workspace_floats) {
    using C = SbCfg<CB, PB, WC, WP, LOG_TW>;
    p.tiles_x = cdiv(p.w, C::TW);
    p.tiles_y = cdiv(p.h, C::TH);
    const int npar = p.up ? 4 : 1;
    const int64_t base = (int64_t)p.tiles_x * p.tiles_y * npar * cdiv(p.cout, C::TN) * p.bs;
    const int ho = p.up ? 2 * p.h : p.h, wo = p.up ? 2 * p.w : p.w;
    const int64_t out_floats = (int64_t)p.bs * p.cout * ho * wo;
    const int nchunk = cdiv(p.cin, CKS);
    int ksplit = 1;
    if (workspace && base < 384) {
        // enough workgroups for one round of the chip: 4 per CU for the small tiles, 1 per CU for the 96 KB tile (more splits there only
        // add rounds of prologue / epilogue and partial sums: the 32 x 32 layer 0.127 -> 0.09 ms with 4 splits instead of 16)
        const int64_t target = C::LDS_BYTES > 64 * 1024 ? 256 : 1024;
        while (ksplit < 16 && base * ksplit * 2 <= target && ksplit * 2 <= nchunk && (int64_t)(ksplit * 2) * out_floats <= workspace_floats) ksplit *= 2;
    }
    if (p.rgb_out) {
        if (WC != 1 || p.cout > C::TN || p.up) return fail(E4S_ERR_ARG, "region_modconv3x3_sb: fused ToRGB needs all %d output channels in one workgroup tile", p.cout);
        ksplit = 1;
    }
    if (p.uni_blocks) ksplit = 1;       // (the skipped blocks are decided per workgroup / lane in the one-pass epilogue)
    p.ksplit = ksplit;
    p.chunks_per = cdiv(nchunk, ksplit);
    p.partial = workspace;
    dim3 grid(p.tiles_x * p.tiles_y * npar * ksplit, cdiv(p.cout, C::TN), p.bs);
    p.perm_mul = p.uni_blocks ? coprime_stride(grid.x) : 0u;
    constexpr int uni_lds = C::LDS_BYTES_UNI;
    if (p.rgb_out) {
        if constexpr (WC == 1) {
            constexpr int MW = C::NT / 256 >= 2 ? 2 : 2;
            if (C::LDS_BYTES > 64 * 1024) {
                static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, MW, false, false, true>),
                                                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
                if (attr2 != hipSuccess) return fail((int)attr2, "region_modconv3x3_sb: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr2));
            }
            if (p.s_next) {   // masked layer handing over to the single-region chain as split planes
                if (!p.labels || p.x_nhwc || !p.out) return fail(E4S_ERR_ARG, "region_modconv3x3_sb: split-plane output is built for the masked fused-ToRGB layer");
                if (C::LDS_BYTES > 64 * 1024) {
                    static const hipError_t attr3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, MW, false, false, true, false, true>),
                                                                        hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
                    if (attr3 != hipSuccess) return fail((int)attr3, "region_modconv3x3_sb: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr3));
                }
                hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, MW, false, false, true, false, true>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
                return check_launch("region_modconv3x3_sb");
            }
            if (!p.labels && p.nreg == 1 && p.x_nhwc)
                hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, MW, true, false, true, true>), grid, dim3(C::NT), uni_lds, st, p);
            else if (p.x_nhwc)
                return fail(E4S_ERR_ARG, "region_modconv3x3_sb: channels-last input is built for the single-region layers only");
            else if (!p.labels && p.nreg == 1)
                hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, MW, true, false, true>), grid, dim3(C::NT), uni_lds, st, p);
            else
                hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, MW, false, false, true>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
            return check_launch("region_modconv3x3_sb");
        }
    }
    constexpr bool BIG = C::LDS_BYTES > 64 * 1024;   // one workgroup per CU
    constexpr int BIGW = C::NT / 256;                // waves per SIMD that workgroup provides
    if constexpr (BIG) {
        if (p.x_nhwc) return fail(E4S_ERR_ARG, "region_modconv3x3_sb: channels-last input is built for the single-region layers only");
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, BIGW, false, false>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (attr != hipSuccess) return fail((int)attr, "region_modconv3x3_sb: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
        hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, BIGW, false, false>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
    } else if (p.x_nhwc) {
        if (p.labels || p.nreg != 1) return fail(E4S_ERR_ARG, "region_modconv3x3_sb: channels-last input is built for the single-region layers only");
        hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, 2, true, false, false, true>), grid, dim3(C::NT), uni_lds, st, p);
    } else if (!p.labels && p.nreg == 1)
        hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, 2, true, false>), grid, dim3(C::NT), uni_lds, st, p);
    else
        hipLaunchKernelGGL((region_modconv_sb_kernel<CB, PB, WC, WP, LOG_TW, 2, false, false>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
    if (ksplit > 1) {
        const int g = (int)(cdiv64(out_floats, 256) < 2048 ? cdiv64(out_floats, 256) : 2048);
        hipLaunchKernelGGL(modconv_sb_finalize_kernel, dim3(g), dim3(256), 0, st, p, ho, wo);
    }
    return check_launch("region_modconv3x3_sb");
}

static inline bool layout_in_out_nhwc(int layout) { return (layout & (E4S_X_NHWC | E4S_OUT_NHWC)) != 0; }

static int region_modconv3x3_sb_impl(float* out, const float* x, const uint16_t* whi, const uint16_t* wlo, const float* s, const float* d,
                                     const uint8_t* labels, int lh, int lw, const float* noise, int noise_bs, const float* noise_weight,
                                     const float* act_bias, int act, int bs, int cin, int cout, int h, int w, int nreg, int up,
                                     float* workspace, int64_t workspace_floats, float* rgb_out, const float* rgb_wt, const float* rgb_s,
                                     const float* rgb_bias, const float* rgb_skip, const float* rgb_up_kernel, const float* s_next,
                                     const uint8_t* uniform_blocks, const int* uniform_ctrl, const void* wmx, int arith, int* flags, void* stream) {
    const int layout = up & (E4S_X_NHWC | E4S_OUT_NHWC | E4S_OUT_SP);
    up &= 1;
    E4S_REQUIRE(!(layout & E4S_OUT_SP) || (s_next && rgb_out && out && cout % 8 == 0 && !(layout & E4S_OUT_NHWC) && ((uintptr_t)out & 15) == 0),
                "region_modconv3x3_sb: split-plane output needs s_next, the fused ToRGB, cout %% 8 == 0 and a 16-byte aligned tensor");
    E4S_REQUIRE((out || rgb_out) && x && ((whi && wlo) || wmx) && s, "region_modconv3x3_sb: null tensor");
    E4S_REQUIRE(!rgb_out || (rgb_wt && rgb_s && rgb_bias && (!rgb_skip || rgb_up_kernel) && w >= 32 && !up), "region_modconv3x3_sb: incomplete fused-ToRGB arguments");
    E4S_REQUIRE(bs >= 0 && cin >= 1 && cout >= 1 && h >= 1 && w >= 1, "region_modconv3x3_sb: bad size");
    E4S_REQUIRE(nreg >= 1 && nreg <= E4S_MAX_REGIONS, "region_modconv3x3_sb: %d regions (max %d)", nreg, E4S_MAX_REGIONS);
    E4S_REQUIRE(labels || nreg == 1, "region_modconv3x3_sb: nreg > 1 needs a label map");
    E4S_REQUIRE(!labels || (lh >= 1 && lw >= 1), "region_modconv3x3_sb: bad label map size");
    E4S_REQUIRE(!noise || (noise_weight && (noise_bs == 1 || noise_bs == bs)), "region_modconv3x3_sb: noise needs its weight and batch 1 or bs");
    E4S_REQUIRE(bs <= 65535, "region_modconv3x3_sb: batch too large");
    E4S_REQUIRE((((uintptr_t)whi | (uintptr_t)wlo) & 15) == 0, "region_modconv3x3_sb: weight slabs must be 16-byte aligned");
    if (bs == 0) return 0;
    SbParams p;
    p.out = out; p.x = x; p.whi = reinterpret_cast<const uint4*>(whi); p.wlo = reinterpret_cast<const uint4*>(wlo); p.s = s; p.d = d;
    p.labels = labels; p.noise = noise; p.noise_weight = noise_weight; p.act_bias = act_bias; p.lh = lh; p.lw = lw; p.act = act;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.nreg = nreg;
    p.x_nhwc = (layout & E4S_X_NHWC) ? 1 : 0;
    p.out_nhwc = (layout & E4S_OUT_NHWC) ? 1 : 0;
    p.up = up;
    E4S_REQUIRE(!p.x_nhwc || (cin % 16 == 0 && ((uintptr_t)x & 15) == 0), "region_modconv3x3_sb: channels-last input needs cin %% 16 == 0 and a 16-byte aligned tensor");
    E4S_REQUIRE(!p.out_nhwc || (out && cout % 8 == 0 && ((uintptr_t)out & 15) == 0 && w >= 32),
                "region_modconv3x3_sb: channel-blocked output needs cout %% 8 == 0, a 16-byte aligned tensor and w >= 32 (no split-K)");
    const int ho = up ? 2 * h : h, wo = up ? 2 * w : w;
    p.lscale_y = labels ? (float)lh / (float)ho : 1.f;
    p.lscale_x = labels ? (float)lw / (float)wo : 1.f;
    p.noise_bstride = (noise && noise_bs > 1) ? ho * wo : 0;
    p.rgb_out = rgb_out; p.rgb_wt = rgb_wt; p.rgb_s = rgb_s; p.rgb_bias = rgb_bias; p.rgb_skip = rgb_skip; p.rgb_upk = rgb_up_kernel;
    p.s_next = (layout & E4S_OUT_SP) ? s_next : nullptr;
    p.plane_out = (int64_t)bs * (cout / 8) * ho * wo;
    E4S_REQUIRE(!uniform_blocks || (up && labels && w >= 32 && cout >= 128 && (h % 8) == 0 && (w % 8) == 0 && !layout),
                "region_modconv3x3_sb: the uniform-block map goes with a masked up layer of width >= 32, cout >= 128, h and w multiples of 8");
    E4S_REQUIRE(!uniform_blocks || uniform_ctrl, "region_modconv3x3_sb: the uniform-block map comes with its control words (e4s_uniform_blocks)");
    p.uni_blocks = uniform_blocks;
    p.uni_ctrl = uniform_ctrl;
    p.perm_mul = 0;
    p.wmx = reinterpret_cast<const unsigned char*>(wmx);
    p.flags = flags;
    hipStream_t st = (hipStream_t)stream;
    float* ws = p.out_nhwc ? nullptr : workspace;     // the split-K partial sums are laid out channels-first
    const int64_t wf = p.out_nhwc ? 0 : workspace_floats;
    if (wmx) {      // the DMA-fed masked kernel (modconv_mx.hip)
        E4S_REQUIRE(labels && w >= 32 && cout >= 128 && cin % CKS == 0 && !layout_in_out_nhwc(layout) && ((uintptr_t)wmx & 15) == 0 && (arith == 0 || arith == 1),
                    "region_modconv3x3_mx: built for masked layers of width >= 32, cout >= 128, cin %% 16 == 0, channels-first activations");
        if (int rc = launch_modconv_mx(p, arith, st, ws, wf)) return rc;
        if (p.ksplit > 1) {
            const int64_t out_floats = (int64_t)bs * cout * ho * wo;
            const int g = (int)(cdiv64(out_floats, 256) < 2048 ? cdiv64(out_floats, 256) : 2048);
            hipLaunchKernelGGL(modconv_sb_finalize_kernel, dim3(g), dim3(256), 0, st, p, ho, wo);
            return check_launch("region_modconv3x3_mx");
        }
        return 0;
    }
    if (w >= 32) {
        // 64 x 4 instead of 32 x 8 pixel tiles (bit 0: single-region layers, bit 1: masked layers): a 66-pixel patch row costs three cache
        // lines per channel like a 34-pixel one.  Tile-read probe: 1.8 -> 2.5-2.9 TB/s; in the pipeline (inputs partly cache-resident)
        // the 1024x1024 / 512x512 layers gain 3.5 % / 2.5 %, the masked layers nothing measurable: on for the single-region layers.
        constexpr int tw64 = 1;
        if (w >= 64 && !labels && (tw64 & 1) && !p.x_nhwc) {   // (a channel-blocked input already reads whole lines: 32 x 8 tiles are then 1-2 % better)
            if (cout > 32) return launch_sb<2, 2, 1, 4, 6>(p, st, ws, wf);   // 64 co x (64 x 4) px
            return launch_sb<1, 2, 1, 4, 6>(p, st, ws, wf);                  // 32 co x (64 x 4) px
        }
        if (w >= 64 && labels && cout >= 128 && (tw64 & 2)) return launch_sb<4, 1, 1, 8, 6>(p, st, ws, wf);
        if (labels && cout >= 128) return launch_sb<4, 1, 1, 8, 5>(p, st, ws, wf);   // 512 threads: 128 co x 256 px; on masked layers
                                                                                              // the on-the-fly split of B feeds 12 MFMAs, not 6
        if (cout > 32) return launch_sb<2, 2, 1, 4, 5>(p, st, ws, wf);   // 64 co x 256 px
        return launch_sb<1, 2, 1, 4, 5>(p, st, ws, wf);                  // 32 co x 256 px
    }
    // masked 16 x 16 layers: 128 co x 256 px (the whole map) per workgroup — a weight byte is then read by 4 workgroups (the batch) instead of 8: the 16 -> 32 up layer
    // streams 151 MB instead of 302 MB (batch 4, in-run: 114 -> 88 us; the same-resolution layer 46 -> 39 us)
    if (w >= 16 && labels && cout >= 128) return launch_sb<4, 1, 1, 8, 4>(p, st, ws, wf);
    if (w >= 16) return launch_sb<1, 2, 2, 2, 4>(p, st, ws, wf);         // 64 co x 128 px (16 x 8)
    if (w >= 8) return launch_sb<1, 1, 2, 2, 3>(p, st, ws, wf);          // 64 co x  64 px (8 x 8)
    return launch_sb<1, 1, 2, 2, 2>(p, st, ws, wf);                      // 64 co x  64 px (4 x 16)
}

extern "C" int e4s_region_modconv3x3_sb(float* out, const float* x, const uint16_t* whi, const uint16_t* wlo, const float* s, const float* d,
                                        const uint8_t* labels, int lh, int lw, const float* noise, int noise_bs, const float* noise_weight,
                                        const float* act_bias, int act, int bs, int cin, int cout, int h, int w, int nreg, int up,
                                        float* workspace, int64_t workspace_floats, float* rgb_out, const float* rgb_wt, const float* rgb_s,
                                        const float* rgb_bias, const float* rgb_skip, const float* rgb_up_kernel, const float* s_next,
                                        const uint8_t* uniform_blocks, const int* uniform_ctrl, void* stream) {
    return region_modconv3x3_sb_impl(out, x, whi, wlo, s, d, labels, lh, lw, noise, noise_bs, noise_weight, act_bias, act, bs, cin, cout, h, w, nreg, up, workspace,
                                     workspace_floats, rgb_out, rgb_wt, rgb_s, rgb_bias, rgb_skip, rgb_up_kernel, s_next, uniform_blocks, uniform_ctrl, nullptr, 0, nullptr, stream);
}

// The same layer on the DMA-fed masked kernel of modconv_mx.hip: `wmx` from e4s_modconv_prep_weights_mx(arith), `flags` (optional, arith 1) receives bit 0
// when a modulated activation left the f16 range (the result is then not to be trusted: repeat with arith 0).
extern "C" int e4s_region_modconv3x3_mx(float* out, const float* x, const void* wmx, int arith, int* flags, const float* s, const float* d,
                                        const uint8_t* labels, int lh, int lw, const float* noise, int noise_bs, const float* noise_weight,
                                        const float* act_bias, int act, int bs, int cin, int cout, int h, int w, int nreg, int up,
                                        float* workspace, int64_t workspace_floats, float* rgb_out, const float* rgb_wt, const float* rgb_s,
                                        const float* rgb_bias, const float* rgb_skip, const float* rgb_up_kernel, const float* s_next,
                                        const uint8_t* uniform_blocks, const int* uniform_ctrl, void* stream) {
    E4S_REQUIRE(wmx, "region_modconv3x3_mx: null weights");
    return region_modconv3x3_sb_impl(out, x, nullptr, nullptr, s, d, labels, lh, lw, noise, noise_bs, noise_weight, act_bias, act, bs, cin, cout, h, w, nreg, up, workspace,
                                     workspace_floats, rgb_out, rgb_wt, rgb_s, rgb_bias, rgb_skip, rgb_up_kernel, s_next, uniform_blocks, uniform_ctrl, wmx, arith, flags, stream);
}

// ============================================================================ single-region up layer in two launches
// (1) transposed conv only -> z [bs, cout, 2h+1, 2w+1] (raw sums, x already modulated by s);  whi/wlo from
//     e4s_modconv_prep_weights_sb(..., up = 0) on the layer's 3x3 weight (NOT blur-composed).
extern "C" int e4s_modconv_tconv_sb(float* z, const float* x, const uint16_t* whi, const uint16_t* wlo, const float* s, int bs, int cin, int cout,
                                    int h, int w, void* stream) {
    E4S_REQUIRE(z && x && whi && wlo && s, "modconv_tconv_sb: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= 1 && cout >= 1 && h >= 1 && w >= 1, "modconv_tconv_sb: bad size");
    E4S_REQUIRE((((uintptr_t)whi | (uintptr_t)wlo) & 15) == 0, "modconv_tconv_sb: weight slabs must be 16-byte aligned");
    if (bs == 0) return 0;
    SbParams p;
    memset(&p, 0, sizeof(p));
    p.out = z; p.x = x; p.whi = reinterpret_cast<const uint4*>(whi); p.wlo = reinterpret_cast<const uint4*>(wlo); p.s = s;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.nreg = 1;
    hipStream_t st = (hipStream_t)stream;
    if (w + 1 > 16) {
        if (cout > 32) return launch_sb_tconv<2, 1, 1, 4, 5>(p, st);   // 64 co x 128 positions, 4 parity accumulators
        return launch_sb_tconv<1, 2, 1, 4, 5>(p, st);                  // 32 co x 256 positions
    }
    return launch_sb_tconv<1, 1, 2, 2, 4>(p, st);
}

// (2) out[b,co,p] = act( d[b,co] * sum_{ty,tx} z[b,co,py+ty-1,px+tx-1] * blur[3-ty][3-tx] + noise_w*noise[p] + bias[co] ),  zero padding:
//     upfirdn2d(z, blur 4x4, pad (1,1)) of models/stylegan2/model.py:300 fused with the StyledConv epilogue (:419-421).
// LDS-tiled: a block owns a 64 x 16 output tile of one (sample, channel) plane; the (16+3) x (64+3) pre-blur window is loaded
// once with row-contiguous accesses (z rows are 2w+1 floats long, so nothing is 16-byte aligned in HBM) and each thread
// produces 4 consecutive outputs from LDS.
constexpr int BE_TW = 128, BE_TH = 16, BE_ZW = BE_TW + 3, BE_ZH = BE_TH + 3;

__global__ __launch_bounds__(256) void blur_epilogue_kernel(float* __restrict__ out, const float* __restrict__ z, const float* __restrict__ blur,
                                                            const float* __restrict__ d, const float* __restrict__ noise, int noise_bstride,
                                                            const float* __restrict__ noise_weight, const float* __restrict__ act_bias, int act,
                                                            int cout, int ho, int wo) {
    __shared__ float kf[16];
    __shared__ float zt[BE_ZH * BE_ZW];
    if (threadIdx.x < 16) kf[threadIdx.x] = blur[15 - threadIdx.x];   // kf[ty*4+tx] = blur[3-ty][3-tx]
    const int zh = ho + 1, zw = wo + 1;
    const int plane = blockIdx.z;  // b*cout + co
    const int b = plane / cout, co = plane - b * cout;
    const int ox0 = blockIdx.x * BE_TW, oy0 = blockIdx.y * BE_TH;
    const float* zp = z + (size_t)plane * zh * zw;
    constexpr int NLD = (BE_ZH * BE_ZW + 255) / 256;
    float ld[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {          // all loads first (independent, in flight together), then the LDS writes
        const int e = threadIdx.x + k * 256;
        const int r = e / BE_ZW, c = e - r * BE_ZW;
        const int zy = oy0 + r - 1, zx = ox0 + c - 1;
        const bool ok = e < BE_ZH * BE_ZW && zy >= 0 && zy < zh && zx >= 0 && zx < zw;
        const float v = zp[ok ? (size_t)zy * zw + zx : 0];
        ld[k] = ok ? v : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = threadIdx.x + k * 256;
        if (e < BE_ZH * BE_ZW) zt[e] = ld[k];
    }
    __syncthreads();
    const float dd = d ? d[plane] : 1.f;
    const float nw = noise ? noise_weight[0] : 0.f;
    const float bi = act_bias ? act_bias[co] : 0.f;
    const int ly = threadIdx.x >> 4;
    const int oy = oy0 + ly;
    if (oy >= ho) return;
    float nzv[2][4];   // loads before stores (one in-order vmcnt for both on gfx9)
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ox = ox0 + (threadIdx.x & 15) * 4 + half * 64 + k;
            nzv[half][k] = (noise && ox < wo) ? nw * noise[(size_t)b * noise_bstride + (size_t)oy * wo + ox] : 0.f;
        }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int lx = (threadIdx.x & 15) * 4 + half * 64;
        const int ox = ox0 + lx;
        if (ox >= wo) continue;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ty = 0; ty < 4; ++ty) {
            float row[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) row[j] = zt[(ly + ty) * BE_ZW + lx + j];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int tx = 0; tx < 4; ++tx) a[k] += row[k + tx] * kf[ty * 4 + tx];
        }
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = a[k] * dd + bi + nzv[half][k];
            if (act) v = (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
            r[k] = v;
        }
        float* op = out + ((size_t)plane * ho + oy) * wo + ox;
        if (ox + 3 < wo && (wo & 3) == 0) {
            *reinterpret_cast<float4*>(op) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            for (int k = 0; k < 4 && ox + k < wo; ++k) op[k] = r[k];
        }
    }
}

extern "C" int e4s_blur_epilogue(float* out, const float* z, const float* blur, const float* d, const float* noise, int noise_bs,
                                 const float* noise_weight, const float* act_bias, int act, int bs, int cout, int ho, int wo, void* stream) {
    E4S_REQUIRE(out && z && blur, "blur_epilogue: null tensor");
    E4S_REQUIRE(bs >= 0 && cout >= 1 && ho >= 1 && wo >= 1 && (int64_t)bs * cout <= 65535, "blur_epilogue: bad size");
    E4S_REQUIRE(!noise || (noise_weight && (noise_bs == 1 || noise_bs == bs)), "blur_epilogue: noise needs its weight and batch 1 or bs");
    if (bs == 0) return 0;
    dim3 grid(cdiv(wo, BE_TW), cdiv(ho, BE_TH), bs * cout);
    hipLaunchKernelGGL(blur_epilogue_kernel, grid, dim3(256), 0, (hipStream_t)stream, out, z, blur, d, noise, (noise && noise_bs > 1) ? ho * wo : 0,
                       noise_weight, act_bias, act, cout, ho, wo);
    return check_launch("blur_epilogue");
}
