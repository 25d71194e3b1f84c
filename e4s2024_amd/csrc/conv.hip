// a8 / a9: plain 2-D convolution (encoder IR-SE units, BiSeNet / ResNet-18) as an implicit GEMM on fp32 MFMA
// (v_mfma_f32_32x32x2_f32), NCHW fp32 in and out.
//
//   D[co][pix] = sum_{k=(ci,tap)} Wt[co][k] * X[k][pix]        (same operand roles as modconv.hip: lanes own pixels)
//
// Fusions: InstanceNorm of the INPUT applied while staging ((x-mean)*rstd for in-bounds pixels, padding stays 0),
// channel-concatenated input (two source tensors), folded-BatchNorm bias, residual add, ReLU / PReLU epilogue.
// Tile shape is chosen per launch so that small feature maps (32x32, 16x16) still put >= ~200 workgroups on the chip.
#include <stdlib.h>

#include "common.h"

using namespace e4s;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------ weight prep
// wt[ci][tap][co] = weight[co][ci][tap] * g[co],  bias_out[co] = beta - mean*g (+ conv_bias*g),  g = gamma / sqrt(var + eps)
__global__ __launch_bounds__(256) void conv_prep_kernel(float* __restrict__ wt, float* __restrict__ bias_out, const float* __restrict__ weight,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                        const float* __restrict__ conv_bias, int cout, int cin, int kk) {
    const int64_t total = (int64_t)cin * kk * cout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % cout);
        const int64_t r = i / cout;
        const int tap = (int)(r % kk);
        const int ci = (int)(r / kk);
        const float g = var ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
        wt[i] = weight[((size_t)co * cin + ci) * kk + tap] * g;
    }
    if (bias_out) {
        for (int co = blockIdx.x * 256 + threadIdx.x; co < cout; co += gridDim.x * 256) {
            const float g = var ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
            float b = var ? beta[co] - mean[co] * g : 0.f;
            if (conv_bias) b += conv_bias[co] * g;
            bias_out[co] = b;
        }
    }
}

extern "C" int e4s_conv_prep_weights(float* wt, float* bias_out, const float* weight, const float* bn_gamma, const float* bn_beta,
                                     const float* bn_mean, const float* bn_var, float bn_eps, const float* conv_bias, int cout, int cin, int kh,
                                     int kw, void* stream) {
    E4S_REQUIRE(wt && weight, "conv_prep_weights: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= 1 && kh >= 1 && kw >= 1, "conv_prep_weights: bad size");
    const bool bn = bn_var != nullptr;
    E4S_REQUIRE(!bn || (bn_gamma && bn_beta && bn_mean && bias_out), "conv_prep_weights: BatchNorm fold needs gamma, beta, mean, var and bias_out");
    E4S_REQUIRE(!conv_bias || bias_out, "conv_prep_weights: conv bias needs bias_out");
    const int64_t total = (int64_t)cin * kh * kw * cout;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(conv_prep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wt, bias_out, weight, bn_gamma, bn_beta, bn_mean, bn_var,
                       bn_eps, conv_bias, cout, cin, kh * kw);
    return check_launch("conv_prep_weights");
}

// ------------------------------------------------------------------------------------ the conv kernel
struct Conv2dParams {
    float* out;
    const float* x0;       // channels [0, cin0)
    const float* x1;       // channels [cin0, cin) or NULL
    const float* wt;       // [cin][KS*KS][cout]
    const float* bias;     // [cout] or NULL
    const float* in_mean;  // [bs][cin] or NULL  (instance-norm-on-load)
    const float* in_rstd;
    const float* slope;    // [cout] PReLU slopes (act == 2)
    const float* residual; // [bs][cout][ho][wo] or NULL
    int act;               // 0 none, 1 relu, 2 prelu
    int bs, cin, cin0, cout, h, w, ho, wo, pad;
    int tiles_x, tiles_y;
};

template <int KS, int S, int CKK, int CB, int PB, int WC, int WP, int LOG_TW>
struct C2Cfg {
    static constexpr int KK = KS * KS;
    static constexpr int TN = WC * CB * 32;
    static constexpr int NPB = WP * PB;
    static constexpr int TW = 1 << LOG_TW;
    static constexpr int RPB = 32 >> LOG_TW;
    static constexpr int TH = NPB * RPB;
    static constexpr int PW = (TW - 1) * S + KS, PH = (TH - 1) * S + KS;
    static constexpr int PATCH = PH * PW;
    static constexpr int EPT = (PATCH + 255) / 256;
    static constexpr int XS = CKK * PATCH, WS = CKK * KK * TN;
    static constexpr int LDS_FLOATS = XS + WS;
    static_assert(WC * WP == 4, "256-thread blocks");
    static_assert(CKK % 2 == 0, "K pairs");
    static_assert(LDS_FLOATS * 4 <= 64 * 1024, "static LDS limit");
};

template <int KS, int S, int CKK, int CB, int PB, int WC, int WP, int LOG_TW>
__global__ __launch_bounds__(256) void conv2d_kernel(const Conv2dParams p) {
    using C = C2Cfg<KS, S, CKK, CB, PB, WC, WP, LOG_TW>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    float* xs = lds;
    float* ws = lds + C::XS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    const int wc = wave / WP, wp = wave % WP;
    const int tile = blockIdx.x;
    const int oy0 = (tile / p.tiles_x) * C::TH, ox0 = (tile % p.tiles_x) * C::TW;
    const int co0 = blockIdx.y * C::TN;
    const int b = blockIdx.z;
    const int hw = p.h * p.w;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;

    int goff[C::EPT];
    bool ginb[C::EPT];
#pragma unroll
    for (int j = 0; j < C::EPT; ++j) {
        const int e = tid + j * 256;
        const int py = e / C::PW, px = e - py * C::PW;
        const int gy = iy0 + py, gx = ix0 + px;
        ginb[j] = (e < C::PATCH) && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        goff[j] = gy * p.w + gx;
    }
    const int cin1 = p.cin - p.cin0;
    const float* xb0 = p.x0 + (size_t)b * p.cin0 * hw;
    const float* xb1 = p.x1 ? p.x1 + (size_t)b * cin1 * hw : nullptr;

    int xoff[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int ty = pbk * C::RPB + (l5 >> LOG_TW), tx = l5 & (C::TW - 1);
        xoff[q] = ty * S * C::PW + tx * S;
    }

    f32x16 acc[CB][PB];
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
        for (int q = 0; q < PB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    const bool wvec = (p.cout & 3) == 0;

    // Register stage of the NEXT K chunk: global loads are unconditional (clamped addresses, masked when written to LDS) and issued
    // before the current chunk's MFMAs, so their latency hides behind the matrix work instead of sitting between two barriers.
    // Same arithmetic, same order as before — results are bit-identical.
    constexpr int NV = CKK * C::KK * C::TN / 4;
    constexpr int WPT = (NV + 255) / 256;
    float xr[CKK][C::EPT];
    float wr[WPT][4];
    float mur[CKK], rsr[CKK];
    int goffs[C::EPT];
#pragma unroll
    for (int j = 0; j < C::EPT; ++j) goffs[j] = ginb[j] ? goff[j] : 0;
    auto load_chunk = [&](int ci0) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CKK; ++c) {
            const int ci = (ci0 + c < p.cin) ? ci0 + c : p.cin - 1;
            const float* xc = (ci < p.cin0) ? xb0 + (size_t)ci * hw : xb1 + (size_t)(ci - p.cin0) * hw;
#pragma unroll
            for (int j = 0; j < C::EPT; ++j) xr[c][j] = xc[goffs[j]];
            mur[c] = p.in_mean ? p.in_mean[(size_t)b * p.cin + ci] : 0.f;
            rsr[c] = p.in_mean ? p.in_rstd[(size_t)b * p.cin + ci] : 1.f;
        }
        if (wvec) {
#pragma unroll
            for (int v = 0; v < WPT; ++v) {
                int idx = tid + v * 256;
                idx = idx < NV ? idx : NV - 1;
                const int n4 = idx % (C::TN / 4);
                int ct = idx / (C::TN / 4);
                const int ctmax = (p.cin - ci0) * C::KK - 1;            // last valid (channel, tap) row of this chunk
                ct = ct < ctmax ? ct : ctmax;
                const int co = (co0 + n4 * 4 < p.cout) ? co0 + n4 * 4 : 0;
                const float4 t4 = *reinterpret_cast<const float4*>(p.wt + ((size_t)ci0 * C::KK + ct) * p.cout + co);
                wr[v][0] = t4.x; wr[v][1] = t4.y; wr[v][2] = t4.z; wr[v][3] = t4.w;
            }
        }
    };
    auto store_chunk = [&](int ci0) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CKK; ++c) {
            const bool cok = ci0 + c < p.cin;
#pragma unroll
            for (int j = 0; j < C::EPT; ++j) {
                const int e = tid + j * 256;
                if (e < C::PATCH) xs[c * C::PATCH + e] = (cok && ginb[j]) ? (xr[c][j] - mur[c]) * rsr[c] : 0.f;
            }
        }
        if (wvec) {
#pragma unroll
            for (int v = 0; v < WPT; ++v) {
                const int idx = tid + v * 256;
                if (idx < NV) {
                    const int n4 = idx % (C::TN / 4);
                    const int ct = idx / (C::TN / 4);
                    const bool ok = ci0 + ct / C::KK < p.cin && co0 + n4 * 4 < p.cout;
                    *reinterpret_cast<float4*>(ws + ct * C::TN + n4 * 4) = ok ? make_float4(wr[v][0], wr[v][1], wr[v][2], wr[v][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {   // cout not a multiple of 4 (the 19-class heads): small layers, staged straight from global memory
            constexpr int NS = CKK * C::KK * C::TN;
            for (int v = tid; v < NS; v += 256) {
                const int n = v % C::TN;
                const int ct = v / C::TN;
                const int c = ct / C::KK;
                ws[v] = (ci0 + c < p.cin && co0 + n < p.cout) ? p.wt[((size_t)ci0 * C::KK + ct) * p.cout + co0 + n] : 0.f;
            }
        }
    };

    load_chunk(0);
    for (int ci0 = 0; ci0 < p.cin; ci0 += CKK) {
        __syncthreads();
        store_chunk(ci0);
        __syncthreads();
        if (ci0 + CKK < p.cin) load_chunk(ci0 + CKK);

#pragma unroll
        for (int cp = 0; cp < CKK / 2; ++cp) {
            const int ci = 2 * cp + khalf;
            const float* xrow = xs + ci * C::PATCH;
            const float* wrow = ws + ci * C::KK * C::TN + wc * CB * 32 + l5;
#pragma unroll
            for (int tap = 0; tap < C::KK; ++tap) {
                const int toff = (tap / KS) * C::PW + (tap % KS);
                float bv[PB], av[CB];
#pragma unroll
                for (int q = 0; q < PB; ++q) bv[q] = xrow[xoff[q] + toff];
#pragma unroll
                for (int i = 0; i < CB; ++i) av[i] = wrow[tap * C::TN + i * 32];
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int q = 0; q < PB; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[q], acc[i][q], 0, 0, 0);
            }
        }
    }

    // Epilogue in two passes: every global load (bias, residual, slope) first, then the stores.  gfx9 tracks loads and stores with
    // one in-order counter (vmcnt), so a load issued after a store cannot complete its wait before that store has reached memory.
    const size_t ohw = (size_t)p.ho * p.wo;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int oy = oy0 + pbk * C::RPB + (l5 >> LOG_TW), ox = ox0 + (l5 & (C::TW - 1));
        const bool pix_ok = oy < p.ho && ox < p.wo;
        const size_t opix = (size_t)oy * p.wo + ox;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (pix_ok && co < p.cout) {
                    float v = acc[i][q][r];
                    if (p.bias) v += p.bias[co];
                    if (p.residual) v += p.residual[((size_t)b * p.cout + co) * ohw + opix];
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    if (p.act == 2) v = v > 0.f ? v : v * p.slope[co];
                    acc[i][q][r] = v;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int oy = oy0 + pbk * C::RPB + (l5 >> LOG_TW), ox = ox0 + (l5 & (C::TW - 1));
        if (oy >= p.ho || ox >= p.wo) continue;
        const size_t opix = (size_t)oy * p.wo + ox;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co < p.cout) p.out[((size_t)b * p.cout + co) * ohw + opix] = acc[i][q][r];
            }
        }
    }
}

template <int KS, int S, int CKK, int CB, int PB, int WC, int WP, int LOG_TW>
static int launch2d(Conv2dParams& p, hipStream_t st) {
    using C = C2Cfg<KS, S, CKK, CB, PB, WC, WP, LOG_TW>;
    p.tiles_x = cdiv(p.wo, C::TW);
    p.tiles_y = cdiv(p.ho, C::TH);
    dim3 grid(p.tiles_x * p.tiles_y, cdiv(p.cout, C::TN), p.bs);
    hipLaunchKernelGGL((conv2d_kernel<KS, S, CKK, CB, PB, WC, WP, LOG_TW>), grid, dim3(256), 0, st, p);
    return check_launch("conv2d");
}

// blocks a (TN x TM-pixel) tiling would launch
static int64_t nblocks(const Conv2dParams& p, int tn, int th, int tw) {
    return (int64_t)cdiv(p.wo, tw) * cdiv(p.ho, th) * cdiv(p.cout, tn) * p.bs;
}

template <int KS, int S, int CKK>
static int dispatch2d(Conv2dParams& p, hipStream_t st) {
    constexpr int64_t FILL = 192;  // ~0.75 x 256 CUs
    if (p.wo >= 32) {
        if (p.cout > 64 && nblocks(p, 128, 4, 32) >= FILL) return launch2d<KS, S, CKK, 2, 2, 2, 2, 5>(p, st);   // 128 co x 128 px
        if (p.cout > 32 && nblocks(p, 64, 8, 32) >= FILL) return launch2d<KS, S, CKK, 2, 2, 1, 4, 5>(p, st);    //  64 co x 256 px
        return launch2d<KS, S, CKK, 1, 1, 2, 2, 5>(p, st);                                                       //  64 co x  64 px
    }
    return launch2d<KS, S, CKK, 1, 1, 2, 2, 4>(p, st);                                                           //  64 co x 64 px (16 x 4)
}

// ------------------------------------------------------------------------------------ the 3-channel stems
// Cin <= 4 with a 3 x 3 kernel (the regional-style encoder's input layer, psp_encoders.py:335: 3 -> 64): K = 27 is far too short for the implicit-GEMM tiles above (their
// 16-channel chunks are 13 / 16 padding), and the layer is bound by its 268 MB of output per batch of 16 images (351 us on the implicit GEMM, 125 us here).  One thread = one output pixel x 64 output channels in registers: its k^2 Cin inputs arrive row by row from global memory (coalesced across the threads of
// a row), the weights [k][64] sit in LDS and are read as broadcast float4s, plain fp32 FMAs (two channels per v_pk_fma_f32), every store a fully used line per channel.
template <int KS>
__global__ __launch_bounds__(256) void conv_small_cin_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                             const float* __restrict__ slope, int act, int cin, int cout, int h, int w, int ho, int wo, int stride, int pad) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [cin * KS * KS][64], then bias [64], slope [64]
    const int nk = cin * KS * KS;
    const int co0 = blockIdx.y * 64, b = blockIdx.z;
    for (int e = threadIdx.x; e < nk * 64; e += 256) {
        const int k = e >> 6, n = e & 63;
        wl[e] = co0 + n < cout ? wt[(size_t)k * cout + co0 + n] : 0.f;
    }
    float* bl = wl + nk * 64;
    float* sl = bl + 64;
    if (threadIdx.x < 64) {
        const int co = co0 + threadIdx.x;
        bl[threadIdx.x] = (bias && co < cout) ? bias[co] : 0.f;
        sl[threadIdx.x] = (act == 2 && co < cout) ? slope[co] : (act == 1 ? 0.f : 1.f);      // (ReLU = PReLU with slope 0)
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const bool ok = pix < ho * wo;
    const int oy = ok ? pix / wo : 0, ox = ok ? pix - (pix / wo) * wo : 0;
    f2 acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = (f2){bl[2 * j], bl[2 * j + 1]};
    const int ix0 = ox * stride - pad, iy0 = oy * stride - pad;
    for (int ci = 0; ci < cin; ++ci) {
        const float* xp = x + ((size_t)b * cin + ci) * h * w;
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = iy0 + ky;
            const bool row_ok = ok && iy >= 0 && iy < h;
            float in[KS];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int ix = ix0 + kx;
                const bool in_ok = row_ok && ix >= 0 && ix < w;
                const float v = xp[in_ok ? (size_t)iy * w + ix : 0];      // (unconditional load from a valid address)
                in[kx] = in_ok ? v : 0.f;
            }
            const float4* wrow = reinterpret_cast<const float4*>(wl + (size_t)((ci * KS + ky) * KS) * 64);
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const f2 vv = (f2){in[kx], in[kx]};
#pragma unroll
                for (int n4 = 0; n4 < 16; ++n4) {
                    const float4 w4 = wrow[kx * 16 + n4];
                    acc[2 * n4] = (f2){w4.x, w4.y} * vv + acc[2 * n4];
                    acc[2 * n4 + 1] = (f2){w4.z, w4.w} * vv + acc[2 * n4 + 1];
                }
            }
        }
    }
    if (!ok) return;
    float* op = out + ((size_t)b * cout + co0) * ho * wo + pix;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int n = 2 * j + e;
            if (co0 + n >= cout) break;
            float v = acc[j][e];
            if (act) v = v > 0.f ? v : v * sl[n];
            op[(size_t)n * ho * wo] = v;
        }
    }
}

template <int KS>
static int launch_small_cin(float* out, const float* x, const float* wt, const float* bias, const float* slope, int act, int bs, int cin, int cout, int h, int w, int ho, int wo,
                            int stride, int pad, hipStream_t st) {
    const int lds = (cin * KS * KS * 64 + 128) * 4;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_cin_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (4 * KS * KS * 64 + 128) * 4);
    if (attr != hipSuccess) return fail((int)attr, "conv2d: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL(conv_small_cin_kernel<KS>, dim3(cdiv(ho * wo, 256), cdiv(cout, 64), bs), dim3(256), lds, st, out, x, wt, bias, slope, act, cin, cout, h, w, ho, wo, stride, pad);
    return check_launch("conv2d (small cin)");
}

extern "C" int e4s_conv2d(float* out, const float* x0, const float* x1, int cin0, const float* wt, const float* bias, const float* in_mean,
                          const float* in_rstd, const float* prelu_slope, const float* residual, int act, int bs, int cin, int cout, int h, int w,
                          int ks, int stride, int pad, void* stream) {
    E4S_REQUIRE(out && x0 && wt, "conv2d: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= 1 && cout >= 1 && h >= 1 && w >= 1, "conv2d: bad size");
    E4S_REQUIRE(stride == 1 || stride == 2, "conv2d: stride %d not supported (1 or 2)", stride);
    E4S_REQUIRE(pad >= 0 && pad <= ks, "conv2d: bad padding");
    E4S_REQUIRE(act >= 0 && act <= 2 && (act != 2 || prelu_slope), "conv2d: bad activation");
    E4S_REQUIRE((in_mean == nullptr) == (in_rstd == nullptr), "conv2d: in_mean and in_rstd go together");
    E4S_REQUIRE(x1 ? (cin0 >= 1 && cin0 < cin) : true, "conv2d: bad channel split");
    if (bs == 0) return 0;
    Conv2dParams p;
    p.out = out; p.x0 = x0; p.x1 = x1; p.wt = wt; p.bias = bias; p.in_mean = in_mean; p.in_rstd = in_rstd; p.slope = prelu_slope;
    p.residual = residual; p.act = act; p.bs = bs; p.cin = cin; p.cin0 = x1 ? cin0 : cin; p.cout = cout; p.h = h; p.w = w; p.pad = pad;
    p.ho = (h + 2 * pad - ks) / stride + 1;
    p.wo = (w + 2 * pad - ks) / stride + 1;
    E4S_REQUIRE(p.ho >= 1 && p.wo >= 1, "conv2d: empty output");
    hipStream_t st = (hipStream_t)stream;
    // the 3 x 3 stem (K = 27): the direct kernel above.  (The 7 x 7 stride-2 stem, K = 147, stays on the implicit GEMM: 19.7 GFLOP per 16 images are 290 us of fp32 VALU at
    // its peak — measured 424 us in this form, 568 us with four pixels per thread, 1 580 us with the weights as scalar loads, against 403 us here.)
    if (cin <= 4 && !x1 && !in_mean && !residual && ks == 3 && (int64_t)p.ho * p.wo >= 4096)
        return launch_small_cin<3>(out, x0, wt, bias, prelu_slope, act, bs, cin, cout, h, w, p.ho, p.wo, stride, pad, st);
    if (ks == 3 && stride == 1) return dispatch2d<3, 1, 8>(p, st);
    if (ks == 3 && stride == 2) return dispatch2d<3, 2, 8>(p, st);
    if (ks == 1 && stride == 1) return dispatch2d<1, 1, 32>(p, st);
    if (ks == 1 && stride == 2) return dispatch2d<1, 2, 8>(p, st);
    if (ks == 7 && stride == 2) return dispatch2d<7, 2, 2>(p, st);
    return fail(E4S_ERR_ARG, "conv2d: kernel %dx%d stride %d not supported", ks, ks, stride);
}

// ====================================================================================================================
// Split-bf16 variant (see modconv_sb.hip for the numerics): operands split into bf16 hi + lo, a*b = hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The activation operand does not depend on the output pixel here, so it is
// transformed (InstanceNorm-on-load), split ONCE while staging and kept in LDS as two bf16 planes [patch pixel][16 channels];
// the main loop is LDS reads + MFMAs only.  Weights are split at preparation time: [Cin/16][tap][half][Cout][8] bf16 x 2.
// Used for every 3x3 / 1x1 convolution with Cin >= 16 (the 3-channel stems stay on the fp32 kernel above).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int CKS2 = 16;

__device__ __forceinline__ unsigned c2_pack_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ void c2_split2(float t0, float t1, unsigned& hi, unsigned& lo) {
    hi = c2_pack_bf16(t0, t1);
    lo = c2_pack_bf16(t0 - __builtin_bit_cast(float, hi << 16), t1 - __builtin_bit_cast(float, hi & 0xffff0000u));
}
typedef _Float16 f16x8c __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned c2_pack_f16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2c));      // v_cvt_pk_f16_f32 (RNE)
}
// the f16 two-term split: a = a1 + a2 with a1 = f16(a), a2 = f16(a - a1): 22 significand bits (while a2 stays a normal f16: |a| >= 2^-3)
__device__ __forceinline__ void c2_split2_f16(float t0, float t1, unsigned& hi, unsigned& lo) {
    const f16x2c h = __builtin_convertvector((f32x2){t0, t1}, f16x2c);
    hi = __builtin_bit_cast(unsigned, h);
    lo = c2_pack_f16(t0 - (float)h[0], t1 - (float)h[1]);
}

__global__ __launch_bounds__(256) void conv_prep_sb_kernel(uint16_t* __restrict__ whi, uint16_t* __restrict__ wlo, uint16_t* __restrict__ wlo2, float* __restrict__ bias_out,
                                                           const float* __restrict__ weight, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ var, float eps, const float* __restrict__ conv_bias,
                                                           int cout, int cin, int kk, int f16 = 0, float wscale = 1.f) {
    const int nchunk = (cin + CKS2 - 1) / CKS2;
    const int64_t total = (int64_t)nchunk * kk * 2 * cout * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i & 7);
        int64_t r = i >> 3;
        const int co = (int)(r % cout); r /= cout;
        const int half = (int)(r & 1); r >>= 1;
        const int tap = (int)(r % kk);
        const int chunk = (int)(r / kk);
        const int ci = chunk * CKS2 + half * 8 + e;
        float v = 0.f;
        if (ci < cin) {
            const float g = var ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
            v = weight[((size_t)co * cin + ci) * kk + tap] * g;
        }
        if (f16) {      // w * 2^k = w1 + w2 in f16 (the power of two keeps w2 a normal f16; the kernel's epilogue takes it out again)
            unsigned h, l;
            c2_split2_f16(v * wscale, 0.f, h, l);
            whi[i] = (uint16_t)(h & 0xffffu);
            wlo[i] = (uint16_t)(l & 0xffffu);
            continue;
        }
        const unsigned hp = c2_pack_bf16(v, 0.f) & 0xffffu;
        whi[i] = (uint16_t)hp;
        const float r1 = v - __builtin_bit_cast(float, hp << 16);
        const unsigned mp = c2_pack_bf16(r1, 0.f) & 0xffffu;
        wlo[i] = (uint16_t)mp;
        if (wlo2) wlo2[i] = (uint16_t)(c2_pack_bf16(r1 - __builtin_bit_cast(float, mp << 16), 0.f) & 0xffffu);   // third term of the 3-way split
    }
    if (bias_out) {
        for (int co = blockIdx.x * 256 + threadIdx.x; co < cout; co += gridDim.x * 256) {
            const float g = var ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
            float b = var ? beta[co] - mean[co] * g : 0.f;
            if (conv_bias) b += conv_bias[co] * g;
            bias_out[co] = b;
        }
    }
}

extern "C" int e4s_conv_prep_weights_sb(uint16_t* whi, uint16_t* wlo, float* bias_out, const float* weight, const float* bn_gamma,
                                        const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps, const float* conv_bias,
                                        int cout, int cin, int kh, int kw, void* stream) {
    E4S_REQUIRE(whi && wlo && weight, "conv_prep_weights_sb: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= 1 && kh >= 1 && kw >= 1, "conv_prep_weights_sb: bad size");
    const bool bn = bn_var != nullptr;
    E4S_REQUIRE(!bn || (bn_gamma && bn_beta && bn_mean && bias_out), "conv_prep_weights_sb: BatchNorm fold needs gamma, beta, mean, var and bias_out");
    E4S_REQUIRE(!conv_bias || bias_out, "conv_prep_weights_sb: conv bias needs bias_out");
    const int64_t total = (int64_t)cdiv(cin, CKS2) * kh * kw * 2 * cout * 8;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(conv_prep_sb_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, whi, wlo, (uint16_t*)nullptr, bias_out, weight, bn_gamma,
                       bn_beta, bn_mean, bn_var, bn_eps, conv_bias, cout, cin, kh * kw);
    return check_launch("conv_prep_weights_sb");
}

// Three-way split: w = w0 + w1 + w2 (each bf16, RNE of the running residual) represents an fp32 weight to ~2^-25.
extern "C" int e4s_conv_prep_weights_sb3(uint16_t* w0, uint16_t* w1, uint16_t* w2, float* bias_out, const float* weight, const float* bn_gamma,
                                         const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps, const float* conv_bias,
                                         int cout, int cin, int kh, int kw, void* stream) {
    E4S_REQUIRE(w0 && w1 && w2 && weight, "conv_prep_weights_sb3: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= 1 && kh >= 1 && kw >= 1, "conv_prep_weights_sb3: bad size");
    const bool bn = bn_var != nullptr;
    E4S_REQUIRE(!bn || (bn_gamma && bn_beta && bn_mean && bias_out), "conv_prep_weights_sb3: BatchNorm fold needs gamma, beta, mean, var and bias_out");
    E4S_REQUIRE(!conv_bias || bias_out, "conv_prep_weights_sb3: conv bias needs bias_out");
    const int64_t total = (int64_t)cdiv(cin, CKS2) * kh * kw * 2 * cout * 8;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(conv_prep_sb_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w0, w1, w2, bias_out, weight, bn_gamma, bn_beta, bn_mean,
                       bn_var, bn_eps, conv_bias, cout, cin, kh * kw);
    return check_launch("conv_prep_weights_sb3");
}

struct Conv2dSbParams {
    float* out;
    const float* x0;
    const float* x1;
    const uint4* wsl[3];   // weight slabs: hi, lo (, lo2 for the three-way split)
    const float* bias;
    const float* in_mean;
    const float* in_rstd;
    const float* slope;
    const float* residual;
    int act;
    int bs, cin, cin0, cout, h, w, ho, wo, pad;
    int tiles_x, tiles_y;
    float out_scale;       // (NS = 4) 2^-k of the weights' power-of-two pre-scale; 1 otherwise
    int* flags;            // (NS = 4) flags[0] |= 1, flags[1] += 1 when an activation leaves the f16 range (ops.MxGuard re-runs the pass on the bf16 x 3 kernels); or NULL
};

// NS = number of bf16 terms per operand.  NS = 2: a*b ~ a0*b0 + a0*b1 + a1*b0 (3 MFMAs per 16-deep step, ~2^-17 per product).
// NS = 3: a0*b0 + a0*b1 + a1*b0 + a0*b2 + a2*b0 + a1*b1 (6 MFMAs, ~2^-24: fp32-class, for the face parser whose argmax must not move)
// — still 2.7x less matrix-pipe time than the 8 fp32 MFMAs of the exact kernel.
// NS = 4 (round 3): TWO f16 terms per operand, a1*b1 + a1*b2 + a2*b1 on v_mfma_f32_32x32x16_f16 — 22 significand bits per operand, ~2^-23 per product:
// the fp32-class error of NS = 3 at HALF its MFMAs (f16 carries 11 bits per term where bf16 carries 8).  Weights are pre-scaled by a power of two so
// that their second term stays a normal f16 (e4s_conv_prep_weights_f16x3); the activations of the networks on this path are O(1) after their norms.
template <int KS, int S, int CB, int PB, int WC, int WP, int LOG_TW, int NS = 2>
struct C2SbCfg {
    static constexpr int NP = NS == 4 ? 2 : NS;      // operand planes / weight slabs
    static constexpr int KK = KS * KS;
    static constexpr int TN = WC * CB * 32;
    static constexpr int NPB = WP * PB;
    static constexpr int TW = 1 << LOG_TW;
    static constexpr int RPB = 32 >> LOG_TW;
    static constexpr int TH = NPB * RPB;
    static constexpr int PW = (TW - 1) * S + KS, PH = (TH - 1) * S + KS;
    static constexpr int PATCH = PH * PW;
    static constexpr int NT = 64 * WC * WP;          // threads per workgroup (256 or 512)
    static constexpr int EPT = (PATCH + NT - 1) / NT;
    static constexpr int W4 = NP * KK * 2 * TN;      // uint4: [term][tap][half][TN]
    static constexpr int WPT = (W4 + NT - 1) / NT;
    static constexpr int LDS_BYTES = W4 * 16 + PATCH * 32 * NP;
    static_assert(WC * WP == 4 || WC * WP == 8, "256- or 512-thread workgroups");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per CU");
};

E4S_PROF_DECL(g_prof_conv)
#ifdef E4S_PHASE_PROF
extern "C" E4S_API int e4s_prof_read_conv(long long* host, int64_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prof_conv), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
extern "C" E4S_API int e4s_prof_clear_conv() {
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_prof_conv));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(ptr, 0, sizeof(long long) * (size_t)E4S_PROF_BLOCKS * E4S_PROF_SLOTS);
}
#endif

template <int KS, int S, int CB, int PB, int WC, int WP, int LOG_TW, int PF = 1, int NS = 2>
__global__ __launch_bounds__(64 * WC * WP, 2) void conv2d_sb_kernel(const Conv2dSbParams p) {
    using C = C2SbCfg<KS, S, CB, PB, WC, WP, LOG_TW, NS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* wsm = reinterpret_cast<uint4*>(lds_raw);                 // [NS][KK][2][TN]
    uint4* xpl = reinterpret_cast<uint4*>(lds_raw + C::W4 * 16);    // NS planes of [PATCH][2]: 16 bf16 per patch pixel, halves swizzled

    const int tid = threadIdx.x;
    E4S_PROF_MARK(g_prof_conv, 0);
    const int lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    const int wc = wave / WP, wp = wave % WP;
    const int tile = blockIdx.x;
    const int oy0 = (tile / p.tiles_x) * C::TH, ox0 = (tile % p.tiles_x) * C::TW;
    const int co0 = blockIdx.y * C::TN;
    const int b = blockIdx.z;
    const int hw = p.h * p.w;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    const int nchunk = (p.cin + CKS2 - 1) / CKS2;

    int goff[C::EPT];
    bool ginb[C::EPT];
#pragma unroll
    for (int j = 0; j < C::EPT; ++j) {
        const int e = tid + j * C::NT;
        const int py = e / C::PW, px = e - py * C::PW;
        const int gy = iy0 + py, gx = ix0 + px;
        ginb[j] = (e < C::PATCH) && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        goff[j] = gy * p.w + gx;
    }
    const int cin1 = p.cin - p.cin0;
    const float* xb0 = p.x0 + (size_t)b * p.cin0 * hw;
    const float* xb1 = p.x1 ? p.x1 + (size_t)b * cin1 * hw : nullptr;

    int xoff[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int ty = pbk * C::RPB + (l5 >> LOG_TW), tx = l5 & (C::TW - 1);
        xoff[q] = ty * S * C::PW + tx * S;
    }

    f32x16 acc[CB][PB];
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
        for (int q = 0; q < PB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    // Register stages of the NEXT TWO chunks (set 0 / set 1 alternate): with one or two workgroups per CU a single chunk of MFMAs
    // (~1 us) does not cover a cold HBM/L2 miss (2-4 us) on the weight slabs, so loads run two chunks ahead.  The per-channel
    // InstanceNorm statistics of a chunk travel with it (they used to be fetched at conversion time, on the critical path).
    // (PF = 1 keeps one stage where the second would spill: the 64 co x 256 px tile and the large stride-2 patches.)
    float xr[PF][CKS2][C::EPT];
    unsigned wr[PF][C::WPT][4];   // scalar components: an array of uint4 ends up in scratch
    float mu[PF][CKS2], rs[PF][CKS2];

    // Unconditional loads (no per-element branches, so they are issued here and really are a prefetch): out-of-image elements read
    // a valid clamped address and are zeroed at conversion time; channels >= cin meet zero-padded weights; output-channel rows
    // >= cout are computed but never stored.
    int goffs[C::EPT];
#pragma unroll
    for (int j = 0; j < C::EPT; ++j) goffs[j] = ginb[j] ? goff[j] : 0;
    const bool split_ok = (p.cin0 % CKS2) == 0;   // a chunk never straddles the two concatenated inputs
    auto load_chunk = [&](int chunk, float (&xs)[CKS2][C::EPT], unsigned (&ws)[C::WPT][4], float (&m)[CKS2], float (&r)[CKS2]) __attribute__((always_inline)) {
        const int ci0 = chunk * CKS2;
        if (split_ok) {
            const float* xcb = (ci0 < p.cin0) ? xb0 + (size_t)ci0 * hw : xb1 + (size_t)(ci0 - p.cin0) * hw;
            const int cmax = ((ci0 < p.cin0) ? p.cin0 : p.cin) - 1 - ci0;
#pragma unroll
            for (int c = 0; c < CKS2; ++c) {
                const float* xc = xcb + (size_t)(c < cmax ? c : cmax) * hw;
#pragma unroll
                for (int j = 0; j < C::EPT; ++j) xs[c][j] = xc[goffs[j]];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CKS2; ++c) {
                const int ci = (ci0 + c < p.cin) ? ci0 + c : p.cin - 1;
                const float* xc = (ci < p.cin0) ? xb0 + (size_t)ci * hw : xb1 + (size_t)(ci - p.cin0) * hw;
#pragma unroll
                for (int j = 0; j < C::EPT; ++j) xs[c][j] = xc[goffs[j]];
            }
        }
        const size_t wbase = (size_t)chunk * C::KK * 2 * p.cout;
#pragma unroll
        for (int v = 0; v < C::WPT; ++v) {
            int idx = tid + v * C::NT;
            idx = idx < C::W4 ? idx : C::W4 - 1;
            const int hl = idx / (C::KK * 2 * C::TN);
            const int rem = idx - hl * C::KK * 2 * C::TN;
            const int th = rem / C::TN, n = rem - th * C::TN;
            const int co = (co0 + n < p.cout) ? co0 + n : p.cout - 1;
            const uint4* slab = hl == 0 ? p.wsl[0] : (hl == 1 ? p.wsl[1] : p.wsl[2]);
            const uint4 t4 = slab[wbase + (size_t)th * p.cout + co];
            ws[v][0] = t4.x; ws[v][1] = t4.y; ws[v][2] = t4.z; ws[v][3] = t4.w;
        }
#pragma unroll
        for (int c = 0; c < CKS2; ++c) {
            const bool on = p.in_mean && ci0 + c < p.cin;
            m[c] = on ? p.in_mean[(size_t)b * p.cin + ci0 + c] : 0.f;
            r[c] = on ? p.in_rstd[(size_t)b * p.cin + ci0 + c] : 1.f;
        }
    };
    bool ovf = false;      // (NS = 4) a staged activation left the f16 range
    auto store_chunk = [&](const float (&xs)[CKS2][C::EPT], const unsigned (&ws)[C::WPT][4], const float (&m)[CKS2], const float (&r)[CKS2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < C::EPT; ++j) {
            const int e = tid + j * C::NT;
            if (e < C::PATCH) {
                unsigned h[8], l[8], l2[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // padding stays exactly 0: the normalisation applies to in-bounds pixels only
                    const float t0 = ginb[j] ? (xs[2 * c][j] - m[2 * c]) * r[2 * c] : 0.f;
                    const float t1 = ginb[j] ? (xs[2 * c + 1][j] - m[2 * c + 1]) * r[2 * c + 1] : 0.f;
                    if constexpr (NS == 4) {
                        c2_split2_f16(t0, t1, h[c], l[c]);
                        const unsigned em = h[c] & 0x7c007c00u;            // an f16 exponent field of 31: the value rounded to infinity
                        ovf |= (em & 0xffffu) == 0x7c00u || (em >> 16) == 0x7c00u;
                    } else c2_split2(t0, t1, h[c], l[c]);
                    if (NS == 3) {   // third term: what the first two leave over
                        const float q0 = (t0 - __builtin_bit_cast(float, h[c] << 16)) - __builtin_bit_cast(float, l[c] << 16);
                        const float q1 = (t1 - __builtin_bit_cast(float, h[c] & 0xffff0000u)) - __builtin_bit_cast(float, l[c] & 0xffff0000u);
                        l2[c] = c2_pack_bf16(q0, q1);
                    }
                }
                const int sw = (e >> 3) & 1;
                uint4* x0p = xpl, *x1p = xpl + 2 * C::PATCH, *x2p = xpl + 4 * C::PATCH;
                x0p[e * 2 + (0 ^ sw)] = make_uint4(h[0], h[1], h[2], h[3]);
                x0p[e * 2 + (1 ^ sw)] = make_uint4(h[4], h[5], h[6], h[7]);
                x1p[e * 2 + (0 ^ sw)] = make_uint4(l[0], l[1], l[2], l[3]);
                x1p[e * 2 + (1 ^ sw)] = make_uint4(l[4], l[5], l[6], l[7]);
                if (NS == 3) {
                    x2p[e * 2 + (0 ^ sw)] = make_uint4(l2[0], l2[1], l2[2], l2[3]);
                    x2p[e * 2 + (1 ^ sw)] = make_uint4(l2[4], l2[5], l2[6], l2[7]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < C::WPT; ++v) {
            const int idx = tid + v * C::NT;
            if (idx < C::W4) wsm[idx] = make_uint4(ws[v][0], ws[v][1], ws[v][2], ws[v][3]);
        }
    };
    auto compute_chunk = [&]() __attribute__((always_inline)) {
        const uint4* whalf = wsm + khalf * C::TN + wc * CB * 32 + l5;
#pragma unroll
        for (int tap = 0; tap < C::KK; ++tap) {
            const int toff = (tap / KS) * C::PW + (tap % KS);
            uint4 bt[C::NP][PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int e = xoff[q] + toff;
                const int slot = e * 2 + (khalf ^ ((e >> 3) & 1));
#pragma unroll
                for (int t = 0; t < C::NP; ++t) bt[t][q] = xpl[t * 2 * C::PATCH + slot];
            }
            uint4 at[C::NP][CB];
#pragma unroll
            for (int t = 0; t < C::NP; ++t)
#pragma unroll
                for (int i = 0; i < CB; ++i) at[t][i] = whalf[t * C::KK * 2 * C::TN + tap * 2 * C::TN + i * 32];
            // products in order of magnitude: (0,0) (0,1) (1,0) [ (0,2) (2,0) (1,1) ]
            constexpr int NPROD = NS == 3 ? 6 : 3;
            constexpr int TA[6] = {0, 0, 1, 0, 2, 1};
            constexpr int TB[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int q = 0; q < PB; ++q) {
                        if constexpr (NS == 4)
                            acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8c, at[TA[pr]][i]), __builtin_bit_cast(f16x8c, bt[TB[pr]][q]), acc[i][q], 0, 0, 0);
                        else
                            acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, at[TA[pr]][i]), __builtin_bit_cast(bf16x8, bt[TB[pr]][q]), acc[i][q], 0, 0, 0);
                    }
        }
    };

    if constexpr (PF == 2) {
        load_chunk(0, xr[0], wr[0], mu[0], rs[0]);
        if (nchunk > 1) load_chunk(1, xr[1], wr[1], mu[1], rs[1]);
        for (int chunk = 0; chunk < nchunk; chunk += 2) {
            __syncthreads();
            store_chunk(xr[0], wr[0], mu[0], rs[0]);
            __syncthreads();
            if (chunk + 2 < nchunk) load_chunk(chunk + 2, xr[0], wr[0], mu[0], rs[0]);
            compute_chunk();
            if (chunk + 1 < nchunk) {
                __syncthreads();
                store_chunk(xr[1], wr[1], mu[1], rs[1]);
                __syncthreads();
                if (chunk + 3 < nchunk) load_chunk(chunk + 3, xr[1], wr[1], mu[1], rs[1]);
                compute_chunk();
            }
        }
    } else {
        load_chunk(0, xr[0], wr[0], mu[0], rs[0]);
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            __syncthreads();
            store_chunk(xr[0], wr[0], mu[0], rs[0]);
            __syncthreads();
            if (chunk == 0) E4S_PROF_MARK(g_prof_conv, 1);
            if (chunk + 1 < nchunk) load_chunk(chunk + 1, xr[0], wr[0], mu[0], rs[0]);
            compute_chunk();
        }
    }

    E4S_PROF_MARK(g_prof_conv, 2);
    if constexpr (NS == 4) {
        if (p.flags && __builtin_amdgcn_ballot_w64(ovf) != 0 && lane == 0) { atomicOr(p.flags, 1); atomicAdd(p.flags + 1, 1); }   // one report per wave
    }
    E4S_PROF_MARK(g_prof_conv, 3);
    // Epilogue in two passes: every global load (bias, residual, slope) first, then the stores.  gfx9 tracks loads and stores with
    // one in-order counter (vmcnt), so a load issued after a store cannot complete its wait before that store has reached memory.
    const size_t ohw = (size_t)p.ho * p.wo;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int oy = oy0 + pbk * C::RPB + (l5 >> LOG_TW), ox = ox0 + (l5 & (C::TW - 1));
        const bool pix_ok = oy < p.ho && ox < p.wo;
        const size_t opix = (size_t)oy * p.wo + ox;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (pix_ok && co < p.cout) {
                    float v = acc[i][q][r];
                    if constexpr (NS == 4) v *= p.out_scale;
                    if (p.bias) v += p.bias[co];
                    if (p.residual) v += p.residual[((size_t)b * p.cout + co) * ohw + opix];
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    if (p.act == 2) v = v > 0.f ? v : v * p.slope[co];
                    acc[i][q][r] = v;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int oy = oy0 + pbk * C::RPB + (l5 >> LOG_TW), ox = ox0 + (l5 & (C::TW - 1));
        if (oy >= p.ho || ox >= p.wo) continue;
        const size_t opix = (size_t)oy * p.wo + ox;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co < p.cout) p.out[((size_t)b * p.cout + co) * ohw + opix] = acc[i][q][r];
            }
        }
    }
    E4S_PROF_MARK(g_prof_conv, 4);
    E4S_PROF_DRAIN();
    E4S_PROF_MARK(g_prof_conv, 5);
}

template <int KS, int S, int CB, int PB, int WC, int WP, int LOG_TW, int NS = 2>
static int launch2d_sb(Conv2dSbParams& p, hipStream_t st) {
    // two chunks of register prefetch wherever both stages fit in 256 registers without spilling (measured with hipcc 7.2)
    constexpr int PF = ((NS == 2 || NS == 4) && CB * PB <= 2 && (S == 1 || KS == 1)) ? 2 : 1;
    using C = C2SbCfg<KS, S, CB, PB, WC, WP, LOG_TW, NS>;
    p.tiles_x = cdiv(p.wo, C::TW);
    p.tiles_y = cdiv(p.ho, C::TH);
    dim3 grid(p.tiles_x * p.tiles_y, cdiv(p.cout, C::TN), p.bs);
    if (C::LDS_BYTES > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_sb_kernel<KS, S, CB, PB, WC, WP, LOG_TW, PF, NS>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (attr != hipSuccess) return fail((int)attr, "conv2d_sb: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    }
    hipLaunchKernelGGL((conv2d_sb_kernel<KS, S, CB, PB, WC, WP, LOG_TW, PF, NS>), grid, dim3(C::NT), C::LDS_BYTES, st, p);
    return check_launch("conv2d_sb");
}

static int64_t nblocks_sb(const Conv2dSbParams& p, int tn, int th, int tw) {
    return (int64_t)cdiv(p.wo, tw) * cdiv(p.ho, th) * cdiv(p.cout, tn) * p.bs;
}

template <int KS, int S>
static int dispatch2d_sb(Conv2dSbParams& p, hipStream_t st) {
    constexpr int64_t FILL = 512;   // two workgroups per CU before a larger tile is chosen (measured: 192 -> 512 = -6 % on the encoder)
    if (p.wo >= 32) {
        // (a 512-thread 128 co x 256 px tile, <KS,S,4,1,1,8,5>, measured the same as two 64 co x 256 px workgroups per CU: not dispatched)
        if (S == 1 && p.cout > 32 && nblocks_sb(p, 64, 8, 32) >= FILL) return launch2d_sb<KS, S, 2, 2, 1, 4, 5>(p, st);   // 64 co x 256 px
        if (p.cout > 32 && nblocks_sb(p, 64, 4, 32) >= FILL) return launch2d_sb<KS, S, 2, 1, 1, 4, 5>(p, st);             // 64 co x 128 px
        return launch2d_sb<KS, S, 1, 1, 2, 2, 5>(p, st);                                                                  // 64 co x  64 px
    }
    return launch2d_sb<KS, S, 1, 1, 2, 2, 4>(p, st);                                                                      // 64 co x 64 px (16 x 4)
}

// f16 two-term split: two planes and two slabs like split-bf16, so its tiles
template <int KS, int S>
static int dispatch2d_f16x3(Conv2dSbParams& p, hipStream_t st) {
    constexpr int64_t FILL = 512;
    if (p.wo >= 32) {
        if (S == 1 && p.cout > 32 && nblocks_sb(p, 64, 8, 32) >= FILL) return launch2d_sb<KS, S, 2, 2, 1, 4, 5, 4>(p, st);   // 64 co x 256 px
        if (p.cout > 32 && nblocks_sb(p, 64, 4, 32) >= FILL) return launch2d_sb<KS, S, 2, 1, 1, 4, 5, 4>(p, st);             // 64 co x 128 px
        return launch2d_sb<KS, S, 1, 1, 2, 2, 5, 4>(p, st);                                                                  // 64 co x  64 px
    }
    return launch2d_sb<KS, S, 1, 1, 2, 2, 4, 4>(p, st);                                                                      // 64 co x 64 px (16 x 4)
}

// three-way split: the 64 co x 128 px tile is the largest whose three planes + three slabs leave room for two workgroups per CU
template <int KS, int S>
static int dispatch2d_sb3(Conv2dSbParams& p, hipStream_t st) {
    if (p.wo >= 32) {
        if (p.cout > 32 && nblocks_sb(p, 64, 4, 32) >= 256) return launch2d_sb<KS, S, 2, 1, 1, 4, 5, 3>(p, st);           // 64 co x 128 px
        return launch2d_sb<KS, S, 1, 1, 2, 2, 5, 3>(p, st);                                                                // 64 co x  64 px
    }
    return launch2d_sb<KS, S, 1, 1, 2, 2, 4, 3>(p, st);                                                                    // 64 co x 64 px (16 x 4)
}

static int conv2d_sb_common(int nterms, float* out, const float* x0, const float* x1, int cin0, const uint16_t* w0, const uint16_t* w1, const uint16_t* w2,
                            const float* bias, const float* in_mean, const float* in_rstd, const float* prelu_slope, const float* residual, int act,
                            int bs, int cin, int cout, int h, int w, int ks, int stride, int pad, void* stream, float out_scale = 1.f, int* flags = nullptr) {
    E4S_REQUIRE(out && x0 && w0 && w1 && (nterms != 3 || w2), "conv2d_sb: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= 1 && cout >= 1 && h >= 1 && w >= 1, "conv2d_sb: bad size");
    E4S_REQUIRE(stride == 1 || stride == 2, "conv2d_sb: stride %d not supported (1 or 2)", stride);
    E4S_REQUIRE(pad >= 0 && pad <= ks, "conv2d_sb: bad padding");
    E4S_REQUIRE(act >= 0 && act <= 2 && (act != 2 || prelu_slope), "conv2d_sb: bad activation");
    E4S_REQUIRE((in_mean == nullptr) == (in_rstd == nullptr), "conv2d_sb: in_mean and in_rstd go together");
    E4S_REQUIRE(x1 ? (cin0 >= 1 && cin0 < cin) : true, "conv2d_sb: bad channel split");
    E4S_REQUIRE((((uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0, "conv2d_sb: weight slabs must be 16-byte aligned");
    if (bs == 0) return 0;
    Conv2dSbParams p;
    p.out = out; p.x0 = x0; p.x1 = x1; p.bias = bias;
    p.wsl[0] = reinterpret_cast<const uint4*>(w0); p.wsl[1] = reinterpret_cast<const uint4*>(w1); p.wsl[2] = reinterpret_cast<const uint4*>(w2 ? w2 : w1);
    p.in_mean = in_mean; p.in_rstd = in_rstd; p.slope = prelu_slope; p.residual = residual; p.act = act;
    p.bs = bs; p.cin = cin; p.cin0 = x1 ? cin0 : cin; p.cout = cout; p.h = h; p.w = w; p.pad = pad;
    p.ho = (h + 2 * pad - ks) / stride + 1;
    p.wo = (w + 2 * pad - ks) / stride + 1;
    p.out_scale = out_scale;
    p.flags = flags;
    E4S_REQUIRE(p.ho >= 1 && p.wo >= 1, "conv2d_sb: empty output");
    hipStream_t st = (hipStream_t)stream;
    if (nterms == 4) {
        if (ks == 3 && stride == 1) return dispatch2d_f16x3<3, 1>(p, st);
        if (ks == 3 && stride == 2) return dispatch2d_f16x3<3, 2>(p, st);
        if (ks == 1 && stride == 1) return dispatch2d_f16x3<1, 1>(p, st);
        if (ks == 1 && stride == 2) return dispatch2d_f16x3<1, 2>(p, st);
    } else if (nterms == 3) {
        if (ks == 3 && stride == 1) return dispatch2d_sb3<3, 1>(p, st);
        if (ks == 3 && stride == 2) return dispatch2d_sb3<3, 2>(p, st);
        if (ks == 1 && stride == 1) return dispatch2d_sb3<1, 1>(p, st);
        if (ks == 1 && stride == 2) return dispatch2d_sb3<1, 2>(p, st);
    } else {
        if (ks == 3 && stride == 1) return dispatch2d_sb<3, 1>(p, st);
        if (ks == 3 && stride == 2) return dispatch2d_sb<3, 2>(p, st);
        if (ks == 1 && stride == 1) return dispatch2d_sb<1, 1>(p, st);
        if (ks == 1 && stride == 2) return dispatch2d_sb<1, 2>(p, st);
    }
    return fail(E4S_ERR_ARG, "conv2d_sb: kernel %dx%d stride %d not supported (3x3 / 1x1, stride 1 / 2)", ks, ks, stride);
}

extern "C" int e4s_conv2d_sb(float* out, const float* x0, const float* x1, int cin0, const uint16_t* whi, const uint16_t* wlo, const float* bias,
                             const float* in_mean, const float* in_rstd, const float* prelu_slope, const float* residual, int act, int bs,
                             int cin, int cout, int h, int w, int ks, int stride, int pad, void* stream) {
    return conv2d_sb_common(2, out, x0, x1, cin0, whi, wlo, nullptr, bias, in_mean, in_rstd, prelu_slope, residual, act, bs, cin, cout, h, w, ks, stride,
                            pad, stream);
}

extern "C" int e4s_conv2d_sb3(float* out, const float* x0, const float* x1, int cin0, const uint16_t* w0, const uint16_t* w1, const uint16_t* w2,
                              const float* bias, const float* in_mean, const float* in_rstd, const float* prelu_slope, const float* residual, int act,
                              int bs, int cin, int cout, int h, int w, int ks, int stride, int pad, void* stream) {
    return conv2d_sb_common(3, out, x0, x1, cin0, w0, w1, w2, bias, in_mean, in_rstd, prelu_slope, residual, act, bs, cin, cout, h, w, ks, stride, pad,
                            stream);
}

// The f16 two-term split (NS = 4 above): fp32-class error at three f16 MFMAs per 16-deep step.  w1 / w2: f16 (as uint16) slabs of weight * 2^wscale_log2 in
// e4s_conv_prep_weights_sb's layout (choose wscale_log2 so that the largest folded weight lands near 2^10); e4s_conv2d_f16x3 takes the same power back out.
extern "C" int e4s_conv_prep_weights_f16x3(uint16_t* w1, uint16_t* w2, float* bias_out, const float* weight, const float* bn_gamma, const float* bn_beta,
                                           const float* bn_mean, const float* bn_var, float bn_eps, const float* conv_bias, int cout, int cin, int kh, int kw,
                                           int wscale_log2, void* stream) {
    E4S_REQUIRE(w1 && w2 && weight, "conv_prep_weights_f16x3: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= 1 && kh >= 1 && kw >= 1 && wscale_log2 >= -40 && wscale_log2 <= 40, "conv_prep_weights_f16x3: bad arguments");
    const bool bn = bn_var != nullptr;
    E4S_REQUIRE(!bn || (bn_gamma && bn_beta && bn_mean && bias_out), "conv_prep_weights_f16x3: BatchNorm fold needs gamma, beta, mean, var and bias_out");
    E4S_REQUIRE(!conv_bias || bias_out, "conv_prep_weights_f16x3: conv bias needs bias_out");
    const int64_t total = (int64_t)cdiv(cin, CKS2) * kh * kw * 2 * cout * 8;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(conv_prep_sb_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w1, w2, (uint16_t*)nullptr, bias_out, weight, bn_gamma, bn_beta,
                       bn_mean, bn_var, bn_eps, conv_bias, cout, cin, kh * kw, 1, ldexpf(1.f, wscale_log2));
    return check_launch("conv_prep_weights_f16x3");
}

extern "C" int e4s_conv2d_f16x3(float* out, const float* x0, const float* x1, int cin0, const uint16_t* w1, const uint16_t* w2, const float* bias,
                                const float* in_mean, const float* in_rstd, const float* prelu_slope, const float* residual, int act, int bs, int cin,
                                int cout, int h, int w, int ks, int stride, int pad, int wscale_log2, int* flags, void* stream) {
    return conv2d_sb_common(4, out, x0, x1, cin0, w1, w2, nullptr, bias, in_mean, in_rstd, prelu_slope, residual, act, bs, cin, cout, h, w, ks, stride, pad,
                            stream, ldexpf(1.f, -wscale_log2), flags);
}
