// f1: gradient kernels of the one-pass region-modulated convolution
//     z[b,o,p] = Σ_{i,k} W[o,i,k] · s[b,c(p),i] · x[b,i,p+k-pad]          y[b,o,p] = d[b,c(p),o] · z[b,o,p]
//     out = leaky_relu(y + nw·noise + bias) · √2                           (StyledConv: noise injection :335, FusedLeakyReLU :421)
// which is Σ_c modconv(x, style_c) ⊙ [label == c] of StyledConv.forward / ToRGB.forward (models/stylegan2/model.py:389-398, 447-454)
// as PTI tuning back-propagates through it (training/video_swap_ft_coach.py:242-299).  The two GEMMs of the backward are plain
// fp32 library GEMMs (the host side calls rocBLAS through torch.matmul); everything around them — what makes a stock-PyTorch
// backward 10 ms per layer (unfold, gather, fold, segmented sums over 300 MB intermediates) — is the three HBM-bound kernels here:
//     cols[g,b,(i,k),q] = s[b,c_g(q),i] · x[b,i,q+k-pad]                                (e4s_mconv_unfold;  z = W_g · cols, dW_g = gz · colsᵀ)
//     gz[g,b,o,q] = gy · act'(out) · d[b,c,o],   q[b,c,o] = Σ_{p∈c} gy·act'·y,  Σ gy·act',  Σ gy·act'·noise      (e4s_mconv_scale)
//     with U_g = W_gᵀ · gz_g:   dx[b,i,t] = Σ_g Σ_k s[b,c_g(t-k+pad),i] · U[g,b,(i,k),t-k+pad]        (e4s_mconv_fold, one pass over U)
//                               ds[b,c,i] = Σ_g Σ_{q: c_g(q)=c} Σ_k U[g,b,(i,k),q] · x[b,i,q+k-pad]
// Up-sampling layers (up = 2) are the four output parities g = (a, b) of the composed form: output pixel (2qy+a, 2qx+b) is a 3x3
// correlation of x around q with the composed weight W_g, under the label of THAT output pixel: c_g(q) = label[2qy+a][2qx+b].
// A label >= nreg belongs to no region: zero output, zero gradient.  All fp32, every sum in a fixed order (no atomics).
#include "common.h"

using namespace e4s;

namespace {

constexpr int NT = 256;

template <int KS>
__global__ __launch_bounds__(NT) void mconv_unfold_kernel(float* __restrict__ cols, const float* __restrict__ x, const float* __restrict__ s,
                                                          const uint8_t* __restrict__ lab, int cin, int h, int w, int nreg, int up) {
    constexpr int KK = KS * KS, PAD = KS / 2;
    const int bs = gridDim.z;
    const int b = blockIdx.z, i = blockIdx.y;
    const int P = h * w;
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= P) return;
    const int py = p / w, px = p - py * w;
    const float* xp = x + ((size_t)b * cin + i) * P;
    float xv[KK];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int yy = py + ky - PAD, xx = px + kx - PAD;
            xv[ky * KS + kx] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? xp[yy * w + xx] : 0.f;
        }
    const int lw = up * w;
    for (int g = 0; g < up * up; ++g) {
        const int c = lab ? lab[((size_t)b * up * h + up * py + g / up) * lw + up * px + g % up] : 0;
        const float sv = c < nreg ? s[((size_t)b * nreg + c) * cin + i] : 0.f;
        float* cp = cols + (((size_t)g * bs + b) * cin + i) * KK * P + p;
#pragma unroll
        for (int k = 0; k < KK; ++k) cp[(size_t)k * P] = sv * xv[k];
    }
}

// one workgroup per (pixel chunk, o, b) of the (up*h) x (up*w) output plane; the sums come out per chunk (the host adds them up)
__global__ __launch_bounds__(NT) void mconv_scale_kernel(float* __restrict__ gz, float* __restrict__ q, float* __restrict__ dbias, float* __restrict__ dnw,
                                                         const float* __restrict__ gy, const float* __restrict__ out, const float* __restrict__ d,
                                                         const uint8_t* __restrict__ lab, const float* __restrict__ noise, int noise_bs,
                                                         const float* __restrict__ nw, const float* __restrict__ bias, int act, int cout, int h, int w,
                                                         int nreg, int up, int chunk_px) {
    __shared__ float dtab[E4S_MAX_REGIONS];
    __shared__ float red[NT / 64][E4S_MAX_REGIONS + 2];
    const int bs = gridDim.z;
    const int b = blockIdx.z, o = blockIdx.y, chunk = blockIdx.x;
    if (threadIdx.x < E4S_MAX_REGIONS)
        dtab[threadIdx.x] = threadIdx.x < nreg ? (d ? d[((size_t)b * nreg + threadIdx.x) * cout + o] : 1.f) : 0.f;
    __syncthreads();
    const int W2 = up * w, P2 = up * h * W2, P = h * w;
    const size_t base = ((size_t)b * cout + o) * P2;
    const uint8_t* lp = lab ? lab + (size_t)b * P2 : nullptr;       // no label map: every pixel is region 0
    const float* np = noise ? noise + (size_t)(noise_bs > 1 ? b : 0) * P2 : nullptr;
    const float nwv = noise ? nw[0] : 0.f, bv = bias ? bias[o] : 0.f;
    const float SQ2 = 1.41421356237309515f;
    float acc[E4S_MAX_REGIONS + 2];
#pragma unroll
    for (int c = 0; c < E4S_MAX_REGIONS + 2; ++c) acc[c] = 0.f;
    const int p_end = min(P2, (chunk + 1) * chunk_px);
    for (int p = chunk * chunk_px + threadIdx.x; p < p_end; p += NT) {
        const int c = lp ? lp[p] : 0;
        float g = gy[base + p];
        float yv = out ? out[base + p] : 0.f;
        if (act) {                                   // out = leaky_relu(pre, 0.2) * sqrt(2): sign(out) = sign(pre)
            const float slope = yv > 0.f ? SQ2 : 0.2f * SQ2;
            g *= slope;
            yv /= slope;
        }
        const float nz = np ? np[p] : 0.f;
        yv -= bv + nwv * nz;                         // y = d * z
        size_t dst = base + p;                       // gz is [up*up, bs, cout, h*w]: parity-planar for the up-sampling layers
        if (up == 2) {
            const int Y = p / W2, X = p - Y * W2;
            dst = (((size_t)((Y & 1) * 2 + (X & 1)) * bs + b) * cout + o) * P + (Y >> 1) * w + (X >> 1);
        }
        gz[dst] = c < nreg ? g * dtab[c] : 0.f;
        acc[E4S_MAX_REGIONS] += g;
        acc[E4S_MAX_REGIONS + 1] += g * nz;
        if (q) {
            const float t = g * yv;
#pragma unroll
            for (int r = 0; r < E4S_MAX_REGIONS; ++r) acc[r] += c == r ? t : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < E4S_MAX_REGIONS + 2; ++r) {
        const float v = wave_sum(acc[r]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][r] = v;
    }
    __syncthreads();
    if (threadIdx.x < E4S_MAX_REGIONS + 2) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv) v += red[wv][threadIdx.x];
        const size_t cb = (size_t)chunk * bs + b;
        if (threadIdx.x < nreg && q) q[(cb * nreg + threadIdx.x) * cout + o] = v;
        if (threadIdx.x == E4S_MAX_REGIONS && dbias) dbias[cb * cout + o] = v;
        if (threadIdx.x == E4S_MAX_REGIONS + 1 && dnw) dnw[cb * cout + o] = v;
    }
}

// one workgroup per (pixel chunk, i, b): reads the rows of U for its chunk once (the shifted reads of the dgrad hit the same lines)
template <int KS>
__global__ __launch_bounds__(NT) void mconv_fold_kernel(float* __restrict__ dx, float* __restrict__ ds_part, const float* __restrict__ U,
                                                        const float* __restrict__ x, const float* __restrict__ s, const uint8_t* __restrict__ lab,
                                                        int cin, int h, int w, int nreg, int up, int chunk_px) {
    constexpr int KK = KS * KS, PAD = KS / 2;
    __shared__ float stab[E4S_MAX_REGIONS + 1];
    __shared__ float red[NT / 64][E4S_MAX_REGIONS];
    const int bs = gridDim.z;
    const int b = blockIdx.z, i = blockIdx.y, chunk = blockIdx.x;
    const int P = h * w, lw = up * w;
    if (threadIdx.x <= E4S_MAX_REGIONS) stab[threadIdx.x] = threadIdx.x < nreg ? s[((size_t)b * nreg + threadIdx.x) * cin + i] : 0.f;
    __syncthreads();
    const uint8_t* lp = lab ? lab + (size_t)b * up * h * lw : nullptr;      // no label map: every pixel is region 0
    const float* xp = x + ((size_t)b * cin + i) * P;
    float acc[E4S_MAX_REGIONS];
#pragma unroll
    for (int c = 0; c < E4S_MAX_REGIONS; ++c) acc[c] = 0.f;
    const int p_end = min(P, (chunk + 1) * chunk_px);
    for (int p = chunk * chunk_px + threadIdx.x; p < p_end; p += NT) {
        const int py = p / w, px = p - py * w;
        float xv[KK];
        if (ds_part) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int yy = py + ky - PAD, xx = px + kx - PAD;
                    xv[ky * KS + kx] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? xp[yy * w + xx] : 0.f;
                }
        }
        float gsum = 0.f;
        for (int g = 0; g < up * up; ++g) {
            const int ga = g / up, gb = g % up;
            const float* ug = U + (((size_t)g * bs + b) * cin + i) * KK * P;
            float t = 0.f;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int k = ky * KS + kx;
                    if (ds_part) t += ug[(size_t)k * P + p] * xv[k];       // this pixel as the OUTPUT position of tap k
                    // dx: this pixel as the INPUT of tap k, i.e. of output position (py - ky + pad, px - kx + pad)
                    const int oy = py - ky + PAD, ox = px - kx + PAD;
                    if (dx && oy >= 0 && oy < h && ox >= 0 && ox < w) {
                        const int c = lp ? lp[(up * oy + ga) * lw + up * ox + gb] : 0;
                        gsum += stab[c < nreg ? c : E4S_MAX_REGIONS] * ug[(size_t)k * P + oy * w + ox];
                    }
                }
            if (ds_part) {
                const int c = lp ? lp[(up * py + ga) * lw + up * px + gb] : 0;
#pragma unroll
                for (int r = 0; r < E4S_MAX_REGIONS; ++r) acc[r] += c == r ? t : 0.f;
            }
        }
        if (dx) dx[((size_t)b * cin + i) * P + p] = gsum;
    }
    if (!ds_part) return;
#pragma unroll
    for (int r = 0; r < E4S_MAX_REGIONS; ++r) {
        const float v = wave_sum(acc[r]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][r] = v;
    }
    __syncthreads();
    if (threadIdx.x < nreg) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv) v += red[wv][threadIdx.x];
        ds_part[(((size_t)chunk * bs + b) * nreg + threadIdx.x) * cin + i] = v;
    }
}

// cols[b, c*KK + k, q] = x[b, c, stride*qy + ky - pad, stride*qx + kx - pad]  (0 outside): the B operand of a plain convolution's weight
// gradient as a library GEMM (the single-region layers: dW = g' colsᵀ, or — transposed conv — dW = x cols(gT)ᵀ with stride 2)
template <int KS>
__global__ __launch_bounds__(NT) void unfold2d_kernel(float* __restrict__ cols, const float* __restrict__ x, int C, int hi, int wi, int ho, int wo,
                                                      int stride, int pad) {
    constexpr int KK = KS * KS;
    const int b = blockIdx.z, c = blockIdx.y;
    const int P = ho * wo;
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= P) return;
    const int py = p / wo, px = p - py * wo;
    const float* xp = x + ((size_t)b * C + c) * hi * wi;
    float* cp = cols + ((size_t)b * C + c) * KK * P + p;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int yy = stride * py + ky - pad, xx = stride * px + kx - pad;
            cp[(size_t)(ky * KS + kx) * P] = (yy >= 0 && yy < hi && xx >= 0 && xx < wi) ? xp[yy * wi + xx] : 0.f;
        }
}

// The 3x3 fold with four pixels of a row per thread (w % 4 == 0): 18 aligned 16-byte loads of U + 6 dword loads per group and four pixels
// instead of 72 dword loads — the scalar kernel above is bound by the number of vector-memory instructions, not by bytes.
__global__ __launch_bounds__(NT) void mconv_fold4_kernel(float* __restrict__ dx, float* __restrict__ ds_part, const float* __restrict__ U,
                                                         const float* __restrict__ x, const float* __restrict__ s, const uint8_t* __restrict__ lab,
                                                         int cin, int h, int w, int nreg, int up, int chunk_px) {
    __shared__ float stab[E4S_MAX_REGIONS + 1];
    __shared__ float red[NT / 64][E4S_MAX_REGIONS];
    const int bs = gridDim.z;
    const int b = blockIdx.z, i = blockIdx.y, chunk = blockIdx.x;
    const int P = h * w, lw = up * w;
    if (threadIdx.x <= E4S_MAX_REGIONS) stab[threadIdx.x] = threadIdx.x < nreg ? s[((size_t)b * nreg + threadIdx.x) * cin + i] : 0.f;
    __syncthreads();
    const uint8_t* lp = lab ? lab + (size_t)b * up * h * lw : nullptr;
    const float* xp = x + ((size_t)b * cin + i) * P;
    float acc[E4S_MAX_REGIONS];
#pragma unroll
    for (int c = 0; c < E4S_MAX_REGIONS; ++c) acc[c] = 0.f;
    const int p_end = min(P, (chunk + 1) * chunk_px);
    for (int p = chunk * chunk_px + 4 * threadIdx.x; p < p_end; p += 4 * NT) {
        const int py = p / w, px = p - py * w;
        // x around the four pixels: rows py-1..py+1, columns px-1..px+4
        float xr[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = py + r - 1;
            const bool in = ds_part && yy >= 0 && yy < h;
            const float4 v = in ? *reinterpret_cast<const float4*>(xp + yy * w + px) : make_float4(0.f, 0.f, 0.f, 0.f);
            xr[r][0] = (in && px >= 1) ? xp[yy * w + px - 1] : 0.f;
            xr[r][1] = v.x; xr[r][2] = v.y; xr[r][3] = v.z; xr[r][4] = v.w;
            xr[r][5] = (in && px + 4 < w) ? xp[yy * w + px + 4] : 0.f;
        }
        float gsum[4] = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < up * up; ++g) {
            const int ga = g / up, gb = g % up;
            const float* ug = U + (((size_t)g * bs + b) * cin + i) * 9 * P;
            // region and modulation of the 3 x 6 output positions around the four pixels (0 outside the image / no region)
            int lr[6];
            float sv[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int yy = py + r - 1, xx = px + c - 1;
                    int lc = E4S_MAX_REGIONS;
                    if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                        lc = lp ? lp[(up * yy + ga) * lw + up * xx + gb] : 0;
                        if (lc >= nreg) lc = E4S_MAX_REGIONS;
                    }
                    sv[r][c] = stab[lc];
                    if (r == 1) lr[c] = lc;
                }
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float* uk = ug + (size_t)(ky * 3 + kx) * P;
                    if (ds_part) {                       // these four pixels as OUTPUT positions of tap (ky, kx)
                        const float4 v = *reinterpret_cast<const float4*>(uk + p);
                        t[0] += v.x * xr[ky][kx]; t[1] += v.y * xr[ky][kx + 1]; t[2] += v.z * xr[ky][kx + 2]; t[3] += v.w * xr[ky][kx + 3];
                    }
                    if (dx) {                            // as INPUT positions: output position = (py - ky + 1, px + j - kx + 1)
                        const int oy = py - ky + 1, rr = 2 - ky, off = 1 - kx;
                        if (oy >= 0 && oy < h) {
                            const float4 v = *reinterpret_cast<const float4*>(uk + oy * w + px);
                            float u0, u1, u2, u3;
                            if (off == 0) { u0 = v.x; u1 = v.y; u2 = v.z; u3 = v.w; }
                            else if (off == 1) { u0 = v.y; u1 = v.z; u2 = v.w; u3 = px + 4 < w ? uk[oy * w + px + 4] : 0.f; }
                            else { u0 = px >= 1 ? uk[oy * w + px - 1] : 0.f; u1 = v.x; u2 = v.y; u3 = v.z; }
                            gsum[0] += sv[rr][off + 1] * u0; gsum[1] += sv[rr][off + 2] * u1;
                            gsum[2] += sv[rr][off + 3] * u2; gsum[3] += sv[rr][off + 4] * u3;
                        }
                    }
                }
            if (ds_part) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < E4S_MAX_REGIONS; ++r) acc[r] += lr[j + 1] == r ? t[j] : 0.f;
            }
        }
        if (dx) *reinterpret_cast<float4*>(dx + ((size_t)b * cin + i) * P + p) = make_float4(gsum[0], gsum[1], gsum[2], gsum[3]);
    }
    if (!ds_part) return;
#pragma unroll
    for (int r = 0; r < E4S_MAX_REGIONS; ++r) {
        const float v = wave_sum(acc[r]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][r] = v;
    }
    __syncthreads();
    if (threadIdx.x < nreg) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv) v += red[wv][threadIdx.x];
        ds_part[(((size_t)chunk * bs + b) * nreg + threadIdx.x) * cin + i] = v;
    }
}

// ---- gradient of a layer's style tables (model.py:276-281 in the one-pass form), three launches instead of ~25 small library ops:
//     s = styles · (mod_w·ms)ᵀ + mod_b·lr        ws = c · weight        d = rsqrt(s² · wsq + 1e-8),   wsq[i,o] = Σ_k ws[o,i,k]²
// given gs = dL/ds, gd = dL/dd, gws = dL/dws (each optional):
//     t = -½ · gd · d³                      gs' = gs + 2 s ⊙ (t · wsqᵀ)             gwsq = tᵀ · s²
//     g_weight = c · (gws + 2 ws ⊙ gwsq)    g_styles = ms · gs' · mod_w             g_mod_w = ms · gs'ᵀ · styles        g_mod_b = lr · Σ_B gs'
// B = bs*nreg rows.  Tables are small (B <= 96 rows, <= 512 channels): one thread per output element, loops over the short dimension.
__global__ __launch_bounds__(NT) void tables_bwd_s_kernel(float* __restrict__ gs_tot, float* __restrict__ t_out, const float* __restrict__ gs,
                                                          const float* __restrict__ gd, const float* __restrict__ s, const float* __restrict__ d,
                                                          const float* __restrict__ wsq, int cin, int cout) {
    extern __shared__ float tl[];                  // t[b, :]
    const int b = blockIdx.y;
    for (int o = threadIdx.x; o < cout; o += NT) {
        const float dv = d[(size_t)b * cout + o];
        const float tv = -0.5f * gd[(size_t)b * cout + o] * dv * dv * dv;
        tl[o] = tv;
        if (blockIdx.x == 0) t_out[(size_t)b * cout + o] = tv;
    }
    __syncthreads();
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= cin) return;
    float acc = 0.f;
    for (int o = 0; o < cout; ++o) acc += tl[o] * wsq[(size_t)i * cout + o];      // wsq is [cin, cout], as e4s_modconv_prep_weights writes it
    const size_t e = (size_t)b * cin + i;
    gs_tot[e] = (gs ? gs[e] : 0.f) + 2.f * s[e] * acc;
}

__global__ __launch_bounds__(NT) void tables_bwd_w_kernel(float* __restrict__ g_weight, const float* __restrict__ gws, const float* __restrict__ t,
                                                          const float* __restrict__ s, const float* __restrict__ weight, float c, int B, int cin,
                                                          int cout, int kk) {
    const int e = blockIdx.x * NT + threadIdx.x;
    if (e >= cout * cin) return;
    const int o = e / cin, i = e - o * cin;
    float gwsq = 0.f;
    if (t)
        for (int b = 0; b < B; ++b) {
            const float sv = s[(size_t)b * cin + i];
            gwsq += t[(size_t)b * cout + o] * sv * sv;
        }
    for (int k = 0; k < kk; ++k) {
        const size_t a = (size_t)e * kk + k;
        g_weight[a] = c * ((gws ? gws[a] : 0.f) + 2.f * c * weight[a] * gwsq);
    }
}

__global__ __launch_bounds__(NT) void tables_bwd_mod_kernel(float* __restrict__ g_styles, float* __restrict__ g_mod_w, float* __restrict__ g_mod_b,
                                                            const float* __restrict__ gs_tot, const float* __restrict__ styles,
                                                            const float* __restrict__ mod_w, float ms, float lr, int B, int sd, int cin) {
    // 64 outputs (j) per workgroup, the reduction split over its four waves (a serial 512-long loop per thread is latency-bound: 150 us)
    __shared__ float part[4][64];
    __shared__ float pb[4];
    const int jj = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jj;
    const bool rows = (int)blockIdx.y < B;         // g_styles[b, j] = ms * Σ_i gs'[b,i] * mod_w[i,j]
    const int b = blockIdx.y, i = blockIdx.y - B;  // else g_mod_w[i, j] = ms * Σ_b gs'[b,i] * styles[b,j];  g_mod_b[i] = lr * Σ_b gs'[b,i]
    float acc = 0.f, accb = 0.f;
    if (j < sd) {
        if (rows) {
            const int n = (cin + 3) / 4, i0 = sl * n, i1 = min(cin, i0 + n);
#pragma unroll 8
            for (int q = i0; q < i1; ++q) acc += gs_tot[(size_t)b * cin + q] * mod_w[(size_t)q * sd + j];
        } else {
            for (int q = sl; q < B; q += 4) {
                const float g = gs_tot[(size_t)q * cin + i];
                acc += g * styles[(size_t)q * sd + j];
                accb += g;
            }
        }
    }
    part[sl][jj] = acc;
    if (jj == 0) pb[sl] = accb;                    // the same for every lane of the wave (does not depend on j)
    __syncthreads();
    if (sl == 0 && j < sd) {
        const float v = ms * (part[0][jj] + part[1][jj] + part[2][jj] + part[3][jj]);
        if (rows) g_styles[(size_t)b * sd + j] = v;
        else g_mod_w[(size_t)i * sd + j] = v;
    }
    if (!rows && blockIdx.x == 0 && threadIdx.x == 0) g_mod_b[i] = lr * (pb[0] + pb[1] + pb[2] + pb[3]);
}

int bad_shape(int bs, int c, int h, int w, int ks, int nreg, int up) {
    return !(bs >= 0 && bs <= 65535 && c >= 1 && c <= 65535 && h >= 1 && w >= 1 && (int64_t)h * w * up * up < ((int64_t)1 << 24) &&
             (ks == 1 || ks == 3) && nreg >= 1 && nreg <= E4S_MAX_REGIONS && (up == 1 || up == 2));
}

}  // namespace

extern "C" int e4s_unfold2d(float* cols, const float* x, int bs, int C, int hi, int wi, int ho, int wo, int ks, int stride, int pad, void* stream) {
    E4S_REQUIRE(cols && x, "unfold2d: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && C >= 1 && C <= 65535 && hi >= 1 && wi >= 1 && ho >= 1 && wo >= 1 && (int64_t)ho * wo < ((int64_t)1 << 24) &&
                    (int64_t)hi * wi < ((int64_t)1 << 24) && (ks == 1 || ks == 3) && stride >= 1 && pad >= 0,
                "unfold2d: bad size (3x3 / 1x1)");
    if (bs == 0) return 0;
    const dim3 grid(cdiv(ho * wo, NT), C, bs);
    if (ks == 3) hipLaunchKernelGGL(unfold2d_kernel<3>, grid, dim3(NT), 0, (hipStream_t)stream, cols, x, C, hi, wi, ho, wo, stride, pad);
    else hipLaunchKernelGGL(unfold2d_kernel<1>, grid, dim3(NT), 0, (hipStream_t)stream, cols, x, C, hi, wi, ho, wo, stride, pad);
    return check_launch("unfold2d");
}

extern "C" int e4s_mconv_unfold(float* cols, const float* x, const float* s, const uint8_t* labels, int bs, int cin, int h, int w, int ks, int nreg,
                                int up, void* stream) {
    E4S_REQUIRE(cols && x && s, "mconv_unfold: null tensor");
    E4S_REQUIRE(!bad_shape(bs, cin, h, w, ks, nreg, up), "mconv_unfold: bad size (3x3 / 1x1, nreg 1..%d, up 1 / 2)", E4S_MAX_REGIONS);
    if (bs == 0) return 0;
    const dim3 grid(cdiv(h * w, NT), cin, bs);
    if (ks == 3) hipLaunchKernelGGL(mconv_unfold_kernel<3>, grid, dim3(NT), 0, (hipStream_t)stream, cols, x, s, labels, cin, h, w, nreg, up);
    else hipLaunchKernelGGL(mconv_unfold_kernel<1>, grid, dim3(NT), 0, (hipStream_t)stream, cols, x, s, labels, cin, h, w, nreg, up);
    return check_launch("mconv_unfold");
}

extern "C" int e4s_mconv_scale(float* gz, float* q, float* dbias, float* dnw, const float* gy, const float* out, const float* d, const uint8_t* labels,
                               const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act, int bs, int cout, int h,
                               int w, int nreg, int up, int chunk_px, void* stream) {
    E4S_REQUIRE(gz && gy, "mconv_scale: null tensor");
    E4S_REQUIRE(chunk_px >= NT && chunk_px % NT == 0, "mconv_scale: chunk_px must be a multiple of %d", NT);
    E4S_REQUIRE((q == nullptr && !act) || out, "mconv_scale: the per-region sums and the activation gradient need the forward output");
    E4S_REQUIRE((noise == nullptr) || (noise_weight && (noise_bs == 1 || noise_bs == bs)), "mconv_scale: noise needs its weight and batch 1 or bs");
    E4S_REQUIRE(act == 0 || act == 1, "mconv_scale: act is 0 or 1");
    E4S_REQUIRE(!bad_shape(bs, cout, h, w, 1, nreg, up), "mconv_scale: bad size");
    if (bs == 0) return 0;
    hipLaunchKernelGGL(mconv_scale_kernel, dim3(cdiv(up * h * up * w, chunk_px), cout, bs), dim3(NT), 0, (hipStream_t)stream, gz, q, dbias, dnw, gy, out,
                       d, labels, noise, noise_bs, noise_weight, act_bias, act, cout, h, w, nreg, up, chunk_px);
    return check_launch("mconv_scale");
}

extern "C" int e4s_mconv_fold(float* dx, float* ds_part, const float* U, const float* x, const float* s, const uint8_t* labels, int bs, int cin, int h,
                              int w, int ks, int nreg, int up, int chunk_px, void* stream) {
    E4S_REQUIRE((dx || ds_part) && U && s, "mconv_fold: null tensor");
    E4S_REQUIRE(!ds_part || x, "mconv_fold: the style gradient needs x");
    E4S_REQUIRE(chunk_px >= NT && chunk_px % NT == 0, "mconv_fold: chunk_px must be a multiple of %d", NT);
    E4S_REQUIRE(!bad_shape(bs, cin, h, w, ks, nreg, up), "mconv_fold: bad size (3x3 / 1x1, nreg 1..%d, up 1 / 2)", E4S_MAX_REGIONS);
    if (bs == 0) return 0;
    const dim3 grid(cdiv(h * w, chunk_px), cin, bs);
    const bool vec4 = ks == 3 && w % 4 == 0 && chunk_px % (4 * NT) == 0 && ((((uintptr_t)U | (uintptr_t)x | (uintptr_t)dx) & 15) == 0);
    if (vec4)
        hipLaunchKernelGGL(mconv_fold4_kernel, grid, dim3(NT), 0, (hipStream_t)stream, dx, ds_part, U, x, s, labels, cin, h, w, nreg, up, chunk_px);
    else if (ks == 3)
        hipLaunchKernelGGL(mconv_fold_kernel<3>, grid, dim3(NT), 0, (hipStream_t)stream, dx, ds_part, U, x, s, labels, cin, h, w, nreg, up, chunk_px);
    else
        hipLaunchKernelGGL(mconv_fold_kernel<1>, grid, dim3(NT), 0, (hipStream_t)stream, dx, ds_part, U, x, s, labels, cin, h, w, nreg, up, chunk_px);
    return check_launch("mconv_fold");
}

extern "C" int e4s_style_tables_bwd(float* g_styles, float* g_mod_w, float* g_mod_b, float* g_weight, float* scratch, const float* gs, const float* gd,
                                    const float* gws, const float* styles, const float* mod_w, const float* s, const float* d, const float* weight,
                                    const float* wsq, float weight_scale, float mod_scale, float mod_lr, int rows, int sdim, int cin, int cout, int kk,
                                    void* stream) {
    E4S_REQUIRE(scratch && styles && mod_w && s && weight, "style_tables_bwd: null tensor");
    E4S_REQUIRE((gd == nullptr) || (d && wsq), "style_tables_bwd: the demodulation gradient needs d and wsq");
    E4S_REQUIRE(rows >= 1 && sdim >= 1 && cin >= 1 && rows + cin <= 65535 && cout >= 1 && cout <= 8192 && kk >= 1,
                "style_tables_bwd: bad size");
    hipStream_t st = (hipStream_t)stream;
    float* t = scratch;                            // [rows, cout]
    float* gs_tot = scratch + (size_t)rows * cout; // [rows, cin]
    const float* gsp = gs;
    if (gd) {
        hipLaunchKernelGGL(tables_bwd_s_kernel, dim3(cdiv(cin, NT), rows), dim3(NT), sizeof(float) * cout, st, gs_tot, t, gs, gd, s, d, wsq, cin, cout);
        gsp = gs_tot;
    }
    if (g_weight && (gd || gws))
        hipLaunchKernelGGL(tables_bwd_w_kernel, dim3(cdiv(cout * cin, NT)), dim3(NT), 0, st, g_weight, gws, gd ? t : nullptr, s, weight, weight_scale, rows,
                           cin, cout, kk);
    if (gsp && g_styles && g_mod_w && g_mod_b)
        hipLaunchKernelGGL(tables_bwd_mod_kernel, dim3(cdiv(sdim, 64), rows + cin), dim3(NT), 0, st, g_styles, g_mod_w, g_mod_b, gsp, styles, mod_w,
                           mod_scale, mod_lr, rows, sdim, cin);
    return check_launch("style_tables_bwd");
}
