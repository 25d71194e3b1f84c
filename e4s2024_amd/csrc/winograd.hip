// Winograd F(2x2, 3x3) around the batched split-bf16 GEMM (gemm_sb.hip) for the regional-style encoder's stride-1 3x3 convolutions (a8:
// models/encoders/helpers.py:122-144, the 512 -> 512 @32^2 units of models/encoders/psp_encoders.py's IR-SE-50 body — 27 launches per batch that
// run at the sustained MFMA rate in their direct form, so only fewer multiplications make them faster):
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 4 x 4 input patch d -> 2 x 2 outputs, summed over input channels
//
// which is 16 independent GEMMs  M_k [cout x T] = U_k [cout x cin] * V_k [cin x T]  over the T = bs * H/2 * W/2 tiles: 2.25x fewer MACs than the
// direct form.  Three small kernels here; the GEMMs are e4s_gemm_sb(batch = 16).
//     e4s_wino_weight : U[k][co][ci] = (G g G^T)[k]                     (once per weight)
//     e4s_wino_input  : V[k][ci][t]  = (B^T d B)[k], d = the (optionally instance-normalised) input patch with the conv's zero padding
//     e4s_wino_output : y[b][co][2ty+i][2tx+j] = act( (A^T m A)[i][j] ),  m = M[.][co][t];  act = PReLU(co) or none
// B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
#include "sb_common.h"

using namespace e4s;

namespace {

__global__ __launch_bounds__(256) void wino_weight_kernel(float* __restrict__ U, const float* __restrict__ w, int cout, int cin) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= cout * cin) return;
    const float* g = w + (size_t)e * 9;
    float gg[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        gg[0][j] = g0;
        gg[1][j] = 0.5f * (g0 + g1 + g2);
        gg[2][j] = 0.5f * (g0 - g1 + g2);
        gg[3][j] = g2;
    }
    const size_t plane = (size_t)cout * cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = gg[i][0], b = gg[i][1], c = gg[i][2];
        U[(size_t)(4 * i + 0) * plane + e] = a;
        U[(size_t)(4 * i + 1) * plane + e] = 0.5f * (a + b + c);
        U[(size_t)(4 * i + 2) * plane + e] = 0.5f * (a - b + c);
        U[(size_t)(4 * i + 3) * plane + e] = c;
    }
}

// one thread = one tile of one channel; consecutive threads = consecutive tiles of a row (coalesced plane writes, 8-byte row reads)
__global__ __launch_bounds__(256) void wino_input_kernel(float* __restrict__ V, const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, int bs, int C, int H, int W) {
    const int th = H >> 1, tw = W >> 1;
    const int T = bs * th * tw;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= T) return;
    const int tx = t % tw, ty = (t / tw) % th, b = t / (tw * th);
    const float* xp = x + ((size_t)b * C + c) * H * W;
    const float mu = mean ? mean[(size_t)b * C + c] : 0.f, rs = rstd ? rstd[(size_t)b * C + c] : 1.f;
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 2 * ty - 1 + i;
        const bool row = y >= 0 && y < H;
        const float* r = xp + (size_t)(row ? y : 0) * W + 2 * tx;
        const float2 mid = *reinterpret_cast<const float2*>(r);
        const float lft = tx > 0 ? r[-1] : 0.f, rgt = 2 * tx + 2 < W ? r[2] : 0.f;
        // the convolution pads the NORMALISED map with zeros
        d[i][0] = (row && tx > 0) ? (lft - mu) * rs : 0.f;
        d[i][1] = row ? (mid.x - mu) * rs : 0.f;
        d[i][2] = row ? (mid.y - mu) * rs : 0.f;
        d[i][3] = (row && 2 * tx + 2 < W) ? (rgt - mu) * rs : 0.f;
    }
    float e[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // B^T d
        e[0][j] = d[0][j] - d[2][j];
        e[1][j] = d[1][j] + d[2][j];
        e[2][j] = d[2][j] - d[1][j];
        e[3][j] = d[1][j] - d[3][j];
    }
    const size_t plane = (size_t)C * T;
    float* vp = V + (size_t)c * T + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {            // (B^T d) B
        vp[(size_t)(4 * i + 0) * plane] = e[i][0] - e[i][2];
        vp[(size_t)(4 * i + 1) * plane] = e[i][1] + e[i][2];
        vp[(size_t)(4 * i + 2) * plane] = e[i][2] - e[i][1];
        vp[(size_t)(4 * i + 3) * plane] = e[i][1] - e[i][3];
    }
}

// The same transform for e4s_gemm_pre: one thread = one tile x EIGHT channels, V written as bf16 hi / lo in channel blocks [16][C / 8][T][8]
// (16-byte pieces, consecutive threads = consecutive tiles).
__global__ __launch_bounds__(256) void wino_input_pre_kernel(uint4* __restrict__ Vhi, uint4* __restrict__ Vlo, const float* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, int bs, int C, int H, int W) {
    const int th = H >> 1, tw = W >> 1;
    const int T = bs * th * tw;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c8 = blockIdx.y;
    if (t >= T) return;
    const int tx = t % tw, ty = (t / tw) % th, b = t / (tw * th);
    const size_t plane = (size_t)gridDim.y * T;
    // Two passes (rows 0-1, then 2-3 of B^T d B) in a loop that is NOT unrolled.  The single-pass, fully unrolled form (112 loads issued before
    // the first use) produced occasional wrong values — one register, sixteen lanes — when another stream's kernels ran beside it
    // (tools/probes/wino_race6.py): what a load landing in an already reused register looks like (vmcnt counts to 63).  This form has not shown it.
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        unsigned oh[8][4], ol[8][4];
        float prev[8];
#pragma unroll
        for (int cj = 0; cj < 8; ++cj) {
            const int c = c8 * 8 + cj;
            const bool cok = c < C;
            const float* xp = x + ((size_t)b * C + (cok ? c : C - 1)) * H * W;
            const float mu = mean ? mean[(size_t)b * C + (cok ? c : C - 1)] : 0.f, rs = rstd ? rstd[(size_t)b * C + (cok ? c : C - 1)] : 1.f;
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int y = 2 * ty - 1 + i;
                const bool row = cok && y >= 0 && y < H;
                const float* r = xp + (size_t)((y >= 0 && y < H) ? y : 0) * W + 2 * tx;
                const float2 mid = *reinterpret_cast<const float2*>(r);
                const float lft = tx > 0 ? r[-1] : 0.f, rgt = 2 * tx + 2 < W ? r[2] : 0.f;
                d[i][0] = (row && tx > 0) ? (lft - mu) * rs : 0.f;
                d[i][1] = row ? (mid.x - mu) * rs : 0.f;
                d[i][2] = row ? (mid.y - mu) * rs : 0.f;
                d[i][3] = (row && 2 * tx + 2 < W) ? (rgt - mu) * rs : 0.f;
            }
            float e[2][4];          // rows 2 half, 2 half + 1 of B^T d
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[0][j] = half ? d[2][j] - d[1][j] : d[0][j] - d[2][j];
                e[1][j] = half ? d[1][j] - d[3][j] : d[1][j] + d[2][j];
            }
            float v[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                v[4 * i + 0] = e[i][0] - e[i][2];
                v[4 * i + 1] = e[i][1] + e[i][2];
                v[4 * i + 2] = e[i][2] - e[i][1];
                v[4 * i + 3] = e[i][1] - e[i][3];
            }
            if (cj & 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) split2(prev[k], v[k], oh[k][cj >> 1], ol[k][cj >> 1]);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) prev[k] = v[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            Vhi[(size_t)(8 * half + k) * plane + (size_t)c8 * T + t] = make_uint4(oh[k][0], oh[k][1], oh[k][2], oh[k][3]);
            Vlo[(size_t)(8 * half + k) * plane + (size_t)c8 * T + t] = make_uint4(ol[k][0], ol[k][1], ol[k][2], ol[k][3]);
        }
    }
}

__global__ __launch_bounds__(256) void wino_output_kernel(float* __restrict__ y, const float* __restrict__ M, const float* __restrict__ prelu, int bs, int CO,
                                                          int H, int W) {
    const int th = H >> 1, tw = W >> 1;
    const int T = bs * th * tw;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int co = blockIdx.y;
    if (t >= T) return;
    const int tx = t % tw, ty = (t / tw) % th, b = t / (tw * th);
    const size_t plane = (size_t)CO * T;
    const float* mp = M + (size_t)co * T + t;
    float m[4][4];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k >> 2][k & 3] = mp[(size_t)k * plane];
    float s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // A^T m
        s[0][j] = m[0][j] + m[1][j] + m[2][j];
        s[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    float o[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {            // (A^T m) A
        o[i][0] = s[i][0] + s[i][1] + s[i][2];
        o[i][1] = s[i][1] - s[i][2] - s[i][3];
    }
    if (prelu) {
        const float a = prelu[co];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) o[i][j] = o[i][j] >= 0.f ? o[i][j] : a * o[i][j];
    }
    float* yp = y + (((size_t)b * CO + co) * H + 2 * ty) * W + 2 * tx;
    *reinterpret_cast<float2*>(yp) = make_float2(o[0][0], o[0][1]);
    *reinterpret_cast<float2*>(yp + W) = make_float2(o[1][0], o[1][1]);
}

}  // namespace

extern "C" int e4s_wino_weight(float* U, const float* w, int cout, int cin, void* stream) {
    E4S_REQUIRE(U && w && cout >= 1 && cin >= 1, "wino_weight: bad arguments");
    hipLaunchKernelGGL(wino_weight_kernel, dim3(cdiv(cout * cin, 256)), dim3(256), 0, (hipStream_t)stream, U, w, cout, cin);
    return check_launch("wino_weight");
}

extern "C" int e4s_wino_input(float* V, const float* x, const float* mean, const float* rstd, int bs, int C, int H, int W, void* stream) {
    E4S_REQUIRE(V && x && (!mean == !rstd), "wino_input: null tensor (mean and rstd come together)");
    E4S_REQUIRE(bs >= 0 && C >= 1 && C <= 65535 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0 && (int64_t)bs * H * W / 4 < ((int64_t)1 << 30),
                "wino_input: even height and width");
    E4S_REQUIRE((((uintptr_t)x) & 7) == 0, "wino_input: input must be 8-byte aligned");
    if (bs == 0) return 0;
    const int T = bs * (H / 2) * (W / 2);
    hipLaunchKernelGGL(wino_input_kernel, dim3(cdiv(T, 256), C), dim3(256), 0, (hipStream_t)stream, V, x, mean, rstd, bs, C, H, W);
    return check_launch("wino_input");
}

extern "C" int e4s_wino_output(float* y, const float* M, const float* prelu, int bs, int cout, int H, int W, void* stream) {
    E4S_REQUIRE(y && M, "wino_output: null tensor");
    E4S_REQUIRE(bs >= 0 && cout >= 1 && cout <= 65535 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0, "wino_output: even height and width");
    E4S_REQUIRE((((uintptr_t)y) & 7) == 0, "wino_output: output must be 8-byte aligned");
    if (bs == 0) return 0;
    const int T = bs * (H / 2) * (W / 2);
    hipLaunchKernelGGL(wino_output_kernel, dim3(cdiv(T, 256), cout), dim3(256), 0, (hipStream_t)stream, y, M, prelu, bs, cout, H, W);
    return check_launch("wino_output");
}
