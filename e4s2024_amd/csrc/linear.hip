// a7: grouped equalised linear (LocalMLP stack, EqualLinear).  Weight-bandwidth bound (195 MB of fp32 weights for the
// 12 MLPs): one wave streams one weight row with 16-byte loads and dots it against up to 8 input rows held in L1/L2,
// wave-shuffle reduction, fused bias / leaky-relu / latent_avg add.
#include "common.h"

using namespace e4s;

constexpr int LIN_BT = 8;  // batch rows per pass over the weights
constexpr int LIN_MAX_GROUPS = 16;

struct GroupPtrs {
    const float* W[LIN_MAX_GROUPS];
    const float* bias[LIN_MAX_GROUPS];
};

template <bool VEC>
__global__ __launch_bounds__(256) void grouped_linear_kernel(float* __restrict__ out, int64_t out_stride_b, int64_t out_stride_g,
                                                             const float* __restrict__ x, int64_t x_stride_b, int64_t x_stride_g,
                                                             const GroupPtrs gp,
                                                             const float* __restrict__ addend, float scale, float bias_mul, int act,
                                                             float slope, int bs, int in_dim, int out_dim) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = blockIdx.y;
    if (o >= out_dim) return;
    const float* __restrict__ wrow = gp.W[g] + (size_t)o * in_dim;
    const float* __restrict__ bias = gp.bias[g];
    for (int b0 = 0; b0 < bs; b0 += LIN_BT) {
        float acc[LIN_BT];
#pragma unroll
        for (int j = 0; j < LIN_BT; ++j) acc[j] = 0.f;
        if (VEC) {
            for (int i = lane * 4; i < in_dim; i += 256) {
                const float4 w4 = *reinterpret_cast<const float4*>(wrow + i);
#pragma unroll
                for (int j = 0; j < LIN_BT; ++j) {
                    if (b0 + j < bs) {
                        const float4 x4 = *reinterpret_cast<const float4*>(x + (b0 + j) * x_stride_b + g * x_stride_g + i);
                        acc[j] += w4.x * x4.x + w4.y * x4.y + w4.z * x4.z + w4.w * x4.w;
                    }
                }
            }
        } else {
            for (int i = lane; i < in_dim; i += 64) {
                const float wv = wrow[i];
#pragma unroll
                for (int j = 0; j < LIN_BT; ++j)
                    if (b0 + j < bs) acc[j] += wv * x[(b0 + j) * x_stride_b + g * x_stride_g + i];
            }
        }
#pragma unroll
        for (int j = 0; j < LIN_BT; ++j) {
            if (b0 + j >= bs) break;
            float v = wave_sum(acc[j]);
            if (lane == 0) {
                v = v * scale;
                if (bias) v += bias[o] * bias_mul;
                if (act == 1) v = v > 0.f ? v : v * slope;
                if (act == 2) v = (v > 0.f ? v : v * slope) * 1.41421356237309515f;
                if (addend) v += addend[o];
                out[(b0 + j) * out_stride_b + g * out_stride_g + o] = v;
            }
        }
    }
}

extern "C" int e4s_grouped_linear(float* out, int64_t out_stride_b, int64_t out_stride_g, const float* x, int64_t x_stride_b,
                                  int64_t x_stride_g, const float* const* W, const float* const* bias, const float* addend, float scale,
                                  float bias_mul, int act, float slope, int bs, int groups, int in_dim, int out_dim, void* stream) {
    E4S_REQUIRE(out && x && W, "grouped_linear: null tensor");
    E4S_REQUIRE(bs >= 0 && groups >= 1 && groups <= LIN_MAX_GROUPS && in_dim >= 1 && out_dim >= 1, "grouped_linear: bad size (groups <= %d)",
                LIN_MAX_GROUPS);
    GroupPtrs gp;
    uintptr_t align = (uintptr_t)x;
    for (int g = 0; g < LIN_MAX_GROUPS; ++g) {
        gp.W[g] = g < groups ? W[g] : nullptr;
        gp.bias[g] = (g < groups && bias) ? bias[g] : nullptr;
        if (g < groups) {
            E4S_REQUIRE(W[g], "grouped_linear: null weight pointer for group %d", g);
            align |= (uintptr_t)W[g];
        }
    }
    E4S_REQUIRE(act >= 0 && act <= 2, "grouped_linear: act must be 0, 1 or 2");
    if (bs == 0) return 0;
    const bool vec = (in_dim % 4 == 0) && (x_stride_b % 4 == 0) && (x_stride_g % 4 == 0) && ((align & 15) == 0);
    dim3 grid(cdiv(out_dim, 4), groups);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(grouped_linear_kernel<true>, grid, dim3(256), 0, st, out, out_stride_b, out_stride_g, x, x_stride_b, x_stride_g, gp,
                           addend, scale, bias_mul, act, slope, bs, in_dim, out_dim);
    else
        hipLaunchKernelGGL(grouped_linear_kernel<false>, grid, dim3(256), 0, st, out, out_stride_b, out_stride_g, x, x_stride_b, x_stride_g,
                           gp, addend, scale, bias_mul, act, slope, bs, in_dim, out_dim);
    return check_launch("grouped_linear");
}

// ------------------------------------------------------------------------------------ backward of the grouped linear (f1: the PTI loop trains the MLPs)
// y[b,g,o] = act(scale * sum_i W[g][o][i] x[b,g,i] + bias_mul * bias[g][o]);  gy = dL/dy, already multiplied by act'(y) by the caller's kernel below.
//
// (1) weight / bias gradients: dW[g][o][i] = scale * sum_b gy[b,g,o] x[b,g,i]  (an outer product per group: written once, 16-byte stores),
//     db[g][o] = bias_mul * sum_b gy[b,g,o].  dW / db are dense [groups][out][in] / [groups][out]: the caller hands out per-group views.
__global__ __launch_bounds__(256) void grouped_linear_wgrad_kernel(float* __restrict__ dW, float* __restrict__ db, const float* __restrict__ gy,
                                                                   const float* __restrict__ x, float scale, float bias_mul, int bs, int groups, int in_dim,
                                                                   int out_dim) {
    const int g = blockIdx.y;
    const int o0 = blockIdx.x * 8;
    const int n4 = in_dim >> 2;
    for (int i4 = threadIdx.x; i4 < n4; i4 += 256) {
        float4 acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < bs; ++b) {
            const float4 xv = *reinterpret_cast<const float4*>(x + ((size_t)b * groups + g) * in_dim + 4 * i4);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float gv = o0 + r < out_dim ? gy[((size_t)b * groups + g) * out_dim + o0 + r] : 0.f;
                acc[r].x += gv * xv.x; acc[r].y += gv * xv.y; acc[r].z += gv * xv.z; acc[r].w += gv * xv.w;
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (o0 + r < out_dim)
                *reinterpret_cast<float4*>(dW + ((size_t)g * out_dim + o0 + r) * in_dim + 4 * i4) = make_float4(acc[r].x * scale, acc[r].y * scale, acc[r].z * scale, acc[r].w * scale);
    }
    if (db && threadIdx.x < 8 && o0 + (int)threadIdx.x < out_dim) {
        float a = 0.f;
        for (int b = 0; b < bs; ++b) a += gy[((size_t)b * groups + g) * out_dim + o0 + threadIdx.x];
        db[(size_t)g * out_dim + o0 + threadIdx.x] = a * bias_mul;
    }
}

// (2) input gradient: dx[b,g,i] = scale * sum_o W[g][o][i] gy[b,g,o], optionally times lrelu'(h[b,g,i]) of the PREVIOUS layer's output h (the
//     sign of a leaky-relu output is the sign of its input).  The sum over o is split over `osplit` workgroups per (group, 1024 inputs); the
//     partial sums are added in a fixed order by the finishing kernel (no atomics).
constexpr int LIN_BWD_BT = 8;
__global__ __launch_bounds__(256) void grouped_linear_dgrad_kernel(float* __restrict__ part, const float* __restrict__ gy, const GroupPtrs gp, int bs, int groups,
                                                                   int in_dim, int out_dim, int osplit) {
    const int g = blockIdx.y, os = blockIdx.x, i4 = blockIdx.z * 256 + threadIdx.x;
    const int n4 = in_dim >> 2;
    if (i4 >= n4) return;
    const int per = (out_dim + osplit - 1) / osplit;
    const int o_begin = os * per, o_end = o_begin + per < out_dim ? o_begin + per : out_dim;
    const float* __restrict__ W = gp.W[g];
    float4 acc[LIN_BWD_BT];
#pragma unroll
    for (int b = 0; b < LIN_BWD_BT; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int o = o_begin; o < o_end; ++o) {
        const float4 w = *reinterpret_cast<const float4*>(W + (size_t)o * in_dim + 4 * i4);
#pragma unroll
        for (int b = 0; b < LIN_BWD_BT; ++b)
            if (b < bs) {
                const float gv = gy[((size_t)b * groups + g) * out_dim + o];
                acc[b].x += gv * w.x; acc[b].y += gv * w.y; acc[b].z += gv * w.z; acc[b].w += gv * w.w;
            }
    }
#pragma unroll
    for (int b = 0; b < LIN_BWD_BT; ++b)
        if (b < bs) *reinterpret_cast<float4*>(part + ((((size_t)os * bs + b) * groups + g) * in_dim) + 4 * i4) = acc[b];
}

__global__ __launch_bounds__(256) void grouped_linear_dgrad_finish_kernel(float* __restrict__ dx, const float* __restrict__ part, const float* __restrict__ h, float scale,
                                                                          float slope, long long total, int osplit) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float a = 0.f;
    for (int k = 0; k < osplit; ++k) a += part[(size_t)k * total + i];
    a *= scale;
    if (h) a *= h[i] > 0.f ? 1.f : slope;
    dx[i] = a;
}

extern "C" int e4s_grouped_linear_bwd(float* dW, float* db, float* dx, float* scratch, const float* gy, const float* x, const float* const* W, const float* h_prev,
                                      float scale, float bias_mul, float slope, int bs, int groups, int in_dim, int out_dim, int osplit, void* stream) {
    E4S_REQUIRE(gy && x, "grouped_linear_bwd: null tensor");
    E4S_REQUIRE(bs >= 1 && bs <= LIN_BWD_BT && groups >= 1 && groups <= LIN_MAX_GROUPS && in_dim >= 4 && (in_dim % 4) == 0 && out_dim >= 1 && osplit >= 1 && osplit <= 256,
                "grouped_linear_bwd: bad size (batch <= %d, groups <= %d, in_dim a multiple of 4)", LIN_BWD_BT, LIN_MAX_GROUPS);
    E4S_REQUIRE(!dx || (W && scratch), "grouped_linear_bwd: the input gradient needs the weights and a scratch of osplit * bs * groups * in_dim floats");
    E4S_REQUIRE((((uintptr_t)x | (uintptr_t)dW | (uintptr_t)scratch) & 15) == 0, "grouped_linear_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dW)
        hipLaunchKernelGGL(grouped_linear_wgrad_kernel, dim3(cdiv(out_dim, 8), groups), dim3(256), 0, st, dW, db, gy, x, scale, bias_mul, bs, groups, in_dim, out_dim);
    if (dx) {
        GroupPtrs gp;
        for (int g = 0; g < LIN_MAX_GROUPS; ++g) {
            gp.W[g] = g < groups ? W[g] : nullptr;
            gp.bias[g] = nullptr;
            if (g < groups) E4S_REQUIRE(W[g] && ((uintptr_t)W[g] & 15) == 0, "grouped_linear_bwd: null / unaligned weight pointer for group %d", g);
        }
        hipLaunchKernelGGL(grouped_linear_dgrad_kernel, dim3(osplit, groups, cdiv(in_dim / 4, 256)), dim3(256), 0, st, scratch, gy, gp, bs, groups, in_dim, out_dim, osplit);
        const long long total = (long long)bs * groups * in_dim;
        hipLaunchKernelGGL(grouped_linear_dgrad_finish_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, dx, scratch, h_prev, scale, slope, total, osplit);
    }
    return check_launch("grouped_linear_bwd");
}

// ------------------------------------------------------------------------------------ small constant linear map over a long axis
// out[j][n] = sum_k T[j][k] * w[n][k]      (trans = 0: w [N][K] -> out [J][N];   the parity composition of an up layer's 3x3 weight with its blur:
//                                           J = 36 composed taps, K = 9, N = cout * cin, T a constant of the blur kernel)
// dw[n][k]  = sum_j T[j][k] * g[j][n]      (trans = 1: its transpose, the gradient of w)
// J, K <= 36: T lives in LDS, one thread per n.
constexpr int SM_MAX = 36;
__global__ __launch_bounds__(256) void small_map_kernel(float* __restrict__ out, const float* __restrict__ T, const float* __restrict__ in, int J, int K, long long N,
                                                        int trans) {
    __shared__ float t[SM_MAX * SM_MAX];
    for (int i = threadIdx.x; i < J * K; i += 256) t[i] = T[i];
    __syncthreads();
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    if (!trans) {
        float w[SM_MAX];
        for (int k = 0; k < K; ++k) w[k] = in[n * K + k];
        for (int j = 0; j < J; ++j) {
            float a = 0.f;
            for (int k = 0; k < K; ++k) a += t[j * K + k] * w[k];
            out[(long long)j * N + n] = a;
        }
    } else {
        float a[SM_MAX];
        for (int k = 0; k < K; ++k) a[k] = 0.f;
        for (int j = 0; j < J; ++j) {
            const float gv = in[(long long)j * N + n];
            for (int k = 0; k < K; ++k) a[k] += t[j * K + k] * gv;
        }
        for (int k = 0; k < K; ++k) out[n * K + k] = a[k];
    }
}

// The shapes the parity composition has (J = 36, K = 9) with everything unrolled and the K-strided sides staged through LDS so that every global
// access is a whole line: 256 columns n per workgroup, w / dw rows [n][9] move as 2304 consecutive floats.  GROUPED: the J side is stored
// [J / 9][N][9] (out[g][n][t] for j = 9 g + t) instead of [J][N] — the four parity weights as [4][cout][cin][3][3], no permuted copy after it.
template <int J, int K, bool GROUPED>
__global__ __launch_bounds__(256) void small_map_fixed_kernel(float* __restrict__ out, const float* __restrict__ T, const float* __restrict__ in, long long N, int trans) {
    __shared__ float t[J * K];
    __shared__ float rows[256 * K];
    constexpr int KO = 9, G = J / KO;
    static_assert(!GROUPED || (K == KO && J % KO == 0), "the grouped layout shares the staging rows: groups of 9");
    for (int i = threadIdx.x; i < J * K; i += 256) t[i] = T[i];
    const long long n0 = (long long)blockIdx.x * 256;
    const int cols = (int)(N - n0 < 256 ? N - n0 : 256);
    const int cnt = cols * K;                                            // floats of the [n][K] side this workgroup owns
    const long long n = n0 + threadIdx.x;
    if (!trans) {
        for (int i = threadIdx.x; i < cnt; i += 256) rows[i] = in[n0 * K + i];
        __syncthreads();
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = rows[threadIdx.x * K + k];
        if (!GROUPED) {
            if (n >= N) return;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) a += t[j * K + k] * w[k];
                out[(long long)j * N + n] = a;
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                __syncthreads();                                         // (rows is reused as the output stage of group g)
#pragma unroll
                for (int q = 0; q < KO; ++q) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) a += t[(g * KO + q) * K + k] * w[k];
                    rows[threadIdx.x * KO + q] = a;
                }
                __syncthreads();
                for (int i = threadIdx.x; i < cols * KO; i += 256) out[((long long)g * N + n0) * KO + i] = rows[i];
            }
        }
    } else {
        __syncthreads();
        float a[K];
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = 0.f;
        if (!GROUPED) {
            if (n < N) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const float gv = in[(long long)j * N + n];
#pragma unroll
                    for (int k = 0; k < K; ++k) a[k] += t[j * K + k] * gv;
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                __syncthreads();
                for (int i = threadIdx.x; i < cols * KO; i += 256) rows[i] = in[((long long)g * N + n0) * KO + i];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < KO; ++q) {
                    const float gv = rows[threadIdx.x * KO + q];         // (columns beyond N read stale LDS and are never written out)
#pragma unroll
                    for (int k = 0; k < K; ++k) a[k] += t[(g * KO + q) * K + k] * gv;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < K; ++k) rows[threadIdx.x * K + k] = a[k];
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) out[n0 * K + i] = rows[i];
    }
}

extern "C" int e4s_small_map(float* out, const float* T, const float* in, int J, int K, int64_t N, int trans, int grouped, void* stream) {
    E4S_REQUIRE(out && T && in, "small_map: null tensor");
    E4S_REQUIRE(J >= 1 && J <= SM_MAX && K >= 1 && K <= SM_MAX && N >= 0, "small_map: J, K in 1..%d", SM_MAX);
    E4S_REQUIRE(!grouped || (J == 36 && K == 9), "small_map: the grouped layout is built for J = 36, K = 9");
    if (N == 0) return 0;
    const dim3 grid((unsigned)cdiv64(N, 256));
    if (J == 36 && K == 9 && grouped)
        hipLaunchKernelGGL((small_map_fixed_kernel<36, 9, true>), grid, dim3(256), 0, (hipStream_t)stream, out, T, in, (long long)N, trans);
    else if (J == 36 && K == 9)
        hipLaunchKernelGGL((small_map_fixed_kernel<36, 9, false>), grid, dim3(256), 0, (hipStream_t)stream, out, T, in, (long long)N, trans);
    else
        hipLaunchKernelGGL(small_map_kernel, grid, dim3(256), 0, (hipStream_t)stream, out, T, in, J, K, (long long)N, trans);
    return check_launch("small_map");
}
