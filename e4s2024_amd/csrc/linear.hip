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
