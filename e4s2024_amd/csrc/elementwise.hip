// a1 fused_bias_act, a2 upfirdn2d, one-hot -> label map.  HBM-bound streaming kernels (gfx950).
#include <stdarg.h>

#include "common.h"

namespace e4s {
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace e4s

using namespace e4s;

extern "C" int e4s_abi_version(void) { return E4S_ABI_VERSION; }
extern "C" const char* e4s_last_error(void) { return err_buf(); }

// ------------------------------------------------------------------------------------------ a1
// One float4 per lane per iteration (16 B/lane, 1 KiB per wave-instruction), grid-stride.
// The bias index (i / step_b) % size_b is evaluated per element; when step_b % 4 == 0 the four
// elements of a float4 share it.
__device__ __forceinline__ float bias_act_one(float v, float r, int mode, float alpha, float scale) {
    float y;
    switch (mode) {
        default:
        case 10:
        case 11: y = v; break;
        case 12: y = 0.f; break;
        case 30: y = (v > 0.f) ? v : v * alpha; break;
        case 31: y = (r > 0.f) ? v : v * alpha; break;
        case 32: y = 0.f; break;
    }
    return y * scale;
}

template <bool VEC>
__global__ __launch_bounds__(256) void fused_bias_act_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                             const float* __restrict__ bias, const float* __restrict__ ref,
                                                             int mode, float alpha, float scale, int64_t size_x,
                                                             int64_t step_b, int64_t size_b) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        const int64_t n4 = size_x >> 2;
        for (; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(x)[i];
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ref) r = reinterpret_cast<const float4*>(ref)[i];
            if (bias) {
                const float b = bias[((i << 2) / step_b) % size_b];
                v.x += b; v.y += b; v.z += b; v.w += b;
            }
            float4 y;
            y.x = bias_act_one(v.x, r.x, mode, alpha, scale);
            y.y = bias_act_one(v.y, r.y, mode, alpha, scale);
            y.z = bias_act_one(v.z, r.z, mode, alpha, scale);
            y.w = bias_act_one(v.w, r.w, mode, alpha, scale);
            reinterpret_cast<float4*>(out)[i] = y;
        }
    } else {
        for (; i < size_x; i += stride) {
            float v = x[i];
            if (bias) v += bias[(i / step_b) % size_b];
            const float r = ref ? ref[i] : 0.f;
            out[i] = bias_act_one(v, r, mode, alpha, scale);
        }
    }
}

extern "C" int e4s_fused_bias_act(float* out, const float* x, const float* bias, const float* ref, int act, int grad,
                                  float alpha, float scale, int64_t size_x, int64_t step_b, int64_t size_b, void* stream) {
    E4S_REQUIRE(size_x >= 0, "fused_bias_act: negative size");
    if (size_x == 0) return 0;
    E4S_REQUIRE(out && x, "fused_bias_act: null tensor");
    if (size_b <= 0) bias = nullptr;
    if (bias) E4S_REQUIRE(step_b > 0, "fused_bias_act: step_b must be positive");
    const int mode = act * 10 + grad;
    const bool vec = (size_x % 4 == 0) && (!bias || step_b % 4 == 0) && ((((uintptr_t)out | (uintptr_t)x | (uintptr_t)ref) & 15) == 0);
    const int64_t work = vec ? size_x / 4 : size_x;
    const int grid = (int)(cdiv64(work, 256) < 2048 ? cdiv64(work, 256) : 2048);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(fused_bias_act_kernel<true>, dim3(grid), dim3(256), 0, st, out, x, bias, ref, mode, alpha, scale, size_x, step_b, size_b);
    else
        hipLaunchKernelGGL(fused_bias_act_kernel<false>, dim3(grid), dim3(256), 0, st, out, x, bias, ref, mode, alpha, scale, size_x, step_b, size_b);
    return check_launch("fused_bias_act");
}

// ------------------------------------------------------------------------------------------ a2
// General upfirdn2d (any up/down/pad, kernel <= 32x32), one output element per thread; the taps sit in LDS
// already flipped, the input window of a 64x4 output tile is shared through L1.  Index arithmetic follows the
// per-output formulas of upfirdn2d_kernel.cu:106-131 of the reference.
__device__ __forceinline__ int floor_div(int a, int b) {
    int c = a / b;
    if (c * b > a) c--;
    return c;
}

struct UpfirdnParams {
    int major, in_h, in_w, kh, kw, out_h, out_w;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
};

__global__ __launch_bounds__(256) void upfirdn2d_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                        const float* __restrict__ kernel, UpfirdnParams p) {
    __shared__ float sk[32 * 32];
    for (int t = threadIdx.x; t < p.kh * p.kw; t += 256) {
        const int ky = t / p.kw, kx = t - ky * p.kw;
        sk[t] = kernel[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];  // flipped: true convolution
    }
    __syncthreads();
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= p.out_w || oy >= p.out_h) return;
    const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
    const int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
    const int ix0 = floor_div(mid_x, p.up_x);
    const int iy0 = floor_div(mid_y, p.up_y);
    const int kx0 = (ix0 + 1) * p.up_x - mid_x - 1;
    const int ky0 = (iy0 + 1) * p.up_y - mid_y - 1;
    for (int m = blockIdx.z; m < p.major; m += gridDim.z) {
        const float* src = in + (size_t)m * p.in_h * p.in_w;
        float v = 0.f;
        for (int ky = ky0, iy = iy0; ky < p.kh; ky += p.up_y, ++iy) {
            if (iy < 0 || iy >= p.in_h) continue;
            for (int kx = kx0, ix = ix0; kx < p.kw; kx += p.up_x, ++ix) {
                if (ix < 0 || ix >= p.in_w) continue;
                v += src[(size_t)iy * p.in_w + ix] * sk[ky * p.kw + kx];
            }
        }
        out[((size_t)m * p.out_h + oy) * p.out_w + ox] = v;
    }
}

// The 4 x 4 FIR without up-sampling (down = 1: the blur and its transpose in the up layers' backward pass; down = 2: the transpose of the
// skip image's FIR up-sampling): four outputs of a row per thread, every input row loaded once for them — 28 / 40 loads per four outputs
// instead of 64.  Taps in the generic kernel's order (ky outer, kx inner, out-of-range inputs contribute +0): the same sums.
template <int DOWN>
__global__ __launch_bounds__(256) void fir4x4_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ kernel, UpfirdnParams p) {
    __shared__ float sk[16];
    if (threadIdx.x < 16) sk[threadIdx.x] = kernel[15 - threadIdx.x];   // flipped: true convolution
    __syncthreads();
    const int ox0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox0 >= p.out_w || oy >= p.out_h) return;
    constexpr int NIN = 3 * DOWN + 4;                                    // input columns under four outputs
    const int ix0 = ox0 * DOWN - p.pad_x0, iy0 = oy * DOWN - p.pad_y0;
    float k[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) k[t] = sk[t];
    for (int m = blockIdx.z; m < p.major; m += gridDim.z) {
        const float* src = in + (size_t)m * p.in_h * p.in_w;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int iy = iy0 + ky;
            const bool row = iy >= 0 && iy < p.in_h;
            const float* r = src + (size_t)(row ? iy : 0) * p.in_w;
            float a[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                const int ix = ix0 + j;
                a[j] = (row && ix >= 0 && ix < p.in_w) ? r[ix] : 0.f;
            }
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                for (int o = 0; o < 4; ++o) v[o] += a[o * DOWN + kx] * k[ky * 4 + kx];
        }
        float* dst = out + ((size_t)m * p.out_h + oy) * p.out_w + ox0;
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (ox0 + o < p.out_w) dst[o] = v[o];
    }
}

extern "C" int e4s_upfirdn2d(float* out, const float* in, const float* kernel, int major, int in_h, int in_w, int kh, int kw,
                             int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                             void* stream) {
    E4S_REQUIRE(kh >= 1 && kw >= 1 && kh <= 32 && kw <= 32, "upfirdn2d: kernel %dx%d not in 1..32", kh, kw);
    E4S_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "upfirdn2d: up/down must be >= 1");
    E4S_REQUIRE(major >= 0 && in_h >= 1 && in_w >= 1, "upfirdn2d: bad input size");
    UpfirdnParams p;
    p.major = major; p.in_h = in_h; p.in_w = in_w; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    const int num_h = in_h * up_y + pad_y0 + pad_y1 - kh, num_w = in_w * up_x + pad_x0 + pad_x1 - kw;
    E4S_REQUIRE(num_h >= 0 && num_w >= 0, "upfirdn2d: output would be empty");
    p.out_h = num_h / down_y + 1;
    p.out_w = num_w / down_x + 1;
    if (major == 0) return 0;
    E4S_REQUIRE(out && in && kernel, "upfirdn2d: null tensor");
    if (kh == 4 && kw == 4 && up_x == 1 && up_y == 1 && down_x == down_y && (down_x == 1 || down_x == 2)) {
        dim3 g4(cdiv(p.out_w, 256), cdiv(p.out_h, 4), major < 65535 ? major : 65535);
        if (down_x == 1) hipLaunchKernelGGL(fir4x4_kernel<1>, g4, dim3(256), 0, (hipStream_t)stream, out, in, kernel, p);
        else hipLaunchKernelGGL(fir4x4_kernel<2>, g4, dim3(256), 0, (hipStream_t)stream, out, in, kernel, p);
        return check_launch("upfirdn2d");
    }
    dim3 grid(cdiv(p.out_w, 64), cdiv(p.out_h, 4), major < 65535 ? major : 65535);
    hipLaunchKernelGGL(upfirdn2d_kernel, grid, dim3(256), 0, (hipStream_t)stream, out, in, kernel, p);
    return check_launch("upfirdn2d");
}

// ------------------------------------------------------------------------------- one-hot -> labels
__global__ __launch_bounds__(256) void onehot_to_labels_kernel(uint8_t* __restrict__ labels, int* __restrict__ flag,
                                                               const float* __restrict__ mask, int ncls, int hw, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / hw;
    const int pix = (int)(i - b * hw);
    const float* m = mask + (size_t)b * ncls * hw + pix;
    int lab = E4S_LABEL_NONE, cnt = 0, bad = 0;
    for (int c = 0; c < ncls; ++c) {
        const float v = m[(size_t)c * hw];
        if (v != 0.f) {
            if (v != 1.f) bad |= 1;
            if (cnt == 0) lab = c;
            ++cnt;
        }
    }
    if (cnt > 1) bad |= 2;
    labels[i] = (uint8_t)lab;
    if (bad && flag) atomicOr(flag, bad);
}

extern "C" int e4s_onehot_to_labels(uint8_t* labels, int* flag, const float* mask, int bs, int ncls, int h, int w, void* stream) {
    E4S_REQUIRE(ncls >= 1 && ncls <= E4S_MAX_REGIONS, "onehot_to_labels: %d classes (max %d)", ncls, E4S_MAX_REGIONS);
    E4S_REQUIRE(bs >= 0 && h >= 1 && w >= 1, "onehot_to_labels: bad size");
    const int64_t total = (int64_t)bs * h * w;
    if (total == 0) return 0;
    E4S_REQUIRE(labels && mask, "onehot_to_labels: null tensor");
    hipLaunchKernelGGL(onehot_to_labels_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, labels, flag, mask,
                       ncls, h * w, total);
    return check_launch("onehot_to_labels");
}
