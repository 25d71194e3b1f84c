// a9 / a10 support kernels (BiSeNet face parser): max-pool, attention gating + nearest upsample, fused bilinear
// (align_corners) upsample + argmax (+ 19->12 remap) writing uint8 labels, bicubic down-sample + clamp + normalise.
#include "common.h"

using namespace e4s;

// ------------------------------------------------------------------------------------ MaxPool2d(3, 2, 1)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(float* __restrict__ out, const float* __restrict__ in, int h, int w, int oh, int ow) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const float* p = in + (size_t)blockIdx.z * h * w;
    float m = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = 2 * y + dy;
        if (iy < 0 || iy >= h) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int ix = 2 * x + dx;
            if (ix < 0 || ix >= w) continue;
            m = fmaxf(m, p[(size_t)iy * w + ix]);
        }
    }
    out[((size_t)blockIdx.z * oh + y) * ow + x] = m;
}

// Two adjacent outputs per thread (w % 4 == 0): output columns 2t, 2t + 1 see input columns 4t - 1 .. 4t + 3 — one aligned 16-byte load and one dword per input row instead
// of nine dwords per output (the same maxima: max is exact in any order).  142 -> ~75 us on the parser's 16 x 64 planes of 256 x 256.
__global__ __launch_bounds__(256) void maxpool3x3s2_x2_kernel(float* __restrict__ out, const float* __restrict__ in, int h, int w, int oh, int ow) {
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);          // output column pair
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * t >= ow || y >= oh) return;
    const float* p = in + (size_t)blockIdx.z * h * w;
    float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = 2 * y + dy;
        if (iy < 0 || iy >= h) continue;
        const float* r = p + (size_t)iy * w + 4 * t;
        const float4 v = *reinterpret_cast<const float4*>(r);
        const float l = t > 0 ? r[-1] : -INFINITY;
        m0 = fmaxf(m0, fmaxf(l, fmaxf(v.x, v.y)));
        m1 = fmaxf(m1, fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    *reinterpret_cast<float2*>(out + ((size_t)blockIdx.z * oh + y) * ow + 2 * t) = make_float2(m0, m1);
}

extern "C" int e4s_maxpool3x3s2(float* out, const float* in, int planes, int h, int w, void* stream) {
    E4S_REQUIRE(out && in, "maxpool3x3s2: null tensor");
    E4S_REQUIRE(planes >= 0 && planes <= 65535 && h >= 1 && w >= 1, "maxpool3x3s2: bad size");
    if (planes == 0) return 0;
    const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
    if (w % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 7) == 0)       // (then ow = w / 2 is even: whole pairs)
        hipLaunchKernelGGL(maxpool3x3s2_x2_kernel, dim3(cdiv(ow / 2, 64), cdiv(oh, 4), planes), dim3(256), 0, (hipStream_t)stream, out, in, h, w, oh, ow);
    else
        hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(cdiv(ow, 64), cdiv(oh, 4), planes), dim3(256), 0, (hipStream_t)stream, out, in, h, w, oh, ow);
    return check_launch("maxpool3x3s2");
}

// ------------------------------------------------------------------------------------ gate * feat + addend, nearest x`up`
// out[b,c,Y,X] = feat[b,c,Y/up,X/up] * gate[b,c] + (add_map ? add_map[b,c,Y/up,X/up] : 0) + (add_vec ? add_vec[b,c] : 0)
__global__ __launch_bounds__(256) void gate_add_up_kernel(float* __restrict__ out, const float* __restrict__ feat, const float* __restrict__ gate,
                                                          const float* __restrict__ add_map, const float* __restrict__ add_vec, int h, int w,
                                                          int up) {
    const int ow = w * up, oh = h * up;
    const int X = blockIdx.x * 64 + (threadIdx.x & 63);
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= ow || Y >= oh) return;
    const int plane = blockIdx.z;
    const size_t src = (size_t)plane * h * w + (size_t)(Y / up) * w + X / up;
    float v = feat[src] * (gate ? gate[plane] : 1.f);
    if (add_map) v += add_map[src];
    if (add_vec) v += add_vec[plane];
    out[((size_t)plane * oh + Y) * ow + X] = v;
}

extern "C" int e4s_gate_add_upsample(float* out, const float* feat, const float* gate, const float* add_map, const float* add_vec, int planes,
                                     int h, int w, int up, void* stream) {
    E4S_REQUIRE(out && feat, "gate_add_upsample: null tensor");
    E4S_REQUIRE(planes >= 0 && planes <= 65535 && h >= 1 && w >= 1 && up >= 1, "gate_add_upsample: bad size");
    if (planes == 0) return 0;
    hipLaunchKernelGGL(gate_add_up_kernel, dim3(cdiv(w * up, 64), cdiv(h * up, 4), planes), dim3(256), 0, (hipStream_t)stream, out, feat, gate,
                       add_map, add_vec, h, w, up);
    return check_launch("gate_add_upsample");
}

// ------------------------------------------------------------------------------------ bilinear(align_corners=True) + argmax
// labels[b,Y,X] = lut[ argmax_c bilinear(logits[b,c])(Y,X) ]   (first maximum wins, like torch.argmax); the ncls x H x W
// up-sampled logits (60 MB at 19 x 512^2 x 3 heads in the reference) never reach HBM.
__global__ __launch_bounds__(256) void bilinear_argmax_kernel(uint8_t* __restrict__ labels, const float* __restrict__ logits,
                                                              const uint8_t* __restrict__ lut, int ncls, int ih, int iw, int oh, int ow,
                                                              float sy, float sx) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const float fy = (float)y * sy, fx = (float)x * sx;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > ih - 1) y0 = ih - 1;
    if (x0 > iw - 1) x0 = iw - 1;
    const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* p = logits + (size_t)blockIdx.z * ncls * ih * iw;
    float best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < ncls; ++c) {
        const float* q = p + (size_t)c * ih * iw;
        const float v = hy * (hx * q[(size_t)y0 * iw + x0] + lx * q[(size_t)y0 * iw + x1]) + ly * (hx * q[(size_t)y1 * iw + x0] + lx * q[(size_t)y1 * iw + x1]);
        if (v > best) {
            best = v;
            bi = c;
        }
    }
    labels[((size_t)blockIdx.z * oh + y) * ow + x] = lut ? lut[bi] : (uint8_t)bi;
}

// The same with the block's corner of the logits staged in LDS first: the 64 x 4 outputs of a block touch (int)(63 sx) + 3 columns and (int)(3 sy) + 3 rows of every class
// plane (10 x 3 at the parser's 64^2 -> 512^2) — 570 coalesced loads per block instead of 256 x 76 cached ones; the interpolation expression is the one above, value for value.
__global__ __launch_bounds__(256) void bilinear_argmax_lds_kernel(uint8_t* __restrict__ labels, const float* __restrict__ logits,
                                                                  const uint8_t* __restrict__ lut, int ncls, int ih, int iw, int oh, int ow,
                                                                  float sy, float sx, int ry, int rx) {
    extern __shared__ float reg[];                 // [ncls][ry][rx]
    const float* p = logits + (size_t)blockIdx.z * ncls * ih * iw;
    int ry0 = (int)((float)(blockIdx.y * 4) * sy), rx0 = (int)((float)(blockIdx.x * 64) * sx);
    if (ry0 > ih - 1) ry0 = ih - 1;
    if (rx0 > iw - 1) rx0 = iw - 1;
    const int rr = ry * rx;
    for (int e = threadIdx.x; e < ncls * rr; e += 256) {
        const int c = e / rr, r = e - c * rr;
        const int yy = min(ry0 + r / rx, ih - 1), xx = min(rx0 + r % rx, iw - 1);
        reg[e] = p[(size_t)c * ih * iw + (size_t)yy * iw + xx];
    }
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const float fy = (float)y * sy, fx = (float)x * sx;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > ih - 1) y0 = ih - 1;
    if (x0 > iw - 1) x0 = iw - 1;
    const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const int i00 = (y0 - ry0) * rx + (x0 - rx0), i01 = (y0 - ry0) * rx + (x1 - rx0), i10 = (y1 - ry0) * rx + (x0 - rx0), i11 = (y1 - ry0) * rx + (x1 - rx0);
    float best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < ncls; ++c) {
        const float* q = reg + c * rr;
        const float v = hy * (hx * q[i00] + lx * q[i01]) + ly * (hx * q[i10] + lx * q[i11]);
        if (v > best) {
            best = v;
            bi = c;
        }
    }
    labels[((size_t)blockIdx.z * oh + y) * ow + x] = lut ? lut[bi] : (uint8_t)bi;
}

extern "C" int e4s_bilinear_argmax(uint8_t* labels, const float* logits, const uint8_t* lut, int bs, int ncls, int ih, int iw, int oh, int ow,
                                   void* stream) {
    E4S_REQUIRE(labels && logits, "bilinear_argmax: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && ncls >= 1 && ncls <= 255 && ih >= 1 && iw >= 1 && oh >= 1 && ow >= 1, "bilinear_argmax: bad size");
    if (bs == 0) return 0;
    const float sy = oh > 1 ? (float)(ih - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(iw - 1) / (float)(ow - 1) : 0.f;
    const int ry = (int)(3.f * sy) + 3, rx = (int)(63.f * sx) + 3;      // rows / columns of a class plane that a block's 4 x 64 outputs can touch (+ 1 for the upper corner, + 1 for rounding)
    const size_t lds = (size_t)ncls * ry * rx * sizeof(float);
    if (lds <= 32 * 1024)
        hipLaunchKernelGGL(bilinear_argmax_lds_kernel, dim3(cdiv(ow, 64), cdiv(oh, 4), bs), dim3(256), lds, (hipStream_t)stream, labels, logits, lut, ncls,
                           ih, iw, oh, ow, sy, sx, ry, rx);
    else
        hipLaunchKernelGGL(bilinear_argmax_kernel, dim3(cdiv(ow, 64), cdiv(oh, 4), bs), dim3(256), 0, (hipStream_t)stream, labels, logits, lut, ncls,
                           ih, iw, oh, ow, sy, sx);
    return check_launch("bilinear_argmax");
}

// ------------------------------------------------------------------------------------ bicubic down-sample + clamp + normalise
// BicubicDownSample (separable 4*f-tap kernel, a = -0.5, reflect padding, vertical pass then horizontal pass) followed by
// clamp(0,1) and (x - mean[c]) / std[c].   in [planes = bs*3, H, W] -> out [planes, H/f, W/f]
__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// One block = a (64 / (F / 2)) x (16 / (F / 2)) tile of outputs (64 x 16 for F = 2, 32 x 8 for F = 4).  The input region (F * tile + K - F on a side, reflect-padded) is read
// ONCE, coalesced, into LDS; the vertical pass writes a [tile rows][region columns] buffer; the horizontal pass reads it.  Sums in the order of the direct form this replaces
// (column: taps i = 0 .. K - 1 ascending; then the row taps j ascending) — value for value the same results; the direct form read 64 (F = 2) scattered values per output
// from global memory: 434 us for the sixteen 1024^2 images of a swap batch whose bytes take 60.
// PM1: the input is the swap pipeline's [-1, 1] image; (v + 1) * 0.5 on load is torch's ``(img + 1) / 2`` value for value (the pass this saves read and wrote the batch twice).
template <int F, bool PM1>
__global__ __launch_bounds__(256) void bicubic_down_norm_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                                const float* __restrict__ taps, const float* __restrict__ mean,
                                                                const float* __restrict__ stdv, int C, int h, int w, int oh, int ow, int do_norm) {
    constexpr int K = 4 * F;
    constexpr int TW = 128 / F, TH = 32 / F;                  // output tile
    constexpr int RW = TW * F + K - F, RH = TH * F + K - F;   // input region
    constexpr int PADT = (K - F) / 2;
    __shared__ float reg[RH][RW + 1];
    __shared__ float colb[TH][RW + 1];
    const int plane = blockIdx.z;
    const float* p = in + (size_t)plane * h * w;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const int ix0 = ox0 * F - PADT, iy0 = oy0 * F - PADT;
    float t[K];
#pragma unroll
    for (int i = 0; i < K; ++i) t[i] = taps[i];
    for (int e = threadIdx.x; e < RH * RW; e += 256) {
        const int ry = e / RW, rx = e - ry * RW;
        // (rows / columns of a partial tile far outside the image feed no stored output: reflected once, then clamped so that the read stays inside the plane)
        const int sy = min(max(reflect_idx(iy0 + ry, h), 0), h - 1), sx = min(max(reflect_idx(ix0 + rx, w), 0), w - 1);
        const float v = p[(size_t)sy * w + sx];
        reg[ry][rx] = PM1 ? (v + 1.f) * 0.5f : v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TH * RW; e += 256) {       // vertical pass (first pass of the reference's separable filter)
        const int ty = e / RW, rx = e - ty * RW;
        float col = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) col += t[i] * reg[ty * F + i][rx];
        colb[ty][rx] = col;
    }
    __syncthreads();
    const int c = plane % C;
    const float m = do_norm ? mean[c] : 0.f, sd = do_norm ? stdv[c] : 1.f;
    for (int e = threadIdx.x; e < TH * TW; e += 256) {       // horizontal pass
        const int ty = e / TW, tx = e - ty * TW;
        const int x = ox0 + tx, y = oy0 + ty;
        if (x >= ow || y >= oh) continue;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) acc += t[j] * colb[ty][tx * F + j];
        if (do_norm) {
            acc = fminf(fmaxf(acc, 0.f), 1.f);
            acc = (acc - m) / sd;
        }
        out[((size_t)plane * oh + y) * ow + x] = acc;
    }
}

__global__ __launch_bounds__(256) void clamp_normalize_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ mean,
                                                              const float* __restrict__ stdv, int C, int hw) {
    const int plane = blockIdx.y, c = plane % C;
    const float m = mean[c], s = stdv[c];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        const float v = fminf(fmaxf(in[(size_t)plane * hw + i], 0.f), 1.f);
        out[(size_t)plane * hw + i] = (v - m) / s;
    }
}

static int bicubic_down_normalize(float* out, const float* in, const float* taps, const float* mean, const float* stdv, int bs, int C, int h, int w, int factor,
                                  bool pm1, void* stream) {
    E4S_REQUIRE(out && in, "bicubic_down_normalize: null tensor");
    E4S_REQUIRE(bs >= 0 && C >= 1 && (int64_t)bs * C <= 65535 && h >= 1 && w >= 1, "bicubic_down_normalize: bad size");
    if (factor == 1) {  // no resampling (input already at the parser resolution): clamp + normalise only
        E4S_REQUIRE(mean && stdv, "bicubic_down_normalize: factor 1 needs mean/std");
        E4S_REQUIRE(!pm1, "bicubic_down_normalize: the [-1, 1] input form needs factor 2 or 4");
        if (bs == 0) return 0;
        const int gx = cdiv(h * w, 256) < 256 ? cdiv(h * w, 256) : 256;
        hipLaunchKernelGGL(clamp_normalize_kernel, dim3(gx, bs * C), dim3(256), 0, (hipStream_t)stream, out, in, mean, stdv, C, h * w);
        return check_launch("clamp_normalize");
    }
    E4S_REQUIRE(taps, "bicubic_down_normalize: null taps");
    E4S_REQUIRE(factor == 2 || factor == 4, "bicubic_down_normalize: factor %d not supported (1, 2 or 4)", factor);
    E4S_REQUIRE((mean == nullptr) == (stdv == nullptr), "bicubic_down_normalize: mean and std go together");
    E4S_REQUIRE(h >= 4 * factor && w >= 4 * factor, "bicubic_down_normalize: image smaller than the filter");
    if (bs == 0) return 0;
    // conv output size with pad (K - F) and stride F: (h + K - F - K) / F + 1 = h / F  (floor)
    const int oh = (h - factor) / factor + 1, ow = (w - factor) / factor + 1;
    dim3 grid(cdiv(ow, 128 / factor), cdiv(oh, 32 / factor), bs * C);
    hipStream_t st = (hipStream_t)stream;
    const int nrm = mean ? 1 : 0;
    if (factor == 2 && pm1)
        hipLaunchKernelGGL((bicubic_down_norm_kernel<2, true>), grid, dim3(256), 0, st, out, in, taps, mean, stdv, C, h, w, oh, ow, nrm);
    else if (factor == 2)
        hipLaunchKernelGGL((bicubic_down_norm_kernel<2, false>), grid, dim3(256), 0, st, out, in, taps, mean, stdv, C, h, w, oh, ow, nrm);
    else if (pm1)
        hipLaunchKernelGGL((bicubic_down_norm_kernel<4, true>), grid, dim3(256), 0, st, out, in, taps, mean, stdv, C, h, w, oh, ow, nrm);
    else
        hipLaunchKernelGGL((bicubic_down_norm_kernel<4, false>), grid, dim3(256), 0, st, out, in, taps, mean, stdv, C, h, w, oh, ow, nrm);
    return check_launch("bicubic_down_normalize");
}

extern "C" int e4s_bicubic_down_normalize(float* out, const float* in, const float* taps, const float* mean, const float* stdv, int bs, int C,
                                          int h, int w, int factor, void* stream) {
    return bicubic_down_normalize(out, in, taps, mean, stdv, bs, C, h, w, factor, false, stream);
}

extern "C" int e4s_bicubic_down_normalize_pm1(float* out, const float* in, const float* taps, const float* mean, const float* stdv, int bs, int C,
                                              int h, int w, int factor, void* stream) {
    return bicubic_down_normalize(out, in, taps, mean, stdv, bs, C, h, w, factor, true, stream);
}

// ------------------------------------------------------------------------------------ tensor2im
// utils/torch_utils.py:64-76 of the reference on the device: ((x + 1) / 2) clamped to [0,1], * 255, TRUNCATED to uint8,
// CHW float -> HWC uint8 (what PIL / the RCCL frame gather consume: 3 MB per 1024^2 frame instead of 12.6 MB).
__global__ __launch_bounds__(256) void tensor2im_kernel(uint8_t* __restrict__ out, const float* __restrict__ img, int hw) {
    const int b = blockIdx.y;
    const float* p = img + (size_t)b * 3 * hw;
    uint8_t* o = out + (size_t)b * 3 * hw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (p[(size_t)c * hw + i] + 1.0f) / 2.0f;
            v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
            o[(size_t)i * 3 + c] = (uint8_t)(v * 255.0f);
        }
    }
}

extern "C" int e4s_tensor2im_u8(uint8_t* out, const float* img, int bs, int h, int w, void* stream) {
    E4S_REQUIRE(out && img, "tensor2im_u8: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && h >= 1 && w >= 1, "tensor2im_u8: bad size");
    if (bs == 0) return 0;
    const int gx = cdiv(h * w, 256) < 1024 ? cdiv(h * w, 256) : 1024;
    hipLaunchKernelGGL(tensor2im_kernel, dim3(gx, bs), dim3(256), 0, (hipStream_t)stream, out, img, h * w);
    return check_launch("tensor2im_u8");
}
