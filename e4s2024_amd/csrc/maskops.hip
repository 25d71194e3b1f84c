// f2 / f3: the integer mask surgery that sits between face parsing and synthesis, on the device.
//   e4s_swap_head_mask   : swap_head_mask_hole_first            (swap_face_fine/swap_face_mask.py:194-333)
//   e4s_foreground_masks : foreground mask + create_masks(..., 'expansion', radius)
//                          (face_swap_video_pipeline.py:456-463, gradio_utils/face_swapping.py:203-221, utils/morphology.py:22-190)
// uint8 label maps in, uint8 / {0,1} float maps out; every value is exact.  HBM-bound byte work: one coalesced pass per map.
#include "common.h"

using namespace e4s;

namespace {

__device__ __forceinline__ bool is_bg_class(int c) { return c == 4 || c == 0 || c == 8 || c == 7 || c == 11; }   // :210-220

// One thread per column: top-most skin row of the target (rows >= 1; the reference's `target_skin * arange(H)` makes row 0
// indistinguishable from "no skin", :282-286) and the lowest rows of the source's eyes / brows / nose (:234-239).
__global__ __launch_bounds__(256) void swap_mask_scan_kernel(int* __restrict__ skin_top, int* __restrict__ lowest, const uint8_t* __restrict__ source,
                                                             const uint8_t* __restrict__ target, int h, int w) {
    const int b = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    int top = h, m3 = -1, m2 = -1, m5 = -1;
    if (x < w) {
        const uint8_t* s = source + (size_t)b * h * w + x;
        const uint8_t* t = target + (size_t)b * h * w + x;
        for (int y = 0; y < h; ++y) {
            const int sv = s[(size_t)y * w], tv = t[(size_t)y * w];
            if (tv == 6 && y >= 1 && y < top) top = y;
            if (sv == 3) m3 = y;
            if (sv == 2) m2 = y;
            if (sv == 5) m5 = y;
        }
        skin_top[(size_t)b * w + x] = top;
    }
    m3 = (int)wave_max((float)m3);   // rows < 2^24: exact in float
    m2 = (int)wave_max((float)m2);
    m5 = (int)wave_max((float)m5);
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&lowest[b * 3 + 0], m3);
        atomicMax(&lowest[b * 3 + 1], m2);
        atomicMax(&lowest[b * 3 + 2], m5);
    }
}

__global__ __launch_bounds__(256) void swap_mask_apply_kernel(uint8_t* __restrict__ res, uint8_t* __restrict__ hole, uint8_t* __restrict__ hole_map,
                                                              int* __restrict__ lines, const uint8_t* __restrict__ source,
                                                              const uint8_t* __restrict__ target, const int* __restrict__ skin_top,
                                                              const int* __restrict__ lowest, int h, int w) {
    const int b = blockIdx.y;
    const int m3 = lowest[b * 3 + 0], m2 = lowest[b * 3 + 1], m5 = lowest[b * 3 + 2];
    const int eye_line = m3 >= 0 ? m3 : (m2 >= 0 ? m2 : (int)(2.0 / 5.0 * (double)h));   // :232-237
    const int nose_line = m5 >= 0 ? m5 : (int)(3.0 / 5.0 * (double)h);                   // :233, 238-239
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        lines[b * 2 + 0] = eye_line;
        lines[b * 2 + 1] = nose_line;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const size_t o = (size_t)b * h * w + i;
    const int s = source[o], t = target[o];
    bool hl = !is_bg_class(t) && is_bg_class(s);          // target face that the source face does not cover (:222-223)
    if (y < eye_line) hl = false;                         // :243-244
    int r = 0;
    if (t == 0) r = 99;                                   // :247-251 (later assignments win, as in the reference)
    if (t == 8) r = 8;
    if (t == 7) r = 7;
    if (t == 11) r = 11;
    if (s == 1) r = 1;                                    // :263-269
    if (s == 2) r = 2;
    if (s == 4 && t == 2) r = 2;
    if (s == 3) r = 3;
    if (s == 5) r = 5;
    if (s == 6) r = 6;
    if (s == 9) r = 9;
    const int top = skin_top[(size_t)b * w + x];
    if (t == 0 && top != h && y <= top) r = 98;           // :287-301 target foreground above its skin (hats)
    if (t == 4) r = 4;                                    // :304-305
    if (t == 10) r = 10;
    if (r == 0) r = 6;                                    // :310-312
    if (r == 99 || r == 98) r = 0;
    res[o] = (uint8_t)r;
    hole[o] = hl ? 1 : 0;
    hole_map[o] = hl ? 17 : (uint8_t)r;                   // :313-314
}

// foreground = not {background, ear-ring, ear, hair, neck} or hole; then flat (2r+1)^2 dilation / erosion that ignore pixels outside the
// image ('geodesic' border).  For a {0,1} map: dilation = any foreground in the window, erosion = all in-image pixels foreground.
constexpr int FM_T = 32;
__global__ __launch_bounds__(256) void foreground_masks_kernel(float* __restrict__ content, float* __restrict__ border, float* __restrict__ full,
                                                               const uint8_t* __restrict__ swapped, const uint8_t* __restrict__ hole, int h, int w,
                                                               int radius) {
    extern __shared__ unsigned char sm[];
    const int pw = FM_T + 2 * radius;
    unsigned char* fg = sm;                   // [pw][pw]: 1 foreground, 0 not, 2 outside the image
    unsigned char* any_r = sm + pw * pw;      // [pw][FM_T] horizontal "any foreground"
    unsigned char* all_r = any_r + pw * FM_T; // [pw][FM_T] horizontal "all foreground"
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * FM_T, y0 = blockIdx.y * FM_T;
    for (int e = threadIdx.x; e < pw * pw; e += 256) {
        const int py = e / pw, px = e - py * pw;
        const int gy = y0 - radius + py, gx = x0 - radius + px;
        unsigned char v = 2;
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
            const size_t o = ((size_t)b * h + gy) * w + gx;
            const int c = swapped[o];
            const bool bgc = c == 0 || c == 11 || c == 7 || c == 4 || c == 8;
            v = (!bgc || (hole && hole[o])) ? 1 : 0;
        }
        fg[e] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < pw * FM_T; e += 256) {
        const int py = e / FM_T, lx = e - py * FM_T;
        bool any = false, all = true;
        for (int d = 0; d <= 2 * radius; ++d) {
            const unsigned char v = fg[py * pw + lx + d];
            any = any || v == 1;
            all = all && v != 0;
        }
        any_r[e] = any;
        all_r[e] = all;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < FM_T * FM_T; e += 256) {
        const int ly = e / FM_T, lx = e - ly * FM_T;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= h || gx >= w) continue;
        bool any = false, all = true;
        for (int d = 0; d <= 2 * radius; ++d) {
            any = any || any_r[(ly + d) * FM_T + lx];
            all = all && all_r[(ly + d) * FM_T + lx];
        }
        const size_t o = ((size_t)b * h + gy) * w + gx;
        const float dil = any ? 1.f : 0.f, ero = all ? 1.f : 0.f;
        if (content) content[o] = fg[(ly + radius) * pw + lx + radius] == 1 ? 1.f : 0.f;
        border[o] = fminf(fmaxf(dil - ero, 0.f), 1.f);
        full[o] = dil;
    }
}

// ---- multi-band blend (swap_face_fine/multi_band_blending.py:5-74): the two pyramid steps of cv2.pyrDown / cv2.pyrUp on float planes.
// pyrDown: 5x5 kernel [1 4 6 4 1]^2 / 256 at the even pixels, BORDER_REFLECT_101; round_u8 = the 8-bit variant's (sum + 128) >> 8.
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__global__ __launch_bounds__(256) void pyr_down_kernel(float* __restrict__ out, const float* __restrict__ in, int h, int w, int oh, int ow, int round_u8) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= oh * ow) return;
    const int oy = i / ow, ox = i - oy * ow;
    const float* ip = in + (size_t)blockIdx.y * h * w;
    const float k[5] = {1.f, 4.f, 6.f, 4.f, 1.f};
    int xs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) xs[t] = reflect101(2 * ox + t - 2, w);
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const float* row = ip + (size_t)reflect101(2 * oy + u - 2, h) * w;
        float r = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) r += k[t] * row[xs[t]];
        acc += k[u] * r;
    }
    out[(size_t)blockIdx.y * oh * ow + i] = round_u8 ? fminf(fmaxf(floorf((acc + 128.f) * (1.f / 256.f)), 0.f), 255.f) : acc * (1.f / 256.f);
}

// pyrUp weights of destination index D over source indices (i-1, i, i+1), i = D >> 1, in eighths: even D: (1, 6, 1) with s[-1] = s[1];
// odd D: (0, 4, 4); the last source pixel: even (1, 7, 0), odd (0, 8, 0); a one-pixel source: (0, 8, 0).
__device__ __forceinline__ void pyr_up_taps(int D, int n, int idx[3], float wgt[3]) {
    const int i = D >> 1;
    idx[0] = i > 0 ? i - 1 : (n > 1 ? 1 : 0);
    idx[1] = i;
    idx[2] = i + 1 < n ? i + 1 : i;
    if (n == 1) { wgt[0] = 0.f; wgt[1] = 8.f; wgt[2] = 0.f; return; }
    const bool last = i == n - 1;
    if (D & 1) { wgt[0] = 0.f; wgt[1] = last ? 8.f : 4.f; wgt[2] = last ? 0.f : 4.f; }
    else { wgt[0] = 1.f; wgt[1] = last ? 7.f : 6.f; wgt[2] = last ? 0.f : 1.f; }
}

// out = up(in), or minuend - up(in) (Laplacian level), or up(in) + addend (reconstruction)
__global__ __launch_bounds__(256) void pyr_up_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ minuend,
                                                     const float* __restrict__ addend, int h, int w) {
    const int ow = 2 * w, oh = 2 * h;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= oh * ow) return;
    const int Y = i / ow, X = i - Y * ow;
    int yi[3], xi[3];
    float yw[3], xw[3];
    pyr_up_taps(Y, h, yi, yw);
    pyr_up_taps(X, w, xi, xw);
    const float* ip = in + (size_t)blockIdx.y * h * w;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const float* row = ip + (size_t)yi[u] * w;
        acc += yw[u] * (xw[0] * row[xi[0]] + xw[1] * row[xi[1]] + xw[2] * row[xi[2]]);
    }
    acc *= (1.f / 64.f);
    const size_t o = (size_t)blockIdx.y * oh * ow + i;
    if (minuend) acc = minuend[o] - acc;
    if (addend) acc += addend[o];
    out[o] = acc;
}

// One level of the blend in one pass:  out = up(prev) + [ (a_hi - up(a_lo)) * m + (b_hi - up(b_lo)) * (1 - m) ]
// = the Laplacian levels of A and B, their mask-weighted mix (la*gm + lb*(1-gm)) and the reconstruction step (multi_band_blending.py:27-46).
__global__ __launch_bounds__(256) void pyr_blend_level_kernel(float* __restrict__ out, const float* __restrict__ prev, const float* __restrict__ a_hi,
                                                              const float* __restrict__ a_lo, const float* __restrict__ b_hi,
                                                              const float* __restrict__ b_lo, const float* __restrict__ m_hi, int h, int w) {
    const int ow = 2 * w, oh = 2 * h;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= oh * ow) return;
    const int Y = i / ow, X = i - Y * ow;
    int yi[3], xi[3];
    float yw[3], xw[3];
    pyr_up_taps(Y, h, yi, yw);
    pyr_up_taps(X, w, xi, xw);
    const size_t lo = (size_t)blockIdx.y * h * w;
    float up_p = 0.f, up_a = 0.f, up_b = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const size_t r = lo + (size_t)yi[u] * w;
        up_p += yw[u] * (xw[0] * prev[r + xi[0]] + xw[1] * prev[r + xi[1]] + xw[2] * prev[r + xi[2]]);
        up_a += yw[u] * (xw[0] * a_lo[r + xi[0]] + xw[1] * a_lo[r + xi[1]] + xw[2] * a_lo[r + xi[2]]);
        up_b += yw[u] * (xw[0] * b_lo[r + xi[0]] + xw[1] * b_lo[r + xi[1]] + xw[2] * b_lo[r + xi[2]]);
    }
    const size_t o = (size_t)blockIdx.y * oh * ow + i;
    const float la = a_hi[o] - up_a * (1.f / 64.f), lb = b_hi[o] - up_b * (1.f / 64.f);
    const float mv = m_hi[o];
    out[o] = up_p * (1.f / 64.f) + (la * mv + lb * (1.f - mv));      // the reference's form: exact at mask 0 and 1
}

// One pass of Pillow's 8-bit resampler (src/libImaging/Resample.c: ImagingResampleHorizontal_8bpc / Vertical_8bpc) over a uint8
// [bs, H, W, C] image: out = clip8((2^21 + sum_j k[o][j] * in[xmin[o] + j]) >> 22) along `axis` (1 = width, 0 = height), integer arithmetic.
__global__ __launch_bounds__(256) void resample_u8_kernel(uint8_t* __restrict__ out, const uint8_t* __restrict__ in, const int* __restrict__ xmin,
                                                          const int* __restrict__ cnt, const int* __restrict__ kk, int ksize, int h, int w, int c,
                                                          int out_size, int axis) {
    const int oh = axis == 0 ? out_size : h, ow = axis == 1 ? out_size : w;
    const int64_t n = (int64_t)oh * ow * c;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ch = (int)(i % c);
    const int x = (int)((i / c) % ow), y = (int)(i / ((int64_t)c * ow));
    const int o = axis == 1 ? x : y;
    const int lo = xmin[o], m = cnt[o];
    const int* k = kk + (size_t)o * ksize;
    const uint8_t* ip = in + (size_t)blockIdx.y * h * w * c;
    const size_t step = axis == 1 ? (size_t)c : (size_t)w * c;
    const uint8_t* src = axis == 1 ? ip + ((size_t)y * w + lo) * c + ch : ip + ((size_t)lo * w + x) * c + ch;
    int acc = 1 << 21;
    for (int j = 0; j < m; ++j) acc += k[j] * (int)src[j * step];
    acc >>= 22;
    out[(size_t)blockIdx.y * n + i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
}

}  // namespace

extern "C" int e4s_resample_u8(uint8_t* out, const uint8_t* in, const int32_t* xmin, const int32_t* count, const int32_t* coeffs, int ksize, int bs,
                               int h, int w, int c, int out_size, int axis, void* stream) {
    E4S_REQUIRE(out && in && xmin && count && coeffs, "resample_u8: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && h >= 1 && w >= 1 && c >= 1 && c <= 4 && out_size >= 1 && ksize >= 1 && (axis == 0 || axis == 1),
                "resample_u8: bad size");
    if (bs == 0) return 0;
    const int64_t n = (int64_t)(axis == 0 ? out_size : h) * (axis == 1 ? out_size : w) * c;
    E4S_REQUIRE(n < ((int64_t)1 << 31), "resample_u8: output too large");
    hipLaunchKernelGGL(resample_u8_kernel, dim3((unsigned)cdiv64(n, 256), bs), dim3(256), 0, (hipStream_t)stream, out, in, xmin, count, coeffs, ksize, h, w,
                       c, out_size, axis);
    return check_launch("resample_u8");
}

extern "C" int e4s_pyr_blend_level(float* out, const float* prev, const float* a_hi, const float* a_lo, const float* b_hi, const float* b_lo,
                                   const float* m_hi, int planes, int h, int w, void* stream) {
    E4S_REQUIRE(out && prev && a_hi && a_lo && b_hi && b_lo && m_hi, "pyr_blend_level: null tensor");
    E4S_REQUIRE(planes >= 0 && planes <= 65535 && h >= 1 && w >= 1 && (int64_t)h * w < ((int64_t)1 << 28), "pyr_blend_level: bad size");
    if (planes == 0) return 0;
    hipLaunchKernelGGL(pyr_blend_level_kernel, dim3(cdiv(4 * h * w, 256), planes), dim3(256), 0, (hipStream_t)stream, out, prev, a_hi, a_lo, b_hi, b_lo,
                       m_hi, h, w);
    return check_launch("pyr_blend_level");
}

extern "C" int e4s_pyr_down(float* out, const float* in, int planes, int h, int w, int round_u8, void* stream) {
    E4S_REQUIRE(out && in, "pyr_down: null tensor");
    E4S_REQUIRE(planes >= 0 && planes <= 65535 && h >= 1 && w >= 1 && (int64_t)h * w < ((int64_t)1 << 30), "pyr_down: bad size");
    if (planes == 0) return 0;
    const int oh = (h + 1) / 2, ow = (w + 1) / 2;
    hipLaunchKernelGGL(pyr_down_kernel, dim3(cdiv(oh * ow, 256), planes), dim3(256), 0, (hipStream_t)stream, out, in, h, w, oh, ow, round_u8);
    return check_launch("pyr_down");
}

extern "C" int e4s_pyr_up(float* out, const float* in, const float* minuend, const float* addend, int planes, int h, int w, void* stream) {
    E4S_REQUIRE(out && in, "pyr_up: null tensor");
    E4S_REQUIRE(planes >= 0 && planes <= 65535 && h >= 1 && w >= 1 && (int64_t)h * w < ((int64_t)1 << 28), "pyr_up: bad size");
    if (planes == 0) return 0;
    hipLaunchKernelGGL(pyr_up_kernel, dim3(cdiv(4 * h * w, 256), planes), dim3(256), 0, (hipStream_t)stream, out, in, minuend, addend, h, w);
    return check_launch("pyr_up");
}

extern "C" int e4s_swap_head_mask(uint8_t* res, uint8_t* hole_mask, uint8_t* hole_map, int32_t* lines, const uint8_t* source, const uint8_t* target,
                                  int32_t* scratch, int bs, int h, int w, void* stream) {
    E4S_REQUIRE(res && hole_mask && hole_map && lines && source && target && scratch, "swap_head_mask: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && h >= 1 && w >= 1 && (int64_t)h * w < ((int64_t)1 << 30), "swap_head_mask: bad size");
    if (bs == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int* lowest = scratch;            // [bs][3]
    int* skin_top = scratch + bs * 3; // [bs][w]
    hipError_t e = hipMemsetAsync(lowest, 0xff, sizeof(int) * bs * 3, st);   // -1
    if (e != hipSuccess) return fail((int)e, "swap_head_mask: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(swap_mask_scan_kernel, dim3(cdiv(w, 256), bs), dim3(256), 0, st, skin_top, lowest, source, target, h, w);
    hipLaunchKernelGGL(swap_mask_apply_kernel, dim3(cdiv(h * w, 256), bs), dim3(256), 0, st, res, hole_mask, hole_map, lines, source, target, skin_top,
                       lowest, h, w);
    return check_launch("swap_head_mask");
}

extern "C" int e4s_foreground_masks(float* content, float* border, float* full, const uint8_t* swapped, const uint8_t* hole_mask, int bs, int h, int w,
                                    int radius, void* stream) {
    E4S_REQUIRE(border && full && swapped, "foreground_masks: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && h >= 1 && w >= 1 && radius >= 0 && radius <= 32, "foreground_masks: bad size (radius 0..32)");
    if (bs == 0) return 0;
    const int pw = FM_T + 2 * radius;
    const size_t lds = (size_t)pw * pw + 2 * (size_t)pw * FM_T;
    hipLaunchKernelGGL(foreground_masks_kernel, dim3(cdiv(w, FM_T), cdiv(h, FM_T), bs), dim3(256), lds, (hipStream_t)stream, content, border, full, swapped,
                       hole_mask, h, w, radius);
    return check_launch("foreground_masks");
}


// ============================================================================ f3: uint8 frames -> network input
// transforms.Compose([ToTensor(), Normalize((.5,.5,.5), (.5,.5,.5))]) (datasets/dataset.py:32, 45; face_swap_video_pipeline.py:338-339):
// out[b][c][y][x] = (in[b][y][x][c] / 255 - 0.5) / 0.5 in float32 (a true division, as torchvision's ToTensor does).
__global__ __launch_bounds__(256) void frames_to_tensor_kernel(float* __restrict__ out, const uint8_t* __restrict__ in, int bs, int hw) {
    const int64_t total = (int64_t)bs * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / hw);
        const int pix = (int)(i - (int64_t)b * hw);
        const uint8_t* q = in + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((int64_t)b * 3 + c) * hw + pix] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)q[c], 255.f), 0.5f), 0.5f);
    }
}

extern "C" int e4s_frames_to_tensor(float* out, const uint8_t* frames_u8, int bs, int h, int w, void* stream) {
    E4S_REQUIRE(out && frames_u8, "frames_to_tensor: null tensor");
    E4S_REQUIRE(bs >= 0 && h >= 1 && w >= 1, "frames_to_tensor: bad size");
    if (bs == 0) return 0;
    const int64_t total = (int64_t)bs * h * w;
    const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
    hipLaunchKernelGGL(frames_to_tensor_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, frames_u8, bs, h * w);
    return check_launch("frames_to_tensor");
}

// ============================================================================ f1: erode_mask of the PTI loop
// erode_mask(mask, img, radius) (training/video_swap_ft_coach.py:64-93): face = not {background 0, hair 4, ear-rings 11}; the face mask is
// eroded with a (2r+1)^2 box (cv2.erode, BORDER_CONSTANT 0: pixels outside the image count as not-face); out = mask where the eroded face
// mask holds, 0 elsewhere.  A flat erosion is a logical AND over the window — no arithmetic to pin beyond its definition.
__global__ __launch_bounds__(256) void erode_labels_kernel(uint8_t* __restrict__ out, const uint8_t* __restrict__ in, int h, int w, int radius, unsigned bg_bits) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const uint8_t* m = in + (size_t)blockIdx.z * h * w;
    bool keep = true;
    for (int dy = -radius; dy <= radius && keep; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) { keep = false; break; }
        for (int dx = -radius; dx <= radius; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w) { keep = false; break; }
            const unsigned c = m[(size_t)yy * w + xx];
            if (c < 32 && ((bg_bits >> c) & 1u)) { keep = false; break; }
        }
    }
    out[(size_t)blockIdx.z * h * w + (size_t)y * w + x] = keep ? m[(size_t)y * w + x] : (uint8_t)0;
}

extern "C" int e4s_erode_labels(uint8_t* out, const uint8_t* labels, int bs, int h, int w, int radius, unsigned bg_class_bits, void* stream) {
    E4S_REQUIRE(out && labels && out != labels, "erode_labels: null tensor / in-place");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && h >= 1 && w >= 1 && radius >= 0 && radius <= 64, "erode_labels: bad size");
    if (bs == 0) return 0;
    hipLaunchKernelGGL(erode_labels_kernel, dim3(cdiv(w, 32), cdiv(h, 8), bs), dim3(256), 0, (hipStream_t)stream, out, labels, h, w, radius, bg_class_bits);
    return check_launch("erode_labels");
}
