// a9, round 4: the face parser's ResNet stem — Conv2d(3, 64, 7, stride 2, pad 3) + BatchNorm (folded) + ReLU (swap_face_fine/face_parsing/resnet.py:57-58, 66) — as an
// implicit GEMM whose K axis is the FLATTENED (channel, ky, kx) index: K = 147, padded to 160 = ten 16-deep MFMA steps (8 % padding).  The tap-per-chunk kernels of
// conv.hip pad the 3 channels to a 16-channel chunk per tap (49 steps, 13 of every 16 K values zero) and ran the layer on the exact-fp32 MFMA: 403 us for the sixteen
// 512 x 512 images of a swap batch.  Arithmetic: the two-term f16 split of the parser's other convolutions (conv.hip NS = 4: a1 b1 + a1 b2 + a2 b1, ~2^-23 per product,
// weights pre-scaled by a power of two); the input is the normalised image (|x| < 3): no range hazard.
// One workgroup = 32 x 4 output pixels x all 64 channels; wave w = output row w (32 pixels) x 2 blocks of 32 channels.  The 13 x 69 input patch of each channel is split
// into f16 (hi, lo) planes in LDS; a lane builds its B fragment — 8 consecutive K values of its pixel — from 8 two-byte LDS reads per plane at compile-time offsets
// (the lane's K half selects between two); the prepared weights [plane][K step][K half][co] x 16 B (40 KB, the same for every workgroup) are read from global memory.
#include <stdlib.h>

#include "common.h"

using namespace e4s;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TW = 32, TH = 4;
constexpr int PW = 2 * TW + 5, PH = 2 * TH + 5;       // 69 x 13 input patch per channel
constexpr int PS = 72;                                 // row stride of the patch planes (halves)
constexpr int KTOT = 147, KSTEPS = 10;
constexpr int W4 = 2 * KSTEPS * 2 * 64;                // uint4: [plane][K step][K half][co]
constexpr int PLANE_H = 3 * PH * PS;                   // halves per patch plane (2 808)
constexpr int LDS_BYTES = 2 * PLANE_H * 2;             // 11 232: the patch planes only — the 40 KB of weights are read from global memory (every workgroup the same bytes:
                                                       // L1 / L2 hits), which leaves room for eight workgroups per CU; staged in LDS they allowed three, and the kernel is a chain of
                                                       // load -> barrier -> 60 MFMAs -> store per workgroup that only occupancy hides (229 -> see DESIGN section 8 item 4)

struct Stem7Params {
    float* out;
    const float* x;
    const uint4* w;
    const float* bias;
    int bs, h, w_, ho, wo, tiles_x, relu;
    float out_scale;
};

// patch offset (halves) of flattened K index k = (c * 7 + ky) * 7 + kx; the padding steps read offset 0 (finite values against zero weights)
__host__ __device__ constexpr int koff(int k) { return k < KTOT ? ((k / 49) * PH + (k % 49) / 7) * PS + (k % 7) : 0; }

__global__ __launch_bounds__(256) void stem7_kernel(const Stem7Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint4* __restrict__ wsm = p.w;
    unsigned short* xh = reinterpret_cast<unsigned short*>(lds_raw);
    unsigned short* xl = xh + PLANE_H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int oy0 = (tile / p.tiles_x) * TH, ox0 = (tile % p.tiles_x) * TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;

    const float* xb = p.x + (size_t)b * 3 * p.h * p.w_;
    for (int e = tid; e < 3 * PH * PW; e += 256) {
        const int c = e / (PH * PW), r = e - c * (PH * PW);
        const int py = r / PW, px = r - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        const float v = (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_) ? xb[((size_t)c * p.h + gy) * p.w_ + gx] : 0.f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        xh[(c * PH + py) * PS + px] = __builtin_bit_cast(unsigned short, hi);
        xl[(c * PH + py) * PS + px] = __builtin_bit_cast(unsigned short, lo);
    }
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int base = (2 * wave) * PS + 2 * l5;           // patch origin of this lane's output pixel (row wave, column l5 of the tile)
    const unsigned short* bh_p = xh + base;
    const unsigned short* bl_p = xl + base;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        unsigned bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ka = ks * 16 + 2 * j, kb = ks * 16 + 8 + 2 * j;      // this lane's K values 2 j, 2 j + 1 in K half 0 / 1
            const int o0 = khalf ? koff(kb) : koff(ka), o1 = khalf ? koff(kb + 1) : koff(ka + 1);
            bh[j] = (unsigned)bh_p[o0] | ((unsigned)bh_p[o1] << 16);
            bl[j] = (unsigned)bl_p[o0] | ((unsigned)bl_p[o1] << 16);
        }
        const uint4 bhv = make_uint4(bh[0], bh[1], bh[2], bh[3]), blv = make_uint4(bl[0], bl[1], bl[2], bl[3]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const uint4 ah = wsm[((0 * KSTEPS + ks) * 2 + khalf) * 64 + cb * 32 + l5];
            const uint4 al = wsm[((1 * KSTEPS + ks) * 2 + khalf) * 64 + cb * 32 + l5];
            // products in order of magnitude, as conv.hip's NS = 4: (w1, a1) (w1, a2) (w2, a1)
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bhv), acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, blv), acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bhv), acc[cb], 0, 0, 0);
        }
    }
    const int oy = oy0 + wave, ox = ox0 + l5;
    if (oy >= p.ho || ox >= p.wo) return;
    const size_t ohw = (size_t)p.ho * p.wo;
    float* op = p.out + (size_t)b * 64 * ohw + (size_t)oy * p.wo + ox;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            float v = acc[cb][r] * p.out_scale + (p.bias ? p.bias[n] : 0.f);
            if (p.relu) v = fmaxf(v, 0.f);
            op[(size_t)n * ohw] = v;
        }
}

}  // namespace

// out [bs, 64, ho, wo] = act(conv2d(x [bs, 3, h, w], W, stride 2, pad 3) * 2^-wscale_log2 + bias), ho = (h - 1) / 2 + 1.
// w: 2 x 10 x 2 x 64 x 8 f16 = the two f16 terms of W * 2^wscale_log2 (BatchNorm folded) as [term][K step][K half][co][8], K = (c * 7 + ky) * 7 + kx zero-padded to 160
// (ops.PreparedConv builds it); bias [64] or NULL.
extern "C" int e4s_conv7x7s2_stem_f16x3(float* out, const float* x, const void* w, const float* bias, int bs, int h, int wd, int relu, int wscale_log2, void* stream) {
    E4S_REQUIRE(out && x && w, "conv7x7s2_stem_f16x3: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && h >= 1 && wd >= 1 && wscale_log2 >= -40 && wscale_log2 <= 40, "conv7x7s2_stem_f16x3: bad size");
    E4S_REQUIRE(((uintptr_t)w & 15) == 0, "conv7x7s2_stem_f16x3: the weights must be 16-byte aligned");
    if (bs == 0) return 0;
    Stem7Params p;
    p.out = out; p.x = x; p.w = reinterpret_cast<const uint4*>(w); p.bias = bias;
    p.bs = bs; p.h = h; p.w_ = wd; p.ho = (h + 6 - 7) / 2 + 1; p.wo = (wd + 6 - 7) / 2 + 1;
    p.tiles_x = cdiv(p.wo, TW); p.relu = relu ? 1 : 0; p.out_scale = ldexpf(1.f, -wscale_log2);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return fail((int)attr, "conv7x7s2_stem_f16x3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    dim3 grid(p.tiles_x * cdiv(p.ho, TH), bs);
    hipLaunchKernelGGL(stem7_kernel, grid, dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    return check_launch("conv7x7s2_stem_f16x3");
}
