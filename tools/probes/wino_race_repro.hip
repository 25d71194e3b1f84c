// Minimal reproducer attempt for the round-2 fault of the FIRST pre-split Winograd input transform (single pass, fully unrolled: 112 loads in flight before
// the first use): occasional wrong values — one register, sixteen lanes — only when another stream's kernels ran beside it.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/wino_race_repro.hip -o tools/probes/wino_race_repro && tools/probes/wino_race_repro [iterations]
// The transform (kernel text as of commit 3af0289) runs alone for the reference, then `iterations` times beside a bandwidth-hungry kernel on a second stream;
// every output word is compared with the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ unsigned pack_bf16_rne(float a, float b) {
    unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    ua += 0x7fffu + ((ua >> 16) & 1u);
    ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
}
__device__ __forceinline__ void split2(float t0, float t1, unsigned& hi, unsigned& lo) {
    hi = pack_bf16_rne(t0, t1);
    const float h0 = __builtin_bit_cast(float, hi << 16);
    const float h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16_rne(t0 - h0, t1 - h1);
}

__global__ __launch_bounds__(256) void wino_input_pre_kernel(uint4* __restrict__ Vhi, uint4* __restrict__ Vlo, const float* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, int bs, int C, int H, int W) {
    const int th = H >> 1, tw = W >> 1;
    const int T = bs * th * tw;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c8 = blockIdx.y;
    if (t >= T) return;
    const int tx = t % tw, ty = (t / tw) % th, b = t / (tw * th);
    unsigned oh[16][4], ol[16][4];
    float prev[16];
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
        float e[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[0][j] = d[0][j] - d[2][j];
            e[1][j] = d[1][j] + d[2][j];
            e[2][j] = d[2][j] - d[1][j];
            e[3][j] = d[1][j] - d[3][j];
        }
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[4 * i + 0] = e[i][0] - e[i][2];
            v[4 * i + 1] = e[i][1] + e[i][2];
            v[4 * i + 2] = e[i][2] - e[i][1];
            v[4 * i + 3] = e[i][1] - e[i][3];
        }
        if (cj & 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) split2(prev[k], v[k], oh[k][cj >> 1], ol[k][cj >> 1]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) prev[k] = v[k];
        }
    }
    const size_t plane = (size_t)gridDim.y * T;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        Vhi[(size_t)k * plane + (size_t)c8 * T + t] = make_uint4(oh[k][0], oh[k][1], oh[k][2], oh[k][3]);
        Vlo[(size_t)k * plane + (size_t)c8 * T + t] = make_uint4(ol[k][0], ol[k][1], ol[k][2], ol[k][3]);
    }
}

// the neighbour: streams a large buffer (read + write) so that memory latency under the transform grows
__global__ __launch_bounds__(256) void hog_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n, int rounds) {
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            float4 v = src[i];
            v.x += 1.f;
            dst[i] = v;
        }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;      // neighbour on the second stream: 0 = bandwidth hog, 1 = the same transform on other buffers, 2 = both
    const int bs = 8, C = 512, H = 32, W = 32, T = bs * (H / 2) * (W / 2);
    const size_t nx = (size_t)bs * C * H * W, nv = (size_t)16 * (C / 8) * T;     // uint4 elements per V plane set
    std::vector<float> hx(nx), hm((size_t)bs * C), hr((size_t)bs * C);
    srand(1);
    for (auto& v : hx) v = (float)(rand() & 0xffffff) / 16777216.f * 4.f - 2.f;
    for (auto& v : hm) v = (float)(rand() & 0xffffff) / 16777216.f - 0.5f;
    for (auto& v : hr) v = (float)(rand() & 0xffffff) / 16777216.f + 0.5f;
    float *x, *mean, *rstd;
    uint4 *vh, *vl, *rh, *rl, *wh, *wl;
    float4 *ha, *hb;
    const size_t nh = (size_t)64 << 20;      // 1 GiB of float4
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&mean, hm.size() * 4)); CK(hipMalloc(&rstd, hr.size() * 4));
    CK(hipMalloc(&vh, nv * 16)); CK(hipMalloc(&vl, nv * 16)); CK(hipMalloc(&rh, nv * 16)); CK(hipMalloc(&rl, nv * 16));
    CK(hipMalloc(&ha, nh * 16)); CK(hipMalloc(&hb, nh * 16)); CK(hipMalloc(&wh, nv * 16)); CK(hipMalloc(&wl, nv * 16));
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(mean, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rstd, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(ha, 0, nh * 16));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    const dim3 grid((T + 255) / 256, C / 8);
    hipLaunchKernelGGL(wino_input_pre_kernel, grid, dim3(256), 0, s1, rh, rl, x, mean, rstd, bs, C, H, W);
    CK(hipStreamSynchronize(s1));
    std::vector<uint4> ref_h(nv), ref_l(nv), out_h(nv), out_l(nv);
    CK(hipMemcpy(ref_h.data(), rh, nv * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ref_l.data(), rl, nv * 16, hipMemcpyDeviceToHost));
    long bad_iters = 0, bad_words = 0;
    for (int it = 0; it < iters; ++it) {
        CK(hipMemsetAsync(vh, 0xff, nv * 16, s1));
        CK(hipMemsetAsync(vl, 0xff, nv * 16, s1));
        if (mode != 1) hipLaunchKernelGGL(hog_kernel, dim3(1024), dim3(256), 0, s2, hb, ha, nh / 8, 1);
        if (mode != 0) for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(wino_input_pre_kernel, grid, dim3(256), 0, s2, wh, wl, x, mean, rstd, bs, C, H, W);
        for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(wino_input_pre_kernel, grid, dim3(256), 0, s1, vh, vl, x, mean, rstd, bs, C, H, W);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out_h.data(), vh, nv * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(out_l.data(), vl, nv * 16, hipMemcpyDeviceToHost));
        long w = 0;
        const unsigned *a = (const unsigned*)out_h.data(), *b = (const unsigned*)ref_h.data(), *c = (const unsigned*)out_l.data(), *d = (const unsigned*)ref_l.data();
        for (size_t i = 0; i < nv * 4; ++i) w += (a[i] != b[i]) + (c[i] != d[i]);
        if (w) { ++bad_iters; bad_words += w; if (bad_iters <= 5) printf("iteration %d: %ld wrong words\n", it, w); }
    }
    printf("%d iterations beside a second stream: %ld with wrong values (%ld words)\n", iters, bad_iters, bad_words);
    return bad_iters ? 1 : 0;
}
