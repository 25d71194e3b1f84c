// How fast can 256-thread workgroups read a [bs, C, H, W] fp32 activation tile by tile (32 x 8 pixels + 1-pixel halo, 16 channels per
// chunk — the staging pattern of the synthesis kernels) in channels-first vs channels-last layout?  Pure read throughput: every loaded
// value is summed, one float per thread is written.  Tuning probe.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int TW = 32, TH = 8, PW = TW + 2, PH = TH + 2, PATCH = PW * PH, NT = 256, CKS = 16;

// channels-first with other tile shapes of the same 256 pixels (wider rows waste less of each 128-byte line on the halo)
template <int W_, int H_>
__global__ __launch_bounds__(NT) void read_nchw_shape(float* out, const float* x, int C, int H, int W) {
    constexpr int pw = W_ + 2, ph = H_ + 2;
    const int tx = blockIdx.x % (W / W_), ty = blockIdx.x / (W / W_), b = blockIdx.y;
    const float* xb = x + (size_t)b * C * H * W;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += CKS) {
        for (int e = threadIdx.x; e < pw * ph; e += NT) {
            const int py = e / pw, px = e - py * pw;
            int gy = ty * H_ - 1 + py, gx = tx * W_ - 1 + px;
            gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy); gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
#pragma unroll
            for (int c = 0; c < CKS; ++c) acc += xb[(size_t)(c0 + c) * H * W + gy * W + gx];
        }
    }
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NT + threadIdx.x] = acc;
}

// channels-first: thread <-> patch pixel, 16 dword loads per pixel and chunk (one per channel plane)
// xcd != 0: workgroups are dealt round-robin to the 8 XCDs (each with its own L2); renumber so that every XCD walks one contiguous band of tiles
__device__ __forceinline__ int tile_of(int id, int ntile, int xcd) {
    if (!xcd || ntile % 8) return id;
    return (id % 8) * (ntile / 8) + id / 8;
}

__global__ __launch_bounds__(NT) void read_nchw(float* out, const float* x, int C, int H, int W, int depth, int xcd) {
    const int t = tile_of(blockIdx.x, gridDim.x, xcd);
    const int tx = t % (W / TW), ty = t / (W / TW), b = blockIdx.y;
    const float* xb = x + (size_t)b * C * H * W;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += CKS * depth) {
        for (int e = threadIdx.x; e < PATCH; e += NT) {
            const int py = e / PW, px = e - py * PW;
            int gy = ty * TH - 1 + py, gx = tx * TW - 1 + px;
            gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy); gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
#pragma unroll
            for (int c = 0; c < CKS; ++c)
                for (int dd = 0; dd < depth; ++dd) acc += xb[(size_t)(c0 + dd * CKS + c) * H * W + gy * W + gx];
        }
    }
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NT + threadIdx.x] = acc;
}

// channels-last: thread <-> (patch pixel, 16-byte quarter of its 64-byte chunk): consecutive lanes read consecutive 16-byte pieces
__global__ __launch_bounds__(NT) void read_nhwc(float* out, const float* x, int C, int H, int W, int depth, int xcd) {
    const int t = tile_of(blockIdx.x, gridDim.x, xcd);
    const int tx = t % (W / TW), ty = t / (W / TW), b = blockIdx.y;
    const float* xb = x + (size_t)b * C * H * W;
    float acc = 0.f;
    const int q4 = CKS * depth / 4;   // float4 per pixel and load phase
    for (int c0 = 0; c0 < C; c0 += CKS * depth) {
        for (int e = threadIdx.x; e < PATCH * q4; e += NT) {
            const int pix = e / q4, q = e - pix * q4;
            const int py = pix / PW, px = pix - py * PW;
            int gy = ty * TH - 1 + py, gx = tx * TW - 1 + px;
            gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy); gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
            const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)gy * W + gx) * C + c0 + 4 * q);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NT + threadIdx.x] = acc;
}

int main() {
    const int bs = 4;
    const int cases[3][2] = {{32, 1024}, {64, 512}, {128, 256}};
    for (auto& cs : cases) {
        const int C = cs[0], H = cs[1], W = cs[1];
        const size_t n = (size_t)bs * C * H * W;
        float *x, *out;
        hipMalloc(&x, n * 4); hipMalloc(&out, (size_t)bs * (H / TH) * (W / TW) * NT * 4);
        hipMemset(x, 0, n * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const dim3 grid((H / TH) * (W / TW), bs);
        for (int xcd = 0; xcd < 2; ++xcd)
        for (int depth = 1; depth <= 2; ++depth)
            for (int layout = 0; layout < 2; ++layout) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0, 0);
                    if (layout == 0) hipLaunchKernelGGL(read_nchw, grid, dim3(NT), 0, 0, out, x, C, H, W, depth, xcd);
                    else hipLaunchKernelGGL(read_nhwc, grid, dim3(NT), 0, 0, out, x, C, H, W, depth, xcd);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("%s%3d ch @ %4d^2 bs %d, %s, %d channels per load phase: %.3f ms = %.2f TB/s of algorithmic bytes (%.0f MB)\n", xcd ? "XCD-banded " : "linear     ", C, H, bs,
                       layout ? "channels-last " : "channels-first", CKS * depth, best, n * 4 / (best * 1e-3) / 1e12, n * 4 / 1e6);
            }
        for (int shape = 0; shape < 3; ++shape) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                if (shape == 0) hipLaunchKernelGGL((read_nchw_shape<64, 4>), dim3((H / 4) * (W / 64), bs), dim3(NT), 0, 0, out, x, C, H, W);
                if (shape == 1) hipLaunchKernelGGL((read_nchw_shape<128, 2>), dim3((H / 2) * (W / 128), bs), dim3(NT), 0, 0, out, x, C, H, W);
                if (shape == 2) hipLaunchKernelGGL((read_nchw_shape<16, 16>), dim3((H / 16) * (W / 16), bs), dim3(NT), 0, 0, out, x, C, H, W);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("%3d ch @ %4d^2 bs %d, channels-first, tile %s: %.3f ms = %.2f TB/s of algorithmic bytes\n", C, H, bs,
                   shape == 0 ? "64 x 4" : (shape == 1 ? "128 x 2" : "16 x 16"), best, n * 4 / (best * 1e-3) / 1e12);
        }
        hipFree(x); hipFree(out);
    }
    return 0;
}
