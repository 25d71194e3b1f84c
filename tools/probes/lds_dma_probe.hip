// Does loading activation tiles straight into LDS (global_load_lds_dword: no destination registers, so several chunks can be in flight)
// beat the register-staged one-chunk prefetch of the synthesis kernels?  Same tile geometry as tile_read_probe (32 x 8 pixels + halo,
// 16 channels per chunk, channels-first), a stand-in "compute" phase of ~0.4 us per chunk, 256-thread workgroups.  Tuning probe.
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int TW = 32, TH = 8, PW = TW + 2, PH = TH + 2, PATCH = PW * PH, NT = 256, CKS = 16, EPT = (PATCH + NT - 1) / NT;
constexpr int SLOTS = EPT * NT;   // patch elements rounded up to whole waves (512)

__device__ __forceinline__ float fake_compute(const float* buf, int tid, int iters) {
    float a = 0.f;
    for (int i = 0; i < iters; ++i) a = __builtin_fmaf(buf[(tid + 37 * i) & (SLOTS * CKS - 1)], 1.0001f, a);
    return a;
}

// register staging, one chunk of prefetch (the product kernels' scheme)
__global__ __launch_bounds__(NT) void staged(float* out, const float* x, int C, int H, int W, int iters) {
    __shared__ float buf[CKS * SLOTS];
    const int tx = blockIdx.x % (W / TW), ty = blockIdx.x / (W / TW), b = blockIdx.y, tid = threadIdx.x;
    const float* xb = x + (size_t)b * C * H * W;
    int goff[EPT];
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * NT, py = e / PW, px = e - py * PW;
        int gy = ty * TH - 1 + py, gx = tx * TW - 1 + px;
        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy); gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        goff[j] = e < PATCH ? gy * W + gx : 0;
    }
    float r[CKS][EPT], acc = 0.f;
    const int nchunk = C / CKS;
#pragma unroll
    for (int c = 0; c < CKS; ++c)
#pragma unroll
        for (int j = 0; j < EPT; ++j) r[c][j] = xb[(size_t)c * H * W + goff[j]];
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CKS; ++c)
#pragma unroll
            for (int j = 0; j < EPT; ++j) buf[c * SLOTS + tid + j * NT] = r[c][j];
        __syncthreads();
        if (ch + 1 < nchunk) {
#pragma unroll
            for (int c = 0; c < CKS; ++c)
#pragma unroll
                for (int j = 0; j < EPT; ++j) r[c][j] = xb[(size_t)((ch + 1) * CKS + c) * H * W + goff[j]];
        }
        acc += fake_compute(buf, tid, iters);
    }
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NT + tid] = acc;
}

// straight to LDS, DEPTH chunks in flight (ring of DEPTH buffers)
template <int DEPTH>
__global__ __launch_bounds__(NT) void direct(float* out, const float* x, int C, int H, int W, int iters) {
    extern __shared__ float ring[];   // [DEPTH][CKS][SLOTS]
    const int tx = blockIdx.x % (W / TW), ty = blockIdx.x / (W / TW), b = blockIdx.y, tid = threadIdx.x;
    const float* xb = x + (size_t)b * C * H * W;
    int goff[EPT];
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * NT, py = e / PW, px = e - py * PW;
        int gy = ty * TH - 1 + py, gx = tx * TW - 1 + px;
        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy); gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        goff[j] = e < PATCH ? gy * W + gx : 0;
    }
    const int nchunk = C / CKS;
    const int wbase = tid & ~63;
    auto issue = [&](int ch) {
        float* dst = ring + (size_t)(ch % DEPTH) * CKS * SLOTS;
#pragma unroll
        for (int c = 0; c < CKS; ++c)
#pragma unroll
            for (int j = 0; j < EPT; ++j)
                __builtin_amdgcn_global_load_lds(xb + (size_t)(ch * CKS + c) * H * W + goff[j], dst + c * SLOTS + j * NT + wbase, 4, 0, 0);
    };
    float acc = 0.f;
    for (int ch = 0; ch < DEPTH && ch < nchunk; ++ch) issue(ch);
    for (int ch = 0; ch < nchunk; ++ch) {
        // wait until chunk ch has landed: at most (DEPTH-1) younger chunks may stay in flight
        const int younger = (nchunk - 1 - ch) < (DEPTH - 1) ? (nchunk - 1 - ch) : (DEPTH - 1);
        if (younger == 0) __builtin_amdgcn_s_waitcnt(0x0f70);                       // vmcnt(0)
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CKS * EPT));
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CKS * EPT > 63 ? 63 : 2 * CKS * EPT));   // the counter has 6 bits
        __syncthreads();
        acc += fake_compute(ring + (size_t)(ch % DEPTH) * CKS * SLOTS, tid, iters);
        __syncthreads();
        if (ch + DEPTH < nchunk) issue(ch + DEPTH);
    }
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NT + tid] = acc;
}

int main() {
    const int bs = 4, iters = 96;
    const int cases[3][2] = {{32, 1024}, {64, 512}, {128, 256}};
    for (auto& cs : cases) {
        const int C = cs[0], H = cs[1], W = cs[1];
        const size_t n = (size_t)bs * C * H * W;
        float *x, *out;
        (void)hipMalloc(&x, n * 4); (void)hipMalloc(&out, (size_t)bs * (H / TH) * (W / TW) * NT * 4);
        (void)hipMemset(x, 0, n * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const dim3 grid((H / TH) * (W / TW), bs);
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            const size_t lds = (size_t)(mode ? mode : 1) * CKS * SLOTS * 4;
            if (mode == 2) (void)hipFuncSetAttribute((const void*)direct<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (mode == 3) (void)hipFuncSetAttribute((const void*)direct<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(staged, grid, dim3(NT), 0, 0, out, x, C, H, W, iters);
                if (mode == 1) hipLaunchKernelGGL(direct<1>, grid, dim3(NT), lds, 0, out, x, C, H, W, iters);
                if (mode == 2) hipLaunchKernelGGL(direct<2>, grid, dim3(NT), lds, 0, out, x, C, H, W, iters);
                if (mode == 3) hipLaunchKernelGGL(direct<3>, grid, dim3(NT), lds, 0, out, x, C, H, W, iters);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("%3d ch @ %4d^2 bs %d: %-44s %.3f ms = %.2f TB/s (%zu KB of LDS per workgroup)\n", C, H, bs,
                   mode == 0 ? "registers, one chunk ahead (product scheme)" : (mode == 1 ? "straight to LDS, 1 chunk in flight" : (mode == 2 ? "straight to LDS, 2 chunks in flight" : "straight to LDS, 3 chunks in flight")),
                   best, n * 4 / (best * 1e-3) / 1e12, lds / 1024);
        }
        (void)hipFree(x); (void)hipFree(out);
    }
    return 0;
}
