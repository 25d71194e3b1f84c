// LDS read throughput per CU on gfx950: conflict-free ds_read_b128 / b64 / b32 streams from 4 or 8 waves.  Tuning probe.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int WIDTH>   // bytes per lane per read
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 16; i += blockDim.x) reinterpret_cast<uint4*>(lds)[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    float acc = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // volatile asm reads so that nothing is hoisted or merged; 16 conflict-free reads in flight, then one wait
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const unsigned base = (unsigned)(size_t)lds + lane * WIDTH;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (WIDTH == 16) { u4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base + r * 1024)); asm volatile("" :: "v"(v)); }
            if (WIDTH == 8)  { u2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(base + r * 512)); asm volatile("" :: "v"(v)); }
            if (WIDTH == 4)  { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(base + r * 256)); asm volatile("" :: "v"(v)); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W>
void run(int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<W>, dim3(256), dim3(threads), 65536, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<W>, dim3(256), dim3(threads), 65536, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    const double bytes = (double)threads * W * 16.0 * iters;   // per CU
    printf("ds_read_b%-3d %d waves/CU: %.1f B per clock64 tick per CU, %.2f TB/s aggregate (kernel %.3f ms)\n", W * 8, threads / 64, bytes / m,
           bytes * 256 / (ms * 1e-3) / 1e12, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int t : {256, 512}) { run<16>(t); run<8>(t); run<4>(t); }
    return 0;
}
