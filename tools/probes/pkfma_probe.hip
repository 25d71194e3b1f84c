// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (is the packed form really two FMAs per lane at the scalar form's issue cost?)
// hipcc -O3 --offload-arch=gfx950 tools/probes/pkfma_probe.hip -o tools/probes/pkfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float s, int iters) {
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = (f32x2){(float)threadIdx.x + i, 1.f};
    f32x2 kk = {s, s * 0.5f};
    f32x2 zz = {s * 0.25f, s * 0.125f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { a[i][0] = __builtin_fmaf(a[i][0], kk[0], zz[0]); }
            else if (MODE == 1) { a[i] = __builtin_elementwise_fma(a[i], kk, zz); }
            else { a[i] = __builtin_elementwise_fma(a[i], __builtin_shufflevector(kk, kk, 1, 1), zz); }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE>
float run(float* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, 1.0001f, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, 1.0001f, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    const int iters = 20000;
    const double n = 1024.0 * 4 * 8 * iters;    // wave-instructions
    float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters);
    // 1024 SIMDs
    printf("v_fma_f32      : %.3f ms  %.2f cycles/instr/SIMD at 2.4 GHz\n", t0, t0 * 1e-3 * 2.4e9 / (n / 1024));
    printf("v_pk_fma_f32   : %.3f ms  %.2f\n", t1, t1 * 1e-3 * 2.4e9 / (n / 1024));
    printf("v_pk_fma op_sel: %.3f ms  %.2f\n", t2, t2 * 1e-3 * 2.4e9 / (n / 1024));
    return 0;
}
