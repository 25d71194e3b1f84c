// What slows a [12 x v_mfma_f32_32x32x16_bf16 + ~30 VALU + 10 ds_read_b128] loop body on gfx950 with 2 waves per SIMD?
// Variants add one ingredient at a time; prints cycles per MFMA per SIMD (32 = matrix-pipe peak).  Tuning probe, not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2)); }

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint4* w = reinterpret_cast<uint4*>(lds);
    float4* x = reinterpret_cast<float4*>(lds + 73728);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < (73728 + 21760) / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float sv[8];
    for (int e = 0; e < 8; ++e) sv[e] = 1.0f + 0.001f * (lane + e);
    uint4 ah[4], al[4], bh, bl;
    for (int i = 0; i < 4; ++i) { ah[i] = w[lane + i * 64]; al[i] = w[lane + 256 + i * 64]; }
    bh = w[lane + 512]; bl = w[lane + 576];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (VARIANT & 2) {   // LDS reads: 8 weight + 2 activation vectors per tap
#pragma unroll
                for (int i = 0; i < 4; ++i) { ah[i] = w[lane + i * 32 + tap * 256]; al[i] = w[2304 + lane + i * 32 + tap * 256]; }
            }
            if (VARIANT & 1) {   // scale + split of 8 activations (the masked path's VALU work)
                float4 x0, x1;
                if (VARIANT & 2) { x0 = x[(lane + tap * 3) * 4]; x1 = x[(lane + tap * 3) * 4 + 1]; }
                else { x0 = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]); x1 = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]); }
                const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = v[2 * j] * sv[2 * j], b = v[2 * j + 1] * sv[2 * j + 1];
                    h[j] = pack(a, b);
                    l[j] = pack(a - __builtin_bit_cast(float, h[j] << 16), b - __builtin_bit_cast(float, h[j] & 0xffff0000u));
                }
                bh = make_uint4(h[0], h[1], h[2], h[3]);
                bl = make_uint4(l[0], l[1], l[2], l[3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh), acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl), acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh), acc[i], 0, 0, 0);
        }
        if (VARIANT & 4) __syncthreads();   // one workgroup barrier per 9 taps, as between K chunks
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, int waves_per_simd) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 200;
    const int threads = waves_per_simd * 256;
    hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<V>, dim3(256), dim3(threads), 98304, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<V>, dim3(256), dim3(threads), 98304, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
    m /= 256;
    // clock64 = s_memtime at 100 MHz on gfx950?  report both raw ticks and ns via hipEvent is overkill: print ticks per MFMA per SIMD
    const double tf = 1024.0 * waves_per_simd * iters * 108.0 * 65536.0 / (ms * 1e-3) / 1e12;
    printf("%-44s %d wave(s)/SIMD: %.1f clock64 ticks per MFMA per SIMD, kernel %.3f ms = %.0f TFLOP/s bf16 (%.0f%% of 2500)\n", name, waves_per_simd,
           m / (iters * 108.0 * waves_per_simd), ms, tf, tf / 25.0);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int wps : {1, 2}) {
        run<0>("MFMA only", wps);
        run<1>("MFMA + scale/split VALU", wps);
        run<2>("MFMA + LDS reads", wps);
        run<3>("MFMA + VALU + LDS reads", wps);
        run<7>("MFMA + VALU + LDS reads + barrier/9 taps", wps);
    }
    return 0;
}
