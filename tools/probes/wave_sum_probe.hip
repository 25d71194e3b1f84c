// The DPP / permlane butterfly of csrc/common.h against the __shfl_xor loop it replaced: bit for bit, on random data (including negative zeros, huge and tiny values).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../e4s2024_amd/csrc/common.h"
using namespace e4s;
__device__ __forceinline__ float ref_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ float ref_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__global__ void k32(const float* x, unsigned* out) {      // 32 values per lane: wave_sum_x32 against 32 x wave_sum
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    float v[32], ref[32];
    for (int j = 0; j < 32; ++j) { v[j] = x[(wave * 32 + j) * 64 + lane]; ref[j] = ref_sum(v[j]); }
    const float got = wave_sum_x32(v);
    const int kk = wave_sum_x32_index(lane);
    float want = 0.f;
    for (int j = 0; j < 32; ++j) if (j == kk) want = ref[j];
    out[2 * (wave * 64 + lane)] = __builtin_bit_cast(unsigned, got);
    out[2 * (wave * 64 + lane) + 1] = __builtin_bit_cast(unsigned, want);
}
__global__ void k(const float* x, unsigned* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float v = x[i];
    out[4 * i] = __builtin_bit_cast(unsigned, wave_sum(v));
    out[4 * i + 1] = __builtin_bit_cast(unsigned, ref_sum(v));
    out[4 * i + 2] = __builtin_bit_cast(unsigned, wave_max(v));
    out[4 * i + 3] = __builtin_bit_cast(unsigned, ref_max(v));
}
int main() {
    const int n = 256 * 4096;
    float* h = (float*)malloc(n * 4);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const int kind = rand() % 8;
        float v = (float)rand() / RAND_MAX * 2.f - 1.f;
        if (kind == 0) v *= 1e30f; else if (kind == 1) v *= 1e-30f; else if (kind == 2) v = -0.f; else if (kind == 3) v *= 1e6f;
        h[i] = v;
    }
    float* dx; unsigned* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 16);
    hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout);
    unsigned* ho = (unsigned*)malloc(n * 16);
    hipMemcpy(ho, dout, n * 16, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) bad += (ho[4 * i] != ho[4 * i + 1]) + (ho[4 * i + 2] != ho[4 * i + 3]);
    printf("wave_sum / wave_max butterfly vs __shfl_xor loop: %ld mismatching values of %d\n", bad, 2 * n);
    const int nw = n / 64 / 32;               // waves of the 32-value test
    hipLaunchKernelGGL(k32, dim3(nw * 64 / 256), dim3(256), 0, 0, dx, dout);
    hipMemcpy(ho, dout, (size_t)nw * 64 * 8, hipMemcpyDeviceToHost);
    long bad32 = 0;
    for (int i = 0; i < nw * 64; ++i) bad32 += ho[2 * i] != ho[2 * i + 1];
    printf("wave_sum_x32 vs 32 x __shfl_xor loop: %ld mismatching lanes of %d\n", bad32, nw * 64);
    return bad != 0 || bad32 != 0;
}
