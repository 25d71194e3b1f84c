// Which wave tile lets the matrix pipe run?  The 3x3 tap loop in isolation (operands resident in LDS, no global traffic, no conversions): a workgroup of NW
// waves, every wave MB x NB blocks of v_mfma_f32_32x32x16_f16 per (tap, 16-channel K step): MB weight fragments + NB activation fragments read from LDS
// per MB * NB MFMAs.  One workgroup per CU (LDS request > 80 KB).  Prints TFLOP/s of the MFMAs issued and the share of the register-only loop's rate.
// Tuning probe (hipcc --offload-arch=gfx950 -O3 tools/probes/wavetile_probe.hip -o tools/probes/wavetile_probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TN = 128, PST = 352, PW = 34;

// MODE bit 0: operands from registers (no LDS reads: the matrix pipe's own rate under this instruction stream); bit 1: a barrier per kernel row (3 per chunk)
template <int NW, int MB, int NB, int MODE>
__global__ __launch_bounds__(NW * 64) void probe(float* out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* wsm = reinterpret_cast<uint4*>(lds_raw);                          // [ring 2][tap 9][k half 2][co 128] x 16 B = 2 x 36 864
    uint4* xsm = reinterpret_cast<uint4*>(lds_raw + 2 * 9 * 2 * TN * 16);    // [ring 2][k half 2][pixel 352] x 16 B   = 2 x 11 264
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    for (int i = tid; i < (2 * 9 * 2 * TN + 2 * 2 * PST); i += NW * 64) reinterpret_cast<uint4*>(lds_raw)[i] = make_uint4(0x3c003c00u, 0x38003800u, 0x34003400u, 0x3c003800u);
    __syncthreads();
    constexpr int CO_W = (MB * 32 >= TN) ? 1 : TN / (MB * 32);              // waves along the output channels
    const int wco = wave % CO_W, wpx = wave / CO_W;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    uint4 areg[MB], breg[NB];
#pragma unroll
    for (int i = 0; i < MB; ++i) areg[i] = make_uint4(0x3c003c00u + i, 0x38003800u, 0x34003400u, 0x3c003800u + lane);
#pragma unroll
    for (int j = 0; j < NB; ++j) breg[j] = make_uint4(0x3c003c00u + j, 0x38003800u + lane, 0x34003400u, 0x3c003800u);
#pragma unroll 1
    for (int chunk = 0; chunk < chunks; ++chunk) {
        const uint4* wv = wsm + (chunk & 1) * 9 * 2 * TN + khalf * TN + wco * MB * 32 + l5;
        const uint4* xv = xsm + (chunk & 1) * 2 * PST + khalf * PST + (wpx * NB) % 8 * PW + l5;
#pragma unroll
        for (int row = 0; row < 3; ++row) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                uint4 a[MB], b[NB];
                if constexpr (MODE & 1) {
#pragma unroll
                    for (int i = 0; i < MB; ++i) { a[i] = areg[i]; asm volatile("" : "+v"(a[i].x)); }
#pragma unroll
                    for (int j = 0; j < NB; ++j) { b[j] = breg[j]; asm volatile("" : "+v"(b[j].x)); }
                } else {
#pragma unroll
                    for (int i = 0; i < MB; ++i) a[i] = wv[(row * 3 + t) * 2 * TN + i * 32];
#pragma unroll
                    for (int j = 0; j < NB; ++j) b[j] = xv[(j % 8) * PW + row * PW + t];
                }
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
            }
            if constexpr (MODE & 2) __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * NW * 64 + tid] = s;
}

template <int NW, int MB, int NB, int MODE>
double run(float* out, int chunks, const char* name) {
    constexpr int lds = 2 * 9 * 2 * TN * 16 + 2 * 2 * PST * 16 + 16;       // 96 272 B: one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NW, MB, NB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL((probe<NW, MB, NB, MODE>), dim3(grid), dim3(NW * 64), lds, 0, out, chunks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((probe<NW, MB, NB, MODE>), dim3(grid), dim3(NW * 64), lds, 0, out, chunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double flop = (double)grid * NW * chunks * 9.0 * MB * NB * 2.0 * 32 * 32 * 16;
    const double tf = flop / (ms * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, tf);
    return tf;
}

#define CASE(NW, MB, NB) { \
    const double r = run<NW, MB, NB, 1>(out, chunks, #NW " waves, " #MB "x" #NB " blocks: registers only"); \
    const double l = run<NW, MB, NB, 0>(out, chunks, #NW " waves, " #MB "x" #NB " blocks: LDS operands"); \
    const double bq = run<NW, MB, NB, 2>(out, chunks, #NW " waves, " #MB "x" #NB " blocks: LDS + 3 barriers/chunk"); \
    printf("    -> LDS %.2f, LDS + barriers %.2f of the register-only rate\n", l / r, bq / r); }

int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 2048;
    float* out;
    hipMalloc(&out, 256 * 512 * 4 * 4);
    CASE(8, 4, 1)      // today's masked kernel: 32 px x 128 co per wave
    CASE(8, 2, 2)      // conv_mx3: 64 px x 64 co
    CASE(4, 4, 2)      // 64 px x 128 co, one wave per SIMD
    CASE(8, 4, 2)      // the same, two waves per SIMD (tile 512 px)
    CASE(4, 4, 4)      // 128 px x 128 co, one wave per SIMD (256 accumulators)
    CASE(4, 2, 4)      // 128 px x 64 co
    return 0;
}
