// The masked kernel's tap loop in isolation (same LDS layout, same addresses, same instruction mix as region_modconv_sb_kernel<4,1,1,8,5>,
// 512 threads, 96 KB of LDS, no global traffic inside the loop): which ingredient keeps the matrix pipe at ~50 %?  Tuning probe.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2)); }
__device__ __forceinline__ void split2(float t0, float t1, unsigned& hi, unsigned& lo) {
    hi = pack(t0, t1);
    lo = pack(t0 - __builtin_bit_cast(float, hi << 16), t1 - __builtin_bit_cast(float, hi & 0xffff0000u));
}

constexpr int TN = 128, PW = 34, PATCH = 10 * 34, W4 = 2 * 9 * 2 * TN;

// MODE bit 0: scale + split B on the fly (masked path) instead of reading pre-split planes;  bit 1: barriers between chunks (2 per chunk);
// bit 2: also re-write the LDS stage every chunk (ds_write traffic of store_chunk, from registers)
// MODE bit 5: the real kernel's global prefetch: per chunk 16 dword loads (activations, 16 channels of a 34-wide patch row set) and
// 9 dwordx4 loads (weight slabs) per thread, issued before the taps and written to LDS at the next chunk boundary.
template <int MODE, int KEEP = 0>
__global__ __launch_bounds__(512, 2) void probe(float* out, int chunks, const float* __restrict__ gx = nullptr, const uint4* __restrict__ gw = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* wsm = reinterpret_cast<uint4*>(lds_raw);
    float4* xf4 = reinterpret_cast<float4*>(lds_raw + W4 * 16);
    uint4* xh4 = reinterpret_cast<uint4*>(lds_raw + W4 * 16);
    uint4* xl4 = xh4 + 2 * PATCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    for (int i = tid; i < (W4 * 16 + PATCH * 64) / 16; i += 512) reinterpret_cast<uint4*>(lds_raw)[i] = make_uint4(0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f00u);
    __syncthreads();
    const int xoff = wave * PW + l5;          // wave = pixel row of the 8 x 32 tile
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float sv[8];
    for (int e = 0; e < 8; ++e) sv[e] = 1.0f + 0.001f * (lane + e);
    uint4 wreg[9];
    for (int v = 0; v < 9; ++v) wreg[v] = make_uint4(0x3f803f80u + v, 0x3f003f00u, 0x3e803e80u, 0x3f803f00u);
    float keep[KEEP > 0 ? KEEP : 1];   // registers held live across the loop, like the real kernel's prefetch stage
#pragma unroll
    for (int k = 0; k < KEEP; ++k) keep[k] = out[threadIdx.x + k * 512];
    float xr[16];
    unsigned wr[9][4];
    const int hw = 64 * 64;
    const int goff = ((blockIdx.x & 15) * 8 * 64 + (tid / 34) * 64 + (tid % 34)) & (hw - 1);
    if (MODE & 64) { gx += (size_t)(blockIdx.x >> 6) * 512 * hw; gw += (size_t)((blockIdx.x >> 4) & 3) * 32 * 4608; }   // 4 samples, 4 output-channel tiles
    if (MODE & 32) {
#pragma unroll
        for (int c = 0; c < 16; ++c) xr[c] = gx[(size_t)c * hw + goff];
#pragma unroll
        for (int v = 0; v < 9; ++v) { const uint4 t = gw[tid + v * 512]; wr[v][0] = t.x; wr[v][1] = t.y; wr[v][2] = t.z; wr[v][3] = t.w; }
    }
    for (int chunk = 0; chunk < chunks; ++chunk) {
        if (KEEP > 0) {
#pragma unroll
            for (int k = 0; k < KEEP; ++k) asm volatile("" : "+v"(keep[k]));
        }
        if (MODE & 2) __syncthreads();
        if (MODE & 32) {
#pragma unroll
            for (int v = 0; v < 9; ++v) wsm[tid + v * 512] = make_uint4(wr[v][0], wr[v][1], wr[v][2], wr[v][3]);
            if (tid < PATCH) {
                const int g = (tid >> 2) & 3;
#pragma unroll
                for (int k = 0; k < 4; ++k) xf4[tid * 4 + (k ^ g)] = make_float4(xr[4 * k], xr[4 * k + 1], xr[4 * k + 2], xr[4 * k + 3]);
            }
        } else if (MODE & 4) {
#pragma unroll
            for (int v = 0; v < 9; ++v) wsm[tid + v * 512] = wreg[v];
            if (tid < PATCH) {
                const int g = (tid >> 2) & 3;
#pragma unroll
                for (int k = 0; k < 4; ++k) xf4[tid * 4 + (k ^ g)] = make_float4(sv[0], sv[1], sv[2], sv[3]);
            }
        }
        if (MODE & 2) __syncthreads();
        if (MODE & 32) {   // prefetch of the next chunk (32 chunks of 16 channels, then wrap)
            const int cn = (chunk + 1) & 31;
#pragma unroll
            for (int c = 0; c < 16; ++c) xr[c] = gx[(size_t)(cn * 16 + c) * hw + goff];
#pragma unroll
            for (int v = 0; v < 9; ++v) { const uint4 t = gw[(size_t)cn * 4608 + tid + v * 512]; wr[v][0] = t.x; wr[v][1] = t.y; wr[v][2] = t.z; wr[v][3] = t.w; }
        }
        const uint4* whalf = wsm + khalf * TN + l5;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if ((MODE & 8) && tap > 0 && tap % 3 == 0) __builtin_amdgcn_s_barrier();     // extra lockstep points inside the chunk
            if ((MODE & 16) && tap > 0) __builtin_amdgcn_s_barrier();
            const int e = xoff + (tap / 3) * PW + (tap % 3);
            uint4 bh, bl;
            if (MODE & 1) {
                const int g = (e >> 2) & 3;
                const float4 x0 = xf4[e * 4 + ((2 * khalf) ^ g)], x1 = xf4[e * 4 + ((2 * khalf + 1) ^ g)];
                split2(x0.x * sv[0], x0.y * sv[1], bh.x, bl.x);
                split2(x0.z * sv[2], x0.w * sv[3], bh.y, bl.y);
                split2(x1.x * sv[4], x1.y * sv[5], bh.z, bl.z);
                split2(x1.z * sv[6], x1.w * sv[7], bh.w, bl.w);
            } else {
                const int slot = e * 2 + (khalf ^ ((e >> 3) & 1));
                bh = xh4[slot];
                bl = xl4[slot];
            }
            uint4 ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { ah[i] = whalf[tap * 2 * TN + i * 32]; al[i] = whalf[18 * TN + tap * 2 * TN + i * 32]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh), acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl), acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh), acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KEEP; ++k) s += keep[k];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + tid] = s;
}

static int g_chunks = 512;
template <int MODE, int KEEP = 0>
void run(const char* name) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    float* gx; uint4* gw;
    const size_t nx = (size_t)4 * 512 * 64 * 64, nw = (size_t)4 * 32 * 4608 * 4;
    hipMalloc(&gx, nx * 4); hipMalloc(&gw, nw * 4);
    if (MODE & 128) {   // random operands (activations ~N(0,1)-ish floats, weights = random bf16 pairs) instead of zeros: data toggling costs power
        float* hx = (float*)malloc(nx * 4); unsigned* hwt = (unsigned*)malloc(nw * 4);
        unsigned st = 12345u;
        for (size_t i = 0; i < nx; ++i) { st ^= st << 13; st ^= st >> 17; st ^= st << 5; hx[i] = ((int)(st & 0xffff) - 32768) / 16384.0f; }
        for (size_t i = 0; i < nw; ++i) { st ^= st << 13; st ^= st >> 17; st ^= st << 5; hwt[i] = (st & 0x807f807fu) | 0x3c003c00u; }
        hipMemcpy(gx, hx, nx * 4, hipMemcpyHostToDevice); hipMemcpy(gw, hwt, nw * 4, hipMemcpyHostToDevice);
        free(hx); free(hwt);
    } else { hipMemset(gx, 0, nx * 4); hipMemset(gw, 0, nw * 4); }
    const int chunks = g_chunks, lds = W4 * 16 + PATCH * 64;
    hipFuncSetAttribute((const void*)probe<MODE, KEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, KEEP>), dim3(256), dim3(512), lds, 0, out, chunks, gx, gw);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, KEEP>), dim3(256), dim3(512), lds, 0, out, chunks, gx, gw);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 1024.0 * 2 * chunks * 108.0;
    printf("%-62s %.3f ms  %.2f us/chunk  %.0f TFLOP/s bf16 (%.0f %% of 2500)\n", name, ms, ms * 1e3 / chunks, mfma * 32768.0 / (ms * 1e-3) / 1e12,
           mfma * 32768.0 / (ms * 1e-3) / 1e12 / 25.0);
    hipFree(out); hipFree(gx); hipFree(gw);
}

int main(int argc, char** argv) {
    if (argc > 1) {   // sustained-load check: the realistic variant for many chunks, repeated
        g_chunks = atoi(argv[1]);
        for (int rep = 0; rep < 2; ++rep) run<3 + 32>("on the fly, barriers, GLOBAL prefetch + restage (sustained)");
        for (int rep = 0; rep < 2; ++rep) run<3 + 32 + 64>("same, 4 samples x 4 output-channel tiles (9.4 MB of weight slabs, 33 MB of activations)");
        for (int rep = 0; rep < 2; ++rep) run<3 + 32 + 64 + 128>("same with RANDOM operand data");
        run<2 + 32 + 64>("pre-split B (no VALU in the loop), zeros");
        run<2 + 32 + 64 + 128>("pre-split B (no VALU in the loop), RANDOM operand data");
        return 0;
    }
    run<0>("pre-split B from LDS, no barriers");
    run<2>("pre-split B, 2 barriers per chunk");
    run<1>("scale+split B on the fly, no barriers");
    run<3>("scale+split B on the fly, 2 barriers per chunk");
    run<7>("scale+split on the fly, barriers, LDS stage re-written per chunk");
    run<6>("pre-split B, barriers, LDS stage re-written per chunk");
    run<7 + 8>("on the fly, barriers, restage, + s_barrier every 3 taps");
    run<7 + 16>("on the fly, barriers, restage, + s_barrier every tap");
    run<1 + 8>("on the fly, NO chunk barriers, s_barrier every 3 taps");
    run<3 + 32>("on the fly, barriers, GLOBAL prefetch + restage (as the real kernel)");
    run<2 + 32>("pre-split B, barriers, GLOBAL prefetch + restage");
    run<7, 32>("on the fly, barriers, restage, 32 extra live registers");
    run<7, 64>("on the fly, barriers, restage, 64 extra live registers");
    run<7, 96>("on the fly, barriers, restage, 96 extra live registers");
    return 0;
}
