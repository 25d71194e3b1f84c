// Shared helpers for libe4s_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/e4s_hip.h"

namespace e4s {

// Thread-local last-error text (returned by e4s_last_error()).
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

#define E4S_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::e4s::fail(E4S_ERR_ARG, __VA_ARGS__); \
    } while (0)

static inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// an odd stride near n / golden ratio that is coprime with n: i -> (i * stride) % n is a permutation of 0..n-1 that sends neighbours far apart
static inline unsigned coprime_stride(unsigned n) {
    if (n < 3) return 1u;
    unsigned s = (unsigned)(n * 0.6180339887) | 1u;
    auto gcd = [](unsigned a, unsigned b) { while (b) { const unsigned t = a % b; a = b; b = t; } return a; };
    while (gcd(s, n) != 1u) s += 2u;
    return s % n ? s % n : 1u;
}
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave64 butterfly reductions: v[l] (op) v[l ^ 32], then ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 — every lane ends with the result.  The partner values come over the VALU's own
// lane-permute paths (v_permlane32_swap / v_permlane16_swap for the two row exchanges, DPP row_ror:8, row_half_mirror + quad_perm[3,2,1,0] = l ^ 4, quad_perm for ^ 2 and
// ^ 1): the same operands in the same order as the __shfl_xor loop these replace — bit-identical sums — without its six ds_bpermute round trips through the LDS pipeline
// per reduction (hipcc lowers every __shfl_xor to ds_bpermute_b32: 240 of them in the style-table kernel, which they bounded).
template <int CTRL>
__device__ __forceinline__ float e4s_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <class Op>
__device__ __forceinline__ float wave_butterfly(float v, Op op) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // [rows 0 1 0 1], [rows 2 3 2 3]
    v = op(__builtin_bit_cast(float, (unsigned)r32[0]), __builtin_bit_cast(float, (unsigned)r32[1]));
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto r16 = __builtin_amdgcn_permlane16_swap(w, w, false, false);      // [rows 0 0 2 2], [rows 1 1 3 3]
    v = op(__builtin_bit_cast(float, (unsigned)r16[0]), __builtin_bit_cast(float, (unsigned)r16[1]));
    v = op(v, e4s_dpp<0x128>(v));                                          // row_ror:8           = l ^ 8
    v = op(v, e4s_dpp<0x1B>(e4s_dpp<0x141>(v)));                           // row_half_mirror (l ^ 7), quad_perm [3,2,1,0] (^ 3) = l ^ 4
    v = op(v, e4s_dpp<0x4E>(v));                                           // quad_perm [2,3,0,1] = l ^ 2
    v = op(v, e4s_dpp<0xB1>(v));                                           // quad_perm [1,0,3,2] = l ^ 1
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_butterfly(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_max(float v) { return wave_butterfly(v, [](float a, float b) { return fmaxf(a, b); }); }

// 32 wave sums at once.  v[k] (k = 0 .. 31) are 32 independent per-lane values; on return lane l holds, in the return value, the wave sum of v[k(l)] with
// k(l) = bits 5..1 of l read as (b5 b4 b3 b2 b1) — every value ends up in one lane pair.  At step `xor o` of the butterfly each lane keeps the half of its values
// whose index bit matches its own lane bit and hands the other half to its partner, so the number of live values halves with every step: 31 exchanges + 31 additions
// for all 32 sums, where 32 separate butterflies cost 192 of each.  Per value the additions are those of wave_sum, in its order: bit-identical.
__device__ __forceinline__ float wave_sum_x32(float (&v)[32]) {
    const int lane = threadIdx.x & 63;
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {            // xor 32: lanes < 32 go on with value k, lanes >= 32 with value k + 16
        auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[k]), __builtin_bit_cast(unsigned, v[k + 16]), false, false);
        a[k] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    float b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {             // xor 16: rows 0 / 2 go on with k, rows 1 / 3 with k + 8
        auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a[k]), __builtin_bit_cast(unsigned, a[k + 8]), false, false);
        b[k] = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {             // xor 8
        const float keep = b3 ? b[k + 4] : b[k], send = b3 ? b[k] : b[k + 4];
        c[k] = keep + e4s_dpp<0x128>(send);
    }
    float d[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {             // xor 4
        const float keep = b2 ? c[k + 2] : c[k], send = b2 ? c[k] : c[k + 2];
        d[k] = keep + e4s_dpp<0x1B>(e4s_dpp<0x141>(send));
    }
    const float keep = b1 ? d[1] : d[0], send = b1 ? d[0] : d[1];      // xor 2
    const float e = keep + e4s_dpp<0x4E>(send);
    return e + e4s_dpp<0xB1>(e);                                        // xor 1
}
// the value index lane l's result of wave_sum_x32 belongs to
__device__ __forceinline__ int wave_sum_x32_index(int lane) { return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1); }

// PyTorch 'nearest' source index: floor(dst * scale) clamped (ATen nearest_neighbor_compute_source_index)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// Phase timestamps for kernel tuning (tools/phase_prof.py).  Only in the -DE4S_PHASE_PROF build (lib/libe4s_hip_prof.so); the
// product library compiles these to nothing.
#ifdef E4S_PHASE_PROF
#define E4S_PROF_SLOTS 8
#define E4S_PROF_BLOCKS (1 << 17)
#define E4S_PROF_DECL(buf) __device__ long long buf[(size_t)E4S_PROF_BLOCKS * E4S_PROF_SLOTS];
#define E4S_PROF_MARK(buf, slot)                                                                                     \
    do {                                                                                                             \
        if (threadIdx.x == 0) {                                                                                      \
            const size_t lin_ = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);      \
            if (lin_ < E4S_PROF_BLOCKS) {                                                                           \
                buf[lin_ * E4S_PROF_SLOTS + (slot)] = (long long)wall_clock64();                                     \
                if ((slot) == 0)   /* HW_ID (reg 4) | XCC_ID (reg 20) << 32: which CU this workgroup landed on */      \
                    buf[lin_ * E4S_PROF_SLOTS + 7] = (long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |  \
                                                     ((long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32); \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define E4S_PROF_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#else
#define E4S_PROF_DECL(buf)
#define E4S_PROF_MARK(buf, slot) do { } while (0)
#define E4S_PROF_DRAIN() do { } while (0)
#endif

}  // namespace e4s
