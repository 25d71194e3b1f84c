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
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave64 sum via DPP-lowered shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// PyTorch 'nearest' source index: floor(dst * scale) clamped (ATen nearest_neighbor_compute_source_index)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

}  // namespace e4s
