// Split-bf16 helpers shared by the modulated-conv kernels (modconv_sb.hip, modconv_upfused.hip).
#pragma once
#include "common.h"

namespace e4s {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int CKS = 16;  // input channels per K chunk (= one 16-deep MFMA step per tap)

__device__ __forceinline__ unsigned pack_bf16_rne(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));  // v_cvt_pk_bf16_f32
}
// (t0, t1) -> packed hi pair, packed lo pair
__device__ __forceinline__ void split2(float t0, float t1, unsigned& hi, unsigned& lo) {
    hi = pack_bf16_rne(t0, t1);
    const float h0 = __builtin_bit_cast(float, hi << 16);
    const float h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16_rne(t0 - h0, t1 - h1);
}

// LDS-DMA with explicit address spaces: a SCALAR 64-bit global base + a 32-bit per-lane byte offset (the saddr form of global_load_lds_*:
// one VGPR, no 64-bit address arithmetic), and the LDS destination as a plain 32-bit LDS address (M0).  Written with generic pointers,
// hipcc 7.2 wraps every DMA in null checks of the generic -> LDS conversion, and a per-lane choice between two source pointers (image /
// zero block) becomes two exec-masked DMAs with a GOT load in between: ~190 cycles per instruction instead of ~10.
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef const __attribute__((address_space(1))) unsigned char gl_byte;
__device__ __forceinline__ void dma16(const void* gbase, unsigned voff, lds_byte* dst) {
    __builtin_amdgcn_global_load_lds((gl_byte*)gbase + voff, dst, 16, 0, 0);      // (C-style cast: an address-space cast)
}
__device__ __forceinline__ void dma4(const void* gbase, unsigned voff, lds_byte* dst) {
    __builtin_amdgcn_global_load_lds((gl_byte*)gbase + voff, dst, 4, 0, 0);
}
// The same requests issued from inline asm.  Through the builtin hipcc 7.2 treats an LDS-DMA as a store every later ds_read may alias and puts
// `s_waitcnt vmcnt(0)` in front of the next LDS read: a ring of stages then degenerates (the chunk just requested has to land before the CURRENT
// chunk's operands are read — up_fused_dma_kernel's K loop ran that way through round 3).  From asm the compiler neither orders LDS reads behind
// the request nor counts it; the kernel waits by itself (E4S_WAIT_VM + E4S_LDS_BARRIER).  `lds_dst` must be wave-uniform (it goes to M0).
__device__ __forceinline__ void dma16_asm(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    // (s_nop 2: with the two s_mov in front of it, five wait states between a VALU write of the base SGPRs — v_readlane of a spilled SGPR right in front of the block, which
    //  hipcc's hazard pass cannot pair with an instruction inside it — and the vector-memory read of them; round 5)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void dma4_asm(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
// s_waitcnt vmcnt(n) with a literal n, and a workgroup barrier that leaves LDS-DMA in flight (a __syncthreads() would drain it: its fence waits
// vmcnt(0) while a DMA is pending)
// Tile walk of the persistent kernels (one workgroup per CU, workgroup b of G takes tiles off(b), off(b) + G, ...).
// XCD-aware start offset: block b runs on XCD b % 8 (observed, speed only) and each XCD has its own L2, so the G / 8 workgroups of one XCD take G / 8 CONSECUTIVE
// tiles of the enumeration per step (off = (b % 8) * G / 8 + b / 8) instead of every eighth one: with the band-blocked enumeration below, what neighbouring tiles share
// (a patch's halo rows / columns) is then fetched into that L2 once.  A pure permutation of the start offsets: every count of the walk stays as it was.
__device__ __forceinline__ int walk_offset(int b, int G, int xcd_aware) {
    return (xcd_aware && (G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
}
// Band-blocked enumeration of an image's tiles_x x tiles_y tiles: bands of `bh` tile rows, column-major inside a band — 32 consecutive indices are a 4 x 8 block of
// tiles (bh = 8) instead of a row segment.  r = index inside the image; bh <= 1: row-major.
__device__ __forceinline__ void walk_tile_xy(int r, int tiles_x, int tiles_y, int bh, int& ty, int& tx) {
    if (bh <= 1) { ty = r / tiles_x; tx = r - ty * tiles_x; return; }
    const int band = r / (bh * tiles_x);
    const int rr = r - band * bh * tiles_x;
    const int rows = tiles_y - band * bh < bh ? tiles_y - band * bh : bh;      // (a ragged last band)
    tx = rr / rows;
    ty = band * bh + (rr - tx * rows);
}

#define E4S_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define E4S_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

}  // namespace e4s
