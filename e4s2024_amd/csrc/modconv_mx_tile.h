// The workgroup tile of the DMA-fed masked / plain-convolution kernel (csrc/modconv_mx.hip: design, operand layouts and measurements are described there) as a device
// function, with the constants and helpers it needs: included by modconv_mx.hip (the kernel proper) and modconv_mx4.hip (which runs it for the tiles of a masked up
// layer that its own four-parity path does not take).  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <stdlib.h>

#include "common.h"
#include "sb_common.h"
#include "modconv_sb.h"

using namespace e4s;

// Tuning builds only (-DMX_ABL=bits, never the product library; results are then meaningless; tools/build_abl.sh): 1 = the plain-convolution f16 + fp6 K loop without
// its MFMAs, 2 = without its LDS operand reads (operands from registers), 4 = without barriers / waits / DMA.  What is left tells which part bounds the loop.
// The MASKED f16 + fp6 loop: 8 = without the per-tap modulate / split / range VALU work (raw patch words as operands), 16 = without the fp6 half (conversions, their
// MFMAs and operand reads), 32 = without the weight refills, waits and barriers, 64 = without the activation prefetch and its LDS store, 128 = without the f16
// weight-fragment reads, 256 = without the f16 MFMAs, 512 = the K loop runs no chunk at all (prologue + epilogue), 1024 = no epilogue.
#ifndef MX_ABL
#define MX_ABL 0
#endif

namespace {


typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int MX_TN = 128;                                    // output channels per workgroup
constexpr int MX_ROWB0 = 2 * 3 * 2 * MX_TN * 16;              // 24 576
constexpr int MX_W1B = 3 * 2 * MX_TN * 16;                    // 12 288: the f16 part of an ARITH 1 row
constexpr int MX_F6LO = 2 * 2 * MX_TN * 16;                   // 8 192
constexpr int MX_F6HI = 2 * 2 * MX_TN * 8;                    // 4 096
constexpr int MX_SCB = 2 * MX_TN * 4;                         // 1 024
constexpr int MX_ROWB1 = MX_W1B + MX_F6LO + MX_F6HI + MX_SCB; // 25 600
__host__ __device__ constexpr int mx_rowb(int arith) { return arith ? MX_ROWB1 : MX_ROWB0; }

__device__ __forceinline__ unsigned pack_f16_rne(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));      // v_cvt_pk_f16_f32
}

// (LDS-DMA is issued from inline asm — dma16_asm, sb_common.h: through the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of the next LDS read, i.e. the row just
// requested would have to land before the NEXT row's first operand read; hipcc's own counted waits for the ordinary loads of the activation patch only become longer,
// never shorter, by uncounted requests: vmcnt retires in order.)

// f16 pair of the split's residuals: lo = f16(xa * sa - a1.lo), hi = f16(xb * sb - a1.hi), each ONE v_fma_mix instruction (fp32 operands and an f16 addend, the
// product not rounded before the subtraction).  Written out through the compiler the same pair is two conversions back to fp32, a packed subtraction and a packed
// conversion — 6 instructions per two values where this is 4 with the product and a1 (hipcc 7.2 does not form the mix instructions from fma(fpext) here).
__device__ __forceinline__ unsigned resid_pair_f16(float xa, float sa, float xb, float sb, unsigned a1) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(xa), "v"(sa), "v"(a1));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(xb), "v"(sb), "v"(a1));
    return r;
}

// "This value exists HERE": an empty asm that claims to rewrite the register(s).  The two-phase loop needs it — LLVM sinks pure conversions to their first
// use, i.e. across the barrier into the phase that must hold nothing but MFMAs (and keeps their 32 source registers alive across it).
template <typename T>
__device__ __forceinline__ void pin_here(T& v) { asm volatile("" : "+v"(v)); }

// fp6 operands of the MX MFMA: six registers of codes in an eight-register tuple whose last two are never read (left undefined: shufflevector index -1)
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i32x8 mx_op6(u32x6 c) {
    typedef unsigned u32x8v __attribute__((ext_vector_type(8)));
    const u32x8v w = __builtin_shufflevector(c, c, 0, 1, 2, 3, 4, 5, -1, -1);
    return __builtin_bit_cast(i32x8, w);
}
__device__ __forceinline__ i32x8 mx_op6(uint4 lo, uint2 hi) {
    const i32x4v a = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w};
    const i32x2v b2 = {(int)hi.x, (int)hi.y};
    const i32x4v b = __builtin_shufflevector(b2, b2, 0, 1, -1, -1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, -1, -1);
}

using C = SbCfg<4, 1, 1, 8, 5>;     // 128 co x (32 x 8) px, 512 threads; wave w = tile row w, its 32 pixels x all 128 output channels
constexpr int MX_PSTRIDE = 352;                               // pixels per slot row (the 340 of the patch, padded)
constexpr int MX_PATCHB = 4 * MX_PSTRIDE * 16;                // 22 528: [16-B slot 4][pixel 352] (a slot = 4 fp32 channels, or 8 f16: plain mode) — slot-major, so that
                                                              // every fragment read of a lane is ONE base register + an immediate and 16 consecutive lanes read 256 consecutive bytes
constexpr int MX_SSB = E4S_MAX_REGIONS * CKS * 4;             // 1 024
template <int ARITH>
struct MxLds {
    static constexpr int ROWB = ARITH ? MX_ROWB1 : MX_ROWB0;
    static constexpr int RING = 3 * ROWB;
    static constexpr int PATCH0 = RING;
    static constexpr int SS0 = PATCH0 + 2 * MX_PATCHB;
    static constexpr int BYTES = SS0 + 2 * MX_SSB;
    static constexpr int NPIECE = ROWB / 1024;                // 1 KB per wave instruction
    static_assert(ROWB % 1024 == 0 && BYTES <= 160 * 1024, "LDS plan");
    static_assert((E4S_MAX_REGIONS + 5) * MX_TN * 4 + 64 <= RING, "the epilogue's tables overlay the weight ring");
};

constexpr int MX_NORM_MAX_CIN = 1024;
constexpr int MX_NORM_BYTES = 2 * MX_NORM_MAX_CIN * 4;        // plain-convolution mode: [mean | rstd] x MX_NORM_MAX_CIN floats behind the regular LDS plan
static_assert(MxLds<0>::BYTES + MX_NORM_BYTES <= 160 * 1024 && MxLds<1>::BYTES + MX_NORM_BYTES <= 160 * 1024, "LDS plan of the plain-convolution mode");

// ENC = plain-convolution mode (the regional-style encoder's stride-1 3x3 convolutions, helpers.py:122-144): no region map and no modulation;
// instance-norm statistics are applied to the input while it is staged ((x - mean) * rstd, padding stays exactly 0), the epilogue is an optional PReLU.
// One workgroup tile of the kernel: K slice `ks` of output parity `par` (0 on same-resolution layers) of the 32 x 8 tile `tile`, output channels 128 cotile .. + 127, image b.
// (A device function so that the four-parity up kernel of modconv_mx4.hip can run it for the tiles it does not take itself.)
template <int ARITH, bool RGB, bool OSP, bool ENC = false>
__device__ __forceinline__ void mx_tile_body(const SbParams& p, unsigned char* lds_raw, const int tile, const int par, const int ks, const int cotile, const int b) {
    static_assert(!ENC || (!RGB && !OSP), "plain-convolution mode has its own epilogue");
    using L = MxLds<ARITH>;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: DMA destinations (M0) and the piece loop stay in SGPRs
    const int l5 = lane & 31, khalf = lane >> 5;

    const int pa = par >> 1, pb_ = par & 1;
    const int y0 = (tile / p.tiles_x) * C::TH, x0 = (tile % p.tiles_x) * C::TW;
    const int co0 = cotile * MX_TN;
    const int hw = p.h * p.w;
    const int ho = p.up ? 2 * p.h : p.h, wo = p.up ? 2 * p.w : p.w;
    const int nchunk = (p.cin + CKS - 1) / CKS;
    const int ncot = (p.cout + MX_TN - 1) / MX_TN;

    unsigned ub_skip = 0;       // (masked up layer) bit j: 16 x 16 output block j of this tile belongs to the block kernel
    if (!ENC && p.up && p.uni_blocks && p.uni_ctrl[2] != 0) {
        const int nbx = wo >> 4, nby = ho >> 4;
        const int by = (2 * y0) >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bxk = ((2 * x0) >> 4) + j;
            const bool uni = by < nby && bxk < nbx && p.uni_blocks[((size_t)b * nby + by) * nbx + bxk] != 255;
            const bool outside = by >= nby || bxk >= nbx;
            if (uni || outside) ub_skip |= 1u << j;
        }
        if (ub_skip == 0xfu) return;
    }

    // (a thread's patch pixel: threads 0..339, see store_x) its region / patch position as an output pixel (lane l5 of tile row `wave`)
    const float* xb = p.x + (size_t)b * p.cin * hw;
    const float* sb = p.s + (size_t)b * p.nreg * p.cin;

    int cls[1], xoff;
    {
        const int ty = wave, tx = l5;
        xoff = ty * C::PW + tx;
        const int y = y0 + ty, x = x0 + tx;
        int c = ENC ? 0 : E4S_LABEL_NONE;
        if (!ENC && y < p.h && x < p.w) {
            const int oy = p.up ? 2 * y + pa : y, ox = p.up ? 2 * x + pb_ : x;
            c = p.labels[((size_t)b * p.lh + nearest_src(oy, p.lscale_y, p.lh)) * p.lw + nearest_src(ox, p.lscale_x, p.lw)];
        }
        cls[0] = (c < p.nreg) ? c : -1;
    }

    f32x16 acc[4][1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    // ---- staging: each patch thread loads its pixel's 16 channels at the chunk's start (registers through the chunk) and writes them to the other patch buffer in
    // row 2.  Round 3 measured two LDS-DMA alternatives (sum of the seven masked launches of a step / the plain 512 -> 512 @32^2 x 16 launch; registers: 1.63 / 0.172 ms):
    //   12 global_load_lds_dword requests per wave (64 pixels of one channel) into a raw fp32 buffer, converted from there in row 2:      1.71 / 0.174 ms
    //   masked: 11 requests per wave of 16 pixels x 4 channels, per-lane addresses, straight into the fp32 patch (no conversion pass):    1.89 ms
    // — 4-byte DMA requests are no cheaper than 4-byte loads, and the conversion pass or the scattered lanes cost more than the 17 registers.
    float xr[CKS];
    float sr = 0.f;
    // this thread's patch pixel (threads 0..339)
    const int ppy = tid / C::PW, ppx = tid - ppy * C::PW;
    const int pgy = y0 - 1 + ppy, pgx = x0 - 1 + ppx;
    const bool p_in = tid < C::PATCH && pgy >= 0 && pgy < p.h && pgx >= 0 && pgx < p.w;
    const int goffs = p_in ? pgy * p.w + pgx : 0;
    const int s_r = tid / CKS < p.nreg ? tid / CKS : p.nreg - 1, s_c = tid % CKS;
    auto load_x = [&](int chunk) __attribute__((always_inline)) {       // next chunk's patch pixel (16 channels) and modulation table entry
        // Unconditional, branch-free loads (as modconv_sb.hip learned): a load under a per-lane condition, or a register that is also written by a plain
        // move (`sr = cond ? load : 0`), makes hipcc wait vmcnt(0) right behind the issue — the whole latency exposed once per chunk.  Out-of-range
        // lanes read a clamped valid address and are zeroed when the chunk is written to LDS; waves 6 and 7 own no patch pixel (wave-uniform branch).
        const int ci0 = chunk * CKS;
        const int cmax = p.cin - 1 - ci0;
        if (wave < (C::PATCH + 63) / 64) {
#pragma unroll
            for (int c = 0; c < CKS; ++c) xr[c] = xb[(size_t)(ci0 + (c < cmax ? c : cmax)) * hw + goffs];
        }
        if constexpr (!ENC) sr = sb[(size_t)s_r * p.cin + ci0 + (s_c < cmax ? s_c : cmax)];
    };
    auto store_x = [&](int buf, int chunk) __attribute__((always_inline)) {     // the staged chunk -> patch buffer `buf` in the K loop's operand form
        float4* xf4 = reinterpret_cast<float4*>(lds_raw + L::PATCH0 + buf * MX_PATCHB);
        if (tid < C::PATCH) {
#pragma unroll
            for (int c = 0; c < CKS; ++c) xr[c] = p_in ? xr[c] : 0.f;
            if constexpr (ENC) {  // instance norm on load: this sample's statistics from the table staged in LDS at kernel start (mean 0 / rstd 1 without),
                                  // read as wave-uniform float4s — as global loads they were 32 vector-memory requests per chunk inside the K loop
                const float4* nm = reinterpret_cast<const float4*>(lds_raw + L::BYTES) + chunk * (CKS / 4);
                const float4* nr = reinterpret_cast<const float4*>(lds_raw + L::BYTES + MX_NORM_BYTES / 2) + chunk * (CKS / 4);
#pragma unroll
                for (int c4 = 0; c4 < CKS / 4; ++c4) {
                    const float4 m4 = nm[c4], r4 = nr[c4];
                    xr[4 * c4] = p_in ? (xr[4 * c4] - m4.x) * r4.x : 0.f;
                    xr[4 * c4 + 1] = p_in ? (xr[4 * c4 + 1] - m4.y) * r4.y : 0.f;
                    xr[4 * c4 + 2] = p_in ? (xr[4 * c4 + 2] - m4.z) * r4.z : 0.f;
                    xr[4 * c4 + 3] = p_in ? (xr[4 * c4 + 3] - m4.w) * r4.w : 0.f;
                }
            }
            if constexpr (ENC && ARITH == 1) {
                // The operand does not depend on the output pixel here, so a1 = f16(a) and a - a1 are made ONCE per staged value (each feeds nine taps):
                // the pixel's four 16-byte slots hold [a1 of channels 0-7 | a - a1 of 0-7 | a1 of 8-15 | a - a1 of 8-15] as f16 —
                // the K loop then reads its two fragments per tap and does no floating-point VALU work at all.
                uint4* xq = reinterpret_cast<uint4*>(xf4);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    unsigned q1[4], q2[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = xr[hh * 8 + 2 * j], bq = xr[hh * 8 + 2 * j + 1];
                        const f16x2 a1 = __builtin_convertvector((f32x2){a, bq}, f16x2);
                        q1[j] = __builtin_bit_cast(unsigned, a1);
                        q2[j] = pack_f16_rne(a - (float)a1[0], bq - (float)a1[1]);
                    }
                    xq[(2 * hh) * MX_PSTRIDE + tid] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
                    xq[(2 * hh + 1) * MX_PSTRIDE + tid] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) xf4[k * MX_PSTRIDE + tid] = make_float4(xr[4 * k], xr[4 * k + 1], xr[4 * k + 2], xr[4 * k + 3]);
            }
        }
        if constexpr (!ENC) {
            if (tid < E4S_MAX_REGIONS * CKS)
                reinterpret_cast<float*>(lds_raw + L::SS0 + buf * MX_SSB)[tid] = (tid / CKS < p.nreg && chunk * CKS + s_c < p.cin) ? sr : 0.f;
        }
    };
    // row `row` of chunk `chunk` -> ring slot `row`: pieces wave, wave + 8, ... of NPIECE
    const unsigned char* wbase = p.wmx + (size_t)par * nchunk * ncot * 3 * L::ROWB;
    auto dma_row = [&](int chunk, int row) __attribute__((always_inline)) {
        const unsigned char* src = wbase + ((size_t)(chunk * ncot + cotile) * 3 + row) * L::ROWB;
#pragma unroll
        for (int k = 0; k < (L::NPIECE + 7) / 8; ++k) {
            const int piece = wave + 8 * k;
            if (piece < L::NPIECE) dma16_asm(src + piece * 1024, (unsigned)(lane * 16), (unsigned)(row * L::ROWB + piece * 1024));      // (one offset register for all pieces)
        }
    };

    const int ch_begin = ks * p.chunks_per;
    const int ch_end = (ch_begin + p.chunks_per < nchunk) ? ch_begin + p.chunks_per : nchunk;
    if constexpr (ENC) {
        float* nm = reinterpret_cast<float*>(lds_raw + L::BYTES);
        float* nr = reinterpret_cast<float*>(lds_raw + L::BYTES + MX_NORM_BYTES / 2);
        for (int c = tid; c < nchunk * CKS; c += 512) {
            const bool in = c < p.cin && p.in_mean;
            nm[c] = in ? p.in_mean[(size_t)b * p.cin + c] : 0.f;
            nr[c] = in ? p.in_rstd[(size_t)b * p.cin + c] : 1.f;
        }
        __syncthreads();
    }
    if (ch_begin < ch_end) {
        dma_row(ch_begin, 0);
        dma_row(ch_begin, 1);
        dma_row(ch_begin, 2);
        load_x(ch_begin);
        store_x(0, ch_begin);
    }
    E4S_WAIT_VM(0);
    E4S_LDS_BARRIER();

    // A two-phase ("ping-pong") form of this loop was built and measured in round 3 (commit 75c84a8): waves 4-7 half a row behind waves 0-3, each row split into a
    // read / convert phase and an MFMA-only phase with a barrier after each, so that every SIMD always has one wave feeding the matrix pipe.  Correct, but slower
    // (plain mode 512 -> 512 @32^2 x 16: 0.220 against 0.178 ms): cycle stamps put the read phase at 1 550 - 2 450 cycles per row against 810 for the row's 20
    // MFMAs — the row's 38 LDS reads (144 LDS-array cycles per wave, 576 per phase for the four waves of a group, in two dependent rounds) and the patch
    // conversion, not the matrix pipe, set the pace, and the MFMA waves waited at the barrier.  What carried over: the DMA staging, the slot-major patch, and pinning
    // the accumulators (LLVM sinks a row's last MFMAs behind the next barrier otherwise — 60 registers).
    bool ovf = false;       // (f16 arithmetic) a modulated activation of this lane left the f16 range
#pragma unroll 1
    for (int chunk = ch_begin; chunk < ((MX_ABL & 512) ? ch_begin : ch_end); ++chunk) {
        const int cur = (chunk - ch_begin) & 1;
        const bool more = chunk + 1 < ch_end;
        if (more && !(!ENC && (MX_ABL & 64))) load_x(chunk + 1);                       // lands during this chunk (row 0's wait), written to the other patch buffer before its last barrier
        // (masked f16 + fp6 loop) the refill of the ring slot the previous row left is requested behind the row's first MFMAs instead of right behind the barrier:
        // a request stalls the issuing in-order wave on the CU's address unit (~30 cycles per 1 KB request: conv_mx3.hip's stamps), and in front of the row's
        // LDS reads that stall delayed the operands of all of its MFMAs (in-run ratio to the split-bf16 variant of this kernel 0.879 -> 0.859; moving the patch
        // prefetch behind row 0's first MFMAs as well gives 0.881: its loads then have less of row 0 to land in)
        auto deferred_refill = [&](int row) __attribute__((always_inline)) {
            if constexpr (!ENC && (MX_ABL & 32)) return;
            if (row == 0) {
                if (chunk > ch_begin) dma_row(chunk, 2);
            } else if (more) {
                dma_row(chunk + 1, row - 1);
            }
        };

        const float4* xf4 = reinterpret_cast<const float4*>(lds_raw + L::PATCH0 + cur * MX_PATCHB);
        const float* ss = reinterpret_cast<const float*>(lds_raw + L::SS0 + cur * MX_SSB);
        float sv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[e] = ENC ? 1.f : (cls[0] >= 0 ? ss[cls[0] * CKS + khalf * 8 + e] : 0.f);

#pragma unroll
        for (int row = 0; row < 3; ++row) {
            const unsigned char* slot = lds_raw + row * L::ROWB;
            if constexpr (ARITH == 0) {
                const uint4* whalf = reinterpret_cast<const uint4*>(slot) + khalf * MX_TN + l5;      // + t * 2 * TN, + 6 * TN for the lo slab
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int e = xoff + row * C::PW + t;
                    const float4 x0v = xf4[(2 * khalf) * MX_PSTRIDE + e], x1v = xf4[(2 * khalf + 1) * MX_PSTRIDE + e];
                    uint4 bh, bl;
                    split2(x0v.x * sv[0], x0v.y * sv[1], bh.x, bl.x);
                    split2(x0v.z * sv[2], x0v.w * sv[3], bh.y, bl.y);
                    split2(x1v.x * sv[4], x1v.y * sv[5], bh.z, bl.z);
                    split2(x1v.z * sv[6], x1v.w * sv[7], bh.w, bl.w);
                    uint4 ah[4], al[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ah[i] = whalf[t * 2 * MX_TN + i * 32];
                        al[i] = whalf[6 * MX_TN + t * 2 * MX_TN + i * 32];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh), acc[i][0], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl), acc[i][0], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh), acc[i][0], 0, 0, 0);
                }
            } else {
                const uint4* w1half = reinterpret_cast<const uint4*>(slot) + khalf * MX_TN + l5;
                const uint4* f6lo = reinterpret_cast<const uint4*>(slot + MX_W1B) + khalf * MX_TN + l5;                    // + term * 2 * TN + i * 32
                const uint2* f6hi = reinterpret_cast<const uint2*>(slot + MX_W1B + MX_F6LO) + khalf * MX_TN + l5;
                const unsigned* wsc = reinterpret_cast<const unsigned*>(slot + MX_W1B + MX_F6LO + MX_F6HI) + khalf * MX_TN + l5;
                u32x16 v1, v2;        // a1 = f16(a) and a - a1 of the row's 24 values, as the f16 pairs the conversions below take (registers 12..15: copies)
                unsigned ex;          // biased fp32 exponent of the largest |a| among them
                if constexpr (ENC) {
                    // The row's 36 LDS reads are ISSUED TOGETHER, ahead of its first MFMA (sched_barrier below): with two in-order waves per SIMD the loop
                    // was bound by LDS latency — a dozen load -> wait -> use hops per row (-DMX_ABL=1: the reads alone took 78 % of the kernel's time, at a
                    // third of the LDS bandwidth); one hop per row (0.198 -> 0.184 ms on the 512 -> 512 @32^2 launch).  Requesting the next row's fragments under this row's
                    // fp6 MFMAs as well was tried: all three taps spill (the activation prefetch holds 17 registers through the chunk), one tap changes nothing.
                    uint4 xb1[3], wv[3][4], flo[2][4];
                    uint2 fhi[2][4];
                    int fsc[4];
                    {
                        const uint4* xq = reinterpret_cast<const uint4*>(xf4);
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const int e = xoff + row * C::PW + t;
                            xb1[t] = xq[(2 * khalf) * MX_PSTRIDE + e];
                            const uint4 b2 = xq[(2 * khalf + 1) * MX_PSTRIDE + e];
                            v2[t * 4] = b2.x; v2[t * 4 + 1] = b2.y; v2[t * 4 + 2] = b2.z; v2[t * 4 + 3] = b2.w;
#pragma unroll
                            for (int i = 0; i < 4; ++i) wv[t][i] = w1half[t * 2 * MX_TN + i * 32];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned m = 0u;  // running maximum of |a1| as f16 BITS (non-negative halves order like unsigned integers), two lanes of 16 bits
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const uint4 b1 = xb1[t];
                        v1[t * 4] = b1.x; v1[t * 4 + 1] = b1.y; v1[t * 4 + 2] = b1.z; v1[t * 4 + 3] = b1.w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                            const u16x2 mm = __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, v1[t * 4 + j] & 0x7fff7fffu));   // v_pk_max_u16
                            m = __builtin_bit_cast(unsigned, mm);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if constexpr (MX_ABL & 1) { asm volatile("" :: "v"(wv[t][i].x), "v"(wv[t][i].y), "v"(wv[t][i].z), "v"(wv[t][i].w), "v"(b1.x)); }
                            else acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv[t][i]), __builtin_bit_cast(f16x8, b1), acc[i][0], 0, 0, 0);
                        }
                        // the fp6 operands of this row are requested under the f16 MFMAs, into the registers the taps' weight fragments leave behind
                        if (t == 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) { flo[1][i] = f6lo[2 * MX_TN + i * 32]; fhi[1][i] = f6hi[2 * MX_TN + i * 32]; fsc[i] = (int)wsc[i * 32]; }
                        }
                        if (t == 1) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) { flo[0][i] = f6lo[i * 32]; fhi[0][i] = f6hi[i * 32]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const unsigned mh = (m & 0xffffu) > (m >> 16) ? (m & 0xffffu) : (m >> 16);
                    const unsigned e16 = mh >> 10;                    // f16 exponent field: 31 = the value left the f16 range (inf)
                    ovf |= e16 >= 31u;
                    ex = (e16 ? e16 : 1u) + 112u;                     // f16 bias 15 -> fp32 bias 127
                    const unsigned e1 = ex > 3u ? ex - 2u : 1u, e2 = ex > 14u ? ex - 13u : 1u;
                    const u32x6 p1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, v1), __builtin_bit_cast(float, e1 << 23));
                    const u32x6 p2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, v2), __builtin_bit_cast(float, e2 << 23));
                    const i32x8 bx1 = mx_op6(p1), bx2 = mx_op6(p2);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {     // fp6(w - w1) x fp6(a1)
                        if constexpr (MX_ABL & 1) { asm volatile("" :: "v"(flo[1][i].x), "v"(fhi[1][i].x), "v"(fsc[i]), "v"(bx1[0])); }
                        else acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(flo[1][i], fhi[1][i]), bx1, acc[i][0], 2, 2, 1, fsc[i], 0, (int)e1);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {     // fp6(w1) x fp6(a - a1)
                        if constexpr (MX_ABL & 1) { asm volatile("" :: "v"(flo[0][i].x), "v"(fhi[0][i].x), "v"(bx2[0])); }
                        else acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(flo[0][i], fhi[0][i]), bx2, acc[i][0], 2, 2, 0, fsc[i], 0, (int)e2);
                    }
                } else {
                // as in the plain-convolution mode: the row's activation and f16 weight fragments are requested together, ahead of the first conversion
                float amax = 0.f;
                float4 xa[3], xb[3];
                uint4 wv[3][4];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int e = xoff + row * C::PW + t;
                    xa[t] = xf4[(2 * khalf) * MX_PSTRIDE + e];
                    xb[t] = xf4[(2 * khalf + 1) * MX_PSTRIDE + e];
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (MX_ABL & 128) { wv[t][i] = make_uint4(tid, t, i, row); asm volatile("" : "+v"(wv[t][i].x)); }
                        else wv[t][i] = w1half[t * 2 * MX_TN + i * 32];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float xv[8] = {xa[t].x, xa[t].y, xa[t].z, xa[t].w, xb[t].x, xb[t].y, xb[t].z, xb[t].w};
                    if constexpr (MX_ABL & 8) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v1[t * 4 + j] = __builtin_bit_cast(unsigned, xv[2 * j]); v2[t * 4 + j] = __builtin_bit_cast(unsigned, xv[2 * j + 1]); }
                        amax = 1.0f;
                    } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = xv[2 * j] * sv[2 * j], bq = xv[2 * j + 1] * sv[2 * j + 1];
                        const f16x2 a1 = __builtin_convertvector((f32x2){a, bq}, f16x2);
                        v1[t * 4 + j] = __builtin_bit_cast(unsigned, a1);
                        v2[t * 4 + j] = resid_pair_f16(xv[2 * j], sv[2 * j], xv[2 * j + 1], sv[2 * j + 1], v1[t * 4 + j]);
                        amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(bq)));
                    }
                    }
                    const uint4 b1 = make_uint4(v1[t * 4], v1[t * 4 + 1], v1[t * 4 + 2], v1[t * 4 + 3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (MX_ABL & 256) { asm volatile("" :: "v"(wv[t][i].x), "v"(wv[t][i].y), "v"(wv[t][i].z), "v"(wv[t][i].w), "v"(b1.x), "v"(b1.y), "v"(b1.z), "v"(b1.w)); }
                        else acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv[t][i]), __builtin_bit_cast(f16x8, b1), acc[i][0], 0, 0, 0);
                    }
                    if (t == 0) deferred_refill(row);
                }
                // block scales of this lane's 24 values: 2^(E - 2) for fp6(a1), 2^(E - 13) for fp6(a - a1)  (|a - a1| <= 2^(E - 11));  E >= 16 leaves f16
                ex = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
                ovf |= amax >= 65520.f;            // exactly the values f16 rounds to infinity (65520 is the tie between 65504 and 2^16)
                }
                if constexpr (!ENC && (MX_ABL & 16)) { asm volatile("" :: "v"(v1[0]), "v"(v2[0]), "v"(ex)); }
                if constexpr (!ENC && !(MX_ABL & 16)) {
                const unsigned e1 = ex > 3u ? ex - 2u : 1u, e2 = ex > 14u ? ex - 13u : 1u;
                // (positions 24..31 meet zero weights and every fp6 code is finite: registers 12..15 of the tuples are left undefined — no moves)
                const u32x6 p1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, v1), __builtin_bit_cast(float, e1 << 23));
                const u32x6 p2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, v2), __builtin_bit_cast(float, e2 << 23));
                // the MX MFMA reads 6 of its operands' 8 registers for fp6: the last two stay undefined (zero-filling them cost 2 moves per operand)
                const i32x8 bx1 = mx_op6(p1), bx2 = mx_op6(p2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {       // fp6(w - w1) x fp6(a1)
                    uint4 lo; uint2 hi; int sc;
                    if constexpr (ENC && (MX_ABL & 2)) { lo = make_uint4(tid, i, row, 1); hi = make_uint2(tid, i); sc = 127 << 8; }
                    else { lo = f6lo[2 * MX_TN + i * 32]; hi = f6hi[2 * MX_TN + i * 32]; sc = (int)wsc[i * 32]; }
                    const i32x8 aw = mx_op6(lo, hi);
                    if constexpr (ENC && (MX_ABL & 1)) { asm volatile("" :: "v"(lo.x), "v"(hi.x), "v"(sc), "v"(bx1[0])); }
                    else acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aw, bx1, acc[i][0], 2, 2, 1, sc, 0, (int)e1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {       // fp6(w1) x fp6(a - a1)
                    uint4 lo; uint2 hi; int sc;
                    if constexpr (ENC && (MX_ABL & 2)) { lo = make_uint4(tid, i, row, 2); hi = make_uint2(i, tid); sc = 127; }
                    else { lo = f6lo[i * 32]; hi = f6hi[i * 32]; sc = (int)wsc[i * 32]; }
                    const i32x8 aw = mx_op6(lo, hi);
                    if constexpr (ENC && (MX_ABL & 1)) { asm volatile("" :: "v"(lo.x), "v"(hi.x), "v"(sc), "v"(bx2[0])); }
                    else acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aw, bx2, acc[i][0], 2, 2, 0, sc, 0, (int)e2);
                }
                }
            }
            if (row == 2 && more && !(!ENC && (MX_ABL & 64))) store_x(cur ^ 1, chunk + 1);         // (its last readers passed the previous chunk's last barrier)
            // everything this wave issued so far has landed; then: every wave is done with this row's slot (and, after row 2, with the patch)
            if constexpr (!(ENC && (MX_ABL & 4)) && !(!ENC && (MX_ABL & 32))) {
                E4S_WAIT_VM(0);
                E4S_LDS_BARRIER();
                if constexpr (ARITH == 0 || ENC) { if (more) dma_row(chunk + 1, row); }
            }

        }
    }

    if constexpr (ARITH == 1) {
        // one report per wave: flags[0] bit 0 = "some launch overflowed" (sticky), flags[1] = a counter that moves whenever one does — the host compares
        // snapshots of it taken before and after a forward pass (ops.MxGuard) and re-runs that pass with the split-bf16 arithmetic
        if (p.flags && __builtin_amdgcn_ballot_w64(ovf) != 0 && lane == 0) { atomicOr(p.flags, 1); atomicAdd(p.flags + 1, 1); }
    }
    if constexpr (ENC) {
        __syncthreads();
        float* sl = reinterpret_cast<float*>(lds_raw);
        if (tid < MX_TN) sl[tid] = (p.slope && co0 + tid < p.cout) ? p.slope[co0 + tid] : 1.f;      // PReLU slope (1 = identity)
        __syncthreads();
        const int y = y0 + wave, x = x0 + l5;
        if (y < p.h && x < p.w) {
            float* op = p.out + (size_t)b * p.cout * hw + (size_t)y * p.w + x;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    float v = acc[i][0][r];
                    v = v > 0.f ? v : v * sl[n];
                    if (co0 + n < p.cout) op[(size_t)(co0 + n) * hw] = v;
                }
        }
        return;
    }
    if (p.ksplit > 1) {
        float* part = p.partial + ((size_t)ks * p.bs + b) * p.cout * ho * wo;
        const int y = y0 + wave, x = x0 + l5;
        if (y < p.h && x < p.w) {
            const size_t opix = (size_t)(p.up ? 2 * y + pa : y) * wo + (p.up ? 2 * x + pb_ : x);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    if (co < p.cout) part[(size_t)co * ho * wo + opix] = acc[i][0][r];
                }
        }
        return;
    }
    if constexpr (MX_ABL & 1024) { if (acc[0][0][0] == 12345.f) p.out[tid] = acc[1][0][1] + acc[2][0][2] + acc[3][0][3]; return; }
    sb_epilogue<C, 4, 1, 8, RGB, OSP>(p, lds_raw, acc, cls, co0, b, y0, x0, pa, pb_, ho, wo, ub_skip);
}

}  // namespace
