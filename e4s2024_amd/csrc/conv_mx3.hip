// a8, round 3: the regional-style encoder's stride-1 3x3 convolutions (models/encoders/helpers.py:128-139: InstanceNorm -> Conv2d 3x3 -> PReLU -> Conv2d 3x3)
// as out = PReLU(conv3x3(norm(x), W)) on the f16 + 2 x MX-fp6 arithmetic of modconv_mx.hip (a w ~ a1 w1 + fp6(a) fp6(w - w1) + fp6(a - a1) fp6(w1)), in the
// form that kernel's measurements asked for ("What the second half of the round measured"):
//   * a TWO-PHASE K loop: waves 4-7 run half a unit behind waves 0-3 (a workgroup's waves go to SIMDs 0,2,1,3,0,2,1,3: wave w and w + 4 share one), a unit is
//     split into a phase that only reads LDS and a phase that only issues MFMAs, a barrier after each — every SIMD always has one wave on the matrix pipe;
//   * 64 pixels x 64 output channels per wave (2 x 2 MFMA blocks) instead of 32 x 128: a third fewer LDS bytes per MFMA;
//   * NO conversion in the loop: a lane of the MX MFMA holds 32 K-values under one scale, so with 32-channel chunks a lane's values are ONE patch pixel's
//     channels — fp6(a), fp6(a - a1) and their block scales are made once per staged value (each feeds nine taps) next to a1 = f16(a), and the loop's read
//     phase is 36 LDS reads in a single round.  (The one-phase kernel converted per wave and tap row: 1 550 - 2 450 cycles of reads + conversions per 810 of MFMA.)
// K structure: chunk = 32 input channels; unit u = 0..4 of a chunk = taps 2u, 2u + 1 (tap 9 does not exist: zero weights): 16 f16 MFMAs (2 taps x 2 K-steps x
// 2 x 2 blocks) + 8 fp6 MFMAs (K = 64 = the two taps' 32 channels, x 2 terms x 2 x 2 blocks) per wave and unit.
// LDS (134 016 B, one workgroup per CU): ONE patch buffer (40 832 B: between two chunks nobody reads while both wave groups convert the next chunk's patch —
// ~4 000 of a chunk's ~17 000 cycles; two buffers do not fit beside a three-slot ring, and a two-slot ring leaves the DMA one unit of cover), the instance-norm
// table as (rstd, -mean rstd), a ring of three UNIT slots refilled by LDS-DMA from inline asm (29 696 B each).
//   unit slot: f16 w1 [tap 2][K-step 2][k half 2][co 128] x 16 B | fp6 codes, first 16 B [term 2][k half 2][co 128] | last 8 B [term][half][co] |
//              E8M0 scales [half 2][co 128] x 4 B (byte 0: fp6(w1), byte 1: fp6(w - w1); k half = tap 2u / 2u + 1, its 32 channels in order)
//   patch:     a1 f16 [16-B slot 4][pixel 352] (slot s = channels 8 s .. 8 s + 7) | codes first 16 B [term 2][pixel] | last 8 B [term][pixel] | scales [pixel] x 4 B
// Every vector-memory request inside the loop is issued from inline asm (weight DMA and the next chunk's activation loads), so hipcc counts none of them and
// the counted vmcnt waits below are exact; the activation loads' destination registers are first read in the store phase, behind such a wait.  Their results are asm
// outputs, so a register copy between the request and that wait would copy stale data: the prefetch is requested at the TOP of the iteration that consumes it (not in the
// previous iteration's store phase, where it was one barrier earlier: measured <= 0.4 % of the encoder) — the registers are defined and used inside one iteration, there
// is no loop-carried value for the compiler to merge, and the loop has ONE call site (the prologue's request is waited for and consumed before the loop).
// Memory formats (round 5; e4s_conv3x3_mx3_ex): the INPUT is fp32 planes [b][c][h][w] (IN = 0: normalised and converted while staging), channel-blocked fp32
// [b][c / 4][h][w][4] (IN = 1: the same staging from 8 x 16-byte requests per thread instead of 32 dwords) or a PREPARED-OPERAND map (IN = 2: the producing convolution's
// epilogue already wrote the patch's entries, 116 bytes per pixel and 32-channel block; staging copies them); the OUTPUT is any of the three (out_c4 / out_prep; fp32 and
// channel-blocked also as the stride-2 consumer's phase planes).  The formats other than fp32 planes exist for ONE edge of the graph: between the two convolutions of a
// bottleneck_IR_SE_Ours unit (helpers.py:128-139), where nothing else reads the map.  Every format gives the same bits.
// Measured (16 images, in-run against e4s_conv3x3_mx): 512 -> 512 @32^2 0.1776 -> 0.1498 ms (516 algorithmic TFLOP/s), 256 -> 256 @64^2 0.184 -> 0.163;
// cycle stamps (-DMX3_PROF): read phase 1 000 - 1 150 cycles (550 of LDS reads + the wave's share of the refill requests: the CU's address unit takes ~30
// cycles per 1 KB request and stalls the issuing wave), MFMA phase 870; without the in-loop refills (-DMX3_NODMA) the kernel takes 0.134 ms.
#include <stdlib.h>

#include "common.h"
#include "sb_common.h"

using namespace e4s;

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TN = 128;                      // output channels per workgroup
constexpr int CK = 32;                       // input channels per chunk
constexpr int TW = 32, TH = 8;               // pixels per workgroup
constexpr int PW = TW + 2, PHH = TH + 2, PATCH = PW * PHH;   // 340
constexpr int PST = 352;                     // pixel stride of the patch's slot rows
constexpr int NUNIT = 5;                     // tap pairs per chunk
constexpr int U_W16 = 2 * 2 * 2 * TN * 16;   // 16 384
constexpr int U_CLO = 2 * 2 * TN * 16;       // 8 192
constexpr int U_CHI = 2 * 2 * TN * 8;        // 4 096
constexpr int U_SC = 2 * TN * 4;             // 1 024
constexpr int UNITB = U_W16 + U_CLO + U_CHI + U_SC;          // 29 696
constexpr int NPIECE = UNITB / 1024;         // 29 wave requests of 1 KB
constexpr int P_A1 = 4 * PST * 16;           // 22 528
constexpr int P_CLO = 2 * PST * 16;          // 11 264
constexpr int P_CHI = 2 * PST * 8;           // 5 632
constexpr int P_SC = PST * 4;                // 1 408
constexpr int PATCHB = P_A1 + P_CLO + P_CHI + P_SC;          // 40 832
constexpr int MAX_CIN = 512;
// (the patch and the table come first: every address in them is one base register + a 16-bit immediate)
constexpr int PATCH0 = 0, NORM0 = PATCH0 + PATCHB, RING0 = NORM0 + 2 * MAX_CIN * 4, LDS_BYTES = RING0 + 3 * UNITB;      // 134 016
static_assert(UNITB % 1024 == 0 && PATCHB % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS plan");

struct Mx3Params {
    float* out;
    const float* x;
    const unsigned char* w;      // prepared weights: [chunk][co tile][unit] x UNITB
    const float* in_mean;        // [bs][cin] or null
    const float* in_rstd;
    const float* slope;          // [cout] or null
    int* flags;
    int bs, cin, cout, h, w_, tiles_x, tiles_y;
    int ho, wo;                  // output size (= h, w_ at stride 1; h / 2, w_ / 2 at stride 2)
    int phased;                  // stride 1: the OUTPUT is written as phase planes; stride 2: the INPUT is read as phase planes  ([b][c][2 py + px][h / 2][w / 2], h and w even)
    int out_prep;                // the OUTPUT is written as PREPARED OPERANDS (see the IN == 2 template flag): what the consumer's staging would compute, computed here once per pixel
    int out_c4;                  // the OUTPUT is channel-blocked: [b][c / 4][plane layout as above][4 floats] — a pixel's four channels are one 16-byte element (the C4 template
                                 // flag says the same of the INPUT): the hand-over between the two convolutions of an IR-SE unit, which nothing else reads (round 5)
};

__device__ __forceinline__ unsigned pack_f16_rne(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
}
__device__ __forceinline__ i32x8 op6(uint4 lo, uint2 hi) {      // six registers of fp6 codes in the MFMA's eight-register operand (the last two are never read)
    const i32x4v a = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w};
    const i32x2v b2 = {(int)hi.x, (int)hi.y};
    const i32x4v b = __builtin_shufflevector(b2, b2, 0, 1, -1, -1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, -1, -1);
}
// LDS-DMA, 16 bytes per lane from (scalar base + 32-bit lane offset) to LDS address `lds_dst` + 16 * lane (see modconv_mx.hip for why it is asm)
__device__ __forceinline__ void dma16(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
// a global load hipcc does not count: the destination is valid only behind one of the kernel's own vmcnt waits
__device__ __forceinline__ float load_uncounted(const float* gbase, unsigned voff) {
    float v;
    // (s_nop 4: five wait states between a VALU write of the base SGPRs — a spilled scalar restored by v_readlane — and this read of them; hipcc's hazard pass does not
    //  look into asm statements.  Round 5: csrc/sb_common.h)
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(gbase) : "memory");
    return v;
}
// the same for a 16-byte element (four channels of one pixel of a channel-blocked map)
__device__ __forceinline__ f32x4 load4_uncounted(const float* gbase, unsigned voff) {
    f32x4 v;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(gbase) : "memory");
    return v;
}
template <typename T>
__device__ __forceinline__ void pin_here(T& v) { asm volatile("" : "+v"(v)); }
// f16 pair of split residuals, lo = f16(a - a1.lo), hi = f16(b - a1.hi): one v_fma_mix each (fp32 a, b; f16 a1)
__device__ __forceinline__ unsigned resid_pair_f16(float a, float b, unsigned a1) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(a1));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(b), "v"(a1));
    return r;
}

// 16 raw bytes / 4 raw bytes of a prepared-operand map (same caveats as load_uncounted)
// (the asm's output registers are returned AS THEY ARE: any conversion here would be a register copy in front of the kernel's wait — stale data)
__device__ __forceinline__ i32x4v load16_uncounted(const unsigned char* gbase, unsigned voff) {
    i32x4v v;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(gbase) : "memory");
    return v;
}
__device__ __forceinline__ uint4 as_uint4(i32x4v v) { return make_uint4((unsigned)v[0], (unsigned)v[1], (unsigned)v[2], (unsigned)v[3]); }
__device__ __forceinline__ unsigned load4u_uncounted(const unsigned char* gbase, unsigned voff) {
    unsigned v;
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(gbase) : "memory");
    return v;
}
// One pixel's 32 channels -> the operands the K loop reads: a1 = f16(a) (q1: pairs), fp6(a1) and fp6(a - a1) under the block scales 2^(E - 2) / 2^(E - 13) (c1, c2),
// the scale bytes, and whether a value left the f16 range.  The staging's arithmetic (store_x below), instruction for instruction: a map prepared with this by its
// PRODUCER gives the consumer the bits its own staging would have made.
__device__ __forceinline__ void encode32(const float (&a)[32], u32x16& q1, u32x6& c1, u32x6& c2, unsigned& scales, unsigned& ovf) {
    u32x16 q2;
    unsigned m = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        q1[j] = pack_f16_rne(a[2 * j], a[2 * j + 1]);
        q2[j] = resid_pair_f16(a[2 * j], a[2 * j + 1], q1[j]);
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const u16x2 mm = __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, q1[j] & 0x7fff7fffu));
        m = __builtin_bit_cast(unsigned, mm);
    }
    const unsigned mh = (m & 0xffffu) > (m >> 16) ? (m & 0xffffu) : (m >> 16);
    const unsigned e16 = mh >> 10;
    ovf |= e16 >= 31u ? 1u : 0u;
    const unsigned ex = (e16 ? e16 : 1u) + 112u;
    const unsigned e1 = ex > 3u ? ex - 2u : 1u, e2 = ex > 14u ? ex - 13u : 1u;
    c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
    c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
    scales = e1 | (e2 << 8);
}
// bytes of a prepared-operand map per pixel and 32-channel block: a1 4 x 16 | codes 2 x 16 | code tails 16 (term 0: 8, term 1: 8) | scales 4
constexpr int PREPB = 116;

// ============================================================================ weight preparation
// One thread per (chunk, co tile, unit, k half, co): tap 2 unit + half, its 32 channels.
// s2: the stride-2 kernel's units (see conv3x3_mx3_kernel<true>): unit u, half d = tap S2TAP[u][d] of the 3 x 3 kernel (-1: none, zero weights)
__global__ __launch_bounds__(256) void prep_weights_mx3_kernel(unsigned char* __restrict__ dst, const float* __restrict__ weight, int cout, int cin, int s2) {
    const int nchunk = cin / CK, ntile = (cout + TN - 1) / TN;
    const int64_t total = (int64_t)nchunk * ntile * NUNIT * 2 * TN;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int n = (int)(r % TN); r /= TN;
        const int half = (int)(r & 1); r >>= 1;
        const int unit = (int)(r % NUNIT); r /= NUNIT;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int s2tap = unit == 0 ? (half ? 2 : 0) : unit == 1 ? (half ? 8 : 6) : unit == 2 ? (half ? 5 : 3) : unit == 3 ? (half ? 7 : 1) : (half ? 9 : 4);
        const int co = tile * TN + n, tap = s2 ? s2tap : 2 * unit + half;
        unsigned char* slot = dst + (((size_t)chunk * ntile + tile) * NUNIT + unit) * UNITB;
        u32x16 q1, q2;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float a = 0.f, b = 0.f;
            if (tap < 9 && co < cout) {
                a = weight[((size_t)co * cin + chunk * CK + 2 * j) * 9 + tap];
                b = weight[((size_t)co * cin + chunk * CK + 2 * j + 1) * 9 + tap];
            }
            const f16x2 hh = __builtin_convertvector((f32x2){a, b}, f16x2);
            // the residual w - w1 is at most 2^-12 |w|: it goes through f16 scaled by 2^12 (a normal f16 whatever the weight scale), the factor comes back out of its block scale
            const float ra = (a - (float)hh[0]) * 4096.f, rb = (b - (float)hh[1]) * 4096.f;
            q1[j] = __builtin_bit_cast(unsigned, hh);
            q2[j] = pack_f16_rne(ra, rb);
            m1 = fmaxf(m1, fmaxf(fabsf((float)hh[0]), fabsf((float)hh[1])));
            m2 = fmaxf(m2, fmaxf(fabsf(ra), fabsf(rb)));
        }
        // f16 part: [tap half][K-step j][k half kb2][co] x 16 B, K-step j / kb2 = channels 16 j + 8 kb2 .. + 7 = registers 4 (2 j + kb2) .. + 3
        uint4* w16 = reinterpret_cast<uint4*>(slot);
#pragma unroll
        for (int s = 0; s < 4; ++s) w16[(half * 4 + s) * TN + n] = make_uint4(q1[4 * s], q1[4 * s + 1], q1[4 * s + 2], q1[4 * s + 3]);
        // block scale 2^(E - 2) with 2^E <= max < 2^(E + 1): the largest value lands in [4, 8) of e2m3's [0, 7.5]
        auto expo = [](float m) { const unsigned ex = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu; return ex > 3u ? ex - 2u : 1u; };
        const unsigned e1 = expo(m1), e2 = expo(m2);
        const u32x6 c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
        const u32x6 c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
        uint4* clo = reinterpret_cast<uint4*>(slot + U_W16);
        uint2* chi = reinterpret_cast<uint2*>(slot + U_W16 + U_CLO);
        unsigned* scp = reinterpret_cast<unsigned*>(slot + U_W16 + U_CLO + U_CHI);
        clo[(0 * 2 + half) * TN + n] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
        chi[(0 * 2 + half) * TN + n] = make_uint2(c1[4], c1[5]);
        clo[(1 * 2 + half) * TN + n] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
        chi[(1 * 2 + half) * TN + n] = make_uint2(c2[4], c2[5]);
        const unsigned e2s = e2 > 12u ? e2 - 12u : 0u;
        scp[half * TN + n] = e1 | (e2s << 8);
    }
}

// ============================================================================ the kernel
// S2 (round 4): the stride-2, pad-1 form (the regional-style encoder's second convolution of a stage's first unit, helpers.py:128-139 with stride 2).  Output (oy, ox)
// reads input row 2 oy + ky - 1 = 2 (oy + dy) + py: with the input seen as four PHASE planes (py, px) at half resolution, tap ky = 0 is (py = 1, dy = -1), ky = 1 is
// (py = 0, dy = 0), ky = 2 is (py = 1, dy = 0), columns alike — every tap is a stride-1 tap (dy, dx) in {-1, 0}^2 of ONE phase plane.  A 32-channel chunk therefore
// takes FOUR stagings of the same 34 x 10 patch buffer (one per phase plane, pixel (ppy, ppx) = half-resolution position (y0 - 1 + ppy, x0 - 1 + ppx)) and the same
// FIVE units, whose two K halves are now:  u0 = taps (0,0) (0,2), u1 = (2,0) (2,2) on phase (1,1);  u2 = (1,0) (1,2) on phase (0,1);  u3 = (0,1) (2,1) on phase (1,0);
// u4 = (1,1) + nothing on phase (0,0) — each of the nine taps once.  (Phase order 3, 1, 2, 0: phases (py, 1) and (py, 0) share their cache lines, so the stagings
// behind a one-unit sub-chunk — (1,0) after (0,1), (0,0) after (1,0) — find theirs in L2, and the cold lines of (0,1) are requested with two units of cover.)  Weight ring, refills, waits and the two-phase schedule are the stride-1 kernel's; a sub-chunk's
// prefetch is requested at the top of its predecessor's first unit, so the one-unit sub-chunks have ONE unit of cover for it (the stride-1 kernel has five).
// C4 (round 5): the input is channel-blocked ([b][c / 4][...][4]): a patch thread requests its pixel's 32 channels of a chunk as EIGHT 16-byte loads instead of 32
// dword loads — a quarter of the wave requests through the CU's address unit (~18 - 30 cycles each, the issuing wave stalled meanwhile: 2 300 - 2 600 cycles per chunk).
// IN == 2 (round 5): the input is a PREPARED-OPERAND map, written by the producing convolution's epilogue (out_prep) — per image, 32-channel block kb and pixel pi of
// the hw pixels (phase-plane order for the stride-2 consumer):  a1 [slot 4][pi] x 16 B | codes [term 2][pi] x 16 B | code tails [pi] x 16 B | scales [pi] x 4 B  = 116 hw
// bytes per block, exactly the patch's entries.  Staging is then eight loads and seven LDS writes per thread: no normalisation, no conversion (the store phase between two
// chunks was 2 600 cycles of a chunk's 16 000 - 19 000; the stride-2 form has four of them per chunk).
template <bool S2, int IN>
__global__ __launch_bounds__(512, 2) void conv3x3_mx3_kernel(const Mx3Params p) {
    constexpr bool C4 = IN == 1, PREP = IN == 2;
#ifdef MX3_PROF
    const unsigned long long tKernel = __builtin_readcyclecounter();
#endif
    constexpr int NLD = IN ? CK / 4 : CK;            // load requests of a prefetch, per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l5 = lane & 31, khalf = lane >> 5;
    const int pr = wave & 3, chh = wave >> 2;        // the wave's pixel-row pair and 64-channel half
    const int grp = chh;                             // waves 4-7 run half a unit behind

    // XCD affinity as in modconv_mx.hip: an XCD's L2 serves one co tile's weights
    unsigned bx_g = blockIdx.x, cot_g = blockIdx.y, b_g = blockIdx.z;
    {
        const unsigned nx = gridDim.x, ncg = gridDim.y;
        const unsigned long long tot = (unsigned long long)nx * ncg * gridDim.z;
        if ((ncg == 2 || ncg == 4 || ncg == 8) && tot % 8 == 0) {
            const unsigned lin = blockIdx.x + nx * (blockIdx.y + ncg * blockIdx.z);
            const unsigned per = 8u / ncg;
            const unsigned xcd = lin & 7u, q = lin >> 3;
            cot_g = xcd / per;
            const unsigned r = q * per + (xcd % per);
            bx_g = r % nx;
            b_g = r / nx;
        }
    }
    const int tile = (int)bx_g;
    const int y0 = (tile / p.tiles_x) * TH, x0 = (tile % p.tiles_x) * TW;
    const int cotile = (int)cot_g, co0 = cotile * TN, b = (int)b_g;
    const int hw = p.h * p.w_;
    const int nchunk = p.cin / CK, ncot = (p.cout + TN - 1) / TN;
    const int nunits = nchunk * NUNIT;
    const float* xb = p.x + (size_t)b * p.cin * hw;

    // instance-norm statistics of this sample, all channels, as (rstd, -mean * rstd): norm(x) = fma(x, rstd, -mean * rstd)  (1, 0 without)
    {
        float* nr = reinterpret_cast<float*>(lds + NORM0);
        float* nb = nr + MAX_CIN;
        for (int c = tid; c < p.cin; c += 512) {
            const float r = p.in_mean ? p.in_rstd[(size_t)b * p.cin + c] : 1.f;
            nr[c] = r;
            nb[c] = p.in_mean ? -p.in_mean[(size_t)b * p.cin + c] * r : 0.f;
        }
    }

    // ---- staging: thread t < 340 owns patch pixel t; its 32 channels of the next chunk sit in registers from the chunk's start to the store phase
    // (the pixel's coordinates are recomputed from a pinned copy of the thread id where they are used: as loop invariants they would sit in registers the K loop needs)
    auto patch_pixel = [&](bool& in, int ph) __attribute__((always_inline)) {      // ph = 2 py + px: the phase plane (stride 2 only)
        int t = tid;
        pin_here(t);
        const int ppy = t / PW, ppx = t - ppy * PW;
        if constexpr (S2) {
            const int qy = y0 - 1 + ppy, qx = x0 - 1 + ppx;                  // half-resolution position
            in = t < PATCH && ppy <= TH && ppx <= TW && qy >= 0 && qy < p.ho && qx >= 0 && qx < p.wo;      // (taps reach rows 0 .. TH and columns 0 .. TW of the patch only)
            // phase planes (the producer wrote them: consecutive lanes read consecutive floats) or the plain map (every other float of a row: twice the lines per request)
            const unsigned off = p.phased ? (unsigned)(ph * (p.ho * p.wo) + qy * p.wo + qx) : (unsigned)((2 * qy + (ph >> 1)) * p.w_ + 2 * qx + (ph & 1));
            return in ? off * (IN ? 16u : 4u) : 0u;
        } else {
            const int pgy = y0 - 1 + ppy, pgx = x0 - 1 + ppx;
            in = t < PATCH && pgy >= 0 && pgy < p.h && pgx >= 0 && pgx < p.w_;
            return in ? (unsigned)(pgy * p.w_ + pgx) * (IN ? 16u : 4u) : 0u;
        }
    };
    float xr[IN ? 1 : CK];
    f32x4 xr4[C4 ? CK / 4 : 1];
    i32x4v xq[PREP ? 7 : 1];
    unsigned xs = 0u;
    // the next chunk's 32 channels of this thread's pixel: requested between the store phase's barriers, where the wave would otherwise idle (spread over the
    // read phases the requests' issue — the address unit takes a wave request per ~18 cycles and stalls the issuing wave — made every read phase longer than
    // the MFMA phase beside it); waves 6 and 7 own no patch pixel and request nothing
    auto load_x = [&](int chunk, int ph) __attribute__((always_inline)) {
        if (wave < 6) {
            bool p_in;
            const unsigned goff = patch_pixel(p_in, ph);
            if constexpr (PREP) {
                const unsigned char* blk = reinterpret_cast<const unsigned char*>(p.x) + ((size_t)b * nchunk + chunk) * ((size_t)PREPB * hw);
                const unsigned plane = 16u * (unsigned)hw;          // (one scalar base, the element's plane in the lane offset: eight 64-bit bases spill scalars; 116 hw < 2^32: the launcher)
#pragma unroll
                for (int e = 0; e < 7; ++e) xq[e] = load16_uncounted(blk, goff + (unsigned)e * plane);
                xs = load4u_uncounted(blk, 7u * plane + (goff >> 2));
            } else if constexpr (C4) {
#pragma unroll
                for (int c4 = 0; c4 < CK / 4; ++c4) xr4[c4] = load4_uncounted(xb + (size_t)(chunk * CK + 4 * c4) * hw, goff);      // (plane c / 4 starts at c hw floats)
            } else {
#pragma unroll
                for (int c = 0; c < CK; ++c) xr[c] = load_uncounted(xb + (size_t)(chunk * CK + c) * hw, goff);
            }
        }
    };
    unsigned ovf = 0u;
    auto store_x = [&](int chunk, int ph) __attribute__((always_inline)) {
        // Branch-free arithmetic for every lane (a first version read the table under per-lane conditions: hipcc made it eight dependent LDS round trips); pixels
        // outside the map and the threads beyond the patch compute on a clamped pixel's data and are zeroed afterwards; only the final writes are predicated.
        bool p_in;
        (void)patch_pixel(p_in, ph);
        if constexpr (PREP) {          // the producer's entries as they are (zeros for the padding)
            if (tid < PATCH) {
                const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                uint4* a1p = reinterpret_cast<uint4*>(lds + PATCH0);
#pragma unroll
                for (int s = 0; s < 4; ++s) a1p[s * PST + tid] = p_in ? as_uint4(xq[s]) : z;
                uint4* clo = reinterpret_cast<uint4*>(lds + PATCH0 + P_A1);
                uint2* chi = reinterpret_cast<uint2*>(lds + PATCH0 + P_A1 + P_CLO);
                clo[tid] = p_in ? as_uint4(xq[4]) : z;
                clo[PST + tid] = p_in ? as_uint4(xq[5]) : z;
                chi[tid] = p_in ? make_uint2((unsigned)xq[6][0], (unsigned)xq[6][1]) : make_uint2(0u, 0u);
                chi[PST + tid] = p_in ? make_uint2((unsigned)xq[6][2], (unsigned)xq[6][3]) : make_uint2(0u, 0u);
                reinterpret_cast<unsigned*>(lds + PATCH0 + P_A1 + P_CLO + P_CHI)[tid] = p_in ? xs : (111u | (100u << 8));      // (what the staging makes of a zero pixel)
            }
            return;
        }
        const float4* nr = reinterpret_cast<const float4*>(lds + NORM0) + chunk * (CK / 4);
        const float4* nb = reinterpret_cast<const float4*>(lds + NORM0 + MAX_CIN * 4) + chunk * (CK / 4);
        u32x16 q1, q2;
        unsigned m = 0u;      // maximum of |a1| as f16 bits, two lanes of 16
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 r4[4], b4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { r4[k] = nr[half * 4 + k]; b4[k] = nb[half * 4 + k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c4 = half * 4 + k;
                float v0, v1, v2, v3;
                if constexpr (C4) { v0 = xr4[c4][0]; v1 = xr4[c4][1]; v2 = xr4[c4][2]; v3 = xr4[c4][3]; }
                else { v0 = xr[4 * c4]; v1 = xr[4 * c4 + 1]; v2 = xr[4 * c4 + 2]; v3 = xr[4 * c4 + 3]; }
                const float a0 = __builtin_fmaf(v0, r4[k].x, b4[k].x), a1v = __builtin_fmaf(v1, r4[k].y, b4[k].y);
                const float a2v = __builtin_fmaf(v2, r4[k].z, b4[k].z), a3 = __builtin_fmaf(v3, r4[k].w, b4[k].w);
                q1[2 * c4] = pack_f16_rne(a0, a1v);
                q1[2 * c4 + 1] = pack_f16_rne(a2v, a3);
                q2[2 * c4] = resid_pair_f16(a0, a1v, q1[2 * c4]);
                q2[2 * c4 + 1] = resid_pair_f16(a2v, a3, q1[2 * c4 + 1]);
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                u16x2 mm = __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, q1[2 * c4] & 0x7fff7fffu));
                mm = __builtin_elementwise_max(mm, __builtin_bit_cast(u16x2, q1[2 * c4 + 1] & 0x7fff7fffu));
                m = __builtin_bit_cast(unsigned, mm);
            }
        }
        if (!p_in) {          // padding: exact zeros (a divergent branch that interior tiles never take)
#pragma unroll
            for (int j = 0; j < 16; ++j) { q1[j] = 0u; q2[j] = 0u; }
            m = 0u;
        }
        const unsigned mh = (m & 0xffffu) > (m >> 16) ? (m & 0xffffu) : (m >> 16);
        const unsigned e16 = mh >> 10;                    // f16 exponent field: 31 = the value left the f16 range
        ovf |= e16 >= 31u ? 1u : 0u;
        const unsigned ex = (e16 ? e16 : 1u) + 112u;      // f16 bias 15 -> fp32 bias 127
        // block scales of the pixel's 32 values: 2^(E - 2) for fp6(a1), 2^(E - 13) for fp6(a - a1)  (|a - a1| <= 2^(E - 11))
        const unsigned e1 = ex > 3u ? ex - 2u : 1u, e2 = ex > 14u ? ex - 13u : 1u;
        const u32x6 c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
        const u32x6 c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
        if (tid < PATCH) {
            uint4* a1p = reinterpret_cast<uint4*>(lds + PATCH0);
#pragma unroll
            for (int s = 0; s < 4; ++s) a1p[s * PST + tid] = make_uint4(q1[4 * s], q1[4 * s + 1], q1[4 * s + 2], q1[4 * s + 3]);
            uint4* clo = reinterpret_cast<uint4*>(lds + PATCH0 + P_A1);
            uint2* chi = reinterpret_cast<uint2*>(lds + PATCH0 + P_A1 + P_CLO);
            clo[tid] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
            chi[tid] = make_uint2(c1[4], c1[5]);
            clo[PST + tid] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
            chi[PST + tid] = make_uint2(c2[4], c2[5]);
            reinterpret_cast<unsigned*>(lds + PATCH0 + P_A1 + P_CLO + P_CHI)[tid] = e1 | (e2 << 8);
        }
    };
    // unit g (global index) -> ring slot `slot`: pieces wave, wave + 8, ... of 29 (waves 0-4 issue four requests, waves 5-7 three)
    // unit g (global index) -> ring slot `slot`: 29 requests of 1 KB, pieces wave, wave + 8, ... (waves 0-4 issue four, waves 5-7 three)
    auto unit_src = [&](int g) __attribute__((always_inline)) {
        const int chunk = g / NUNIT, u = g - chunk * NUNIT;
        return p.w + (((size_t)chunk * ncot + cotile) * NUNIT + u) * UNITB;
    };
    // (Measured and dropped in round 5: a wave's last piece requested behind its MFMA phase instead of in its read phase — the read phase is the longer one with prepared
    //  operands, 1 190 against 880 cycles — made the 512 -> 512 @32 launch 3 % SLOWER (144.6 -> 149.3 us): the request costs the MFMA phase what it saves the read phase.
    //  A ninth loader wave does not fit the register file: two waves per SIMD hold 2 x 256 registers.)
    auto dma_unit = [&](int g, int slot) __attribute__((always_inline)) {
        const unsigned char* src = unit_src(g);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = wave + 8 * k;
            if (piece < NPIECE) dma16(src + piece * 1024, (unsigned)(lane * 16), (unsigned)(RING0 + slot * UNITB + piece * 1024));
        }
    };
    // counted wait: the refill requested in the previous read phase has landed; younger requests stay in flight: `d` refills of this wave (waves 0-4: four
    // pieces, waves 5-7: three) and (lx) the activation prefetch (32 loads, waves 0-5)
    auto wait_units = [&](bool d, bool lx) __attribute__((always_inline)) {
        if (wave < 5) {
            if (d && lx) E4S_WAIT_VM(4 + NLD); else if (d) E4S_WAIT_VM(4); else if (lx) E4S_WAIT_VM(NLD); else E4S_WAIT_VM(0);
        } else if (wave == 5) {
            if (d && lx) E4S_WAIT_VM(3 + NLD); else if (d) E4S_WAIT_VM(3); else if (lx) E4S_WAIT_VM(NLD); else E4S_WAIT_VM(0);
        } else {
            if (d) E4S_WAIT_VM(3); else E4S_WAIT_VM(0);
        }
    };

    f32x16 acc[2][2];        // [co block][pixel block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: the first three units' weights, the first chunk's patch
    dma_unit(0, 0);
    dma_unit(1, 1);
    dma_unit(2, 2);
    load_x(0, 3);
    E4S_WAIT_VM(0);
    E4S_LDS_BARRIER();          // (also: the norm table is written)
    store_x(0, 3);
    E4S_LDS_BARRIER();
    if (grp) E4S_LDS_BARRIER(); // waves 4-7: half a unit behind from here on

    // per-lane LDS bases: activations (pixel (2 pr, l5) of the tile = patch pixel (2 pr) * 34 + l5, tap offsets are immediates), weights (co half, lane)
    const int ebase = (2 * pr) * PW + l5;
    const unsigned char* a1b = lds + PATCH0 + (khalf * PST + ebase) * 16;                 // + (2 j) * PST * 16 for K-step j, + pixel offset * 16
    const unsigned char* wl = lds + RING0 + (khalf * TN + chh * 64 + l5) * 16;            // f16: + ((d * 2 + j) * 2) * TN * 16 + cb * 512
    int slot = 0, g = 0;
#ifdef MX3_PROF
    unsigned long long tR = 0, tWR = 0, tM = 0, tWM = 0, tST = 0, tS1 = 0, tS2 = 0, tS3 = 0, tS = __builtin_readcyclecounter(), tLoop = tS;
#define MX3_STAMP(accum) { const unsigned long long tn = __builtin_readcyclecounter(); accum += tn - tS; tS = tn; }
#else
#define MX3_STAMP(accum)
#endif
#pragma unroll 1
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const bool more = chunk + 1 < nchunk;
        // The next chunk's activations are requested HERE and consumed by this iteration's store phase: the prefetch registers are defined and used inside one iteration
        // (no loop-carried value whose merge the compiler could materialise as a copy before the data has landed: round-3 advisor finding), with ONE in-loop call site.
        if constexpr (!S2) { if (more) load_x(chunk + 1, 0); }
#pragma unroll
        for (int u = 0; u < NUNIT; ++u, ++g) {
            // sub-chunks (stagings of the patch): stride 1: the chunk's five units; stride 2: units {0, 1} {2} {3} {4} on the phase planes 3, 1, 2, 0
            const bool first = S2 ? u != 1 : u == 0, last = S2 ? u != 0 : u == NUNIT - 1;
            const bool nxt_here = S2 && u != NUNIT - 1;                        // the next staging belongs to this chunk
            const bool have_next = nxt_here || more;
            const int nchunk_i = nxt_here ? chunk : chunk + 1;
            const int nph = u == 0 ? 1 : u == 2 ? 2 : u == 3 ? 0 : 3;
            if constexpr (S2) { if (first && have_next) load_x(nchunk_i, nph); }
            // ---------------- R phase: every operand of the unit into registers, one round of LDS reads
            const int t0 = 2 * u, t1 = 2 * u + 1 < 9 ? 2 * u + 1 : 8;          // (tap 9: zero weights; its activations are tap 8's)
            // patch-pixel offsets of the two taps (stride 2: (dy + 1) * PW + (dx + 1) of the unit's taps, see the kernel's header)
            const int o0 = S2 ? (u == 0 ? 0 : u == 1 ? PW : u == 2 ? PW : u == 3 ? 1 : PW + 1) : (t0 / 3) * PW + t0 % 3;
            const int o1 = S2 ? (u == 0 ? 1 : PW + 1) : (t1 / 3) * PW + t1 % 3;
            const int ok = khalf ? o1 : o0;                                       // the tap this lane's fp6 K half belongs to
            uint4 xa[2][2][2], wv[2][2][2];          // [pixel / co block][tap][K-step]
            uint4 calo[2][2], wclo[2][2];            // [block][term]
            uint2 cahi[2][2], wchi[2][2];
            int sca[2], scw[2];
            const unsigned char* ws = wl + slot * UNITB;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    xa[pb][0][j] = *reinterpret_cast<const uint4*>(a1b + ((2 * j) * PST + pb * PW + o0) * 16);
                    xa[pb][1][j] = *reinterpret_cast<const uint4*>(a1b + ((2 * j) * PST + pb * PW + o1) * 16);
                }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int j = 0; j < 2; ++j) wv[cb][d][j] = *reinterpret_cast<const uint4*>(ws + ((d * 2 + j) * 2 * TN + cb * 32) * 16);
            {
                const unsigned char* cl = lds + PATCH0 + P_A1 + (ebase + ok) * 16;
                const unsigned char* ch = lds + PATCH0 + P_A1 + P_CLO + (ebase + ok) * 8;
                const unsigned char* sc = lds + PATCH0 + P_A1 + P_CLO + P_CHI + (ebase + ok) * 4;
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
                    for (int term = 0; term < 2; ++term) {
                        calo[pb][term] = *reinterpret_cast<const uint4*>(cl + (term * PST + pb * PW) * 16);
                        cahi[pb][term] = *reinterpret_cast<const uint2*>(ch + (term * PST + pb * PW) * 8);
                    }
                    sca[pb] = *reinterpret_cast<const int*>(sc + pb * PW * 4);
                }
                const unsigned char* wc = lds + RING0 + slot * UNITB + U_W16 + (khalf * TN + chh * 64 + l5) * 16;
                const unsigned char* wh = lds + RING0 + slot * UNITB + U_W16 + U_CLO + (khalf * TN + chh * 64 + l5) * 8;
                const unsigned char* wsc = lds + RING0 + slot * UNITB + U_W16 + U_CLO + U_CHI + (khalf * TN + chh * 64 + l5) * 4;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                    for (int term = 0; term < 2; ++term) {
                        wclo[cb][term] = *reinterpret_cast<const uint4*>(wc + (term * 2 * TN + cb * 32) * 16);
                        wchi[cb][term] = *reinterpret_cast<const uint2*>(wh + (term * 2 * TN + cb * 32) * 8);
                    }
                    scw[cb] = *reinterpret_cast<const int*>(wsc + cb * 32 * 4);
                }
            }
            // vector-memory requests behind the LDS reads (they issue while the LDS data returns): the refill of the slot the previous unit left — both groups
            // have read it: waves 0-3 are two barriers past their read of it, waves 4-7 one barrier past theirs, which was the later one — then the prefetch
#ifndef MX3_NODMA     // (tuning builds: -DMX3_NODMA leaves the in-loop refills out — results are then meaningless; measured: their issue costs 7 % of the kernel)
            if (g >= 1 && g + 2 < nunits) dma_unit(g + 2, slot == 0 ? 2 : slot - 1);
#endif
            // which requests are younger than the refill that must have landed (the one of the previous read phase): this phase's refill and, in a chunk's first
            // unit, the prefetch requested in the store phase between the two
            const bool d_younger = g + 2 < nunits;
            const bool lx_younger = first && have_next;
            if (grp) wait_units(d_younger, lx_younger);
            __builtin_amdgcn_sched_barrier(0);
#ifdef MX3_PROF
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            MX3_STAMP(tR)
            E4S_LDS_BARRIER();
            MX3_STAMP(tWR)
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- M phase: 16 f16 + 8 fp6 MFMAs, nothing else (a refill request in front of or between the MFMAs kept the matrix pipe idle for
            // ~350 cycles per unit: vector-memory issue stalls the in-order wave)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                if (d == 1 && u == NUNIT - 1) break;          // tap 9
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int pb = 0; pb < 2; ++pb)
                            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv[cb][d][j]), __builtin_bit_cast(f16x8, xa[pb][d][j]), acc[cb][pb], 0, 0, 0);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    // fp6(w - w1) x fp6(a1): weight term 1 (scale byte 1), activation term 0 (scale byte 0)
                    acc[cb][pb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op6(wclo[cb][1], wchi[cb][1]), op6(calo[pb][0], cahi[pb][0]), acc[cb][pb], 2, 2, 1, scw[cb], 0, sca[pb]);
                    // fp6(w1) x fp6(a - a1)
                    acc[cb][pb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op6(wclo[cb][0], wchi[cb][0]), op6(calo[pb][1], cahi[pb][1]), acc[cb][pb], 2, 2, 0, scw[cb], 1, sca[pb]);
                }
            // (LLVM sinks a unit's last MFMAs to their first use — behind the next barrier, their operand registers with them — unless the accumulators are pinned)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) pin_here(acc[cb][pb]);
            if (!grp) wait_units(d_younger, lx_younger && !last);          // (a sub-chunk's last unit: the prefetch must have landed — the store phase follows)
            // waves 4-7 convert their pixels of the next chunk's patch right behind the chunk's last MFMAs — while waves 0-3 (one barrier ahead, done with the
            // patch like everyone: this group's last read phase was the last) convert theirs: see the store phase below
            if (grp && last && have_next) {
                if (lx_younger) wait_units(d_younger, false);               // (one-unit sub-chunk: this group's wait in the read phase let the prefetch stay in flight)
                store_x(nchunk_i, nph);
            }
            __builtin_amdgcn_sched_barrier(0);
            MX3_STAMP(tM)
            E4S_LDS_BARRIER();
            MX3_STAMP(tWM)
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == 2 ? 0 : slot + 1;
            if (S2 && last && have_next) {
            // ---------------- store phase: the patch has ONE buffer, so between two chunks nobody reads.  Both groups convert at the same time — waves 0-3 after
            // their barrier, waves 4-7 (above) behind their last MFMA phase, in front of the same barrier — and one more barrier lets waves 0-3 run ahead
            // again.  (The prefetch registers are refilled at the top of the next iteration: see there.)
            // (measured against the serial form — waves 0-3 convert during waves 4-7's last MFMA phase, then waves 4-7, two barriers: 0.1498 vs 0.1518 ms)
            if (!grp) store_x(nchunk_i, nph);
            E4S_LDS_BARRIER();
            MX3_STAMP(tST)
            }
        }
        if constexpr (!S2) {
            if (more) {          // (stride 1: the store phase between two chunks, as above)
                if (!grp) store_x(chunk + 1, 0);
                E4S_LDS_BARRIER();
                MX3_STAMP(tST)
            }
        }
    }
    if (!grp) E4S_LDS_BARRIER();
    E4S_WAIT_VM(0);
#ifdef MX3_PROF
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0)
        printf("wave %d: prologue %llu  loop %llu cyc | R %llu  wait after R %llu  M %llu  wait after M %llu | store: vm wait %llu  store_x %llu  barrier 1 %llu  barrier 2 %llu  (units %d)\n", wave, tLoop - tKernel, __builtin_readcyclecounter() - tLoop, tR, tWR, tM, tWM, tS1, tS2, tS3, tST, nunits);
#endif
    auto report_ovf = [&]() __attribute__((always_inline)) {
        if (p.flags && __builtin_amdgcn_ballot_w64(ovf != 0u) != 0 && (threadIdx.x & 63) == 0) { atomicOr(p.flags, 1); atomicAdd(p.flags + 1, 1); }   // one report per wave: sticky bit + moving counter (ops.MxGuard)
    };
    report_ovf();
    ovf = 0u;

    // ---- epilogue: PReLU, stores (lane = pixel column: consecutive lanes write consecutive floats of one channel plane)
    __syncthreads();
    float* sl = reinterpret_cast<float*>(lds);
    if (tid < TN) sl[tid] = (p.slope && co0 + tid < p.cout) ? p.slope[co0 + tid] : 1.f;
    __syncthreads();
    const int x = x0 + l5;
    if constexpr (!S2) {
        if (p.out_prep) {
            // PREPARED OPERANDS for the consuming convolution (see IN == 2): a lane holds 16 of a pixel's 32 channels of a block (those of its k half) for each of
            // its two pixel rows; one v_permlane32_swap per register pair gives lanes 0-31 all 32 channels of row 0 and lanes 32-63 those of row 1
            const int y = y0 + 2 * pr + khalf;
            const size_t ohw = (size_t)hw;
            const bool ok = y < p.h && x < p.w_;
            const size_t pi = p.phased ? (size_t)(2 * (y & 1) + (x & 1)) * (ohw >> 2) + (size_t)(y >> 1) * (p.w_ >> 1) + (x >> 1) : (size_t)y * p.w_ + x;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                float a[32];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = chh * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    float v0 = acc[cb][0][r], v1 = acc[cb][1][r];
                    v0 = v0 > 0.f ? v0 : v0 * sl[n];
                    v1 = v1 > 0.f ? v1 : v1 * sl[n];
                    // lanes 32-63 of the first <-> lanes 0-31 of the second: [0] = (own row 0 | lower lanes' row 1), [1] = (upper lanes' row 0 | own row 1)
                    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1), false, false);
                    a[8 * (r >> 2) + (r & 3)] = __builtin_bit_cast(float, (unsigned)sw[0]);           // channels 8 q + i      (the k half 0 lanes' registers)
                    a[8 * (r >> 2) + 4 + (r & 3)] = __builtin_bit_cast(float, (unsigned)sw[1]);       // channels 8 q + 4 + i  (the k half 1 lanes')
                }
                u32x16 q1;
                u32x6 c1, c2;
                unsigned sc;
                encode32(a, q1, c1, c2, sc, ovf);
                const int kb = (co0 + chh * 64 + cb * 32) >> 5;
                if (ok && kb * 32 < p.cout) {
                    unsigned char* blk = reinterpret_cast<unsigned char*>(p.out) + ((size_t)b * (p.cout >> 5) + kb) * ((size_t)PREPB * ohw);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        *reinterpret_cast<uint4*>(blk + ((size_t)s4 * ohw + pi) * 16) = make_uint4(q1[4 * s4], q1[4 * s4 + 1], q1[4 * s4 + 2], q1[4 * s4 + 3]);
                    *reinterpret_cast<uint4*>(blk + ((size_t)4 * ohw + pi) * 16) = make_uint4(c1[0], c1[1], c1[2], c1[3]);
                    *reinterpret_cast<uint4*>(blk + ((size_t)5 * ohw + pi) * 16) = make_uint4(c2[0], c2[1], c2[2], c2[3]);
                    *reinterpret_cast<uint4*>(blk + ((size_t)6 * ohw + pi) * 16) = make_uint4(c1[4], c1[5], c2[4], c2[5]);
                    *reinterpret_cast<unsigned*>(blk + (size_t)112 * ohw + pi * 4) = sc;
                }
            }
            if (!ok) ovf = 0u;          // (a pixel outside the map: whatever the tile computed there is not part of the result)
            report_ovf();
            return;
        }
    }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int y = y0 + 2 * pr + pb;
        const int oh = S2 ? p.ho : p.h, ow = S2 ? p.wo : p.w_, ohw = S2 ? p.ho * p.wo : hw;      // (stride 1: the same registers as the input's)
        if (y < oh && x < ow) {
            // (stride 1, phased: pixel (y, x) of a channel plane goes to plane 2 (y & 1) + (x & 1), position (y / 2, x / 2) — the layout the stride-2 form reads coalesced)
            const size_t pix = (!S2 && p.phased) ? (size_t)(2 * (y & 1) + (x & 1)) * (ohw >> 2) + (size_t)(y >> 1) * (ow >> 1) + (x >> 1) : (size_t)y * ow + x;
            float* op = p.out + (size_t)b * p.cout * ohw + (p.out_c4 ? pix * 4 : pix);
            if (p.out_c4) {
                // channel-blocked: registers 4 q .. 4 q + 3 of a block are channels 8 q + 4 khalf .. + 3 — one 16-byte element of the pixel (cout % 4 == 0: the launcher)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = chh * 64 + cb * 32 + 8 * q + 4 * khalf;
                        f32x4 v4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { const float v = acc[cb][pb][4 * q + i]; v4[i] = v > 0.f ? v : v * sl[n + i]; }
                        if (co0 + n < p.cout) *reinterpret_cast<f32x4*>(op + (size_t)(co0 + n) * ohw) = v4;      // (plane (co0 + n) / 4 starts at (co0 + n) ohw floats)
                    }
            } else {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = chh * 64 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    float v = acc[cb][pb][r];
                    v = v > 0.f ? v : v * sl[n];
                    if (co0 + n < p.cout) op[(size_t)(co0 + n) * ohw] = v;
                }
            }
        }
    }
#ifdef MX3_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) printf("wave %d: whole kernel %llu cyc\n", wave, __builtin_readcyclecounter() - tKernel);
#endif
}

}  // namespace

extern "C" int e4s_conv3x3_mx3_weight_bytes(int cout, int cin, int64_t* bytes) {
    E4S_REQUIRE(bytes && cout >= 1 && cin >= CK && cin % CK == 0, "conv3x3_mx3_weight_bytes: bad arguments (cin %% 32 == 0)");
    *bytes = (int64_t)(cin / CK) * cdiv(cout, TN) * NUNIT * UNITB;
    return 0;
}

// weight [cout][cin][3][3] fp32 -> unit slots (see the header)
static int prep_weights_mx3(void* dst, const float* weight, int cout, int cin, int s2, void* stream) {
    E4S_REQUIRE(dst && weight && cout >= 1 && cin >= CK && cin % CK == 0, "conv_prep_weights_mx3: bad arguments (cin %% 32 == 0)");
    E4S_REQUIRE(((uintptr_t)dst & 15) == 0, "conv_prep_weights_mx3: the destination must be 16-byte aligned");
    const int64_t total = (int64_t)(cin / CK) * cdiv(cout, TN) * NUNIT * 2 * TN;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(prep_weights_mx3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<unsigned char*>(dst), weight, cout, cin, s2);
    return check_launch("conv_prep_weights_mx3");
}

extern "C" int e4s_conv_prep_weights_mx3(void* dst, const float* weight, int cout, int cin, void* stream) {
    return prep_weights_mx3(dst, weight, cout, cin, 0, stream);
}

// the same bytes per layer (e4s_conv3x3_mx3_weight_bytes), units in the stride-2 kernel's tap order
extern "C" int e4s_conv_prep_weights_mx3_s2(void* dst, const float* weight, int cout, int cin, void* stream) {
    return prep_weights_mx3(dst, weight, cout, cin, 1, stream);
}

// out = PReLU(conv3x3(norm(x), W)), stride 1, pad 1.  in_mean / in_rstd [bs][cin] (optional, together), prelu_slope [cout] optional; flags[0] bit 0 is raised
// when a normalised activation leaves the f16 range (the result is then invalid; the kernel does not fall back by itself — ops.MxGuard notices the counter flags[1] moving and the entry points re-run the pass on the split-bf16 kernels).
// layouts: bit 0 = phase planes (stride 1: of the output; stride 2: of the input), bit 1 = channel-blocked [c / 4][...][4]
template <bool S2, int IN>
static int conv3x3_mx3(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                       int bs, int cin, int cout, int h, int w, int phased, int out_c4, int out_prep, void* stream) {
    E4S_REQUIRE(out && x && wmx3, "conv3x3_mx3: null tensor");
    E4S_REQUIRE((in_mean == nullptr) == (in_rstd == nullptr), "conv3x3_mx3: in_mean and in_rstd go together");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= CK && cin % CK == 0 && cin <= MAX_CIN && cout >= 1 && h >= 1 && w >= 1, "conv3x3_mx3: bad size (cin %% 32 == 0, cin <= 512)");
    E4S_REQUIRE(!(S2 || phased) || (h % 2 == 0 && w % 2 == 0), "conv3x3_mx3: stride 2 / phase planes need an even height and width");
    E4S_REQUIRE(((uintptr_t)wmx3 & 15) == 0, "conv3x3_mx3: the weights must be 16-byte aligned");
    E4S_REQUIRE((int64_t)cin * h * w * 4 < (int64_t)1 << 32, "conv3x3_mx3: a sample's input must stay below 4 GB (32-bit lane offsets)");
    E4S_REQUIRE(!out_c4 || cout % 4 == 0, "conv3x3_mx3: a channel-blocked output needs cout %% 4 == 0");
    E4S_REQUIRE(!(IN || out_c4 || out_prep) || ((((uintptr_t)x | (uintptr_t)out) & 15) == 0), "conv3x3_mx3: channel-blocked / prepared maps must be 16-byte aligned");
    E4S_REQUIRE(!(IN == 2) || (in_mean == nullptr && (h * w) % 4 == 0), "conv3x3_mx3: a prepared-operand input takes no normalisation and needs h * w %% 4 == 0");
    E4S_REQUIRE(!out_prep || (!S2 && !out_c4 && cout % 32 == 0 && (h * w) % 4 == 0), "conv3x3_mx3: prepared-operand output: stride 1, cout %% 32 == 0, h * w %% 4 == 0");
    E4S_REQUIRE((int64_t)PREPB * h * w < (int64_t)1 << 32, "conv3x3_mx3: map too large for 32-bit lane offsets");
    if (bs == 0) return 0;
    Mx3Params p;
    memset(&p, 0, sizeof(p));
    p.out = out; p.x = x; p.w = reinterpret_cast<const unsigned char*>(wmx3); p.flags = flags;
    p.in_mean = in_mean; p.in_rstd = in_rstd; p.slope = prelu_slope;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w_ = w;
    p.ho = S2 ? h / 2 : h; p.wo = S2 ? w / 2 : w;
    p.phased = phased ? 1 : 0;
    p.out_c4 = out_c4 ? 1 : 0;
    p.out_prep = out_prep ? 1 : 0;
    p.tiles_x = cdiv(p.wo, TW); p.tiles_y = cdiv(p.ho, TH);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mx3_kernel<S2, IN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return fail((int)attr, "conv3x3_mx3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    dim3 grid(p.tiles_x * p.tiles_y, cdiv(cout, TN), bs);
    hipLaunchKernelGGL((conv3x3_mx3_kernel<S2, IN>), grid, dim3(512), LDS_BYTES, (hipStream_t)stream, p);
    return check_launch(S2 ? "conv3x3_s2_mx3" : "conv3x3_mx3");
}

extern "C" int e4s_conv3x3_mx3(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                               int bs, int cin, int cout, int h, int w, void* stream) {
    return conv3x3_mx3<false, 0>(out, x, wmx3, flags, in_mean, in_rstd, prelu_slope, bs, cin, cout, h, w, 0, 0, 0, stream);
}

// e4s_conv3x3_mx3 whose result is written as PHASE PLANES, out[b][c][2 py + px][h / 2][w / 2] = result[b][c][2 y + py][2 x + px] (h, w even): the layout
// e4s_conv3x3_s2_mx3(in_phased = 1) reads with consecutive lanes on consecutive floats
extern "C" int e4s_conv3x3_mx3_phased(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                                      int bs, int cin, int cout, int h, int w, void* stream) {
    return conv3x3_mx3<false, 0>(out, x, wmx3, flags, in_mean, in_rstd, prelu_slope, bs, cin, cout, h, w, 1, 0, 0, stream);
}

// e4s_conv3x3_mx3 with explicit memory layouts (round 5).  Bit 1 (value 2) of either = CHANNEL-BLOCKED: the map is [bs][c / 4][h][w][4 floats] (c % 4 == 0, 16-byte
// aligned) — a pixel's four channels are one 16-byte element, so the consumer's patch threads request 8 elements per 32-channel chunk instead of 32 floats and the
// producer stores 16 bytes per request.  Bit 0 (value 1) of out_layout = phase planes as e4s_conv3x3_mx3_phased ([bs][c (/ 4)][2 py + px][h / 2][w / 2]([4])).
// Bit 2 (value 4) = PREPARED OPERANDS (alone, or 5 = in phase-plane pixel order): per image and 32-channel block 116 h w bytes — a1 [slot 4][pixel] x 16 | codes [term 2]
// [pixel] x 16 | code tails [pixel] x 16 | scales [pixel] x 4 — the entries the consumer's staging would compute from the fp32 map (f16 part, two fp6 code sets, their
// block scales), computed once per pixel in the producer's epilogue instead (cout % 32 == 0; as an input: no in_mean / in_rstd).
// in_layout: 0, 2 or 4.  The arithmetic does not depend on the layouts: the same values as e4s_conv3x3_mx3, bit for bit.
extern "C" int e4s_conv3x3_mx3_ex(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                                  int bs, int cin, int cout, int h, int w, int in_layout, int out_layout, void* stream) {
    E4S_REQUIRE((in_layout == 0 || in_layout == 2 || in_layout == 4) && out_layout >= 0 && out_layout <= 5, "conv3x3_mx3_ex: in_layout is 0, 2 or 4, out_layout 0 .. 5");
    const int ph = out_layout & 1, c4 = out_layout & 2, prep = out_layout & 4;
    if (in_layout == 4) return conv3x3_mx3<false, 2>(out, x, wmx3, flags, in_mean, in_rstd, prelu_slope, bs, cin, cout, h, w, ph, c4, prep, stream);
    if (in_layout == 2) return conv3x3_mx3<false, 1>(out, x, wmx3, flags, in_mean, in_rstd, prelu_slope, bs, cin, cout, h, w, ph, c4, prep, stream);
    return conv3x3_mx3<false, 0>(out, x, wmx3, flags, in_mean, in_rstd, prelu_slope, bs, cin, cout, h, w, ph, c4, prep, stream);
}

// out [bs, cout, h / 2, w / 2] = PReLU(conv3x3(norm(x), W, stride 2, pad 1)); h, w even; weights from e4s_conv_prep_weights_mx3_s2; in_phased: x is in the phase-plane
// layout of e4s_conv3x3_mx3_phased; in_phased is a layout word like e4s_conv3x3_mx3_ex's (bit 0: phase planes, bit 1: channel-blocked).  Everything else as e4s_conv3x3_mx3.
extern "C" int e4s_conv3x3_s2_mx3(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                                  int bs, int cin, int cout, int h, int w, int in_phased, void* stream) {
    E4S_REQUIRE(in_phased >= 0 && in_phased <= 5, "conv3x3_s2_mx3: in_phased is a layout, 0 .. 5");
    if (in_phased & 4) return conv3x3_mx3<true, 2>(out, x, wmx3, flags, in_mean, in_rstd, prelu_slope, bs, cin, cout, h, w, in_phased & 1, 0, 0, stream);
    if (in_phased & 2) return conv3x3_mx3<true, 1>(out, x, wmx3, flags, in_mean, in_rstd, prelu_slope, bs, cin, cout, h, w, in_phased & 1, 0, 0, stream);
    return conv3x3_mx3<true, 0>(out, x, wmx3, flags, in_mean, in_rstd, prelu_slope, bs, cin, cout, h, w, in_phased & 1, 0, 0, stream);
}
