// a3/a4, round 5: the region-uniform 16 x 16 output blocks of a masked up layer (modconv_upblock.hip, round 2) on the f16 + 2 x MX-fp6 arithmetic with operands
// prepared ONCE at staging.
//
// Reference: ModulatedConv2d.forward's upsample branch (models/stylegan2/model.py:287-300: conv_transpose2d stride 2, then Blur) under StyledConv.forward's per-region
// mixing (:385-400).  Inside a block whose 16 x 16 output pixels all carry region r the layer is the single-region form
//     out = lrelu( d[r] * blur( conv_transpose(x * s[r], W, stride 2) ) + noise_weight * noise + act_bias ) * sqrt(2)
// at 1x the transposed conv's MACs (x 2.0 for the block's halo and the idle lanes of its 100 positions in 128) instead of the composed form's 4x.
//
// Why a second kernel.  masked_up_block_kernel<2, 1> still runs the split-bf16 arithmetic (three bf16 MFMAs per product: 6 matrix-pipe units per algorithmic MAC at its
// 2.0x, against 4 x 1.667 = 6.67 of the composed f16 + fp6 kernels — the "fast path" was no faster than the general one: round-4 review) and splits every staged value
// per chunk on its way into LDS.  Here the block's one region makes the operand independent of the output pixel, so conv_mx3.hip's discipline applies as it is:
// x * s[r] -> a1 = f16, fp6(a1), fp6(x s - a1) with one block scale per patch pixel and 32-channel chunk, made once per staged value; the K loop reads LDS and issues
// MFMAs (1.667 units per MAC: 3.3 per algorithmic MAC at 2.0x).  The weights arrive by LDS-DMA (a ring of three tap-pair units of 15 KB) instead of through registers.
//
//   workgroup: a PAIR of horizontally adjacent blocks x 64 output channels, 512 threads = 2 groups of 4 waves x 32 positions (10 x 10 positions feed a block's 19 x 19
//              pre-blur window; a position owns z[2a + i][2b + j]); the two groups share ONE weight ring of six 15 KB units.  (The first version — one block per
//              256-thread workgroup, two per CU, a ring of three — was 5 % faster than the round-2 kernel, not 40 %: a block streams 600 KB of weights for 100
//              positions, 1.2 GB of LDS-DMA per launch, at the ~5 TB/s that 60 KB in flight per CU sustain (in flight / latency): the kernel was bound by weight
//              ingest, like its predecessor.  A pair halves the bytes per MFMA to the other kernels' 160 B and the deeper ring keeps 75 KB in flight.)
//   K loop:    chunk = 32 input channels; five units = tap pairs that share an accumulator parity — (0,0)|(0,2) and (2,0)|(2,2) -> (0,0); (1,0)|(1,2) -> (1,0);
//              (0,1)|(2,1) -> (0,1); (1,1) -> (1,1) — 8 f16 + 4 fp6 MFMAs per wave and unit, one barrier per unit (the ring slot's hand-over)
//   epilogue:  the round-2 kernel's (8 channels at a time through a pre-blur tile in LDS, 4 x 4 FIR with a rolling four-row window, demodulation / noise / bias /
//              leaky ReLU, fp32 NCHW stores)
// f16 range: a wave that sees |x s| >= 65520 raises flags[0] bit 0 and bumps flags[1] (ops.MxGuard re-runs the pass on the split-bf16 kernels).
#include "modconv_mx_tile.h"

// Tuning builds only (-DUX_ABL=bits; results are then meaningless): 1 = no K loop, 2 = no epilogue, 4 = no activation loads / conversions after the first chunk
#ifndef UX_ABL
#define UX_ABL 0
#endif

namespace {

constexpr int UX_OUT = 16;
constexpr int UX_TN = 64, UX_CK = 32;
constexpr int UX_T = UX_OUT / 2 + 2;              // 10 positions per side
constexpr int UX_PW = UX_T + 1;                   // 11: activation patch side
constexpr int UX_PATCH = UX_PW * UX_PW;           // 121
constexpr int UX_NPOS = UX_T * UX_T;              // 100
constexpr int UX_EST = 128;                       // entry stride of the patch planes
constexpr int UX_ZR = 2 * UX_T, UX_ZS = UX_ZR + 2;
constexpr int UX_NUNIT = 5;
constexpr int UX_U_W16 = 2 * 2 * 2 * UX_TN * 16;  // 8 192
constexpr int UX_U_CLO = 2 * 2 * UX_TN * 16;      // 4 096
constexpr int UX_U_CHI = 2 * 2 * UX_TN * 8;       // 2 048
constexpr int UX_U_SC = 2 * UX_TN * 4;            // 512
constexpr int UX_UNITB = 15 * 1024;               // 14 848 used: a whole number of 1 KB DMA pieces
constexpr int UX_NPIECE = UX_UNITB / 1024;        // 15
static_assert(UX_U_W16 + UX_U_CLO + UX_U_CHI + UX_U_SC <= UX_UNITB, "unit slot");
// LDS plan
constexpr int UX_NSLOT = 6;                                   // ring slots (units in flight: five)
constexpr int UX_RING = 0;
constexpr int UX_GRP0 = UX_RING + UX_NSLOT * UX_UNITB;        // 92 160: per block of the pair (group = wave / 4):
constexpr int UX_A1 = 0;                                      //   a1 f16 [16-B slot 4][pixel 128]
constexpr int UX_CLO = UX_A1 + 4 * UX_EST * 16;
constexpr int UX_CHI = UX_CLO + 2 * UX_EST * 16;
constexpr int UX_SC = UX_CHI + 2 * UX_EST * 8;
constexpr int UX_SROW = UX_SC + UX_EST * 4;                   //   the block's modulation row s[r][cin] (cin <= 512)
constexpr int UX_MAX_CIN = 512;
constexpr int UX_EP = UX_SROW + UX_MAX_CIN * 4;               //   d [64] | bias [64] | noise [256]
constexpr int UX_RAW = UX_EP + (2 * UX_TN + UX_OUT * UX_OUT) * 4;       //   the NEXT chunk's raw activations [channel 32][pixel 128] fp32, landed by LDS-DMA
constexpr int UX_GRPB = UX_RAW + UX_CK * UX_EST * 4;          // 34 816 per group
constexpr int UX_LDS = UX_GRP0 + 2 * UX_GRPB;                 // 161 792
constexpr int UX_ZTB = 16 * UX_ZR * UX_ZS * 4;                // 28 160: a group's pre-blur tile of 16 channels (epilogue, over the weight ring)
static_assert(UX_LDS <= 160 * 1024 && 2 * UX_ZTB <= UX_GRP0 && UX_GRPB % 16 == 0, "LDS plan; the pre-blur tiles overlay the weight ring");

struct UpBlockMxParams {
    float* out;
    const float* x;
    const unsigned char* wmx;  // [cin / 32][cout / 64][unit 5] x UX_UNITB (e4s_modconv_prep_weights_upblock_mx)
    const float* s;            // [bs][nreg][cin]
    const float* d;            // [bs][nreg][cout]
    const uint8_t* blocks;     // [bs][nby][nbx]: region of a uniform block, anything >= nreg: not this kernel's
    const int* ctrl;           // ctrl[2] == 0: the layer stays in the composed form (e4s_uniform_blocks)
    const float* blur;         // [4][4]
    const float* noise;
    const float* noise_weight;
    const float* act_bias;
    int* flags;
    int noise_bstride, act;
    int bs, cin, cout, h, w, nreg;
    int nbx, nby;
    unsigned perm_mul;
};

__device__ __forceinline__ void ux_wait_vm(int n) {      // s_waitcnt vmcnt(n) for a wave-uniform n (anything unexpected waits for everything)
    switch (n) {
#define UX_CASE(v) case v: E4S_WAIT_VM(v); break;
        UX_CASE(0) UX_CASE(1) UX_CASE(2) UX_CASE(3) UX_CASE(4) UX_CASE(5) UX_CASE(6) UX_CASE(8) UX_CASE(10)
        UX_CASE(32) UX_CASE(33) UX_CASE(34) UX_CASE(35) UX_CASE(36) UX_CASE(38) UX_CASE(40)
#undef UX_CASE
        default: E4S_WAIT_VM(0); break;
    }
}

// unit u, K half d -> tap (ky, kx) of the 3 x 3 kernel (ky = 3: none, zero weights) and the unit's accumulator parity 2 (ky & 1) + (kx & 1)
__host__ __device__ constexpr int ux_tap_ky(int u, int d) { return u == 0 ? 0 : u == 1 ? 2 : u == 2 ? 1 : u == 3 ? (d ? 2 : 0) : (d ? 3 : 1); }
__host__ __device__ constexpr int ux_tap_kx(int u, int d) { return u == 0 ? (d ? 2 : 0) : u == 1 ? (d ? 2 : 0) : u == 2 ? (d ? 2 : 0) : 1; }
__host__ __device__ constexpr int ux_parity(int u) { return u == 0 ? 0 : u == 1 ? 0 : u == 2 ? 2 : u == 3 ? 1 : 3; }

// ============================================================================ weight preparation: transposed-conv taps (NOT blur-composed) as tap-pair units
__global__ __launch_bounds__(256) void prep_weights_upblock_mx_kernel(unsigned char* __restrict__ dst, const float* __restrict__ weight, int cout, int cin, float scale) {
    const int nchunk = cin / UX_CK, ntile = (cout + UX_TN - 1) / UX_TN;
    const int64_t total = (int64_t)nchunk * ntile * UX_NUNIT * 2 * UX_TN;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int n = (int)(r % UX_TN); r /= UX_TN;
        const int half = (int)(r & 1); r >>= 1;
        const int unit = (int)(r % UX_NUNIT); r /= UX_NUNIT;
        const int tile = (int)(r % ntile);
        const int chunk = (int)(r / ntile);
        const int co = tile * UX_TN + n;
        const int ky = ux_tap_ky(unit, half), kx = ux_tap_kx(unit, half);
        unsigned char* slot = dst + (((size_t)chunk * ntile + tile) * UX_NUNIT + unit) * UX_UNITB;
        u32x16 q1, q2;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float a = 0.f, b = 0.f;
            if (ky < 3 && co < cout) {
                a = weight[((size_t)co * cin + chunk * UX_CK + 2 * j) * 9 + ky * 3 + kx] * scale;
                b = weight[((size_t)co * cin + chunk * UX_CK + 2 * j + 1) * 9 + ky * 3 + kx] * scale;
            }
            const f16x2 hh = __builtin_convertvector((f32x2){a, b}, f16x2);
            const float ra = (a - (float)hh[0]) * 4096.f, rb = (b - (float)hh[1]) * 4096.f;      // (the residual goes through f16 scaled by 2^12: conv_mx3.hip)
            q1[j] = __builtin_bit_cast(unsigned, hh);
            q2[j] = pack_f16_rne(ra, rb);
            m1 = fmaxf(m1, fmaxf(fabsf((float)hh[0]), fabsf((float)hh[1])));
            m2 = fmaxf(m2, fmaxf(fabsf(ra), fabsf(rb)));
        }
        uint4* w16 = reinterpret_cast<uint4*>(slot);
#pragma unroll
        for (int s = 0; s < 4; ++s) w16[(half * 4 + s) * UX_TN + n] = make_uint4(q1[4 * s], q1[4 * s + 1], q1[4 * s + 2], q1[4 * s + 3]);
        auto expo = [](float m) { const unsigned ex = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu; return ex > 3u ? ex - 2u : 1u; };
        const unsigned e1 = expo(m1), e2 = expo(m2);
        const u32x6 c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
        const u32x6 c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
        uint4* clo = reinterpret_cast<uint4*>(slot + UX_U_W16);
        uint2* chi = reinterpret_cast<uint2*>(slot + UX_U_W16 + UX_U_CLO);
        unsigned* scp = reinterpret_cast<unsigned*>(slot + UX_U_W16 + UX_U_CLO + UX_U_CHI);
        clo[(0 * 2 + half) * UX_TN + n] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
        chi[(0 * 2 + half) * UX_TN + n] = make_uint2(c1[4], c1[5]);
        clo[(1 * 2 + half) * UX_TN + n] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
        chi[(1 * 2 + half) * UX_TN + n] = make_uint2(c2[4], c2[5]);
        const unsigned e2s = e2 > 12u ? e2 - 12u : 0u;
        scp[half * UX_TN + n] = e1 | (e2s << 8);
    }
}

// ============================================================================ the kernel
__global__ __launch_bounds__(512, 2) void masked_up_block_mx_kernel(const UpBlockMxParams p) {
    const int npx = p.nbx >> 1;                                 // block pairs per row
    const int pr_i = (int)(((unsigned long long)blockIdx.x * p.perm_mul) % gridDim.x);
    const int tyt = pr_i / npx, txp = pr_i - tyt * npx;
    const int b = blockIdx.z;
    if (p.ctrl[2] == 0) return;
    const int reg_a = p.blocks[((size_t)b * p.nby + tyt) * p.nbx + 2 * txp], reg_b = p.blocks[((size_t)b * p.nby + tyt) * p.nbx + 2 * txp + 1];
    if (reg_a >= p.nreg && reg_b >= p.nreg) return;             // neither block is region-uniform

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3;                   // block of the pair, wave inside its group
    const int lt = tid & 255;                                   // thread inside the group
    const int txt = 2 * txp + grp;
    const int reg = grp ? reg_b : reg_a;
    const bool live = reg < p.nreg;                             // (a pair with one mixed block: that group only keeps the barriers company)
    unsigned char* gl = lds + UX_GRP0 + grp * UX_GRPB;          // this group's planes
    const int l5 = lane & 31, khalf = lane >> 5;
    const int cot = blockIdx.y, co0 = cot * UX_TN;
    const int ncot = (p.cout + UX_TN - 1) / UX_TN;
    const int hw = p.h * p.w, ho = 2 * p.h, wo = 2 * p.w;
    const int nchunk = p.cin / UX_CK, nunits = nchunk * UX_NUNIT;
    const float* xb = p.x + (size_t)b * p.cin * hw;

    // ---- weight ring: unit g -> slot g % 6, 15 pieces of 1 KB over the eight waves (waves 0-6: two, wave 7: one)
    const unsigned char* wbase = p.wmx + (size_t)cot * UX_NUNIT * UX_UNITB;
    auto dma_unit = [&](int g, int slot) __attribute__((always_inline)) {
        const int chunk = g / UX_NUNIT, u = g - chunk * UX_NUNIT;
        const unsigned char* src = wbase + ((size_t)chunk * ncot * UX_NUNIT + u) * UX_UNITB;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wave + 8 * k;
            if (piece < UX_NPIECE) dma16_asm(src + piece * 1024, (unsigned)(lane * 16), (unsigned)(UX_RING + slot * UX_UNITB + piece * 1024));
        }
    };

    // ---- staging thread = patch pixel (threads 0..120 of a group: its waves 0 and 1)
    const bool has_x = gw < 2 && live;
    const int PCS = wave < 7 ? 2 : 1;
    const int NLD = has_x ? UX_CK : 0;
    auto patch_pixel = [&](bool& in) __attribute__((always_inline)) {
        int t = lt;
        pin_here(t);
        const int se_y = t / UX_PW, se_x = t - se_y * UX_PW;
        const int sgy = tyt * (UX_OUT / 2) - 2 + se_y, sgx = txt * (UX_OUT / 2) - 2 + se_x;
        in = t < UX_PATCH && sgy >= 0 && sgy < p.h && sgx >= 0 && sgx < p.w;
        return in ? (unsigned)(sgy * p.w + sgx) * 4u : 0u;
    };
    // the next chunk's activations go straight to LDS (global_load_lds_dword: one channel x this wave's 64 patch pixels per request, per-lane addresses) instead of
    // through 32 registers per thread — the accumulators (128) and a unit's operands leave no room for them; the conversion reads them back at the chunk's end
    const unsigned raw_dst = (unsigned)(UX_GRP0 + grp * UX_GRPB + UX_RAW + gw * 64 * 4);
    auto load_x = [&](int chunk) __attribute__((always_inline)) {
        if (has_x) {
            bool in;
            const unsigned goff = patch_pixel(in);
#pragma unroll
            for (int c = 0; c < UX_CK; ++c) dma4_asm(xb + (size_t)(chunk * UX_CK + c) * hw, goff, raw_dst + (unsigned)(c * UX_EST * 4));
        }
    };
    unsigned ovf = 0u;
    auto store_x = [&](int chunk) __attribute__((always_inline)) {
        if (!has_x) return;
        bool in;
        (void)patch_pixel(in);
        const float4* st = reinterpret_cast<const float4*>(gl + UX_SROW) + chunk * (UX_CK / 4);
        float xr[UX_CK];
        {
            int t = lt;
            pin_here(t);
            const float* raw = reinterpret_cast<const float*>(gl + UX_RAW) + t;
#pragma unroll
            for (int c = 0; c < UX_CK; ++c) xr[c] = raw[c * UX_EST];
        }
        u32x16 q1, q2;
        unsigned m = 0u;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 s4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k] = st[half * 4 + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c4 = half * 4 + k;
                const float a0 = xr[4 * c4] * s4[k].x, a1v = xr[4 * c4 + 1] * s4[k].y, a2v = xr[4 * c4 + 2] * s4[k].z, a3 = xr[4 * c4 + 3] * s4[k].w;
                q1[2 * c4] = pack_f16_rne(a0, a1v);
                q1[2 * c4 + 1] = pack_f16_rne(a2v, a3);
                q2[2 * c4] = resid_pair_f16(xr[4 * c4], s4[k].x, xr[4 * c4 + 1], s4[k].y, q1[2 * c4]);
                q2[2 * c4 + 1] = resid_pair_f16(xr[4 * c4 + 2], s4[k].z, xr[4 * c4 + 3], s4[k].w, q1[2 * c4 + 1]);
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                u16x2 mm = __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, q1[2 * c4] & 0x7fff7fffu));
                mm = __builtin_elementwise_max(mm, __builtin_bit_cast(u16x2, q1[2 * c4 + 1] & 0x7fff7fffu));
                m = __builtin_bit_cast(unsigned, mm);
            }
        }
        if (!in) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { q1[j] = 0u; q2[j] = 0u; }
            m = 0u;
        }
        const unsigned mh = (m & 0xffffu) > (m >> 16) ? (m & 0xffffu) : (m >> 16);
        const unsigned e16 = mh >> 10;
        ovf |= e16 >= 31u ? 1u : 0u;
        const unsigned ex = (e16 ? e16 : 1u) + 112u;
        const unsigned e1 = ex > 3u ? ex - 2u : 1u, e2 = ex > 14u ? ex - 13u : 1u;
        const u32x6 c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
        const u32x6 c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
        int t = lt;
        pin_here(t);
        if (t < UX_PATCH) {
            uint4* a1p = reinterpret_cast<uint4*>(gl + UX_A1);
#pragma unroll
            for (int s = 0; s < 4; ++s) a1p[s * UX_EST + t] = make_uint4(q1[4 * s], q1[4 * s + 1], q1[4 * s + 2], q1[4 * s + 3]);
            uint4* clo = reinterpret_cast<uint4*>(gl + UX_CLO);
            uint2* chi = reinterpret_cast<uint2*>(gl + UX_CHI);
            clo[t] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
            chi[t] = make_uint2(c1[4], c1[5]);
            clo[UX_EST + t] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
            chi[UX_EST + t] = make_uint2(c2[4], c2[5]);
            reinterpret_cast<unsigned*>(gl + UX_SC)[t] = e1 | (e2 << 8);
        }
    };

    // ---- the block's modulation row and epilogue operands (plain loads: counted by the compiler, landed at the wait below)
    if (live) {
        const float* srow = p.s + ((size_t)b * p.nreg + reg) * p.cin;
        float* sl = reinterpret_cast<float*>(gl + UX_SROW);
        for (int c = lt; c < p.cin; c += 256) sl[c] = srow[c];
        float* ep_d = reinterpret_cast<float*>(gl + UX_EP);
        float* ep_b = ep_d + UX_TN;
        float* ep_n = ep_b + UX_TN;
        if (lt < UX_TN) {
            const int co = co0 + lt;
            ep_d[lt] = (co < p.cout && p.d) ? p.d[((size_t)b * p.nreg + reg) * p.cout + co] : 1.f;
            ep_b[lt] = (co < p.cout && p.act_bias) ? p.act_bias[co] : 0.f;
        }
        {
            const int ny = tyt * UX_OUT + (lt >> 4), nx = txt * UX_OUT + (lt & 15);
            ep_n[lt] = (p.noise && ny < ho && nx < wo) ? p.noise_weight[0] * p.noise[(size_t)b * p.noise_bstride + (size_t)ny * wo + nx] : 0.f;
        }
    }
    // request order: (tables above: plain loads, consumed already) the first chunk's activations, then the first six weight units — the wait below lets units 1 .. 5 stay
    // in flight: the first unit's operands and the activations are what the first MFMAs need
    load_x(0);
#pragma unroll
    for (int k = 0; k < UX_NSLOT; ++k)
        if (k < nunits) dma_unit(k, k);
    ux_wait_vm(((nunits < UX_NSLOT ? nunits : UX_NSLOT) - 1) * PCS);
    E4S_LDS_BARRIER();
    store_x(0);
    E4S_LDS_BARRIER();

    // this lane's position (gw * 32 + l5 of the group's 100) and its patch element
    const int pos = gw * 32 + l5;
    const bool pos_ok = pos < UX_NPOS;
    const int posc = pos_ok ? pos : UX_NPOS - 1;
    const int pty = posc / UX_T, ptx = posc - pty * UX_T;
    const int xoff = pty * UX_PW + ptx;

    f32x16 accs[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[a][i][r] = 0.f;

    // One phase per unit: operands, MFMAs, the wait for the next unit's weights, one barrier (the ring slot's hand-over); the next chunk's conversion behind the chunk's
    // last barrier.  (A two-phase schedule — group 1 half a unit behind group 0, R / M phases, each group converting behind its own last M phase — was built and measured
    // in this round: correct, 307 against 281 us on the 128 -> 256 layer.  Neither schedule is what bounds this tile: see the note on weight ingest in the header.)
    const unsigned char* wl = lds + UX_RING + (khalf * UX_TN + l5) * 16;
    int slot = 0, g = 0;
#pragma unroll 1
    for (int chunk = 0; chunk < ((UX_ABL & 1) ? 0 : nchunk); ++chunk) {
        const bool more = chunk + 1 < nchunk;
        if (more && !(UX_ABL & 4)) load_x(chunk + 1);
#pragma unroll
        for (int u = 0; u < UX_NUNIT; ++u, ++g) {
            // the slot unit g - 1 left (every wave passed that unit's barrier) takes unit g - 1 + NSLOT
            if (g >= 1 && g - 1 + UX_NSLOT < nunits) dma_unit(g - 1 + UX_NSLOT, slot == 0 ? UX_NSLOT - 1 : slot - 1);
            // patch elements of the unit's two taps: position (pty, ptx) reads x[a - (ky >> 1)] = patch row pty + 1 - (ky >> 1)
            // (recomputed from an opaque copy in every unit: as loop invariants the five units' fifteen LDS addresses would sit in registers the loop needs)
            int xo = xoff;
            pin_here(xo);
            const int e0 = xo + (1 - (ux_tap_ky(u, 0) >> 1)) * UX_PW + (1 - (ux_tap_kx(u, 0) >> 1));
            const int e1 = u == UX_NUNIT - 1 ? e0 : xo + (1 - (ux_tap_ky(u, 1) >> 1)) * UX_PW + (1 - (ux_tap_kx(u, 1) >> 1));
            const int ai = ux_parity(u);
            const unsigned char* ws = wl + slot * UX_UNITB;
            if (live) {
                {   // f16 part: a1 x w1
                    uint4 xa[2][2], wv[2][2][2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        xa[0][j] = *reinterpret_cast<const uint4*>(gl + UX_A1 + ((2 * j + khalf) * UX_EST + e0) * 16);
                        xa[1][j] = *reinterpret_cast<const uint4*>(gl + UX_A1 + ((2 * j + khalf) * UX_EST + e1) * 16);
                    }
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int d = 0; d < 2; ++d)
#pragma unroll
                            for (int j = 0; j < 2; ++j) wv[cb][d][j] = *reinterpret_cast<const uint4*>(ws + ((d * 2 + j) * 2 * UX_TN + cb * 32) * 16);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        if (d == 1 && u == UX_NUNIT - 1) break;          // (unit 4 has one tap)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb)
                                accs[ai][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv[cb][d][j]), __builtin_bit_cast(f16x8, xa[d][j]), accs[ai][cb], 0, 0, 0);
                    }
                }
                {   // fp6 part: fp6(w - w1) x fp6(a1) and fp6(w1) x fp6(a - a1)
                    const int ek = khalf ? e1 : e0;
                    uint4 calo[2], wclo[2][2];
                    uint2 cahi[2], wchi[2][2];
                    int scw[2];
#pragma unroll
                    for (int term = 0; term < 2; ++term) {
                        calo[term] = *reinterpret_cast<const uint4*>(gl + UX_CLO + (term * UX_EST + ek) * 16);
                        cahi[term] = *reinterpret_cast<const uint2*>(gl + UX_CHI + (term * UX_EST + ek) * 8);
                    }
                    const int sca = *reinterpret_cast<const int*>(gl + UX_SC + ek * 4);
                    const unsigned char* wc = lds + UX_RING + slot * UX_UNITB + UX_U_W16 + (khalf * UX_TN + l5) * 16;
                    const unsigned char* wh = lds + UX_RING + slot * UX_UNITB + UX_U_W16 + UX_U_CLO + (khalf * UX_TN + l5) * 8;
                    const unsigned char* wsc = lds + UX_RING + slot * UX_UNITB + UX_U_W16 + UX_U_CLO + UX_U_CHI + (khalf * UX_TN + l5) * 4;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                        for (int term = 0; term < 2; ++term) {
                            wclo[cb][term] = *reinterpret_cast<const uint4*>(wc + (term * 2 * UX_TN + cb * 32) * 16);
                            wchi[cb][term] = *reinterpret_cast<const uint2*>(wh + (term * 2 * UX_TN + cb * 32) * 8);
                        }
                        scw[cb] = *reinterpret_cast<const int*>(wsc + cb * 32 * 4);
                    }
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        accs[ai][cb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(wclo[cb][1], wchi[cb][1]), mx_op6(calo[0], cahi[0]), accs[ai][cb], 2, 2, 1, scw[cb], 0, sca);
                        accs[ai][cb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(wclo[cb][0], wchi[cb][0]), mx_op6(calo[1], cahi[1]), accs[ai][cb], 2, 2, 0, scw[cb], 1, sca);
                    }
                }
            }
            // Unit g + 1 must have landed before the next unit reads it.  Requests younger than it: the units requested after it (up to g - 1 + NSLOT, the last one
            // requested so far) and — while g + 1 belongs to this chunk, i.e. it was requested before the chunk's prefetch — the prefetch.
            {
                const int last_req = (g - 1 + UX_NSLOT < nunits - 1) ? (g >= 1 ? g - 1 + UX_NSLOT : UX_NSLOT - 1) : nunits - 1;
                const int younger = last_req > g + 1 ? last_req - (g + 1) : 0;
                ux_wait_vm(younger * PCS + ((u < UX_NUNIT - 1 && more && !(UX_ABL & 4)) ? NLD : 0));
            }
            E4S_LDS_BARRIER();
            slot = slot == UX_NSLOT - 1 ? 0 : slot + 1;
        }
        if (more) {                 // (every wave is past its last read of this chunk's patch; the prefetch is older than the unit just waited for: it has landed)
            if (!(UX_ABL & 4)) store_x(chunk + 1);
            E4S_LDS_BARRIER();
        }
    }
    E4S_WAIT_VM(0);
    if (p.flags && __builtin_amdgcn_ballot_w64(ovf != 0u) != 0 && lane == 0) { atomicOr(p.flags, 1); atomicAdd(p.flags + 1, 1); }

    if (UX_ABL & 2) { if (accs[0][0][0] == 12345.f) p.out[tid] = accs[1][1][1] + accs[2][0][2] + accs[3][1][3]; return; }
    // ---- epilogue: pre-blur tiles of 16 channels at a time (four passes), z[2 pty + i][2 ptx + j]; output pixel (y, x) = sum_{t,u} k[t][u] z[y + 1 + t][x + 1 + u] with a
    // rolling four-row window.  (The round-2 kernel's epilogue — 8 channels per pass, one FIR chain per thread in a loop the compiler must not unroll — cost 84 of this
    // kernel's 283 us on the 128 -> 256 layer: tuning builds.  Here a thread runs two independent chains, fully unrolled, and the workgroup synchronises half as often.)
    __syncthreads();
    float* zt = reinterpret_cast<float*>(lds + grp * UX_ZTB);       // this group's [16][ZR][ZS] over the weight ring
    const float* ep_d = reinterpret_cast<const float*>(gl + UX_EP);
    const float* ep_b = ep_d + UX_TN;
    const float* ep_n = ep_b + UX_TN;
    float kf[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = p.blur[15 - t];
    const int bx = lt & 15, bco = (lt >> 4) & 7, byg = (lt >> 7) & 1;
    const int oy0 = tyt * UX_OUT + 8 * byg, ox = txt * UX_OUT + bx;
    int nrow = ho - oy0;
    nrow = (!live || nrow < 0) ? 0 : (nrow > 8 ? 8 : nrow);
    const bool col_ok = ox < wo;
    const float* nzp = ep_n + 8 * byg * UX_OUT + bx;
    float nzr[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) nzr[r] = nzp[r * UX_OUT];
    const float neg = p.act ? 0.2f : 1.f, gain = p.act ? 1.41421356237309515f : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int gh = 0; gh < 2; ++gh) {               // channels i * 32 + 16 gh .. + 15
            if (pos_ok && live) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int col = 8 * g2 + 4 * khalf + rr;
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci)
                            *reinterpret_cast<float2*>(&zt[(col * UX_ZR + 2 * pty + ci) * UX_ZS + 2 * ptx]) =
                                make_float2(accs[2 * ci][i][4 * (2 * gh + g2) + rr], accs[2 * ci + 1][i][4 * (2 * gh + g2) + rr]);
                    }
            }
            __syncthreads();
            if (nrow > 0 && col_ok) {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int lc = bco + 8 * c2;                      // channel of the pass
                    const int cl = i * 32 + 16 * gh + lc;
                    const int co = co0 + cl;
                    if (co >= p.cout) continue;
                    const float dd = ep_d[cl], bi = ep_b[cl];
                    const float* zc = zt + (lc * UX_ZR + 8 * byg + 1) * UX_ZS + bx + 1;
                    float* orow = p.out + ((size_t)b * p.cout + co) * ho * wo + (size_t)oy0 * wo + ox;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3;
#pragma unroll
                    for (int zr = 0; zr < 8 + 3; ++zr) {
                        const float* zp = zc + zr * UX_ZS;
                        const float z0 = zp[0], z1 = zp[1], z2 = zp[2], z3 = zp[3];
                        a3 = 0.f;
                        a0 = __builtin_fmaf(z0, kf[12], a0); a1 = __builtin_fmaf(z0, kf[8], a1); a2 = __builtin_fmaf(z0, kf[4], a2); a3 = __builtin_fmaf(z0, kf[0], a3);
                        a0 = __builtin_fmaf(z1, kf[13], a0); a1 = __builtin_fmaf(z1, kf[9], a1); a2 = __builtin_fmaf(z1, kf[5], a2); a3 = __builtin_fmaf(z1, kf[1], a3);
                        a0 = __builtin_fmaf(z2, kf[14], a0); a1 = __builtin_fmaf(z2, kf[10], a1); a2 = __builtin_fmaf(z2, kf[6], a2); a3 = __builtin_fmaf(z2, kf[2], a3);
                        a0 = __builtin_fmaf(z3, kf[15], a0); a1 = __builtin_fmaf(z3, kf[11], a1); a2 = __builtin_fmaf(z3, kf[7], a2); a3 = __builtin_fmaf(z3, kf[3], a3);
                        if (zr >= 3 && zr - 3 < nrow) {
                            float v = __builtin_fmaf(a0, dd, bi) + nzr[zr - 3];
                            v = fmaxf(v, v * neg) * gain;
                            orow[(size_t)(zr - 3) * wo] = v;
                        }
                        a0 = a1; a1 = a2; a2 = a3;
                    }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace

extern "C" int e4s_upblock_mx_weight_bytes(int cout, int cin, int64_t* bytes) {
    E4S_REQUIRE(bytes && cout >= 1 && cin >= UX_CK && cin % UX_CK == 0, "upblock_mx_weight_bytes: bad arguments (cin %% 32 == 0)");
    *bytes = (int64_t)(cin / UX_CK) * cdiv(cout, UX_TN) * UX_NUNIT * UX_UNITB;
    return 0;
}

// weight [1 or none, cout, cin, 3, 3] fp32 (the transposed-conv taps, NOT blur-composed) -> tap-pair unit slots, equalised-lr scale 1 / sqrt(9 cin) folded in
extern "C" int e4s_modconv_prep_weights_upblock_mx(void* dst, const float* weight, int cout, int cin, void* stream) {
    E4S_REQUIRE(dst && weight && cout >= 1 && cin >= UX_CK && cin % UX_CK == 0, "modconv_prep_weights_upblock_mx: bad arguments (cin %% 32 == 0)");
    E4S_REQUIRE(((uintptr_t)dst & 15) == 0, "modconv_prep_weights_upblock_mx: the destination must be 16-byte aligned");
    (void)hipMemsetAsync(dst, 0, (size_t)(cin / UX_CK) * cdiv(cout, UX_TN) * UX_NUNIT * UX_UNITB, (hipStream_t)stream);      // (the pad bytes of every unit are DMA'd too)
    const int64_t total = (int64_t)(cin / UX_CK) * cdiv(cout, UX_TN) * UX_NUNIT * 2 * UX_TN;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(prep_weights_upblock_mx_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst, weight, cout, cin, 1.0f / sqrtf((float)cin * 9.f));
    return check_launch("modconv_prep_weights_upblock_mx");
}

// The region-uniform 16 x 16 output blocks of a masked up layer (blocks[b][by][bx] < nreg; everything else is left to the composed kernel with the same map), f16 + 2 x MX
// fp6 arithmetic.  wmx from e4s_modconv_prep_weights_upblock_mx; cin % 32 == 0, cin <= 512; flags as for e4s_region_modconv3x3_mx; the other arguments as e4s_masked_upconv_blocks.
extern "C" int e4s_masked_upconv_blocks_mx(float* out, const float* x, const void* wmx, int* flags, const float* s, const float* d, const uint8_t* blocks, const int* ctrl,
                                           const float* blur, const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act, int bs, int cin,
                                           int cout, int h, int w, int nreg, void* stream) {
    E4S_REQUIRE(out && x && wmx && s && blocks && ctrl && blur, "masked_upconv_blocks_mx: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= UX_CK && cin % UX_CK == 0 && cin <= UX_MAX_CIN && cout >= 1 && h >= 8 && w >= 8 && (h % 8) == 0 && (w % 8) == 0,
                "masked_upconv_blocks_mx: bad size (cin %% 32 == 0, cin <= 512, h multiple of 8, w multiple of 16)");
    E4S_REQUIRE(w % 16 == 0, "masked_upconv_blocks_mx: the width must be a multiple of 16 (pairs of 16 x 16 output blocks)");
    E4S_REQUIRE(nreg >= 1 && nreg <= E4S_MAX_REGIONS, "masked_upconv_blocks_mx: %d regions (max %d)", nreg, E4S_MAX_REGIONS);
    E4S_REQUIRE(!noise || (noise_weight && (noise_bs == 1 || noise_bs == bs)), "masked_upconv_blocks_mx: noise needs its weight and batch 1 or bs");
    E4S_REQUIRE(((uintptr_t)wmx & 15) == 0, "masked_upconv_blocks_mx: the weights must be 16-byte aligned");
    if (bs == 0) return 0;
    UpBlockMxParams p;
    memset(&p, 0, sizeof(p));
    p.out = out; p.x = x; p.wmx = reinterpret_cast<const unsigned char*>(wmx); p.flags = flags; p.s = s; p.d = d; p.blocks = blocks; p.ctrl = ctrl;
    p.blur = blur; p.noise = noise; p.noise_weight = noise_weight; p.act_bias = act_bias;
    p.noise_bstride = (noise && noise_bs == bs) ? 4 * h * w : 0; p.act = act;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.nreg = nreg;
    p.nbx = 2 * w / UX_OUT; p.nby = 2 * h / UX_OUT;
    p.perm_mul = coprime_stride((unsigned)((p.nbx / 2) * p.nby));
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&masked_up_block_mx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, UX_LDS);
    if (attr != hipSuccess) return fail((int)attr, "masked_upconv_blocks_mx: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    const dim3 grid((p.nbx / 2) * p.nby, cdiv(cout, UX_TN), bs);           // one workgroup per PAIR of blocks
    hipLaunchKernelGGL(masked_up_block_mx_kernel, grid, dim3(512), UX_LDS, (hipStream_t)stream, p);
    return check_launch("masked_upconv_blocks_mx");
}
