// a3/a4, round 5: the masked 3x3 modulated conv with CLASS-PREPARED operands ("entries") on the two-phase, conversion-free K loop of conv_mx3.hip.
//
// Reference: StyledConv.forward's per-region loop (models/stylegan2/model.py:385-400) over ModulatedConv2d.forward (:276-320), in the one-pass form of DESIGN.md section 2:
//     out[b,o,p] = d[b,c(p),o] * sum_{i,k} W[o,i,k] / sqrt(9 Cin) * s[b,c(p),i] * x[b,i,p+k]
//
// Why.  region_modconv_mx_kernel (modconv_mx_tile.h) multiplies the activation by its OUTPUT pixel's modulation between the LDS read and the MFMA: every staged value is
// modulated, split into f16 + residual and converted to fp6 NINE times (once per tap that reads it) — 11 VALU instructions per MFMA, 25 % matrix-pipe occupancy (round-4
// counters), and its six loop parts add up instead of overlapping.  The plain-convolution kernel of conv_mx3.hip does none of that in its loop (operands are made once per
// staged value, two wave groups alternate between an LDS-read phase and an MFMA-only phase) and runs the same arithmetic twice as fast.  What keeps the masked layer from
// that form is only that the operand depends on the consumer's region.  But a staged value has few DISTINCT consumers' regions: one inside a region, two along a border,
// up to four at a corner.  So the unit of staging here is an ENTRY = (patch pixel q, region c) for every region c that occurs among the <= 9 output pixels of the tile that
// read q: 1.0x the patch inside a region, 1.56x on 8-pixel cells, 9x in the worst case (i.i.d. labels) — against a fixed 9x per-tap cost in the old loop.  Entries are
// built once per workgroup (labels do not depend on the channel), prepared once per 32-channel chunk (x * s[c] -> a1 = f16, fp6(a1), fp6(x s - a1), block scales per
// entry), and an output lane reads tap t of ITS region through a per-lane LDS offset (a table written once per tile): the K loop has no floating-point VALU work at all.
//
//   tile:      32 x 8 output pixels x 128 output channels, 512 threads; wave = (pixel-row pair pr, 64-channel half chh): 2 x 2 MFMA blocks (conv_mx3.hip's wave tile)
//   entries:   e = base(q) + rank of c among q's regions (pixel-major: a wave's global loads stay coalesced); at most XE_EMAX = 512 = one per thread.  A tile with more
//              (4-pixel cells and finer at this tile size; i.i.d. labels) runs the round-3/4 tile body (mx_tile_body) inside the same launch — per workgroup, no second launch.
//   K loop:    conv_mx3.hip's: chunk = 32 input channels, five tap-pair units, ring of three unit slots refilled by LDS-DMA from inline asm, waves 4-7 half a unit behind.
//   s table:   the chunk's modulation rows [region][32] staged two chunks ahead (one float per thread), read as float4 at conversion time.
//   epilogue:  sb_epilogue (modconv_sb.h) with the 2 x 2-block wave tile; fused single-region ToRGB and split-plane output as the round-3 kernel; split-K partial sums.
// Arithmetic: f16 main term + two MX-fp6 cross terms, the block = one entry's 32 channels (conv_mx3.hip's rule) — the round-3 kernel's block is 3 taps x 8 channels of
// one output pixel, so results differ from it in the last bits (same error class: tests compare both with the fp64 form).
#include "modconv_mx_tile.h"

namespace {

constexpr int XE_TN = 128, XE_CK = 32, XE_TW = 32, XE_TH = 8, XE_PW = XE_TW + 2, XE_PATCH = XE_PW * (XE_TH + 2);     // 340
constexpr int XE_EMAX = 512;
constexpr int XE_NUNIT = 5;
constexpr int XE_U_W16 = 2 * 2 * 2 * XE_TN * 16;     // 16 384
constexpr int XE_U_CLO = 2 * 2 * XE_TN * 16;         // 8 192
constexpr int XE_U_CHI = 2 * 2 * XE_TN * 8;          // 4 096
constexpr int XE_U_SC = 2 * XE_TN * 4;               // 1 024
constexpr int XE_UNITB = XE_U_W16 + XE_U_CLO + XE_U_CHI + XE_U_SC;      // 29 696 (conv_mx3.hip's unit slot)
constexpr int XE_NPIECE = XE_UNITB / 1024;           // 29
// LDS plan
constexpr int XE_A1 = 0;                                    // a1 f16 [16-B slot 4][entry 512]
constexpr int XE_CLO = XE_A1 + 4 * XE_EMAX * 16;            // fp6 codes, first 16 B [term 2][entry]
constexpr int XE_CHI = XE_CLO + 2 * XE_EMAX * 16;           // last 8 B [term 2][entry]
constexpr int XE_SC = XE_CHI + 2 * XE_EMAX * 8;             // scales [entry] x 4 B (byte 0: fp6(a1), byte 1: fp6(a - a1))
constexpr int XE_TBL = XE_SC + XE_EMAX * 4;                 // per-lane tap table [unit 5][pixel 256] x 4 B: entry of tap 2u | entry of tap 2u + 1 << 16
constexpr int XE_SSROW = 36;                                // floats per region row of the chunk's modulation table (32 + 4: rows start 4 banks apart)
constexpr int XE_SSB = E4S_MAX_REGIONS * XE_SSROW * 4;      // 2 304
constexpr int XE_SS = XE_TBL + XE_NUNIT * 256 * 4;          // two buffers
constexpr int XE_RING = XE_SS + 2 * XE_SSB;
constexpr int XE_LDS = XE_RING + 3 * XE_UNITB;              // 158 208
// prologue scratch inside the (not yet written) entry planes
constexpr int XE_T_CLS = 0, XE_T_PM = 256, XE_T_WSUM = 2048, XE_T_ENT = 4096;
static_assert(XE_LDS <= 160 * 1024 && MxLds<1>::BYTES <= XE_LDS, "LDS plan (the fallback tile runs inside this allocation)");
static_assert(XE_T_PM + XE_PATCH * 4 <= XE_T_WSUM && XE_T_ENT + XE_EMAX * 2 <= XE_CLO, "prologue scratch");
static_assert((E4S_MAX_REGIONS + 5) * XE_TN * 4 + 64 + 3 * 256 * 4 <= XE_TBL, "the epilogue's tables overlay the entry planes");
static_assert(E4S_MAX_REGIONS * XE_CK == 512, "one modulation value per thread and chunk");

using CE = SbCfg<2, 2, 2, 4, 5>;        // 2 x 2 blocks per wave, waves = 2 channel halves x 4 row pairs: 128 co x (32 x 8) px

// A global load hipcc does not count (see conv_mx3.hip: the result is valid only behind one of the kernel's own vmcnt waits).  The s_nop: gfx9 requires five wait states
// between a VALU instruction that writes an SGPR (v_readlane of a spilled SGPR, v_readfirstlane) and a vector-memory instruction that reads it; hipcc inserts them for its
// own instructions but does not look into an asm block — without them a base restored from a spill lane right in front of the block is read before it is written
// (observed: memory faults at addresses with a stale upper half, in builds whose register allocation put such a restore there).
__device__ __forceinline__ float xe_load_uncounted(const float* gbase, unsigned voff) {
    float v;
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(gbase) : "memory");
    return v;
}
__device__ __forceinline__ unsigned xe_resid_pair(float a, float b, float sa, float sb_, unsigned a1) { return resid_pair_f16(a, sa, b, sb_, a1); }
// s_waitcnt vmcnt(n) for a wave-uniform n from this kernel's small set of request counts
__device__ __forceinline__ void xe_wait_vm(int n) {
    switch (n) {
        case 0: E4S_WAIT_VM(0); break;
        case 1: E4S_WAIT_VM(1); break;
        case 3: E4S_WAIT_VM(3); break;
        case 4: E4S_WAIT_VM(4); break;
        case 5: E4S_WAIT_VM(5); break;
        case 33: E4S_WAIT_VM(33); break;
        case 36: E4S_WAIT_VM(36); break;
        case 37: E4S_WAIT_VM(37); break;
        default: E4S_WAIT_VM(0); break;
    }
}

// ============================================================================ weight preparation (conv_mx3.hip's unit slots, per output parity of an up layer)
// One thread per (parity, chunk, co tile, unit, k half, co): tap 2 unit + half (tap 9: zeros), its 32 channels.
__global__ __launch_bounds__(256) void prep_weights_mxe_kernel(unsigned char* __restrict__ dst, const float* __restrict__ weight, const float* __restrict__ blur,
                                                               int cout, int cin, int up, float scale) {
    const int npar = up ? 4 : 1;
    const int nchunk = cin / XE_CK, ntile = (cout + XE_TN - 1) / XE_TN;
    const int64_t total = (int64_t)npar * nchunk * ntile * XE_NUNIT * 2 * XE_TN;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int n = (int)(r % XE_TN); r /= XE_TN;
        const int half = (int)(r & 1); r >>= 1;
        const int unit = (int)(r % XE_NUNIT); r /= XE_NUNIT;
        const int tile = (int)(r % ntile); r /= ntile;
        const int chunk = (int)(r % nchunk);
        const int par = (int)(r / nchunk);
        const int co = tile * XE_TN + n, tap = 2 * unit + half;
        unsigned char* slot = dst + ((((size_t)par * nchunk + chunk) * ntile + tile) * XE_NUNIT + unit) * XE_UNITB;
        u32x16 q1, q2;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float a = 0.f, b = 0.f;
            if (tap < 9 && co < cout) {
                a = sb_weff(weight, blur, cin, co, chunk * XE_CK + 2 * j, tap, par, up) * scale;
                b = sb_weff(weight, blur, cin, co, chunk * XE_CK + 2 * j + 1, tap, par, up) * scale;
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
        for (int s = 0; s < 4; ++s) w16[(half * 4 + s) * XE_TN + n] = make_uint4(q1[4 * s], q1[4 * s + 1], q1[4 * s + 2], q1[4 * s + 3]);
        auto expo = [](float m) { const unsigned ex = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu; return ex > 3u ? ex - 2u : 1u; };
        const unsigned e1 = expo(m1), e2 = expo(m2);
        const u32x6 c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
        const u32x6 c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
        uint4* clo = reinterpret_cast<uint4*>(slot + XE_U_W16);
        uint2* chi = reinterpret_cast<uint2*>(slot + XE_U_W16 + XE_U_CLO);
        unsigned* scp = reinterpret_cast<unsigned*>(slot + XE_U_W16 + XE_U_CLO + XE_U_CHI);
        clo[(0 * 2 + half) * XE_TN + n] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
        chi[(0 * 2 + half) * XE_TN + n] = make_uint2(c1[4], c1[5]);
        clo[(1 * 2 + half) * XE_TN + n] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
        chi[(1 * 2 + half) * XE_TN + n] = make_uint2(c2[4], c2[5]);
        const unsigned e2s = e2 > 12u ? e2 - 12u : 0u;
        scp[half * XE_TN + n] = e1 | (e2s << 8);
    }
}

// ============================================================================ the kernel
// p.wmx = unit slots (e4s_modconv_prep_weights_mxe); p.whi carries the round-3 kernel's row slots (e4s_modconv_prep_weights_mx, arith 1) for the tiles that fall back.
// p.chunks_per counts 16-channel chunks (the fallback tile's unit) and is even.
template <bool RGB, bool OSP>
__global__ __launch_bounds__(512, 2) void region_conv_mxe_kernel(const SbParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l5 = lane & 31, khalf = lane >> 5;
    const int pr = wave & 3, chh = wave >> 2;
    const int grp = chh;                                      // waves 4-7 run half a unit behind

    // ---- workgroup -> (K slice, parity, tile, co tile, image): modconv_mx.hip's decode (XCD affinity: an XCD's L2 serves one co tile's weights)
    const int ntile = p.tiles_x * p.tiles_y;
    const int npar = p.up ? 4 : 1;
    unsigned bx_g = blockIdx.x, cot_g = blockIdx.y, b_g = blockIdx.z;
    if (p.xcd_remap) {
        const unsigned nx = gridDim.x, ncg = gridDim.y;
        const unsigned lin = blockIdx.x + nx * (blockIdx.y + ncg * blockIdx.z);
        const unsigned per = 8u / ncg;
        const unsigned xcd = lin & 7u, q = lin >> 3;
        cot_g = xcd / per;
        const unsigned r = q * per + (xcd % per);
        bx_g = r % nx;
        b_g = r / nx;
    }
    const int ks = (int)bx_g / (ntile * npar);
    const int bx = (int)bx_g - ks * ntile * npar;
    const int tile = bx % ntile, par = bx / ntile;
    const int cotile = (int)cot_g, b = (int)b_g;
    const int pa = par >> 1, pb_ = par & 1;
    const int y0 = (tile / p.tiles_x) * XE_TH, x0 = (tile % p.tiles_x) * XE_TW;
    const int co0 = cotile * XE_TN;
    const int hw = p.h * p.w;
    const int ho = p.up ? 2 * p.h : p.h, wo = p.up ? 2 * p.w : p.w;
    const int nchunk = p.cin / XE_CK, ncot = (p.cout + XE_TN - 1) / XE_TN;
    const int ch_begin = ks * (p.chunks_per >> 1);
    const int ch_end = (ch_begin + (p.chunks_per >> 1) < nchunk) ? ch_begin + (p.chunks_per >> 1) : nchunk;
    const int nunits = (ch_end - ch_begin) * XE_NUNIT;
    const float* xb = p.x + (size_t)b * p.cin * hw;
    const float* sbase = p.s + (size_t)b * p.nreg * p.cin;

#ifdef XE_PROF
    unsigned long long tR = 0, tWR = 0, tM = 0, tWM = 0, tST = 0, tTop = 0, tS = __builtin_readcyclecounter(), tStart = tS, tPro = 0, tPro2 = 0, tLoop = 0;
#define XE_STAMP(accum) { const unsigned long long tn = __builtin_readcyclecounter(); accum += tn - tS; tS = tn; }
#else
#define XE_STAMP(accum)
#endif
    // ---- the first three weight units are requested before anything else: they land (3.5 us for 87 KB at a CU's LDS-DMA rate) while the entry table is built
    const unsigned char* wbase = p.wmx + ((size_t)par * nchunk + ch_begin) * ncot * XE_NUNIT * XE_UNITB;
    auto dma_unit = [&](int g, int slot) __attribute__((always_inline)) {      // unit g of this workgroup's K slice -> ring slot
        const int chunk = g / XE_NUNIT, u = g - chunk * XE_NUNIT;
        const unsigned char* src = wbase + (((size_t)chunk * ncot + cotile) * XE_NUNIT + u) * XE_UNITB;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int piece = wave + 8 * k;
            if (piece < XE_NPIECE) dma16_asm(src + piece * 1024, (unsigned)(lane * 16), (unsigned)(XE_RING + slot * XE_UNITB + piece * 1024));
        }
    };
    if (nunits > 0) dma_unit(0, 0);
    if (nunits > 1) dma_unit(1, 1);
    if (nunits > 2) dma_unit(2, 2);
    // ================================================================ prologue: the tile's entries
    auto out_class = [&](int ty, int tx) __attribute__((always_inline)) {        // region of output pixel (ty, tx) of the tile (255: none)
        const int y = y0 + ty, x = x0 + tx;
        int c = E4S_LABEL_NONE;
        if (y < p.h && x < p.w) {
            const int oy = p.up ? 2 * y + pa : y, ox = p.up ? 2 * x + pb_ : x;
            c = p.labels[((size_t)b * p.lh + nearest_src(oy, p.lscale_y, p.lh)) * p.lw + nearest_src(ox, p.lscale_x, p.lw)];
        }
        return c < p.nreg ? c : 255;
    };
    unsigned char* cls_t = lds + XE_T_CLS;
    unsigned* pm = reinterpret_cast<unsigned*>(lds + XE_T_PM);
    int* wsum = reinterpret_cast<int*>(lds + XE_T_WSUM);
    unsigned short* entl = reinterpret_cast<unsigned short*>(lds + XE_T_ENT);
    if (tid < 256) cls_t[tid] = (unsigned char)out_class(tid >> 5, tid & 31);
    __syncthreads();
    unsigned mask = 0u;
    if (tid < XE_PATCH) {
        const int ppy = tid / XE_PW, ppx = tid - ppy * XE_PW;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ty = ppy - dy, tx = ppx - dx;
                if (ty >= 0 && ty < XE_TH && tx >= 0 && tx < XE_TW) {
                    const unsigned c = cls_t[ty * XE_TW + tx];
                    if (c != 255u) mask |= 1u << c;
                }
            }
    }
    const int cnt = __builtin_popcount(mask);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = incl - cnt, E = 0;
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) {
        const int t = wsum[w2];
        if (w2 < wave) base += t;
        E += t;
    }
    E = __builtin_amdgcn_readfirstlane(E);
    if (E > XE_EMAX) {
        // too many distinct (pixel, region) pairs for one entry per thread: the round-3 tile (per-tap operand preparation) computes this workgroup's tile
        E4S_WAIT_VM(0);          // (the requested units land in the ring, whose bytes that tile's own LDS plan reuses)
        __syncthreads();
        SbParams pc = p;
        pc.wmx = reinterpret_cast<const unsigned char*>(p.whi);
        mx_tile_body<1, RGB, OSP, false>(pc, lds, tile, par, ks, cotile, b);
        return;
    }
    if (tid < XE_PATCH) {
        pm[tid] = (unsigned)base | (mask << 16);
        unsigned m = mask;
        int k = 0;
        while (m) {
            const int c = __builtin_ctz(m);
            m &= m - 1u;
            entl[base + k] = (unsigned short)(tid | (c << 9));
            ++k;
        }
    }
    __syncthreads();
    // this thread's entry: patch pixel | region << 9 | valid << 15
    unsigned einfo = tid < E ? ((unsigned)entl[tid] | 0x8000u) : 0u;
    // the tap table of output pixel tid (threads 0..255): entry of (pixel + tap, own region) for the nine taps, two per unit
    if (tid < 256) {
        const int ty = tid >> 5, tx = tid & 31;
        const unsigned c = cls_t[tid];
        unsigned ent[10];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned w = pm[(ty + t / 3) * XE_PW + tx + t % 3];
            ent[t] = c != 255u ? (w & 0xffffu) + (unsigned)__builtin_popcount((w >> 16) & ((1u << c) - 1u)) : 0u;
        }
        ent[9] = ent[8];          // (tap 9 meets zero weights)
        unsigned* tbl = reinterpret_cast<unsigned*>(lds + XE_TBL);
#pragma unroll
        for (int u = 0; u < XE_NUNIT; ++u) tbl[u * 256 + tid] = ent[2 * u] | (ent[2 * u + 1] << 16);
    }
    __syncthreads();          // (everyone is done with the scratch: the entry planes may be written)

    XE_STAMP(tPro)
    f32x16 acc[2][2];        // [co block][pixel block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned ovf = 0u;
#ifdef XE_PROF
    const unsigned dbg = p.perm_mul;      // (tuning build, E4S_MXE_DBG: 1 = no K loop, 2 = no epilogue)
#else
    constexpr unsigned dbg = 0u;
#endif
    if (E > 0 && nunits > 0 && !(dbg & 1u)) {
    const bool has_x = wave * 64 < E;                        // this wave owns entries (wave-uniform)
    const int PCS = wave < 5 ? 4 : 3;                        // this wave's pieces of a unit's 29 DMA requests
    const int NLD = (has_x ? XE_CK : 0) + 1;                 // this wave's requests of one prefetch: 32 activation loads + the modulation value

    // ---- staging
    auto entry_pixel = [&](bool& in) __attribute__((always_inline)) {      // byte offset of this thread's entry's pixel in a channel plane (recomputed where used: see conv_mx3.hip)
        unsigned ei = einfo;
        pin_here(ei);
        const int q = (int)(ei & 0x1ffu);
        const int ppy = q / XE_PW, ppx = q - ppy * XE_PW;
        const int pgy = y0 - 1 + ppy, pgx = x0 - 1 + ppx;
        in = (ei & 0x8000u) && pgy >= 0 && pgy < p.h && pgx >= 0 && pgx < p.w;
        return in ? (unsigned)(pgy * p.w + pgx) * 4u : 0u;
    };
    float xr[XE_CK];
    float sr = 0.f;
    // chunk `chunk`'s 32 channels of this thread's entry + one value of chunk `chunk + 1`'s modulation table (region tid / 32, channel tid % 32)
    auto load_x = [&](int chunk) __attribute__((always_inline)) {
        if (has_x) {
            bool p_in;
            const unsigned goff = entry_pixel(p_in);
#pragma unroll
            for (int c = 0; c < XE_CK; ++c) xr[c] = xe_load_uncounted(xb + (size_t)(chunk * XE_CK + c) * hw, goff);
        }
        {
            int t = tid;
            pin_here(t);
            const int r = (t >> 5) < p.nreg ? (t >> 5) : p.nreg - 1;
            const int cs = chunk + 1 < nchunk ? chunk + 1 : nchunk - 1;
            sr = xe_load_uncounted(sbase, (unsigned)((r * p.cin + cs * XE_CK + (t & 31)) * 4));
        }
    };
    auto store_s = [&](int buf) __attribute__((always_inline)) {
        int t = tid;
        pin_here(t);
        reinterpret_cast<float*>(lds + XE_SS + buf * XE_SSB)[(t >> 5) * XE_SSROW + (t & 31)] = sr;
    };
    // the staged chunk -> this thread's entry in operand form, modulated by the entry's region (table buffer `sbuf`); then the next table goes into the other buffer
    auto store_x = [&](int sbuf) __attribute__((always_inline)) {
        if (has_x) {
        bool p_in;
        (void)entry_pixel(p_in);
        unsigned ei = einfo;
        pin_here(ei);
        const int creg = (int)((ei >> 9) & 0xfu);
        const float4* st = reinterpret_cast<const float4*>(lds + XE_SS + sbuf * XE_SSB + creg * (XE_SSROW * 4));
        u32x16 q1, q2;
        unsigned m = 0u;      // maximum of |a1| as f16 bits, two lanes of 16
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
                q2[2 * c4] = xe_resid_pair(xr[4 * c4], xr[4 * c4 + 1], s4[k].x, s4[k].y, q1[2 * c4]);
                q2[2 * c4 + 1] = xe_resid_pair(xr[4 * c4 + 2], xr[4 * c4 + 3], s4[k].z, s4[k].w, q1[2 * c4 + 1]);
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                u16x2 mm = __builtin_elementwise_max(__builtin_bit_cast(u16x2, m), __builtin_bit_cast(u16x2, q1[2 * c4] & 0x7fff7fffu));
                mm = __builtin_elementwise_max(mm, __builtin_bit_cast(u16x2, q1[2 * c4 + 1] & 0x7fff7fffu));
                m = __builtin_bit_cast(unsigned, mm);
            }
        }
        if (!p_in) {          // padding / no entry: exact zeros
#pragma unroll
            for (int j = 0; j < 16; ++j) { q1[j] = 0u; q2[j] = 0u; }
            m = 0u;
        }
        const unsigned mh = (m & 0xffffu) > (m >> 16) ? (m & 0xffffu) : (m >> 16);
        const unsigned e16 = mh >> 10;                    // f16 exponent field: 31 = the value left the f16 range
        ovf |= e16 >= 31u ? 1u : 0u;
        const unsigned ex = (e16 ? e16 : 1u) + 112u;
        const unsigned e1 = ex > 3u ? ex - 2u : 1u, e2 = ex > 14u ? ex - 13u : 1u;
        const u32x6 c1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q1), __builtin_bit_cast(float, e1 << 23));
        const u32x6 c2 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, q2), __builtin_bit_cast(float, e2 << 23));
        if (ei & 0x8000u) {
            int t = tid;
            pin_here(t);
            uint4* a1p = reinterpret_cast<uint4*>(lds + XE_A1);
#pragma unroll
            for (int s = 0; s < 4; ++s) a1p[s * XE_EMAX + t] = make_uint4(q1[4 * s], q1[4 * s + 1], q1[4 * s + 2], q1[4 * s + 3]);
            uint4* clo = reinterpret_cast<uint4*>(lds + XE_CLO);
            uint2* chi = reinterpret_cast<uint2*>(lds + XE_CHI);
            clo[t] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
            chi[t] = make_uint2(c1[4], c1[5]);
            clo[XE_EMAX + t] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
            chi[XE_EMAX + t] = make_uint2(c2[4], c2[5]);
            reinterpret_cast<unsigned*>(lds + XE_SC)[t] = e1 | (e2 << 8);
        }
        }
        store_s(sbuf ^ 1);
    };
    auto wait_units = [&](bool d, bool lx) __attribute__((always_inline)) { xe_wait_vm((d ? PCS : 0) + (lx ? NLD : 0)); };

    // ---- the first two modulation tables, the first chunk's entries
    {
        const int r = (tid >> 5) < p.nreg ? (tid >> 5) : p.nreg - 1;
        const int c1 = ch_begin + 1 < nchunk ? ch_begin + 1 : nchunk - 1;
        const float s0v = sbase[(size_t)r * p.cin + ch_begin * XE_CK + (tid & 31)];
        sr = sbase[(size_t)r * p.cin + c1 * XE_CK + (tid & 31)];
        reinterpret_cast<float*>(lds + XE_SS)[(tid >> 5) * XE_SSROW + (tid & 31)] = s0v;
    }
    if (has_x) {
        bool p_in;
        const unsigned goff = entry_pixel(p_in);
#pragma unroll
        for (int c = 0; c < XE_CK; ++c) xr[c] = xe_load_uncounted(xb + (size_t)(ch_begin * XE_CK + c) * hw, goff);
    }
    E4S_WAIT_VM(0);
    E4S_LDS_BARRIER();
    store_x(0);              // (reads table 0, writes table 1)
    E4S_LDS_BARRIER();
    if (grp) E4S_LDS_BARRIER();

    // per-lane LDS bases
    const int pix0 = (2 * pr) * XE_TW + l5;                                                        // the lane's first pixel; the second is one tile row down
    const unsigned char* wl = lds + XE_RING + (khalf * XE_TN + chh * 64 + l5) * 16;
    const unsigned* tbl = reinterpret_cast<const unsigned*>(lds + XE_TBL) + pix0;
    unsigned tw0 = tbl[0], tw1 = tbl[XE_TW];                 // tap entries of unit 0 for the two pixels (the next unit's are requested one unit ahead)
    int slot = 0, g = 0;
    XE_STAMP(tPro2)
#pragma unroll 1
    for (int chunk = ch_begin; chunk < ch_end; ++chunk) {
        const bool more = chunk + 1 < ch_end;
        const int kk = chunk - ch_begin;
        if (more) load_x(chunk + 1);
        XE_STAMP(tTop)
#pragma unroll
        for (int u = 0; u < XE_NUNIT; ++u, ++g) {
            const bool first = u == 0, last = u == XE_NUNIT - 1;
            // ---------------- R phase: every operand of the unit into registers, one round of LDS reads
            uint4 xa[2][2][2], wv[2][2][2];          // [pixel / co block][tap][K-step]
            uint4 calo[2][2], wclo[2][2];            // [block][term]
            uint2 cahi[2][2], wchi[2][2];
            int sca[2], scw[2];
            const unsigned char* ws = wl + slot * XE_UNITB;
            {
                const unsigned tw[2] = {tw0, tw1};
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const unsigned e0 = tw[pb] & 0xffffu, e1 = tw[pb] >> 16;
                    const unsigned char* a0p = lds + XE_A1 + (khalf * XE_EMAX + e0) * 16;
                    const unsigned char* a1p = lds + XE_A1 + (khalf * XE_EMAX + e1) * 16;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        xa[pb][0][j] = *reinterpret_cast<const uint4*>(a0p + (2 * j) * XE_EMAX * 16);
                        xa[pb][1][j] = *reinterpret_cast<const uint4*>(a1p + (2 * j) * XE_EMAX * 16);
                    }
                    const unsigned ek = khalf ? e1 : e0;                  // the tap this lane's fp6 K half belongs to
#pragma unroll
                    for (int term = 0; term < 2; ++term) {
                        calo[pb][term] = *reinterpret_cast<const uint4*>(lds + XE_CLO + (term * XE_EMAX + ek) * 16);
                        cahi[pb][term] = *reinterpret_cast<const uint2*>(lds + XE_CHI + (term * XE_EMAX + ek) * 8);
                    }
                    sca[pb] = *reinterpret_cast<const int*>(lds + XE_SC + ek * 4);
                }
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int j = 0; j < 2; ++j) wv[cb][d][j] = *reinterpret_cast<const uint4*>(ws + ((d * 2 + j) * 2 * XE_TN + cb * 32) * 16);
            {
                const unsigned char* wc = lds + XE_RING + slot * XE_UNITB + XE_U_W16 + (khalf * XE_TN + chh * 64 + l5) * 16;
                const unsigned char* wh = lds + XE_RING + slot * XE_UNITB + XE_U_W16 + XE_U_CLO + (khalf * XE_TN + chh * 64 + l5) * 8;
                const unsigned char* wsc = lds + XE_RING + slot * XE_UNITB + XE_U_W16 + XE_U_CLO + XE_U_CHI + (khalf * XE_TN + chh * 64 + l5) * 4;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                    for (int term = 0; term < 2; ++term) {
                        wclo[cb][term] = *reinterpret_cast<const uint4*>(wc + (term * 2 * XE_TN + cb * 32) * 16);
                        wchi[cb][term] = *reinterpret_cast<const uint2*>(wh + (term * 2 * XE_TN + cb * 32) * 8);
                    }
                    scw[cb] = *reinterpret_cast<const int*>(wsc + cb * 32 * 4);
                }
            }
            {   // the next unit's tap entries (the table does not depend on the chunk)
                const int un = u + 1 < XE_NUNIT ? u + 1 : 0;
                tw0 = tbl[un * 256];
                tw1 = tbl[un * 256 + XE_TW];
            }
            // vector-memory requests behind the LDS reads: the refill of the slot the previous unit left (conv_mx3.hip)
            if (g >= 1 && g + 2 < nunits) dma_unit(g + 2, slot == 0 ? 2 : slot - 1);
            const bool d_younger = g >= 1 && g + 2 < nunits;
            const bool lx_younger = first && more;
            if (grp) wait_units(d_younger, lx_younger);
            __builtin_amdgcn_sched_barrier(0);
#ifdef XE_PROF
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            XE_STAMP(tR)
            E4S_LDS_BARRIER();
            XE_STAMP(tWR)
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- M phase: 16 f16 + 8 fp6 MFMAs, nothing else
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                if (d == 1 && u == XE_NUNIT - 1) break;          // tap 9
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
                    acc[cb][pb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(wclo[cb][1], wchi[cb][1]), mx_op6(calo[pb][0], cahi[pb][0]), acc[cb][pb], 2, 2, 1, scw[cb], 0, sca[pb]);
                    acc[cb][pb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(wclo[cb][0], wchi[cb][0]), mx_op6(calo[pb][1], cahi[pb][1]), acc[cb][pb], 2, 2, 0, scw[cb], 1, sca[pb]);
                }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) pin_here(acc[cb][pb]);
            if (!grp) wait_units(d_younger, lx_younger && !last);
            // waves 4-7 convert their entries of the next chunk right behind the chunk's last MFMAs, waves 0-3 behind the barrier (conv_mx3.hip's store phase)
            XE_STAMP(tM)
            if (grp && last && more) { store_x((kk + 1) & 1); XE_STAMP(tST) }
            __builtin_amdgcn_sched_barrier(0);
            E4S_LDS_BARRIER();
            XE_STAMP(tWM)
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (more) {
            if (!grp) { store_x((kk + 1) & 1); XE_STAMP(tST) }
            E4S_LDS_BARRIER();
            XE_STAMP(tWM)
        }
    }
    if (!grp) E4S_LDS_BARRIER();
    E4S_WAIT_VM(0);
    XE_STAMP(tLoop)
    }
    E4S_WAIT_VM(0);          // (a tile without entries ran no loop: its prologue's requests must land before the workgroup's LDS is released)
    if (p.flags && __builtin_amdgcn_ballot_w64(ovf != 0u) != 0 && lane == 0) { atomicOr(p.flags, 1); atomicAdd(p.flags + 1, 1); }      // one report per wave (ops.MxGuard)

    // ================================================================ epilogue
    int cls[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = out_class(2 * pr + q, l5);
        cls[q] = c != 255 ? c : -1;
    }
    if (dbg & 2u) return;
    if (p.ksplit > 1) {
        float* part = p.partial + ((size_t)ks * p.bs + b) * p.cout * ho * wo;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int y = y0 + 2 * pr + q, x = x0 + l5;
            if (y < p.h && x < p.w) {
                const size_t opix = (size_t)(p.up ? 2 * y + pa : y) * wo + (p.up ? 2 * x + pb_ : x);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + (chh * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                        if (co < p.cout) part[(size_t)co * ho * wo + opix] = acc[i][q][r];
                    }
            }
        }
        return;
    }
    sb_epilogue<CE, 2, 2, 4, RGB, OSP>(p, lds, acc, cls, co0, b, y0, x0, pa, pb_, ho, wo, 0u);
#ifdef XE_PROF
    if (blockIdx.x == 5 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0)
        printf("wave %d: E %d total %llu | prologue %llu + %llu | top %llu  R %llu  wait after R %llu  M %llu  store %llu  wait after M %llu  tail %llu | epilogue %llu  (units %d)\n", wave, E,
               __builtin_readcyclecounter() - tStart, tPro, tPro2, tTop, tR, tWR, tM, tST, tWM, tLoop, __builtin_readcyclecounter() - tS, nunits);
#endif
}

template <bool RGB, bool OSP>
int launch_mxe_variant(const SbParams& p, dim3 grid, hipStream_t st) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&region_conv_mxe_kernel<RGB, OSP>), hipFuncAttributeMaxDynamicSharedMemorySize, XE_LDS);
    if (attr != hipSuccess) return fail((int)attr, "region_modconv3x3_mxe: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL((region_conv_mxe_kernel<RGB, OSP>), grid, dim3(512), XE_LDS, st, p);
    return check_launch("region_modconv3x3_mxe");
}

}  // namespace

extern "C" int e4s_modconv_mxe_weight_bytes(int cout, int cin, int up, int64_t* bytes) {
    E4S_REQUIRE(bytes && cout >= 1 && cin >= XE_CK && cin % XE_CK == 0, "modconv_mxe_weight_bytes: bad arguments (cin %% 32 == 0)");
    *bytes = (int64_t)(up ? 4 : 1) * (cin / XE_CK) * cdiv(cout, XE_TN) * XE_NUNIT * XE_UNITB;
    return 0;
}

extern "C" int e4s_modconv_prep_weights_mxe(void* dst, const float* weight, const float* blur, int cout, int cin, int up, void* stream) {
    E4S_REQUIRE(dst && weight, "modconv_prep_weights_mxe: null tensor");
    E4S_REQUIRE(cout >= 1 && cin >= XE_CK && cin % XE_CK == 0, "modconv_prep_weights_mxe: bad size (cin %% 32 == 0)");
    E4S_REQUIRE(!up || blur, "modconv_prep_weights_mxe: an up layer needs the 4x4 blur kernel");
    E4S_REQUIRE(((uintptr_t)dst & 15) == 0, "modconv_prep_weights_mxe: the destination must be 16-byte aligned");
    const int64_t total = (int64_t)(up ? 4 : 1) * (cin / XE_CK) * cdiv(cout, XE_TN) * XE_NUNIT * 2 * XE_TN;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipLaunchKernelGGL(prep_weights_mxe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst, weight, blur, cout, cin, up ? 1 : 0,
                       1.0f / sqrtf((float)cin * 9.f));
    return check_launch("modconv_prep_weights_mxe");
}

// Launch of the entry kernel for an SbParams filled by region_modconv3x3_sb_impl (modconv_sb.hip): p.wmx = unit slots, p.whi = the fallback's row slots.
int e4s::launch_modconv_mxe(SbParams& p, hipStream_t st, float* workspace, int64_t workspace_floats) {
    p.tiles_x = cdiv(p.w, XE_TW);
    p.tiles_y = cdiv(p.h, XE_TH);
    const int npar = p.up ? 4 : 1;
    const int64_t base = (int64_t)p.tiles_x * p.tiles_y * npar * cdiv(p.cout, XE_TN) * p.bs;
    const int ho = p.up ? 2 * p.h : p.h, wo = p.up ? 2 * p.w : p.w;
    const int64_t out_floats = (int64_t)p.bs * p.cout * ho * wo;
    const int nchunk16 = p.cin / CKS;
    int ksplit = 1;
    if (workspace && base < 384)         // one workgroup per CU: split K until one round of the chip is full; a slice is a whole number of 32-channel chunks
        while (ksplit < 16 && base * ksplit * 2 <= 256 && ksplit * 4 <= nchunk16 && (nchunk16 / (ksplit * 2)) % 2 == 0 && nchunk16 % (ksplit * 2) == 0 &&
               (int64_t)(ksplit * 2) * out_floats <= workspace_floats)
            ksplit *= 2;
    if (p.rgb_out) {
        if (p.cout > XE_TN || p.up) return fail(E4S_ERR_ARG, "region_modconv3x3_mxe: fused ToRGB needs all %d output channels in one workgroup tile", p.cout);
        ksplit = 1;
    }
    p.ksplit = ksplit;
    p.chunks_per = nchunk16 / ksplit;
    p.partial = workspace;
    p.uni_blocks = nullptr; p.uni_ctrl = nullptr; p.perm_mul = 0u;
#ifdef XE_PROF
    { const char* e = getenv("E4S_MXE_DBG"); if (e) p.perm_mul = (unsigned)atoi(e); }
#endif
    dim3 grid(p.tiles_x * p.tiles_y * npar * ksplit, cdiv(p.cout, XE_TN), p.bs);
    const unsigned long long tot = (unsigned long long)grid.x * grid.y * grid.z;
    p.xcd_remap = ((grid.y == 2 || grid.y == 4 || grid.y == 8) && tot % 8 == 0) ? 1 : 0;
    const bool rgb = p.rgb_out != nullptr, osp = p.s_next != nullptr;
    if (osp && !rgb) return fail(E4S_ERR_ARG, "region_modconv3x3_mxe: split-plane output is built for the masked fused-ToRGB layer");
    return osp ? launch_mxe_variant<true, true>(p, grid, st) : rgb ? launch_mxe_variant<true, false>(p, grid, st) : launch_mxe_variant<false, false>(p, grid, st);
}
