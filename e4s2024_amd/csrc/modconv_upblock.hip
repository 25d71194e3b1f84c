// Masked up-sampling StyledConv on REGION-UNIFORM output blocks (a3/a4, reference models/stylegan2/model.py:287-300, 389-398): where all
// 16 x 16 output pixels of a block carry one region r, the layer is, inside that block, the single-region form
//     out = lrelu( d[r] * blur( conv_transpose(x * s[r], W, stride 2) ) + noise_weight * noise + act_bias ) * sqrt(2)
// and costs the transposed conv's MACs once (x 2.0 for the block's halo and the idle lanes of its 100 positions in 128) instead of the four
// times of the parity-composed masked form (modconv_sb.hip), which keeps the blocks that mix regions: the same launch pair for every mask,
// each block computed by exactly one of the two kernels (e4s_uniform_blocks writes the map both read).
//
// One workgroup = one block x 32*CB output channels; 256 threads = 4 waves of 32 positions (a position (a, b) owns the 2 x 2 pre-blur values
// z[2a+i][2b+j], 10 x 10 positions feed the block's 19 x 19 window).  K loop as in modconv_upfused.hip: 16 input channels per chunk, the
// 11 x 11 activation patch multiplied by s[r], split into bf16 hi / lo while staged, weights [hi/lo][tap][half][co] from the transposed-conv
// preparation, one MFMA triple per tap into the accumulator of the tap's parity; register prefetch of the next chunk.  Epilogue: 8 channels
// at a time through a pre-blur tile in LDS, 4 x 4 FIR with a rolling four-row window, demodulation / noise / bias / leaky ReLU, fp32 NCHW stores.
#include "sb_common.h"

using namespace e4s;

namespace {

constexpr int MB_OUT = 16;                 // output pixels per block side
constexpr int MB_ZCO = 8;
constexpr int MB_QUAD = 254;               // block map value: four region-uniform 8 x 8 sub-blocks with different regions (SUB = 2 variant)

struct UpBlockParams {
    float* out;
    const float* x;
    const uint4* whi;
    const uint4* wlo;
    const float* s;            // [bs][nreg][cin]
    const float* d;            // [bs][nreg][cout]
    const uint8_t* blocks;     // [bs][nby][nbx]: region of a uniform block, MB_QUAD = four uniform 8 x 8 sub-blocks, 255 = mixed (composed kernel)
    const uint8_t* sub;        // [bs][2 nby][2 nbx]: region of every 8 x 8 sub-block (255 = mixed); read by the SUB = 2 variant
    const int* ctrl;           // ctrl[2] == 0: too few blocks qualify, the layer stays in the composed form (e4s_uniform_blocks)
    const float* blur;         // [4][4]
    const float* noise;
    const float* noise_weight;
    const float* act_bias;
    int noise_bstride, act;
    int bs, cin, cout, h, w, nreg;
    int nbx, nby;
    unsigned perm_mul;         // workgroup i works on block (i * perm_mul) % (nbx * nby): see SbParams::perm_mul
};

// SUB = 1: the block is one region (10 x 10 positions); SUB = 2: its four 8 x 8 sub-blocks are each one region (4 x 6 x 6 positions, every
// sub-block with its own activation patch, modulation and demodulation: 2.5x the algorithmic MACs)
template <int CB, int SUB>
struct UbCfg {
    static constexpr int TN = CB * 32;
    static constexpr int NSB = SUB * SUB;
    static constexpr int SBO = MB_OUT / SUB;               // output pixels per sub-block side
    static constexpr int T = SBO / 2 + 2;                  // positions per sub-block side
    static constexpr int PW = T + 1;                       // activation patch side of a sub-block (a position reads x[a-1], x[a])
    static constexpr int PATCH = NSB * PW * PW;            // 121 / 196 staged pixels
    static constexpr int NPOS = NSB * T * T;               // 100 / 144
    static constexpr int NWAVE = (NPOS + 31) / 32;         // 4 / 5
    static constexpr int NT = 64 * NWAVE;
    static constexpr int ZR = 2 * T;                       // pre-blur rows / columns of a sub-block
    static constexpr int ZS = ZR + 2;
    static constexpr int W4 = 2 * 9 * 2 * TN;
    static constexpr int WPT = (W4 + NT - 1) / NT;
    static constexpr int MAIN_BYTES = W4 * 16 + PATCH * 64;
    static constexpr int ZT_BYTES = MB_ZCO * NSB * ZR * ZS * 4;
    static constexpr int BODY = MAIN_BYTES > ZT_BYTES ? MAIN_BYTES : ZT_BYTES;
    static constexpr int EP_FLOATS = (NSB + 1) * TN + MB_OUT * MB_OUT;
    static constexpr int LDS_BYTES = BODY + EP_FLOATS * 4;
    static_assert(PATCH <= NT && MB_OUT * MB_OUT <= NT, "one staging thread per patch pixel, one per noise value");
};

template <int CB, int SUB>
__global__ __launch_bounds__(64 * ((SUB * SUB * (MB_OUT / SUB / 2 + 2) * (MB_OUT / SUB / 2 + 2) + 31) / 32), 2) void masked_up_block_kernel(const UpBlockParams p) {
    using C = UbCfg<CB, SUB>;
    const int blk = (int)(((unsigned long long)blockIdx.x * p.perm_mul) % gridDim.x);
    const int tyt = blk / p.nbx, txt = blk - tyt * p.nbx;
    const int b = blockIdx.z;
    if (p.ctrl[2] == 0) return;
    const int flag = p.blocks[((size_t)b * p.nby + tyt) * p.nbx + txt];
    if (SUB == 1 ? flag >= p.nreg : flag != MB_QUAD) return;            // not this variant's block

    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* wsm = reinterpret_cast<uint4*>(lds_raw);
    uint4* xh4 = reinterpret_cast<uint4*>(lds_raw + C::W4 * 16);
    uint4* xl4 = xh4 + 2 * C::PATCH;
    float* ep_d = reinterpret_cast<float*>(lds_raw + C::BODY);           // [NSB][TN] demodulation of each sub-block's region
    float* ep_b = ep_d + C::NSB * C::TN;
    float* ep_n = ep_b + C::TN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    const int co0 = blockIdx.y * C::TN;
    const int hw = p.h * p.w;
    const int ho = 2 * p.h, wo = 2 * p.w;
    const int nchunk = (p.cin + CKS - 1) / CKS;
    // region of sub-block j of this block
    auto region_of = [&](int j) -> int {
        if (SUB == 1) return flag;
        return p.sub[((size_t)b * 2 * p.nby + 2 * tyt + (j >> 1)) * (2 * p.nbx) + 2 * txt + (j & 1)];
    };

    // staging element of this thread: one patch pixel of one sub-block, 16 channels per chunk
    const int ssb = tid < C::PATCH ? tid / (C::PW * C::PW) : 0;
    const int se = tid - ssb * C::PW * C::PW;
    const int se_y = se / C::PW, se_x = se - se_y * C::PW;
    // first position of the sub-block: (block origin + sub-block offset) / 2 - 1; patch row 0 is one further up
    const int sgy = tyt * (MB_OUT / 2) + (ssb / SUB) * (C::SBO / 2) - 2 + se_y, sgx = txt * (MB_OUT / 2) + (ssb % SUB) * (C::SBO / 2) - 2 + se_x;
    const bool s_in = tid < C::PATCH && sgy >= 0 && sgy < p.h && sgx >= 0 && sgx < p.w;
    const int sgoff = s_in ? sgy * p.w + sgx : 0;
    const float* xb = p.x + (size_t)b * p.cin * hw;
    const float* sb = p.s + ((size_t)b * p.nreg + region_of(ssb)) * p.cin;

    const int pos = wave * 32 + l5;
    const bool pos_ok = pos < C::NPOS;
    const int posc = pos_ok ? pos : C::NPOS - 1;
    const int psb = posc / (C::T * C::T), plp = posc - psb * C::T * C::T;
    const int pty = plp / C::T, ptx = plp - pty * C::T;
    const int xoff = psb * C::PW * C::PW + pty * C::PW + ptx;

    f32x16 accs[4][CB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[a][i][r] = 0.f;

    float xr[CKS];
    unsigned wr[C::WPT][4];
    const ptrdiff_t wdelta = p.wlo - p.whi;
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const int ci0 = chunk * CKS;
        const int cmax = p.cin - 1 - ci0;
#pragma unroll
        for (int c = 0; c < CKS; ++c) xr[c] = xb[(size_t)(ci0 + (c < cmax ? c : cmax)) * hw + sgoff];
        const size_t wbase = (size_t)chunk * 18 * p.cout;
#pragma unroll
        for (int v = 0; v < C::WPT; ++v) {
            int idx = tid + v * C::NT;
            idx = idx < C::W4 ? idx : C::W4 - 1;
            const int hl = idx / (18 * C::TN);
            const int rem = idx - hl * 18 * C::TN;
            const int th = rem / C::TN, n = rem - th * C::TN;
            const int co = (co0 + n < p.cout) ? co0 + n : p.cout - 1;
            const uint4 t4 = p.whi[(ptrdiff_t)hl * wdelta + (ptrdiff_t)(wbase + (size_t)th * p.cout + co)];
            wr[v][0] = t4.x; wr[v][1] = t4.y; wr[v][2] = t4.z; wr[v][3] = t4.w;
        }
    };
    auto store_chunk = [&](int chunk) __attribute__((always_inline)) {
        if (tid < C::PATCH) {
            unsigned hi[8], lo[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int c0 = chunk * CKS + 2 * c;
                const float s0 = c0 < p.cin ? sb[c0] : 0.f, s1 = c0 + 1 < p.cin ? sb[c0 + 1] : 0.f;   // the sub-block's region (SUB = 1: workgroup-uniform)
                split2(s_in ? xr[2 * c] * s0 : 0.f, s_in ? xr[2 * c + 1] * s1 : 0.f, hi[c], lo[c]);
            }
            const int sw = (tid >> 3) & 1;
            xh4[tid * 2 + (0 ^ sw)] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            xh4[tid * 2 + (1 ^ sw)] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
            xl4[tid * 2 + (0 ^ sw)] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            xl4[tid * 2 + (1 ^ sw)] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
        }
#pragma unroll
        for (int v = 0; v < C::WPT; ++v) {
            const int idx = tid + v * C::NT;
            if (idx < C::W4) wsm[idx] = make_uint4(wr[v][0], wr[v][1], wr[v][2], wr[v][3]);
        }
    };

    // epilogue operands, fetched next to the first chunk's loads and parked in LDS (no global load after a store)
    float ep_r0[C::NSB], ep_r1 = 0.f, ep_r2 = 0.f;
    {
        const int co = co0 + tid;
#pragma unroll
        for (int j = 0; j < C::NSB; ++j) ep_r0[j] = (tid < C::TN && co < p.cout && p.d) ? p.d[((size_t)b * p.nreg + region_of(j)) * p.cout + co] : 1.f;
        if (tid < C::TN && co < p.cout && p.act_bias) ep_r1 = p.act_bias[co];
        if (p.noise && tid < MB_OUT * MB_OUT) {
            const int ny = tyt * MB_OUT + (tid >> 4), nx = txt * MB_OUT + (tid & 15);
            if (ny < ho && nx < wo) ep_r2 = p.noise_weight[0] * p.noise[(size_t)b * p.noise_bstride + (size_t)ny * wo + nx];
        }
    }
    load_chunk(0);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();
        store_chunk(chunk);
        if (chunk == 0) {
            if (tid < C::TN) {
#pragma unroll
                for (int j = 0; j < C::NSB; ++j) ep_d[j * C::TN + tid] = ep_r0[j];
                ep_b[tid] = ep_r1;
            }
            if (tid < MB_OUT * MB_OUT) ep_n[tid] = ep_r2;
        }
        __syncthreads();
        if (chunk + 1 < nchunk) load_chunk(chunk + 1);
        const uint4* whalf = wsm + khalf * C::TN + l5;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int e = xoff + (1 - (ky >> 1)) * C::PW + (1 - (kx >> 1));
            const int ai = 2 * (ky & 1) + (kx & 1);
            const int slot = e * 2 + (khalf ^ ((e >> 3) & 1));
            const uint4 bh = xh4[slot], bl = xl4[slot];
            uint4 ah[CB], al[CB];
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                ah[i] = whalf[tap * 2 * C::TN + i * 32];
                al[i] = whalf[18 * C::TN + tap * 2 * C::TN + i * 32];
            }
#pragma unroll
            for (int i = 0; i < CB; ++i)
                accs[ai][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh), accs[ai][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CB; ++i)
                accs[ai][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl), accs[ai][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CB; ++i)
                accs[ai][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh), accs[ai][i], 0, 0, 0);
        }
    }

    // ---- epilogue: the pre-blur tiles of 8 channels, per sub-block z[2 pty + i][2 ptx + j]; output pixel (y, x) of a sub-block =
    // sum_{t,u} k[t][u] z[y + 1 + t][x + 1 + u]
    __syncthreads();
    float* zt = reinterpret_cast<float*>(lds_raw);       // [8][NSB][ZR][ZS]
    float kf[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) kf[t] = p.blur[15 - t];
    // blur item (threads 0..255): column bx of the block, channel bco of the pass, rows 8 byg .. +7
    const int bx = tid & 15, bco = (tid >> 4) & 7, byg = (tid >> 7) & 1;
    const bool blur_thread = tid < 256;
    const int bsb = SUB == 1 ? 0 : 2 * byg + (bx >> 3);                       // sub-block of these 8 rows / this column
    const int brow0 = SUB == 1 ? 8 * byg : 0, blx = SUB == 1 ? bx : (bx & 7);  // first row / column inside the sub-block
    const int oy0 = tyt * MB_OUT + 8 * byg, ox = txt * MB_OUT + bx;
    int nrow = ho - oy0;
    nrow = (!blur_thread || nrow < 0) ? 0 : (nrow > 8 ? 8 : nrow);
    const bool col_ok = ox < wo;
    const float* zc = zt + ((bco * C::NSB + bsb) * C::ZR + brow0 + 1) * C::ZS + blx + 1;
    const float* nzp = ep_n + 8 * byg * MB_OUT + bx;
    const float neg = p.act ? 0.2f : 1.f, gain = p.act ? 1.41421356237309515f : 1.f;
#pragma unroll
    for (int i = 0; i < CB; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (pos_ok) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int col = 4 * khalf + rr;
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
                        *reinterpret_cast<float2*>(&zt[((col * C::NSB + psb) * C::ZR + 2 * pty + ci) * C::ZS + 2 * ptx]) =
                            make_float2(accs[2 * ci][i][4 * g + rr], accs[2 * ci + 1][i][4 * g + rr]);
                }
            }
            __syncthreads();
            const int cl = i * 32 + 8 * g + bco;
            const int co = co0 + cl;
            if (nrow > 0 && col_ok && co < p.cout) {
                const float dd = ep_d[bsb * C::TN + cl], bi = ep_b[cl];
                float* orow = p.out + ((size_t)b * p.cout + co) * ho * wo + (size_t)oy0 * wo + ox;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3;
#pragma unroll 1
                for (int zr = 0; zr < 8 + 3; ++zr) {
                    const float* zp = zc + zr * C::ZS;
                    const float z0 = zp[0], z1 = zp[1], z2 = zp[2], z3 = zp[3];
                    a3 = 0.f;
                    a0 = __builtin_fmaf(z0, kf[12], a0); a1 = __builtin_fmaf(z0, kf[8], a1); a2 = __builtin_fmaf(z0, kf[4], a2); a3 = __builtin_fmaf(z0, kf[0], a3);
                    a0 = __builtin_fmaf(z1, kf[13], a0); a1 = __builtin_fmaf(z1, kf[9], a1); a2 = __builtin_fmaf(z1, kf[5], a2); a3 = __builtin_fmaf(z1, kf[1], a3);
                    a0 = __builtin_fmaf(z2, kf[14], a0); a1 = __builtin_fmaf(z2, kf[10], a1); a2 = __builtin_fmaf(z2, kf[6], a2); a3 = __builtin_fmaf(z2, kf[2], a3);
                    a0 = __builtin_fmaf(z3, kf[15], a0); a1 = __builtin_fmaf(z3, kf[11], a1); a2 = __builtin_fmaf(z3, kf[7], a2); a3 = __builtin_fmaf(z3, kf[3], a3);
                    const int r = zr - 3;
                    if (r >= 0 && r < nrow) {
                        float v = __builtin_fmaf(a0, dd, bi) + nzp[r * MB_OUT];
                        v = fmaxf(v, v * neg) * gain;
                        orow[(size_t)r * wo] = v;
                    }
                    a0 = a1; a1 = a2; a2 = a3;
                }
            }
            __syncthreads();
        }
    }
}

// One workgroup per ROW OF FOUR 16 x 16 output blocks (= the 64 x 16 output pixels of one tile of the composed kernel, modconv_sb.hip: an
// 8 x 32 input tile at the four parities), one thread per pixel column of each block (labels sampled 'nearest' at ho x wo exactly as the
// masked kernels do):
//   sub[b][2 by + sy][2 bx + sx] = the region shared by the 8 x 8 sub-block's pixels, 255 if they differ or are no region (>= nreg);
//   blocks[b][by][bx] = that region if all four sub-blocks share one, MB_QUAD (with want_quad) if each sub-block is uniform but they differ,
//   else 255 — and 255 for ALL FOUR blocks of the row unless every one of them qualifies: the composed kernel can only leave a tile out as
//   a whole, so a tile with one mixed block is cheaper entirely in the composed form than partly in both.
__global__ __launch_bounds__(256) void uniform_blocks_kernel(uint8_t* __restrict__ blocks, uint8_t* __restrict__ sub, const uint8_t* __restrict__ labels, int lh,
                                                             int lw, int ho, int wo, float lsy, float lsx, int nreg, int want_quad, int* __restrict__ ctrl,
                                                             int min_percent) {
    // wave k = block k of the row; lane = (sub-block q = lane >> 4, its pixel column x8 = lane & 7, rows 4 yh .. 4 yh + 3 with yh = (lane >> 3) & 1):
    // four labels per lane, everything else with wave ballots — one barrier in the whole kernel
    __shared__ int flag[4];
    const int b = blockIdx.z, by = blockIdx.y, tile = blockIdx.x;
    const int nbx = wo / MB_OUT, nby = ho / MB_OUT;
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int q = lane >> 4, x8 = lane & 7, yh = (lane >> 3) & 1;
    const int bxk = tile * 4 + k;
    int ox = bxk * MB_OUT + (q & 1) * 8 + x8;
    ox = ox < wo ? ox : wo - 1;
    const int sx = nearest_src(ox, lsx, lw);
    int c0 = -1;
    bool same = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int oy = by * MB_OUT + (q >> 1) * 8 + yh * 4 + i;
        oy = oy < ho ? oy : ho - 1;
        const int c = labels[((size_t)b * lh + nearest_src(oy, lsy, lh)) * lw + sx];
        if (i == 0) c0 = c;
        same = same && c == c0;
    }
    const int refq = __shfl(c0, q * 16, 64);                              // the sub-block's first label
    const unsigned long long okm = __ballot(same && c0 == refq);
    int r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rj = __shfl(c0, j * 16, 64);
        r[j] = (((okm >> (16 * j)) & 0xffffull) == 0xffffull && rj < nreg) ? rj : 255;
    }
    int f = 255;
    if (bxk < nbx) {
        const bool all_uni = r[0] != 255 && r[1] != 255 && r[2] != 255 && r[3] != 255;
        const bool one = all_uni && r[0] == r[1] && r[0] == r[2] && r[0] == r[3];
        f = one ? r[0] : ((all_uni && want_quad) ? MB_QUAD : 255);
        if (lane < 4) sub[((size_t)b * 2 * nby + 2 * by + (lane >> 1)) * (2 * nbx) + 2 * bxk + (lane & 1)] = (uint8_t)r[lane];
    }
    if (lane == 0) flag[k] = bxk < nbx ? f : 0;                           // (a block beyond the edge does not veto its row)
    __syncthreads();
    const bool all = flag[0] != 255 && flag[1] != 255 && flag[2] != 255 && flag[3] != 255;
    if (lane == 0 && bxk < nbx) blocks[((size_t)b * nby + by) * nbx + bxk] = (uint8_t)(all ? f : 255);
    // ctrl (zeroed once by the caller, left zeroed by every launch): [0] = rows of four blocks that qualify, [1] = workgroups done, [2] = the verdict for the two consumers: 1 if
    // at least min_percent of the rows qualify.  Below that the transposed-conv form does not pay (measured: scattered blocks run at half
    // the rate of contiguous ones in both kernels) and the whole layer stays in the composed form.
    if (threadIdx.x == 0) {
        if (all) atomicAdd(&ctrl[0], 1);
        __threadfence();
        const int total = gridDim.x * gridDim.y * gridDim.z;
        if (atomicAdd(&ctrl[1], 1) == total - 1) {
            __threadfence();
            const int q = atomicAdd(&ctrl[0], 0);
            ctrl[2] = (q * 100 >= min_percent * total) ? 1 : 0;
            ctrl[0] = 0;        // ready for the next layer on this stream (its map kernel runs after this layer's consumers)
            ctrl[1] = 0;
        }
    }
}

}  // namespace

extern "C" int e4s_uniform_blocks(uint8_t* blocks, uint8_t* sub, int* ctrl, const uint8_t* labels, int bs, int lh, int lw, int ho, int wo, int nreg, int want_quad,
                                  int min_percent, void* stream) {
    E4S_REQUIRE(blocks && sub && labels && ctrl, "uniform_blocks: null tensor");
    E4S_REQUIRE(min_percent >= 0 && min_percent <= 100, "uniform_blocks: min_percent in 0..100");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && lh >= 1 && lw >= 1 && ho >= 16 && wo >= 16 && (ho % 16) == 0 && (wo % 16) == 0 && nreg >= 1 && nreg <= E4S_MAX_REGIONS,
                "uniform_blocks: bad size (output height / width multiples of 16)");
    if (bs == 0) return 0;
    hipLaunchKernelGGL(uniform_blocks_kernel, dim3(cdiv(wo, 64), ho / 16, bs), dim3(256), 0, (hipStream_t)stream, blocks, sub, labels, lh, lw, ho, wo,
                       (float)lh / (float)ho, (float)lw / (float)wo, nreg, want_quad, ctrl, min_percent);
    return check_launch("uniform_blocks");
}

// The 16 x 16 output blocks of a masked up-sampling StyledConv that lie under one region, or whose four 8 x 8 sub-blocks each do (blocks[b][by][bx]
// != 255; the others are left untouched for e4s_region_modconv3x3_sb with the same block map).  Weights: the transposed-conv preparation (e4s_modconv_prep_weights_sb, k = 3, not composed).
extern "C" int e4s_masked_upconv_blocks(float* out, const float* x, const uint16_t* whi, const uint16_t* wlo, const float* s, const float* d,
                                        const uint8_t* blocks, const uint8_t* sub, const int* ctrl, const float* blur, const float* noise, int noise_bs,
                                        const float* noise_weight, const float* act_bias, int act, int bs, int cin, int cout, int h, int w, int nreg,
                                        int sub_blocks, void* stream) {
    E4S_REQUIRE(out && x && whi && wlo && s && blocks && sub && ctrl && blur, "masked_upconv_blocks: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= 1 && cout >= 1 && h >= 8 && w >= 8 && (h % 8) == 0 && (w % 8) == 0, "masked_upconv_blocks: bad size (h, w multiples of 8)");
    E4S_REQUIRE(nreg >= 1 && nreg <= E4S_MAX_REGIONS, "masked_upconv_blocks: %d regions (max %d)", nreg, E4S_MAX_REGIONS);
    E4S_REQUIRE(!noise || (noise_weight && (noise_bs == 1 || noise_bs == bs)), "masked_upconv_blocks: noise needs its weight and batch 1 or bs");
    E4S_REQUIRE((((uintptr_t)whi | (uintptr_t)wlo) & 15) == 0, "masked_upconv_blocks: weight slabs must be 16-byte aligned");
    if (bs == 0) return 0;
    UpBlockParams p;
    p.out = out; p.x = x; p.whi = reinterpret_cast<const uint4*>(whi); p.wlo = reinterpret_cast<const uint4*>(wlo); p.s = s; p.d = d; p.blocks = blocks; p.sub = sub; p.ctrl = ctrl;
    p.blur = blur; p.noise = noise; p.noise_weight = noise_weight; p.act_bias = act_bias;
    p.noise_bstride = (noise && noise_bs == bs) ? 4 * h * w : 0; p.act = act;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.nreg = nreg;
    p.nbx = 2 * w / MB_OUT; p.nby = 2 * h / MB_OUT;
    p.perm_mul = coprime_stride((unsigned)(p.nbx * p.nby));
    using C1 = UbCfg<2, 1>;
    using C2 = UbCfg<2, 2>;
    const dim3 grid(p.nbx * p.nby, cdiv(cout, C1::TN), bs);
    hipLaunchKernelGGL((masked_up_block_kernel<2, 1>), grid, dim3(C1::NT), C1::LDS_BYTES, (hipStream_t)stream, p);   // blocks under one region
    // blocks of four uniform 8 x 8 sub-blocks (only present in a map made with want_quad).  Measured on the benchmark maps (the 64 -> 128 layer,
    // 8-pixel cells): 0.57 ms against the composed form's 0.47 — 144 positions are five waves on four SIMDs at 252 registers, one workgroup per
    // CU; kept behind E4S_UP_SUBBLOCKS for masks where it is the 16 x 16 blocks that are rare.
    if (sub_blocks) hipLaunchKernelGGL((masked_up_block_kernel<2, 2>), grid, dim3(C2::NT), C2::LDS_BYTES, (hipStream_t)stream, p);
    return check_launch("masked_upconv_blocks");
}
