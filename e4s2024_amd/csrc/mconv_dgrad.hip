// Data and style gradient of the masked modulated 3x3 convolution in ONE kernel (the reference gets both from autograd through
// ModulatedConv2d.forward, models/stylegan2/model.py:276-320, inside the loop of training/video_swap_ft_coach.py:268-299):
//     U_g[i,k,p]  = sum_o W_g[o,i,k] * gz_g[o,p]                                  (the GEMM of e4s_gemm_sb, never stored here)
//     dx[i,q]     = sum_g sum_k s[c_g(q-k+1), i] * U_g[i,k,q-k+1]                   (e4s_mconv_fold)
//     ds[r,i]     = sum_g sum_{p: c_g(p) = r} sum_k U_g[i,k,p] * x[i,p+k-1]
// The modulation of the data gradient depends on the SOURCE position p and the input channel i, not on the tap, so each of the nine taps keeps
// its own accumulator over the output channels (M = 32 input channels, N = 32 positions, K = cout in chunks of 16: 27 MFMAs per chunk and wave,
// every product as hi*hi + hi*lo + lo*hi of bf16 halves), the accumulators are multiplied by s[c_g(p), i] once, and the nine shifted planes are
// summed through LDS (col2im).  Without U in HBM: the unfused pair moved 9 * cin * P floats out and twice back in per group.
//
// One workgroup = 32 input channels x a tile of 8 x 32 positions (wave = row), all groups g one after the other.  A tile owns the outputs whose
// nine source positions it holds — rows 1..6, columns 1..30 of the tile (columns 0..31 when the map is 32 wide: there is nothing beyond) — so
// neighbouring tiles overlap by two positions and recompute them (1.42x the MACs of the bare GEMM); every output and every position's ds term
// is produced by exactly one workgroup, in a fixed order.
//
// STATUS (round 2, measured with tools/time_dgrad.py, batch 1): correct and bit-reproducible, but NOT the default (E4S_DGRAD_FUSED=1 turns it on).
// 144 accumulator registers per wave leave one workgroup of two waves per SIMD on a CU, so the load -> split -> barrier -> MFMA chain of a chunk,
// the col2im passes and the next tile's first loads never overlap: 2.3 us per chunk where the MFMAs need 0.7, 22 us of epilogue per tile.  Against
// e4s_gemm_sb + e4s_mconv_fold: 128 -> 128 at 256^2 0.33 vs 0.36 ms, the 128^2 -> 256^2 up layer 0.66 vs 0.73, 256 -> 256 at 128^2 0.26 vs 0.18,
// 512 -> 512 at 64^2 0.29 vs 0.13 (tiles of 6 x 30 owned outputs fit small maps badly).  What it needs to win: operands split to bf16 once by their
// producers and brought in by LDS-DMA several chunks ahead, and a smaller LDS footprint so that two workgroups share a CU.
#include "sb_common.h"

using namespace e4s;

namespace {

constexpr int DG_NT = 512;
constexpr int DG_POS = 256;                     // 8 rows x 32 columns
constexpr int DG_OY = 6;
constexpr int DG_MI = 32;
constexpr int DG_AROW = 33;                     // uint4 per (tap, k-half) row of the weight operand (32 + 1: the staging writes stride over taps)
constexpr int DG_A4 = 18 * DG_AROW;
constexpr int DG_B4 = DG_POS * 2;
constexpr int DG_MAIN_BYTES = (2 * DG_A4 + 2 * DG_B4) * 16;
constexpr int DG_PS = DG_POS + 8;               // plane stride in floats: the two k-halves of a wave (channels 4 apart) land 32 banks apart
constexpr int DG_V_BYTES = 9 * 8 * DG_PS * 4;
constexpr int DG_BODY = DG_V_BYTES > DG_MAIN_BYTES ? DG_V_BYTES : DG_MAIN_BYTES;
constexpr int DG_X_BYTES = DG_MI * DG_PS * 4;
constexpr int DG_DSW_BYTES = 8 * E4S_MAX_REGIONS * DG_MI * 4;
constexpr int DG_STAB_BYTES = (E4S_MAX_REGIONS + 1) * DG_MI * 4;
constexpr int DG_LDS_BYTES = DG_BODY + DG_X_BYTES + DG_DSW_BYTES + DG_STAB_BYTES;

struct DgradParams {
    float* dx;                 // [bs][cin][h][w] or null
    float* ds_part;            // [tiles][bs][nreg][cin] or null
    const float* gz;           // [G][bs][cout][h*w]
    const float* wg;           // [G][cout][cin][3][3]
    const float* x;            // [bs][cin][h][w] (ds only)
    const float* s;            // [bs][nreg][cin]
    const uint8_t* lab;        // [bs][up*h][up*w] or null (one region)
    int bs, cin, cout, h, w, nreg, up;
    int ntx, step_x, x_base, own_lo, own_hi;
};

__global__ __launch_bounds__(DG_NT, 2) void mconv_dgrad_kernel(const DgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* a_hi = reinterpret_cast<uint4*>(lds_raw);
    uint4* a_lo = a_hi + DG_A4;
    uint4* b_hi = a_lo + DG_A4;
    uint4* b_lo = b_hi + DG_B4;
    float* vt = reinterpret_cast<float*>(lds_raw);                                  // [9 taps][8 channels][DG_PS], after the K loop
    float* xt = reinterpret_cast<float*>(lds_raw + DG_BODY);                        // [32 channels][DG_PS]
    float* dsw = reinterpret_cast<float*>(lds_raw + DG_BODY + DG_X_BYTES);          // [8 waves][MAX_REGIONS][32]
    float* stab = reinterpret_cast<float*>(lds_raw + DG_BODY + DG_X_BYTES + DG_DSW_BYTES);   // [MAX_REGIONS + 1][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l5 = lane & 31, khalf = lane >> 5;
    const int ty = blockIdx.x / p.ntx, tx = blockIdx.x - ty * p.ntx;
    const int i0 = blockIdx.y * DG_MI, b = blockIdx.z;
    const int y0 = ty * DG_OY - 1, x0 = tx * p.step_x + p.x_base;
    const int P = p.h * p.w, lw = p.up * p.w, G = p.up * p.up;
    const int nchunk = (p.cout + CKS - 1) / CKS;

    // this lane's position in the MFMA layout (wave = row, l5 = column; both k-halves hold the same position)
    const int py = y0 + wave, px = x0 + l5;
    const bool p_in = py >= 0 && py < p.h && px >= 0 && px < p.w;
    const bool p_own = p_in && wave >= 1 && wave <= DG_OY && l5 >= p.own_lo && l5 <= p.own_hi;

    // staging roles.  gz: one position and 8 of the chunk's 16 output channels per thread; W: one (channel, tap) and all 16 per thread < 288
    const int spos = tid & 255, soh = tid >> 8;
    const int sgy = y0 + (spos >> 5), sgx = x0 + (spos & 31);
    const bool s_in = sgy >= 0 && sgy < p.h && sgx >= 0 && sgx < p.w;
    const int sgoff = s_in ? sgy * p.w + sgx : 0;
    const int wil = tid / 9, wk = tid - wil * 9;
    const bool w_thr = tid < DG_MI * 9;
    const bool w_ok = w_thr && i0 + wil < p.cin;

    // x tile, modulation table, zeroed per-wave style sums
    if (p.ds_part) {
        for (int e = tid; e < DG_MI * DG_POS; e += DG_NT) {
            const int il = e >> 8, pos = e & 255;
            const int gy = y0 + (pos >> 5), gx = x0 + (pos & 31);
            const bool in = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w && i0 + il < p.cin;
            xt[il * DG_PS + pos] = in ? p.x[((size_t)b * p.cin + i0 + il) * P + gy * p.w + gx] : 0.f;
        }
        for (int e = tid; e < 8 * E4S_MAX_REGIONS * DG_MI; e += DG_NT) dsw[e] = 0.f;
    }
    for (int e = tid; e < (E4S_MAX_REGIONS + 1) * DG_MI; e += DG_NT) {
        const int r = e >> 5, il = e & 31;
        stab[e] = (r < p.nreg && i0 + il < p.cin) ? p.s[((size_t)b * p.nreg + r) * p.cin + i0 + il] : 0.f;
    }

    const int gC = tid & 31;                                   // column of this thread's col2im outputs (512 and 192 are multiples of 32)

    float gr[8], wr[16];
    for (int g = 0; g < G; ++g) {
        const int ga = g / p.up, gb = g - ga * p.up;
        const float* gzb = p.gz + ((size_t)g * p.bs + b) * p.cout * P;                      // (uniform bases, 32-bit per-lane offsets: one add per load)
        const float* wgb = p.wg + (size_t)g * p.cout * p.cin * 9 + (size_t)i0 * 9;
        const unsigned woff = (unsigned)((w_ok ? wil : 0) * 9 + wk);
        const unsigned wstride = (unsigned)(p.cin * 9);
        const int soh_u = __builtin_amdgcn_readfirstlane(soh);
        auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
            const int o0 = chunk * CKS;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int o = o0 + soh_u * 8 + j;
                gr[j] = gzb[(unsigned)(o < p.cout ? o : p.cout - 1) * (unsigned)P + (unsigned)sgoff];
            }
            if (w_thr) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int o = o0 + j;
                    wr[j] = wgb[(unsigned)(o < p.cout ? o : p.cout - 1) * wstride + woff];
                }
            }
        };
        auto store_chunk = [&](int chunk) __attribute__((always_inline)) {
            const int o0 = chunk * CKS;
            {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = o0 + soh * 8 + 2 * j;
                    split2((s_in && o < p.cout) ? gr[2 * j] : 0.f, (s_in && o + 1 < p.cout) ? gr[2 * j + 1] : 0.f, hi[j], lo[j]);
                }
                const int slot = spos * 2 + (soh ^ ((spos >> 3) & 1));
                b_hi[slot] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                b_lo[slot] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
            if (w_thr) {
                unsigned hi[8], lo[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int o = o0 + 2 * j;
                    split2((w_ok && o < p.cout) ? wr[2 * j] : 0.f, (w_ok && o + 1 < p.cout) ? wr[2 * j + 1] : 0.f, hi[j], lo[j]);
                }
                const int e = wk * 2 * DG_AROW + wil;
                a_hi[e] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                a_hi[e + DG_AROW] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
                a_lo[e] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                a_lo[e + DG_AROW] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
            }
        };

        f32x16 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        load_chunk(0);
        const int bpos = wave * 32 + l5;
        const int bslot = bpos * 2 + (khalf ^ ((bpos >> 3) & 1));
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            __syncthreads();
            store_chunk(chunk);
            __syncthreads();
            if (chunk + 1 < nchunk) load_chunk(chunk + 1);
            const uint4 bh = b_hi[bslot], bl = b_lo[bslot];
            const uint4* ah_p = a_hi + khalf * DG_AROW + l5;
            const uint4* al_p = a_lo + khalf * DG_AROW + l5;
            // taps in pairs: two independent accumulators alternate, so that no MFMA waits for the one before it
#define DG_MFMA(A, B, T) acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[T], 0, 0, 0)
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                const uint4 ah0 = ah_p[t * 2 * DG_AROW], al0 = al_p[t * 2 * DG_AROW];
                const uint4 ah1 = ah_p[(t + 1) * 2 * DG_AROW], al1 = al_p[(t + 1) * 2 * DG_AROW];
                DG_MFMA(ah0, bh, t); DG_MFMA(ah1, bh, t + 1);
                DG_MFMA(ah0, bl, t); DG_MFMA(ah1, bl, t + 1);
                DG_MFMA(al0, bh, t); DG_MFMA(al1, bh, t + 1);
            }
            {
                const uint4 ah0 = ah_p[16 * DG_AROW], al0 = al_p[16 * DG_AROW];
                DG_MFMA(ah0, bh, 8); DG_MFMA(ah0, bl, 8); DG_MFMA(al0, bh, 8);
            }
#undef DG_MFMA
        }
        __syncthreads();                                       // the staging area becomes the plane area

        // region of this lane's position in group g
        int c = E4S_MAX_REGIONS;
        if (p_in) {
            c = p.lab ? p.lab[((size_t)b * p.up * p.h + p.up * py + ga) * lw + p.up * px + gb] : 0;
            if (c >= p.nreg) c = E4S_MAX_REGIONS;
        }

        // ---- style gradient: t[i] = sum_k U[i,k,p] x[i,p+k-1] at the positions this tile owns, summed per region over the wave's row
        if (p.ds_part) {
            unsigned long long todo = __ballot(p_own && c < p.nreg);
            if (todo) {                                                     // (t is computed where it is used: in a block of its own, its FMAs sink
            float t[16];                                                    // below a branch and all 144 reads are issued, and spilled, first)
#pragma unroll
            for (int q = 0; q < 16; ++q) t[q] = 0.f;
            // (lanes that own nothing compute on clamped addresses and are masked out of the sums below; the opaque offset keeps the 144 reads
            // inside the loop over g — hoisted, they live in scratch)
            int xbase = (4 * khalf) * DG_PS;
            asm volatile("" : "+v"(xbase));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int row = wave + ky - 1;
                row = row < 0 ? 0 : (row > 7 ? 7 : row);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int cc = l5 + kx - 1;
                    const bool cin_tile = cc >= 0 && cc < 32;               // (false only beyond the edge of a 32-wide map: x = 0 there)
                    const float* xo = xt + xbase + row * 32 + (cin_tile ? cc : l5);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        float xv = xo[(8 * (q >> 2) + (q & 3)) * DG_PS];
                        xv = cin_tile ? xv : 0.f;
                        t[q] = __builtin_fmaf(acc[ky * 3 + kx][q], xv, t[q]);
                    }
                    __builtin_amdgcn_sched_barrier(0);                      // (one tap's 16 LDS reads in flight at a time: registers)
                }
            }
            while (todo) {
                const int first = __ffsll((long long)todo) - 1;
                const int r = __shfl(c, first, 64);
                const bool m = p_own && c == r;
                float mine = 0.f;                                           // lane l5 = q < 16 of each half keeps the sum of value q
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float v = m ? t[q] : 0.f;
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    mine = l5 == q ? v : mine;
                }
                if (l5 < 16) dsw[(wave * E4S_MAX_REGIONS + r) * DG_MI + 8 * (l5 >> 2) + 4 * khalf + (l5 & 3)] += mine;
                todo &= ~__ballot(m);
            }
            }
        }

        // ---- data gradient: the nine planes times s[c(p), i], eight channels per pass, summed with their shifts
        if (p.dx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 sv = *reinterpret_cast<const float4*>(&stab[c * DG_MI + 8 * j + 4 * khalf]);
                const float sq[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) vt[(t * 8 + 4 * khalf + rr) * DG_PS + wave * 32 + l5] = acc[t][4 * j + rr] * sq[rr];
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
#pragma unroll
                for (int n3 = 0; n3 < 3; ++n3) {
                    // three outputs per thread and pass (8 channels x 6 rows x 32 columns)
                    const int n = tid + DG_NT * n3;
                    const int il8 = n / (DG_OY * 32), r = n - il8 * DG_OY * 32;
                    const int R = (r >> 5) + 1;
                    const int gbase = il8 * DG_PS + R * 32 + gC;
                    float sum = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int cc = gC - kx + 1;
                            const float v = vt[(ky * 3 + kx) * 8 * DG_PS + gbase + (1 - ky) * 32 + ((cc >= 0 && cc < 32) ? 1 - kx : 0)];
                            sum += (cc >= 0 && cc < 32) ? v : 0.f;
                        }
                    const int qy = y0 + R, qx = x0 + gC, i = i0 + 8 * j + il8;
                    if (gC >= p.own_lo && gC <= p.own_hi && qy < p.h && qx >= 0 && qx < p.w && i < p.cin) {
                        float* dst = p.dx + ((size_t)b * p.cin + i) * P + qy * p.w + qx;        // (qy >= 0: R >= 1)
                        if (g == 0) *dst = sum;                                                  // the same thread adds the groups, in order
                        else *dst += sum;
                    }
                }
                __syncthreads();
            }
        }
    }

    if (p.ds_part) {
        __syncthreads();
        for (int e = tid; e < p.nreg * DG_MI; e += DG_NT) {
            const int r = e >> 5, il = e & 31;
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) v += dsw[(wv * E4S_MAX_REGIONS + r) * DG_MI + il];
            if (i0 + il < p.cin) p.ds_part[(((size_t)blockIdx.x * p.bs + b) * p.nreg + r) * p.cin + i0 + il] = v;
        }
    }
}

}  // namespace

extern "C" int e4s_mconv_dgrad_tiles(int h, int w) {
    if (h < 1 || w < 32) return 0;
    return cdiv(h, DG_OY) * (w == 32 ? 1 : cdiv(w, 30));
}

extern "C" int e4s_mconv_dgrad(float* dx, float* ds_part, const float* gz, const float* wg, const float* x, const float* s, const uint8_t* labels,
                               int bs, int cin, int cout, int h, int w, int nreg, int up, void* stream) {
    E4S_REQUIRE((dx || ds_part) && gz && wg && s, "mconv_dgrad: null tensor");
    E4S_REQUIRE(!ds_part || x, "mconv_dgrad: the style gradient needs x");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= 1 && cout >= 1 && h >= 1 && w >= 32 && (int64_t)h * w * up * up < ((int64_t)1 << 24) && nreg >= 1 &&
                    nreg <= E4S_MAX_REGIONS && (up == 1 || up == 2),
                "mconv_dgrad: bad size (w >= 32, nreg 1..%d, up 1 / 2)", E4S_MAX_REGIONS);
    if (bs == 0) return 0;
    DgradParams p;
    p.dx = dx; p.ds_part = ds_part; p.gz = gz; p.wg = wg; p.x = x; p.s = s; p.lab = labels;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.nreg = nreg; p.up = up;
    if (w == 32) { p.ntx = 1; p.step_x = 0; p.x_base = 0; p.own_lo = 0; p.own_hi = 31; }
    else { p.ntx = cdiv(w, 30); p.step_x = 30; p.x_base = -1; p.own_lo = 1; p.own_hi = 30; }
    const int nty = cdiv(h, DG_OY);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&mconv_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DG_LDS_BYTES);
    if (attr != hipSuccess) return fail((int)attr, "mconv_dgrad: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL(mconv_dgrad_kernel, dim3(p.ntx * nty, cdiv(cin, DG_MI), bs), dim3(DG_NT), DG_LDS_BYTES, (hipStream_t)stream, p);
    return check_launch("mconv_dgrad");
}
