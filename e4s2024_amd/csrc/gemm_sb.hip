// Batched fp32 GEMM on the bf16 matrix cores (row f1: the contractions of the PTI backward, which were library GEMMs):
//
//     C[b] (M x N, row-major) = opA(A[b]) (M x K) * opB(B[b]) (K x N)
//
// fp32 in, fp32 out; every product is a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (the arithmetic of modconv_sb.hip: ~2^-17 per product).  Both operands are split while they are staged into LDS, in the layout the
// MFMA fragments read with one ds_read_b128: plane[row][32 k] bf16 = 4 uint4 per row, the uint4 index XOR-swizzled with (row >> 2) & 3.
//
// Either operand may be stored with K contiguous ("KC": A as [M][K], B as [N][K]) or with its M / N index contiguous ("MC": A as [K][M],
// B as [K][N]).  KC rows are read as 64-byte pieces (4 x float4 per thread), MC tiles as dwords with consecutive lanes on consecutive
// rows (coalesced 256-byte lines, transposed for free: a thread collects 16 k of one row).  Workgroup = 128 x 128 of C, 256 threads,
// 2 x 2 waves of 64 x 64; K in chunks of 32 with a one-chunk register prefetch.  A long K with few tiles (weight gradients: K = all
// pixels of a layer) is split over workgroups; the partial products are summed in a fixed order (no atomics: run-to-run identical).
#include "sb_common.h"

using namespace e4s;

namespace {

constexpr int GT = 128;    // tile side
constexpr int GK = 32;     // k per chunk
constexpr int GNT = 256;

struct GemmParams {
    float* c;
    const float* a;
    const float* b;
    float* partial;
    int M, N, K;
    int lda, ldb;
    long long sa, sb, sc;
    int ksplit, chunks_per;
    int tiles_m, tiles_n;
    int batch;
};

__device__ __forceinline__ int g_slot(int r, int j) { return r * 4 + (j ^ ((r >> 2) & 3)); }

// KC operand: rows x K, k contiguous.  Thread: row = t >> 1, k = 16 (t & 1) .. +15 (four float4).
template <bool KC>
__device__ __forceinline__ void g_load(float (&v)[16], const float* __restrict__ base, int ld, int row0, int rows, int k0, int K, int tid) {
    if constexpr (KC) {
        const int r = row0 + (tid >> 1);
        const int rr = r < rows ? r : rows - 1;
        const int kb = k0 + 16 * (tid & 1);
        const float* src = base + (size_t)rr * ld;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = kb + 4 * q;
            const int kk = k < K ? k : K - 4;     // K % 4 == 0: a float4 is wholly inside or wholly outside; outside ones are zeroed at the split
            const float4 f = *reinterpret_cast<const float4*>(src + kk);
            v[4 * q + 0] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
        }
    } else {
        // MC operand: K x rows, the row index contiguous.  Thread: row = t & 127, k = 16 (t >> 7) .. +15 (sixteen dwords)
        const int r = row0 + (tid & 127);
        const int rr = r < rows ? r : rows - 1;
        const int kb = k0 + 16 * (tid >> 7);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kb + e;
            v[e] = base[(size_t)(k < K ? k : K - 1) * ld + rr];
        }
    }
}

template <bool KC>
__device__ __forceinline__ void g_store(const float (&v)[16], uint4* __restrict__ hi, uint4* __restrict__ lo, int k0, int K, int tid) {
    const int r = KC ? (tid >> 1) : (tid & 127);
    const int kh = KC ? (tid & 1) : (tid >> 7);
    const int kb = k0 + 16 * kh;
    unsigned h[8], l[8];
    {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float t0 = (kb + 2 * c < K) ? v[2 * c] : 0.f, t1 = (kb + 2 * c + 1 < K) ? v[2 * c + 1] : 0.f;
            split2(t0, t1, h[c], l[c]);
        }
    }
    hi[g_slot(r, 2 * kh)] = make_uint4(h[0], h[1], h[2], h[3]);
    hi[g_slot(r, 2 * kh + 1)] = make_uint4(h[4], h[5], h[6], h[7]);
    lo[g_slot(r, 2 * kh)] = make_uint4(l[0], l[1], l[2], l[3]);
    lo[g_slot(r, 2 * kh + 1)] = make_uint4(l[4], l[5], l[6], l[7]);
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(GNT, 2) void gemm_sb_kernel(const GemmParams p) {
    __shared__ uint4 lds[4 * GT * 4];          // A hi | A lo | B hi | B lo: 32 KB
    uint4* ahi = lds;
    uint4* alo = lds + GT * 4;
    uint4* bhi = lds + 2 * GT * 4;
    uint4* blo = lds + 3 * GT * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, kg = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x;
    const int ks = bx % p.ksplit; bx /= p.ksplit;
    const int tn = bx % p.tiles_n, tm = bx / p.tiles_n;
    const int bz = blockIdx.z;
    const float* A = p.a + (size_t)bz * p.sa;
    const float* B = p.b + (size_t)bz * p.sb;
    const int m0 = tm * GT, n0 = tn * GT;
    const int nchunk = (p.K + GK - 1) / GK;
    const int ch_begin = ks * p.chunks_per;
    const int ch_end = ch_begin + p.chunks_per < nchunk ? ch_begin + p.chunks_per : nchunk;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    float va[16], vb[16];
    if (ch_begin < ch_end) {
        g_load<AKC>(va, A, p.lda, m0, p.M, ch_begin * GK, p.K, tid);
        g_load<BKC>(vb, B, p.ldb, n0, p.N, ch_begin * GK, p.K, tid);
    }
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        __syncthreads();
        g_store<AKC>(va, ahi, alo, ch * GK, p.K, tid);
        g_store<BKC>(vb, bhi, blo, ch * GK, p.K, tid);
        __syncthreads();
        if (ch + 1 < ch_end) {
            g_load<AKC>(va, A, p.lda, m0, p.M, (ch + 1) * GK, p.K, tid);
            g_load<BKC>(vb, B, p.ldb, n0, p.N, (ch + 1) * GK, p.K, tid);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = 2 * t + kg;
            uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + l5, rb = wn * 64 + i * 32 + l5;
                ah[i] = ahi[g_slot(ra, j)]; al[i] = alo[g_slot(ra, j)];
                bh[i] = bhi[g_slot(rb, j)]; bl[i] = blo[g_slot(rb, j)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[q]), acc[i][q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[q]), acc[i][q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[q]), acc[i][q], 0, 0, 0);
        }
    }

    // lanes = columns (n), registers = rows (m): every store instruction writes 128-byte row segments
    float* C = p.ksplit > 1 ? p.partial + ((size_t)ks * p.batch + bz) * (size_t)p.M * p.N : p.c + (size_t)bz * p.sc;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = n0 + wn * 64 + q * 32 + l5;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (m < p.M) C[(size_t)m * p.N + n] = acc[i][q][r];
            }
    }
}

// Skinny variant for the weight gradient of ToRGB (M = 3 output channels, K = all pixels), both operands K-contiguous: 32 x 256 of C
// per workgroup; wave w owns columns 64 w .. 64 w + 63.  (Measured for M = 32 / 64 as well: no better than the 128-row tile there —
// both stream 128-byte pieces of 4 MB-strided rows; tools/time_bwd_parts.py.)  A: 32 rows x 32 k (one float4 per thread); B: 256 rows x 32 k (a 128-byte line per thread).
constexpr int SKM = 32, SKN = 256;

__global__ __launch_bounds__(GNT, 2) void gemm_sb_skinny_kernel(const GemmParams p) {
    __shared__ uint4 lds[2 * (SKM + SKN) * 4];   // A hi | A lo | B hi | B lo: 36 KB
    uint4* ahi = lds;
    uint4* alo = lds + SKM * 4;
    uint4* bhi = lds + 2 * SKM * 4;
    uint4* blo = bhi + SKN * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, kg = lane >> 5;
    int bx = blockIdx.x;
    const int ks = bx % p.ksplit; bx /= p.ksplit;
    const int tn = bx % p.tiles_n, tm = bx / p.tiles_n;
    const int bz = blockIdx.z;
    const int m0 = tm * SKM, n0 = tn * SKN;
    const int nchunk = (p.K + GK - 1) / GK;
    const int ch_begin = ks * p.chunks_per;
    const int ch_end = ch_begin + p.chunks_per < nchunk ? ch_begin + p.chunks_per : nchunk;
    const int ar = m0 + (tid >> 3) < p.M ? m0 + (tid >> 3) : p.M - 1, akq = tid & 7;
    const int br = n0 + tid < p.N ? n0 + tid : p.N - 1;
    const float* arow = p.a + (size_t)bz * p.sa + (size_t)ar * p.lda;
    const float* brow = p.b + (size_t)bz * p.sb + (size_t)br * p.ldb;

    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    float4 va, vb[8];
    auto load = [&](int ch) __attribute__((always_inline)) {
        const int k0 = ch * GK;
        const int ka = k0 + 4 * akq;
        va = *reinterpret_cast<const float4*>(arow + (ka < p.K ? ka : p.K - 4));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + 4 * q;
            vb[q] = *reinterpret_cast<const float4*>(brow + (k < p.K ? k : p.K - 4));
        }
    };
    auto store = [&](int ch) __attribute__((always_inline)) {
        const int k0 = ch * GK;
        {
            const bool ok = k0 + 4 * akq < p.K;
            unsigned h0, l0, h1, l1;
            split2(ok ? va.x : 0.f, ok ? va.y : 0.f, h0, l0);
            split2(ok ? va.z : 0.f, ok ? va.w : 0.f, h1, l1);
            const int sl = g_slot(tid >> 3, akq >> 1) * 2 + (akq & 1);
            reinterpret_cast<uint2*>(ahi)[sl] = make_uint2(h0, h1);
            reinterpret_cast<uint2*>(alo)[sl] = make_uint2(l0, l1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float4 f = vb[2 * j + e];
                const bool ok = k0 + 8 * j + 4 * e < p.K;
                split2(ok ? f.x : 0.f, ok ? f.y : 0.f, h[2 * e], l[2 * e]);
                split2(ok ? f.z : 0.f, ok ? f.w : 0.f, h[2 * e + 1], l[2 * e + 1]);
            }
            bhi[g_slot(tid, j)] = make_uint4(h[0], h[1], h[2], h[3]);
            blo[g_slot(tid, j)] = make_uint4(l[0], l[1], l[2], l[3]);
        }
    };

    if (ch_begin < ch_end) load(ch_begin);
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        __syncthreads();
        store(ch);
        __syncthreads();
        if (ch + 1 < ch_end) load(ch + 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = 2 * t + kg;
            const uint4 ah = ahi[g_slot(l5, j)], al = alo[g_slot(l5, j)];
            uint4 bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int rb = wave * 64 + q * 32 + l5;
                bh[q] = bhi[g_slot(rb, j)]; bl[q] = blo[g_slot(rb, j)];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[q]), acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[q]), acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[q]), acc[q], 0, 0, 0);
        }
    }
    float* C = p.ksplit > 1 ? p.partial + ((size_t)ks * p.batch + bz) * (size_t)p.M * p.N : p.c + (size_t)bz * p.sc;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = n0 + wave * 64 + q * 32 + l5;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (m < p.M) C[(size_t)m * p.N + n] = acc[q][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of a (masked) modulated convolution without the unfolded operand:
//     dW[g][b][co][(ci, k)] = sum_p gz[g][b][co][p] * s[b][c_g(p)][ci] * x[b][ci][p + k - pad]
// = the GEMM above with A = gz (K-contiguous) and a VIRTUAL B: row n = (ci, k) of the modulated im2col matrix is produced while it is
// staged — 16 consecutive pixels of one image row per thread: 16 dwords of x (shifted by the tap, zero outside the image), their 16
// labels (one 16-byte load; two for the up layers, whose labels live at the output resolution: c_g(p) = label[2 py + gy][2 px + gx]) and
// the modulation from a table in LDS (this tile's input channels — at most 16 for 3x3, 128 for 1x1 — x 16 regions).  Reads x instead of the 9x larger unfolded
// matrix and needs no unfold pass.  labels == null: one region (s[b][0][ci], or 1 if s == null: a plain convolution).  w % 16 == 0.
struct WgradParams {
    GemmParams g;          // a = gz, c / partial / M = cout, N = cin * KK, K = h * w; lda = K; batch = G * bs
    const float* x;        // [bs][cin][h][w]
    const float* s;        // [bs][nreg][cin] or null
    const uint8_t* lab;    // [bs][up h][up w] or null
    int bs, cin, h, w, nreg, up;
};

// TM = rows (output channels) of a tile: 128 (waves 2 x 2, 64 x 64 each), 64 (waves 1 x 4, 64 x 32 each) or 32 (1 x 4, 32 x 32 each).  The
// 64- and 32-channel layers at 512^2 / 1024^2 fill a 128-row tile to a half / a quarter: their MFMAs, not the building of B, were the time.
template <int KS, int TM>
__global__ __launch_bounds__(GNT, 2) void mconv_wgrad_kernel(const WgradParams q) {
    constexpr int KK = KS * KS, PAD = KS / 2;
    constexpr int WAVES_M = TM == GT ? 2 : 1, WAVES_N = 4 / WAVES_M;
    constexpr int MI = TM / WAVES_M / 32, NJ = GT / WAVES_N / 32;     // 32 x 32 blocks per wave: 2 x 2, 2 x 1, 1 x 1
    const GemmParams& p = q.g;
    __shared__ uint4 lds[4 * GT * 4];
    constexpr int TAB_CI = KS == 1 ? GT : 16;            // distinct input channels among a tile's 128 rows
    __shared__ float s_tab[TAB_CI * E4S_MAX_REGIONS];
    uint4* ahi = lds;
    uint4* alo = lds + GT * 4;
    uint4* bhi = lds + 2 * GT * 4;
    uint4* blo = lds + 3 * GT * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, kg = lane >> 5;
    const int wm = WAVES_M == 2 ? wave >> 1 : 0, wn = WAVES_M == 2 ? wave & 1 : wave;
    int bx = blockIdx.x;
    const int ks = bx % p.ksplit; bx /= p.ksplit;
    const int tn = bx % p.tiles_n, tm = bx / p.tiles_n;
    const int bz = blockIdx.z;
    const int gpar = bz / q.bs, b = bz - gpar * q.bs;
    const int gy = q.up == 2 ? gpar >> 1 : 0, gx = q.up == 2 ? gpar & 1 : 0;
    const float* A = p.a + (size_t)bz * p.sa;
    const int m0 = tm * TM, n0 = tn * GT;
    const bool a_thr = (tid >> 1) < TM;                  // threads that stage a row of A
    const int nchunk = (p.K + GK - 1) / GK;
    const int ch_begin = ks * p.chunks_per;
    const int ch_end = ch_begin + p.chunks_per < nchunk ? ch_begin + p.chunks_per : nchunk;

    // the modulation of this tile's input channels (ci0 .. ci0 + 15) for every region
    const int ci0 = n0 / KK;
    for (int i = tid; i < TAB_CI * E4S_MAX_REGIONS; i += GNT) {
        const int cl = i >> 4, r = i & 15;
        float v = 0.f;
        if (ci0 + cl < q.cin && r < q.nreg) v = q.s ? q.s[((size_t)b * q.nreg + r) * q.cin + ci0 + cl] : 1.f;
        s_tab[i] = v;
    }
    // this thread's row of the virtual B: n = (ci, tap), 16 pixels from kb
    const int n = n0 + (tid >> 1);
    const bool n_ok = n < p.N;
    const int ci = n_ok ? n / KK : 0, tap = n_ok ? n - ci * KK : 0;
    const int ky = tap / KS - PAD, kx = tap % KS - PAD;
    const int cl16 = (ci - ci0) * 16;
    const float* xc = q.x + ((size_t)b * q.cin + ci) * q.h * q.w;
    const int lw = q.up * q.w;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float va[16], vb[16], vl = 0.f, vr = 0.f;
    uint4 lb[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    bool row_ok = false;
    // 16 pixels of one image row, aligned (w % 16 == 0): four 16-byte loads + the left / right neighbour for the shifted taps
    auto load_b = [&](int ch) __attribute__((always_inline)) {
        const int p0 = ch * GK + 16 * (tid & 1);
        const int py = p0 / q.w, px0 = p0 - py * q.w;
        const int yy = py + ky;
        row_ok = n_ok && p0 < p.K && yy >= 0 && yy < q.h;
        const float* xr = xc + (size_t)(row_ok ? yy : 0) * q.w + (p0 < p.K ? px0 : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 f = *reinterpret_cast<const float4*>(xr + 4 * j);
            vb[4 * j] = f.x; vb[4 * j + 1] = f.y; vb[4 * j + 2] = f.z; vb[4 * j + 3] = f.w;
        }
        if (KS == 3) {
            const int pxs = p0 < p.K ? px0 : 0;
            vl = xr[pxs > 0 ? -1 : 0];
            vr = xr[pxs + 16 < q.w ? 16 : 15];
        }
        if (q.lab) {
            const int pyc = py < q.h ? py : q.h - 1;
            const uint8_t* lr = q.lab + ((size_t)b * q.up * q.h + (size_t)q.up * pyc + gy) * lw + q.up * (p0 < p.K ? px0 : 0);
            lb[0] = *reinterpret_cast<const uint4*>(lr);
            if (q.up == 2) lb[1] = *reinterpret_cast<const uint4*>(lr + 16);
        }
    };
    auto store_b = [&](int ch) __attribute__((always_inline)) {
        const int p0 = ch * GK + 16 * (tid & 1);
        const int px0 = p0 % q.w;
        unsigned h[8], l[8];
        float t[16];
        // the tap's view of the row: element e reads pixel px0 + e + kx
        float xs[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float ctr = vb[e];
            const float lft = e > 0 ? vb[e - 1] : vl, rgt = e < 15 ? vb[e + 1] : vr;
            xs[e] = kx == 0 ? ctr : (kx < 0 ? lft : rgt);
        }
        const bool ok_l = row_ok && (px0 + kx >= 0), ok_r = row_ok && (px0 + 15 + kx < q.w);
        const unsigned lw4[8] = {lb[0].x, lb[0].y, lb[0].z, lb[0].w, lb[1].x, lb[1].y, lb[1].z, lb[1].w};
        bool uni = true;
        if (q.lab) {
            const unsigned b0 = lw4[0] & 0xffu, rep = b0 * 0x01010101u;
            uni = lw4[0] == rep && lw4[1] == rep && lw4[2] == rep && lw4[3] == rep;
            if (q.up == 2) uni = uni && lw4[4] == rep && lw4[5] == rep && lw4[6] == rep && lw4[7] == rep;
        }
        if (uni) {          // one region under these 16 pixels (or no label map): one table read
            const int c = q.lab ? (int)(lw4[0] & 0xffu) : 0;
            const float sv = c < q.nreg ? s_tab[cl16 + (c < E4S_MAX_REGIONS ? c : 0)] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) t[e] = xs[e] * sv;
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int byte = q.up == 2 ? 2 * e + gx : e;
                const int c = (int)((lw4[byte >> 2] >> (8 * (byte & 3))) & 0xffu);
                t[e] = c < q.nreg ? xs[e] * s_tab[cl16 + (c < E4S_MAX_REGIONS ? c : 0)] : 0.f;
            }
        }
        if (!row_ok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) t[e] = 0.f;
        }
        if (!ok_l) t[0] = 0.f;
        if (!ok_r) t[15] = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) split2(t[2 * c], t[2 * c + 1], h[c], l[c]);
        const int r = tid >> 1, kh = tid & 1;
        bhi[g_slot(r, 2 * kh)] = make_uint4(h[0], h[1], h[2], h[3]);
        bhi[g_slot(r, 2 * kh + 1)] = make_uint4(h[4], h[5], h[6], h[7]);
        blo[g_slot(r, 2 * kh)] = make_uint4(l[0], l[1], l[2], l[3]);
        blo[g_slot(r, 2 * kh + 1)] = make_uint4(l[4], l[5], l[6], l[7]);
    };

    if (ch_begin < ch_end) {
        if (a_thr) g_load<true>(va, A, p.lda, m0, p.M, ch_begin * GK, p.K, tid);
        load_b(ch_begin);
    }
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        __syncthreads();                       // (also orders the s_tab fill before its first use)
        if (a_thr) g_store<true>(va, ahi, alo, ch * GK, p.K, tid);
        store_b(ch);
        __syncthreads();
        if (ch + 1 < ch_end) {
            if (a_thr) g_load<true>(va, A, p.lda, m0, p.M, (ch + 1) * GK, p.K, tid);
            load_b(ch + 1);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = 2 * t + kg;
            uint4 ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ra = wm * 64 + i * 32 + l5;
                ah[i] = ahi[g_slot(ra, j)]; al[i] = alo[g_slot(ra, j)];
            }
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                const int rb = wn * 32 * NJ + i * 32 + l5;
                bh[i] = bhi[g_slot(rb, j)]; bl[i] = blo[g_slot(rb, j)];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jq = 0; jq < NJ; ++jq)
                    acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[jq]), acc[i][jq], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jq = 0; jq < NJ; ++jq)
                    acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[jq]), acc[i][jq], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jq = 0; jq < NJ; ++jq)
                    acc[i][jq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[jq]), acc[i][jq], 0, 0, 0);
        }
    }
    float* C = p.ksplit > 1 ? p.partial + ((size_t)ks * p.batch + bz) * (size_t)p.M * p.N : p.c + (size_t)bz * p.sc;
#pragma unroll
    for (int jq = 0; jq < NJ; ++jq) {
        const int nn = n0 + wn * 32 * NJ + jq * 32 + l5;
        if (nn >= p.N) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (m < p.M) C[(size_t)m * p.N + nn] = acc[i][jq][r];
            }
    }
}

// C = sum of the ksplit partial products, always in the order 0, 1, 2, ... (one thread per element; a workgroup of 256 threads shares
// the work of an element when there are few elements and many partials: 4 x 64 strided partial sums, then a fixed tree)
__global__ __launch_bounds__(256) void gemm_sb_finalize_wide_kernel(float* __restrict__ c, const float* __restrict__ partial, long long mn, long long sc, int batch,
                                                                   int ksplit) {
    __shared__ float red[256];
    const long long total = mn * batch;
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;      // 64 elements per workgroup, 4 threads per element
    const long long i = (long long)blockIdx.x * 64 + e;
    float a = 0.f;
    if (i < total)
        for (int k = part; k < ksplit; k += 4) a += partial[(size_t)k * total + i];
    red[threadIdx.x] = a;
    __syncthreads();
    if (part == 0 && i < total) {
        const long long b = i / mn;
        c[(size_t)b * sc + (i - b * mn)] = (red[e] + red[64 + e]) + (red[128 + e] + red[192 + e]);
    }
}

__global__ __launch_bounds__(256) void gemm_sb_finalize_kernel(float* __restrict__ c, const float* __restrict__ partial, long long mn, long long sc, int batch, int ksplit) {
    const long long total = mn * batch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float a = 0.f;
        for (int k = 0; k < ksplit; ++k) a += partial[(size_t)k * total + i];
        const long long b = i / mn;
        c[(size_t)b * sc + (i - b * mn)] = a;
    }
}

}  // namespace

// C[b] = opA(A[b]) opB(B[b]) for b < batch.  a_kc != 0: A is stored [M][K] (row stride lda), else [K][M]; b_kc != 0: B is stored [N][K]
// (row stride ldb), else [K][N].  Batch strides in floats (0 = the same matrix for every b); C is dense [batch][M][N] at stride_c.
// workspace: split-K scratch (may be null: no split); the choice of split depends on sizes only, so results are reproducible.
extern "C" int e4s_gemm_sb(float* c, const float* a, const float* b, int M, int N, int K, int a_kc, int b_kc, int lda, int ldb, int64_t stride_a,
                           int64_t stride_b, int64_t stride_c, int batch, float* workspace, int64_t workspace_floats, void* stream) {
    E4S_REQUIRE(c && a && b, "gemm_sb: null tensor");
    E4S_REQUIRE(M >= 1 && N >= 1 && K >= 1 && batch >= 0 && batch <= 65535, "gemm_sb: bad size");
    E4S_REQUIRE(lda >= (a_kc ? K : M) && ldb >= (b_kc ? K : N), "gemm_sb: leading dimension smaller than the row");
    if (a_kc) E4S_REQUIRE((lda % 4) == 0 && (((uintptr_t)a | (uintptr_t)(stride_a * 4)) & 15) == 0 && (K % 4) == 0, "gemm_sb: a K-contiguous A needs 16-byte aligned rows (lda %% 4 == 0) and K %% 4 == 0");
    if (b_kc) E4S_REQUIRE((ldb % 4) == 0 && (((uintptr_t)b | (uintptr_t)(stride_b * 4)) & 15) == 0 && (K % 4) == 0, "gemm_sb: a K-contiguous B needs 16-byte aligned rows (ldb %% 4 == 0) and K %% 4 == 0");
    if (batch == 0) return 0;
    GemmParams p;
    p.c = c; p.a = a; p.b = b; p.partial = workspace;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.sa = stride_a; p.sb = stride_b; p.sc = stride_c; p.batch = batch;
    const bool skinny = a_kc && b_kc && M <= 8;
    p.tiles_m = cdiv(M, skinny ? SKM : GT); p.tiles_n = cdiv(N, skinny ? SKN : GT);
    const int nchunk = cdiv(K, GK);
    const int64_t base = (int64_t)p.tiles_m * p.tiles_n * batch;
    int ksplit = 1;
    if (workspace)
        while (base * ksplit < 512 && ksplit * 2 * 4 <= nchunk && (int64_t)(ksplit * 2) * batch * M * N <= workspace_floats && ksplit < 1024) ksplit *= 2;
    p.ksplit = ksplit;
    p.chunks_per = cdiv(nchunk, ksplit);
    const int64_t gx = (int64_t)p.tiles_m * p.tiles_n * ksplit;
    E4S_REQUIRE(gx <= 0x7fffffff, "gemm_sb: grid too large");
    dim3 grid((unsigned)gx, 1, batch);
    hipStream_t st = (hipStream_t)stream;
    if (skinny) hipLaunchKernelGGL(gemm_sb_skinny_kernel, grid, dim3(GNT), 0, st, p);
    else if (a_kc && b_kc) hipLaunchKernelGGL((gemm_sb_kernel<true, true>), grid, dim3(GNT), 0, st, p);
    else if (a_kc) hipLaunchKernelGGL((gemm_sb_kernel<true, false>), grid, dim3(GNT), 0, st, p);
    else if (b_kc) hipLaunchKernelGGL((gemm_sb_kernel<false, true>), grid, dim3(GNT), 0, st, p);
    else hipLaunchKernelGGL((gemm_sb_kernel<false, false>), grid, dim3(GNT), 0, st, p);
    if (ksplit > 1) {
        const int64_t mn = (int64_t)M * N, total = mn * batch;
        if (total <= 65536 && ksplit >= 16) {
            hipLaunchKernelGGL(gemm_sb_finalize_wide_kernel, dim3((unsigned)cdiv64(total, 64)), dim3(256), 0, st, c, workspace, (long long)mn, (long long)stride_c, batch, ksplit);
        } else {
            const int g = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
            hipLaunchKernelGGL(gemm_sb_finalize_kernel, dim3(g), dim3(256), 0, st, c, workspace, (long long)mn, (long long)stride_c, batch, ksplit);
        }
    }
    return check_launch("gemm_sb");
}

// dW[g][b][co][(ci, k)] (dense [G * bs][cout][cin * ks * ks]) of the masked modulated convolution, G = up * up (the composed weight of each
// output parity for up = 2), from gz [G][bs][cout][h * w] (e4s_mconv_scale), x [bs][cin][h][w], s [bs][nreg][cin] (null: 1) and the labels
// ([bs][up h][up w] uint8; null: one region) — the unfolded operand of e4s_mconv_unfold is never materialised.  w % 16 == 0, ks 1 / 3.
extern "C" int e4s_mconv_wgrad(float* dw, const float* gz, const float* x, const float* s, const uint8_t* labels, int bs, int cin, int cout, int h,
                               int w, int ks, int nreg, int up, float* workspace, int64_t workspace_floats, void* stream) {
    E4S_REQUIRE(dw && gz && x, "mconv_wgrad: null tensor");
    E4S_REQUIRE(bs >= 0 && cin >= 1 && cout >= 1 && h >= 1 && w >= 16 && (w % 16) == 0, "mconv_wgrad: bad size (w must be a multiple of 16)");
    E4S_REQUIRE((ks == 1 || ks == 3) && (up == 1 || up == 2) && nreg >= 1 && nreg <= E4S_MAX_REGIONS, "mconv_wgrad: ks 1 / 3, up 1 / 2, nreg 1..%d", E4S_MAX_REGIONS);
    E4S_REQUIRE(labels || nreg == 1, "mconv_wgrad: several regions need a label map");
    E4S_REQUIRE(up == 1 || labels, "mconv_wgrad: up = 2 is the masked composed form");
    E4S_REQUIRE((((uintptr_t)gz | (uintptr_t)labels | (uintptr_t)x) & 15) == 0, "mconv_wgrad: gz, x and labels must be 16-byte aligned");
    const int G = up * up, batch = G * bs;
    E4S_REQUIRE(batch <= 65535, "mconv_wgrad: batch too large");
    if (bs == 0) return 0;
    WgradParams q;
    GemmParams& p = q.g;
    p.c = dw; p.a = gz; p.b = nullptr; p.partial = workspace;
    p.M = cout; p.N = cin * ks * ks; p.K = h * w; p.lda = p.K; p.ldb = 0;
    p.sa = (long long)cout * p.K; p.sb = 0; p.sc = (long long)p.M * p.N; p.batch = batch;
    const int tm_rows = p.M <= 32 ? 32 : (p.M <= 64 ? 64 : GT);
    p.tiles_m = cdiv(p.M, tm_rows); p.tiles_n = cdiv(p.N, GT);
    q.x = x; q.s = s; q.lab = labels; q.bs = bs; q.cin = cin; q.h = h; q.w = w; q.nreg = nreg; q.up = up;
    const int nchunk = cdiv(p.K, GK);
    const int64_t base = (int64_t)p.tiles_m * p.tiles_n * batch;
    int ksplit = 1;
    if (workspace)
        while (base * ksplit < 512 && ksplit * 2 * 4 <= nchunk && (int64_t)(ksplit * 2) * batch * p.M * p.N <= workspace_floats && ksplit < 1024) ksplit *= 2;
    p.ksplit = ksplit;
    p.chunks_per = cdiv(nchunk, ksplit);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * ksplit), 1, batch);
    hipStream_t st = (hipStream_t)stream;
#define E4S_WGRAD(KS_, TM_) hipLaunchKernelGGL((mconv_wgrad_kernel<KS_, TM_>), grid, dim3(GNT), 0, st, q)
    if (ks == 3) { if (tm_rows == 32) E4S_WGRAD(3, 32); else if (tm_rows == 64) E4S_WGRAD(3, 64); else E4S_WGRAD(3, 128); }
    else { if (tm_rows == 32) E4S_WGRAD(1, 32); else if (tm_rows == 64) E4S_WGRAD(1, 64); else E4S_WGRAD(1, 128); }
#undef E4S_WGRAD
    if (ksplit > 1) {
        const int64_t mn = (int64_t)p.M * p.N, total = mn * batch;
        if (total <= 65536 && ksplit >= 16) {
            hipLaunchKernelGGL(gemm_sb_finalize_wide_kernel, dim3((unsigned)cdiv64(total, 64)), dim3(256), 0, st, dw, workspace, (long long)mn, (long long)p.sc, batch, ksplit);
        } else {
            const int g = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
            hipLaunchKernelGGL(gemm_sb_finalize_kernel, dim3(g), dim3(256), 0, st, dw, workspace, (long long)mn, (long long)p.sc, batch, ksplit);
        }
    }
    return check_launch("mconv_wgrad");
}
