// a3-a6: region-aware modulated synthesis kernels (gfx950).
//
//   e4s_modconv_prep_weights : parameter re-layout [Cout,Cin,k,k] -> K-major [par][Cin][k*k][Cout] (+ blur composition for up layers)
//   e4s_style_demod          : per (sample, region) modulation vector s and demodulation vector d (wave-shuffle reductions)
//   e4s_region_modconv3x3    : implicit-GEMM 3x3 conv on fp32 MFMA (v_mfma_f32_32x32x2_f32), LDS-tiled, one pass per layer
//   e4s_region_torgb         : 1x1 modulated conv to 3 channels + bias + fused x2 FIR upsample of the skip (HBM-bound)
//
// GEMM view of the conv:  D[co][pix] = sum_{k=(ci,tap)} A[co][k] * B[k][pix]
//   A = shared un-modulated weights (scale folded in), B = x[ci][pix+tap] * s[region(pix)][ci]  (scaled on the LDS read path),
//   epilogue: * d[region(pix)][co] + noise_w*noise[pix] + bias[co], leaky-relu * sqrt2.
// Pixels are the MFMA "column" index so that each lane owns one pixel: region, noise and the (coalesced along x) store
// address are per-lane constants, and a wave stores 32 consecutive pixels of one channel per instruction.
#include "common.h"

using namespace e4s;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ============================================================================ weight preparation
// Up layers: conv_transpose2d(stride 2, 3x3) followed by upfirdn2d(blur 4x4, pad (1,1)) is, per output parity (a,b),
//   y[2m+a, 2n+b] = sum_{dy,dx in {-1,0,1}} Weff[a,b][dy+1][dx+1] * x[m+dy, n+dx]     (zero padded x)
//   Weff[a,b][dy+1][dx+1] = sum_{ty-ky = 2dy+1-a} sum_{tx-kx = 2dx+1-b} blur[3-ty][3-tx] * W[ky][kx]
// (q = 2m' + k indexes the transposed conv output, y[p] = sum_t z[p+t-1]*blur_flipped[t]).
__global__ __launch_bounds__(256) void prep_weights_kernel(float* __restrict__ wt, const float* __restrict__ weight,
                                                           const float* __restrict__ blur, int cout, int cin, int k, int up,
                                                           float scale) {
    const int kk = k * k;
    const int npar = up ? 4 : 1;
    const int64_t total = (int64_t)npar * cin * kk * cout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % cout);
        int64_t r = i / cout;
        const int tap = (int)(r % kk);
        r /= kk;
        const int ci = (int)(r % cin);
        const int par = (int)(r / cin);
        const float* w = weight + ((size_t)co * cin + ci) * kk;
        float v;
        if (!up) {
            v = w[tap];
        } else {
            const int a = par >> 1, b = par & 1;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            v = 0.f;
            for (int ky = 0; ky < 3; ++ky) {
                const int ty = ky + 2 * dy + 1 - a;
                if (ty < 0 || ty > 3) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int tx = kx + 2 * dx + 1 - b;
                    if (tx < 0 || tx > 3) continue;
                    v += blur[(3 - ty) * 4 + (3 - tx)] * w[ky * 3 + kx];
                }
            }
        }
        wt[i] = v * scale;
    }
}

__global__ __launch_bounds__(256) void wsq_kernel(float* __restrict__ wsq, const float* __restrict__ weight, int cout, int cin, int kk,
                                                  float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // over cin*cout, co fastest
    if (i >= cin * cout) return;
    const int co = i % cout, ci = i / cout;
    const float* w = weight + ((size_t)co * cin + ci) * kk;
    float a = 0.f;
    for (int t = 0; t < kk; ++t) {
        const float v = w[t] * scale;
        a += v * v;
    }
    wsq[i] = a;
}

extern "C" int e4s_modconv_prep_weights(float* wt, float* wsq, const float* weight, const float* blur, int cout, int cin, int k, int up,
                                        void* stream) {
    E4S_REQUIRE(wt && weight, "modconv_prep_weights: null tensor");
    E4S_REQUIRE(k == 1 || k == 3, "modconv_prep_weights: kernel size %d not supported (1 or 3)", k);
    E4S_REQUIRE(cout >= 1 && cin >= 1, "modconv_prep_weights: bad channel counts");
    E4S_REQUIRE(!up || (k == 3 && blur), "modconv_prep_weights: up-conv needs k=3 and the 4x4 blur kernel");
    const float scale = 1.0f / sqrtf((float)cin * k * k);
    const int64_t total = (int64_t)(up ? 4 : 1) * cin * k * k * cout;
    const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prep_weights_kernel, dim3(grid), dim3(256), 0, st, wt, weight, blur, cout, cin, k, up, scale);
    if (wsq) hipLaunchKernelGGL(wsq_kernel, dim3(cdiv(cin * cout, 256)), dim3(256), 0, st, wsq, weight, cout, cin, k * k, scale);
    return check_launch("modconv_prep_weights");
}

// ============================================================================ style + demod tables
// style: one wave per (input channel, group of 8 (sample, region) rows): the channel's modulation row stays in registers and
// is dotted against 8 style vectors with independent accumulators and wave-shuffle reductions.
constexpr int STYLE_ROWS = 8;

__global__ __launch_bounds__(256) void style_kernel(float* __restrict__ s, const float* __restrict__ styles, int64_t stride_b,
                                                    int64_t stride_r, const float* __restrict__ mod_weight,
                                                    const float* __restrict__ mod_bias, int nbr, int nreg, int cin, int sdim, float scale) {
    const int lane = threadIdx.x & 63;
    const int ci = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ci >= cin) return;
    const int br0 = blockIdx.y * STYLE_ROWS;
    const float* wrow = mod_weight + (size_t)ci * sdim;
    const float mb = mod_bias ? mod_bias[ci] : 0.f;
    float acc[STYLE_ROWS];
#pragma unroll
    for (int q = 0; q < STYLE_ROWS; ++q) acc[q] = 0.f;
    const bool vec = (sdim & 3) == 0 && (stride_b & 3) == 0 && (stride_r & 3) == 0 && ((((uintptr_t)styles | (uintptr_t)mod_weight) & 15) == 0);
    if (vec) {
        for (int j = lane * 4; j < sdim; j += 256) {
            const float4 w4 = *reinterpret_cast<const float4*>(wrow + j);
#pragma unroll
            for (int q = 0; q < STYLE_ROWS; ++q) {
                const int br = br0 + q;
                if (br < nbr) {
                    const int b = br / nreg, r = br - b * nreg;
                    const float4 v = *reinterpret_cast<const float4*>(styles + b * stride_b + r * stride_r + j);
                    acc[q] += (v.x * w4.x + v.y * w4.y) + (v.z * w4.z + v.w * w4.w);
                }
            }
        }
    } else {
        for (int j = lane; j < sdim; j += 64) {
            const float wv = wrow[j];
#pragma unroll
            for (int q = 0; q < STYLE_ROWS; ++q) {
                const int br = br0 + q;
                if (br < nbr) {
                    const int b = br / nreg, r = br - b * nreg;
                    acc[q] += styles[b * stride_b + r * stride_r + j] * wv;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < STYLE_ROWS; ++q) {
        const float a = wave_sum(acc[q]);
        if (lane == 0 && br0 + q < nbr) s[(size_t)(br0 + q) * cin + ci] = a * scale + mb;
    }
}

// demod: block = (one (sample, region) row) x (64 output channels); the 4 waves split the input-channel range, lanes own
// output channels (coalesced wsq rows), partial sums meet in LDS.
__global__ __launch_bounds__(256) void demod_kernel(float* __restrict__ d, const float* __restrict__ s, const float* __restrict__ wsq,
                                                    int cin, int cout) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co = blockIdx.x * 64 + lane;
    const int br = blockIdx.y;
    const float* sv = s + (size_t)br * cin;
    const int per = (cin + 3) / 4;
    const int c0 = wave * per, c1 = (c0 + per < cin) ? c0 + per : cin;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (co < cout) {
        int ci = c0;
        for (; ci + 3 < c1; ci += 4) {
            const float t0 = sv[ci], t1 = sv[ci + 1], t2 = sv[ci + 2], t3 = sv[ci + 3];
            a0 += t0 * t0 * wsq[(size_t)ci * cout + co];
            a1 += t1 * t1 * wsq[(size_t)(ci + 1) * cout + co];
            a2 += t2 * t2 * wsq[(size_t)(ci + 2) * cout + co];
            a3 += t3 * t3 * wsq[(size_t)(ci + 3) * cout + co];
        }
        for (; ci < c1; ++ci) {
            const float t = sv[ci];
            a0 += t * t * wsq[(size_t)ci * cout + co];
        }
    }
    part[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && co < cout) d[(size_t)br * cout + co] = rsqrtf(((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) + 1e-8f);
}

extern "C" int e4s_style_demod(float* s, float* d, const float* styles, int64_t stride_b, int64_t stride_r, const float* mod_weight,
                               const float* mod_bias, const float* wsq, int bs, int nreg, int cin, int cout, int sdim, void* stream) {
    E4S_REQUIRE(s && styles && mod_weight, "style_demod: null tensor");
    E4S_REQUIRE(nreg >= 1 && nreg <= E4S_MAX_REGIONS, "style_demod: %d regions (max %d)", nreg, E4S_MAX_REGIONS);
    E4S_REQUIRE(bs >= 0 && cin >= 1 && sdim >= 1, "style_demod: bad size");
    E4S_REQUIRE((d == nullptr) == (wsq == nullptr), "style_demod: d and wsq must both be given or both be NULL");
    if (bs == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int nbr = bs * nreg;
    hipLaunchKernelGGL(style_kernel, dim3(cdiv(cin, 4), cdiv(nbr, STYLE_ROWS)), dim3(256), 0, st, s, styles, stride_b, stride_r, mod_weight,
                       mod_bias, nbr, nreg, cin, sdim, 1.0f / sqrtf((float)sdim));
    if (d) {
        E4S_REQUIRE(cout >= 1, "style_demod: bad cout");
        hipLaunchKernelGGL(demod_kernel, dim3(cdiv(cout, 64), nbr), dim3(256), 0, st, d, s, wsq, cin, cout);
    }
    return check_launch("style_demod");
}

// ---------------------------------------------------------------------------- all layers of a generator in two launches
struct JobTable {
    E4sStyleJob j[E4S_MAX_STYLE_JOBS];
};

constexpr int STYLE_CI = 4;     // input channels per wave: the 8 style rows of a block are loaded once and reused for all of them
constexpr int DEMOD_ROWS = 8;   // (batch, region) rows per block of the demodulation kernel: wsq is re-read nbr / 8 times instead of nbr times

__global__ __launch_bounds__(256) void style_batched_kernel(const JobTable t, int bs, int sdim, float scale) {
    const E4sStyleJob& J = t.j[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int ci0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * STYLE_CI;
    const int nbr = bs * J.nreg;
    const int br0 = blockIdx.y * STYLE_ROWS;
    if (ci0 >= J.cin || br0 >= nbr) return;
    const bool vec = sdim == 512 && (J.stride_b & 3) == 0 && (J.stride_r & 3) == 0 && ((((uintptr_t)J.styles | (uintptr_t)J.mod_weight) & 15) == 0);
    if (vec) {   // the path's shape (style_dim 512): a lane owns 8 of the 512 style entries of every row, in registers
        float4 sv[STYLE_ROWS][2];
#pragma unroll
        for (int q = 0; q < STYLE_ROWS; ++q) {
            const int br = br0 + q < nbr ? br0 + q : nbr - 1;
            const int b = br / J.nreg, r = br - b * J.nreg;
            const float* sp = J.styles + b * J.stride_b + r * J.stride_r;
            sv[q][0] = *reinterpret_cast<const float4*>(sp + lane * 4);
            sv[q][1] = *reinterpret_cast<const float4*>(sp + 256 + lane * 4);
        }
        // the block's 4 channels x 8 rows = 32 dot products: per-lane partial sums first, then ONE transposing butterfly for all 32 (wave_sum_x32: the additions of 32
        // separate wave sums, bit-identical, at a sixth of their cross-lane traffic — the reductions were this kernel's time)
        float part[32];
#pragma unroll
        for (int c = 0; c < STYLE_CI; ++c) {
            const int ci = ci0 + c < J.cin ? ci0 + c : J.cin - 1;
            const float* wrow = J.mod_weight + (size_t)ci * sdim;
            const float4 w0 = *reinterpret_cast<const float4*>(wrow + lane * 4), w1 = *reinterpret_cast<const float4*>(wrow + 256 + lane * 4);
#pragma unroll
            for (int q = 0; q < STYLE_ROWS; ++q) {
                // (same summation order as the one-channel-per-wave version: two float4 steps of j = lane * 4 and 256 + lane * 4)
                float a = (sv[q][0].x * w0.x + sv[q][0].y * w0.y) + (sv[q][0].z * w0.z + sv[q][0].w * w0.w);
                a += (sv[q][1].x * w1.x + sv[q][1].y * w1.y) + (sv[q][1].z * w1.z + sv[q][1].w * w1.w);
                part[c * STYLE_ROWS + q] = a;
            }
        }
        static_assert(STYLE_CI * STYLE_ROWS == 32, "wave_sum_x32 reduces 32 values");
        const float tot = wave_sum_x32(part);
        const int k = wave_sum_x32_index(lane);
        const int c = k / STYLE_ROWS, q = k - c * STYLE_ROWS;
        if ((lane & 1) == 0 && ci0 + c < J.cin && br0 + q < nbr) {
            const float mb = J.mod_bias ? J.mod_bias[ci0 + c] : 0.f;
            J.s[(size_t)(br0 + q) * J.cin + ci0 + c] = tot * scale + mb;
        }
        return;
    }
    for (int c = 0; c < STYLE_CI; ++c) {
        const int ci = ci0 + c;
        if (ci >= J.cin) break;
        const float* wrow = J.mod_weight + (size_t)ci * sdim;
        const float mb = J.mod_bias ? J.mod_bias[ci] : 0.f;
        float acc[STYLE_ROWS];
#pragma unroll
        for (int q = 0; q < STYLE_ROWS; ++q) acc[q] = 0.f;
        for (int j = lane; j < sdim; j += 64) {
            const float wv = wrow[j];
#pragma unroll
            for (int q = 0; q < STYLE_ROWS; ++q) {
                const int br = br0 + q;
                if (br < nbr) {
                    const int b = br / J.nreg, r = br - b * J.nreg;
                    acc[q] += J.styles[b * J.stride_b + r * J.stride_r + j] * wv;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < STYLE_ROWS; ++q) {
            const float a = wave_sum(acc[q]);
            if (lane == 0 && br0 + q < nbr) J.s[(size_t)(br0 + q) * J.cin + ci] = a * scale + mb;
        }
    }
}

constexpr int DEMOD_MAX_CIN = 512;      // widest layer whose style rows a block keeps in LDS (wider ones read them as scalar loads, as before round 4)
__global__ __launch_bounds__(256) void demod_batched_kernel(const JobTable t, int bs) {
    const E4sStyleJob& J = t.j[blockIdx.z];
    __shared__ float part[DEMOD_ROWS][4][64];
    __shared__ __attribute__((aligned(16))) float srow[DEMOD_MAX_CIN][DEMOD_ROWS];     // the block's eight style rows, [ci][row]: one channel's eight values = two 16-byte reads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co = blockIdx.x * 64 + lane;
    const int nbr = bs * J.nreg;
    const int br0 = blockIdx.y * DEMOD_ROWS;
    if (!J.d || br0 >= nbr || blockIdx.x * 64 >= J.cout) return;   // block-uniform exits
    const int per = (J.cin + 3) / 4;
    const int c0 = wave * per, c1 = (c0 + per < J.cin) ? c0 + per : J.cin;
    // The style rows used to come in as wave-uniform scalar loads inside the channel loop: 16 per step, and scalar loads return out of order, so every step waited for
    // all of its own (64 dependent round trips for a 512-channel layer: 38-43 us for tables whose bytes take 5).  Staged in LDS once, the loop's only memory traffic is
    // the coalesced wsq stream, several steps in flight.
    const bool in_lds = J.cin <= DEMOD_MAX_CIN;
    if (in_lds) {
        for (int i = threadIdx.x; i < DEMOD_ROWS * J.cin; i += 256) {
            const int q = i / J.cin, ci = i - q * J.cin;
            srow[ci][q] = J.s[(size_t)(br0 + q < nbr ? br0 + q : nbr - 1) * J.cin + ci];
        }
        __syncthreads();
    }
    // per row the two partial sums (even / odd input channels of this wave's quarter) of the one-row-per-block version, in its order
    float a0[DEMOD_ROWS], a1[DEMOD_ROWS];
#pragma unroll
    for (int q = 0; q < DEMOD_ROWS; ++q) a0[q] = a1[q] = 0.f;
    if (co < J.cout) {
        int ci = c0;
        if (in_lds) {
#pragma unroll 4
            for (; ci + 1 < c1; ci += 2) {
                const float w0 = J.wsq[(size_t)ci * J.cout + co], w1 = J.wsq[(size_t)(ci + 1) * J.cout + co];
#pragma unroll
                for (int q = 0; q < DEMOD_ROWS; ++q) {
                    const float t0 = srow[ci][q], t1 = srow[ci + 1][q];
                    a0[q] += t0 * t0 * w0;
                    a1[q] += t1 * t1 * w1;
                }
            }
        } else {
            for (; ci + 1 < c1; ci += 2) {
                const float w0 = J.wsq[(size_t)ci * J.cout + co], w1 = J.wsq[(size_t)(ci + 1) * J.cout + co];
#pragma unroll
                for (int q = 0; q < DEMOD_ROWS; ++q) {
                    const float* sv = J.s + (size_t)(br0 + q < nbr ? br0 + q : nbr - 1) * J.cin;     // wave-uniform: scalar loads
                    const float t0 = sv[ci], t1 = sv[ci + 1];
                    a0[q] += t0 * t0 * w0;
                    a1[q] += t1 * t1 * w1;
                }
            }
        }
        if (ci < c1) {
            const float w0 = J.wsq[(size_t)ci * J.cout + co];
#pragma unroll
            for (int q = 0; q < DEMOD_ROWS; ++q) {
                const float t0 = J.s[(size_t)(br0 + q < nbr ? br0 + q : nbr - 1) * J.cin + ci];
                a0[q] += t0 * t0 * w0;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < DEMOD_ROWS; ++q) part[q][wave][lane] = a0[q] + a1[q];
    __syncthreads();
    // the four waves finish two rows each
    for (int q = wave; q < DEMOD_ROWS; q += 4)
        if (co < J.cout && br0 + q < nbr)
            J.d[(size_t)(br0 + q) * J.cout + co] = rsqrtf(((part[q][0][lane] + part[q][1][lane]) + (part[q][2][lane] + part[q][3][lane])) + 1e-8f);
}

extern "C" int e4s_style_demod_batched(const E4sStyleJob* jobs, int n_jobs, int bs, int sdim, void* stream) {
    E4S_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= E4S_MAX_STYLE_JOBS, "style_demod_batched: 1..%d jobs", E4S_MAX_STYLE_JOBS);
    E4S_REQUIRE(bs >= 0 && sdim >= 1, "style_demod_batched: bad size");
    if (bs == 0) return 0;
    JobTable t;
    memset(&t, 0, sizeof(t));
    int max_cin = 0, max_cout = 0, max_nbr = 0, any_d = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const E4sStyleJob& J = jobs[i];
        E4S_REQUIRE(J.s && J.styles && J.mod_weight, "style_demod_batched: job %d has a null tensor", i);
        E4S_REQUIRE(J.nreg >= 1 && J.nreg <= E4S_MAX_REGIONS && J.cin >= 1, "style_demod_batched: job %d has a bad size", i);
        E4S_REQUIRE((J.d == nullptr) == (J.wsq == nullptr) && (!J.d || J.cout >= 1), "style_demod_batched: job %d: d and wsq go together", i);
        t.j[i] = J;
        max_cin = J.cin > max_cin ? J.cin : max_cin;
        max_nbr = bs * J.nreg > max_nbr ? bs * J.nreg : max_nbr;
        if (J.d) {
            any_d = 1;
            max_cout = J.cout > max_cout ? J.cout : max_cout;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(style_batched_kernel, dim3(cdiv(max_cin, 4 * STYLE_CI), cdiv(max_nbr, STYLE_ROWS), n_jobs), dim3(256), 0, st, t, bs, sdim,
                       1.0f / sqrtf((float)sdim));
    if (any_d) hipLaunchKernelGGL(demod_batched_kernel, dim3(cdiv(max_cout, 64), cdiv(max_nbr, DEMOD_ROWS), n_jobs), dim3(256), 0, st, t, bs);
    return check_launch("style_demod_batched");
}

// ============================================================================ region-aware 3x3 modulated conv
struct ConvParams {
    float* out;
    const float* x;
    const float* wt;
    const float* s;
    const float* d;
    const uint8_t* labels;
    const float* noise;
    const float* noise_weight;
    const float* act_bias;
    int lh, lw;
    float lscale_y, lscale_x;  // label-map size / output size (PyTorch 'nearest' scale)
    int noise_bstride;
    int act;
    int bs, cin, cout, h, w, nreg, up;
    int tiles_x, tiles_y;
    int ksplit, cin_per;   // split-K over input channels for small feature maps: block ks handles [ks*cin_per, (ks+1)*cin_per)
    float* partial;        // [ksplit][bs][cout][ho*wo] raw accumulators when ksplit > 1
};

constexpr int CK = 8;  // input channels staged per K-chunk (K per chunk = 72)

template <int CB, int PB, int WC, int WP, int LOG_TW>
struct ConvCfg {
    static constexpr int TN = WC * CB * 32;       // output channels per block
    static constexpr int NPB = WP * PB;           // 32-pixel blocks per block
    static constexpr int TW = 1 << LOG_TW;        // tile width in pixels
    static constexpr int RPB = 32 >> LOG_TW;      // rows per pixel block
    static constexpr int TH = NPB * RPB;          // tile height
    static constexpr int PW = TW + 2, PH = TH + 2;
    static constexpr int PATCH = PH * PW;         // staged input patch per channel (with halo)
    static constexpr int EPT = (PATCH + 255) / 256;
    static constexpr int XS = CK * PATCH, WS = CK * 9 * TN, SS = E4S_MAX_REGIONS * CK;
    static constexpr int LDS_FLOATS = XS + WS + SS;
    static_assert(WC * WP == 4, "256-thread blocks");
    static_assert(E4S_MAX_REGIONS * TN <= WS, "demod table overlays the weight stage");
};

template <int CB, int PB, int WC, int WP, int LOG_TW>
__global__ __launch_bounds__(256) void region_modconv_kernel(const ConvParams p) {
    using C = ConvCfg<CB, PB, WC, WP, LOG_TW>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    float* xs = lds;
    float* ws = lds + C::XS;
    float* ss = lds + C::XS + C::WS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l5 = lane & 31, khalf = lane >> 5;
    const int wc = wave / WP, wp = wave % WP;

    const int ntile = p.tiles_x * p.tiles_y;
    const int npar = p.up ? 4 : 1;
    const int ks = blockIdx.x / (ntile * npar);
    const int bx = blockIdx.x - ks * ntile * npar;
    const int tile = bx % ntile;
    const int par = bx / ntile;  // 0 when !up
    const int pa = par >> 1, pb_ = par & 1;
    const int y0 = (tile / p.tiles_x) * C::TH, x0 = (tile % p.tiles_x) * C::TW;
    const int co0 = blockIdx.y * C::TN;
    const int b = blockIdx.z;
    const int hw = p.h * p.w;
    const int ho = p.up ? 2 * p.h : p.h, wo = p.up ? 2 * p.w : p.w;

    // ---- per-thread staging map of the input patch (independent of the channel)
    int goff[C::EPT];
    bool ginb[C::EPT];
#pragma unroll
    for (int j = 0; j < C::EPT; ++j) {
        const int e = tid + j * 256;
        const int py = e / C::PW, px = e - py * C::PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        ginb[j] = (e < C::PATCH) && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        goff[j] = gy * p.w + gx;
    }
    const float* xb = p.x + (size_t)b * p.cin * hw;
    const float* wpar = p.wt + (size_t)par * p.cin * 9 * p.cout;
    const float* sb = p.s + (size_t)b * p.nreg * p.cin;

    // ---- per-lane pixel bookkeeping
    int xoff[PB], cls[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int ty = pbk * C::RPB + (l5 >> LOG_TW), tx = l5 & (C::TW - 1);
        xoff[q] = ty * C::PW + tx;
        const int y = y0 + ty, x = x0 + tx;
        int c = 0;
        if (p.labels) {
            c = E4S_LABEL_NONE;
            if (y < p.h && x < p.w) {
                const int oy = p.up ? 2 * y + pa : y, ox = p.up ? 2 * x + pb_ : x;
                c = p.labels[((size_t)b * p.lh + nearest_src(oy, p.lscale_y, p.lh)) * p.lw + nearest_src(ox, p.lscale_x, p.lw)];
            }
        }
        cls[q] = (c < p.nreg) ? c : -1;
    }

    f32x16 acc[CB][PB];
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
        for (int q = 0; q < PB; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    const bool wvec = (p.cout & 3) == 0;

    const int ci_begin = ks * p.cin_per;
    const int ci_end = (ci_begin + p.cin_per < p.cin) ? ci_begin + p.cin_per : p.cin;
    for (int ci0 = ci_begin; ci0 < ci_end; ci0 += CK) {
        __syncthreads();
        // stage x patch: CK channels x PATCH (zero padded)
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const bool cok = ci0 + c < p.cin;
            const float* xc = xb + (size_t)(ci0 + c) * hw;
#pragma unroll
            for (int j = 0; j < C::EPT; ++j) {
                const int e = tid + j * 256;
                if (e < C::PATCH) xs[c * C::PATCH + e] = (cok && ginb[j]) ? xc[goff[j]] : 0.f;
            }
        }
        // stage weights: ws[c][tap][n] <- wt[par][ci0+c][tap][co0+n]
        if (wvec) {
            constexpr int NV = CK * 9 * C::TN / 4;
            for (int v = tid; v < NV; v += 256) {
                const int n4 = v % (C::TN / 4);
                const int ct = v / (C::TN / 4);  // c*9 + tap
                const int c = ct / 9;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci0 + c < p.cin && co0 + n4 * 4 < p.cout)
                    val = *reinterpret_cast<const float4*>(wpar + ((size_t)ci0 * 9 + ct) * p.cout + co0 + n4 * 4);
                *reinterpret_cast<float4*>(ws + ct * C::TN + n4 * 4) = val;
            }
        } else {
            constexpr int NS = CK * 9 * C::TN;
            for (int v = tid; v < NS; v += 256) {
                const int n = v % C::TN;
                const int ct = v / C::TN;
                const int c = ct / 9;
                ws[v] = (ci0 + c < p.cin && co0 + n < p.cout) ? wpar[((size_t)ci0 * 9 + ct) * p.cout + co0 + n] : 0.f;
            }
        }
        // stage the modulation chunk: ss[r][c] <- s[b][r][ci0+c]
        if (tid < E4S_MAX_REGIONS * CK) {
            const int r = tid / CK, c = tid % CK;
            ss[tid] = (r < p.nreg && ci0 + c < p.cin) ? sb[(size_t)r * p.cin + ci0 + c] : 0.f;
        }
        __syncthreads();

#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
            const int ci = 2 * cp + khalf;  // k index within the MFMA: lanes 0-31 even channel, 32-63 odd channel
            float sv[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) sv[q] = cls[q] >= 0 ? ss[cls[q] * CK + ci] : 0.f;
            const float* xrow = xs + ci * C::PATCH;
            const float* wrow = ws + ci * 9 * C::TN + wc * CB * 32 + l5;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = (tap / 3) * C::PW + (tap % 3);
                float bv[PB], av[CB];
#pragma unroll
                for (int q = 0; q < PB; ++q) bv[q] = xrow[xoff[q] + toff] * sv[q];
#pragma unroll
                for (int i = 0; i < CB; ++i) av[i] = wrow[tap * C::TN + i * 32];
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int q = 0; q < PB; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[q], acc[i][q], 0, 0, 0);
            }
        }
    }

    if (p.ksplit > 1) {  // raw partial sums; modconv_finalize_kernel applies the epilogue after the K-slices are summed
        float* part = p.partial + ((size_t)ks * p.bs + b) * p.cout * ho * wo;
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int pbk = wp * PB + q;
            const int y = y0 + pbk * C::RPB + (l5 >> LOG_TW), x = x0 + (l5 & (C::TW - 1));
            if (y >= p.h || x >= p.w) continue;
            const size_t opix = (size_t)(p.up ? 2 * y + pa : y) * wo + (p.up ? 2 * x + pb_ : x);
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    if (co < p.cout) part[(size_t)co * ho * wo + opix] = acc[i][q][r];
                }
        }
        return;
    }

    // ---- epilogue: demod table through LDS (overlays the weight stage), noise, bias, activation, store
    __syncthreads();
    float* dt = ws;  // [E4S_MAX_REGIONS][TN]
    for (int v = tid; v < E4S_MAX_REGIONS * C::TN; v += 256) {
        const int r = v / C::TN, n = v % C::TN;
        float val = 0.f;
        if (r < p.nreg && co0 + n < p.cout) val = p.d ? p.d[((size_t)b * p.nreg + r) * p.cout + co0 + n] : 1.f;
        dt[v] = val;
    }
    __syncthreads();
    // two passes: all global loads (noise, bias) before the first store — gfx9's single in-order vmcnt would otherwise make every
    // load wait for the stores issued before it
    const float nw = p.noise ? p.noise_weight[0] : 0.f;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int y = y0 + pbk * C::RPB + (l5 >> LOG_TW), x = x0 + (l5 & (C::TW - 1));
        const bool pix_ok = y < p.h && x < p.w;
        const int oy = p.up ? 2 * y + pa : y, ox = p.up ? 2 * x + pb_ : x;
        const size_t opix = (size_t)oy * wo + ox;
        const float nz = (p.noise && pix_ok) ? nw * p.noise[(size_t)b * p.noise_bstride + opix] : 0.f;
        const float* drow = dt + (cls[q] >= 0 ? cls[q] : 0) * C::TN;
        const float dz = cls[q] >= 0 ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int co = co0 + n;
                if (pix_ok && co < p.cout) {
                    float v = acc[i][q][r] * drow[n] * dz + nz;
                    if (p.act_bias) v += p.act_bias[co];
                    if (p.act) v = (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
                    acc[i][q][r] = v;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int pbk = wp * PB + q;
        const int y = y0 + pbk * C::RPB + (l5 >> LOG_TW), x = x0 + (l5 & (C::TW - 1));
        if (y >= p.h || x >= p.w) continue;
        const int oy = p.up ? 2 * y + pa : y, ox = p.up ? 2 * x + pb_ : x;
        const size_t opix = (size_t)oy * wo + ox;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (co < p.cout) p.out[((size_t)b * p.cout + co) * ho * wo + opix] = acc[i][q][r];
            }
        }
    }
}

// Sum the K-slices in a fixed order and apply the StyledConv epilogue (demod, noise, bias, leaky-relu * sqrt2).
__global__ __launch_bounds__(256) void modconv_finalize_kernel(const ConvParams p, int ho, int wo) {
    const size_t ohw = (size_t)ho * wo;
    const size_t per_b = (size_t)p.cout * ohw;
    const size_t total = (size_t)p.bs * per_b;
    const float nw = p.noise ? p.noise_weight[0] : 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / per_b);
        const size_t r = i - (size_t)b * per_b;
        const int co = (int)(r / ohw);
        const size_t opix = r - (size_t)co * ohw;
        const int oy = (int)(opix / wo), ox = (int)(opix - (size_t)oy * wo);
        float a = 0.f;
        for (int k = 0; k < p.ksplit; ++k) a += p.partial[(size_t)k * total + i];
        int c = 0;
        if (p.labels) c = p.labels[((size_t)b * p.lh + nearest_src(oy, p.lscale_y, p.lh)) * p.lw + nearest_src(ox, p.lscale_x, p.lw)];
        float v = 0.f;
        if (c < p.nreg) v = a * (p.d ? p.d[((size_t)b * p.nreg + c) * p.cout + co] : 1.f);
        if (p.noise) v += nw * p.noise[(size_t)b * p.noise_bstride + opix];
        if (p.act_bias) v += p.act_bias[co];
        if (p.act) v = (v > 0.f ? v : v * 0.2f) * 1.41421356237309515f;
        p.out[i] = v;
    }
}

template <int CB, int PB, int WC, int WP, int LOG_TW>
static int launch_conv(ConvParams& p, hipStream_t st, float* workspace, int64_t workspace_floats) {
    using C = ConvCfg<CB, PB, WC, WP, LOG_TW>;
    p.tiles_x = cdiv(p.w, C::TW);
    p.tiles_y = cdiv(p.h, C::TH);
    const int npar = p.up ? 4 : 1;
    const int64_t base = (int64_t)p.tiles_x * p.tiles_y * npar * cdiv(p.cout, C::TN) * p.bs;
    const int ho = p.up ? 2 * p.h : p.h, wo = p.up ? 2 * p.w : p.w;
    const int64_t out_floats = (int64_t)p.bs * p.cout * ho * wo;
    int ksplit = 1;
    if (workspace && base < 192) {  // too few workgroups for 256 CUs: split the reduction over input channels
        const int chunks = cdiv(p.cin, CK);
        while (ksplit < 16 && base * ksplit * 2 <= 512 && ksplit * 2 <= chunks && (int64_t)(ksplit * 2) * out_floats <= workspace_floats) ksplit *= 2;
    }
    p.ksplit = ksplit;
    p.cin_per = cdiv(cdiv(p.cin, ksplit), CK) * CK;
    p.partial = workspace;
    dim3 grid(p.tiles_x * p.tiles_y * npar * ksplit, cdiv(p.cout, C::TN), p.bs);
    hipLaunchKernelGGL((region_modconv_kernel<CB, PB, WC, WP, LOG_TW>), grid, dim3(256), 0, st, p);
    if (ksplit > 1) {
        const int g = (int)(cdiv64(out_floats, 256) < 2048 ? cdiv64(out_floats, 256) : 2048);
        hipLaunchKernelGGL(modconv_finalize_kernel, dim3(g), dim3(256), 0, st, p, ho, wo);
    }
    return check_launch("region_modconv3x3");
}

extern "C" int e4s_region_modconv3x3(float* out, const float* x, const float* wt, const float* s, const float* d, const uint8_t* labels,
                                     int lh, int lw, const float* noise, int noise_bs, const float* noise_weight, const float* act_bias,
                                     int act, int bs, int cin, int cout, int h, int w, int nreg, int up, float* workspace,
                                     int64_t workspace_floats, void* stream) {
    E4S_REQUIRE(out && x && wt && s, "region_modconv3x3: null tensor");
    E4S_REQUIRE(!workspace || workspace_floats >= 0, "region_modconv3x3: bad workspace size");
    E4S_REQUIRE(bs >= 0 && cin >= 1 && cout >= 1 && h >= 1 && w >= 1, "region_modconv3x3: bad size");
    E4S_REQUIRE(nreg >= 1 && nreg <= E4S_MAX_REGIONS, "region_modconv3x3: %d regions (max %d)", nreg, E4S_MAX_REGIONS);
    E4S_REQUIRE(labels || nreg == 1, "region_modconv3x3: nreg > 1 needs a label map");
    E4S_REQUIRE(!labels || (lh >= 1 && lw >= 1), "region_modconv3x3: bad label map size");
    E4S_REQUIRE(!noise || (noise_weight && (noise_bs == 1 || noise_bs == bs)), "region_modconv3x3: noise needs its weight and batch 1 or bs");
    E4S_REQUIRE(bs <= 65535, "region_modconv3x3: batch too large");
    if (bs == 0) return 0;
    ConvParams p;
    p.out = out; p.x = x; p.wt = wt; p.s = s; p.d = d; p.labels = labels; p.noise = noise; p.noise_weight = noise_weight;
    p.act_bias = act_bias; p.lh = lh; p.lw = lw; p.act = act;
    p.bs = bs; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.nreg = nreg; p.up = up ? 1 : 0;
    const int ho = up ? 2 * h : h, wo = up ? 2 * w : w;
    p.lscale_y = labels ? (float)lh / (float)ho : 1.f;
    p.lscale_x = labels ? (float)lw / (float)wo : 1.f;
    p.noise_bstride = (noise && noise_bs > 1) ? ho * wo : 0;
    hipStream_t st = (hipStream_t)stream;
    float* ws = workspace;
    const int64_t wf = workspace_floats;
    if (w >= 32) {
        if (cout > 64) return launch_conv<2, 2, 2, 2, 5>(p, st, ws, wf);   // 128 co x 128 px
        if (cout > 32) return launch_conv<2, 2, 1, 4, 5>(p, st, ws, wf);   //  64 co x 256 px
        return launch_conv<1, 2, 1, 4, 5>(p, st, ws, wf);                  //  32 co x 256 px
    }
    if (w >= 16) return launch_conv<2, 2, 2, 2, 4>(p, st, ws, wf);
    if (w >= 8) return launch_conv<2, 1, 2, 2, 3>(p, st, ws, wf);
    return launch_conv<2, 1, 2, 2, 2>(p, st, ws, wf);
}

// ============================================================================ ToRGB
// out[b,o,p] = sum_ci wt[ci][o] * s[b,c(p),ci] * x[b,ci,p] + bias[o] + upfirdn2d(skip, up=2, pad=(2,1))[p]
// One thread per 4 consecutive pixels (float4 loads of x); the s table [nreg][cin] and wt [cin][3] sit in LDS.
template <bool SINGLE>
__global__ __launch_bounds__(256) void region_torgb_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ wt,
                                                           const float* __restrict__ s, const uint8_t* __restrict__ labels, int lh, int lw,
                                                           float lsy, float lsx, const float* __restrict__ bias,
                                                           const float* __restrict__ skip, const float* __restrict__ upk, int cin, int h,
                                                           int w, int nreg) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* st = sm;                     // [nreg][cin]
    float* wl = sm + nreg * cin;        // [cin][3]
    float* kf = wl + cin * 3;           // [16] flipped upsample taps
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < nreg * cin; i += 256) st[i] = s[(size_t)b * nreg * cin + i];
    // single-region layers (the unmasked 256^2..1024^2 ToRGBs): fold the modulation into the weights once per block
    for (int i = threadIdx.x; i < cin * 3; i += 256) wl[i] = SINGLE ? wt[i] * s[(size_t)b * cin + i / 3] : wt[i];
    if (threadIdx.x < 16) kf[threadIdx.x] = upk ? upk[15 - threadIdx.x] : 0.f;  // kf[ky*4+kx] = k[3-ky][3-kx]
    __syncthreads();
    const int hw = h * w;
    const int q = blockIdx.x * 256 + threadIdx.x;  // pixel quad
    if (q * 4 >= hw) return;
    const int pix = q * 4;
    const int y = pix / w, x0 = pix - y * w;  // w % 4 == 0: the quad stays in one row
    int cls[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = 0;
        if (labels) c = labels[((size_t)b * lh + nearest_src(y, lsy, lh)) * lw + nearest_src(x0 + j, lsx, lw)];
        cls[j] = c < nreg ? c : -1;
    }
    float acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
    const float* xb = x + (size_t)b * cin * hw + pix;
#pragma unroll 8
    for (int ci = 0; ci < cin; ++ci) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)ci * hw);
        const float w0 = wl[ci * 3 + 0], w1 = wl[ci * 3 + 1], w2 = wl[ci * 3 + 2];
        const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = SINGLE ? xv[j] : (cls[j] >= 0 ? xv[j] * st[cls[j] * cin + ci] : 0.f);
            acc[j][0] += t * w0;
            acc[j][1] += t * w1;
            acc[j][2] += t * w2;
        }
    }
    const int hs = h >> 1, wsk = w >> 1;
    float r[3][4];   // all skip loads before the first store (one in-order vmcnt for loads and stores)
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[j][o] + bias[o];
            if (skip) {
                // upfirdn2d(up=2, pad=(2,1), 4x4 kernel): out[p] = sum_{t} in[i0+t] * kflip[k0 + 2t],  mid = p - 1
                const int xx = x0 + j;
                const int iy0 = (y - 1) >> 1, ix0 = (xx - 1) >> 1;  // floor((p-1)/2), arithmetic shift
                const int ky0 = 2 * iy0 + 2 - y, kx0 = 2 * ix0 + 2 - xx;
                const float* sp = skip + ((size_t)b * 3 + o) * hs * wsk;
                float u = 0.f;
#pragma unroll
                for (int ty = 0; ty < 2; ++ty) {
                    const int iy = iy0 + ty;
                    if (iy < 0 || iy >= hs) continue;
#pragma unroll
                    for (int tx = 0; tx < 2; ++tx) {
                        const int ix = ix0 + tx;
                        if (ix < 0 || ix >= wsk) continue;
                        u += sp[(size_t)iy * wsk + ix] * kf[(ky0 + 2 * ty) * 4 + kx0 + 2 * tx];
                    }
                }
                v += u;
            }
            r[o][j] = v;
        }
    }
#pragma unroll
    for (int o = 0; o < 3; ++o)
        *reinterpret_cast<float4*>(out + ((size_t)b * 3 + o) * hw + pix) = make_float4(r[o][0], r[o][1], r[o][2], r[o][3]);
}

// Masked ToRGB at low resolution (<= 128^2, Cin 256..512): few pixels and a long reduction, so the block is 64 pixels wide and
// its NW waves split the input-channel range; partial RGB sums meet in LDS.  NW = 16 (round 4; 4 before): the launch is a chain of dependent round trips
// (table staging, then cin / NW channels per wave, eight loads at a time) and there are never enough pixels to fill the chip — 14 us per launch at 4 waves,
// eight launches per synthesis step.
constexpr int TORGB_NW = 16;
__global__ __launch_bounds__(64 * TORGB_NW) void region_torgb_splitc_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                                  const float* __restrict__ wt, const float* __restrict__ s,
                                                                  const uint8_t* __restrict__ labels, int lh, int lw, float lsy, float lsx,
                                                                  const float* __restrict__ bias, const float* __restrict__ skip,
                                                                  const float* __restrict__ upk, int cin, int h, int w, int nreg) {
    constexpr int NW = TORGB_NW, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* st = sm;                     // [nreg][cin]
    float* wl = sm + nreg * cin;        // [cin][3]
    float* kf = wl + cin * 3;           // [16]
    float* red = kf + 16;               // [NW][3][64]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < nreg * cin; i += NT) st[i] = s[(size_t)b * nreg * cin + i];
    for (int i = threadIdx.x; i < cin * 3; i += NT) wl[i] = wt[i];
    if (threadIdx.x < 16) kf[threadIdx.x] = upk ? upk[15 - threadIdx.x] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hw = h * w;
    const int pix = blockIdx.x * 64 + lane;
    const bool ok = pix < hw;
    const int y = ok ? pix / w : 0, xx = ok ? pix - (pix / w) * w : 0;
    int c = 0;
    if (labels && ok) c = labels[((size_t)b * lh + nearest_src(y, lsy, lh)) * lw + nearest_src(xx, lsx, lw)];
    const int cls = c < nreg ? c : -1;
    const int per = (cin + NW - 1) / NW;
    const int c0 = wave * per < cin ? wave * per : cin, c1 = (c0 + per < cin) ? c0 + per : cin;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (ok && cls >= 0) {
        const float* xb = x + (size_t)b * cin * hw + pix;
        const float* sr = st + cls * cin;
#pragma unroll 8
        for (int ci = c0; ci < c1; ++ci) {
            const float t = xb[(size_t)ci * hw] * sr[ci];
            a0 += t * wl[ci * 3 + 0];
            a1 += t * wl[ci * 3 + 1];
            a2 += t * wl[ci * 3 + 2];
        }
    }
    red[(wave * 3 + 0) * 64 + lane] = a0;
    red[(wave * 3 + 1) * 64 + lane] = a1;
    red[(wave * 3 + 2) * 64 + lane] = a2;
    __syncthreads();
    if (wave != 0 || !ok) return;
    const int hs = h >> 1, wsk = w >> 1;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float q4[4];                    // four groups of four consecutive channel slices, each ((a + b) + (c + d)), then the same over the groups
#pragma unroll
        for (int g = 0; g < 4; ++g)
            q4[g] = (red[((4 * g + 0) * 3 + o) * 64 + lane] + red[((4 * g + 1) * 3 + o) * 64 + lane]) + (red[((4 * g + 2) * 3 + o) * 64 + lane] + red[((4 * g + 3) * 3 + o) * 64 + lane]);
        float v = ((q4[0] + q4[1]) + (q4[2] + q4[3])) + bias[o];
        if (skip) {
            const int iy0 = (y - 1) >> 1, ix0 = (xx - 1) >> 1;
            const int ky0 = 2 * iy0 + 2 - y, kx0 = 2 * ix0 + 2 - xx;
            const float* sp = skip + ((size_t)b * 3 + o) * hs * wsk;
            float u = 0.f;
#pragma unroll
            for (int ty = 0; ty < 2; ++ty) {
                const int iy = iy0 + ty;
                if (iy < 0 || iy >= hs) continue;
#pragma unroll
                for (int tx = 0; tx < 2; ++tx) {
                    const int ix = ix0 + tx;
                    if (ix < 0 || ix >= wsk) continue;
                    u += sp[(size_t)iy * wsk + ix] * kf[(ky0 + 2 * ty) * 4 + kx0 + 2 * tx];
                }
            }
            v += u;
        }
        out[((size_t)b * 3 + o) * hw + pix] = v;
    }
}

extern "C" int e4s_region_torgb(float* out, const float* x, const float* wt, const float* s, const uint8_t* labels, int lh, int lw,
                                const float* bias, const float* skip, const float* up_kernel, int bs, int cin, int h, int w, int nreg,
                                void* stream) {
    E4S_REQUIRE(out && x && wt && s && bias, "region_torgb: null tensor");
    E4S_REQUIRE(bs >= 0 && cin >= 1 && h >= 1 && w >= 1, "region_torgb: bad size");
    E4S_REQUIRE(nreg >= 1 && nreg <= E4S_MAX_REGIONS, "region_torgb: %d regions (max %d)", nreg, E4S_MAX_REGIONS);
    E4S_REQUIRE(labels || nreg == 1, "region_torgb: nreg > 1 needs a label map");
    E4S_REQUIRE(!skip || (up_kernel && (h % 2) == 0), "region_torgb: skip needs the 4x4 upsample kernel and even size");
    E4S_REQUIRE(bs <= 65535, "region_torgb: batch too large");
    if (bs == 0) return 0;
    const size_t shm = ((size_t)nreg * cin + cin * 3 + 16 + TORGB_NW * 3 * 64) * sizeof(float);
    E4S_REQUIRE(shm <= 64 * 1024, "region_torgb: style table does not fit LDS (cin=%d nreg=%d)", cin, nreg);
    const float lsy = labels ? (float)lh / (float)h : 1.f, lsx = labels ? (float)lw / (float)w : 1.f;
    if ((w % 4) != 0 || (int64_t)h * w * bs < 262144) {  // small maps (or ragged widths): 64-pixel blocks, channels split over the waves
        hipLaunchKernelGGL(region_torgb_splitc_kernel, dim3(cdiv(h * w, 64), bs), dim3(64 * TORGB_NW), shm, (hipStream_t)stream, out, x, wt, s, labels, lh,
                           lw, lsy, lsx, bias, skip, up_kernel, cin, h, w, nreg);
        return check_launch("region_torgb");
    }
    dim3 grid(cdiv(h * w / 4, 256), bs);
    if (!labels && nreg == 1)
        hipLaunchKernelGGL(region_torgb_kernel<true>, grid, dim3(256), shm, (hipStream_t)stream, out, x, wt, s, labels, lh, lw, lsy, lsx, bias,
                           skip, up_kernel, cin, h, w, nreg);
    else
        hipLaunchKernelGGL(region_torgb_kernel<false>, grid, dim3(256), shm, (hipStream_t)stream, out, x, wt, s, labels, lh, lw, lsy, lsx, bias,
                           skip, up_kernel, cin, h, w, nreg);
    return check_launch("region_torgb");
}
