// a8 support kernels (regional-style encoder): per-plane statistics (InstanceNorm), squeeze-excite gate, fused
// InstanceNorm-apply * gate + shortcut, masked average pooling per region, bilinear resize.  All HBM/L2-streaming.
#include "common.h"

using namespace e4s;

// block-wide sum of one float per thread (256 threads); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------ plane statistics
// One block per (b, c) plane.  Two passes (the plane is L2-resident): mean, then centred second moment — the same
// biased variance InstanceNorm2d uses (no affine, no running stats).  nmean = mean of the NORMALISED plane, which is what
// the encoder's SE squeeze sees (it is 0 up to rounding; computed, not assumed).
__global__ __launch_bounds__(256) void plane_stats_kernel(float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ nmean,
                                                          const float* __restrict__ x, int hw, float eps) {
    __shared__ float sh[4];
    const float* xp = x + (size_t)blockIdx.x * hw;
    float s = 0.f;
    if ((hw & 3) == 0) {
        for (int i = threadIdx.x * 4; i < hw; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(xp + i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += 256) s += xp[i];
    }
    const float m = block_sum(s, sh) / (float)hw;
    if (threadIdx.x == 0) mean[blockIdx.x] = m;
    if (!rstd) return;
    float q = 0.f, c1 = 0.f;
    if ((hw & 3) == 0) {
        for (int i = threadIdx.x * 4; i < hw; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(xp + i);
            const float a = v.x - m, b = v.y - m, c = v.z - m, d = v.w - m;
            q += (a * a + b * b) + (c * c + d * d);
            c1 += (a + b) + (c + d);
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += 256) {
            const float a = xp[i] - m;
            q += a * a;
            c1 += a;
        }
    }
    const float var = block_sum(q, sh) / (float)hw;
    const float r = 1.0f / sqrtf(var + eps);
    const float cs = block_sum(c1, sh);
    if (threadIdx.x == 0) {
        rstd[blockIdx.x] = r;
        if (nmean) nmean[blockIdx.x] = cs * r / (float)hw;
    }
}

extern "C" int e4s_plane_stats(float* mean, float* rstd, float* nmean, const float* x, int planes, int hw, float eps, void* stream) {
    E4S_REQUIRE(mean && x, "plane_stats: null tensor");
    E4S_REQUIRE(planes >= 0 && hw >= 1, "plane_stats: bad size");
    E4S_REQUIRE(!nmean || rstd, "plane_stats: nmean needs rstd");
    if (planes == 0) return 0;
    hipLaunchKernelGGL(plane_stats_kernel, dim3(planes), dim3(256), 0, (hipStream_t)stream, mean, rstd, nmean, x, hw, eps);
    return check_launch("plane_stats");
}

// ------------------------------------------------------------------------------------ small fully-connected on [bs, C] vectors
// y[b, o] = act( bn( sum_i W[o, i] * x[b, i] ) );  bn (optional, eval-mode BatchNorm) = (v - mean)*gamma/sqrt(var+eps) + beta.
// One wave per (b, o).  Used for the SE gate, ARM / FFM attentions and the global-context 1x1 ConvBNReLU of BiSeNet.
__global__ __launch_bounds__(256) void vec_fc_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ var, float eps, int act,
                                                     int cin, int cout) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (o >= cout) return;
    const float* xr = x + (size_t)b * cin;
    const float* wr = W + (size_t)o * cin;
    float a = 0.f;
    for (int i = lane; i < cin; i += 64) a += wr[i] * xr[i];
    a = wave_sum(a);
    if (lane == 0) {
        if (var) a = (a - mean[o]) * (gamma[o] / sqrtf(var[o] + eps)) + beta[o];
        if (act == 1) a = fmaxf(a, 0.f);
        if (act == 3) a = 1.0f / (1.0f + expf(-a));
        y[(size_t)b * cout + o] = a;
    }
}

extern "C" int e4s_vec_fc(float* y, const float* x, const float* W, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                          const float* bn_var, float bn_eps, int act, int bs, int cin, int cout, void* stream) {
    E4S_REQUIRE(y && x && W, "vec_fc: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && cin >= 1 && cout >= 1, "vec_fc: bad size");
    E4S_REQUIRE(act == 0 || act == 1 || act == 3, "vec_fc: act must be 0 (none), 1 (relu) or 3 (sigmoid)");
    E4S_REQUIRE(!bn_var || (bn_gamma && bn_beta && bn_mean), "vec_fc: incomplete BatchNorm parameters");
    if (bs == 0) return 0;
    hipLaunchKernelGGL(vec_fc_kernel, dim3(cdiv(cout, 4), bs), dim3(256), 0, (hipStream_t)stream, y, x, W, bn_gamma, bn_beta, bn_mean, bn_var,
                       bn_eps, act, cin, cout);
    return check_launch("vec_fc");
}

// ------------------------------------------------------------------------------------ fused normalise * gate + shortcut (+ PReLU)
// out[b,c,y,x] = act( ((x - mean[b,c]) * rstd[b,c]) * gate[b,c] + sc' )
//   sc' = shortcut[b,c,y*ss,x*ss] (MaxPool2d(1, ss) == strided subsample), optionally instance-normalised with (sc_mean, sc_rstd)
// Any of mean/rstd, gate, shortcut, prelu may be NULL.
__global__ __launch_bounds__(256) void norm_gate_add_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gate,
                                                            const float* __restrict__ sc, const float* __restrict__ sc_mean,
                                                            const float* __restrict__ sc_rstd, int ss, const float* __restrict__ prelu, int C,
                                                            int h, int w) {
    const int plane = blockIdx.y;  // b*C + c
    const int c = plane % C;
    const int hw = h * w;
    const float m = mean ? mean[plane] : 0.f, r = rstd ? rstd[plane] : 1.f, g = gate ? gate[plane] : 1.f;
    const float sm = sc_mean ? sc_mean[plane] : 0.f, sr = sc_rstd ? sc_rstd[plane] : 1.f;
    const float sl = prelu ? prelu[c] : 1.f;
    const float* xp = x + (size_t)plane * hw;
    float* op = out + (size_t)plane * hw;
    const float* sp = sc ? sc + (size_t)plane * hw * ss * ss : nullptr;
    const int ws_ = w * ss;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < hw; i += gridDim.x * 1024) {
        float v[4];
        if ((hw & 3) == 0) {
            const float4 t = *reinterpret_cast<const float4*>(xp + i);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = i + j < hw ? xp[i + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = (v[j] - m) * r * g;
            if (sp && i + j < hw) {
                const int yy = (i + j) / w, xx = (i + j) - yy * w;
                const float s = (ss == 1) ? sp[i + j] : sp[(size_t)yy * ss * ws_ + xx * ss];
                t += (s - sm) * sr;
            }
            v[j] = t > 0.f ? t : t * sl;
        }
        if ((hw & 3) == 0) {
            *reinterpret_cast<float4*>(op + i) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i + j < hw) op[i + j] = v[j];
        }
    }
}

extern "C" int e4s_norm_gate_add(float* out, const float* x, const float* mean, const float* rstd, const float* gate, const float* shortcut,
                                 const float* sc_mean, const float* sc_rstd, int sc_stride, const float* prelu, int bs, int C, int h, int w,
                                 void* stream) {
    E4S_REQUIRE(out && x, "norm_gate_add: null tensor");
    E4S_REQUIRE(bs >= 0 && C >= 1 && h >= 1 && w >= 1 && (int64_t)bs * C <= 65535, "norm_gate_add: bad size");
    E4S_REQUIRE((mean == nullptr) == (rstd == nullptr) && (sc_mean == nullptr) == (sc_rstd == nullptr), "norm_gate_add: mean/rstd go together");
    E4S_REQUIRE(!shortcut || sc_stride >= 1, "norm_gate_add: bad shortcut stride");
    if (bs == 0) return 0;
    const int gx = cdiv(h * w, 1024) < 64 ? cdiv(h * w, 1024) : 64;
    hipLaunchKernelGGL(norm_gate_add_kernel, dim3(gx, bs * C), dim3(256), 0, (hipStream_t)stream, out, x, mean, rstd, gate, shortcut, sc_mean,
                       sc_rstd, shortcut ? sc_stride : 1, prelu, C, h, w);
    return check_launch("norm_gate_add");
}

// ------------------------------------------------------------------------------------ squeeze-excite gate in one launch
// gate[b, o] = sigmoid( sum_j W2[o, j] * relu( sum_i W1[j, i] * pooled[b, i] ) )      (W1 [H, C], W2 [C, H], no biases; H <= 64)
// Workgroup = (64 outputs, image): its four waves first compute the H hidden units (every workgroup of an image repeats them: H*C MACs),
// then 16 outputs each.  Per value the same lane-strided products and wave reduction as vec_fc_kernel: bit-identical to the two launches.
__global__ __launch_bounds__(256) void se_gate_kernel(float* __restrict__ gate, const float* __restrict__ pooled, const float* __restrict__ W1,
                                                      const float* __restrict__ W2, int C, int H) {
    __shared__ float hid[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const float* xr = pooled + (size_t)b * C;
    // Eight hidden units / outputs at a time with independent accumulators: written one value after the other (round 2) every dot product waited for its own
    // loads — 8 + 16 dependent memory round trips, 31 us for a launch whose work is microseconds (192 launches per batch of 8 swaps).  Same products, same order
    // per value: bit-identical.
    for (int j0 = wave; j0 < H; j0 += 32) {
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = 0.f;
#pragma unroll 4
        for (int i = lane; i < C; i += 64) {
            const float xv = xr[i];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = j0 + 4 * q;
                const float wv = W1[(size_t)(j < H ? j : H - 1) * C + i];      // (unconditional load: no branch between the requests)
                if (j < H) a[q] += wv * xv;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j0 + 4 * q;
            const float t = wave_sum(a[q]);
            if (j < H && lane == 0) hid[j] = fmaxf(t, 0.f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 8) {
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int o = blockIdx.x * 64 + wave * 16 + k0 + q;
            a[q] = 0.f;
            if (o < C)
                for (int i = lane; i < H; i += 64) a[q] += W2[(size_t)o * H + i] * hid[i];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int o = blockIdx.x * 64 + wave * 16 + k0 + q;
            const float t = wave_sum(a[q]);
            if (o < C && lane == 0) gate[(size_t)b * C + o] = 1.0f / (1.0f + expf(-t));
        }
    }
}

extern "C" int e4s_se_gate(float* gate, const float* pooled, const float* w1, const float* w2, int bs, int C, int H, void* stream) {
    E4S_REQUIRE(gate && pooled && w1 && w2, "se_gate: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && C >= 1 && H >= 1 && H <= 64, "se_gate: bad size (hidden width 1..64)");
    if (bs == 0) return 0;
    hipLaunchKernelGGL(se_gate_kernel, dim3(cdiv(C, 64), bs), dim3(256), 0, (hipStream_t)stream, gate, pooled, w1, w2, C, H);
    return check_launch("se_gate");
}

// ------------------------------------------------------------------------------------ norm_gate_add that also returns the statistics of its OUTPUT
// The next IR-SE unit starts with an InstanceNorm of this output: one workgroup per plane keeps the plane's output values in registers
// (IT float4 per thread: planes up to 1024 * IT pixels) and produces mean / rstd with the very sums of plane_stats_kernel (same thread ->
// element mapping, same two passes), so a unit no longer needs a statistics launch of its own.
// SELF: the statistics of the INPUT plane are computed here too (in_eps; `mean` / `rstd` unused) — the plane sits in registers anyway, and its sums are plane_stats_kernel's
// (same thread -> element mapping, same two passes): a unit whose gate does not depend on them (ops.SE_GATE_IS_HALF) then needs no statistics launch between its second
// convolution and this kernel.
template <int IT, bool SELF = false>
__global__ __launch_bounds__(256) void norm_gate_add_stats_kernel(float* __restrict__ out, float* __restrict__ omean, float* __restrict__ orstd,
                                                                  const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gate, const float* __restrict__ sc, const float* __restrict__ sc_mean,
                                                                  const float* __restrict__ sc_rstd, int ss, const float* __restrict__ prelu, int C, int h, int w,
                                                                  float eps, float in_eps) {
    __shared__ float sh[4];
    const int plane = blockIdx.x;  // b*C + c
    const int c = plane % C;
    const int hw = h * w;
    float m = mean ? mean[plane] : 0.f, r = rstd ? rstd[plane] : 1.f;
    const float g = gate ? gate[plane] : 1.f;
    const float sm = sc_mean ? sc_mean[plane] : 0.f, sr = sc_rstd ? sc_rstd[plane] : 1.f;
    const float sl = prelu ? prelu[c] : 1.f;
    const float* xp = x + (size_t)plane * hw;
    float* op = out + (size_t)plane * hw;
    const float* sp = sc ? sc + (size_t)plane * hw * ss * ss : nullptr;
    const int ws_ = w * ss;
    float4 v[IT];
    // (round 5) a same-resolution shortcut is requested up front, 16 bytes per lane, beside the plane itself: it used to be read element by element BEHIND the two
    // reductions of the statistics — a second round trip to memory per workgroup and four times the requests (128 -> 128 @128: 124 -> 9x us per launch)
    const bool sc4 = SELF && sp && ss == 1;
    float4 sv4[SELF ? IT : 1];
    if constexpr (SELF) {
        float s0 = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = threadIdx.x * 4 + it * 1024;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            sv4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < hw) {
                v[it] = *reinterpret_cast<const float4*>(xp + i);
                if (sc4) sv4[it] = *reinterpret_cast<const float4*>(sp + i);
                s0 += (v[it].x + v[it].y) + (v[it].z + v[it].w);
            }
        }
        m = block_sum(s0, sh) / (float)hw;
        float q0 = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = threadIdx.x * 4 + it * 1024;
            if (i < hw) {
                const float a = v[it].x - m, b2 = v[it].y - m, c2 = v[it].z - m, d = v[it].w - m;
                q0 += (a * a + b2 * b2) + (c2 * c2 + d * d);
            }
        }
        r = 1.0f / sqrtf(block_sum(q0, sh) / (float)hw + in_eps);
    }
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x * 4 + it * 1024;
        if constexpr (!SELF) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < hw) {
            const float4 t = SELF ? v[it] : *reinterpret_cast<const float4*>(xp + i);
            float e[4] = {t.x, t.y, t.z, t.w};
            const float pre[4] = {sv4[SELF ? it : 0].x, sv4[SELF ? it : 0].y, sv4[SELF ? it : 0].z, sv4[SELF ? it : 0].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = (e[j] - m) * r * g;
                if (sc4) {
                    u += (pre[j] - sm) * sr;
                } else if (sp) {
                    const int yy = (i + j) / w, xx = (i + j) - yy * w;
                    const float sv = (ss == 1) ? sp[i + j] : sp[(size_t)yy * ss * ws_ + xx * ss];
                    u += (sv - sm) * sr;
                }
                e[j] = u > 0.f ? u : u * sl;
            }
            v[it] = make_float4(e[0], e[1], e[2], e[3]);
            *reinterpret_cast<float4*>(op + i) = v[it];
            s += (e[0] + e[1]) + (e[2] + e[3]);
        }
    }
    const float mo = block_sum(s, sh) / (float)hw;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x * 4 + it * 1024;
        if (i < hw) {
            const float a = v[it].x - mo, b2 = v[it].y - mo, c2 = v[it].z - mo, d = v[it].w - mo;
            q += (a * a + b2 * b2) + (c2 * c2 + d * d);
        }
    }
    const float var = block_sum(q, sh) / (float)hw;
    if (threadIdx.x == 0) {
        omean[plane] = mo;
        orstd[plane] = 1.0f / sqrtf(var + eps);
    }
}

extern "C" int e4s_norm_gate_add_stats(float* out, float* out_mean, float* out_rstd, const float* x, const float* mean, const float* rstd,
                                       const float* gate, const float* shortcut, const float* sc_mean, const float* sc_rstd, int sc_stride,
                                       const float* prelu, int bs, int C, int h, int w, float eps, void* stream) {
    E4S_REQUIRE(out && out_mean && out_rstd && x, "norm_gate_add_stats: null tensor");
    E4S_REQUIRE(bs >= 0 && C >= 1 && h >= 1 && w >= 1 && (int64_t)bs * C <= 0x7fffffff, "norm_gate_add_stats: bad size");
    E4S_REQUIRE(((h * w) & 3) == 0 && h * w <= 16384, "norm_gate_add_stats: planes of 4 .. 16384 pixels, a multiple of 4 (use norm_gate_add + plane_stats otherwise)");
    E4S_REQUIRE((mean == nullptr) == (rstd == nullptr) && (sc_mean == nullptr) == (sc_rstd == nullptr), "norm_gate_add_stats: mean/rstd go together");
    E4S_REQUIRE(!shortcut || sc_stride >= 1, "norm_gate_add_stats: bad shortcut stride");
    if (bs == 0) return 0;
    const int hw = h * w, ss = shortcut ? sc_stride : 1;
    const dim3 grid(bs * C), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (hw <= 1024)
        hipLaunchKernelGGL((norm_gate_add_stats_kernel<1, false>), grid, block, 0, st, out, out_mean, out_rstd, x, mean, rstd, gate, shortcut, sc_mean, sc_rstd, ss, prelu, C, h, w, eps, 0.f);
    else if (hw <= 4096)
        hipLaunchKernelGGL((norm_gate_add_stats_kernel<4, false>), grid, block, 0, st, out, out_mean, out_rstd, x, mean, rstd, gate, shortcut, sc_mean, sc_rstd, ss, prelu, C, h, w, eps, 0.f);
    else
        hipLaunchKernelGGL((norm_gate_add_stats_kernel<16, false>), grid, block, 0, st, out, out_mean, out_rstd, x, mean, rstd, gate, shortcut, sc_mean, sc_rstd, ss, prelu, C, h, w, eps, 0.f);
    return check_launch("norm_gate_add_stats");
}

// ... and for planes of at most 1 024 pixels (the 32 x 32 and 16 x 16 units: 17 of the encoder's 24) ONE WAVE per plane, four planes per block: the plane is 16 values per
// lane, every reduction a wave butterfly, no barrier at all (a block per plane spent its time in four block reductions of two barriers each: 21 us per launch).
__global__ __launch_bounds__(256) void norm_self_wave_kernel(float* __restrict__ out, float* __restrict__ omean, float* __restrict__ orstd, const float* __restrict__ x,
                                                             const float* __restrict__ gate, const float* __restrict__ sc, const float* __restrict__ sc_mean,
                                                             const float* __restrict__ sc_rstd, int ss, const float* __restrict__ prelu, int planes, int C, int h, int w,
                                                             float eps, float in_eps) {
    const int lane = threadIdx.x & 63;
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int c = plane % C;
    const int hw = h * w;
    const float g = gate ? gate[plane] : 1.f;
    const float sm = sc_mean ? sc_mean[plane] : 0.f, sr = sc_rstd ? sc_rstd[plane] : 1.f;
    const float sl = prelu ? prelu[c] : 1.f;
    const float* xp = x + (size_t)plane * hw;
    float* op = out + (size_t)plane * hw;
    const float* sp = sc ? sc + (size_t)plane * hw * ss * ss : nullptr;
    const int ws_ = w * ss;
    float4 v[4];
    float s0 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = lane * 4 + it * 256;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < hw) {
            v[it] = *reinterpret_cast<const float4*>(xp + i);
            s0 += (v[it].x + v[it].y) + (v[it].z + v[it].w);
        }
    }
    const float m = wave_sum(s0) / (float)hw;
    float q0 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = lane * 4 + it * 256;
        if (i < hw) {
            const float a = v[it].x - m, b2 = v[it].y - m, c2 = v[it].z - m, d = v[it].w - m;
            q0 += (a * a + b2 * b2) + (c2 * c2 + d * d);
        }
    }
    const float r = 1.0f / sqrtf(wave_sum(q0) / (float)hw + in_eps);
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = lane * 4 + it * 256;
        if (i < hw) {
            float e[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = (e[j] - m) * r * g;
                if (sp) {
                    const int yy = (i + j) / w, xx = (i + j) - yy * w;
                    const float sv = (ss == 1) ? sp[i + j] : sp[(size_t)yy * ss * ws_ + xx * ss];
                    u += (sv - sm) * sr;
                }
                e[j] = u > 0.f ? u : u * sl;
            }
            v[it] = make_float4(e[0], e[1], e[2], e[3]);
            *reinterpret_cast<float4*>(op + i) = v[it];
            s += (e[0] + e[1]) + (e[2] + e[3]);
        }
    }
    const float mo = wave_sum(s) / (float)hw;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = lane * 4 + it * 256;
        if (i < hw) {
            const float a = v[it].x - mo, b2 = v[it].y - mo, c2 = v[it].z - mo, d = v[it].w - mo;
            q += (a * a + b2 * b2) + (c2 * c2 + d * d);
        }
    }
    const float var = wave_sum(q) / (float)hw;
    if (lane == 0) {
        omean[plane] = mo;
        orstd[plane] = 1.0f / sqrtf(var + eps);
    }
}

// The same for planes of up to 65 536 pixels without a shortcut (the encoder's input layer, psp_encoders.py:335-336: InstanceNorm2d(64) + PReLU on 256 x 256 maps): 1 024 threads per
// plane, 16 float4 each — statistics, normalisation, PReLU and the statistics of the result in one launch instead of plane_stats + norm_gate_add + plane_stats (three passes
// over 268 MB per batch of 16 images).
__device__ __forceinline__ float block_sum16(float v, float* sh) {       // 1 024 threads; result valid in every thread
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k];
    return t;
}
__global__ __launch_bounds__(1024) void norm_self_stats_big_kernel(float* __restrict__ out, float* __restrict__ omean, float* __restrict__ orstd, const float* __restrict__ x,
                                                                   const float* __restrict__ gate, const float* __restrict__ prelu, int C, int hw, float eps, float in_eps) {
    constexpr int IT = 16;
    __shared__ float sh[16];
    const int plane = blockIdx.x, c = plane % C;
    const float g = gate ? gate[plane] : 1.f, sl = prelu ? prelu[c] : 1.f;
    const float* xp = x + (size_t)plane * hw;
    float* op = out + (size_t)plane * hw;
    float4 v[IT];
    float s0 = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x * 4 + it * 4096;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < hw) {
            v[it] = *reinterpret_cast<const float4*>(xp + i);
            s0 += (v[it].x + v[it].y) + (v[it].z + v[it].w);
        }
    }
    const float m = block_sum16(s0, sh) / (float)hw;
    float q0 = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x * 4 + it * 4096;
        if (i < hw) {
            const float a = v[it].x - m, b2 = v[it].y - m, c2 = v[it].z - m, d = v[it].w - m;
            q0 += (a * a + b2 * b2) + (c2 * c2 + d * d);
        }
    }
    const float r = 1.0f / sqrtf(block_sum16(q0, sh) / (float)hw + in_eps);
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = threadIdx.x * 4 + it * 4096;
        if (i < hw) {
            float e[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = (e[j] - m) * r * g;
                e[j] = u > 0.f ? u : u * sl;
            }
            *reinterpret_cast<float4*>(op + i) = make_float4(e[0], e[1], e[2], e[3]);
            s += (e[0] + e[1]) + (e[2] + e[3]);
        }
    }
    const float mo = block_sum16(s, sh) / (float)hw;          // (the barrier inside also orders this thread's stores before its re-reads below)
    float q = 0.f;
#pragma unroll 4
    for (int it = 0; it < IT; ++it) {                          // second moment of the result: its values come back from the cache (128 registers per thread hold one copy of the plane, not two)
        const int i = threadIdx.x * 4 + it * 4096;
        if (i < hw) {
            const float4 w4 = *reinterpret_cast<const float4*>(op + i);
            const float a = w4.x - mo, b2 = w4.y - mo, c2 = w4.z - mo, d = w4.w - mo;
            q += (a * a + b2 * b2) + (c2 * c2 + d * d);
        }
    }
    const float var = block_sum16(q, sh) / (float)hw;
    if (threadIdx.x == 0) {
        omean[plane] = mo;
        orstd[plane] = 1.0f / sqrtf(var + eps);
    }
}

extern "C" int e4s_norm_self_gate_add_stats(float* out, float* out_mean, float* out_rstd, const float* x, float in_eps, const float* gate, const float* shortcut,
                                            const float* sc_mean, const float* sc_rstd, int sc_stride, const float* prelu, int bs, int C, int h, int w, float eps,
                                            void* stream) {
    E4S_REQUIRE(out && out_mean && out_rstd && x, "norm_self_gate_add_stats: null tensor");
    E4S_REQUIRE(bs >= 0 && C >= 1 && h >= 1 && w >= 1 && (int64_t)bs * C <= 0x7fffffff, "norm_self_gate_add_stats: bad size");
    E4S_REQUIRE(((h * w) & 3) == 0 && (h * w <= 16384 || (h * w <= 65536 && !shortcut)), "norm_self_gate_add_stats: planes of 4 .. 16384 pixels (65536 without a shortcut), a multiple of 4");
    E4S_REQUIRE((sc_mean == nullptr) == (sc_rstd == nullptr), "norm_self_gate_add_stats: sc_mean / sc_rstd go together");
    E4S_REQUIRE(!shortcut || sc_stride >= 1, "norm_self_gate_add_stats: bad shortcut stride");
    if (bs == 0) return 0;
    const int hw = h * w, ss = shortcut ? sc_stride : 1;
    const dim3 grid(bs * C), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (hw > 16384) {
        hipLaunchKernelGGL(norm_self_stats_big_kernel, grid, dim3(1024), 0, st, out, out_mean, out_rstd, x, gate, prelu, C, hw, eps, in_eps);
        return check_launch("norm_self_gate_add_stats");
    }
    if (hw <= 1024)
        hipLaunchKernelGGL(norm_self_wave_kernel, dim3(cdiv(bs * C, 4)), block, 0, st, out, out_mean, out_rstd, x, gate, shortcut, sc_mean, sc_rstd, ss, prelu, bs * C, C, h, w, eps, in_eps);
    else if (hw <= 4096)
        hipLaunchKernelGGL((norm_gate_add_stats_kernel<4, true>), grid, block, 0, st, out, out_mean, out_rstd, x, nullptr, nullptr, gate, shortcut, sc_mean, sc_rstd, ss, prelu, C, h, w, eps, in_eps);
    else
        hipLaunchKernelGGL((norm_gate_add_stats_kernel<16, true>), grid, block, 0, st, out, out_mean, out_rstd, x, nullptr, nullptr, gate, shortcut, sc_mean, sc_rstd, ss, prelu, C, h, w, eps, in_eps);
    return check_launch("norm_self_gate_add_stats");
}

// ------------------------------------------------------------------------------------ masked average pooling per region
// out[b, r, c] = mean over {p : label(p) == r} of feats[b, c, p]  (0 when the region is empty); labels sampled 'nearest'.
// One block per (b, four channels): the label tile is staged once, then every wave pools ONE plane in one pass — a lane keeps a sum per region in registers and
// adds every element to all of them under a select; the region's pixel count comes from wave ballots (scalar), the sums from one wave reduction each: 12
// reductions per plane and a single barrier.  (Until round 4: one block per plane, one pass per region with two block reductions each — 96 barriers and
// 98 us per launch whatever the map size.)
__global__ __launch_bounds__(256) void masked_avg_pool_kernel(float* __restrict__ out, const float* __restrict__ feats,
                                                              const uint8_t* __restrict__ labels, int lh, int lw, float lsy, float lsx, int C,
                                                              int h, int w, int nreg) {
    extern __shared__ uint8_t lab[];  // [h*w] labels at feature resolution
    const int b = blockIdx.y;
    const int hw = h * w;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const int y = i / w, x = i - y * w;
        lab[i] = labels[((size_t)b * lh + nearest_src(y, lsy, lh)) * lw + nearest_src(x, lsx, lw)];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    const float* fp = feats + ((size_t)b * C + c) * hw;
    float s[E4S_MAX_REGIONS];
    int n[E4S_MAX_REGIONS];            // wave-uniform
#pragma unroll
    for (int r = 0; r < E4S_MAX_REGIONS; ++r) { s[r] = 0.f; n[r] = 0; }
    for (int i0 = 0; i0 < hw; i0 += 64) {
        const int i = i0 + lane;
        const float v = i < hw ? fp[i] : 0.f;
        const int l = i < hw ? lab[i] : 255;
#pragma unroll
        for (int r = 0; r < E4S_MAX_REGIONS; ++r) {
            s[r] += l == r ? v : 0.f;
            n[r] += __popcll(__ballot(l == r));
        }
    }
#pragma unroll
    for (int r = 0; r < E4S_MAX_REGIONS; ++r) {
        const float ts = wave_sum(s[r]);
        if (lane == 0 && r < nreg) out[((size_t)b * nreg + r) * C + c] = n[r] > 0 ? ts / (float)n[r] : 0.f;
    }
}

extern "C" int e4s_masked_avg_pool(float* out, const float* feats, const uint8_t* labels, int lh, int lw, int bs, int C, int h, int w, int nreg,
                                   void* stream) {
    E4S_REQUIRE(out && feats && labels, "masked_avg_pool: null tensor");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && C >= 1 && h >= 1 && w >= 1 && lh >= 1 && lw >= 1, "masked_avg_pool: bad size");
    E4S_REQUIRE(nreg >= 1 && nreg <= E4S_MAX_REGIONS, "masked_avg_pool: %d regions (max %d)", nreg, E4S_MAX_REGIONS);
    E4S_REQUIRE(h * w <= 48 * 1024, "masked_avg_pool: feature map %dx%d too large for the LDS label tile", h, w);
    if (bs == 0) return 0;
    hipLaunchKernelGGL(masked_avg_pool_kernel, dim3(cdiv(C, 4), bs), dim3(256), (size_t)h * w, (hipStream_t)stream, out, feats, labels, lh, lw,
                       (float)lh / (float)h, (float)lw / (float)w, C, h, w, nreg);
    return check_launch("masked_avg_pool");
}

// ------------------------------------------------------------------------------------ bilinear resize
// F.interpolate(mode='bilinear') for both align_corners settings (no antialias), planes = bs*C.
//   align_corners=False: src = max(0, (dst + 0.5) * in/out - 0.5)      align_corners=True: src = dst * (in-1)/(out-1)
__device__ __forceinline__ void bilinear_coord(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
    float src = align ? (float)dst * scale : ((float)dst + 0.5f) * scale - 0.5f;
    if (!align && src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear_kernel(float* __restrict__ out, const float* __restrict__ in, int ih, int iw, int oh, int ow,
                                                       float sy, float sx, int align) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_coord(y, sy, align, ih, y0, y1, ly);
    bilinear_coord(x, sx, align, iw, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p = in + (size_t)blockIdx.z * ih * iw;
    // ATen upsample_bilinear2d: w00*v00 + w01*v01 + w10*v10 + w11*v11 grouped per row
    const float v = hy * (hx * p[(size_t)y0 * iw + x0] + lx * p[(size_t)y0 * iw + x1]) + ly * (hx * p[(size_t)y1 * iw + x0] + lx * p[(size_t)y1 * iw + x1]);
    out[((size_t)blockIdx.z * oh + y) * ow + x] = v;
}

extern "C" int e4s_bilinear_resize(float* out, const float* in, int planes, int ih, int iw, int oh, int ow, int align_corners, void* stream) {
    E4S_REQUIRE(out && in, "bilinear_resize: null tensor");
    E4S_REQUIRE(planes >= 0 && planes <= 65535 && ih >= 1 && iw >= 1 && oh >= 1 && ow >= 1, "bilinear_resize: bad size");
    if (planes == 0) return 0;
    float sy, sx;
    if (align_corners) {
        sy = oh > 1 ? (float)(ih - 1) / (float)(oh - 1) : 0.f;
        sx = ow > 1 ? (float)(iw - 1) / (float)(ow - 1) : 0.f;
    } else {
        sy = (float)ih / (float)oh;
        sx = (float)iw / (float)ow;
    }
    hipLaunchKernelGGL(bilinear_kernel, dim3(cdiv(ow, 64), cdiv(oh, 4), planes), dim3(256), 0, (hipStream_t)stream, out, in, ih, iw, oh, ow, sy, sx,
                       align_corners ? 1 : 0);
    return check_launch("bilinear_resize");
}
