/* libe4s_hip.so — C ABI of the MI355X (gfx950) kernels behind the E4S regional-GAN-inversion
 * hot path.  Plain pointers and sizes only: every pointer is a DEVICE pointer to fp32 data
 * (unless the name says otherwise), every tensor is dense NCHW, the caller owns all buffers
 * and passes the HIP stream to launch on (`stream` = hipStream_t, 0 = default stream).
 * Every entry point returns 0 on success, E4S_ERR_ARG (-1) for a rejected argument, or the
 * positive hipError_t of a failed launch; e4s_last_error() returns the message (thread-local).
 * Nothing here synchronises the device, allocates device memory or keeps global state.
 *
 * Each entry point names the reference interface (paths relative to the reference tree) it
 * replaces; INTEGRATION.md shows the Python/ctypes binding a maintainer would add there.
 */
#ifndef E4S_HIP_H
#define E4S_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E4S_API __attribute__((visibility("default")))
#define E4S_ABI_VERSION 1
#define E4S_ERR_ARG (-1)
#define E4S_MAX_REGIONS 16 /* regions (segmentation classes) a masked layer can mix; the reference uses 12 */
#define E4S_LABEL_NONE 255 /* label of a pixel whose one-hot column is all zero (output of the masked sum is 0) */

E4S_API int e4s_abi_version(void);
E4S_API const char* e4s_last_error(void);

/* ------------------------------------------------------------------------------------ a1
 * Replaces `fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)` —
 * models/stylegan2/op/fused_bias_act.cpp:11-21, kernel fused_bias_act_kernel.cu:18-49, launcher :52-99.
 *   out[i] = f(x[i] + bias[(i / step_b) % size_b]) * scale,  selector act*10+grad:
 *   10/11 linear, 12 zero, 30 leaky-relu, 31 leaky-relu backward (slope chosen by sign of ref[i]), 32 zero.
 * bias may be NULL (size_b = 0), ref may be NULL (treated as 0). In-place (out == x) is allowed. */
E4S_API int e4s_fused_bias_act(float* out, const float* x, const float* bias, const float* ref,
                       int act, int grad, float alpha, float scale,
                       int64_t size_x, int64_t step_b, int64_t size_b, void* stream);

/* ------------------------------------------------------------------------------------ a2
 * Replaces `upfirdn2d_op.upfirdn2d(input[major,H,W,minor], kernel[kh,kw], up_x, up_y, down_x, down_y,
 * pad_x0, pad_x1, pad_y0, pad_y1)` — models/stylegan2/op/upfirdn2d.cpp:12-23, kernel
 * upfirdn2d_kernel.cu:52-137, launcher :140-272.  minor must be 1 (the only layout the reference's
 * Python wrapper produces: op/upfirdn2d.py:96).  out is [major, out_h, out_w] with
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1 (op/upfirdn2d.py:100-101); kh,kw <= 32. */
E4S_API int e4s_upfirdn2d(float* out, const float* in, const float* kernel,
                  int major, int in_h, int in_w, int kh, int kw,
                  int up_x, int up_y, int down_x, int down_y,
                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* --------------------------------------------------------------------------- a3 - a6
 * Region-aware modulated synthesis.  The reference evaluates a masked layer as
 *   out = sum_c ModulatedConv2d(x, style[:, c]) * nearest(mask)[:, c]      (models/stylegan2/model.py:385-400, 442-456)
 * i.e. 12 full convolutions.  With one-hot masks this equals, per output pixel p of class c(p),
 *   out[b,o,p] = d[b,c(p),o] * sum_{i,k} (W[o,i,k]/sqrt(Cin k^2)) * s[b,c(p),i] * x[b,i,p+k]
 * which the kernels below evaluate ONCE per layer.  */

/* One-hot mask [bs, ncls, h, w] (values 0/1, utils/torch_utils.py:207-213) -> uint8 labels [bs, h, w].
 * A pixel with no class set gets E4S_LABEL_NONE.  *flag (device int, caller zeroes it) is OR-ed with
 * 1 if any value is neither 0 nor 1, with 2 if a pixel has more than one class set. */
E4S_API int e4s_onehot_to_labels(uint8_t* labels, int* flag, const float* mask, int bs, int ncls, int h, int w, void* stream);

/* Weight preparation, once per parameter version (replaces the per-call weight materialisation of
 * models/stylegan2/model.py:277-294).
 *   weight : [cout, cin, k, k]  (the ModulatedConv2d parameter without its leading 1), k = 3 or 1
 *   blur   : [4,4] FIR of the layer's Blur (model.py:212-213) when up != 0, else NULL
 *   wt     : out, [npar, cin, k*k, cout] = weight / sqrt(cin k^2), K-major; npar = 4 when up: the stride-2
 *            transposed 3x3 conv (model.py:295-297) composed with the blur (upfirdn2d pad (1,1), model.py:300)
 *            is, for each output parity (y&1, x&1), a 3x3 correlation over the INPUT grid; parity index = 2*(y&1)+(x&1)
 *   wsq    : out, [cin, cout] = sum_k (weight/sqrt(cin k^2))^2 for the demodulation table (may be NULL) */
E4S_API int e4s_modconv_prep_weights(float* wt, float* wsq, const float* weight, const float* blur,
                             int cout, int cin, int k, int up, void* stream);

/* Style and demodulation tables of one layer (model.py:276-281 for every (sample, region) at once).
 *   styles : W+ codes for this layer, element (b, r, j) at styles[b*stride_b + r*stride_r + j], j < sdim
 *   s      : out, [bs, nreg, cin]  = styles @ (mod_weight/sqrt(sdim))^T + mod_bias      (EqualLinear, model.py:154-162)
 *   d      : out, [bs, nreg, cout] = rsqrt(sum_i s^2 * wsq[i, o] + 1e-8); pass d = NULL / wsq = NULL for demodulate=False */
E4S_API int e4s_style_demod(float* s, float* d, const float* styles, int64_t stride_b, int64_t stride_r,
                    const float* mod_weight, const float* mod_bias, const float* wsq,
                    int bs, int nreg, int cin, int cout, int sdim, void* stream);

/* The same for up to E4S_MAX_STYLE_JOBS layers in two launches (all 26 modulated convs of a 1024x1024 generator: their W+ codes
 * are known before the first layer runs).  jobs is a HOST array; every pointer inside is a device pointer with the meaning of the
 * e4s_style_demod argument of the same name (d/wsq NULL for demodulate=False). */
#define E4S_MAX_STYLE_JOBS 32
typedef struct E4sStyleJob {
    float* s;
    float* d;
    const float* styles;
    int64_t stride_b, stride_r;
    const float* mod_weight;
    const float* mod_bias;
    const float* wsq;
    int nreg, cin, cout, _pad;
} E4sStyleJob;
E4S_API int e4s_style_demod_batched(const E4sStyleJob* jobs, int n_jobs, int bs, int sdim, void* stream);

/* StyledConv forward in one pass (model.py:382-423): 3x3 modulated conv (same resolution, or x2 up-conv
 * + blur when up != 0) with per-pixel region modulation, demodulation, noise injection, bias, leaky-relu*sqrt2.
 *   x       : [bs, cin, h, w]            out : [bs, cout, ho, wo]  (ho = h, or 2h when up)
 *   wt, s, d: from the two calls above (d may be NULL: no demodulation)
 *   labels  : uint8 [bs, lh, lw] region map, sampled nearest at the OUTPUT pixel (model.py:389-391); NULL = every pixel is
 *             region 0 (unmasked layer, nreg must be 1)
 *   noise   : [noise_bs (1 or bs), 1, ho, wo] or NULL;  noise_weight: device pointer to the NoiseInjection scalar (model.py:335)
 *   act_bias: [cout] FusedLeakyReLU bias or NULL; act != 0 applies leaky_relu(0.2)*sqrt(2) (model.py:421)
 *   workspace: optional scratch of workspace_floats floats (NULL/0 = none).  Feature maps too small to fill 256 CUs (4x4..32x32)
 *             are then reduced split-K over input channels: K-slices write raw partial sums to the workspace and a second
 *             launch sums them in a fixed order and applies the epilogue (deterministic; 16*bs*cout*ho*wo floats is always enough) */
E4S_API int e4s_region_modconv3x3(float* out, const float* x, const float* wt, const float* s, const float* d,
                          const uint8_t* labels, int lh, int lw,
                          const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act,
                          int bs, int cin, int cout, int h, int w, int nreg, int up,
                          float* workspace, int64_t workspace_floats, void* stream);

/* Split-bf16 variant of the two calls above (the default fast path; numerics in DESIGN.md §4): every fp32 operand is split into
 * bf16 hi + bf16 lo and a*b is evaluated as hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation — 3 MFMAs at 16x the fp32
 * MFMA rate.  whi / wlo : out, bf16 (as uint16) [npar][ceil(cin/16)][9][2][cout][8]: element (par, chunk, tap, half, co, e) holds input
 * channel chunk*16 + half*8 + e (zero beyond cin).  k = 3 only.  All other arguments as e4s_modconv_prep_weights / e4s_region_modconv3x3. */
E4S_API int e4s_modconv_prep_weights_sb(uint16_t* whi, uint16_t* wlo, float* wsq, const float* weight, const float* blur,
                                        int cout, int cin, int up, void* stream);
/* Layout flags OR-ed into the `up` argument of e4s_region_modconv3x3_sb (and the `act` argument of e4s_modconv_up_fused_sb): the activation
 * is channel-blocked, [bs, c/8, h, w, 8] instead of [bs, c, h, w] — a pixel's 8 channels are 32 contiguous bytes and consecutive pixels
 * follow, so a tile's halo columns share their cache lines with all 8 channels and both producer and consumer move contiguous bytes.
 * cin % 16 == 0 / cout % 8 == 0, w >= 32, single-region layers (input) / any layer (output). */
#define E4S_X_NHWC 2
#define E4S_OUT_NHWC 4
/* ... or a split-plane tensor (see "The single-region chain" below): E4S_X_SP (e4s_modconv_up_fused_sb) = x is split planes already carrying
 * this layer's modulation (s is then not applied again); E4S_OUT_SP = out is written as split planes modulated by s_next[bs][cout]. */
#define E4S_X_SP 8
#define E4S_OUT_SP 16
E4S_API int e4s_region_modconv3x3_sb(float* out, const float* x, const uint16_t* whi, const uint16_t* wlo, const float* s, const float* d,
                                     const uint8_t* labels, int lh, int lw,
                                     const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act,
                                     int bs, int cin, int cout, int h, int w, int nreg, int up,
                                     float* workspace, int64_t workspace_floats,
                                     float* rgb_out, const float* rgb_wt, const float* rgb_s, const float* rgb_bias,
                                     const float* rgb_skip, const float* rgb_up_kernel, const float* s_next,
                                     const uint8_t* uniform_blocks, const int* uniform_ctrl, void* stream);

/* Round 3: the same masked layer on a DMA-fed kernel (csrc/modconv_mx.hip; masked layers of width >= 32, cout >= 128, cin % 16 == 0, channels-first
 * activations): the weights arrive as ready-to-DMA row slots (one kernel row of a 16-channel chunk for 128 output channels: 24-25 KB) from
 * e4s_modconv_prep_weights_mx.  arith 0 = the split-bf16 arithmetic above, bit-identical results; arith 1 = a1*w1 on the f16 MFMA plus the two cross
 * terms fp6(a)*fp6(w - w1) and fp6(a - a1)*fp6(w1) on the block-scaled MX fp6 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4): about half the matrix-pipe
 * time, 2-3x the split-bf16 error (1.9e-4 max-abs on the 1024^2 generator against 8e-5; bar 1e-3).  flags (optional, arith 1; int[2]): a wave that sees a
 * modulated activation >= 65520 (what f16 rounds to infinity) does flags[0] |= 1 and flags[1] += 1 — the result is then not to be trusted.  The library does not
 * fall back by itself (no host synchronisation in the ABI): the host layer snapshots flags[1] before and after a forward pass and re-runs a pass that moved
 * it with arith 0 (e4s2024_amd/ops.py MxGuard; the reference computes these layers in fp32, models/stylegan2/model.py:276-320).
 * e4s_modconv_mx_weight_bytes: size of the prepared copy.  All other arguments as e4s_region_modconv3x3_sb. */
E4S_API int e4s_modconv_mx_weight_bytes(int cout, int cin, int up, int arith, int64_t* bytes);
E4S_API int e4s_modconv_prep_weights_mx(void* dst, const float* weight, const float* blur, int cout, int cin, int up, int arith, void* stream);
E4S_API int e4s_region_modconv3x3_mx(float* out, const float* x, const void* wmx, int arith, int* flags, const float* s, const float* d,
                                     const uint8_t* labels, int lh, int lw,
                                     const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act,
                                     int bs, int cin, int cout, int h, int w, int nreg, int up,
                                     float* workspace, int64_t workspace_floats,
                                     float* rgb_out, const float* rgb_wt, const float* rgb_s, const float* rgb_bias, const float* rgb_skip,
                                     const float* rgb_up_kernel, const float* s_next, const uint8_t* uniform_blocks, const int* uniform_ctrl,
                                     void* stream);
/* Round 4: the masked up layer (models/stylegan2/model.py:287-300 with the per-region mixing of :385-400) with the four output parities of a position in ONE workgroup
 * (csrc/modconv_mx4.hip): where the 2 x 2 outputs of every position of a 32 x 8-position tile share a region, the modulated / split / fp6-converted activation operand
 * is prepared once for the four composed 3x3 kernels; the other tiles are computed, inside the same launch, exactly as e4s_region_modconv3x3_mx (arith 1) computes them.
 * The result is bit-identical to that call on every map.  f16 + 2 x MX fp6 arithmetic; wmx4 from e4s_modconv_prep_weights_mx4 (weight [1,cout,cin,3,3], blur [4,4];
 * size e4s_modconv_mx4_weight_bytes), wmx from e4s_modconv_prep_weights_mx(up = 1, arith = 1); cin % 16 == 0, cout % 128 == 0, w >= 32, out 16-byte aligned, noise
 * 8-byte aligned; flags and all other arguments as for e4s_region_modconv3x3_mx. */
E4S_API int e4s_modconv_mx4_weight_bytes(int cout, int cin, int64_t* bytes);
E4S_API int e4s_modconv_prep_weights_mx4(void* dst, const float* weight, const float* blur, int cout, int cin, void* stream);
E4S_API int e4s_region_upconv_mx4(float* out, const float* x, const void* wmx4, const void* wmx, int* flags, const float* s, const float* d, const uint8_t* labels,
                                  int lh, int lw, const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act, int bs, int cin,
                                  int cout, int h, int w, int nreg, void* stream);
/* Round 5: the region-uniform 16 x 16 output blocks of a masked up layer (models/stylegan2/model.py:287-300, 385-400; the map: e4s_uniform_blocks) on the f16 + 2 x MX-fp6
 * arithmetic with operands prepared once at staging (csrc/modconv_upblock_mx.hip): x * s[region] -> f16 + two fp6 terms per patch pixel and 32-channel chunk, weights as
 * tap-pair units by LDS-DMA, the same 1x transposed-conv form and blur epilogue.  wmx from e4s_modconv_prep_weights_upblock_mx (weight [cout,cin,3,3] or [1,cout,cin,3,3],
 * NOT blur-composed; size e4s_upblock_mx_weight_bytes); cin % 32 == 0, cin <= 512; blocks / ctrl from e4s_uniform_blocks (want_quad = 0); flags as for
 * e4s_region_modconv3x3_mx; s [bs][nreg][cin], d [bs][nreg][cout], blur [4][4], out fp32 [bs][cout][2h][2w] (only the blocks < nreg are written); w % 16 == 0. */
E4S_API int e4s_upblock_mx_weight_bytes(int cout, int cin, int64_t* bytes);
E4S_API int e4s_modconv_prep_weights_upblock_mx(void* dst, const float* weight, int cout, int cin, void* stream);
E4S_API int e4s_masked_upconv_blocks_mx(float* out, const float* x, const void* wmx, int* flags, const float* s, const float* d, const uint8_t* blocks, const int* ctrl,
                                        const float* blur, const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act, int bs, int cin,
                                        int cout, int h, int w, int nreg, void* stream);
/* The regional-style encoder's stride-1, pad-1 3x3 convolutions (models/encoders/helpers.py:128-139) on the same kernel in its plain-convolution mode:
 *   out[bs,cout,h,w] = PReLU( conv3x3( (x - in_mean[b,ci]) * in_rstd[b,ci], W ) )          in_mean / in_rstd (together) and prelu_slope optional
 * cin % 16 == 0; padding is exactly 0 (the normalisation applies to in-image pixels only).  wmx from e4s_conv_prep_weights_mx (weight [cout,cin,3,3],
 * no scale; size: e4s_modconv_mx_weight_bytes(cout, cin, 0, arith)).  arith and flags as above. */
E4S_API int e4s_conv_prep_weights_mx(void* dst, const float* weight, int cout, int cin, int arith, void* stream);
E4S_API int e4s_conv3x3_mx(float* out, const float* x, const void* wmx, int arith, int* flags, const float* in_mean, const float* in_rstd,
                           const float* prelu_slope, int bs, int cin, int cout, int h, int w, void* stream);

/* The same operator (f16 + 2 x MX fp6 only) on the two-phase kernel of csrc/conv_mx3.hip: 32-channel chunks (cin % 32 == 0, cin <= 512), activations converted
 * to fp6 once per staged value, waves 4-7 half a unit behind waves 0-3 so that every SIMD always has a wave on the matrix pipe.  Replaces the same reference
 * statements (models/encoders/helpers.py:128-139).  wmx3 from e4s_conv_prep_weights_mx3 (weight [cout,cin,3,3]; size: e4s_conv3x3_mx3_weight_bytes);
 * flags as for e4s_region_modconv3x3_mx (flags[0] |= 1, flags[1] += 1 when a normalised activation leaves the f16 range). */
E4S_API int e4s_conv3x3_mx3_weight_bytes(int cout, int cin, int64_t* bytes);
E4S_API int e4s_conv_prep_weights_mx3(void* dst, const float* weight, int cout, int cin, void* stream);
E4S_API int e4s_conv3x3_mx3(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                            int bs, int cin, int cout, int h, int w, void* stream);
/* The stride-2 form on the same kernel: out [bs,cout,h/2,w/2] = PReLU(conv3x3(norm(x), W, stride 2, pad 1)), h and w even — the second convolution of a stage's
 * first bottleneck_IR_SE_Ours unit (models/encoders/helpers.py:128-139 with stride = 2, psp_encoders.py get_blocks).  The input is read as its four phase planes
 * (x[2y+py][2x+px]), each a stride-1 operand of a subset of the nine taps; wmx3 from e4s_conv_prep_weights_mx3_s2 (same size as the stride-1 copy). */
E4S_API int e4s_conv_prep_weights_mx3_s2(void* dst, const float* weight, int cout, int cin, void* stream);
E4S_API int e4s_conv3x3_s2_mx3(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                               int bs, int cin, int cout, int h, int w, int in_phased, void* stream);
/* e4s_conv3x3_mx3 with its result stored as PHASE PLANES: out[b][c][2 py + px][h/2][w/2] = result[b][c][2y+py][2x+px] (h, w even) — the hand-over between the two
 * convolutions of a stride-2 unit (helpers.py:128-139): e4s_conv3x3_s2_mx3(in_phased = 1) then reads consecutive floats with consecutive lanes, where the plain
 * map costs it two cache lines per useful one. */
E4S_API int e4s_conv3x3_mx3_phased(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                                   int bs, int cin, int cout, int h, int w, void* stream);
/* e4s_conv3x3_mx3 with explicit memory layouts (round 5): a layout word's bit 0 (1) = phase planes as above, bit 1 (2) = CHANNEL-BLOCKED — the map is
 * [bs][c/4][plane layout][4 floats], a pixel's four channels one 16-byte element (c % 4 == 0, 16-byte aligned).  It is the hand-over between the two convolutions
 * of a bottleneck_IR_SE_Ours unit (models/encoders/helpers.py:128-139: Conv2d -> PReLU -> Conv2d with nothing between them and no other reader): the producer
 * stores 16 bytes per request, the consumer's patch threads request 8 elements per 32-channel chunk instead of 32 floats.
 * Bit 2 (4) = PREPARED OPERANDS (4, or 5 = in phase-plane pixel order; not combined with bit 1): the map is stored as what the consuming convolution's staging would
 * compute from it — per image and 32-channel block 116 h w bytes: f16 part [slot 4][pixel] x 16 B (slot s = channels 8s..8s+7) | MX-fp6 codes of the f16 part and of the
 * residual, first 16 B [term 2][pixel] | their last 8 B each [pixel] x 16 B | the two E8M0 block scales [pixel] x 4 B (bytes 0, 1) — made once per pixel in the
 * producer's epilogue (cout % 32 == 0, h w % 4 == 0; the producer raises the f16 flag for an output beyond the f16 range); as an input it takes no in_mean / in_rstd.
 * in_layout: 0, 2 or 4; out_layout: 0..5; e4s_conv3x3_s2_mx3's in_phased argument is such a word too (0..5).  Values do not depend on the layouts (bit for bit those
 * of e4s_conv3x3_mx3). */
E4S_API int e4s_conv3x3_mx3_ex(float* out, const float* x, const void* wmx3, int* flags, const float* in_mean, const float* in_rstd, const float* prelu_slope,
                               int bs, int cin, int cout, int h, int w, int in_layout, int out_layout, void* stream);
/* Masked up-sampling layers, region-uniform output blocks (model.py:287-300 per region == one transposed conv + blur where a block of output
 * pixels has ONE region):
 *   e4s_uniform_blocks: sub[b][2by+sy][2bx+sx] = the region of an 8 x 8 output sub-block (labels uint8 [bs][lh][lw] sampled 'nearest' at
 *   ho x wo), 255 = mixed / no region; blocks[b][by][bx] = the region of a 16 x 16 block if its four sub-blocks share one, 254 if each of
 *   them is uniform but they differ (only with want_quad), 255 otherwise — and 255 for a
 *   whole row of four blocks (one tile of the composed kernel) unless all four qualify; ctrl: four ints, [0] and [1] zero on entry (every launch leaves them zero again: one buffer per stream serves all layers), ctrl[2] becomes 1
 *   if at least min_percent of those rows qualify, else 0 (both consumers then leave the layer in the composed form);
 *   e4s_masked_upconv_blocks_mx (above) computes exactly the blocks < nreg in the transposed-conv form; e4s_region_modconv3x3_mx / _upconv_mx4(..., uniform_blocks = the
 *   same block map) compute the remaining blocks in the composed form.  (want_quad = 1 additionally marks blocks of four uniform 8 x 8 sub-blocks with 254: the round-2
 *   sub-block kernel that consumed them was measured slower than the composed form and deleted in round 5 — pass 0.) */
E4S_API int e4s_uniform_blocks(uint8_t* blocks, uint8_t* sub, int* ctrl, const uint8_t* labels, int bs, int lh, int lw, int ho, int wo, int nreg,
                               int want_quad, int min_percent, void* stream);
/* rgb_* (all NULL = off): fuse the single-region ToRGB that follows this layer (model.py:439-479) into the epilogue — allowed for
 * same-resolution layers of width >= 32 whose Cout fits one workgroup tile (<= 64, or <= 128 on masked layers): rgb_out [bs,3,h,w] =
 * sum_co out[co] * rgb_wt[co][o] * rgb_s[b][co] + rgb_bias[o] + upfirdn2d(rgb_skip, rgb_up_kernel, up=2, pad=(2,1)), so the layer's
 * output is not read back for the 1x1 conv.  rgb_wt from e4s_modconv_prep_weights(k=1), rgb_s = the ToRGB's s table [bs,1,cout].
 * With rgb_out given, out may be NULL: the layer's own activation is then not written at all (the last layer of the generator, whose
 * output only feeds its ToRGB). */

/* Single-region (unmasked) up layer at 1x the transposed conv's MACs, in two launches (the parity-composed kernel above spends 4x):
 *   e4s_modconv_tconv_sb : z[bs,cout,2h+1,2w+1] = conv_transpose2d(x * s, W/sqrt(9 cin), stride 2)   (model.py:287-299; raw sums)
 *                          whi/wlo from e4s_modconv_prep_weights_sb(up = 0) on the layer's 3x3 weight; s [bs,1,cin]
 *   e4s_blur_epilogue    : out[bs,cout,ho,wo] = act( d * upfirdn2d(z, blur 4x4, pad (1,1)) + noise_weight*noise + act_bias )
 *                          (model.py:300 + 419-421); d [bs,1,cout] or NULL; ho = 2h, wo = 2w */
E4S_API int e4s_modconv_tconv_sb(float* z, const float* x, const uint16_t* whi, const uint16_t* wlo, const float* s,
                                 int bs, int cin, int cout, int h, int w, void* stream);
E4S_API int e4s_blur_epilogue(float* out, const float* z, const float* blur, const float* d,
                              const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act,
                              int bs, int cout, int ho, int wo, void* stream);

/* The same single-region up layer in ONE launch: the transposed conv's pre-blur tile stays in LDS and the 4x4 blur, demodulation,
 * noise, bias and activation are applied before the only write (no [bs,cout,2h+1,2w+1] round trip).  Arguments as the pair above. */
E4S_API int e4s_modconv_up_fused_sb(float* out, const float* x, const uint16_t* whi, const uint16_t* wlo, const float* s, const float* d,
                                    const float* blur, const float* noise, int noise_bs, const float* noise_weight,
                                    const float* act_bias, int act, int bs, int cin, int cout, int h, int w, const float* s_next, void* stream);

/* ---- The single-region chain (layers past remaining_layer_idx) on split planes.
 * A single-region layer's modulation s[b][ci] belongs to the INPUT channel alone (model.py:276-283 with one style per sample) and all
 * tables are known up front, so the producer of an activation applies the CONSUMER's modulation and the bf16 hi/lo split in its epilogue
 * and writes "split planes":  sp[plane hi|lo][bs][c/8][h][w][8 x bf16]  (c % 16 == 0; same bytes as the fp32 tensor), whose 16-byte
 * element is one lane's MFMA B fragment.  A split-plane buffer carries 16 more bytes behind the second plane, written as zeros by whoever
 * produces it (the padding source of the consumers' LDS-DMA): allocate 2 * bs * c * h * w * 2 + 16 bytes.  Consumers stage them with LDS-DMA only (no registers, no VALU) from persistent workgroups.
 * Arithmetic is that of e4s_region_modconv3x3_sb / e4s_modconv_up_fused_sb on the fp32 tensor (fl(x * s), RNE split).
 *   e4s_to_split_planes : fp32 [bs,c,h,w] (or channel-blocked [bs,c/8,h,w,8] when x_nhwc) times s[bs][c] -> split planes
 *   e4s_chain_conv3x3   : StyledConv (same resolution) reading split planes; out_sp (optional) = its activation as split planes modulated
 *                         by s_next[bs][cout]; rgb_* (optional) = the following single-region ToRGB fused as in e4s_region_modconv3x3_sb.
 *                         h % 16 == 0, w % 32 == 0; built for 32 -> 32 and 64 -> 64 channels (the 1024 / 512 stages of Generator(1024)).
 *   (the chain's up-sampling layers: e4s_modconv_up_hc below; a rank-1 blur kernel is required there, anything else takes e4s_modconv_up_fused_sb)
 * L is a HOST struct; every pointer inside is a device pointer.  whi / wlo from e4s_modconv_prep_weights_sb(up = 0). */
typedef struct E4sChainLayer {
    const uint16_t* x_sp;
    const uint16_t* whi;
    const uint16_t* wlo;
    const float* d;            /* [bs][cout] demodulation or NULL */
    const float* noise;        /* [noise_bs (1 or bs)][ho*wo] or NULL */
    const float* noise_weight;
    const float* act_bias;     /* [cout] or NULL */
    uint16_t* out_sp;          /* or NULL */
    const float* s_next;       /* [bs][cout], required with out_sp */
    float* rgb_out;            /* [bs,3,h,w] or NULL (e4s_chain_conv3x3 only) */
    const float* rgb_wt;       /* [cout][3]  (e4s_modconv_prep_weights, k = 1) */
    const float* rgb_s;        /* [bs][cout] */
    const float* rgb_bias;     /* [3] */
    const float* rgb_skip;     /* [bs,3,h/2,w/2] or NULL */
    const float* rgb_up_kernel;/* [4,4] */
    int noise_bs, act, bs, cin, cout, h, w, _pad;
} E4sChainLayer;
E4S_API int e4s_to_split_planes(uint16_t* out_sp, const float* x, const float* s, int bs, int c, int h, int w, int x_nhwc, void* stream);
E4S_API int e4s_chain_conv3x3(const E4sChainLayer* L, void* stream);

/* The chain's up layer in the HALF-COMPOSED form (csrc/modconv_uphc.hip; model.py:287-301 + 417-421, single-region case).  The 4 x 4 blur of the
 * reference is an outer product kv x kh (model.py:23-31 make_kernel of a 1-D list): its VERTICAL factor is composed into the weights — two output-row
 * parities x 9 taps over the input rows m-1, m, m+1 (2x the MACs of the bare transposed conv, no vertical tile overlap) — and its HORIZONTAL factor is
 * applied to the MFMA accumulators in registers (three DPP row shifts per register), so the epilogue needs no LDS round trip and no barrier; outputs leave
 * as 16-byte stores of split planes.
 *   e4s_modconv_prep_weights_hc : whi / wlo, bf16 (as uint16) [2 row parities][cin/16][9 = (dy+1)*3 + kx][2][cout][8] from the layer's 3x3 weight
 *                                 [1,cout,cin,3,3] (equalised-lr scale folded in) and the blur kernel [4,4]
 *   e4s_modconv_up_hc           : x_sp [2][bs][cin/8][h][w][8] split planes carrying this layer's modulation -> out_sp [2][bs][cout/8][2h][2w][8]
 *                                 (+ 16 zero bytes) modulated by s_next[bs][cout]; d [bs][cout]; cin % 16 == 0, cout % 32 == 0.
 * `blur` MUST be rank 1 (blur[r][c] == rowsum[r] * colsum[c] / sum): the caller checks that on the host (e4s2024_amd/ops.py PreparedHc) and keeps
 * e4s_modconv_up_fused_sb for any other kernel. */
E4S_API int e4s_modconv_prep_weights_hc(uint16_t* whi, uint16_t* wlo, const float* weight, const float* blur, int cout, int cin, void* stream);
E4S_API int e4s_modconv_up_hc(uint16_t* out_sp, const uint16_t* x_sp, const uint16_t* whi, const uint16_t* wlo, const float* d, const float* blur,
                              const float* noise, int noise_bs, const float* noise_weight, const float* act_bias, int act,
                              int bs, int cin, int cout, int h, int w, const float* s_next, void* stream);

/* ToRGB forward in one pass (model.py:439-479): 1x1 modulated conv without demodulation, + bias, + upsampled skip.
 *   x : [bs, cin, h, w]   wt : [cin, 3] from e4s_modconv_prep_weights(k=1)   s : [bs, nreg, cin]   bias : [3]
 *   skip : previous RGB [bs, 3, h/2, w/2] or NULL; up_kernel : [4,4] FIR of Upsample (model.py:34-53; up=2, pad=(2,1))
 *   out : [bs, 3, h, w] */
E4S_API int e4s_region_torgb(float* out, const float* x, const float* wt, const float* s,
                     const uint8_t* labels, int lh, int lw, const float* bias,
                     const float* skip, const float* up_kernel,
                     int bs, int cin, int h, int w, int nreg, void* stream);

/* ------------------------------------------------------------------------------------ a7
 * Grouped equalised linear: for g < groups, out[b,g,:] = act(x[b,g,:] @ (W[g]*scale)^T + bias[g]*bias_mul) (+ addend)
 * Replaces the 12 LocalMLP layers of models/networks.py:32-36, 226-230 (groups = 12) and EqualLinear
 * (models/stylegan2/model.py:154-164, groups = 1).
 *   x   : element (b,g,i) at x[b*x_stride_b + g*x_stride_g + i], i < in_dim
 *   W   : HOST array of `groups` (<= 16) device pointers, W[g] -> [out_dim, in_dim] (the 12 MLPs are separate parameters);
 *   bias: HOST array of `groups` device pointers to [out_dim] (entries may be NULL), or NULL
 *   act : 0 none, 1 leaky_relu(slope), 2 leaky_relu(slope) * sqrt(2) (fused_lrelu)
 *   addend : [out_dim] added to every (b,g) row, or NULL (latent_avg, networks.py:247)
 *   out : element (b,g,o) at out[b*out_stride_b + g*out_stride_g + o] */
E4S_API int e4s_grouped_linear(float* out, int64_t out_stride_b, int64_t out_stride_g,
                       const float* x, int64_t x_stride_b, int64_t x_stride_g,
                       const float* const* W, const float* const* bias, const float* addend,
                       float scale, float bias_mul, int act, float slope,
                       int bs, int groups, int in_dim, int out_dim, void* stream);

/* Backward of e4s_grouped_linear for dense [bs][groups][...] tensors (f1: PTI trains the LocalMLPs, training/video_swap_ft_coach.py:297-299).
 * gy = dL/d(pre-activation) [bs][groups][out_dim].  dW [groups][out_dim][in_dim] = scale * sum_b gy x^T and db [groups][out_dim] =
 * bias_mul * sum_b gy (either may be NULL);  dx [bs][groups][in_dim] = scale * W^T gy, times leaky_relu'(h_prev) when h_prev (the previous
 * layer's OUTPUT, same shape as dx) is given — or NULL for no input gradient.  scratch: osplit * bs * groups * in_dim floats (the sum over
 * out_dim is split over osplit workgroups and finished in a fixed order).  bs <= 8, in_dim % 4 == 0. */
E4S_API int e4s_grouped_linear_bwd(float* dW, float* db, float* dx, float* scratch, const float* gy, const float* x, const float* const* W,
                                   const float* h_prev, float scale, float bias_mul, float slope, int bs, int groups, int in_dim, int out_dim,
                                   int osplit, void* stream);

/* out[j][n] = sum_k T[j][k] w[n][k] (trans = 0; w [N][K], out [J][N]) or its transpose dw[n][k] = sum_j T[j][k] g[j][n] (trans = 1): a constant
 * J x K map (J, K <= 36) along a long axis — the composition of an up layer's 3x3 weight with its blur kernel into the four parity weights
 * (J = 36, K = 9, N = cout * cin; models/stylegan2/model.py:287-300 composed) and its gradient.  grouped != 0 (J = 36, K = 9 only):
 * the J side is stored [4][N][9] — out[g][n][t] for j = 9 g + t — i.e. the four parity weights as [4][cout][cin][3][3]. */
E4S_API int e4s_small_map(float* out, const float* T, const float* in, int J, int K, int64_t N, int trans, int grouped, void* stream);

/* ------------------------------------------------------------------------------------ a8 / a9: plain convolutions
 * Replaces the F.conv2d calls of the regional-style encoder (models/encoders/psp_encoders.py:334, helpers.py:128-139)
 * and of BiSeNet / ResNet-18 (swap_face_fine/face_parsing/model.py:23-35, resnet.py:15-49) with one implicit-GEMM kernel on
 * fp32 MFMA.  Weights are first re-laid out K-major (once per parameter version):
 *   wt[ci][tap][co] = weight[co][ci][tap] * g[co];  bias_out[co] = beta - mean*g (+ conv_bias*g);  g = gamma/sqrt(var+eps)
 * (BatchNorm2d in eval mode folds into the bias-free conv in front of it; pass bn_* = NULL for a plain conv). */
E4S_API int e4s_conv_prep_weights(float* wt, float* bias_out, const float* weight,
                                  const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps,
                                  const float* conv_bias, int cout, int cin, int kh, int kw, void* stream);

/* out[bs,cout,ho,wo] = act( conv(x', wt) + bias + residual ),  ho = (h + 2*pad - ks)/stride + 1.
 *   x0 / x1 : input channels [0,cin0) come from x0 [bs,cin0,h,w], [cin0,cin) from x1 [bs,cin-cin0,h,w] (x1 = NULL: all from x0) —
 *             the channel concatenation of FeatureFusionModule (face_parsing/model.py:207) without a copy
 *   in_mean / in_rstd : [bs,cin] or NULL — InstanceNorm2d of the INPUT applied while staging, x' = (x - mean)*rstd inside the
 *             image and 0 in the padding (helpers.py:134: InstanceNorm2d -> Conv2d)
 *   act : 0 none, 1 ReLU, 2 PReLU(prelu_slope[cout]);  residual : [bs,cout,ho,wo] added before the activation (resnet.py:46-48)
 *   ks/stride : (3,1) (3,2) (1,1) (1,2) (7,2) */
E4S_API int e4s_conv2d(float* out, const float* x0, const float* x1, int cin0, const float* wt, const float* bias,
                       const float* in_mean, const float* in_rstd, const float* prelu_slope, const float* residual, int act,
                       int bs, int cin, int cout, int h, int w, int ks, int stride, int pad, void* stream);

/* Split-bf16 variants (3 bf16 MFMAs per fp32 product, fp32 accumulate) of the two calls above, for 3x3 / 1x1
 * kernels.  whi / wlo: bf16 (as uint16) [ceil(cin/16)][kh*kw][2][cout][8] — element (chunk, tap, half, co, e) is input channel
 * chunk*16 + half*8 + e.  Same fusions as e4s_conv2d. */
E4S_API int e4s_conv_prep_weights_sb(uint16_t* whi, uint16_t* wlo, float* bias_out, const float* weight,
                                     const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps,
                                     const float* conv_bias, int cout, int cin, int kh, int kw, void* stream);
E4S_API int e4s_conv2d_sb(float* out, const float* x0, const float* x1, int cin0, const uint16_t* whi, const uint16_t* wlo, const float* bias,
                          const float* in_mean, const float* in_rstd, const float* prelu_slope, const float* residual, int act,
                          int bs, int cin, int cout, int h, int w, int ks, int stride, int pad, void* stream);
/* Three-way split (w = w0 + w1 + w2, x likewise; 6 bf16 MFMAs per 16-deep step: a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1, fp32
 * accumulate): fp32-class error (~2^-24 per product) at 2.7x less matrix-pipe time than the exact fp32 kernel.  Meant for the face
 * parser (swap_face_fine/face_parsing/model.py:20-260), whose argmax must not move.  Slabs from e4s_conv_prep_weights_sb3. */
E4S_API int e4s_conv_prep_weights_sb3(uint16_t* w0, uint16_t* w1, uint16_t* w2, float* bias_out, const float* weight,
                                      const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var,
                                      float bn_eps, const float* conv_bias, int cout, int cin, int kh, int kw, void* stream);
E4S_API int e4s_conv2d_sb3(float* out, const float* x0, const float* x1, int cin0, const uint16_t* w0, const uint16_t* w1,
                           const uint16_t* w2, const float* bias, const float* in_mean, const float* in_rstd,
                           const float* prelu_slope, const float* residual, int act, int bs, int cin, int cout, int h, int w,
                           int ks, int stride, int pad, void* stream);
/* Round 3: the fp32-class convolution at half the MFMAs of the three-way bf16 split — TWO f16 terms per operand (11 significand bits each where bf16 has 8),
 * a1*b1 + a1*b2 + a2*b1 on v_mfma_f32_32x32x16_f16, ~2^-23 per product.  w1 / w2: f16 (as uint16) slabs of weight * 2^wscale_log2 in the layout of
 * e4s_conv_prep_weights_sb — the power of two keeps the second term a normal f16; pick it so that the largest (BatchNorm-folded) weight lands near 2^10 and
 * pass the same value to e4s_conv2d_f16x3, which takes it out again.  Activations are used as they are (|x| < 65520 — a wave that stages a larger one raises
 * flags[0] bit 0 and bumps the counter flags[1] (flags may be NULL), the result is then invalid and the caller re-runs the pass on e4s_conv2d_sb3; their second term loses bits below
 * |x| ~ 2^-3, harmless next to the O(1) activations of the networks on this path).  Same fusions as e4s_conv2d. */
E4S_API int e4s_conv_prep_weights_f16x3(uint16_t* w1, uint16_t* w2, float* bias_out, const float* weight,
                                        const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps,
                                        const float* conv_bias, int cout, int cin, int kh, int kw, int wscale_log2, void* stream);
E4S_API int e4s_conv2d_f16x3(float* out, const float* x0, const float* x1, int cin0, const uint16_t* w1, const uint16_t* w2, const float* bias,
                             const float* in_mean, const float* in_rstd, const float* prelu_slope, const float* residual, int act,
                             int bs, int cin, int cout, int h, int w, int ks, int stride, int pad, int wscale_log2, int* flags, void* stream);

/* Per-plane statistics of x [planes = bs*C, hw]: mean, rstd = 1/sqrt(biased var + eps) (InstanceNorm2d without affine / running
 * stats, helpers.py:133,138), nmean = mean of the normalised plane (what SEModule's avg_pool sees, helpers.py:66).  rstd and nmean
 * may be NULL (plain global average pooling: face_parsing/model.py:83, 116, 209). */
E4S_API int e4s_plane_stats(float* mean, float* rstd, float* nmean, const float* x, int planes, int hw, float eps, void* stream);

/* y[bs,cout] = act( bn( x[bs,cin] @ W[cout,cin]^T ) ), bn optional (eval BatchNorm), act: 0 none, 1 ReLU, 3 sigmoid.
 * The 1x1 convolutions on pooled vectors: SEModule fc1/fc2 (helpers.py:67-71), ARM attention (face_parsing/model.py:84-86),
 * FFM attention (:210-213), conv_avg (:117). */
E4S_API int e4s_vec_fc(float* y, const float* x, const float* W, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                       const float* bn_var, float bn_eps, int act, int bs, int cin, int cout, void* stream);

/* out = prelu( ((x - mean)*rstd) * gate + shortcut' ),  every modifier optional (NULL):
 *   mean,rstd,gate,sc_mean,sc_rstd : [bs,C];  shortcut : [bs,C,h*sc_stride,w*sc_stride] sampled at (y*sc_stride, x*sc_stride)
 *   (MaxPool2d(1,stride), helpers.py:126) and instance-normalised when sc_mean is given (helpers.py:128-131);  prelu : [C].
 * Tail of bottleneck_IR_SE_Ours.forward (helpers.py:141-144) and InstanceNorm+PReLU of the input layer (psp_encoders.py:335-336). */
E4S_API int e4s_norm_gate_add(float* out, const float* x, const float* mean, const float* rstd, const float* gate,
                              const float* shortcut, const float* sc_mean, const float* sc_rstd, int sc_stride, const float* prelu,
                              int bs, int C, int h, int w, void* stream);
/* SEModule's gate (helpers.py:56-72) in one launch: gate[b, o] = sigmoid(fc2 . relu(fc1 . pooled[b])), fc1 [H, C], fc2 [C, H], no biases, H <= 64;
 * value for value what two e4s_vec_fc calls give. */
E4S_API int e4s_se_gate(float* gate, const float* pooled, const float* w1, const float* w2, int bs, int C, int H, void* stream);
/* e4s_norm_gate_add that also returns InstanceNorm statistics (mean, 1/sqrt(var + eps), as e4s_plane_stats computes them) of its OUTPUT planes:
 * the next bottleneck_IR_SE unit (helpers.py:122-144) normalises exactly that tensor.  Planes of at most 16384 pixels, a multiple of 4. */
E4S_API int e4s_norm_gate_add_stats(float* out, float* out_mean, float* out_rstd, const float* x, const float* mean, const float* rstd,
                                    const float* gate, const float* shortcut, const float* sc_mean, const float* sc_rstd, int sc_stride,
                                    const float* prelu, int bs, int C, int h, int w, float eps, void* stream);
/* The same with the InstanceNorm statistics of the INPUT x computed in the launch as well (mean and 1 / sqrt(var + in_eps) of every plane, exactly e4s_plane_stats' sums):
 * out = prelu( IN(x) * gate + shortcut' ) and the statistics of out — InstanceNorm2d(depth) + SEModule + the shortcut add of bottleneck_IR_SE_Ours (helpers.py:128-144) in one
 * launch, for a gate that does not depend on x (the host passes the constant 1/2: see ops.SE_GATE_IS_HALF). */
E4S_API int e4s_norm_self_gate_add_stats(float* out, float* out_mean, float* out_rstd, const float* x, float in_eps, const float* gate, const float* shortcut,
                                         const float* sc_mean, const float* sc_rstd, int sc_stride, const float* prelu, int bs, int C, int h, int w, float eps,
                                         void* stream);

/* Masked average pooling per region (psp_encoders.py:355-375): out[bs,nreg,C] = mean of feats[bs,C,h,w] over the pixels whose
 * label (uint8 [bs,lh,lw], sampled nearest at h x w) equals the region, zeros for an empty region. */
E4S_API int e4s_masked_avg_pool(float* out, const float* feats, const uint8_t* labels, int lh, int lw,
                                int bs, int C, int h, int w, int nreg, void* stream);

/* F.interpolate(mode='bilinear', align_corners=...) on [planes, ih, iw] -> [planes, oh, ow], no antialias
 * (models/networks.py:217 uses align_corners=False; face_parsing/model.py:257-259 uses True). */
E4S_API int e4s_bilinear_resize(float* out, const float* in, int planes, int ih, int iw, int oh, int ow, int align_corners, void* stream);

/* ------------------------------------------------------------------------------------ a9 / a10: parser glue */
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on [planes,h,w] (resnet.py:65, 75). */
E4S_API int e4s_maxpool3x3s2(float* out, const float* in, int planes, int h, int w, void* stream);

/* out[p,Y,X] = feat[p,Y/up,X/up]*gate[p] + add_map[p,Y/up,X/up] + add_vec[p]  (gate/add_map/add_vec optional), planes = bs*C:
 * ARM gating + context add + nearest upsample of ContextPath (face_parsing/model.py:118-128) and FFM's feat*atten+feat (:214-215). */
E4S_API int e4s_gate_add_upsample(float* out, const float* feat, const float* gate, const float* add_map, const float* add_vec,
                                  int planes, int h, int w, int up, void* stream);

/* The face parser's ResNet stem, Conv2d(3, 64, 7, stride 2, pad 3) + folded BatchNorm + ReLU (swap_face_fine/face_parsing/resnet.py:57-58, 66), as an implicit
 * GEMM over the flattened (channel, ky, kx) axis (K = 147 in ten 16-deep steps) in the two-term f16 split of e4s_conv2d_f16x3:
 *   out [bs,64,ho,wo] = act(conv(x [bs,3,h,w]) * 2^-wscale_log2 + bias);  w = the two f16 terms of W * 2^wscale_log2 as [term 2][K step 10][K half 2][co 64][8],
 *   K = (c*7 + ky)*7 + kx zero-padded to 160; bias [64] or NULL; relu 0 / 1. */
E4S_API int e4s_conv7x7s2_stem_f16x3(float* out, const float* x, const void* w, const float* bias, int bs, int h, int wd, int relu, int wscale_log2, void* stream);

/* labels[bs,oh,ow] (uint8) = lut[ argmax_c bilinear_align_corners(logits[bs,ncls,ih,iw]) ] — fuses F.interpolate(..., align_corners=True)
 * (face_parsing/model.py:257), torch.argmax (face_parsing_demo.py:170) and, through the optional 256-entry lut, the 19->12 remap
 * (datasets/dataset.py:58-108).  First maximum wins ties. */
E4S_API int e4s_bilinear_argmax(uint8_t* labels, const float* logits, const uint8_t* lut, int bs, int ncls, int ih, int iw, int oh, int ow,
                                void* stream);

/* BicubicDownSample(factor) + clamp(0,1) + (x-mean)/std of FaceParser.preprocess_img (face_parsing_demo.py:46-84, 151-156):
 * in [bs,C,h,w] in [0,1] -> out [bs,C,h/factor,w/factor]; taps = the 4*factor normalised 1-D weights; mean/std [C] or NULL
 * (NULL: plain down-sample, no clamp).  factor 2 or 4; factor 1 = clamp + normalise only (taps ignored). */
E4S_API int e4s_bicubic_down_normalize(float* out, const float* in, const float* taps, const float* mean, const float* stdv,
                                       int bs, int C, int h, int w, int factor, void* stream);
/* The same on an image in [-1, 1] (what the swap pipeline holds): the [0, 1] image the parser is given there
 * (face_swap_video_pipeline.py:217-219: PIL frames -> ToTensor, face_parsing_demo.py:151-156) is (v + 1) * 0.5, applied on load — value for value
 * what ``e4s_bicubic_down_normalize`` computes from ``(img + 1) / 2``.  factor 2 or 4. */
E4S_API int e4s_bicubic_down_normalize_pm1(float* out, const float* in, const float* taps, const float* mean, const float* stdv,
                                           int bs, int C, int h, int w, int factor, void* stream);

/* tensor2im (utils/torch_utils.py:64-76): img [bs,3,h,w] in ~[-1,1] -> uint8 [bs,h,w,3] = trunc(clamp((x+1)/2, 0, 1) * 255). */
E4S_API int e4s_tensor2im_u8(uint8_t* out, const float* img, int bs, int h, int w, void* stream);

/* ---- f2 / f3: mask surgery between face parsing and synthesis (integer maps, exact) --------------------------------------------
 * e4s_swap_head_mask: swap_head_mask_hole_first(source, target) (swap_face_fine/swap_face_mask.py:194-333) for a batch.
 *   source, target : uint8 [bs, h, w] 12-class maps (driven face, target face)
 *   res            : uint8 [bs, h, w] swapped map;  hole_mask : uint8 {0,1};  hole_map : res with 17 on the hole
 *   lines          : int32 [bs, 2] = (eye_line, nose_line) as computed at :232-239
 *   scratch        : int32 [bs * (3 + w)] work buffer
 * e4s_foreground_masks: foreground = not {0, 11, 7, 4, 8} or hole (face_swap_video_pipeline.py:456-461), then
 *   create_masks(foreground, 'expansion', radius) (gradio_utils/face_swapping.py:203-221; flat (2r+1)^2 dilation / erosion with the
 *   'geodesic' border of utils/morphology.py): content = foreground, full = dilation, border = clip(dilation - erosion, 0, 1).
 *   content (may be NULL), border, full : float32 [bs, 1, h, w] in {0, 1};  hole_mask may be NULL. */
E4S_API int e4s_swap_head_mask(uint8_t* res, uint8_t* hole_mask, uint8_t* hole_map, int32_t* lines, const uint8_t* source,
                               const uint8_t* target, int32_t* scratch, int bs, int h, int w, void* stream);
/* One pass of Pillow's 8-bit image resampler (PIL.Image.resize, src/libImaging/Resample.c) — the reference softens every swapped face with
 * `.resize((512, 512)).resize((1024, 1024))` (face_swap_video_pipeline.py:447, default filter BICUBIC) — on uint8 [bs, h, w, c] images:
 * along axis 1 (width) or 0 (height), out[o] = clip8((2^21 + sum_{j < count[o]} coeffs[o*ksize + j] * in[xmin[o] + j]) >> 22).
 * The tables are Pillow's precompute_coeffs + normalize_coeffs_8bpc (22-bit fixed point), computed by the host (ops.pil_resize). */
E4S_API int e4s_resample_u8(uint8_t* out, const uint8_t* in, const int32_t* xmin, const int32_t* count, const int32_t* coeffs, int ksize,
                            int bs, int h, int w, int c, int out_size, int axis, void* stream);
/* The two pyramid steps of the multi-band blend (swap_face_fine/multi_band_blending.py:5-48; cv2.pyrDown / cv2.pyrUp, OpenCV
 * modules/imgproc/src/pyramids.cpp) on float planes [planes, h, w]:
 *   e4s_pyr_down : out [planes, (h+1)/2, (w+1)/2] = 5x5 kernel [1 4 6 4 1]^2 / 256 at the even pixels, BORDER_REFLECT_101;
 *                  round_u8 != 0 = the 8-bit variant's rounding, floor((sum + 128) / 256) (the reference's pyramid of the uint8 frame)
 *   e4s_pyr_up   : out [planes, 2h, 2w] = up(in), or minuend - up(in) (a Laplacian level), or up(in) + addend (reconstruction);
 *                  up = zero insertion convolved with 4x the same kernel (OpenCV's border rule for the last source pixel) */
E4S_API int e4s_pyr_down(float* out, const float* in, int planes, int h, int w, int round_u8, void* stream);
E4S_API int e4s_pyr_up(float* out, const float* in, const float* minuend, const float* addend, int planes, int h, int w, void* stream);
/* One level of the blend (multi_band_blending.py:27-46) in one pass over [planes, 2h, 2w]: with la = a_hi - up(a_lo), lb = b_hi - up(b_lo),
 * out = up(prev) + la*m_hi + lb*(1 - m_hi);  *_lo and prev are [planes, h, w], *_hi [planes, 2h, 2w]. */
E4S_API int e4s_pyr_blend_level(float* out, const float* prev, const float* a_hi, const float* a_lo, const float* b_hi, const float* b_lo,
                                const float* m_hi, int planes, int h, int w, void* stream);
/* uint8 frames [bs, h, w, 3] -> the network's input [bs, 3, h, w] = (x / 255 - 0.5) / 0.5: transforms.Compose([ToTensor(), Normalize(.5, .5)])
 * of datasets/dataset.py:32, 45 as used at face_swap_video_pipeline.py:338-339 (float32, true division like torchvision). */
E4S_API int e4s_frames_to_tensor(float* out, const uint8_t* frames_u8, int bs, int h, int w, void* stream);
/* erode_mask(mask, img, radius) of the PTI loop (training/video_swap_ft_coach.py:64-93) on uint8 [bs, h, w] 12-class maps: the face mask
 * (every class whose bit is NOT set in bg_class_bits; the reference's background set {0, 4, 11} = 0x811) is eroded with a (2r+1)^2 box,
 * pixels outside the image counting as not-face (cv2.erode, BORDER_CONSTANT 0); out = label where the eroded mask holds, 0 elsewhere. */
E4S_API int e4s_erode_labels(uint8_t* out, const uint8_t* labels, int bs, int h, int w, int radius, unsigned bg_class_bits, void* stream);
E4S_API int e4s_foreground_masks(float* content, float* border, float* full, const uint8_t* swapped, const uint8_t* hole_mask,
                                 int bs, int h, int w, int radius, void* stream);

/* ---- f1: gradients of the one-pass region-modulated convolution (PTI tuning, training/video_swap_ft_coach.py:242-299) -----------
 *     z[b,o,p] = sum_{i,k} W[o,i,k] * s[b,c(p),i] * x[b,i,p+k-pad],   y = d[b,c(p),o] * z        (model.py:389-398, 447-454)
 *     out = leaky_relu(y + noise_weight*noise + act_bias, 0.2) * sqrt(2)                          (model.py:335, 421)
 * The two GEMMs of the backward (U_g = W_g^T gz_g, dW_g = gz_g cols_g^T) are library fp32 GEMMs on the host side; these are the passes
 * around them.  x [bs,cin,h,w], s [bs,nreg,cin], d [bs,nreg,cout] (NULL = no demodulation), ks 1 or 3 (pad ks/2, stride 1).
 * up = 1: labels uint8 [bs,h,w], one group.  up = 2 (the up-sampling layers in their composed form): labels
 * [bs,2h,2w] at the OUTPUT resolution, four groups g = 2a+b, one per output parity: output pixel (2qy+a, 2qx+b) is a 3x3 correlation
 * of x around q with the composed weight W_g under c_g(q) = labels[2qy+a][2qx+b].  A label >= nreg is no region (zero, no gradient).
 *   e4s_mconv_unfold : cols[G, bs, cin*ks*ks, h*w] = s[c_g(q),i] * x[i, q+k-pad]
 *   e4s_mconv_scale  : from gy = dL/d(out) [bs,cout,up*h,up*w] and the forward output `out`:  g' = gy * act'(out);
 *                      gz[G,bs,cout,h*w] = g' * d[c(p),o] (parity-planar);  per pixel chunk (nchunk = ceil(up*h*up*w / chunk_px), the
 *                      caller adds the chunks up): q[nchunk,bs,nreg,cout] = per-region sums of g'*y (dL/dd = q / d),
 *                      dbias[nchunk,bs,cout] = sum g',  dnw[nchunk,bs,cout] = sum g'*noise.  q, dbias, dnw, out, noise, act_bias optional;
 *                      with act = 0 and no noise / bias, `out` is y itself.  noise [noise_bs, up*h*up*w], noise_bs 1 or bs.
 *   labels NULL (all three) = one region, every pixel is region 0: the single-region layers of the generator (nreg = 1).
 *   e4s_mconv_fold   : from U[G, bs, cin*ks*ks, h*w]:  dx[b,i,t] = sum_g sum_k s[c_g(t-k+pad),i] * U[g,b,(i,k),t-k+pad]  (may be NULL)
 *                      ds_part[nchunk, bs, nreg, cin] = per pixel-chunk partial sums of sum_g sum_k U[g,b,(i,k),q] * x[i,q+k-pad] over
 *                      each region's pixels (may be NULL), nchunk = ceil(h*w / chunk_px); dL/ds = ds_part.sum(0) */
/* Gradient of a layer's style tables (EqualLinear modulation + demodulation, model.py:150-161, 276-281, in the one-pass form):
 *     s = styles (mod_w*mod_scale)^T + mod_b*mod_lr,   ws = weight_scale*weight,   d = rsqrt(s^2 wsq + 1e-8),  wsq[i,o] = sum_k ws[o,i,k]^2 ([cin,cout], as e4s_modconv_prep_weights writes it)
 * gs [rows,cin], gd [rows,cout], gws [cout,cin,kk] = dL/ds, dL/dd, dL/dws (each may be NULL; gd needs d and wsq), rows = bs*nreg.
 * Outputs: g_styles [rows,sdim], g_mod_w [cin,sdim], g_mod_b [cin] (written when gs or gd is given), g_weight [cout,cin,kk] w.r.t. the
 * unscaled weight (written when gws or gd is given).  scratch: rows*(cout+cin) floats. */
E4S_API int e4s_style_tables_bwd(float* g_styles, float* g_mod_w, float* g_mod_b, float* g_weight, float* scratch, const float* gs,
                                 const float* gd, const float* gws, const float* styles, const float* mod_w, const float* s, const float* d,
                                 const float* weight, const float* wsq, float weight_scale, float mod_scale, float mod_lr, int rows, int sdim,
                                 int cin, int cout, int kk, void* stream);
E4S_API int e4s_mconv_unfold(float* cols, const float* x, const float* s, const uint8_t* labels, int bs, int cin, int h, int w, int ks,
                             int nreg, int up, void* stream);
E4S_API int e4s_mconv_scale(float* gz, float* q, float* dbias, float* dnw, const float* gy, const float* out, const float* d,
                            const uint8_t* labels, const float* noise, int noise_bs, const float* noise_weight, const float* act_bias,
                            int act, int bs, int cout, int h, int w, int nreg, int up, int chunk_px, void* stream);
/* cols[bs, C*ks*ks, ho*wo] = x[bs, C, stride*qy + ky - pad, stride*qx + kx - pad] (0 outside): the unfolded operand of a plain convolution's
 * weight gradient as a library GEMM — the single-region layers past remaining_layer_idx (conv: dW = g' cols(x)^T, stride 1 pad 1;
 * transposed conv, model.py:296-306: dW = x cols(gT)^T with stride 2 pad 0). */
E4S_API int e4s_unfold2d(float* cols, const float* x, int bs, int C, int hi, int wi, int ho, int wo, int ks, int stride, int pad, void* stream);
E4S_API int e4s_mconv_fold(float* dx, float* ds_part, const float* U, const float* x, const float* s, const uint8_t* labels, int bs, int cin,
                           int h, int w, int ks, int nreg, int up, int chunk_px, void* stream);


/* The contractions of the backward (f1) — replaces the library GEMMs (torch.matmul) behind the reference's autograd of
 * ModulatedConv2d (models/stylegan2/model.py:276-320 differentiated by PyTorch; training/video_swap_ft_coach.py:268-299 is the loop):
 *   C[b] (M x N, row-major, dense [batch][M][N] at stride_c) = opA(A[b]) (M x K) * opB(B[b]) (K x N),  fp32 in / out,
 * each product as three bf16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulation).  a_kc != 0: A is stored [M][K] with row stride lda,
 * else [K][M]; b_kc != 0: B is stored [N][K] with row stride ldb, else [K][N].  K-contiguous operands need 16-byte aligned rows and
 * K % 4 == 0.  Batch strides in floats (0 = one matrix for all b).  workspace (may be NULL): scratch of workspace_floats floats for
 * the split of a long K over workgroups; partial sums are added in a fixed order. */
E4S_API int e4s_gemm_sb(float* c, const float* a, const float* b, int M, int N, int K, int a_kc, int b_kc, int lda, int ldb, int64_t stride_a,
                        int64_t stride_b, int64_t stride_c, int batch, float* workspace, int64_t workspace_floats, void* stream);

/* Weight gradient of the (masked) modulated convolution as one implicit GEMM — the modulated im2col operand of e4s_mconv_unfold is produced
 * while it is staged, never stored:  dW[g][b][co][(ci, k)] = sum_p gz[g][b][co][p] * s[b][c_g(p)][ci] * x[b][ci][p + k - pad],
 * g < up * up (composed weights of the four output parities for up = 2, labels at the output resolution).  s == NULL: no modulation;
 * labels == NULL: one region.  dw: dense [up * up * bs][cout][cin * ks * ks].  w % 16 == 0.  workspace as for e4s_gemm_sb. */
E4S_API int e4s_mconv_wgrad(float* dw, const float* gz, const float* x, const float* s, const uint8_t* labels, int bs, int cin, int cout, int h,
                            int w, int ks, int nreg, int up, float* workspace, int64_t workspace_floats, void* stream);

/* Data and style gradient of the masked modulated 3x3 convolution in one kernel — e4s_gemm_sb (U = W^T gz) + e4s_mconv_fold without U in memory
 * (same reference path: autograd through ModulatedConv2d.forward, models/stylegan2/model.py:276-320):
 *     dx[b,i,q]            = sum_g sum_k s[b,c_g(q-k+1),i] * sum_o wg[g,o,i,k] * gz[g,b,o,q-k+1]               (may be NULL)
 *     ds_part[t,b,r,i]     = tile t's part of sum_g sum_{p: c_g(p) = r} sum_k (sum_o wg[g,o,i,k] gz[g,b,o,p]) * x[b,i,p+k-1]   (may be NULL)
 * gz [up*up, bs, cout, h*w] (e4s_mconv_scale), wg [up*up, cout, cin, 3, 3] fp32, labels as for e4s_mconv_fold (NULL = one region), w >= 32.
 * t < e4s_mconv_dgrad_tiles(h, w); dL/ds = ds_part.sum(0).  Every product as three bf16 MFMAs with fp32 accumulation; sums in a fixed order. */
E4S_API int e4s_mconv_dgrad_tiles(int h, int w);
E4S_API int e4s_mconv_dgrad(float* dx, float* ds_part, const float* gz, const float* wg, const float* x, const float* s, const uint8_t* labels,
                            int bs, int cin, int cout, int h, int w, int nreg, int up, void* stream);

/* Winograd F(2x2, 3x3) pieces around e4s_gemm_sb(batch = 16) for stride-1, pad-1 3x3 convolutions (the regional-style encoder's IR-SE units,
 * models/encoders/helpers.py:122-144: Conv2d(c, d, 3, 1, 1) after InstanceNorm2d / before PReLU):
 *   e4s_wino_weight : U[16][cout][cin] = G g G^T of w [cout][cin][3][3]
 *   e4s_wino_input  : V[16][C][T]      = B^T d B of every 4x4 patch (stride 2) of x [bs][C][H][W], T = bs * H/2 * W/2, tile t = (b * H/2 + ty) * W/2 + tx;
 *                     mean / rstd [bs][C] (both or neither): the patch is taken from (x - mean) * rstd, zero beyond the border
 *   then M[k] (cout x T) = U[k] (cout x cin) * V[k] (cin x T) for k < 16 — e4s_gemm_sb(M, U, V, cout, T, cin, 1, 0, cin, T, cout*cin, cin*T, cout*T, 16, ...)
 *   e4s_wino_output : y [bs][cout][H][W] = act(A^T m A) of m = M[.][co][t]; prelu [cout] or NULL (no activation)
 * H, W even. */
E4S_API int e4s_wino_weight(float* U, const float* w, int cout, int cin, void* stream);
E4S_API int e4s_wino_input(float* V, const float* x, const float* mean, const float* rstd, int bs, int C, int H, int W, void* stream);
E4S_API int e4s_wino_output(float* y, const float* M, const float* prelu, int bs, int cout, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* E4S_HIP_H */
