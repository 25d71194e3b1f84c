/* CPU oracle for the reference's two native kernels.  TEST INFRASTRUCTURE ONLY —
 * linked by tests/ and by bench.py's cpu_baseline leg, never by the product.
 *
 * Plain-C restatement of
 *   models/stylegan2/op/fused_bias_act_kernel.cu:18-49  (fused_bias_act_kernel)
 *   models/stylegan2/op/upfirdn2d_kernel.cu:52-137      (upfirdn2d_kernel)
 * written per output element from the kernels' index arithmetic (no tiling, no
 * shared memory).  Pinned by tests/golden/g1_fused_act.npz and g2_upfirdn2d.npz, which
 * hold outputs of the reference's own CPU path for the same ops.
 */
#include <stddef.h>
#include <stdint.h>

/* act*10+grad selector of the .cu: 10/11 linear, 12 zero, 30 lrelu fwd, 31 lrelu bwd
 * (sign taken from `ref`), 32 zero.  bias index = (i / step_b) % size_b (:29). */
void oracle_fused_bias_act(float *out, const float *x, const float *b, const float *ref,
                           int act, int grad, float alpha, float scale,
                           int64_t size_x, int64_t step_b, int64_t size_b)
{
    for (int64_t i = 0; i < size_x; ++i) {
        float v = x[i];
        if (b && size_b > 0) v += b[(i / step_b) % size_b];
        float r = ref ? ref[i] : 0.0f;
        float y;
        switch (act * 10 + grad) {
        default:
        case 10: case 11: y = v; break;
        case 12: y = 0.0f; break;
        case 30: y = (v > 0.0f) ? v : v * alpha; break;
        case 31: y = (r > 0.0f) ? v : v * alpha; break;
        case 32: y = 0.0f; break;
        }
        out[i] = y * scale;
    }
}

static int floor_div(int a, int b) { int c = a / b; if (c * b > a) c--; return c; }

/* input [major, in_h, in_w] (minor == 1, the only layout the Python wrapper produces:
 * op/upfirdn2d.py:96), kernel [kh, kw]; out [major, out_h, out_w] with
 * out = (in*up + pad0 + pad1 - k) / down + 1 (op/upfirdn2d.py:100-101).
 * Taps are applied flipped (upfirdn2d_kernel.cu:77), i.e. a true convolution. */
void oracle_upfirdn2d(float *out, const float *in, const float *kernel,
                      int major, int in_h, int in_w, int kh, int kw,
                      int up_x, int up_y, int down_x, int down_y,
                      int pad_x0, int pad_x1, int pad_y0, int pad_y1)
{
    int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
    int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
    for (int m = 0; m < major; ++m)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox) {
                /* :114-121 */
                int mid_x = ox * down_x + up_x - 1 - pad_x0;
                int mid_y = oy * down_y + up_y - 1 - pad_y0;
                int ix0 = floor_div(mid_x, up_x);
                int iy0 = floor_div(mid_y, up_y);
                int kx0 = (ix0 + 1) * up_x - mid_x - 1;
                int ky0 = (iy0 + 1) * up_y - mid_y - 1;
                float v = 0.0f;
                for (int y = 0; ky0 + y * up_y < kh; ++y)
                    for (int x = 0; kx0 + x * up_x < kw; ++x) {
                        int iy = iy0 + y, ix = ix0 + x;
                        if (iy < 0 || ix < 0 || iy >= in_h || ix >= in_w) continue;
                        int ky = ky0 + y * up_y, kx = kx0 + x * up_x;
                        v += in[((size_t)m * in_h + iy) * in_w + ix] *
                             kernel[(kh - 1 - ky) * kw + (kw - 1 - kx)];
                    }
                out[((size_t)m * out_h + oy) * out_w + ox] = v;
            }
}
