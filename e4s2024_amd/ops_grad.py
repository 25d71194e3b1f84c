"""Row f1 of the scope table: the gradients of the synthesis path on this library's kernels (``csrc/modconv_bwd.hip``, ``csrc/gemm_sb.hip``,
``csrc/mconv_dgrad.hip``) as ``torch.autograd.Function``s around the forward kernels' own outputs, plus the gradients of the per-region MLPs and
the small style tables.  Reference: the autograd graph of ``models/stylegan2/model.py:276-320, 382-479`` and ``models/networks.py:23-49`` as
``training/video_swap_ft_coach.py:253-299`` differentiates it.

Everything here is re-exported by ``ops`` (``ops.gemm_sb``, ``ops.masked_styled_conv_grad`` ...); the stage's switches (``ops.NATIVE_BWD``,
``ops.DGRAD_FUSED`` ...) live in ``ops`` and are read there at call time, so tests and tools keep setting them as attributes of ``ops``.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import ops
from ._lib import lib
from .ops import PreparedConv, _c, _p, _stream, conv2d, grouped_linear, upfirdn2d_raw
from .ops_post import _labels_u8

# ------------------------------------------------------------------------------------ f1: native gradients of the masked conv


def _mconv_unfold(x, s, lab, ks: int, up: int = 1):
    bs, cin, h, w = x.shape
    cols = torch.empty((up * up, bs, cin * ks * ks, h * w), dtype=torch.float32, device=x.device)
    lib().call("e4s_mconv_unfold", _p(cols), _p(x), _p(s), _p(lab), bs, cin, h, w, ks, s.shape[1], up, _stream())
    return cols


def _mconv_scale(gy, out, d, lab, nreg: int, up: int = 1, want_q: bool = False, noise=None, noise_weight=None, act_bias=None, act: bool = False,
                 want_sums: bool = False):
    """-> gz [up*up, bs, cout, h*w], q [bs, nreg, cout] | None, dbias [bs, cout] | None, dnw [bs, cout] | None  (csrc/modconv_bwd.hip)"""
    bs, cout, ho, wo = gy.shape
    h, w = ho // up, wo // up
    nchunk = -(-(ho * wo) // ops._SCALE_CHUNK_PX)
    gz = torch.empty((up * up, bs, cout, h * w), dtype=torch.float32, device=gy.device)
    q = torch.empty((nchunk, bs, nreg, cout), dtype=torch.float32, device=gy.device) if want_q else None
    dbias = torch.empty((nchunk, bs, cout), dtype=torch.float32, device=gy.device) if want_sums else None
    dnw = torch.empty((nchunk, bs, cout), dtype=torch.float32, device=gy.device) if want_sums and noise is not None else None
    lib().call("e4s_mconv_scale", _p(gz), _p(q), _p(dbias), _p(dnw), _p(gy), _p(out), _p(d), _p(lab), _p(noise), 0 if noise is None else noise.shape[0],
               _p(noise_weight), _p(act_bias), int(act), bs, cout, h, w, nreg, up, ops._SCALE_CHUNK_PX, _stream())
    return (gz,) + tuple(None if t is None else (t.sum(0) if nchunk > 1 else t[0]) for t in (q, dbias, dnw))


def _sum_dim(t, dim: int):
    """``t.sum(dim)`` without a launch when that dimension has one element (batch 1 is the PTI case)."""
    return t.select(dim, 0) if t.shape[dim] == 1 else t.sum(dim)




def gemm_sb(a: torch.Tensor, b: torch.Tensor, a_kc: bool, b_kc: bool, split_k: bool = True) -> torch.Tensor:
    """``C[i] = opA(a[i]) @ opB(b[i])`` on the bf16 matrix cores with the three-term split (``e4s_gemm_sb``, csrc/gemm_sb.hip) — the
    contractions of the backward pass.  ``a``: ``[Ba, M, K]`` if ``a_kc`` else ``[Ba, K, M]``; ``b``: ``[Bb, N, K]`` if ``b_kc`` else
    ``[Bb, K, N]``; ``Ba``, ``Bb`` are the batch or 1 (shared).  Returns fp32 ``[batch, M, N]``.  ``split_k=False``: every output element is one
    pass over K in a fixed order whatever the other dimensions are (a column's value does not depend on how many columns there are)."""
    a, b = _c(a, "a"), _c(b, "b")
    if a.dim() != 3 or b.dim() != 3:
        raise ValueError("gemm_sb takes 3-D operands [batch or 1, rows, cols]")
    M, K = (a.shape[1], a.shape[2]) if a_kc else (a.shape[2], a.shape[1])
    N, Kb = (b.shape[1], b.shape[2]) if b_kc else (b.shape[2], b.shape[1])
    batch = max(a.shape[0], b.shape[0])
    if K != Kb or a.shape[0] not in (1, batch) or b.shape[0] not in (1, batch):
        raise ValueError(f"gemm_sb: a {tuple(a.shape)} (a_kc={a_kc}) and b {tuple(b.shape)} (b_kc={b_kc}) do not fit")
    c = torch.empty((batch, M, N), dtype=torch.float32, device=a.device)
    # the split of a long K (sizes only: reproducible); the library applies the same rule under the workspace it is given
    skinny = a_kc and b_kc and M <= 8
    tm, tn = (32, 256) if skinny else (128, 128)
    base, nchunk, ks = -(-M // tm) * -(-N // tn) * batch, -(-K // 32), 1
    while split_k and base * ks < 512 and ks * 2 * 4 <= nchunk and ks < 1024 and ks * 2 * batch * M * N <= ops.GEMM_SPLITK_CAP_FLOATS:
        ks *= 2
    ws = torch.empty((ks * batch * M * N,), dtype=torch.float32, device=a.device) if ks > 1 else None
    lib().call("e4s_gemm_sb", _p(c), _p(a), _p(b), M, N, K, int(a_kc), int(b_kc), a.shape[2], b.shape[2],
               0 if a.shape[0] == 1 and batch > 1 else a.shape[1] * a.shape[2], 0 if b.shape[0] == 1 and batch > 1 else b.shape[1] * b.shape[2],
               M * N, batch, _p(ws), 0 if ws is None else ws.numel(), _stream())
    return c


def _gemm_nt(a, b):
    """``a [..., M, K] @ b [..., N, K]ᵀ`` (the weight gradients: K = all the pixels of a layer, split over workgroups inside the kernel)."""
    lead = a.shape[:-2]
    return gemm_sb(a.reshape(-1, *a.shape[-2:]), b.reshape(-1, *b.shape[-2:]), True, True).view(*lead, a.shape[-2], b.shape[-2])


def unfold2d(x, ks: int, stride: int, pad: int, ho: int, wo: int):
    """``cols [bs, C*ks*ks, ho*wo] = x[bs, C, stride*q + k - pad]`` (``e4s_unfold2d``)."""
    x = _c(x, "x")
    bs, ch, hi, wi = x.shape
    cols = torch.empty((bs, ch * ks * ks, ho * wo), dtype=torch.float32, device=x.device)
    lib().call("e4s_unfold2d", _p(cols), _p(x), bs, ch, hi, wi, ho, wo, ks, stride, pad, _stream())
    return cols




def _mconv_input_grads(gz, wg, x, s, lab, up: int, need_x: bool, need_s: bool, need_w: bool):
    """The part of the backward that follows ``gz``: U_g = W_gᵀ gz_g (``e4s_gemm_sb``), dx / ds from one pass over U (``e4s_mconv_fold``) — or both
    from ``e4s_mconv_dgrad`` without U in memory (``ops.DGRAD_FUSED``) — and dW_g = gz_g cols_gᵀ as an implicit GEMM (``e4s_mconv_wgrad``; the 4x4 / 8x8
    maps: unfold kernel + ``e4s_gemm_sb``)."""
    bs, cin, h, w = x.shape
    G, cout, ks, nreg = wg.shape[0], wg.shape[1], wg.shape[-1], s.shape[1]
    dx = ds = dw = None
    if (need_x or need_s) and ops.DGRAD_FUSED and ks == 3 and w >= ops.DGRAD_FUSED_MIN_WIDTH:
        # one kernel: the nine taps' U in accumulators, modulated and summed with their shifts in LDS (csrc/mconv_dgrad.hip)
        dx = torch.empty_like(x) if need_x else None
        ntile = lib().cdll.e4s_mconv_dgrad_tiles(h, w)
        part = torch.empty((ntile, bs, nreg, cin), dtype=torch.float32, device=x.device) if need_s else None
        lib().call("e4s_mconv_dgrad", _p(dx), _p(part), _p(_c(gz, "gz")), _p(_c(wg, "wg")), _p(x), _p(s), _p(lab), bs, cin, cout, h, w, nreg, up,
                   _stream())
        if need_s:
            ds = _sum_dim(part.view(ntile, -1), 0).view(bs, nreg, cin) if ntile > 1 else part[0]
    elif need_x or need_s:
        # U_g = W_gᵀ gz_g: [G, bs, cin*KK, P]; the weight is stored [cout][cin*KK] = [K][M], gz [cout][P] = [K][N]
        w2 = wg.reshape(G, cout, cin * ks * ks)
        if bs == 1:
            u = gemm_sb(w2, gz.view(G, cout, -1), False, False).view(G, 1, cin * ks * ks, -1)
        else:
            u = torch.stack([gemm_sb(w2[g:g + 1], gz[g], False, False) for g in range(G)])
        dx = torch.empty_like(x) if need_x else None
        nchunk = -(-(h * w) // ops._FOLD_CHUNK_PX)
        part = torch.empty((nchunk, bs, nreg, cin), dtype=torch.float32, device=x.device) if need_s else None
        lib().call("e4s_mconv_fold", _p(dx), _p(part), _p(u), _p(x), _p(s), _p(lab), bs, cin, h, w, ks, nreg, up, ops._FOLD_CHUNK_PX, _stream())
        del u
        if need_s:
            ds = part.sum(0) if nchunk > 1 else part[0]
    if need_w:
        if w % 16 == 0 and w >= 16:
            dw = _sum_dim(mconv_wgrad(gz, x, s, lab, cout, ks, up).view(G, bs, cout, -1), 1).view_as(wg)
        else:                                   # the 4x4 / 8x8 maps: explicit unfold (tiny)
            cols = _mconv_unfold(x, s, lab, ks, up)
            dw = _sum_dim(_gemm_nt(gz, cols), 1).view_as(wg)
    return dx, ds, dw


def mconv_wgrad(gz, x, s, lab, cout: int, ks: int, up: int = 1) -> torch.Tensor:
    """``dW [G*bs, cout, cin*ks*ks]`` of the (masked) modulated convolution from ``gz [G, bs, cout, h*w]`` (``_mconv_scale``) without the
    unfolded operand (``e4s_mconv_wgrad``: the modulated im2col rows are produced while the GEMM stages them).  ``s`` / ``lab`` None: a plain
    convolution / one region."""
    x, gz = _c(x, "x"), _c(gz, "gz")
    bs, cin, h, w = x.shape
    G = up * up
    nreg = 1 if s is None else s.shape[1]
    M, N, K, batch = cout, cin * ks * ks, h * w, G * bs
    if gz.numel() != batch * cout * K:
        raise ValueError(f"mconv_wgrad: gz {tuple(gz.shape)} is not [{G}, {bs}, {cout}, {K}]")
    dw = torch.empty((batch, M, N), dtype=torch.float32, device=x.device)
    base, nchunk, kspl = -(-M // 128) * -(-N // 128) * batch, -(-K // 32), 1
    while base * kspl < 512 and kspl * 2 * 4 <= nchunk and kspl < 1024 and kspl * 2 * batch * M * N <= ops.GEMM_SPLITK_CAP_FLOATS:
        kspl *= 2
    ws = torch.empty((kspl * batch * M * N,), dtype=torch.float32, device=x.device) if kspl > 1 else None
    lib().call("e4s_mconv_wgrad", _p(dw), _p(gz), _p(x), _p(None if s is None else _c(s, "s")), _p(None if lab is None else _labels_u8(lab, "labels")),
               bs, cin, cout, h, w, ks, nreg, up, _p(ws), 0 if ws is None else ws.numel(), _stream())
    return dw


def _check_mconv(x, wg, s, d, lab, up):
    bs, cin, h, w = x.shape
    G, cout, ks = wg.shape[0], wg.shape[1], wg.shape[-1]
    if wg.shape != (up * up, cout, cin, ks, ks) or s.dim() != 3 or s.shape[0] != bs or s.shape[2] != cin or lab.shape != (bs, up * h, up * w):
        raise ValueError(f"masked conv: x {tuple(x.shape)}, w {tuple(wg.shape)}, s {tuple(s.shape)}, labels {tuple(lab.shape)}, up {up} do not fit")
    if d is not None and d.shape != (bs, s.shape[1], cout):
        raise ValueError(f"masked conv: d {tuple(d.shape)} is not [bs, nreg, cout]")


class _MaskedConvCore(torch.autograd.Function):
    """``y[b,o,p] = d[b,c(p),o] · Σ_{i,k} W[o,i,k] · s[b,c(p),i] · x[b,i,p+k-pad]`` evaluated AND differentiated with the kernels of
    ``csrc/modconv_bwd.hip`` and the split-bf16 MFMA GEMM of ``csrc/gemm_sb.hip`` — the differentiable core ``torch_ref._region_sum`` uses on the
    device when a backward pass has to re-evaluate a masked layer (ToRGB).  The inference forward is the fused MFMA kernel, not this.

    ``x [bs,cin,h,w]``, ``w [cout,cin,ks,ks]`` (already scaled), ``s [bs,nreg,cin]``, ``d [bs,nreg,cout]`` or None, ``lab`` uint8 ``[bs,h,w]``."""

    @staticmethod
    def forward(ctx, x, w, s, d, lab):
        x, wg, s = _c(x, "x"), _c(w, "w")[None], _c(s, "s")
        d = _c(d, "d") if d is not None else None
        lab = _labels_u8(lab, "labels")
        _check_mconv(x, wg, s, d, lab, 1)
        bs, cin, h, wd = x.shape
        cout = wg.shape[1]
        z = gemm_sb(wg.reshape(1, cout, -1), _mconv_unfold(x, s, lab, wg.shape[-1]).view(bs, -1, h * wd), True, False).view(bs, cout, h, wd)
        y = _mconv_scale(z, None, d, lab, s.shape[1])[0].view(bs, cout, h, wd)        # y = z * d[c(p)], zero where the label is no region
        ctx.save_for_backward(x, wg, s, d, lab, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wg, s, d, lab, y = ctx.saved_tensors
        need_x, need_w, need_s, need_d = ctx.needs_input_grad[:4]
        gz, q, _, _ = _mconv_scale(gy.contiguous(), y, d, lab, s.shape[1], want_q=d is not None and need_d)
        dx, ds, dw = _mconv_input_grads(gz, wg, x, s, lab, 1, need_x, need_s, need_w)
        return dx, None if dw is None else dw[0], ds, None if q is None else q / d, None


def masked_conv_core(x, w, s, d, lab):
    return _MaskedConvCore.apply(x, w, s, d, lab)


class _MaskedStyledConvGrad(torch.autograd.Function):
    """A masked ``StyledConv`` whose forward value is already known (``out``, from the fused MFMA kernel) and whose gradients come from
    ``csrc/modconv_bwd.hip`` + ``csrc/gemm_sb.hip`` — no re-evaluation of the layer, no library GEMM:

        out = leaky_relu(d[c(p)] · Σ W_g · s[c(p)] · x  +  noise_weight · noise  +  act_bias) · √2

    ``wg [G,cout,cin,ks,ks]``: G = 1 (plain layer) or 4 (up-sampling layer: the composed weight of each output parity, labels at the output
    resolution).  Differentiable inputs: x, wg, s, d, noise_weight, act_bias."""

    @staticmethod
    def forward(ctx, x, wg, s, d, noise_weight, act_bias, lab, noise, act, out):
        x, wg, s = _c(x, "x"), _c(wg, "w"), _c(s, "s")
        d = _c(d, "d") if d is not None else None
        lab = _labels_u8(lab, "labels")
        up = {1: 1, 4: 2}[wg.shape[0]]
        _check_mconv(x, wg, s, d, lab, up)
        out = _c(out, "out")
        if out.shape != (x.shape[0], wg.shape[1], up * x.shape[2], up * x.shape[3]):
            raise ValueError(f"masked conv: forward output {tuple(out.shape)} does not fit")
        if noise is not None:
            noise = _c(noise, "noise").reshape(noise.shape[0], -1)
            if noise_weight is None or noise.shape[1] != out.shape[2] * out.shape[3] or noise.shape[0] not in (1, x.shape[0]):
                raise ValueError("masked conv: noise must be [1 or bs, 1, H, W] of the output and come with its weight")
        ctx.save_for_backward(x, wg, s, d, noise_weight, act_bias, lab, noise, out)
        ctx.act, ctx.up = bool(act), up
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        x, wg, s, d, nw, bias, lab, noise, out = ctx.saved_tensors
        need_x, need_w, need_s, need_d, need_nw, need_b = ctx.needs_input_grad[:6]
        gz, q, dbias, dnw = _mconv_scale(grad.contiguous(), out, d, lab, s.shape[1], ctx.up, want_q=d is not None and need_d, noise=noise,
                                         noise_weight=None if nw is None else nw.reshape(-1), act_bias=None if bias is None else bias.reshape(-1),
                                         act=ctx.act, want_sums=True)
        dx, ds, dw = _mconv_input_grads(gz, wg, x, s, lab, ctx.up, need_x, need_s, need_w)
        g_nw = dnw.sum().view_as(nw) if (need_nw and dnw is not None) else None
        g_b = _sum_dim(dbias, 0).view_as(bias) if (need_b and bias is not None) else None
        return dx, dw, ds, None if q is None else q / d, g_nw, g_b, None, None, None, None


def masked_styled_conv_grad(x, wg, s, d, noise_weight, act_bias, lab, noise, act, out):
    return _MaskedStyledConvGrad.apply(x, wg, s, d, noise_weight, act_bias, lab, noise, act, out)


class _StyleTablesSaved(torch.autograd.Function):
    """A layer's style tables ``(s, ws, d)`` when ``s`` and ``d`` are already known (the forward kernels computed them for the fused layer):
    only ``ws = weight / sqrt(cin k²)`` is evaluated, and the gradient w.r.t. styles, conv weight, modulation weight and bias is three
    launches of ``e4s_style_tables_bwd`` instead of ~25 small library ops per layer (a PTI step does this for 26 layers)."""

    @staticmethod
    def forward(ctx, styles, weight, mod_w, mod_b, s, d, wsq, mod_scale, mod_lr):
        cout, cin, k = weight.shape[1], weight.shape[2], weight.shape[-1]
        c = 1.0 / math.sqrt(cin * k * k)
        styles_c, s = _c(styles, "styles"), _c(s, "s")
        if s.shape != (styles.shape[0], styles.shape[1], cin) or (d is not None and (wsq is None or d.shape != s.shape[:2] + (cout,))):
            raise ValueError(f"style tables: s {tuple(s.shape)} / d do not fit styles {tuple(styles.shape)} and weight {tuple(weight.shape)}")
        ctx.save_for_backward(styles_c, _c(weight, "weight"), _c(mod_w, "modulation.weight"), s, d, wsq)
        ctx.consts = (c, float(mod_scale), float(mod_lr))
        ws = weight[0] * c
        if d is None:
            empty = s.new_empty(0)
            ctx.mark_non_differentiable(empty)
            return s.view_as(s), ws, empty
        return s.view_as(s), ws, d.view_as(d)

    @staticmethod
    def backward(ctx, gs, gws, gd):
        styles, weight, mod_w, s, d, wsq = ctx.saved_tensors
        c, ms, lr = ctx.consts
        cout, cin, k = weight.shape[1], weight.shape[2], weight.shape[-1]
        rows, sdim = styles.shape[0] * styles.shape[1], styles.shape[2]
        if d is None:
            gd = None
        have_s = gs is not None or gd is not None
        g_styles = torch.empty_like(styles) if have_s else None
        g_mod_w = torch.empty_like(mod_w) if have_s else None
        g_mod_b = torch.empty((cin,), dtype=torch.float32, device=styles.device) if have_s else None
        g_weight = torch.empty_like(weight) if (gws is not None or gd is not None) else None
        scratch = torch.empty((rows * (cout + cin),), dtype=torch.float32, device=styles.device)
        lib().call("e4s_style_tables_bwd", _p(g_styles), _p(g_mod_w), _p(g_mod_b), _p(g_weight), _p(scratch),
                   _p(None if gs is None else gs.contiguous()), _p(None if gd is None else gd.contiguous()),
                   _p(None if gws is None else gws.contiguous()), _p(styles), _p(mod_w), _p(s), _p(d), _p(weight), _p(wsq), c, ms, lr, rows, sdim,
                   cin, cout, k * k, _stream())
        return g_styles, g_weight, g_mod_w, g_mod_b, None, None, None, None, None


def style_tables_saved(styles, weight, mod_w, mod_b, s, d, wsq, mod_scale, mod_lr):
    s_out, ws, d_out = _StyleTablesSaved.apply(styles, weight, mod_w, mod_b, s, d, wsq, mod_scale, mod_lr)
    return s_out, ws, (d_out if d is not None else None)


_flip_cache = {}


def _flipped(k: torch.Tensor) -> torch.Tensor:
    """``torch.flip(k, (0, 1))`` of a constant FIR kernel (a registered buffer: ``blur.kernel`` / ``upsample.kernel``), made once per tensor version instead of once
    per layer and step (12 launches of a PTI step).  An entry lives as long as its source tensor (a captured hipGraph may hold the flipped copy's address) and is
    valid for that very tensor object at that version only."""
    import weakref
    key = id(k)
    ent = _flip_cache.get(key)
    if ent is not None and ent[0]() is k and ent[1] == k._version:
        return ent[2]
    flipped = torch.flip(k.detach(), (0, 1))
    if torch.cuda.is_current_stream_capturing():
        return flipped                       # (made inside the capture: it belongs to the graph's pool, not to this cache)
    for dead in [i for i, e in _flip_cache.items() if e[0]() is None]:
        del _flip_cache[dead]
    _flip_cache[key] = (weakref.ref(k), k._version, flipped)
    return flipped


class _ToRGBGrad(torch.autograd.Function):
    """``ToRGB.forward`` (model.py:439-479) with a known forward value: ``out = Σ_c [c(p)=c] · W · (s_c ⊙ x) + bias + upsample(skip)``
    (1x1, no demodulation; ``lab`` None = one region).  Gradients of x, the scaled weight ``w [3,cin,1,1]``, ``s [bs,nreg,cin]``, the
    bias and the skip image from the kernels of ``csrc/modconv_bwd.hip`` and the FIR kernel — the layer is not re-evaluated."""

    @staticmethod
    def forward(ctx, x, w, s, bias, skip, lab, up_kernel, out):
        x, wg, s = _c(x, "x"), _c(w, "w")[None], _c(s, "s")
        if lab is not None:
            lab = _labels_u8(lab, "labels")
            _check_mconv(x, wg, s, None, lab, 1)
        elif s.shape[1] != 1 or wg.shape[2] != x.shape[1] or s.shape[2] != x.shape[1]:
            raise ValueError("ToRGB without a label map takes one style per sample")
        ctx.save_for_backward(x, wg, s, lab, up_kernel)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.skip_shape = None if skip is None else tuple(skip.shape)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        x, wg, s, lab, up_kernel = ctx.saved_tensors
        need_x, need_w, need_s, need_b, need_skip = ctx.needs_input_grad[:5]
        g = grad.contiguous()
        gz, _, dbias, _ = _mconv_scale(g, None, None, lab, s.shape[1], 1, want_sums=True)
        dx, ds, dw = _mconv_input_grads(gz, wg, x, s, lab, 1, need_x, need_s, need_w)
        g_b = _sum_dim(dbias, 0).view(ctx.bias_shape) if (need_b and ctx.bias_shape is not None) else None
        g_skip = None
        if need_skip and ctx.skip_shape is not None:      # transpose of upfirdn2d(skip, k, up=2, pad=(2,1)) (op/upfirdn2d.py:100-105)
            g_skip = upfirdn2d_raw(g, _flipped(up_kernel), (1, 1), (2, 2), (1, 1, 1, 1)).view(ctx.skip_shape)
        return dx, None if dw is None else dw[0], ds, g_b, g_skip, None, None, None


def torgb_grad(x, w, s, bias, skip, lab, up_kernel, out):
    return _ToRGBGrad.apply(x, w, s, bias, skip, lab, up_kernel, out)


class _SingleStyledConvGrad(torch.autograd.Function):
    """A single-region ``StyledConv`` (the layers past ``remaining_layer_idx``) whose forward value ``out`` is already known: gradients
    without re-evaluating the layer.  ``wmod [bs,cout,cin,k,k]`` is the modulated (and demodulated) weight, built under autograd by the
    caller from the tiny style tensors, so this only has to return dL/dx and dL/dwmod:

        out = leaky_relu(conv(x, wmod)  [or blur(conv_transpose(x, wmod, stride 2)) for the up-sampling layers]  + nw·noise + bias) · √2

    g' = dL/dout · act'(out), Σ g', Σ g'·noise come from ``e4s_mconv_scale``; the blur's transpose is the same FIR kernel
    (``e4s_upfirdn2d``); the data gradient runs on the three-way-split MFMA conv kernel, the weight gradient on ``e4s_mconv_wgrad`` /
    ``e4s_unfold2d`` + ``e4s_gemm_sb``; shapes those do not cover raise unless ``E4S_ALLOW_MIOPEN_BWD=1`` admits ``aten.convolution_backward``."""

    @staticmethod
    def forward(ctx, x, wmod, noise_weight, act_bias, noise, act, blur, out):
        bs, cin, h, w = x.shape
        up = 1 if blur is None else 2
        if wmod.dim() != 5 or wmod.shape[0] != bs or wmod.shape[2] != cin or out.shape != (bs, wmod.shape[1], up * h, up * w):
            raise ValueError(f"single-region conv: x {tuple(x.shape)}, wmod {tuple(wmod.shape)}, out {tuple(out.shape)} do not fit")
        if noise is not None:
            noise = _c(noise, "noise").reshape(noise.shape[0], -1)
        ctx.save_for_backward(_c(x, "x"), wmod, noise_weight, act_bias, noise, blur, _c(out, "out"))
        ctx.act = bool(act)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        x, wmod, nw, bias, noise, blur, out = ctx.saved_tensors
        need_x, need_w, need_nw, need_b = ctx.needs_input_grad[:4]
        bs, cin, h, w = x.shape
        cout, k = wmod.shape[1], wmod.shape[-1]
        gz, _, dbias, dnw = _mconv_scale(grad.contiguous(), out, None, None, 1, 1, noise=noise, noise_weight=None if nw is None else nw.reshape(-1),
                                         act_bias=None if bias is None else bias.reshape(-1), act=ctx.act, want_sums=True)
        g = gz.view(bs, cout, out.shape[2], out.shape[3])
        xin = x.view(1, bs * cin, h, w)
        conv_bwd = torch.ops.aten.convolution_backward
        dx = dw = None
        if blur is not None:
            # out = fir(conv_transpose(x), pad (1,1)): the FIR's transpose is the FIR with the flipped kernel and pad (2,2)
            g = upfirdn2d_raw(g.view(bs * cout, 1, out.shape[2], out.shape[3]), _flipped(blur), (1, 1), (1, 1), (2, 2, 2, 2))
            g = g.view(bs, cout, 2 * h + 1, 2 * w + 1)
        if need_x:
            # data gradient on the MFMA conv kernel (three-way bf16 split: fp32-class), one sample at a time (its weights are per sample):
            # a 3x3 correlation of g with the transposed + flipped weight, or — for the transposed conv — a stride-2 correlation of g
            wd = wmod.detach().transpose(1, 2)                                     # [bs, cin, cout, k, k]
            if blur is None:
                wd = wd.flip(3, 4)
            if cout >= 16 and k in (1, 3):
                parts = [conv2d(g[b:b + 1], PreparedConv(exact=ops.DGRAD_SINGLE_SPLIT).get(wd[b].contiguous()), 1 if blur is None else 2,
                                k // 2 if blur is None else 0) for b in range(bs)]
                dx = parts[0] if bs == 1 else torch.cat(parts)      # (batch 1 is the PTI case: no 134 MB copy of the 1024^2 gradient)
        if need_w and k in (1, 3):
            # weight gradient as an implicit GEMM (e4s_mconv_wgrad; odd widths: one unfold + e4s_gemm_sb per sample group): dW[o,(i,k)] = Σ_p g'[o,p] · x[i,p+k-pad], or for the
            # transposed conv dWt[i,(o,k)] = Σ_q x[i,q] · gT[o,2q+k]
            if blur is None and w % 16 == 0:
                dw = mconv_wgrad(g.reshape(1, bs, cout, h * w), x, None, None, cout, k).view(bs, cout, cin, k, k)
            elif blur is None:
                dw = _gemm_nt(g.reshape(bs, cout, h * w), unfold2d(x, k, 1, k // 2, h, w)).view(bs, cout, cin, k, k)
            else:
                dw = _gemm_nt(x.reshape(bs, cin, h * w), unfold2d(g, k, 2, 0, h, w)).view(bs, cin, cout, k, k).transpose(1, 2)
        want_dx, want_dw = need_x and dx is None, need_w and dw is None
        if want_dx or want_dw:                                  # shapes the kernels above do not cover
            if not ops.ALLOW_LIBRARY_BWD:
                raise NotImplementedError(f"single-region conv backward: no native kernel for cout {cout}, kernel size {k}"
                                          f"{' (up)' if blur is not None else ''}; set E4S_ALLOW_MIOPEN_BWD=1 to let MIOpen compute it")
            g1 = g.reshape(1, bs * cout, g.shape[2], g.shape[3])
            if blur is None:
                dxm, dwm, _ = conv_bwd(g1, xin, wmod.reshape(bs * cout, cin, k, k), None, [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], bs,
                                       [bool(want_dx), bool(want_dw), False])
                if dwm is not None:
                    dwm = dwm.view(bs, cout, cin, k, k)
            else:
                dxm, dwm, _ = conv_bwd(g1, xin, wmod.transpose(1, 2).reshape(bs * cin, cout, k, k), None, [2, 2], [0, 0], [1, 1], True, [0, 0], bs,
                                       [bool(want_dx), bool(want_dw), False])
                if dwm is not None:
                    dwm = dwm.view(bs, cin, cout, k, k).transpose(1, 2)
            dx = dxm if want_dx else dx
            dw = dwm if want_dw else dw
        g_nw = dnw.sum().view_as(nw) if (need_nw and dnw is not None) else None
        g_b = _sum_dim(dbias, 0).view_as(bias) if (need_b and bias is not None) else None
        return None if dx is None else dx.view_as(x), dw, g_nw, g_b, None, None, None, None


def single_styled_conv_grad(x, wmod, noise_weight, act_bias, noise, act, blur, out):
    return _SingleStyledConvGrad.apply(x, wmod, noise_weight, act_bias, noise, act, blur, out)


class _LocalMLPsGrad(torch.autograd.Function):
    """The per-region LocalMLP stack (networks.py:23-49, 236-244) with known forward values: ``h = lrelu(scale0 W0 x + lr0 b0)`` and
    ``out = scale2 W2 h + lr2 b2 (+ addend)`` came from two ``grouped_linear`` launches; the gradients of x and of the 4 n parameters come from
    ``e4s_grouped_linear_bwd`` (outer products for the weights, a split transposed mat-vec for the inputs): no stacking of the 12 x 13.6 MB
    weights, no library GEMM, no re-evaluation."""

    @staticmethod
    def forward(ctx, x, out, h, scale0, scale2, lr0, lr2, slope, *params):
        n = len(params) // 4
        ctx.save_for_backward(x, h, *params[:n], *params[2 * n:3 * n])          # x, h, W0 (n), W2 (n)
        ctx.consts = (n, float(scale0), float(scale2), float(lr0), float(lr2), float(slope))
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        n, scale0, scale2, lr0, lr2, slope = ctx.consts
        x, h = ctx.saved_tensors[0], ctx.saved_tensors[1]
        w0, w2 = ctx.saved_tensors[2:2 + n], ctx.saved_tensors[2 + n:2 + 2 * n]
        bs, _, in0 = x.shape
        hid, out2 = h.shape[2], g.shape[2]
        g = g.contiguous()
        xc, hc = x.contiguous(), h.contiguous()
        dev = g.device
        PtrArr = ctypes.c_void_p * n
        # layer 2: dW2, db2, and dL/d(pre-activation of layer 0) = scale2 W2^T g * lrelu'(h)
        dW2 = torch.empty((n, out2, hid), dtype=torch.float32, device=dev)
        db2 = torch.empty((n, out2), dtype=torch.float32, device=dev)
        gy0 = torch.empty((bs, n, hid), dtype=torch.float32, device=dev)
        os2 = 32
        scratch = torch.empty((os2 * bs * n * max(hid, in0),), dtype=torch.float32, device=dev)
        lib().call("e4s_grouped_linear_bwd", _p(dW2), _p(db2), _p(gy0), _p(scratch), _p(g), _p(hc), PtrArr(*[w.data_ptr() for w in w2]), _p(hc),
                   scale2, lr2, slope, bs, n, hid, out2, os2, _stream())
        # layer 0: dW0, db0, dx
        dW0 = torch.empty((n, hid, in0), dtype=torch.float32, device=dev)
        db0 = torch.empty((n, hid), dtype=torch.float32, device=dev)
        dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        lib().call("e4s_grouped_linear_bwd", _p(dW0), _p(db0), _p(dx), _p(scratch), _p(gy0), _p(xc), PtrArr(*[w.data_ptr() for w in w0]), None,
                   scale0, lr0, slope, bs, n, in0, hid, 8, _stream())
        return (dx, None, None, None, None, None, None, None) + tuple(dW0.unbind(0)) + tuple(db0.unbind(0)) + tuple(dW2.unbind(0)) + tuple(db2.unbind(0))


class _SmallMap(torch.autograd.Function):
    """``out[j, ...] = sum_k T[j, k] * w[..., k]`` for a constant ``T [J, K]`` (J, K <= 36) — the parity composition of an up layer's weight
    (``torch_ref._parity_weights``) — and its gradient, one launch each (``e4s_small_map``).  ``grouped`` (J = 36, K = 9): the result is laid out
    ``[4, ..., 9]`` (``out[g, ..., t]`` for ``j = 9 g + t``), the four parity weights in the layout their consumers read."""

    @staticmethod
    def forward(ctx, w, T, grouped):
        w, T = _c(w, "w"), _c(T, "T")
        J, K = T.shape
        if w.shape[-1] != K:
            raise ValueError(f"small_map: last dimension {w.shape[-1]} != {K}")
        if grouped and (J, K) != (36, 9):
            raise ValueError("small_map: the grouped layout is built for T [36, 9]")
        n = w.numel() // K
        shape = (J // 9,) + tuple(w.shape[:-1]) + (9,) if grouped else (J,) + tuple(w.shape[:-1])
        out = torch.empty(shape, dtype=torch.float32, device=w.device)
        lib().call("e4s_small_map", _p(out), _p(T), _p(w), J, K, n, 0, int(grouped), _stream())
        ctx.save_for_backward(T)
        ctx.wshape, ctx.grouped = tuple(w.shape), bool(grouped)
        return out

    @staticmethod
    def backward(ctx, g):
        (T,) = ctx.saved_tensors
        J, K = T.shape
        g = g.contiguous()
        dw = torch.empty(ctx.wshape, dtype=torch.float32, device=g.device)
        lib().call("e4s_small_map", _p(dw), _p(T), _p(g), J, K, dw.numel() // K, 1, int(ctx.grouped), _stream())
        return dw, None, None


def small_map(w, T, grouped: bool = False):
    return _SmallMap.apply(w, T, grouped)


class _EqualLinearGrad(torch.autograd.Function):
    """``scale * x @ W^T + lr_mul * bias`` (EqualLinear without activation, model.py:154-162) for ``x [bs <= 8, in]`` with the grouped-linear
    kernels in both directions (one group): the modulation vectors of the single-region layers under autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, lr_mul):
        xc = _c(x, "x")
        out = grouped_linear(xc[:, None, :], [weight], [bias], scale=scale, bias_mul=lr_mul, act=0)[:, 0]
        ctx.save_for_backward(xc, weight)
        ctx.consts = (float(scale), float(lr_mul))
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        scale, lr = ctx.consts
        bs, in_dim = x.shape
        out_dim = w.shape[0]
        g = g.contiguous()
        dW = torch.empty((1, out_dim, in_dim), dtype=torch.float32, device=g.device)
        db = torch.empty((1, out_dim), dtype=torch.float32, device=g.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        osplit = 8
        scratch = torch.empty((osplit * bs * in_dim,), dtype=torch.float32, device=g.device)
        lib().call("e4s_grouped_linear_bwd", _p(dW), _p(db), _p(dx), _p(scratch), _p(g), _p(x), (ctypes.c_void_p * 1)(_c(w, "weight").data_ptr()), None,
                   scale, lr, 0.0, bs, 1, in_dim, out_dim, osplit, _stream())
        return dx, dW[0], db[0], None, None


def equal_linear_grad(x, weight, bias, scale, lr_mul):
    return _EqualLinearGrad.apply(x, weight, bias, scale, lr_mul)


def local_mlps_grad(x, out, h, w0, b0, w2, b2, scale0, scale2, lr0, lr2, slope):
    return _LocalMLPsGrad.apply(x, out, h, scale0, scale2, lr0, lr2, slope, *w0, *b0, *w2, *b2)


__all__ = ['_mconv_unfold', '_mconv_scale', '_sum_dim', 'gemm_sb', '_gemm_nt', 'unfold2d', '_mconv_input_grads', 'mconv_wgrad', '_check_mconv', '_MaskedConvCore', 'masked_conv_core', '_MaskedStyledConvGrad', 'masked_styled_conv_grad', '_StyleTablesSaved', 'style_tables_saved', '_ToRGBGrad', 'torgb_grad', '_SingleStyledConvGrad', 'single_styled_conv_grad', '_LocalMLPsGrad', '_SmallMap', 'small_map', '_EqualLinearGrad', 'equal_linear_grad', 'local_mlps_grad']
