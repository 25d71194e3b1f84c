"""Differentiable PyTorch forms of the fused synthesis layers: the GLUE of the backward pass, not its arithmetic.

SURVEY §8 row f1.  A backward through the drop-in modules (PTI tuning, ``training/video_swap_ft_coach.py:242-299``; W-optimisation,
``optimization.py:321-349``) differentiates each layer from the fused forward kernel's own output: the expressions below are evaluated under
autograd on the saved inputs, and every contraction in them is one of this library's kernels (``ops.masked_conv_core``, ``ops.mconv_wgrad``,
``ops.gemm_sb``, ``ops.small_map``, ``ops.local_mlps_grad``, ``ops.equal_linear_grad`` ...) — with the default ``E4S_NATIVE_BWD=1`` no rocBLAS /
MIOpen kernel runs in a PTI step (asserted under torch.profiler by ``tests/test_gpu_backward.py``).  ``E4S_NATIVE_BWD=0`` selects the stock ATen
forms kept next to them: the comparison arm of the gradient tests, never the default.  Each function states the reference lines it evaluates;
none of this runs under ``torch.no_grad()`` inference.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def fir_resample(x: torch.Tensor, kernel: torch.Tensor, up: int = 1, pad=(0, 0)) -> torch.Tensor:
    """``upfirdn2d(x, kernel, up=up, down=1, pad=pad)`` (op/upfirdn2d.py:85-147) with ATen ops: zero-insert, pad, correlate with the
    flipped kernel, every channel on its own."""
    bs, ch, h, w = x.shape
    if up > 1:
        z = x.new_zeros(bs, ch, h, up, w, up)
        z[:, :, :, 0, :, 0] = x
        x = z.reshape(bs, ch, h * up, w * up)
    p0, p1 = pad
    x = F.pad(x, (max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)))
    if p0 < 0 or p1 < 0:
        x = x[:, :, max(-p0, 0): x.shape[2] - max(-p1, 0), max(-p0, 0): x.shape[3] - max(-p1, 0)]
    k = torch.flip(kernel, (0, 1)).to(x.dtype)[None, None].expand(ch, 1, -1, -1)
    return F.conv2d(x, k, groups=ch)


def equal_linear(x, weight, bias, scale: float, lr_mul: float, activation: bool):
    """``EqualLinear.forward`` (model.py:150-161)."""
    if activation:
        out = F.linear(x, weight * scale)
        return F.leaky_relu(out + (bias * lr_mul if bias is not None else 0.0), 0.2) * SQRT2
    return F.linear(x, weight * scale, None if bias is None else bias * lr_mul)


def local_mlps(x, w0: Sequence[torch.Tensor], b0, w2, b2, scale0: float, scale2: float, lr0: float, lr2: float, slope: float, addend=None):
    """The per-region ``LocalMLP`` stack of ``Net3.cal_style_codes`` (networks.py:23-49, 236-244): ``x [bs, n, dim]`` -> ``[bs, n, out]``."""
    n = len(w0)
    if n > 1 and all(w.shape == w0[0].shape for w in w0) and all(w.shape == w2[0].shape for w in w2):
        # same-shaped groups: two batched GEMMs instead of 2n small ones (the backward of a PTI step is bound by launch count)
        xg = x.transpose(0, 1)                                                               # [n, bs, dim]
        # (the equalised-lr scales ride in the GEMMs' alpha: no extra pass over the 160 MB of stacked weights)
        h = F.leaky_relu(torch.baddbmm((torch.stack(tuple(b0)) * lr0)[:, None, :], xg, torch.stack(tuple(w0)).transpose(1, 2), alpha=scale0), slope)
        bias2 = torch.stack(tuple(b2)) * lr2
        if addend is not None:
            bias2 = bias2 + addend
        return torch.baddbmm(bias2[:, None, :], h, torch.stack(tuple(w2)).transpose(1, 2), alpha=scale2).transpose(0, 1)
    outs = []
    for g in range(n):
        h = F.leaky_relu(F.linear(x[:, g], w0[g] * scale0, b0[g] * lr0), slope)
        o = F.linear(h, w2[g] * scale2, b2[g] * lr2)
        outs.append(o if addend is None else o + addend)
    return torch.stack(outs, dim=1)


def _modulated(x, s, weight, demodulate: bool, upsample: bool, blur: Optional[torch.Tensor], eps: float = 1e-8):
    """One region's ``ModulatedConv2d`` fused branch (model.py:276-320): ``s [bs, cin]`` is the modulation output."""
    bs, cin, h, w = x.shape
    cout, k = weight.shape[1], weight.shape[-1]
    wm = (weight * (1.0 / math.sqrt(cin * k * k))) * s.view(bs, 1, cin, 1, 1)
    if demodulate:
        wm = wm * torch.rsqrt(wm.pow(2).sum((2, 3, 4), keepdim=True) + eps)
    xin = x.reshape(1, bs * cin, h, w)
    if upsample:
        y = F.conv_transpose2d(xin, wm.transpose(1, 2).reshape(bs * cin, cout, k, k), stride=2, padding=0, groups=bs)
        return fir_resample(y.view(bs, cout, y.shape[-2], y.shape[-1]), blur, pad=(1, 1))
    y = F.conv2d(xin, wm.reshape(bs * cout, cin, k, k), padding=k // 2, groups=bs)
    return y.view(bs, cout, h, w)


_SHIFT_CACHE = {}


def _blur_shift(blur, dtype):
    """shift[m,n,ky,kx] = blur[m-ky][n-kx] (6x6x3x3) and its parity-gathered form T[g=2a+b,dy,dx,ky,kx] = shift[4-2dy+a, 4-2dx+b, ky, kx];
    constants of the layer's blur kernel, built once per (buffer, dtype)."""
    key = (blur.data_ptr(), blur._version, blur.device, dtype)
    hit = _SHIFT_CACHE.get(key)
    if hit is None:
        shift = torch.zeros(6, 6, 3, 3, dtype=torch.float64)
        b = blur.detach().double().cpu()
        for ky in range(3):
            for kx in range(3):
                shift[ky:ky + 4, kx:kx + 4, ky, kx] = b
        par = torch.stack([shift[a::2, b_::2].flip(0, 1) for a in (0, 1) for b_ in (0, 1)])            # [4, 3(dy), 3(dx), 3, 3]
        hit = (shift.to(blur.device, dtype), par.to(blur.device, dtype))
        if len(_SHIFT_CACHE) > 64:
            _SHIFT_CACHE.clear()
        _SHIFT_CACHE[key] = hit
    return hit


def _composed_up_weights(ws, blur, dtype):
    """C2[o,i,m,n] = Σ_{ky,kx} blur[m-ky][n-kx] · Ws[o,i,ky,kx]: the full 2-D convolution of the 3x3 weight with the 4x4 blur (6x6)."""
    return torch.einsum("mnkl,oikl->oimn", _blur_shift(blur, dtype)[0], ws)


def _parity_weights(ws, blur, dtype):
    """The composed 3x3 weight of each output parity g = 2a+b of an up-sampling layer, ``[4, cout, cin, 3, 3]``:
    ``W_g[dy,dx] = C2[4-2dy+a][4-2dx+b]`` (= ``C2[:, :, a::2, b::2].flip(2, 3)``), as one contraction with a cached constant."""
    par = _blur_shift(blur, dtype)[1]
    if ws.is_cuda and ws.dtype == torch.float32:
        from . import ops
        if ops.NATIVE_BWD:      # one launch (and one for its gradient) instead of a library GEMM of 36 x 9 x (cout cin)
            co, ci = ws.shape[:2]
            return ops.small_map(ws.reshape(co, ci, 9), par.reshape(36, 9), grouped=True).view(4, co, ci, 3, 3)    # written in this layout: no copy
    return torch.einsum("gyxkl,oikl->goiyx", par, ws)


def _region_sum(x, styles, labels, weight, mod_w, mod_b, mod_scale, mod_lr, demodulate, upsample, blur):
    """Σ_c modconv(x, style_c) ⊙ [label == c]  (StyledConv.forward :389-398 / ToRGB.forward :447-454).  ``labels`` None = one region
    (plain modulated conv).  With a region map the sum is evaluated in its ONE-PASS form so that autograd sees one
    GEMM per layer instead of twelve convolutions: ``out[b,o,p] = d[b,c(p),o] · Σ_{i,k} Ws[o,i,k] · s[b,c(p),i] · x[b,i,p+k]``."""
    if labels is None:
        s = F.linear(styles[:, 0], mod_w * mod_scale, mod_b * mod_lr)
        return _modulated(x, s, weight, demodulate, upsample, blur)
    bs, cin, h, w = x.shape
    cout, k = weight.shape[1], weight.shape[-1]
    nreg = styles.shape[1]
    s, ws, d = _tables(x, styles, weight, mod_w, mod_b, mod_scale, mod_lr, demodulate)
    ho, wo = (2 * h, 2 * w) if upsample else (h, w)
    lab = _labels_at(labels, ho, wo)
    if x.is_cuda:
        from . import ops
        if ops.NATIVE_BWD:                                                        # §8 f1: HIP gradient kernels + the split-bf16 MFMA GEMM
            lab8 = _labels_at(labels, ho, wo, as_u8=True)
            if not upsample:
                return ops.masked_conv_core(x, ws, s, d, lab8)
            wg = _parity_weights(ws, blur, x.dtype)
            out = x.new_zeros(bs, cout, ho, wo)
            for a in (0, 1):
                for b in (0, 1):
                    out[:, :, a::2, b::2] = ops.masked_conv_core(x, wg[2 * a + b], s, d, lab8[:, a::2, b::2])
            return out
    lab = lab.long()
    valid = (lab < nreg)
    lab = lab.clamp(max=nreg - 1)

    def per_pixel(table, lab_flat):          # table [bs, nreg, C] -> [bs, C, P] rows picked by each pixel's region
        return torch.gather(table.transpose(1, 2), 2, lab_flat[:, None, :].expand(-1, table.shape[2], -1))

    xu = F.unfold(x, k, padding=k // 2).view(bs, cin, k * k, h * w)               # x[b, i, p + tap]
    if not upsample:
        lf = lab.reshape(bs, -1)
        y = torch.matmul(ws.reshape(cout, cin * k * k), (xu * per_pixel(s, lf)[:, :, None, :]).reshape(bs, cin * k * k, h * w))
        if d is not None:
            y = y * per_pixel(d, lf)
        return (y * valid.reshape(bs, 1, -1).to(y.dtype)).view(bs, cout, h, w)
    # stride-2 transposed conv followed by the 4x4 blur = four 3x3 correlations over the input grid, one per output parity (a, b):
    # Weff[a,b][dy,dx] = C2[2-2dy+a][2-2dx+b] with C2 the full 2-D convolution of the 3x3 weight with the blur kernel (6x6).
    # (as one small GEMM: C2[o,i,m,n] = Σ_{ky,kx} blur[m-ky][n-kx] · Ws[o,i,ky,kx])
    c2 = _composed_up_weights(ws, blur, x.dtype)
    out = x.new_zeros(bs, cout, ho, wo)
    for a in (0, 1):
        for b in (0, 1):
            weff = c2[:, :, a::2, b::2].flip(2, 3)                                  # rows (4+a, 2+a, a), columns likewise: [cout, cin, 3(dy), 3(dx)]
            lf = lab[:, a::2, b::2].reshape(bs, -1)
            y = torch.matmul(weff.reshape(cout, cin * 9), (xu * per_pixel(s, lf)[:, :, None, :]).reshape(bs, cin * 9, h * w))
            if d is not None:
                y = y * per_pixel(d, lf)
            y = y * valid[:, a::2, b::2].reshape(bs, 1, -1).to(y.dtype)
            out[:, :, a::2, b::2] = y.view(bs, cout, h, w)
    return out


def _tables_autograd(x, styles, weight, mod_w, mod_b, mod_scale, mod_lr, demodulate):
    """(s [bs,nreg,cin], scaled weight [cout,cin,k,k], d [bs,nreg,cout] | None) of the one-pass form, op by op under autograd."""
    cin, k = x.shape[1], weight.shape[-1]
    s = F.linear(styles, mod_w * mod_scale, mod_b * mod_lr)
    ws = weight[0] * (1.0 / math.sqrt(cin * k * k))
    d = torch.rsqrt(torch.einsum("bri,oi->bro", s * s, (ws * ws).sum((2, 3))) + 1e-8) if demodulate else None      # :280
    return s, ws, d


class _StyleTables(torch.autograd.Function):
    """``_tables_autograd`` with its gradient written out (half the launches of the op-by-op graph; a PTI step evaluates it for 26 layers):

        s = styles · (mod_w·scale)ᵀ + mod_b·lr        ws = weight / sqrt(cin k²)        d = rsqrt(s² · wsqᵀ + 1e-8),  wsq = Σ_k ws²
    """

    @staticmethod
    def forward(ctx, styles, weight, mod_w, mod_b, mod_scale, mod_lr, demodulate):
        cin, k = weight.shape[2], weight.shape[-1]
        c = 1.0 / math.sqrt(cin * k * k)
        s = F.linear(styles, mod_w * mod_scale, mod_b * mod_lr)
        ws = weight[0] * c
        wsq = d = None
        if demodulate:
            wsq = (ws * ws).sum((2, 3))
            d = torch.rsqrt(torch.matmul(s * s, wsq.t()) + 1e-8)
        ctx.save_for_backward(styles, mod_w, s, ws, wsq, d)
        ctx.consts = (c, mod_scale, mod_lr)
        if d is None:
            ctx.mark_non_differentiable(empty := s.new_empty(0))
            return s, ws, empty
        return s, ws, d

    @staticmethod
    def backward(ctx, gs, gws, gd):
        styles, mod_w, s, ws, wsq, d = ctx.saved_tensors
        c, scale, lr = ctx.consts
        if d is not None and gd is not None:
            t = gd * d * d * d * (-0.5)                                               # dL/d(s²·wsqᵀ)
            back = (torch.matmul(t, wsq) * s) * 2.0
            gs = back if gs is None else gs + back
            gwsq = torch.matmul(t.reshape(-1, t.shape[-1]).t(), (s * s).reshape(-1, s.shape[-1]))      # [cout, cin]
            back_w = ws * gwsq[:, :, None, None] * 2.0
            gws = back_w if gws is None else gws + back_w
        g_weight = None if gws is None else (gws * c)[None]
        g_styles = g_mod_w = g_mod_b = None
        if gs is not None:
            g2 = gs.reshape(-1, gs.shape[-1])
            g_styles = (torch.matmul(g2, mod_w) * scale).view_as(styles)
            g_mod_w = torch.matmul(g2.t(), styles.reshape(-1, styles.shape[-1])) * scale
            g_mod_b = g2.sum(0) * lr
        return g_styles, g_weight, g_mod_w, g_mod_b, None, None, None


def _tables(x, styles, weight, mod_w, mod_b, mod_scale, mod_lr, demodulate, saved=None):
    """(s [bs,nreg,cin], scaled weight [cout,cin,k,k], d [bs,nreg,cout] | None) of the one-pass form.  ``saved = (s, d, wsq)``: the
    tables the forward kernels already computed for this layer (device only) — then nothing is re-evaluated and the gradient is
    ``e4s_style_tables_bwd``."""
    if saved is not None and x.is_cuda:
        from . import ops
        if ops.NATIVE_BWD:
            return ops.style_tables_saved(styles, weight, mod_w, mod_b, saved[0], saved[1] if demodulate else None, saved[2] if demodulate else None,
                                          mod_scale, mod_lr)
    s, ws, d = _StyleTables.apply(styles, weight, mod_w, mod_b, mod_scale, mod_lr, bool(demodulate))
    return s, ws, (d if demodulate else None)


def _labels_at(labels, ho, wo, as_u8: bool = False):
    """Nearest resize of the region map (:391).  The resized maps are kept on the tensor object: every layer of a pass shares the same
    ``labels`` tensor, and several layers share a resolution."""
    if labels.shape[-2:] == (ho, wo) and not (as_u8 and labels.dtype != torch.uint8):
        return labels
    cache = getattr(labels, "_e4s_resized", None)
    if cache is None:
        cache = {}
        try:
            labels._e4s_resized = cache
        except AttributeError:
            pass
    # an entry made while a hipGraph is being captured belongs to that graph (and one made before must not be baked into it)
    key = (ho, wo, as_u8, labels._version, labels.is_cuda and torch.cuda.is_current_stream_capturing())
    if len(cache) > 32:
        cache.clear()
    out = cache.get(key)
    if out is None:
        out = labels
        if labels.shape[-2:] != (ho, wo):
            out = F.interpolate(labels[:, None].float(), size=(ho, wo), mode="nearest")[:, 0]
        if as_u8:
            out = out.to(torch.uint8)
        cache[key] = out
    return out


def styled_conv(x, styles, weight, mod_w, mod_b, noise_weight, act_bias, *, labels, noise, act: bool, upsample: bool, blur, demodulate: bool,
                mod_scale: float, mod_lr: float, fwd_out=None, tables=None):
    """``StyledConv.forward`` (model.py:382-423): region sum + noise injection (:335) + FusedLeakyReLU (:421).

    With ``fwd_out`` (the value the fused forward kernel produced) a masked layer on the device is not re-evaluated at all: only the small
    (s, d, weight) tables are rebuilt under autograd and the layer's gradients come from ``ops.masked_styled_conv_grad`` (§8 f1)."""
    if fwd_out is not None and labels is not None and x.is_cuda:
        from . import ops
        if ops.NATIVE_BWD:
            s, ws, d = _tables(x, styles, weight, mod_w, mod_b, mod_scale, mod_lr, demodulate, saved=tables)
            up = 2 if upsample else 1
            lab8 = _labels_at(labels, up * x.shape[2], up * x.shape[3], as_u8=True)
            wg = _parity_weights(ws, blur, x.dtype) if upsample else ws[None]
            use_noise = noise is not None and noise_weight is not None
            return ops.masked_styled_conv_grad(x, wg, s, d, noise_weight if use_noise else None, act_bias, lab8, noise if use_noise else None, act,
                                               fwd_out)
    if fwd_out is not None and labels is None and x.is_cuda:
        from . import ops
        if ops.NATIVE_BWD:
            bs, cin, k = x.shape[0], x.shape[1], weight.shape[-1]
            if bs <= 8 and styles.shape[-1] % 4 == 0 and mod_b is not None:
                s = ops.equal_linear_grad(styles[:, 0], mod_w, mod_b, mod_scale, mod_lr)
            else:
                s = F.linear(styles[:, 0], mod_w * mod_scale, mod_b * mod_lr)
            wm = (weight * (1.0 / math.sqrt(cin * k * k))) * s.view(bs, 1, cin, 1, 1)                    # model.py:276-281
            if demodulate:
                wm = wm * torch.rsqrt(wm.pow(2).sum((2, 3, 4), keepdim=True) + 1e-8)
            use_noise = noise is not None and noise_weight is not None
            return ops.single_styled_conv_grad(x, wm, noise_weight if use_noise else None, act_bias, noise if use_noise else None, act,
                                               blur if upsample else None, fwd_out)
    out = _region_sum(x, styles, labels, weight, mod_w, mod_b, mod_scale, mod_lr, demodulate, upsample, blur)
    if noise is not None and noise_weight is not None:
        out = out + noise_weight * noise
    if act_bias is not None:
        out = out + act_bias.view(1, -1, 1, 1)
    if act:
        out = F.leaky_relu(out, 0.2) * SQRT2
    return out


def to_rgb(x, styles, skip, weight, mod_w, mod_b, bias, *, labels, up_kernel, mod_scale: float, mod_lr: float, fwd_out=None, tables=None):
    """``ToRGB.forward`` (model.py:439-479): 1x1 modulated conv without demodulation, + bias, + upsampled skip (Upsample :34-53).
    With ``fwd_out`` (the fused kernel's value) on the device: gradients from ``ops.torgb_grad`` without re-evaluating the layer."""
    if fwd_out is not None and x.is_cuda:
        from . import ops
        if ops.NATIVE_BWD and (skip is None or (tuple(up_kernel.shape) == (4, 4) and skip.shape[-1] * 2 == x.shape[-1])):
            s, ws, _ = _tables(x, styles, weight, mod_w, mod_b, mod_scale, mod_lr, False, saved=tables)
            lab8 = None if labels is None else _labels_at(labels, x.shape[2], x.shape[3], as_u8=True)
            return ops.torgb_grad(x, ws, s, bias, skip, lab8, up_kernel, fwd_out)
    out = _region_sum(x, styles, labels, weight, mod_w, mod_b, mod_scale, mod_lr, False, False, None)
    out = out + bias
    if skip is not None:
        out = out + fir_resample(skip, up_kernel, up=2, pad=(2, 1))
    return out
