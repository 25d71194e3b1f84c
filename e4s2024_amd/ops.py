"""Tensor-level wrappers over the C ABI (``include/e4s_hip.h``).

PyTorch is plumbing here: it owns device memory (``torch.empty`` for outputs) and the stream
(``torch.cuda.current_stream()`` is handed to every launch).  All arithmetic of the hot path runs in
``libe4s_hip.so``.  Tensors must be fp32 CUDA tensors; anything else raises (no fallback).

One module per stage of the path (split in round 5; ``ops`` re-exports all of them, so ``ops.<name>`` is the one spelling callers use):

* ``ops``         rows a1 - a7: fused bias / activation, upfirdn2d, region maps, prepared weights, the f16 range guard, style / demodulation tables,
                  the masked and single-region modulated convolutions, ToRGB, the per-region MLPs' grouped linear, the kernel timing hook —
                  and EVERY switch of every stage (module attributes and their environment variables: tests and tools set them here)
* ``ops_encode``  rows a8 - a10: the regional-style encoder's and the face parser's operators
* ``ops_post``    rows f2 / f3: mask surgery, paste-back masks, Pillow's resize, multi-band blend
* ``ops_grad``    row f1: the native gradients of the synthesis path
"""
from __future__ import annotations

import ctypes
import math
import os
import threading
import weakref
from typing import Optional, Sequence, Tuple

import torch

from ._lib import lib

SQRT2 = 2.0 ** 0.5
MAX_REGIONS = 16
LABEL_NONE = 255


# ----------------------------------------------------------------------------- helpers
def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # same failure class as the reference's TORCH_CHECK(x.is_cuda())
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype} (the MI355X path computes in fp32)")
    return t


def _c(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    return _req(t, name, dtype).contiguous()


class _ForwardOnly(torch.autograd.Function):
    """Keeps a fused forward-only kernel in the autograd graph so that a backward through it fails loudly
    instead of silently producing no gradient (backward kernels are the next row)."""

    @staticmethod
    def forward(ctx, name, out, *deps):
        ctx.name = name
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        raise NotImplementedError(
            f"backward of {ctx.name} is not built yet (SURVEY §8 f1); run the fused synthesis path under torch.no_grad()")


class _TorchBackward(torch.autograd.Function):
    """Forward: the fused HIP kernel's result.  Backward: re-evaluates ``ref(*deps)`` — the same layer written with stock PyTorch
    ops (``torch_ref.py``) — under autograd on the saved inputs and back-propagates ``grad`` through it (interim:
    native forward, PyTorch gradients)."""

    @staticmethod
    def forward(ctx, ref, out, *deps):
        ctx.ref = ref
        ctx.is_tensor = [isinstance(d, torch.Tensor) for d in deps]
        ctx.with_out = bool(getattr(ref, "takes_fwd_out", False))       # a ref that differentiates from the forward value (no re-evaluation)
        ctx.save_for_backward(*[d for d in deps if isinstance(d, torch.Tensor)], *([out] if ctx.with_out else []))
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        tensors = ctx.saved_tensors
        extra = {"fwd_out": tensors[-1]} if ctx.with_out else {}
        saved = iter(tensors)
        needs = ctx.needs_input_grad[2:]
        ins, wanted = [], []
        for flag, need in zip(ctx.is_tensor, needs):
            if not flag:
                ins.append(None)
                continue
            t = next(saved).detach()
            if need:
                t = t.requires_grad_(True)
                wanted.append(t)
            ins.append(t)
        ev = _timed("backward:" + getattr(ctx.ref, "__qualname__", "ref").split(".")[0] + f"{tuple(grad.shape[1:])}")
        with torch.enable_grad():
            ref_out = ctx.ref(*ins, **extra)
            grads = torch.autograd.grad(ref_out, wanted, grad, allow_unused=True)
        if ev is not None:
            ev.record()
        it = iter(grads)
        res = [next(it) if (flag and need) else None for flag, need in zip(ctx.is_tensor, needs)]
        return (None, None, *res)


def _attach(name: str, out: torch.Tensor, *deps: Optional[torch.Tensor], ref=None) -> torch.Tensor:
    """Put a fused kernel's output into the autograd graph of its inputs.  With ``ref`` (a differentiable PyTorch form of the same
    computation taking ``*deps``) a backward pass works through ``_TorchBackward``; without, it fails loudly."""
    if torch.is_grad_enabled() and any(d is not None and d.requires_grad for d in deps):
        if ref is not None:
            return _TorchBackward.apply(ref, out, *deps)
        return _ForwardOnly.apply(name, out, *[d for d in deps if d is not None and d.requires_grad])
    return out


# ------------------------------------------------------------------------------------ a1
def fused_bias_act(x: torch.Tensor, bias: Optional[torch.Tensor], ref: Optional[torch.Tensor], act: int, grad: int,
                   alpha: float, scale: float) -> torch.Tensor:
    """``fused.fused_bias_act`` of the reference (models/stylegan2/op/fused_bias_act.cpp:11-21)."""
    x = _c(x, "input")
    b = _c(bias, "bias") if bias is not None and bias.numel() else None
    r = _c(ref, "refer") if ref is not None and ref.numel() else None
    out = torch.empty_like(x)
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    lib().call("e4s_fused_bias_act", _p(out), _p(x), _p(b), _p(r), act, grad, float(alpha), float(scale), x.numel(), step_b,
               0 if b is None else b.numel(), _stream())
    return out


class _FusedLeakyReLUBackward(torch.autograd.Function):
    # models/stylegan2/op/fused_act.py:18-47
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        grad_input = fused_bias_act(grad_output, None, out, 3, 1, negative_slope, scale)
        dim = [0] + list(range(2, grad_input.ndim))
        return grad_input, grad_input.sum(dim).detach()

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        return fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale), None, None, None


class _FusedLeakyReLU(torch.autograd.Function):
    # models/stylegan2/op/fused_act.py:50-69
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        gi, gb = _FusedLeakyReLUBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale)
        return gi, gb, None, None


def fused_leaky_relu(input: torch.Tensor, bias: torch.Tensor, negative_slope: float = 0.2, scale: float = SQRT2) -> torch.Tensor:
    """Drop-in for ``models.stylegan2.op.fused_leaky_relu`` (op/fused_act.py:84-85)."""
    return _FusedLeakyReLU.apply(input, bias, negative_slope, scale)


# ------------------------------------------------------------------------------------ a2
def upfirdn2d_raw(x: torch.Tensor, kernel: torch.Tensor, up: Tuple[int, int], down: Tuple[int, int], pad: Tuple[int, int, int, int]) -> torch.Tensor:
    """``upfirdn2d_op.upfirdn2d`` on an NCHW tensor; up/down = (x, y), pad = (x0, x1, y0, y1)."""
    x = _c(x, "input")
    k = _c(kernel, "kernel")
    n, c, h, w = x.shape
    kh, kw = k.shape
    oh = (h * up[1] + pad[2] + pad[3] - kh) // down[1] + 1
    ow = (w * up[0] + pad[0] + pad[1] - kw) // down[0] + 1
    out = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device)
    lib().call("e4s_upfirdn2d", _p(out), _p(x), _p(k), n * c, h, w, kh, kw, up[0], up[1], down[0], down[1], pad[0], pad[1], pad[2], pad[3],
               _stream())
    return out


class _UpFirDn2dBackward(torch.autograd.Function):
    # models/stylegan2/op/upfirdn2d.py:17-82
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        grad_input = upfirdn2d_raw(grad_output, grad_kernel, down, up, g_pad)
        grad_input = grad_input.view(in_size[0], in_size[1], in_size[2], in_size[3])
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad, ctx.in_size, ctx.out_size = up, down, pad, in_size, out_size
        return grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        gg = upfirdn2d_raw(gradgrad_input, kernel, ctx.up, ctx.down, ctx.pad)
        return gg, None, None, None, None, None, None, None, None


class _UpFirDn2d(torch.autograd.Function):
    # models/stylegan2/op/upfirdn2d.py:85-139
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kernel_h, kernel_w = kernel.shape
        _, _, in_h, in_w = input.shape
        ctx.in_size = input.shape
        out = upfirdn2d_raw(input, kernel, up, down, pad)
        out_h, out_w = out.shape[2:]
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = up, down, pad
        g_pad_x0 = kernel_w - pad_x0 - 1
        g_pad_y0 = kernel_h - pad_y0 - 1
        g_pad_x1 = in_w * up_x - out_w * down_x + pad_x0 - up_x + 1
        g_pad_y1 = in_h * up_y - out_h * down_y + pad_y0 - up_y + 1
        ctx.g_pad = (g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        gi = _UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down, ctx.pad, ctx.g_pad, ctx.in_size, ctx.out_size)
        return gi, None, None, None, None


def upfirdn2d(input: torch.Tensor, kernel: torch.Tensor, up: int = 1, down: int = 1, pad: Sequence[int] = (0, 0)) -> torch.Tensor:
    """Drop-in for ``models.stylegan2.op.upfirdn2d`` (op/upfirdn2d.py:142-147)."""
    return _UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


# ------------------------------------------------------------------------- per-stream host state
class _StreamCtx:
    """Host-side state of the wrappers, one instance per (device, HIP stream): the region-map cache, the style-table plan of the forward
    pass in progress, its identity stamp and the split-K workspace.  The C ABI itself is stateless; with this, the Python layer is too as
    far as concurrent users are concerned — two host threads each driving their own stream share nothing here (the supported concurrency
    model: ONE host thread per stream; the prepared weight copies, which ARE shared, publish themselves atomically and are ordered across
    streams by an event, see ``_Prepared``)."""

    __slots__ = ("label_cache", "table_plan", "forward_stamp", "workspace", "up_ctrl", "pinned")

    def __init__(self):
        self.label_cache, self.table_plan, self.forward_stamp, self.workspace, self.up_ctrl, self.pinned = {}, {}, None, None, None, False


_ctxs = {}
_ctx_lock = threading.Lock()


def _ctx(stream=None) -> _StreamCtx:
    if stream is None:
        stream = torch.cuda.current_stream()
    key = (stream.device.index, stream.cuda_stream)
    c = _ctxs.get(key)
    if c is None:
        with _ctx_lock:
            c = _ctxs.setdefault(key, _StreamCtx())
    return c


def release_stream_context(stream) -> bool:
    """Drop ``stream``'s host context — its 128 MB split-K workspace, control words, region-map cache (a long-running process that keeps
    creating streams — ``StreamPipeline``, ``run_clip_streamed`` — would otherwise pin one workspace per stream id it ever used).  The caller vouches
    that no launch of this library is still queued on the stream (``StreamPipeline.close()`` synchronises first).  A context that was prepared for a
    hipGraph capture (``prepare_stream_context``) is PINNED and stays: the graph baked its workspace / control-word pointers in, and torch hands stream
    handles out of a pool of 32 per device, so another stream object can carry the same handle as a live graph's (round-3 advisor finding).
    Returns whether a context was dropped."""
    key = (stream.device.index, stream.cuda_stream)
    with _ctx_lock:
        c = _ctxs.get(key)
        if c is None or c.pinned:
            return False
        del _ctxs[key]
        return True


def prepare_stream_context(stream) -> None:
    """Create ``stream``'s context, its split-K workspace and the device's f16 range words NOW (eagerly, outside any capture), so that a hipGraph
    captured on that stream later bakes in pointers that outlive the graph's private memory pool; the context is pinned (``release_stream_context``
    leaves it alone)."""
    with torch.cuda.stream(stream):
        _workspace(stream.device, 1)
        mx_flags(stream.device)
        _ctx().pinned = True


# ------------------------------------------------------------------------------ region map
STRICT_MASK = os.environ.get("E4S_STRICT_MASK", "1") != "0"


def mask_to_labels(mask: torch.Tensor, strict: Optional[bool] = None) -> torch.Tensor:
    """One-hot ``[bs, ncls, H, W]`` float mask (``labelMap2OneHot``, utils/torch_utils.py:207-213) → uint8 ``[bs, H, W]``
    region map.  The result is cached per mask *object* and stream (all 26 layers of one ``Generator.forward`` share it).
    ``strict`` (default on; ``E4S_STRICT_MASK=0`` disables) checks on the device that the mask really is one-hot and
    raises otherwise — the one-pass kernels are only equivalent to the reference's masked sum for one-hot masks."""
    if mask.dtype == torch.uint8 and mask.dim() == 3:
        return _req(mask, "labels", torch.uint8).contiguous()
    _req(mask, "mask")
    _label_cache = _ctx().label_cache
    key = id(mask)
    ent = _label_cache.get(key)
    if ent is not None and ent[0]() is mask and ent[1] == mask._version:
        return ent[2]
    for k in [k for k, e in _label_cache.items() if e[0]() is None]:
        del _label_cache[k]
    m = _c(mask, "mask")
    if m.dim() != 4:
        raise ValueError(f"mask must be [bs, n_cls, H, W], got {tuple(m.shape)}")
    bs, ncls, h, w = m.shape
    if ncls > MAX_REGIONS:
        raise ValueError(f"{ncls} regions > {MAX_REGIONS}")
    labels = torch.empty((bs, h, w), dtype=torch.uint8, device=m.device)
    check = STRICT_MASK if strict is None else strict
    flag = torch.zeros(1, dtype=torch.int32, device=m.device) if check else None      # (unchecked: no flag word, no fill launch)
    lib().call("e4s_onehot_to_labels", _p(labels), _p(flag), _p(m), bs, ncls, h, w, _stream())
    if check:
        f = int(flag.item())
        if f:
            raise ValueError("mask is not one-hot (" + ("values other than 0/1; " if f & 1 else "") + ("several classes per pixel" if f & 2 else "") +
                             "): the region-aware kernels require labelMap2OneHot-style masks")
    if len(_label_cache) > 8:
        _label_cache.clear()
    _label_cache[key] = (weakref.ref(mask), mask._version, labels)
    return labels


# ------------------------------------------------------------------------- prepared weights
# Arithmetic of the 3x3 modulated convolutions: "sb" = split-bf16 (3 bf16 MFMAs per fp32 product, fp32 accumulate; default),
# "f32" = exact fp32 MFMA.  Both meet the 1e-3 pixel bar (sb: ~8e-5 end to end, f32: ~2e-5); f32 is ~3x slower.
MODCONV_MODE = os.environ.get("E4S_MODCONV", "sb")
CONV_MODE = os.environ.get("E4S_CONV", "sb")      # same switch for the plain convolutions of the regional-style encoder
# BiSeNet's argmax must not move: "sb3" (default) = three-way bf16 split (6 MFMAs per product, fp32-class error, 2.7x less matrix time
# than fp32 MFMA; measured against the CPU oracle it differs on exactly the same kind of pixel as the exact kernel does — true ties,
# top-2 gap 4e-8 of the logit scale: tests/test_gpu_parser.py), "f32" = exact fp32 MFMA, "sb" = the two-way split of the other
# convolutions (flips a handful of near-tie pixels)
PARSER_EXACT = {"f32": True, "sb3": "sb3", "sb": False, "f16x3": "f16x3"}[os.environ.get("E4S_PARSER_CONV", "f16x3")]


def _volatile(t: torch.Tensor) -> bool:
    """A parameter that is being trained: its re-laid-out copy must not be cached across calls.  The version counter the caches key on
    is not a reliable change signal there — fused optimisers (``torch.optim.Adam(fused=True)``) and ``p.data`` updates write the
    parameter without bumping it (measured on this build) — so under autograd every forward prepares its weights from their current
    values and leaves nothing behind; caching resumes with the first ``no_grad`` / frozen-weight forward."""
    return torch.is_grad_enabled() and t.requires_grad


class one_forward:
    """``with ops.one_forward():`` around one forward pass during which the parameters do not change: a trained layer's weights are then
    prepared once in that pass even if several call sites ask for them (tables plan, layer, backward hand-over), instead of once per ask.
    The stamp identifying the pass lives in the current stream's context."""

    def __enter__(self):
        self.ctx = _ctx()
        self.prev, self.ctx.forward_stamp = self.ctx.forward_stamp, object()
        return self

    def __exit__(self, *exc):
        self.ctx.forward_stamp = self.prev


def invalidate_weight_caches(module: torch.nn.Module) -> int:
    """Drop every re-laid-out weight copy held by the drop-in modules under ``module`` (they are rebuilt by the next forward).  Needed only
    after writing parameters behind autograd's back between two ``no_grad`` forwards — ``p.data.copy_(...)``, an EMA update, a replay of a
    captured optimiser step (``pti.GraphedPTIStep`` does it itself) — which leaves no trace the caches could key on; ``load_state_dict``,
    ordinary in-place ops and any training forward are tracked."""
    seen = set()

    def drop(v) -> int:
        if isinstance(v, _Prepared):
            first = id(v) not in seen
            seen.add(id(v))
            v.key = None
            return int(first)
        if isinstance(v, (list, tuple)):       # e.g. bottleneck_IR_SE_Ours._wino: a list of (PreparedConv, PreparedWinograd, PreparedWinogradSplit)
            return sum(drop(c) for c in v)
        if isinstance(v, dict):
            return sum(drop(c) for c in v.values())
        return 0

    return sum(drop(v) for m in module.modules() for v in vars(m).values())


class _Prepared:
    """Base of the prepared-weight caches.  The cached copy is ONE tuple ``(key, payload, stream id, event, streams that may read it)``
    stored with a single attribute assignment, so a reader on another host thread sees either the old or the new copy, never a mix.
    Streams: the copy is built by kernels on the stream that first asks for it; a hit from a different stream (``pipeline.swap_batch`` runs
    the driven and the target chain on two streams over the same encoder / parser weights) first makes that stream wait for the build's
    event and marks the tensors as in use there (they belong to the building stream's allocator pool)."""

    __slots__ = ("_state",)
    _EMPTY = (None, None, None, None, frozenset())

    def __init__(self):
        self._state = self._EMPTY

    @property
    def key(self):
        return self._state[0]

    @key.setter
    def key(self, value):
        if value is not None:
            raise ValueError("a prepared-weight key can only be reset to None")
        self._state = self._EMPTY

    def _lookup(self, key):
        """Payload of the cached copy if it was built for ``key`` (ordered after its build on the current stream), else None."""
        st = self._state
        if key is None or st[0] != key:
            return None
        cur = torch.cuda.current_stream()
        sid = cur.cuda_stream
        if sid != st[2] and sid not in st[4] and st[3] is not None and not torch.cuda.is_current_stream_capturing():
            cur.wait_event(st[3])
            for t in self._tensors(st[1]):
                t.record_stream(cur)
            self._state = st[:4] + (st[4] | {sid},)
        return st[1]

    def _publish(self, key, payload):
        cur = torch.cuda.current_stream()
        ev = None
        if not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(cur)
        self._state = (key, payload, cur.cuda_stream, ev, frozenset())
        return payload

    def __reduce__(self):                 # copy / deepcopy / pickle of a module: the copy starts with an empty cache (no tensors, no events)
        return (self.__class__, ())

    @staticmethod
    def _tensors(payload):
        out = []
        for v in payload:
            for t in (v if isinstance(v, (tuple, list)) else (v,)):
                if isinstance(t, torch.Tensor):
                    out.append(t)
        return out


class PreparedWeights(_Prepared):
    """K-major, scale-folded copy of a ModulatedConv2d weight (+ blur-composed parity kernels for up layers, + the
    squared-sum table for demodulation), as fp32 (``wt``) or as split-bf16 slabs (``wt = (whi, wlo)``).  Rebuilt when the
    parameter (or blur buffer) changes version or storage, or the arithmetic mode changes."""

    __slots__ = ()

    @property
    def wt(self):
        return None if self._state[1] is None else self._state[1][0]

    @property
    def wsq(self):
        return None if self._state[1] is None else self._state[1][1]

    def get(self, weight: torch.Tensor, blur: Optional[torch.Tensor], up: bool, demodulate: bool, tconv: bool = False):
        """``tconv``: slabs of the bare 3x3 weight of an up layer (for the transposed-conv + blur-epilogue pair)."""
        sb = MODCONV_MODE == "sb" and weight.shape[-1] == 3
        if tconv:
            up, blur = False, None
        key = (weight.data_ptr(), weight._version, weight.device, None if blur is None else (blur.data_ptr(), blur._version), up, demodulate, sb)
        vol = _volatile(weight)
        if vol:
            # a parameter under training: valid for the forward pass in progress only (its stamp is part of the key), never across passes
            stamp = _ctx().forward_stamp
            key = None if stamp is None else ("volatile", id(stamp)) + key
        hit = self._lookup(key)
        if hit is not None and (not vol or hit[2] is stamp):
            return hit[0], hit[1]
        w = _c(weight.detach(), "weight")
        _, cout, cin, k, _ = w.shape
        npar = 4 if up else 1
        wsq = torch.empty((cin, cout), dtype=torch.float32, device=w.device) if demodulate else None
        bk = _c(blur, "blur kernel") if up else None
        if up and tuple(bk.shape) != (4, 4):
            raise NotImplementedError(f"up-conv blur kernel must be 4x4, got {tuple(bk.shape)}")
        if sb:
            shape = (npar, (cin + 15) // 16, 9, 2, cout, 8)
            whi = torch.empty(shape, dtype=torch.int16, device=w.device)
            wlo = torch.empty(shape, dtype=torch.int16, device=w.device)
            lib().call("e4s_modconv_prep_weights_sb", _p(whi), _p(wlo), _p(wsq), _p(w), _p(bk), cout, cin, 1 if up else 0, _stream())
            wt = (whi, wlo)
        else:
            wt = torch.empty((npar, cin, k * k, cout), dtype=torch.float32, device=w.device)
            lib().call("e4s_modconv_prep_weights", _p(wt), _p(wsq), _p(w), _p(bk), cout, cin, k, 1 if up else 0, _stream())
        self._publish(key, (wt, wsq, stamp if vol else None))      # (holding the stamp object keeps its id unique while this copy lives)
        return wt, wsq


# The DMA-fed masked kernel (csrc/modconv_mx.hip) for the masked layers of width >= 32 and >= 128 output channels — the seven launches that dominate
# a synthesis step.  E4S_MX: 0 = off (modconv_sb.hip's register-staged kernel), 1 = its pipeline with the split-bf16 arithmetic (bit-identical
# results), 2 = with the f16 + 2 x MX-fp6 arithmetic (about half the matrix-pipe time; 2-3x the split-bf16 error, still 5x inside the 1e-3 bar).
MX_MODE = int(os.environ.get("E4S_MX", "2"))


def mx_arith() -> Optional[int]:
    """Arithmetic selector of the mx kernel under the current settings, or None when it is off."""
    if MX_MODE <= 0 or MODCONV_MODE != "sb":
        return None
    return 1 if (MX_MODE >= 2 and not mx_exact_active()) else 0


def mx_eligible(cin: int, cout: int, w: int, masked: bool) -> bool:
    return mx_arith() is not None and masked and w >= 32 and cout >= 128 and cin % 16 == 0 and not torch.is_grad_enabled()


class PreparedMx(_Prepared):
    """A ModulatedConv2d weight as the row slots ``e4s_region_modconv3x3_mx`` DMAs (``e4s_modconv_prep_weights_mx``); rebuilt when the parameter,
    the blur buffer or the arithmetic changes.  Inference only (``mx_eligible``), so the copy is always cacheable."""

    __slots__ = ()

    def get(self, weight: torch.Tensor, blur: Optional[torch.Tensor], up: bool, arith: int) -> torch.Tensor:
        """``weight``: a ModulatedConv2d weight ``[1, cout, cin, 3, 3]`` (equalised-lr scale folded in, up layers composed with ``blur``) or a plain
        convolution weight ``[cout, cin, 3, 3]`` (as it is: the encoder's convolutions)."""
        key = (weight.data_ptr(), weight._version, weight.device, None if blur is None else (blur.data_ptr(), blur._version), up, arith)
        hit = self._lookup(key)
        if hit is not None:
            return hit[0]
        w = _c(weight.detach(), "weight")
        plain = w.dim() == 4
        cout, cin, k = w.shape[-4], w.shape[-3], w.shape[-1]
        if k != 3 or w.shape[-2] != 3 or (plain and up):
            raise ValueError("the mx kernel is a 3x3 kernel")
        bk = _c(blur, "blur kernel") if up else None
        nbytes = ctypes.c_int64(0)
        if arith in (3, 5):       # the two-phase plain-convolution kernel's unit slots (csrc/conv_mx3.hip); 5: in the tap order of its stride-2 form
            if not plain:
                raise ValueError("arith 3 / 5 (conv_mx3) is a plain-convolution layout")
            lib().call("e4s_conv3x3_mx3_weight_bytes", cout, cin, ctypes.byref(nbytes))
            wmx = torch.empty((nbytes.value,), dtype=torch.uint8, device=w.device)
            lib().call("e4s_conv_prep_weights_mx3" if arith == 3 else "e4s_conv_prep_weights_mx3_s2", _p(wmx), _p(w), cout, cin, _stream())
            self._publish(key, (wmx,))
            return wmx
        if arith == 7:            # the region-uniform block kernel's tap-pair units (csrc/modconv_upblock_mx.hip): the transposed-conv taps, not blur-composed
            if plain:
                raise ValueError("arith 7 (modconv_upblock_mx) is a ModulatedConv2d layout")
            lib().call("e4s_upblock_mx_weight_bytes", cout, cin, ctypes.byref(nbytes))
            wmx = torch.empty((nbytes.value,), dtype=torch.uint8, device=w.device)
            lib().call("e4s_modconv_prep_weights_upblock_mx", _p(wmx), _p(w), cout, cin, _stream())
            self._publish(key, (wmx,))
            return wmx
        if arith == 4:            # the four-parity up kernel's row slots (csrc/modconv_mx4.hip)
            if plain or not up:
                raise ValueError("arith 4 (modconv_mx4) is the layout of a masked up layer")
            lib().call("e4s_modconv_mx4_weight_bytes", cout, cin, ctypes.byref(nbytes))
            wmx = torch.empty((nbytes.value,), dtype=torch.uint8, device=w.device)
            lib().call("e4s_modconv_prep_weights_mx4", _p(wmx), _p(w), _p(bk), cout, cin, _stream())
            self._publish(key, (wmx,))
            return wmx
        lib().call("e4s_modconv_mx_weight_bytes", cout, cin, 1 if up else 0, arith, ctypes.byref(nbytes))
        wmx = torch.empty((nbytes.value,), dtype=torch.uint8, device=w.device)
        if plain:
            lib().call("e4s_conv_prep_weights_mx", _p(wmx), _p(w), cout, cin, arith, _stream())
        else:
            lib().call("e4s_modconv_prep_weights_mx", _p(wmx), _p(w), _p(bk), cout, cin, 1 if up else 0, arith, _stream())
        self._publish(key, (wmx,))          # (a tuple: the base class walks the payload's tensors when another stream first uses the copy)
        return wmx


_mx_words = {}            # device index -> int32[4]: [0] sticky "some launch left the f16 range" bit, [1] a counter that moves whenever one does
_mx_tls = threading.local()
mx_fallbacks = 0          # passes re-run with the exact arithmetic since import (host counter; ``bench.py`` reports it)
mx_false_trips = 0        # trips of guards whose window overlapped another live guard's: the device-wide counter may have been moved by the OTHER pass (upper bound of the false trips)
_open_guards = []         # live guards between their first snapshot and their check (host bookkeeping for the line above)


def mx_flags(device) -> torch.Tensor:
    """The device words the f16 arithmetic reports to (``include/e4s_hip.h``: flags[0] |= 1, flags[1] += 1 per reporting wave).  ONE tensor per device,
    shared by every stream, never reset by the guard — ``MxGuard`` compares two snapshots of the counter, so concurrent streams cannot hide each other's
    reports (a report on another stream can at worst cause a spurious re-run)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _mx_words.get(idx)
    if t is None:
        with _ctx_lock:
            t = _mx_words.get(idx)
            if t is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("the f16 range words must exist before a hipGraph capture: call ops.prepare_stream_context(stream) first")
                t = _mx_words[idx] = torch.zeros((4,), dtype=torch.int32, device=dev)
    return t


def mx_overflowed(reset: bool = True) -> bool:
    """Did any f16-arithmetic launch on this device since the last reset see a value outside the f16 range?  (Synchronises the device word; tests / tools.
    The product path does not poll this: see ``MxGuard``.)"""
    idx = torch.cuda.current_device()
    t = _mx_words.get(idx)
    if t is None:
        return False
    v = int(t[0].item())
    if reset and v:
        t[0:1].zero_()
    return bool(v & 1)


def mx_exact_active() -> bool:
    """Inside ``with ops.mx_exact():`` — every route that would use the f16 (+ fp6) arithmetic takes its split-bf16 form instead."""
    return getattr(_mx_tls, "exact", 0) > 0


class mx_exact:
    """``with ops.mx_exact():`` re-runs of a pass whose f16 arithmetic overflowed: the masked 3x3 layers on the mx kernel's split-bf16 arithmetic
    (bit-identical to modconv_sb.hip), the encoder's convolutions on the Winograd / direct split-bf16 kernels, the parser on the three-way bf16 split.
    Thread-local (one host thread per stream is the supported concurrency model)."""

    def __enter__(self):
        _mx_tls.exact = getattr(_mx_tls, "exact", 0) + 1
        return self

    def __exit__(self, *exc):
        _mx_tls.exact -= 1


class MxGuard:
    """Self-healing of the f16 arithmetic (default since round 3 for the masked 3x3 layers, the encoder's stride-1 3x3 convolutions and the parser).
    A modulated activation >= 65520 becomes an f16 infinity and the frame carries inf / NaN; StyleGAN2 with trained weights is the textbook network for
    that, the reference computes everything in fp32 (models/stylegan2/model.py:276-320) and the seeded networks of the tests peak at |x s| ~ 10.  The kernels
    report an overflow by bumping a device counter; a guard snapshots that counter (4-byte asynchronous copies into pinned memory, ordered on the
    current stream) before and after a pass, and whoever owns the pass re-runs it under ``mx_exact()`` when the two differ:

        g = ops.MxGuard()            # snapshot "before" (a no-op object when the f16 arithmetic is off)
        out = forward(...)
        if g.tripped():              # snapshot "after" + wait for it (a host synchronisation with this stream)
            with ops.mx_exact(): out = forward(...)

    ``Generator.forward`` / ``FSEncoder_PSP.forward`` / ``FaceParser`` do exactly that by themselves — unless a caller up the stack owns a guard already
    (``with ops.mx_guard_scope() as g:``): pipelines that must not synchronise per pass (``pipeline.swap_batch``, ``runner``, ``bench.py``) take the two
    snapshots around their whole unit of work, call ``g.arm()`` when it is queued, and look at ``g.tripped()`` where they synchronise anyway.
    Graph replays: take the guard around ``graph.replay()`` (the snapshots are ordinary stream-ordered copies).

    The counter is DEVICE-WIDE, not per stream: an overflow on stream B between stream A's two snapshots trips A's guard as well.  That is conservative — A's pass is
    re-run in the exact arithmetic although its own values were in range (never a missed overflow) — and such re-runs are counted separately: a pass whose exact
    a trip of a guard whose window overlapped another live guard's (two batches in flight on two streams) MAY have been caused by the other one's pass;
    ``ops.mx_false_trips`` counts those trips (an upper bound of the false ones: with a single guard in flight a trip is always the pass's own)."""

    __slots__ = ("_before", "_after", "_ev", "_live", "_shared", "_done", "__weakref__")

    def __init__(self):
        self._before = self._after = self._ev = None
        self._shared = self._done = False
        # (inside a hipGraph capture a guard is a no-op: no pinned allocation, no event wait — bracket graph.replay() instead)
        self._live = torch.cuda.is_available() and (MX_MODE >= 2 or PARSER_EXACT == "f16x3") and not mx_exact_active() \
            and not torch.cuda.is_current_stream_capturing()
        if self._live:
            self._before = self._snap()
            for ref in _open_guards:               # windows overlap: either pass can move the counter the other one watches
                other = ref()
                if other is not None and not other._done:
                    other._shared = self._shared = True
            _open_guards[:] = [r for r in _open_guards if r() is not None and not r()._done]
            _open_guards.append(weakref.ref(self))

    @staticmethod
    def _snap():
        words = mx_flags(torch.cuda.current_device())
        host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
        host.copy_(words[1:2], non_blocking=True)
        return host

    def arm(self) -> "MxGuard":
        """Queue the "after" snapshot behind everything issued on the current stream so far."""
        if self._live and self._after is None:
            self._after = self._snap()
            self._ev = torch.cuda.Event()
            self._ev.record()
        return self

    def tripped(self) -> bool:
        """Did the counter move between the two snapshots?  Arms the guard if the caller has not, then waits for the "after" copy."""
        if not self._live:
            return False
        global mx_false_trips
        self.arm()
        self._ev.synchronize()
        moved = int(self._after[0]) != int(self._before[0])
        if not self._done:
            self._done = True
            if moved and self._shared:
                mx_false_trips += 1
        return moved


class mx_guard_scope:
    """``with ops.mx_guard_scope() as g:`` — the caller owns the guard of everything inside: the modules' own per-pass guards (and their host
    synchronisation) are switched off for this thread; the caller arms ``g`` when its unit of work is queued and checks ``g.tripped()`` where it
    synchronises anyway, re-running the unit under ``ops.mx_exact()``."""

    def __enter__(self) -> MxGuard:
        self.guard = MxGuard()
        _mx_tls.owned = getattr(_mx_tls, "owned", 0) + 1
        return self.guard

    def __exit__(self, *exc):
        _mx_tls.owned -= 1


def mx_guard_owned() -> bool:
    """Is this thread inside somebody's ``mx_guard_scope`` (who will check and re-run), or already in the exact arithmetic?"""
    return getattr(_mx_tls, "owned", 0) > 0 or mx_exact_active()


def guarded(fn, f16_under_grad: bool = False):
    """Run ``fn()`` under its own guard unless the caller owns one (or the exact arithmetic is already on): one re-run under ``mx_exact()`` if the f16
    arithmetic overflowed.  What the drop-in modules wrap their forward passes in.  ``f16_under_grad``: the pass takes an f16 route whatever the grad mode
    (the parser's two-term f16 convolutions); the generator's and the encoder's f16 routes are inference-only (``mx_eligible`` / ``mx_conv_eligible``), so
    under autograd they need no guard.  Cost: two 4-byte copies and ONE host synchronisation per call — callers that keep several calls in flight own a scope
    instead (``mx_guard_scope``; ``StreamPipeline.submit`` does it for them)."""
    global mx_fallbacks
    if getattr(_mx_tls, "owned", 0) > 0 or mx_exact_active() or (torch.is_grad_enabled() and not f16_under_grad) or not torch.cuda.is_available() \
            or torch.cuda.is_current_stream_capturing():
        return fn()
    g = MxGuard()
    out = fn()
    if g.tripped():
        mx_fallbacks += 1
        with mx_exact():
            out = fn()
    return out


# The chain's up layers in the half-composed form (csrc/modconv_uphc.hip): vertical blur factor composed into the weights, horizontal factor applied to the
# MFMA accumulators in registers.  Needs a rank-1 blur kernel (the reference's always is: outer([1,3,3,1]), model.py:23-31); anything else keeps the fused kernel.
UP_HC = os.environ.get("E4S_UP_HC", "1") != "0"
_blur_rank1 = {}


def _blur_rank1_known(blur: torch.Tensor) -> bool:
    """Has ``blur_is_rank1`` already looked at this kernel tensor?  (It costs a device -> host copy, which a stream capture cannot take.)"""
    ent = _blur_rank1.get(id(blur))
    return ent is not None and ent[0]() is blur and ent[1] == blur._version


def blur_is_rank1(blur: torch.Tensor) -> bool:
    """Is the 4 x 4 blur kernel an outer product (to fp32 rounding)?  One device -> host copy per kernel tensor OBJECT and version, then cached (keyed by the object, with
    a weak reference: a data pointer alone comes back when the allocator reuses a freed tensor's memory — a different kernel would then inherit the old answer)."""
    ent = _blur_rank1.get(id(blur))
    if ent is not None and ent[0]() is blur and ent[1] == blur._version:
        return ent[2]
    k = blur.detach().double().cpu()
    tot = float(k.sum())
    ok = tuple(k.shape) == (4, 4) and tot > 0
    if ok:
        outer = k.sum(1, keepdim=True) * k.sum(0, keepdim=True) / tot
        ok = bool((outer - k).abs().max() <= 1e-6 * k.abs().max())
    if len(_blur_rank1) > 64:
        _blur_rank1.clear()
    _blur_rank1[id(blur)] = (weakref.ref(blur), blur._version, ok)
    return ok


class PreparedHc(_Prepared):
    """Half-composed weight slabs of a single-region up layer (``e4s_modconv_prep_weights_hc``): ``(whi, wlo)`` int16
    ``[2, cin/16, 9, 2, cout, 8]``, or None when the route does not apply (switch off, exact-fp32 mode, blur kernel not rank 1, channel counts)."""

    __slots__ = ()

    def get(self, weight: torch.Tensor, blur: torch.Tensor):
        if not (UP_HC and MODCONV_MODE == "sb"):         # (the only caller is the split-plane chain, an inference route: the copy is always cacheable)
            return None
        _, cout, cin, k, _ = weight.shape
        if k != 3 or cin % 16 or cout % 32 or torch.cuda.is_current_stream_capturing() and not _blur_rank1_known(blur):
            return None
        if not blur_is_rank1(blur):
            return None
        key = (weight.data_ptr(), weight._version, weight.device, blur.data_ptr(), blur._version)
        hit = self._lookup(key)
        if hit is not None:
            return hit[0]
        w = _c(weight.detach(), "weight")
        shape = (2, cin // 16, 9, 2, cout, 8)
        whi = torch.empty(shape, dtype=torch.int16, device=w.device)
        wlo = torch.empty(shape, dtype=torch.int16, device=w.device)
        lib().call("e4s_modconv_prep_weights_hc", _p(whi), _p(wlo), _p(w), _p(_c(blur, "blur kernel")), cout, cin, _stream())
        self._publish(key, ((whi, wlo),))
        return (whi, wlo)


UP_FUSED = True          # (module attributes, not environment switches: the tests flip them to reach the comparison kernels)  # single-region up layers: one launch (blur in LDS) instead of tconv + blur epilogue
UP_TWO_STAGE = True


def modconv_up_single(x, wt, s, d, blur, noise, noise_weight, act_bias, act: bool, cout: int, x_nhwc: bool = False, out_nhwc: bool = False,
                      s_next=None, hc=None) -> torch.Tensor:
    """Single-region up layer: transposed conv (1x MACs) into a pre-blur buffer, then blur + demod + noise + bias + act.
    ``x_nhwc`` / ``out_nhwc``: channel-blocked activations ``[bs, c/8, h, w, 8]`` (fused kernel only).
    ``s_next [bs, 1, cout]``: the chain form — ``x`` is split planes ``[2, bs, cin/8, h, w, 8]`` (already carrying this layer's modulation)
    and the result is written as split planes modulated for the next layer (csrc/modconv_chain.hip); ``hc`` = ``PreparedHc.get(...)``: the
    half-composed slabs — the layer then runs on csrc/modconv_uphc.hip (``d`` must be given)."""
    if s_next is not None:
        _req(x, "x_sp", torch.int16)
        if x.dim() != 6 or x.shape[0] != 2 or x.shape[-1] != 8 or not x.is_contiguous() or not UP_FUSED:
            raise ValueError("modconv_up_single: the chain form takes contiguous split planes [2, bs, C/8, H, W, 8] and the fused kernel")
        _, bs, cb, h, w, _ = x.shape
        cin = cb * 8
        sn = _c(s_next, "s_next").reshape(bs, cout)
        out = _alloc_split_planes(bs, cout, 2 * h, 2 * w, x.device)
        nz = nbs = None
        if noise is not None:
            nz = _c(noise, "noise")
            nbs = nz.shape[0]
            if nz.numel() != nbs * 4 * h * w:
                raise ValueError(f"noise shape {tuple(nz.shape)} does not match output {2 * h}x{2 * w}")
        if hc is not None:
            ev = _timed("modconv_up_hc")
            lib().call("e4s_modconv_up_hc", _p(out), _p(x), _p(hc[0]), _p(hc[1]), _p(_c(d, "d").reshape(bs, cout)), _p(_c(blur, "blur kernel")), _p(nz), nbs or 0,
                       _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cin, cout, h, w, _p(sn), _stream())
        else:
            ev = _timed("modconv_up_fused_sb")
            lib().call("e4s_modconv_up_fused_sb", _p(out), _p(x), _p(wt[0]), _p(wt[1]), _p(s), _p(d), _p(_c(blur, "blur kernel")), _p(nz), nbs or 0,
                       _p(noise_weight) if nz is not None else None, _p(act_bias), (1 if act else 0) | 8 | 16, bs, cin, cout, h, w, _p(sn), _stream())
        if ev is not None:
            ev.record()
        return out
    x = _c(x, "input")
    if x_nhwc:
        bs, cb, h, w, _ = x.shape          # channel-blocked [bs, cin/8, h, w, 8]
        cin = cb * 8
    else:
        bs, cin, h, w = x.shape
    if (x_nhwc or out_nhwc) and not (UP_FUSED and cin % 16 == 0 and cout % 8 == 0):
        raise ValueError("channel-blocked activations need the fused up kernel, cin % 16 == 0 and cout % 8 == 0")
    out = torch.empty((bs, cout // 8, 2 * h, 2 * w, 8) if out_nhwc else (bs, cout, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    nz = nbs = None
    if noise is not None:
        nz = _c(noise, "noise")
        nbs = nz.shape[0]
        if nz.numel() != nbs * 4 * h * w:
            raise ValueError(f"noise shape {tuple(nz.shape)} does not match output {2 * h}x{2 * w}")
    if UP_FUSED:
        ev = _timed("modconv_up_fused_sb")
        lib().call("e4s_modconv_up_fused_sb", _p(out), _p(x), _p(wt[0]), _p(wt[1]), _p(s), _p(d), _p(_c(blur, "blur kernel")), _p(nz), nbs or 0,
                   _p(noise_weight) if nz is not None else None, _p(act_bias), (1 if act else 0) | (2 if x_nhwc else 0) | (4 if out_nhwc else 0),
                   bs, cin, cout, h, w, None, _stream())
        if ev is not None:
            ev.record()
        return out
    z = torch.empty((bs, cout, 2 * h + 1, 2 * w + 1), dtype=torch.float32, device=x.device)
    ev = _timed("modconv_tconv_sb")
    lib().call("e4s_modconv_tconv_sb", _p(z), _p(x), _p(wt[0]), _p(wt[1]), _p(s), bs, cin, cout, h, w, _stream())
    if ev is not None:
        ev.record()
    lib().call("e4s_blur_epilogue", _p(out), _p(z), _p(_c(blur, "blur kernel")), _p(d), _p(nz), nbs or 0,
               _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cout, 2 * h, 2 * w, _stream())
    return out


# the style-table plan of the forward pass in progress lives in the stream context (``_StreamCtx.table_plan``):
# id(ModulatedConv2d) -> (styles data_ptr/shape/stride key, s, d); filled by style_demod_plan for one forward


def style_demod_plan(jobs):
    """jobs: list of (key, styles [bs,nreg,sdim], mod_weight, mod_bias, wsq or None, cout).  Computes every layer's (s, d) in two
    launches and remembers them under ``key`` for the ``style_demod`` calls of the same forward pass."""
    from ._lib import StyleJob
    _table_plan = _ctx().table_plan
    _table_plan.clear()
    if not jobs:
        return
    bs, _, sdim = jobs[0][1].shape
    dev = jobs[0][1].device
    n_s = sum(bs * j[1].shape[1] * j[2].shape[0] for j in jobs)
    n_d = sum(bs * j[1].shape[1] * j[5] for j in jobs if j[4] is not None)
    sbuf = torch.empty(n_s, dtype=torch.float32, device=dev)
    dbuf = torch.empty(max(n_d, 1), dtype=torch.float32, device=dev)
    arr = (StyleJob * len(jobs))()
    so = do = 0
    keep = []
    for i, (key, styles, mw, mb, wsq, cout) in enumerate(jobs):
        _req(styles, "style")
        if styles.stride(-1) != 1 or tuple(styles.shape[::2]) != (bs, sdim):
            raise ValueError("style_demod_plan: all styles must be [bs, nreg, sdim] with unit inner stride")
        nreg, cin = styles.shape[1], mw.shape[0]
        mwc, mbc = _c(mw.detach(), "modulation.weight"), _c(mb.detach(), "modulation.bias")
        keep += [mwc, mbc]
        s_t = sbuf[so: so + bs * nreg * cin].view(bs, nreg, cin)
        so += bs * nreg * cin
        d_t = None
        if wsq is not None:
            d_t = dbuf[do: do + bs * nreg * cout].view(bs, nreg, cout)
            do += bs * nreg * cout
        arr[i] = StyleJob(s_t.data_ptr(), None if d_t is None else d_t.data_ptr(), styles.data_ptr(), styles.stride(0), styles.stride(1),
                          mwc.data_ptr(), mbc.data_ptr(), None if wsq is None else wsq.data_ptr(), nreg, cin, cout, 0)
        _table_plan[key] = ((styles.data_ptr(), tuple(styles.shape), styles.stride(), styles._version), s_t, d_t)
    for i0 in range(0, len(jobs), 32):
        n = min(32, len(jobs) - i0)
        lib().call("e4s_style_demod_batched", ctypes.byref(arr, i0 * ctypes.sizeof(StyleJob)), n, bs, sdim, _stream())
    del keep


def table_plan_clear() -> None:
    _ctx().table_plan.clear()


def style_demod_planned(key, styles):
    ent = _ctx().table_plan.get(key)
    if ent is not None and ent[0] == (styles.data_ptr(), tuple(styles.shape), styles.stride(), styles._version):
        return ent[1], ent[2]
    return None


def style_demod(styles: torch.Tensor, mod_weight: torch.Tensor, mod_bias: torch.Tensor, wsq: Optional[torch.Tensor], cout: int):
    """styles ``[bs, nreg, sdim]`` (any batch/region strides, unit inner stride) → (s ``[bs,nreg,cin]``, d ``[bs,nreg,cout]`` or None)."""
    _req(styles, "style")
    if styles.stride(-1) != 1:
        styles = styles.contiguous()
    bs, nreg, sdim = styles.shape
    mw = _c(mod_weight, "modulation.weight")
    mb = _c(mod_bias, "modulation.bias")
    cin = mw.shape[0]
    s = torch.empty((bs, nreg, cin), dtype=torch.float32, device=styles.device)
    d = torch.empty((bs, nreg, cout), dtype=torch.float32, device=styles.device) if wsq is not None else None
    lib().call("e4s_style_demod", _p(s), _p(d), _p(styles), styles.stride(0), styles.stride(1), _p(mw), _p(mb), _p(wsq), bs, nreg, cin, cout,
               sdim, _stream())
    return s, d


SPLITK_MAX_OUT_FLOATS = 1 << 21   # only feature maps up to 8 MB of output (<= 32x32 at 512 ch, bs 4) are candidates for split-K


def _workspace(device, floats: int) -> torch.Tensor:
    """Split-K partial sums of the stream's launches.  Allocated ONCE per stream at its maximum (16 slices x SPLITK_MAX_OUT_FLOATS
    floats = 128 MB) and never replaced: a hipGraph captured earlier keeps the pointer baked in, so growing the buffer on demand would
    leave such a graph writing into freed memory; per stream because launches on two streams run concurrently."""
    c = _ctx()
    ws = c.workspace
    if ws is None or ws.device != torch.device(device):
        ws = c.workspace = torch.empty(16 * SPLITK_MAX_OUT_FLOATS, dtype=torch.float32, device=device)
    if ws.numel() < floats:
        raise RuntimeError(f"split-K workspace request of {floats} floats exceeds its fixed size {ws.numel()}")
    return ws


# Fusing the single-region ToRGB into the preceding conv's epilogue is correct but measured neutral on MI355X (the longer epilogue
# costs what the separate HBM-bound ToRGB launch costs), so it is off by default.
# Channel-blocked activations ([bs, C/8, H, W, 8]) between the single-region layers of Generator.forward (inference).  A pixel's 8 channels
# are 32 contiguous bytes and consecutive pixels follow: the fused up-sampling kernel's blur passes (8 channels each) and the conv epilogues
# write contiguous lines, and the following conv reads a patch row of 34 pixels as ~9 fully used cache lines per 8 channels instead of 3 partly
# used ones per channel (tile-read probe: 1.8 -> 5 TB/s).  Measured in the pipeline (bench.py, batch 4), hand-overs into the second conv of
# the 512x512 and 1024x1024 stages ("c" links): 958 -> 990 faces/s (fused up-sampling 0.705 -> 0.672 ms, 1024x1024 conv 0.515 -> 0.470,
# 512x512 conv 0.361 -> 0.324); with the hand-overs into the up-sampling kernels as well ("u" links; the kernel reads its patch pixel's two
# 8-channel blocks with four 16-byte loads) another +1.3 % (fused up-sampling 0.707 -> 0.640 ms).
NHWC_CHAIN = True
# The single-region stages as a split-plane chain (csrc/modconv_chain.hip): each producer writes its activation already multiplied by the
# consumer's modulation and split into bf16 hi / lo planes; the 512 x 512 and 1024 x 1024 convs then run on persistent, LDS-DMA-fed kernels.
SP_CHAIN = os.environ.get("E4S_SP_CHAIN", "1") != "0"
# which hand-overs are channel-blocked: "all" (default), or a list of "c" / "u" (every hand-over into a second conv / into an up-conv),
# "u<J>" (into the up-conv of stage J, resolution 2^(J+3)) and "c<J>" (into that stage's second conv)
NHWC_LINKS = "all"


def nhwc_link(kind: str, stage: int) -> bool:
    items = NHWC_LINKS.split(",")
    return NHWC_CHAIN and (NHWC_LINKS == "all" or kind in items or f"{kind}{stage}" in items)
FUSE_RGB = True


def can_fuse_rgb(cout: int, w: int, up: bool, masked: bool) -> bool:
    """Can the single-region ToRGB that follows a layer ride in its epilogue?  (the layer's Cout must fit one workgroup tile)"""
    return FUSE_RGB and MODCONV_MODE == "sb" and not up and w >= 32 and (cout <= 64 or (masked and cout == 128))


UP_BLOCKS = os.environ.get("E4S_UP_BLOCKS", "1") != "0"    # masked up layers: region-uniform 16 x 16 output blocks in the transposed-conv form (csrc/modconv_upblock_mx.hip: f16 + fp6 only —
                                                           # under the split-bf16 arithmetic, incl. the exact re-runs of ops.mx_exact, the whole layer runs in the composed form)
UP_BLOCK = 16
# smallest input width of a masked up layer that tries the block path (128: the 128 -> 256 layer only).  A layer that tries costs two small
# launches and gives up its K split, and below 128 a tile of the composed kernel is half or all of the map's width: on portrait-shaped and
# parser-made maps those layers never qualify (0 % of their tiles).  Measured, faces/s at batch 4 with the path off / from width 32 / 64 / 128:
# portrait-shaped maps 1086 / 1080 / 1082 / 1084 (a second box: 1044 / - / 1031 / -), 4 x 4-cell maps 1092 / 1229 / 1203 / 1139, the bench's
# blocky maps 1084 / 1113 / 1126 / 1127 (tools/sweep_blocks_minw.sh).  E4S_UP_BLOCKS_MINW=64 / 32 for maps made of large cells.
UP_BLOCKS_MIN_WIDTH = int(os.environ.get("E4S_UP_BLOCKS_MINW", "128"))



UP_BLOCKS_ONLY_ONE_KERNEL = None   # tests: "blocks" / "composed" = NaN-prefill the output and launch only that kernel of the pair (who writes which block)
UP_BLOCKS_MIN_PERCENT_SMALL = 90   # the same for a layer whose composed launch fits the chip at once
UP_BLOCKS_MIN_PERCENT = 40   # below this share of qualifying tiles a layer stays entirely in the composed form


def uniform_blocks(labels: torch.Tensor, ho: int, wo: int, nreg: int, with_ctrl: bool = False, min_percent: int = None):
    """``(blocks [bs, ho/16, wo/16], sub [bs, ho/8, wo/8])`` uint8 (``e4s_uniform_blocks``): ``sub`` = the region shared by all pixels of an
    8 x 8 output sub-block (labels sampled 'nearest' at ``ho`` x ``wo``), 255 if they differ; ``blocks`` = the region of a 16 x 16 block whose
    four sub-blocks share one, 255 otherwise — and 255 for a whole row of four blocks (a tile
    of the composed kernel) unless all four qualify.  ``with_ctrl``: also the control words (``ctrl[2]`` = 1 if at least
    ``UP_BLOCKS_MIN_PERCENT`` of those rows qualify: the consumers leave the layer in the composed form otherwise)."""
    lab = _labels_u8(labels, "labels")
    bs, lh, lw = lab.shape
    if ho % UP_BLOCK or wo % UP_BLOCK:
        raise ValueError("uniform_blocks: output size must be a multiple of 16")
    blocks = torch.empty((bs, ho // UP_BLOCK, wo // UP_BLOCK), dtype=torch.uint8, device=lab.device)
    sub = torch.empty((bs, ho // 8, wo // 8), dtype=torch.uint8, device=lab.device)
    ctx = _ctx()
    ctrl = getattr(ctx, "up_ctrl", None)
    if ctrl is None or ctrl.device != lab.device:
        ctrl = ctx.up_ctrl = torch.zeros((4,), dtype=torch.int32, device=lab.device)      # per stream; every launch leaves its counters zeroed
    lib().call("e4s_uniform_blocks", _p(blocks), _p(sub), _p(ctrl), _p(lab), bs, lh, lw, ho, wo, nreg, 0,
               UP_BLOCKS_MIN_PERCENT if min_percent is None else int(min_percent), _stream())
    return (blocks, sub, ctrl) if with_ctrl else (blocks, sub)


def region_modconv3x3(x, wt, s, d, labels, noise, noise_weight, act_bias, act: bool, cout: int, up: bool, rgb=None, want_out: bool = True,
                      x_nhwc: bool = False, out_nhwc: bool = False, s_next=None, up_blocks=None, mx=None, mx4=None):
    """``mx = (wmx, arith)`` (``PreparedMx``, a layer ``mx_eligible`` accepts): run on the DMA-fed kernel of csrc/modconv_mx.hip.
    ``mx4`` (with ``mx``, arith 1, an up layer ``mx4_eligible`` accepts; ``PreparedMx.get(..., arith=4)``): one launch of csrc/modconv_mx4.hip — the tiles whose
    positions' 2 x 2 outputs share a region as four-parity tiles, the others as the composed kernel's tiles (bit-identical results either way).
    ``rgb = (wt_rgb [cout,3], s_rgb [bs,1,cout], bias [1,3,1,1], skip or None, up_kernel)`` fuses the following single-region
    ToRGB; the call then returns ``(out, rgb_image)``.  ``want_out=False`` (with ``rgb``) skips writing the layer's own activation
    and returns ``(None, rgb_image)``.  ``x_nhwc`` / ``out_nhwc``: the activation is channel-blocked, ``[bs, c/8, h, w, 8]`` (split-bf16
    kernel, width >= 32; the 256x256-and-up layers can chain in this layout inside ``Generator.forward``).  ``s_next [bs, 1, cout]`` (masked
    layer with a fused ToRGB): the activation is written as split planes ``[2, bs, cout/8, h, w, 8]`` modulated for a single-region consumer.
    ``up_blocks = (wmx_blocks, blur_kernel)`` (masked up layer, inference, f16 + fp6 arithmetic; ``PreparedMx.get(..., arith=7)``): the 16 x 16 output blocks under ONE
    region are computed in the transposed-conv form (``e4s_masked_upconv_blocks_mx``: a quarter of the composed form's MACs per block), the composed kernel keeps the rest."""
    x = _c(x, "input")
    if x_nhwc:
        bs, cb, h, w, _ = x.shape          # channel-blocked [bs, cin/8, h, w, 8]
        cin = cb * 8
    else:
        bs, cin, h, w = x.shape
    nreg = s.shape[1]
    ho, wo = (2 * h, 2 * w) if up else (h, w)
    if not want_out and rgb is None:
        raise ValueError("want_out=False only makes sense together with a fused ToRGB")
    if (x_nhwc or out_nhwc) and not (isinstance(wt, tuple) and w >= 32 and cin % 16 == 0 and cout % 8 == 0):
        raise ValueError("channel-blocked activations need the split-bf16 kernel, width >= 32, cin % 16 == 0 and cout % 8 == 0")
    sn = None
    if s_next is not None:
        if rgb is None or not want_out or out_nhwc or up or cout % 8:
            raise ValueError("region_modconv3x3: split-plane output goes with the fused ToRGB of a same-resolution layer")
        sn = _c(s_next, "s_next").reshape(bs, cout)
        out = _alloc_split_planes(bs, cout, ho, wo, x.device)
    else:
        out = torch.empty((bs, cout // 8, ho, wo, 8) if out_nhwc else (bs, cout, ho, wo), dtype=torch.float32, device=x.device) if want_out else None
    lh = lw = 0
    if labels is not None:
        lh, lw = labels.shape[1:]
        if labels.shape[0] != bs:
            raise ValueError(f"mask batch {labels.shape[0]} != input batch {bs}")
    nz = nbs = None
    if noise is not None:
        nz = _c(noise, "noise")
        nbs = nz.shape[0]
        if nz.numel() != nbs * ho * wo:
            raise ValueError(f"noise shape {tuple(nz.shape)} does not match output {ho}x{wo}")
    ws, wsn = None, 0
    if bs * cout * ho * wo <= SPLITK_MAX_OUT_FLOATS and sn is None:
        wsn = 16 * bs * cout * ho * wo
        ws = _workspace(x.device, wsn)
    sb = isinstance(wt, tuple)
    blocks = bctrl = None
    if (up_blocks is not None and UP_BLOCKS and sb and up and labels is not None and mx is not None and mx[1] == 1 and w >= max(32, UP_BLOCKS_MIN_WIDTH) and cout >= 128
            and h % 8 == 0 and w % 16 == 0 and cin % 32 == 0 and cin <= 512 and rgb is None and not (x_nhwc or out_nhwc) and sn is None):
        wmx_blocks, blur_k = up_blocks
        # A layer whose composed launch is one round of workgroups (two per CU) gains nothing from losing some of them — `tools/time_blocks.py 4 tophalf`:
        # 64 -> 128 at batch 4 with half its tiles moved to the block kernel costs 0.42 + 0.21 ms against 0.47 — so it needs (nearly) all tiles to qualify
        composed_wgs = (wo // 64) * (ho // 16) * 4 * -(-cout // 128) * bs
        blocks, sub, bctrl = uniform_blocks(labels, ho, wo, nreg, with_ctrl=True,
                                            min_percent=max(UP_BLOCKS_MIN_PERCENT, UP_BLOCKS_MIN_PERCENT_SMALL) if composed_wgs <= 512 else UP_BLOCKS_MIN_PERCENT)
        if UP_BLOCKS_ONLY_ONE_KERNEL is not None:          # tests: which kernel writes which output block (every block by exactly one of the two)
            out.fill_(float("nan"))
        evb = _timed("masked_upconv_blocks", f"{cin}->{cout} @{h} up")
        if UP_BLOCKS_ONLY_ONE_KERNEL != "composed":
            lib().call("e4s_masked_upconv_blocks_mx", _p(out), _p(x), _p(wmx_blocks), _p(mx_flags(x.device)), _p(s), _p(d), _p(blocks), _p(bctrl), _p(_c(blur_k, "blur kernel")),
                       _p(nz), nbs or 0, _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cin, cout, h, w, nreg, _stream())
        if evb is not None:
            evb.record()
        if UP_BLOCKS_ONLY_ONE_KERNEL == "blocks":
            return out
    ev = _timed(modconv_kernel_name(cout, w, sb, labels is not None, cin, mx[1] if (sb and mx is not None) else None), f"{cin}->{cout} @{h}{' up' if up else ''}")
    rgb_out = None
    if rgb is not None:
        if not sb:
            raise RuntimeError("fused ToRGB needs the split-bf16 kernel")
        r_wt, r_s, r_bias, r_skip, r_upk = rgb
        rgb_out = torch.empty((bs, 3, ho, wo), dtype=torch.float32, device=x.device)
        if r_skip is not None and tuple(r_skip.shape) != (bs, 3, ho // 2, wo // 2):
            raise ValueError(f"skip shape {tuple(r_skip.shape)} != {(bs, 3, ho // 2, wo // 2)}")
        rgb_args = (_p(rgb_out), _p(r_wt), _p(r_s), _p(_c(r_bias.detach(), "bias")), _p(_c(r_skip, "skip")) if r_skip is not None else None,
                    _p(_c(r_upk, "upsample.kernel")) if r_skip is not None else None)
    else:
        rgb_args = (None,) * 6
    if sb and mx is not None:
        if labels is None or w < 32 or cout < 128 or cin % 16 or x_nhwc or out_nhwc:
            raise ValueError("region_modconv3x3: the mx kernel is built for masked layers of width >= 32, cout >= 128, cin % 16 == 0, channels-first")
        wmx, arith = mx
        if mx4 is not None and up and arith == 1 and blocks is None and rgb is None and sn is None and cout % 128 == 0:
            lib().call("e4s_region_upconv_mx4", _p(out), _p(x), _p(mx4), _p(wmx), _p(mx_flags(x.device)), _p(s), _p(d), _p(labels), lh, lw, _p(nz), nbs or 0,
                       _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cin, cout, h, w, nreg, _stream())
        else:
            lib().call("e4s_region_modconv3x3_mx", _p(out), _p(x), _p(wmx), arith, _p(mx_flags(x.device)) if arith else None, _p(s), _p(d), _p(labels), lh, lw,
                       _p(nz), nbs or 0, _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cin, cout, h, w, nreg,
                       (1 if up else 0) | (16 if sn is not None else 0), _p(ws), wsn, *rgb_args, _p(sn),
                       _p(blocks), _p(bctrl) if blocks is not None else None, _stream())
    elif sb:
        lib().call("e4s_region_modconv3x3_sb", _p(out), _p(x), _p(wt[0]), _p(wt[1]), _p(s), _p(d), _p(labels), lh, lw, _p(nz), nbs or 0,
                   _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cin, cout, h, w, nreg,
                   (1 if up else 0) | (2 if x_nhwc else 0) | (4 if out_nhwc else 0) | (16 if sn is not None else 0), _p(ws), wsn, *rgb_args, _p(sn),
                   _p(blocks), _p(bctrl) if blocks is not None else None, _stream())
    else:
        lib().call("e4s_region_modconv3x3", _p(out), _p(x), _p(wt), _p(s), _p(d), _p(labels), lh, lw, _p(nz), nbs or 0,
                   _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cin, cout, h, w, nreg, 1 if up else 0,
                   _p(ws), wsn, _stream())
    if ev is not None:
        ev.record()
    return out if rgb is None else (out, rgb_out)


# ------------------------------------------------------------------------------------ the single-region chain on split planes
def _alloc_split_planes(bs: int, c: int, h: int, w: int, device) -> torch.Tensor:
    """Uninitialised split planes ``[2, bs, c/8, h, w, 8]`` int16 — a view of a flat buffer with 16 more bytes behind the second plane, which
    the producing kernel zeroes (the consumers' padding source; include/e4s_hip.h)."""
    n = 2 * bs * c * h * w
    return torch.empty(n + 8, dtype=torch.int16, device=device)[:n].view(2, bs, c // 8, h, w, 8)


def to_split_planes(x: torch.Tensor, s: torch.Tensor, x_nhwc: bool = False) -> torch.Tensor:
    """fp32 activation ``[bs, C, H, W]`` (or channel-blocked ``[bs, C/8, H, W, 8]``) times the consumer's modulation ``s [bs, 1, C]`` ->
    "split planes" int16 ``[2 (hi, lo), bs, C/8, H, W, 8]`` of bf16 bits (csrc/modconv_chain.hip)."""
    x = _c(x, "input")
    if x_nhwc:
        bs, cb, h, w, _ = x.shape
        c = cb * 8
    else:
        bs, c, h, w = x.shape
    s2 = _c(s, "s").reshape(bs, -1)
    if s2.shape[1] != c or c % 8:
        raise ValueError(f"to_split_planes: modulation {tuple(s.shape)} does not fit {c} channels (multiple of 8)")
    out = _alloc_split_planes(bs, c, h, w, x.device)
    lib().call("e4s_to_split_planes", _p(out), _p(x), _p(s2), bs, c, h, w, int(x_nhwc), _stream())
    return out


def from_split_planes(sp: torch.Tensor) -> torch.Tensor:
    """hi + lo of split planes as fp32 ``[bs, C, H, W]`` (still multiplied by the modulation they were written with); tests / debugging."""
    v = sp.view(torch.bfloat16).float().sum(0)                      # [bs, C/8, H, W, 8]
    bs, cb, h, w, _ = v.shape
    return v.permute(0, 1, 4, 2, 3).reshape(bs, cb * 8, h, w)


def chain_supported(cin: int, cout: int, h: int, w: int, up: bool, last: bool = False) -> bool:
    """Is there a persistent split-plane kernel for this single-region layer?  (Generator(1024): 64 -> 64 @ 512, 32 -> 32 @ 1024.)
    ``last``: the layer is the generator's final convolution (fused ToRGB, no split-plane output).  ``e4s_chain_conv3x3`` is built in exactly
    two variants — 64 -> 64 with ToRGB AND a split-plane hand-over, 32 -> 32 with ToRGB and NO hand-over — so a 64-channel last layer
    (``Generator(512)``) or a 32-channel one that is not the last has no chain kernel and stays on ``region_modconv3x3``."""
    if MODCONV_MODE != "sb":
        return False
    if up:
        return UP_FUSED and UP_TWO_STAGE and cin % 16 == 0 and cout % 32 == 0
    return (cin, cout) == ((32, 32) if last else (64, 64)) and h % 16 == 0 and w % 32 == 0


def chain_conv3x3(x_sp: torch.Tensor, wt, d, noise, noise_weight, act_bias, act: bool, cout: int, s_next=None, rgb=None):
    """Single-region ``StyledConv`` (same resolution) on split planes.  ``s_next [bs, 1, cout]``: also write the activation as split planes
    modulated for the next layer; ``rgb = (wt_rgb, s_rgb, bias, skip or None, up_kernel)``: the following ToRGB fused (as in
    ``region_modconv3x3``).  Returns ``(out_sp or None, rgb image or None)``."""
    from ._lib import ChainLayer
    _req(x_sp, "x_sp", torch.int16)
    if x_sp.dim() != 6 or x_sp.shape[0] != 2 or x_sp.shape[-1] != 8 or not x_sp.is_contiguous():
        raise ValueError("chain_conv3x3: x_sp must be contiguous split planes [2, bs, C/8, H, W, 8]")
    _, bs, cb, h, w, _ = x_sp.shape
    cin = cb * 8
    L = ChainLayer()
    keep = [x_sp]
    L.x_sp, L.whi, L.wlo = x_sp.data_ptr(), wt[0].data_ptr(), wt[1].data_ptr()
    if d is not None:
        dd = _c(d, "d").reshape(bs, cout)
        keep.append(dd)
        L.d = dd.data_ptr()
    if noise is not None:
        nz = _c(noise, "noise")
        if nz.numel() != nz.shape[0] * h * w or nz.shape[0] not in (1, bs):
            raise ValueError(f"noise shape {tuple(nz.shape)} does not match output {h}x{w}")
        keep += [nz, noise_weight]
        L.noise, L.noise_bs, L.noise_weight = nz.data_ptr(), nz.shape[0], noise_weight.data_ptr()
    if act_bias is not None:
        L.act_bias = act_bias.data_ptr()
    L.act = 1 if act else 0
    out_sp = rgb_out = None
    if s_next is not None:
        sn = _c(s_next, "s_next").reshape(bs, cout)
        out_sp = _alloc_split_planes(bs, cout, h, w, x_sp.device)
        keep += [sn, out_sp]
        L.out_sp, L.s_next = out_sp.data_ptr(), sn.data_ptr()
    if rgb is not None:
        r_wt, r_s, r_bias, r_skip, r_upk = rgb
        rgb_out = torch.empty((bs, 3, h, w), dtype=torch.float32, device=x_sp.device)
        rs, rb = _c(r_s, "rgb s").reshape(bs, cout), _c(r_bias.detach(), "rgb bias")
        keep += [rs, rb, rgb_out]
        L.rgb_out, L.rgb_wt, L.rgb_s, L.rgb_bias = rgb_out.data_ptr(), r_wt.data_ptr(), rs.data_ptr(), rb.data_ptr()
        if r_skip is not None:
            sk, uk = _c(r_skip, "skip"), _c(r_upk, "upsample.kernel")
            if tuple(sk.shape) != (bs, 3, h // 2, w // 2):
                raise ValueError(f"skip shape {tuple(sk.shape)} != {(bs, 3, h // 2, w // 2)}")
            keep += [sk, uk]
            L.rgb_skip, L.rgb_up_kernel = sk.data_ptr(), uk.data_ptr()
    L.bs, L.cin, L.cout, L.h, L.w = bs, cin, cout, h, w
    ev = _timed(f"chain_conv3x3<{cin}>")
    lib().call("e4s_chain_conv3x3", ctypes.byref(L), _stream())
    if ev is not None:
        ev.record()
    del keep
    return out_sp, rgb_out


def region_torgb(x, wt, s, labels, bias, skip, up_kernel) -> torch.Tensor:
    x = _c(x, "input")
    bs, cin, h, w = x.shape
    nreg = s.shape[1]
    out = torch.empty((bs, 3, h, w), dtype=torch.float32, device=x.device)
    lh = lw = 0
    if labels is not None:
        lh, lw = labels.shape[1:]
    sk = uk = None
    if skip is not None:
        sk = _c(skip, "skip")
        if tuple(sk.shape) != (bs, 3, h // 2, w // 2):
            raise ValueError(f"skip shape {tuple(sk.shape)} != {(bs, 3, h // 2, w // 2)}")
        uk = _c(up_kernel, "upsample.kernel")
        if tuple(uk.shape) != (4, 4):
            raise NotImplementedError("ToRGB skip upsample kernel must be 4x4")
    lib().call("e4s_region_torgb", _p(out), _p(x), _p(wt), _p(s), _p(labels), lh, lw, _p(_c(bias, "bias")), _p(sk), _p(uk), bs, cin, h, w, nreg,
               _stream())
    return out


# ------------------------------------------------------------------------------------ a7
def grouped_linear(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Optional[Sequence[Optional[torch.Tensor]]], *, scale: float,
                   bias_mul: float = 1.0, act: int = 0, slope: float = 0.2, addend: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x ``[bs, groups, in]`` → ``[bs, groups, out]``; ``weights[g]`` is ``[out, in]``."""
    _req(x, "input")
    if x.stride(-1) != 1:
        x = x.contiguous()
    bs, groups, in_dim = x.shape
    ws = [_c(w.detach() if not w.requires_grad else w, "weight") for w in weights]
    out_dim = ws[0].shape[0]
    if out is None:
        out = torch.empty((bs, groups, out_dim), dtype=torch.float32, device=x.device)
    PtrArr = ctypes.c_void_p * groups
    wp = PtrArr(*[w.data_ptr() for w in ws])
    bp = None
    keep = []
    if biases is not None:
        bl = [None if b is None else _c(b, "bias") for b in biases]
        keep = bl
        bp = PtrArr(*[None if b is None else b.data_ptr() for b in bl])
    ad = _c(addend, "addend") if addend is not None else None
    lib().call("e4s_grouped_linear", _p(out), out.stride(0), out.stride(1), _p(x), x.stride(0), x.stride(1), wp, bp, _p(ad), float(scale),
               float(bias_mul), act, float(slope), bs, groups, in_dim, out_dim, _stream())
    del keep
    return out


# ----------------------------------------------------------------------------- kernel timing hook
class KernelTimer:
    """Optional HIP-event timing of individual launches on the current stream (used by bench.py for the roofline of
    the dominant kernel).  ``with KernelTimer() as kt: ...`` then ``kt.summary()`` → {name: (calls, total_ms)}.  ``only``: time these kernel
    names only (two event records per launch cost ~4 % of a step when every launch is timed)."""

    active = None

    def __init__(self, only=None):
        self.events = []
        self.only = None if only is None else set(only)       # names to time (None: every instrumented launch)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b, _ in self.events:
            c, t = out.get(name, (0, 0.0))
            out[name] = (c + 1, t + a.elapsed_time(b))
        return out

    def by_detail(self, name: str):
        """{detail: (calls, total_ms)} of one kernel's launches (detail = the layer a launch belongs to)."""
        torch.cuda.synchronize()
        out = {}
        for n, a, b, detail in self.events:
            if n == name and detail is not None:
                c, t = out.get(detail, (0, 0.0))
                out[detail] = (c + 1, t + a.elapsed_time(b))
        return out


def _timed(name: str, detail: Optional[str] = None):
    kt = KernelTimer.active
    if kt is None or (kt.only is not None and name not in kt.only):
        return None
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    kt.events.append((name, a, b, detail))
    a.record()
    return b


def modconv_kernel_name(cout: int, w: int, sb: Optional[bool] = None, masked: bool = True, cin: Optional[int] = None, mx: Optional[int] = -1) -> str:
    """Template instantiation the dispatch picks (mirrors the switches in csrc/modconv.hip, csrc/modconv_sb.hip and csrc/modconv_mx.hip).
    ``mx``: arithmetic of the DMA-fed masked kernel when the launch takes that route (None: it does not; -1: what an inference forward would do)."""
    if sb is None:
        sb = MODCONV_MODE == "sb"
    if mx == -1:
        mx = mx_arith() if (sb and masked and w >= 32 and cout >= 128 and (cin is None or cin % 16 == 0)) else None
    if sb and mx is not None:
        return f"region_modconv_mx_kernel<{mx}>"
    if sb:
        if w >= 32:
            cfg = "4,1,1,8,5" if (masked and cout >= 128) else ("2,2,1,4,5" if cout > 32 else "1,2,1,4,5")
        else:
            cfg = "1,2,2,2,4" if w >= 16 else ("1,1,2,2,3" if w >= 8 else "1,1,2,2,2")
        return f"region_modconv_sb_kernel<{cfg}>"
    if w >= 32:
        cfg = "2,2,2,2,5" if cout > 64 else ("2,2,1,4,5" if cout > 32 else "1,2,1,4,5")
    else:
        cfg = "2,2,2,2,4" if w >= 16 else ("2,1,2,2,3" if w >= 8 else "2,1,2,2,2")
    return f"region_modconv_kernel<{cfg}>"


# ------------------------------------------------------------------------------------ switches of the encoder / parser stage (ops_encode.py reads them here)
STEM7 = True            # the parser's 7x7 stride-2 stem on csrc/stem7.hip (attribute; off: the exact-fp32 implicit GEMM of conv.hip)
WINOGRAD = os.environ.get("E4S_WINOGRAD", "1") != "0"
WINOGRAD_MIN_CIN = 256
# Where it pays (tools/time_winograd.py, tools/time_swap.py): the direct kernel runs at the board's sustained MFMA rate once a launch fills the chip
# (16 faces: 0.255 ms per 512 -> 512 @32^2 launch against 0.239 for Winograd, and slower end to end with its three launches and 268 MB of
# transformed operands), the 16 GEMMs on e4s_gemm_sb reach two thirds of it — so Winograd is the route of SMALL batches, where the direct launch is
# latency-bound: one swap (two faces) 7.61 -> 6.65 ms, two swaps 10.04 -> 8.75, four 14.16 -> 13.83, eight 25.3 -> 25.5 (off).
WINOGRAD_MIN_TILES = 256
WINOGRAD_MAX_TILES = 2048
# E4S_ENC_ROUTE_BY_IMAGE=1: the encoder's convolution routes (Winograd / DMA-fed f16 + fp6 / direct) are chosen from ONE image's shape, never from the batch:
# a face's style vectors are then bit-identical whatever batch it travels in (the reference processes one frame at a time, face_swap_video_pipeline.py:337).
# Off by default: the batch-aware choice is 5-10 % faster on the full swap's 16-image launches and changes style vectors by <= 5e-5 (tests/test_gpu_encoder.py).
ENC_ROUTE_BY_IMAGE = os.environ.get("E4S_ENC_ROUTE_BY_IMAGE", "0") != "0"
MX3 = os.environ.get("E4S_MX3", "1") != "0"     # plain f16 + fp6 convolutions on the two-phase kernel (0: the one-phase kernel of modconv_mx.hip)
MX_CONV_MIN_WORKGROUPS = 128       # (layers that can also take the Winograd route) below half a round of the chip Winograd / the direct kernel serve a launch better
MX_CONV_MIN_WORKGROUPS_PER_IMAGE = 64
UP_MX4 = True      # masked up layers: tiles whose positions' 2 x 2 outputs share a region on the four-parity kernel (csrc/modconv_mx4.hip); no environment switch — bench.py's in-run A/B flips it
# the map between the two 3x3 convolutions of an IR-SE unit is handed over channel-blocked ([bs, c / 4, h, w, 4]) when both run on the two-phase kernel
# (ops_encode.conv3x3_s1_c4_pair; csrc/conv_mx3.hip, round 5).  0: plain NCHW planes as before — the values are the same, bit for bit
ENC_C4_LINK = os.environ.get("E4S_ENC_C4", "1") != "0"
# ... and, where the unit's width allows (depth % 32 == 0), as the consumer's OPERANDS (ops_encode.MxOperandMap: f16 part, fp6 codes, block scales — the producer's epilogue
# computes once per pixel what the consumer's staging computed per patch pixel and chunk; the same bits again).  0: the channel-blocked / plain hand-over
ENC_PREP_LINK = os.environ.get("E4S_ENC_PREP", "1") != "0"
S2_MX3 = True           # the encoder's stride-2 3x3 convolutions on the stride-2 form of csrc/conv_mx3.hip (attribute; off: the direct split-bf16 kernel)
# The squeeze-excite gate of an IR-SE unit is sigmoid(fc2 . relu(fc1 . mean(IN(r)))) with bias-free 1x1 convolutions (helpers.py:56-72) behind an affine-free
# InstanceNorm2d (helpers.py:128-139): the pooled vector is the mean of an instance-normalised plane — exactly 0 — so the gate is sigmoid(0) = 1/2 for every channel of
# every image.  What the reference's own launches compute there is the rounding noise of that mean (1e-8 .. 1e-6 of a unit: whatever order its sums ran in) pushed through
# two small matrices: 0.5 to within 1e-6.  True: the unit multiplies by the constant and launches no gate kernel (24 latency-bound launches, 0.78 ms of the encoder's 10.3 ms
# per 16 faces); False: the gate is computed from the normalised plane's measured mean as before (tests/test_gpu_encoder.py measures both and their difference).
SE_GATE_IS_HALF = True
NGA_STATS_MAX_PIXELS = 16384          # planes a single workgroup holds in registers (e4s_norm_gate_add_stats)


# ------------------------------------------------------------------------------------ switches of the gradient stage (ops_grad.py reads them here)
NATIVE_BWD = os.environ.get("E4S_NATIVE_BWD", "1") != "0"
# The backward of the path runs on this library's kernels.  The two ways a vendor library can still enter it are both opt-in: E4S_NATIVE_BWD=0 (the
# stock-PyTorch forms of torch_ref.py, kept as the comparison arm of the gradient tests) and E4S_ALLOW_MIOPEN_BWD=1 (aten.convolution_backward for the
# single-region shapes the hand-written kernels do not cover: cout < 16 or a kernel size other than 1 / 3); without the latter such a shape raises.
ALLOW_LIBRARY_BWD = os.environ.get("E4S_ALLOW_MIOPEN_BWD", "0") != "0"
_FOLD_CHUNK_PX = 1024     # pixels per workgroup of e4s_mconv_fold: one pass per thread (4096: the up layers 0.73 -> 0.61 ms)
_SCALE_CHUNK_PX = 8192
GEMM_SPLITK_CAP_FLOATS = 64 << 20     # at most 256 MB of split-K partial products per call
# the data / style gradient of a masked 3x3 layer as one kernel instead of the GEMM that writes U + the fold that reads it twice.  Off by default:
# measured slower on all but the two largest masked layers (csrc/mconv_dgrad.hip, STATUS)
# arithmetic of the single-region layers' data gradient (a 3x3 correlation on csrc/conv.hip): False = the two-way bf16 split the forward itself uses (3 MFMAs per
# product; default since round 6: every gradient bar keeps its margin — worst parameter at 0.48 of its bar either way — and the PTI step loses 0.23 ms), "sb3" = the
# three-way split (6 MFMAs, fp32-class; E4S_DGRAD_SPLIT=sb3)
DGRAD_SINGLE_SPLIT = "sb3" if os.environ.get("E4S_DGRAD_SPLIT", "sb") == "sb3" else False
DGRAD_FUSED = False           # e4s_mconv_dgrad (csrc/mconv_dgrad.hip): measured slower than GEMM + fold on all but the two largest masked layers; no environment
DGRAD_FUSED_MIN_WIDTH = 32    # switch any more — the parity tests and tools/time_dgrad.py set the attribute


# ------------------------------------------------------------------------------------ the other stages' wrappers, re-exported
# (at the end: both modules import this one's helpers)
from .ops_encode import *     # noqa: E402,F401,F403   a8 - a10: encoder and parser operators
from .ops_post import *       # noqa: E402,F401,F403   f2 / f3: mask surgery, paste-back masks, Pillow resize, multi-band blend
from .ops_post import _labels_u8, _pil_bicubic_tables, _pil_tables      # noqa: E402,F401
from .ops_grad import *       # noqa: E402,F401,F403   f1: native gradients


def __getattr__(name):        # PEP 562
    """The star imports above only see what a stage module has defined when THIS module finishes importing — if ``e4s2024_amd.ops_grad`` (or ``ops_encode`` /
    ``ops_post``) is imported first, it is still empty at that point.  Names they define later resolve here, on first use."""
    import importlib
    import sys
    if not name.startswith("__"):
        for mod in ("ops_encode", "ops_post", "ops_grad"):
            m = sys.modules.get(f"{__package__}.{mod}") or importlib.import_module(f".{mod}", __package__)
            if name in m.__dict__:
                globals()[name] = m.__dict__[name]
                return m.__dict__[name]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
