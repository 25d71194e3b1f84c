"""Tensor-level wrappers over the C ABI (``include/e4s_hip.h``).

PyTorch is plumbing here: it owns device memory (``torch.empty`` for outputs) and the stream
(``torch.cuda.current_stream()`` is handed to every launch).  All arithmetic of the hot path runs in
``libe4s_hip.so``.  Tensors must be fp32 CUDA tensors; anything else raises (no fallback).
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


# Round 5: the masked same-resolution layers with class-prepared operands (csrc/modconv_mxe.hip): entries = (patch pixel, region) pairs prepared once per 32-channel chunk,
# conversion-free two-phase K loop; tiles with more than 512 entries run the kernel above's tile inside the same launch.  OPT-IN (E4S_MXE=1): measured in round 5 it ties
# with the round-3 kernel inside the pipeline (1 359-1 365 against 1 354-1 374 faces/s on the benchmark's maps, 0.90-0.97 of its time on portrait-shaped maps layer by
# layer: DESIGN.md section 4) — the loop's VALU work is gone but its read phase (38 LDS reads + 4 DMA requests per unit) is longer than the MFMA phase beside it.
MXE = os.environ.get("E4S_MXE", "0") != "0"


def mxe_eligible(cin: int, cout: int, w: int, masked: bool, up: bool) -> bool:
    return MXE and mx_arith() == 1 and mx_eligible(cin, cout, w, masked) and not up and cin % 32 == 0


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
        if arith == 6:            # the entry kernel's unit slots (csrc/modconv_mxe.hip)
            if plain:
                raise ValueError("arith 6 (modconv_mxe) is a ModulatedConv2d layout")
            lib().call("e4s_modconv_mxe_weight_bytes", cout, cin, 1 if up else 0, ctypes.byref(nbytes))
            wmx = torch.empty((nbytes.value,), dtype=torch.uint8, device=w.device)
            lib().call("e4s_modconv_prep_weights_mxe", _p(wmx), _p(w), _p(bk), cout, cin, 1 if up else 0, _stream())
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
                      x_nhwc: bool = False, out_nhwc: bool = False, s_next=None, up_blocks=None, mx=None, mx4=None, mxe=None):
    """``mx = (wmx, arith)`` (``PreparedMx``, a layer ``mx_eligible`` accepts): run on the DMA-fed kernel of csrc/modconv_mx.hip.
    ``mx4`` (with ``mx``, arith 1, an up layer ``mx4_eligible`` accepts; ``PreparedMx.get(..., arith=4)``): one launch of csrc/modconv_mx4.hip — the tiles whose
    positions' 2 x 2 outputs share a region as four-parity tiles, the others as the composed kernel's tiles (bit-identical results either way).
    ``mxe`` (with ``mx``, arith 1; ``PreparedMx.get(..., arith=6)``, a layer ``mxe_eligible`` accepts): the launch of csrc/modconv_mxe.hip — class-prepared operands,
    the tiles with too many (pixel, region) pairs as the round-3 kernel's tiles.
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
        evb = _timed("masked_upconv_blocks", f"{cin}->{cout} @{h} up")
        lib().call("e4s_masked_upconv_blocks_mx", _p(out), _p(x), _p(wmx_blocks), _p(mx_flags(x.device)), _p(s), _p(d), _p(blocks), _p(bctrl), _p(_c(blur_k, "blur kernel")),
                   _p(nz), nbs or 0, _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cin, cout, h, w, nreg, _stream())
        if evb is not None:
            evb.record()
    use_mxe = sb and mx is not None and mxe is not None and mx[1] == 1 and blocks is None and cin % 32 == 0
    # (timing key: the entry kernel's launches count with the masked kernel they replace — bench.py prices the group as one kernel and says so)
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
        if use_mxe:
            lib().call("e4s_region_modconv3x3_mxe", _p(out), _p(x), _p(mxe), _p(wmx), _p(mx_flags(x.device)), _p(s), _p(d), _p(labels), lh, lw,
                       _p(nz), nbs or 0, _p(noise_weight) if nz is not None else None, _p(act_bias), 1 if act else 0, bs, cin, cout, h, w, nreg,
                       (1 if up else 0) | (16 if sn is not None else 0), _p(ws), wsn, *rgb_args, _p(sn), _stream())
        elif mx4 is not None and up and arith == 1 and blocks is None and rgb is None and sn is None and cout % 128 == 0:
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


def chain_upconv(x_sp: torch.Tensor, wt, d, blur, noise, noise_weight, act_bias, act: bool, cout: int, s_next) -> torch.Tensor:
    """Single-region up-sampling ``StyledConv`` (transposed conv + blur) on split planes ``[2, bs, cin/8, h, w, 8]`` -> split planes of its
    ``[bs, cout, 2h, 2w]`` activation, modulated by ``s_next [bs, 1, cout]`` for the next layer.  ``wt`` = bare 3x3 slabs
    (``PreparedWeights.get(..., tconv=True)``)."""
    from ._lib import ChainLayer
    _req(x_sp, "x_sp", torch.int16)
    if x_sp.dim() != 6 or x_sp.shape[0] != 2 or x_sp.shape[-1] != 8 or not x_sp.is_contiguous():
        raise ValueError("chain_upconv: x_sp must be contiguous split planes [2, bs, C/8, H, W, 8]")
    _, bs, cb, h, w, _ = x_sp.shape
    L = ChainLayer()
    dd, sn, bk = _c(d, "d").reshape(bs, cout), _c(s_next, "s_next").reshape(bs, cout), _c(blur, "blur kernel")
    out_sp = _alloc_split_planes(bs, cout, 2 * h, 2 * w, x_sp.device)
    keep = [x_sp, dd, sn, bk, out_sp]
    L.x_sp, L.whi, L.wlo, L.d, L.out_sp, L.s_next = x_sp.data_ptr(), wt[0].data_ptr(), wt[1].data_ptr(), dd.data_ptr(), out_sp.data_ptr(), sn.data_ptr()
    if noise is not None:
        nz = _c(noise, "noise")
        if nz.numel() != nz.shape[0] * 4 * h * w or nz.shape[0] not in (1, bs):
            raise ValueError(f"noise shape {tuple(nz.shape)} does not match output {2 * h}x{2 * w}")
        keep += [nz, noise_weight]
        L.noise, L.noise_bs, L.noise_weight = nz.data_ptr(), nz.shape[0], noise_weight.data_ptr()
    if act_bias is not None:
        L.act_bias = act_bias.data_ptr()
    L.act = 1 if act else 0
    L.bs, L.cin, L.cout, L.h, L.w = bs, cb * 8, cout, h, w
    ev = _timed(f"chain_upconv<{cb * 8}>")
    lib().call("e4s_chain_upconv", ctypes.byref(L), _p(bk), _stream())
    if ev is not None:
        ev.record()
    del keep
    return out_sp


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


# --------------------------------------------------------------------------- a8 / a9 (conv.hip, norm.hip, parser.hip)
class PreparedConv(_Prepared):
    """K-major copy of a plain conv weight ``[cout, cin, k, k]`` (optionally with an eval-mode BatchNorm2d folded in),
    rebuilt when a parameter or BN buffer changes version or storage.  ``get`` returns the prepared copy as an immutable record
    ``(wt, bias, shape)`` (attributes), which is what ``conv2d`` takes."""

    __slots__ = ("exact",)

    class Copy(tuple):
        __slots__ = ()
        wt = property(lambda self: self[0])
        bias = property(lambda self: self[1])
        shape = property(lambda self: self[2])
        kexp = property(lambda self: self[3] if len(self) > 3 else None)       # (f16x3) log2 of the weights' pre-scale

    def __init__(self, exact=False):
        """``exact=True`` pins this convolution to the exact-fp32 MFMA kernel whatever ``CONV_MODE`` says (the face parser:
        its argmax must match the reference pixel for pixel, and split-bf16's ~2e-5 relative logit error flips near-ties);
        ``exact="sb3"`` asks for the three-way bf16 split (fp32-class error) where a split kernel exists, fp32 elsewhere; ``exact="f16x3"`` for the
        two-term f16 split (the same error class at half the MFMAs; its preparation reads the largest folded weight back to pick a power-of-two
        scale — one host sync per weight version, so not for weights prepared inside a graph capture)."""
        super().__init__()
        self.exact = exact

    def __reduce__(self):
        return (self.__class__, (self.exact,))

    def use_sb(self, cin: int, kh: int, kw: int) -> int:
        """Number of bf16 terms per operand: 2 (``wt = (whi, wlo)``) or 3 (``(w0, w1, w2)``) for 3x3 / 1x1 kernels with at least 16
        input channels, 0 = exact-fp32 kernel (always for the 3-channel stems: 7x7 ResNet stem, encoder input layer)."""
        if not (kh == kw and kh in (1, 3) and cin >= 16):
            return 0
        if self.exact == "sb3":
            return 3
        if self.exact == "f16x3":
            return 3 if mx_exact_active() else 4          # (the re-run of a pass whose f16 arithmetic overflowed: three-way bf16 split)
        return 2 if (CONV_MODE == "sb" and not self.exact) else 0

    def get(self, weight: torch.Tensor, bn=None, conv_bias: Optional[torch.Tensor] = None):
        ts = [weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else []) + ([conv_bias] if conv_bias is not None else [])
        key = tuple((t.data_ptr(), t._version) for t in ts) + (weight.device, CONV_MODE, self.exact == "f16x3" and mx_exact_active())
        if any(_volatile(t) for t in ts):
            key = None
        hit = self._lookup(key)
        if hit is not None:
            return hit
        w = _c(weight.detach(), "weight")
        cout, cin, kh, kw = w.shape
        if STEM7 and self.exact == "f16x3" and not mx_exact_active() and (cout, cin, kh, kw) == (64, 3, 7, 7) and conv_bias is None and w.is_cuda:
            # the parser's 7x7 stem on its own kernel (csrc/stem7.hip): K = (c, ky, kx) flattened; the two f16 terms of the BN-folded weight x 2^kexp, built here
            # (64 x 147 values, once per weight version; one host read of the largest folded weight like the f16x3 route below)
            with torch.no_grad():
                if bn is not None:
                    if bn.training:
                        raise RuntimeError("BatchNorm2d must be in eval mode to be folded into the convolution (the parser runs in eval mode)")
                    sc = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + float(bn.eps))
                    wf = w.float() * sc[:, None, None, None]
                    bias = (bn.bias.detach().float() - bn.running_mean.float() * sc).contiguous()
                else:
                    wf, bias = w.float(), None
                m = float(wf.abs().max().item())
                kexp = 10 - int(math.ceil(math.log2(m))) if m > 0 and math.isfinite(m) else 0
                kexp = max(-30, min(30, kexp))
                wk = torch.zeros((64, 160), dtype=torch.float32, device=w.device)
                wk[:, :147] = wf.reshape(64, 147) * float(2.0 ** kexp)
                hi = wk.half()
                lo = (wk - hi.float()).half()
                wt = torch.stack([hi, lo]).view(2, 64, 10, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(torch.int16)
            return self._publish(key, PreparedConv.Copy((wt, bias, (cout, cin, kh, kw), kexp, "stem7")))
        sb = self.use_sb(cin, kh, kw)
        if sb:
            shape = ((cin + 15) // 16, kh * kw, 2, cout, 8)
            wt = tuple(torch.empty(shape, dtype=torch.int16, device=w.device) for _ in range(2 if sb == 4 else sb))
        else:
            wt = torch.empty((cin, kh * kw, cout), dtype=torch.float32, device=w.device)
        bias = torch.empty((cout,), dtype=torch.float32, device=w.device) if (bn is not None or conv_bias is not None) else None
        if bn is not None:
            if bn.training:
                raise RuntimeError("BatchNorm2d must be in eval mode to be folded into the convolution (the parser runs in eval mode)")
            g, be, mu, var, eps = _c(bn.weight.detach(), "bn.weight"), _c(bn.bias.detach(), "bn.bias"), _c(bn.running_mean, "bn.running_mean"), \
                _c(bn.running_var, "bn.running_var"), float(bn.eps)
        else:
            g = be = mu = var = None
            eps = 0.0
        cb = _c(conv_bias.detach(), "conv bias") if conv_bias is not None else None
        kexp = 0
        if sb == 4:
            # power-of-two pre-scale: the largest folded weight lands in (2^9, 2^10], so that every weight's second f16 term stays normal
            with torch.no_grad():
                wmax = w.abs().flatten(1).amax(1)
                if bn is not None:
                    wmax = wmax * (g / torch.sqrt(var + eps)).abs()
                m = float(wmax.max().item())
            kexp = 10 - int(math.ceil(math.log2(m))) if m > 0 and math.isfinite(m) else 0
            kexp = max(-30, min(30, kexp))
            lib().call("e4s_conv_prep_weights_f16x3", _p(wt[0]), _p(wt[1]), _p(bias), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, _p(cb), cout, cin, kh, kw,
                       kexp, _stream())
        elif sb == 3:
            lib().call("e4s_conv_prep_weights_sb3", _p(wt[0]), _p(wt[1]), _p(wt[2]), _p(bias), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, _p(cb),
                       cout, cin, kh, kw, _stream())
        elif sb:
            lib().call("e4s_conv_prep_weights_sb", _p(wt[0]), _p(wt[1]), _p(bias), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, _p(cb), cout,
                       cin, kh, kw, _stream())
        else:
            lib().call("e4s_conv_prep_weights", _p(wt), _p(bias), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, _p(cb), cout, cin, kh, kw,
                       _stream())
        return self._publish(key, PreparedConv.Copy((wt, bias, (cout, cin, kh, kw), kexp, "f16x3" if sb == 4 else "")))


STEM7 = True            # the parser's 7x7 stride-2 stem on csrc/stem7.hip (attribute; off: the exact-fp32 implicit GEMM of conv.hip)


def _is_f16x3(prepared) -> bool:
    return len(prepared) > 4 and prepared[4] == "f16x3"


def conv2d(x: torch.Tensor, prepared: PreparedConv, stride: int = 1, pad: int = 0, *, x1: Optional[torch.Tensor] = None, in_norm=None,
           prelu: Optional[torch.Tensor] = None, relu: bool = False, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(conv2d(cat(x, x1), W) + bias + residual)``; ``in_norm=(mean, rstd)`` applies InstanceNorm to the input on load."""
    x = _c(x, "input")
    cout, cin, kh, kw = prepared.shape
    bs, c0, h, w = x.shape
    if kh != kw:
        raise NotImplementedError("square kernels only")
    if x1 is not None:
        x1 = _c(x1, "input (second half)")
        if x1.shape[0] != bs or tuple(x1.shape[2:]) != (h, w):
            raise ValueError("concatenated inputs must share batch and spatial size")
    if c0 + (0 if x1 is None else x1.shape[1]) != cin:
        raise ValueError(f"conv expects {cin} input channels, got {c0 + (0 if x1 is None else x1.shape[1])}")
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kh) // stride + 1
    out = torch.empty((bs, cout, ho, wo), dtype=torch.float32, device=x.device)
    mean = rstd = None
    if in_norm is not None:
        mean, rstd = _c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")
    act = 2 if prelu is not None else (1 if relu else 0)
    res = None
    if residual is not None:
        res = _c(residual, "residual")
        if tuple(res.shape) != tuple(out.shape):
            raise ValueError(f"residual shape {tuple(res.shape)} != output {tuple(out.shape)}")
    if len(prepared) > 4 and prepared[4] == "stem7":
        if (stride, pad) != (2, 3) or x1 is not None or in_norm is not None or prelu is not None or residual is not None:
            raise ValueError("this weight copy is the parser stem's (7x7, stride 2, pad 3, ReLU or nothing)")
        ev = _timed("conv7x7s2_stem_f16x3")
        lib().call("e4s_conv7x7s2_stem_f16x3", _p(out), _p(x), _p(prepared.wt), _p(prepared.bias), bs, h, w, 1 if relu else 0, prepared.kexp, _stream())
        if ev is not None:
            ev.record()
        return out
    sb = isinstance(prepared.wt, tuple)
    ev = _timed(f"conv2d_{'sb_' if sb else ''}kernel<{kh},{stride}>")
    pr = _p(_c(prelu.detach(), "prelu")) if prelu is not None else None
    if sb and _is_f16x3(prepared):
        lib().call("e4s_conv2d_f16x3", _p(out), _p(x), _p(x1), c0, _p(prepared.wt[0]), _p(prepared.wt[1]), _p(prepared.bias), _p(mean), _p(rstd),
                   pr, _p(res), act, bs, cin, cout, h, w, kh, stride, pad, prepared.kexp, _p(mx_flags(x.device)), _stream())
    elif sb and len(prepared.wt) == 3:
        lib().call("e4s_conv2d_sb3", _p(out), _p(x), _p(x1), c0, _p(prepared.wt[0]), _p(prepared.wt[1]), _p(prepared.wt[2]), _p(prepared.bias),
                   _p(mean), _p(rstd), pr, _p(res), act, bs, cin, cout, h, w, kh, stride, pad, _stream())
    elif sb:
        lib().call("e4s_conv2d_sb", _p(out), _p(x), _p(x1), c0, _p(prepared.wt[0]), _p(prepared.wt[1]), _p(prepared.bias), _p(mean), _p(rstd),
                   pr, _p(res), act, bs, cin, cout, h, w, kh, stride, pad, _stream())
    else:
        lib().call("e4s_conv2d", _p(out), _p(x), _p(x1), c0, _p(prepared.wt), _p(prepared.bias), _p(mean), _p(rstd), pr, _p(res), act, bs, cin,
                   cout, h, w, kh, stride, pad, _stream())
    if ev is not None:
        ev.record()
    return out


# ---- Winograd F(2x2, 3x3) route of the encoder's stride-1 3x3 convolutions (csrc/winograd.hip + the batched split-bf16 GEMM)
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


class PreparedWinograd(_Prepared):
    """``U [16, cout, cin] = G g G^T`` of a 3x3 conv weight, rebuilt when the parameter changes version or storage (``e4s_wino_weight``)."""

    __slots__ = ()

    def get(self, weight: torch.Tensor) -> torch.Tensor:
        key = None if _volatile(weight) else ((weight.data_ptr(), weight._version), weight.device)
        hit = self._lookup(key)
        if hit is not None:
            return hit[0]
        w = _c(weight.detach(), "weight")
        cout, cin, kh, kw = w.shape
        if (kh, kw) != (3, 3):
            raise ValueError("PreparedWinograd: 3x3 kernels only")
        U = torch.empty((16, cout, cin), dtype=torch.float32, device=w.device)
        lib().call("e4s_wino_weight", _p(U), _p(w), cout, cin, _stream())
        return self._publish(key, (U,))[0]


def winograd_route(x: torch.Tensor, cin: int, stride: int):
    """Which route a 3x3, pad-1 convolution of ``x`` takes: ``"f32"`` (Winograd F(2x2, 3x3) on the general split-bf16 GEMM) or ``None`` (the direct /
    DMA-fed kernels).  Stride 1, even maps, inference only; at least ``WINOGRAD_MIN_CIN`` channels and between ``WINOGRAD_MIN_TILES`` and
    ``WINOGRAD_MAX_TILES`` 2 x 2 output tiles (above that the two transforms cost more than the GEMMs save once the direct kernel fills the chip:
    256 -> 256 @64^2 at 16 faces 0.31 against 0.28 ms).  ``ENC_ROUTE_BY_IMAGE``: the tile count is taken PER IMAGE, so a face's style vectors do
    not depend on how many faces share the launch.  (Round 2-3 also carried a route with operands split to bf16 by their producers; it gave rare wrong
    values beside a second stream, was never root-caused and stayed off — removed in round 4.)"""
    bs, _, h, w = x.shape
    if not (WINOGRAD and stride == 1 and x.is_cuda and not torch.is_grad_enabled() and cin >= WINOGRAD_MIN_CIN and h % 2 == 0 and w % 2 == 0):
        return None
    tiles = (1 if ENC_ROUTE_BY_IMAGE else bs) * (h // 2) * (w // 2)
    if tiles < WINOGRAD_MIN_TILES or tiles > WINOGRAD_MAX_TILES:
        return None
    return "f32"


MX3 = os.environ.get("E4S_MX3", "1") != "0"     # plain f16 + fp6 convolutions on the two-phase kernel (0: the one-phase kernel of modconv_mx.hip)
MX_CONV_MIN_WORKGROUPS = 128       # (layers that can also take the Winograd route) below half a round of the chip Winograd / the direct kernel serve a launch better
MX_CONV_MIN_WORKGROUPS_PER_IMAGE = 64


UP_MX4 = True      # masked up layers: tiles whose positions' 2 x 2 outputs share a region on the four-parity kernel (csrc/modconv_mx4.hip); no environment switch — bench.py's in-run A/B flips it


def mx4_eligible(cin: int, cout: int, h: int, w: int, bs: int) -> bool:
    """Does a masked up layer ``[bs, cin, h, w] -> [bs, cout, 2h, 2w]`` (one ``mx_eligible`` accepts, f16 + fp6 arithmetic in force) try the four-parity kernel?
    Its workgroup is 64 output channels x (32 x 8) positions x 4 parities — twice the composed kernel's work — so the launch must still fill the chip (one workgroup
    per CU), and the layer must not be one the region-uniform block path takes (``UP_BLOCKS_MIN_WIDTH``)."""
    if not UP_MX4 or cin % 16 or cout % 128 or w < 32 or (UP_BLOCKS and w >= UP_BLOCKS_MIN_WIDTH and cout >= 128):
        return False
    return (-(-w // 32)) * (-(-h // 8)) * (-(-cout // 64)) * bs >= 256


def mx_conv_eligible(x: torch.Tensor, cout: int) -> bool:
    """Does a stride-1 3x3 convolution of ``x`` run on the DMA-fed kernel's plain-convolution mode?  (inference, the split arithmetic in force,
    16-channel chunks, >= 128 output channels, maps at least 32 wide.)  Layers below ``WINOGRAD_MIN_CIN`` input channels decide from the image alone
    — at least ``MX_CONV_MIN_WORKGROUPS_PER_IMAGE`` 128 co x (32 x 8) px tiles per image — so that a face's style vectors do not depend on how many
    faces share the batch; the 256- / 512-channel layers, whose Winograd route already depends on the launch size, take it when the whole launch
    has ``MX_CONV_MIN_WORKGROUPS`` tiles (the full swap's 16 images; smaller batches keep Winograd / the direct kernel)."""
    bs, cin, h, w = x.shape
    if mx_arith() is None or CONV_MODE != "sb" or torch.is_grad_enabled() or not x.is_cuda:
        return False
    if cin % 16 or cout < 128 or w < 32:
        return False
    per_image = (-(-w // 32)) * (-(-h // 8)) * (-(-cout // 128))
    if cin < WINOGRAD_MIN_CIN:
        return per_image >= MX_CONV_MIN_WORKGROUPS_PER_IMAGE
    if ENC_ROUTE_BY_IMAGE:
        return per_image >= MX_CONV_MIN_WORKGROUPS_PER_IMAGE
    return bs * per_image >= MX_CONV_MIN_WORKGROUPS


def conv3x3_mx(x: torch.Tensor, wmx: torch.Tensor, arith: int, cout: int, *, in_norm=None, prelu: Optional[torch.Tensor] = None,
               out_phased: bool = False) -> torch.Tensor:
    """``PReLU(conv3x3(norm(x), W))``, stride 1, pad 1, on ``e4s_conv3x3_mx`` / ``e4s_conv3x3_mx3`` (``arith`` 3) (``wmx`` from ``PreparedMx.get`` of the
    plain weight with the same ``arith``).  ``out_phased`` (``arith`` 3, even maps): the result's MEMORY is phase planes — ``[bs, cout, 2, 2, h / 2, w / 2]``,
    plane ``(py, px)`` = ``result[..., py::2, px::2]`` — which only ``conv3x3_s2_mx(in_phased=True)`` reads; the returned tensor has that shape."""
    x = _c(x, "input")
    bs, cin, h, w = x.shape
    if out_phased:
        if arith != 3 or h % 2 or w % 2:
            raise ValueError("conv3x3_mx: out_phased needs the two-phase kernel (arith 3) and an even map")
        out = torch.empty((bs, cout, 2, 2, h // 2, w // 2), dtype=torch.float32, device=x.device)
        mean, rstd = (_c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")) if in_norm is not None else (None, None)
        ev = _timed("conv3x3_mx<3>", f"{cin}->{cout} @{h}")
        lib().call("e4s_conv3x3_mx3_phased", _p(out), _p(x), _p(wmx), _p(mx_flags(x.device)), _p(mean), _p(rstd),
                   _p(_c(prelu.detach(), "prelu")) if prelu is not None else None, bs, cin, cout, h, w, _stream())
        if ev is not None:
            ev.record()
        return out
    out = torch.empty((bs, cout, h, w), dtype=torch.float32, device=x.device)
    mean = rstd = None
    if in_norm is not None:
        mean, rstd = _c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")
    ev = _timed(f"conv3x3_mx<{arith}>", f"{cin}->{cout} @{h}")
    pr = _p(_c(prelu.detach(), "prelu")) if prelu is not None else None
    if arith == 3:
        lib().call("e4s_conv3x3_mx3", _p(out), _p(x), _p(wmx), _p(mx_flags(x.device)), _p(mean), _p(rstd), pr, bs, cin, cout, h, w, _stream())
    else:
        lib().call("e4s_conv3x3_mx", _p(out), _p(x), _p(wmx), arith, _p(mx_flags(x.device)) if arith else None, _p(mean), _p(rstd), pr, bs, cin, cout, h, w, _stream())
    if ev is not None:
        ev.record()
    return out


def conv3x3_s2_mx(x: torch.Tensor, wmx: torch.Tensor, cout: int, *, in_norm=None, prelu: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``PReLU(conv3x3(norm(x), W, stride 2, pad 1))`` on ``e4s_conv3x3_s2_mx3`` (f16 + 2 x MX fp6; ``wmx`` from ``PreparedMx.get(weight, None, False, 5)``);
    the input's height and width must be even, ``cin % 32 == 0``, ``cin <= 512``.  A 6-D input ``[bs, cin, 2, 2, h / 2, w / 2]`` is the phase-plane
    hand-over of ``conv3x3_mx(out_phased=True)``."""
    x = _c(x, "input")
    in_phased = x.dim() == 6
    if in_phased:
        if x.shape[2] != 2 or x.shape[3] != 2:
            raise ValueError("conv3x3_s2_mx: a phase-plane input is [bs, cin, 2, 2, h / 2, w / 2]")
        bs, cin, h, w = x.shape[0], x.shape[1], 2 * x.shape[4], 2 * x.shape[5]
    else:
        bs, cin, h, w = x.shape
    if h % 2 or w % 2:
        raise ValueError("conv3x3_s2_mx: the input height and width must be even")
    out = torch.empty((bs, cout, h // 2, w // 2), dtype=torch.float32, device=x.device)
    mean = rstd = None
    if in_norm is not None:
        mean, rstd = _c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")
    ev = _timed("conv3x3_s2_mx<3>", f"{cin}->{cout} @{h}")
    pr = _p(_c(prelu.detach(), "prelu")) if prelu is not None else None
    lib().call("e4s_conv3x3_s2_mx3", _p(out), _p(x), _p(wmx), _p(mx_flags(x.device)), _p(mean), _p(rstd), pr, bs, cin, cout, h, w, 1 if in_phased else 0, _stream())
    if ev is not None:
        ev.record()
    return out


S2_MX3 = True           # the encoder's stride-2 3x3 convolutions on the stride-2 form of csrc/conv_mx3.hip (attribute; off: the direct split-bf16 kernel)


def conv3x3_s2_takes_mx(bs: int, cin: int, cout: int, h: int, w: int, device) -> bool:
    """Does a stride-2 3x3 convolution of a ``[bs, cin, h, w]`` map run on ``e4s_conv3x3_s2_mx3``?  (``mx_conv_eligible`` of the output-sized launch.)"""
    if not (S2_MX3 and MX3 and mx_arith() == 1 and cin % 32 == 0 and cin <= 512 and h % 2 == 0 and w % 2 == 0):
        return False
    return mx_conv_eligible(_ShapeOnly(bs, cin, h // 2, w // 2, device), cout)


class _ShapeOnly:
    """What ``mx_conv_eligible`` looks at of its input (shape, device kind) for a map that does not exist yet."""
    __slots__ = ("shape", "is_cuda")

    def __init__(self, bs, c, h, w, device):
        self.shape = (bs, c, h, w)
        self.is_cuda = torch.device(device).type == "cuda"


def conv3x3_s2(x: torch.Tensor, weight: torch.Tensor, caches) -> torch.Tensor:
    """A stride-2, pad-1 3x3 convolution: the DMA-fed f16 + fp6 kernel where it fits and fills the chip (``conv3x3_s2_takes_mx``), else the direct kernel.
    ``caches = (PreparedConv, PreparedWinograd, PreparedMx)`` of the layer.  ``x`` may be the phase-plane hand-over of ``conv3x3_s1(out_phased=True)``."""
    if x.dim() == 6:
        return conv3x3_s2_mx(x, caches[2].get(weight, None, False, 5), weight.shape[0])
    bs, cin, h, w = x.shape
    if len(caches) > 2 and conv3x3_s2_takes_mx(bs, cin, weight.shape[0], h, w, x.device):
        return conv3x3_s2_mx(x, caches[2].get(weight, None, False, 5), weight.shape[0])
    return conv2d(x, caches[0].get(weight), 2, 1)


def conv3x3_s1_takes_mx3(x: torch.Tensor, cout: int) -> bool:
    """``conv3x3_s1`` runs this layer on the two-phase kernel (``e4s_conv3x3_mx3``)."""
    return (winograd_route(x, x.shape[1], 1) != "f32" and mx_conv_eligible(x, cout) and mx_arith() == 1 and MX3
            and x.shape[1] % 32 == 0 and x.shape[1] <= 512)


def conv3x3_s1(x: torch.Tensor, weight: torch.Tensor, caches, *, in_norm=None, prelu: Optional[torch.Tensor] = None, out_phased: bool = False) -> torch.Tensor:
    """A stride-1, pad-1 3x3 convolution by whichever route fits the launch: Winograd (``winograd_route``: small batches), the DMA-fed kernel
    (``mx_conv_eligible``: launches that fill the chip) or the direct kernel; ``caches = (PreparedConv, PreparedWinograd, PreparedMx)`` of the layer.
    ``out_phased``: see ``conv3x3_mx`` — the caller has checked ``conv3x3_s1_takes_mx3``."""
    if out_phased:
        if not (len(caches) > 2 and conv3x3_s1_takes_mx3(x, weight.shape[0])):
            raise RuntimeError("conv3x3_s1: out_phased on a layer that does not run on the two-phase kernel")
        return conv3x3_mx(x, caches[2].get(weight, None, False, 3), 3, weight.shape[0], in_norm=in_norm, prelu=prelu, out_phased=True)
    route = winograd_route(x, x.shape[1], 1)
    if route == "f32":
        return conv2d_winograd(x, caches[1].get(weight), in_norm=in_norm, prelu=prelu)
    if len(caches) > 2 and mx_conv_eligible(x, weight.shape[0]):
        arith = mx_arith()
        if arith == 1 and MX3 and x.shape[1] % 32 == 0 and x.shape[1] <= 512:
            arith = 3             # same arithmetic, the two-phase kernel (csrc/conv_mx3.hip)
        return conv3x3_mx(x, caches[2].get(weight, None, False, arith), arith, weight.shape[0], in_norm=in_norm, prelu=prelu)
    return conv2d(x, caches[0].get(weight), 1, 1, in_norm=in_norm, prelu=prelu)


def conv2d_winograd(x: torch.Tensor, U: torch.Tensor, *, in_norm=None, prelu: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``act(conv2d(norm(x), W, stride 1, pad 1))`` for a 3x3 kernel through Winograd F(2x2, 3x3): input transform (with the InstanceNorm
    of ``in_norm=(mean, rstd)`` applied on load), 16 GEMMs ``U_k [cout, cin] @ V_k [cin, tiles]`` on the split-bf16 MFMA GEMM, output
    transform with the PReLU.  2.25x fewer multiplications than the direct kernel; same results to ~1e-5 relative."""
    x = _c(x, "input")
    bs, cin, h, w = x.shape
    if U.dim() != 3 or U.shape[0] != 16 or U.shape[2] != cin:
        raise ValueError(f"conv2d_winograd: U {tuple(U.shape)} does not fit {cin} input channels")
    cout = U.shape[1]
    T = bs * (h // 2) * (w // 2)
    mean = rstd = None
    if in_norm is not None:
        mean, rstd = _c(in_norm[0], "in_mean"), _c(in_norm[1], "in_rstd")
    V = torch.empty((16, cin, T), dtype=torch.float32, device=x.device)
    ev = _timed("conv2d_winograd<3,1>")
    lib().call("e4s_wino_input", _p(V), _p(x), _p(mean), _p(rstd), bs, cin, h, w, _stream())
    M = gemm_sb(U, V, True, False, split_k=False)                                    # [16, cout, T]; no K split: a face's result does not depend on the batch
    del V
    out = torch.empty((bs, cout, h, w), dtype=torch.float32, device=x.device)
    lib().call("e4s_wino_output", _p(out), _p(M), _p(_c(prelu.detach(), "prelu")) if prelu is not None else None, bs, cout, h, w, _stream())
    if ev is not None:
        ev.record()
    return out


def plane_stats(x: torch.Tensor, eps: Optional[float] = None, want_nmean: bool = False):
    """Per-(b, c) mean [bs, C] (``eps=None``: mean only = global average pooling), or (mean, rstd[, nmean])."""
    x = _c(x, "input")
    bs, C = x.shape[:2]
    hw = x[0, 0].numel()
    mean = torch.empty((bs, C), dtype=torch.float32, device=x.device)
    if eps is None:
        lib().call("e4s_plane_stats", _p(mean), None, None, _p(x), bs * C, hw, 0.0, _stream())
        return mean
    rstd = torch.empty_like(mean)
    nmean = torch.empty_like(mean) if want_nmean else None
    lib().call("e4s_plane_stats", _p(mean), _p(rstd), _p(nmean), _p(x), bs * C, hw, float(eps), _stream())
    return (mean, rstd, nmean) if want_nmean else (mean, rstd)


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 3


def vec_fc(x: torch.Tensor, weight: torch.Tensor, bn=None, act: int = ACT_NONE) -> torch.Tensor:
    """``act(bn(x @ W^T))`` for ``x [bs, cin]`` and a 1x1 conv weight ``[cout, cin, 1, 1]`` (or ``[cout, cin]``)."""
    x = _c(x, "input")
    w = _c(weight.detach(), "weight")
    cout, cin = w.shape[0], w.shape[1]
    bs = x.shape[0]
    y = torch.empty((bs, cout), dtype=torch.float32, device=x.device)
    if bn is not None:
        if bn.training:
            raise RuntimeError("BatchNorm2d must be in eval mode")
        g, be, mu, var, eps = _c(bn.weight.detach(), "bn.weight"), _c(bn.bias.detach(), "bn.bias"), _c(bn.running_mean, "bn.running_mean"), \
            _c(bn.running_var, "bn.running_var"), float(bn.eps)
    else:
        g = be = mu = var = None
        eps = 0.0
    lib().call("e4s_vec_fc", _p(y), _p(x), _p(w), _p(g), _p(be), _p(mu), _p(var), eps, act, bs, cin, cout, _stream())
    return y


def se_gate(pooled: torch.Tensor, fc1_weight: torch.Tensor, fc2_weight: torch.Tensor) -> torch.Tensor:
    """``sigmoid(fc2 . relu(fc1 . pooled))`` for ``pooled [bs, C]`` and the two bias-free 1x1 conv weights of an SEModule, one launch
    (``e4s_se_gate``); value for value the two ``vec_fc`` calls."""
    x = _c(pooled, "pooled")
    w1, w2 = _c(fc1_weight.detach(), "fc1.weight"), _c(fc2_weight.detach(), "fc2.weight")
    bs, C = x.shape
    H = w1.shape[0]
    if w1.numel() != H * C or w2.numel() != C * H or w2.shape[0] != C or H > 64:
        raise ValueError(f"se_gate: fc1 {tuple(w1.shape)} / fc2 {tuple(w2.shape)} do not fit {C} channels (hidden width <= 64)")
    gate = torch.empty((bs, C), dtype=torch.float32, device=x.device)
    lib().call("e4s_se_gate", _p(gate), _p(x), _p(w1), _p(w2), bs, C, H, _stream())
    return gate


# The squeeze-excite gate of an IR-SE unit is sigmoid(fc2 . relu(fc1 . mean(IN(r)))) with bias-free 1x1 convolutions (helpers.py:56-72) behind an affine-free
# InstanceNorm2d (helpers.py:128-139): the pooled vector is the mean of an instance-normalised plane — exactly 0 — so the gate is sigmoid(0) = 1/2 for every channel of
# every image.  What the reference's own launches compute there is the rounding noise of that mean (1e-8 .. 1e-6 of a unit: whatever order its sums ran in) pushed through
# two small matrices: 0.5 to within 1e-6.  True: the unit multiplies by the constant and launches no gate kernel (24 latency-bound launches, 0.78 ms of the encoder's 10.3 ms
# per 16 faces); False: the gate is computed from the normalised plane's measured mean as before (tests/test_gpu_encoder.py measures both and their difference).
SE_GATE_IS_HALF = True
_half_gates = {}


def half_gate(bs: int, C: int, device) -> torch.Tensor:
    """``[bs, C]`` filled with 0.5, cached per shape and device for the life of the process (a few KB each; never evicted: a captured hipGraph may have the
    pointer baked in).  The fill runs on the stream that first asks; any other stream waits for its event before the first use (as ``_Prepared._lookup``
    does for the weight copies).  Inside a stream capture an uncached shape gets a fresh tensor that is not kept."""
    key = (torch.device(device), bs, C)
    ent = _half_gates.get(key)
    if ent is None:
        t = torch.full((bs, C), 0.5, dtype=torch.float32, device=device)
        if torch.cuda.is_current_stream_capturing():
            return t
        ev = torch.cuda.Event()
        ev.record()
        ent = _half_gates[key] = [t, ev, {torch.cuda.current_stream().cuda_stream}]
        return t
    t, ev, seen = ent
    sid = torch.cuda.current_stream().cuda_stream
    if sid not in seen:
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_event(ev)
            seen.add(sid)
        else:
            torch.cuda.current_stream().wait_event(ev)
    return t


NGA_STATS_MAX_PIXELS = 16384          # planes a single workgroup holds in registers (e4s_norm_gate_add_stats)


def norm_gate_add(x, mean=None, rstd=None, gate=None, shortcut=None, sc_stats=None, sc_stride: int = 1, prelu=None, stats_eps: Optional[float] = None,
                  self_eps: Optional[float] = None):
    """``prelu(((x - mean) * rstd) * gate + shortcut')``.  With ``stats_eps`` the InstanceNorm statistics of the RESULT come back as well:
    ``(out, mean_out, rstd_out)`` — from the same launch for planes of up to 16384 pixels, from ``plane_stats`` otherwise.  ``self_eps`` (instead of ``mean`` / ``rstd``,
    with ``stats_eps``): the statistics of ``x`` itself are computed in that launch too (``e4s_norm_self_gate_add_stats``; ``plane_stats`` first where the plane does not fit)."""
    x = _c(x, "input")
    bs, C, h, w = x.shape
    if self_eps is not None:
        if mean is not None or rstd is not None or stats_eps is None:
            raise ValueError("norm_gate_add: self_eps replaces mean / rstd and goes with stats_eps")
        if not ((h * w) % 4 == 0 and (h * w <= NGA_STATS_MAX_PIXELS or (h * w <= 4 * NGA_STATS_MAX_PIXELS and shortcut is None))):
            mean, rstd = plane_stats(x, self_eps)
            self_eps = None
    out = torch.empty_like(x)
    sc = scm = scr = None
    if shortcut is not None:
        sc = _c(shortcut, "shortcut")
        if tuple(sc.shape) != (bs, C, h * sc_stride, w * sc_stride):
            raise ValueError(f"shortcut shape {tuple(sc.shape)} != {(bs, C, h * sc_stride, w * sc_stride)}")
        if sc_stats is not None:
            scm, scr = _c(sc_stats[0], "sc_mean"), _c(sc_stats[1], "sc_rstd")
    pr = _p(_c(prelu.detach(), "prelu")) if prelu is not None else None
    if self_eps is not None:
        om = torch.empty((bs, C), dtype=torch.float32, device=x.device)
        orr = torch.empty_like(om)
        lib().call("e4s_norm_self_gate_add_stats", _p(out), _p(om), _p(orr), _p(x), float(self_eps), _p(gate), _p(sc), _p(scm), _p(scr), sc_stride, pr,
                   bs, C, h, w, float(stats_eps), _stream())
        return out, om, orr
    if stats_eps is not None and (h * w) % 4 == 0 and h * w <= NGA_STATS_MAX_PIXELS:
        om = torch.empty((bs, C), dtype=torch.float32, device=x.device)
        orr = torch.empty_like(om)
        lib().call("e4s_norm_gate_add_stats", _p(out), _p(om), _p(orr), _p(x), _p(mean), _p(rstd), _p(gate), _p(sc), _p(scm), _p(scr), sc_stride, pr,
                   bs, C, h, w, float(stats_eps), _stream())
        return out, om, orr
    lib().call("e4s_norm_gate_add", _p(out), _p(x), _p(mean), _p(rstd), _p(gate), _p(sc), _p(scm), _p(scr), sc_stride, pr, bs, C, h, w, _stream())
    if stats_eps is not None:
        return (out,) + tuple(plane_stats(out, stats_eps))
    return out


def masked_avg_pool(feats: torch.Tensor, labels: torch.Tensor, nreg: int) -> torch.Tensor:
    feats = _c(feats, "features")
    labels = _c(labels, "labels", torch.uint8)
    bs, C, h, w = feats.shape
    out = torch.empty((bs, nreg, C), dtype=torch.float32, device=feats.device)
    lib().call("e4s_masked_avg_pool", _p(out), _p(feats), _p(labels), labels.shape[1], labels.shape[2], bs, C, h, w, nreg, _stream())
    return out


def _out_like(out: Optional[torch.Tensor], shape, device, name: str) -> torch.Tensor:
    """``out=`` of the resize ops: a contiguous float32 CUDA tensor of exactly ``shape`` (e.g. one half of a batch buffer), or a new one."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == tuple(shape)):
        raise ValueError(f"{name}: out= must be a contiguous float32 CUDA tensor of shape {tuple(shape)}")
    return out


def bilinear_resize(x: torch.Tensor, size, align_corners: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _c(x, "input")
    bs, C, h, w = x.shape
    out = _out_like(out, (bs, C, size[0], size[1]), x.device, "bilinear_resize")
    lib().call("e4s_bilinear_resize", _p(out), _p(x), bs * C, h, w, size[0], size[1], 1 if align_corners else 0, _stream())
    return out


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    x = _c(x, "input")
    bs, C, h, w = x.shape
    out = torch.empty((bs, C, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    lib().call("e4s_maxpool3x3s2", _p(out), _p(x), bs * C, h, w, _stream())
    return out


def gate_add_upsample(feat, gate=None, add_map=None, add_vec=None, up: int = 1) -> torch.Tensor:
    feat = _c(feat, "feat")
    bs, C, h, w = feat.shape
    out = torch.empty((bs, C, h * up, w * up), dtype=torch.float32, device=feat.device)
    am = _c(add_map, "add_map") if add_map is not None else None
    if am is not None and tuple(am.shape) != tuple(feat.shape):
        raise ValueError("add_map must have the shape of feat")
    lib().call("e4s_gate_add_upsample", _p(out), _p(feat), _p(gate), _p(am), _p(add_vec), bs * C, h, w, up, _stream())
    return out


def bilinear_argmax(logits: torch.Tensor, size, lut: Optional[torch.Tensor] = None) -> torch.Tensor:
    logits = _c(logits, "logits")
    bs, ncls, h, w = logits.shape
    out = torch.empty((bs, size[0], size[1]), dtype=torch.uint8, device=logits.device)
    lib().call("e4s_bilinear_argmax", _p(out), _p(logits), _p(lut), bs, ncls, h, w, size[0], size[1], _stream())
    return out


def bicubic_down_normalize(img01: torch.Tensor, taps: torch.Tensor, factor: int, mean: Optional[torch.Tensor] = None,
                           std: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, pm1: bool = False) -> torch.Tensor:
    """``pm1``: the image is in [-1, 1] and ``(img + 1) / 2`` is applied on load (the same values; saves the pass that makes the [0, 1] copy)."""
    x = _c(img01, "image")
    bs, C, h, w = x.shape
    out = _out_like(out, (bs, C, h // factor, w // factor), x.device, "bicubic_down_normalize")
    lib().call("e4s_bicubic_down_normalize_pm1" if pm1 else "e4s_bicubic_down_normalize", _p(out), _p(x),
               _p(_c(taps, "taps")) if taps is not None else None, _p(mean), _p(std), bs, C, h, w, factor, _stream())
    return out


def tensor2im_u8(img: torch.Tensor) -> torch.Tensor:
    """``[bs, 3, H, W]`` float -> uint8 ``[bs, H, W, 3]`` with the reference's ``tensor2im`` arithmetic (truncating cast)."""
    x = _c(img, "image")
    bs, c, h, w = x.shape
    if c != 3:
        raise ValueError("tensor2im_u8 expects 3 channels")
    out = torch.empty((bs, h, w, 3), dtype=torch.uint8, device=x.device)
    lib().call("e4s_tensor2im_u8", _p(out), _p(x), bs, h, w, _stream())
    return out


# ------------------------------------------------------------------------------------ f2 / f3 (maskops.hip)
def _labels_u8(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if t.dtype != torch.uint8 or t.dim() != 3:
        raise ValueError(f"{name}: expected a uint8 [bs, H, W] label map, got {t.dtype} {tuple(t.shape)}")
    return t.contiguous()


def swap_head_mask(source: torch.Tensor, target: torch.Tensor):
    """``swap_head_mask_hole_first`` (swap_face_fine/swap_face_mask.py:194-333) for a batch of 12-class maps on the device.
    ``source`` = the driven face's map, ``target`` = the target frame's map, both uint8 ``[bs, H, W]``.
    Returns ``(res, hole_mask, hole_map, lines)``: uint8 maps (``hole_mask`` in {0,1}) and int32 ``[bs, 2]`` = (eye_line, nose_line)."""
    s, t = _labels_u8(source, "source"), _labels_u8(target, "target")
    if s.shape != t.shape:
        raise ValueError(f"source {tuple(s.shape)} and target {tuple(t.shape)} maps differ in shape")
    bs, h, w = t.shape
    res, hole, hole_map = torch.empty_like(t), torch.empty_like(t), torch.empty_like(t)
    lines = torch.empty((bs, 2), dtype=torch.int32, device=t.device)
    scratch = torch.empty((bs * (3 + w),), dtype=torch.int32, device=t.device)
    if bs == 0:
        return res, hole, hole_map, lines
    lib().call("e4s_swap_head_mask", _p(res), _p(hole), _p(hole_map), _p(lines), _p(s), _p(t), _p(scratch), bs, h, w, _stream())
    return res, hole, hole_map, lines


def foreground_masks(swapped: torch.Tensor, hole_mask: Optional[torch.Tensor] = None, radius: int = 5):
    """Foreground of a swapped map (everything but background / ear-ring / ear / hair / neck, plus the hole:
    face_swap_video_pipeline.py:456-461) and ``create_masks(foreground, operation='expansion', radius)``
    (gradio_utils/face_swapping.py:203-221).  Returns float32 ``[bs, 1, H, W]`` ``(content, border, full)``."""
    m = _labels_u8(swapped, "swapped")
    hm = _labels_u8(hole_mask, "hole_mask") if hole_mask is not None else None
    if hm is not None and hm.shape != m.shape:
        raise ValueError("hole_mask and swapped map differ in shape")
    bs, h, w = m.shape
    content = torch.empty((bs, 1, h, w), dtype=torch.float32, device=m.device)
    border, full = torch.empty_like(content), torch.empty_like(content)
    if bs == 0:
        return content, border, full
    lib().call("e4s_foreground_masks", _p(content), _p(border), _p(full), _p(m), _p(hm), bs, h, w, int(radius), _stream())
    return content, border, full


def frames_to_tensor(frames_u8: torch.Tensor) -> torch.Tensor:
    """uint8 frames ``[bs, H, W, 3]`` -> ``[bs, 3, H, W]`` float in [-1, 1]: ``Compose([ToTensor(), Normalize(.5, .5)])`` (datasets/dataset.py:32, 45;
    face_swap_video_pipeline.py:338-339) on the device, bit for bit (``(x / 255 - 0.5) / 0.5`` in float32)."""
    if not isinstance(frames_u8, torch.Tensor) or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
        raise ValueError("frames_to_tensor: uint8 [bs, H, W, 3] frames")
    if not frames_u8.is_cuda:
        raise RuntimeError("frames must be a CUDA tensor")
    x = frames_u8.contiguous()
    bs, h, w, _ = x.shape
    out = torch.empty((bs, 3, h, w), dtype=torch.float32, device=x.device)
    lib().call("e4s_frames_to_tensor", _p(out), _p(x), bs, h, w, _stream())
    return out


PTI_BG_CLASSES = (0, 4, 11)        # background, hair, ear-rings: what erode_mask / the PTI foreground leave out (video_swap_ft_coach.py:72, 277)


def erode_labels(labels: torch.Tensor, radius: int, bg_classes: Sequence[int] = PTI_BG_CLASSES) -> torch.Tensor:
    """``erode_mask(mask, img, radius)[0]`` (training/video_swap_ft_coach.py:64-93) for a batch of uint8 ``[bs, H, W]`` 12-class maps."""
    m = _labels_u8(labels, "labels")
    bits = 0
    for c in bg_classes:
        bits |= 1 << int(c)
    out = torch.empty_like(m)
    if m.shape[0]:
        lib().call("e4s_erode_labels", _p(out), _p(m), m.shape[0], m.shape[1], m.shape[2], int(radius), bits, _stream())
    return out


# ------------------------------------------------------------------------------------ f1: native gradients of the masked conv
NATIVE_BWD = os.environ.get("E4S_NATIVE_BWD", "1") != "0"
# The backward of the path runs on this library's kernels.  The two ways a vendor library can still enter it are both opt-in: E4S_NATIVE_BWD=0 (the
# stock-PyTorch forms of torch_ref.py, kept as the comparison arm of the gradient tests) and E4S_ALLOW_MIOPEN_BWD=1 (aten.convolution_backward for the
# single-region shapes the hand-written kernels do not cover: cout < 16 or a kernel size other than 1 / 3); without the latter such a shape raises.
ALLOW_LIBRARY_BWD = os.environ.get("E4S_ALLOW_MIOPEN_BWD", "0") != "0"
_FOLD_CHUNK_PX = 1024     # pixels per workgroup of e4s_mconv_fold: one pass per thread (4096: the up layers 0.73 -> 0.61 ms)
_SCALE_CHUNK_PX = 8192


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
    nchunk = -(-(ho * wo) // _SCALE_CHUNK_PX)
    gz = torch.empty((up * up, bs, cout, h * w), dtype=torch.float32, device=gy.device)
    q = torch.empty((nchunk, bs, nreg, cout), dtype=torch.float32, device=gy.device) if want_q else None
    dbias = torch.empty((nchunk, bs, cout), dtype=torch.float32, device=gy.device) if want_sums else None
    dnw = torch.empty((nchunk, bs, cout), dtype=torch.float32, device=gy.device) if want_sums and noise is not None else None
    lib().call("e4s_mconv_scale", _p(gz), _p(q), _p(dbias), _p(dnw), _p(gy), _p(out), _p(d), _p(lab), _p(noise), 0 if noise is None else noise.shape[0],
               _p(noise_weight), _p(act_bias), int(act), bs, cout, h, w, nreg, up, _SCALE_CHUNK_PX, _stream())
    return (gz,) + tuple(None if t is None else (t.sum(0) if nchunk > 1 else t[0]) for t in (q, dbias, dnw))


def _sum_dim(t, dim: int):
    """``t.sum(dim)`` without a launch when that dimension has one element (batch 1 is the PTI case)."""
    return t.select(dim, 0) if t.shape[dim] == 1 else t.sum(dim)


GEMM_SPLITK_CAP_FLOATS = 64 << 20     # at most 256 MB of split-K partial products per call


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
    while split_k and base * ks < 512 and ks * 2 * 4 <= nchunk and ks < 1024 and ks * 2 * batch * M * N <= GEMM_SPLITK_CAP_FLOATS:
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


# the data / style gradient of a masked 3x3 layer as one kernel instead of the GEMM that writes U + the fold that reads it twice.  Off by default:
# measured slower on all but the two largest masked layers (csrc/mconv_dgrad.hip, STATUS)
DGRAD_FUSED = False           # e4s_mconv_dgrad (csrc/mconv_dgrad.hip): measured slower than GEMM + fold on all but the two largest masked layers; no environment
DGRAD_FUSED_MIN_WIDTH = 32    # switch any more — the parity tests and tools/time_dgrad.py set the attribute


def _mconv_input_grads(gz, wg, x, s, lab, up: int, need_x: bool, need_s: bool, need_w: bool):
    """The part of the backward that follows ``gz``: U_g = W_gᵀ gz_g (``e4s_gemm_sb``), dx / ds from one pass over U (``e4s_mconv_fold``) — or both
    from ``e4s_mconv_dgrad`` without U in memory (``DGRAD_FUSED``) — and dW_g = gz_g cols_gᵀ as an implicit GEMM (``e4s_mconv_wgrad``; the 4x4 / 8x8
    maps: unfold kernel + ``e4s_gemm_sb``)."""
    bs, cin, h, w = x.shape
    G, cout, ks, nreg = wg.shape[0], wg.shape[1], wg.shape[-1], s.shape[1]
    dx = ds = dw = None
    if (need_x or need_s) and DGRAD_FUSED and ks == 3 and w >= DGRAD_FUSED_MIN_WIDTH:
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
        nchunk = -(-(h * w) // _FOLD_CHUNK_PX)
        part = torch.empty((nchunk, bs, nreg, cin), dtype=torch.float32, device=x.device) if need_s else None
        lib().call("e4s_mconv_fold", _p(dx), _p(part), _p(u), _p(x), _p(s), _p(lab), bs, cin, h, w, ks, nreg, up, _FOLD_CHUNK_PX, _stream())
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
    while base * kspl < 512 and kspl * 2 * 4 <= nchunk and kspl < 1024 and kspl * 2 * batch * M * N <= GEMM_SPLITK_CAP_FLOATS:
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
            g_skip = upfirdn2d_raw(g, torch.flip(up_kernel, (0, 1)), (1, 1), (2, 2), (1, 1, 1, 1)).view(ctx.skip_shape)
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
            g = upfirdn2d_raw(g.view(bs * cout, 1, out.shape[2], out.shape[3]), torch.flip(blur, (0, 1)), (1, 1), (1, 1), (2, 2, 2, 2))
            g = g.view(bs, cout, 2 * h + 1, 2 * w + 1)
        if need_x:
            # data gradient on the MFMA conv kernel (three-way bf16 split: fp32-class), one sample at a time (its weights are per sample):
            # a 3x3 correlation of g with the transposed + flipped weight, or — for the transposed conv — a stride-2 correlation of g
            wd = wmod.detach().transpose(1, 2)                                     # [bs, cin, cout, k, k]
            if blur is None:
                wd = wd.flip(3, 4)
            if cout >= 16 and k in (1, 3):
                dx = torch.cat([conv2d(g[b:b + 1], PreparedConv(exact="sb3").get(wd[b].contiguous()), 1 if blur is None else 2,
                                       k // 2 if blur is None else 0) for b in range(bs)])
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
            if not ALLOW_LIBRARY_BWD:
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


# ------------------------------------------------------------------------------------ f3: Pillow's resize on the device
_pil_tables = {}


def _pil_bicubic_tables(in_size: int, out_size: int, device):
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the BICUBIC filter (src/libImaging/Resample.c): per output index the
    first input index, the tap count and the taps in 22-bit fixed point.  Computed once per (in, out, device) in float64 like the library."""
    key = (in_size, out_size, str(device))
    hit = _pil_tables.get(key)
    if hit is None:
        scale = in_size / out_size
        fscale = max(scale, 1.0)
        support = 2.0 * fscale
        ksize = int(math.ceil(support)) * 2 + 1
        xmin, cnt, kk = [], [], []
        ss = 1.0 / fscale
        for xx in range(out_size):
            center = (xx + 0.5) * scale
            lo = max(int(center - support + 0.5), 0)
            hi = min(int(center + support + 0.5), in_size)
            ws = []
            for x in range(hi - lo):
                t = abs((x + lo - center + 0.5) * ss)
                ws.append(((1.5 * t - 2.5) * t * t + 1.0) if t < 1.0 else ((((t - 5.0) * t + 8.0) * t - 4.0) * -0.5 if t < 2.0 else 0.0))
            tot = sum(ws)
            if tot != 0.0:
                ws = [v / tot for v in ws]
            row = [int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in ws]
            xmin.append(lo); cnt.append(hi - lo); kk.append(row + [0] * (ksize - len(row)))
        hit = (torch.tensor(xmin, dtype=torch.int32, device=device), torch.tensor(cnt, dtype=torch.int32, device=device),
               torch.tensor(kk, dtype=torch.int32, device=device), ksize)
        if len(_pil_tables) > 32:
            _pil_tables.clear()
        _pil_tables[key] = hit
    return hit


def pil_resize(img_u8: torch.Tensor, size) -> torch.Tensor:
    """``PIL.Image.resize(size)`` (size = (width, height); Pillow's default BICUBIC with its 8-bit fixed-point arithmetic) of uint8
    ``[bs, H, W, C]`` frames on the device, bit for bit: a horizontal then a vertical pass, each rounded to 8 bits
    (face_swap_video_pipeline.py:447 softens the swapped face with ``.resize((512, 512)).resize((1024, 1024))``)."""
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or not img_u8.is_cuda:
        raise ValueError("pil_resize: uint8 [bs, H, W, C] CUDA frames")
    wd, ht = int(size[0]), int(size[1])
    out = img_u8.contiguous()
    for axis, target in ((1, wd), (0, ht)):
        bs, h, w, c = out.shape
        if target == (w if axis == 1 else h):
            continue
        xmin, cnt, kk, ksize = _pil_bicubic_tables(w if axis == 1 else h, target, out.device)
        nxt = torch.empty((bs, h, target, c) if axis == 1 else (bs, target, w, c), dtype=torch.uint8, device=out.device)
        lib().call("e4s_resample_u8", _p(nxt), _p(out), _p(xmin), _p(cnt), _p(kk), ksize, bs, h, w, c, target, axis, _stream())
        out = nxt
    return out


# ------------------------------------------------------------------------------------ f3: multi-band blend
def pyr_down(x: torch.Tensor, round_u8: bool = False) -> torch.Tensor:
    """``cv2.pyrDown`` on ``[..., H, W]`` float planes (``round_u8``: the 8-bit variant's rounding, for a pyramid of a uint8 image)."""
    x = _c(x, "image")
    h, w = x.shape[-2:]
    out = torch.empty(x.shape[:-2] + ((h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device)
    lib().call("e4s_pyr_down", _p(out), _p(x), x.numel() // (h * w), h, w, int(round_u8), _stream())
    return out


def pyr_up(x: torch.Tensor, minuend: Optional[torch.Tensor] = None, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``cv2.pyrUp`` on ``[..., H, W]`` float planes -> ``[..., 2H, 2W]``; ``minuend - up(x)`` or ``up(x) + addend`` when given."""
    x = _c(x, "image")
    h, w = x.shape[-2:]
    out = torch.empty(x.shape[:-2] + (2 * h, 2 * w), dtype=torch.float32, device=x.device)
    for name, t in (("minuend", minuend), ("addend", addend)):
        if t is not None and (tuple(t.shape) != tuple(out.shape) or not t.is_contiguous() or t.dtype != torch.float32):
            raise ValueError(f"pyr_up: {name} must be a contiguous float32 tensor of the output shape {tuple(out.shape)}")
    lib().call("e4s_pyr_up", _p(out), _p(x), _p(minuend), _p(addend), x.numel() // (h * w), h, w, _stream())
    return out


def laplacian_blend(a_u8: torch.Tensor, b: torch.Tensor, mask: torch.Tensor, num_levels: int = 10) -> torch.Tensor:
    """``Laplacian_Pyramid_Blending_with_mask(A, B, m, num_levels)`` (swap_face_fine/multi_band_blending.py:5-48) on the device, with the
    types of its call site: ``a_u8`` uint8 ``[bs, 3, H, W]`` (its Gaussian pyramid is rounded to 8 bits per level like cv2's), ``b`` float
    ``[bs, 3, H, W]`` in [0, 255], ``mask`` float ``[bs, 1 or 3, H, W]``.  Returns the float blend ``[bs, 3, H, W]``."""
    if a_u8.dtype != torch.uint8 or a_u8.dim() != 4 or b.shape != a_u8.shape:
        raise ValueError("laplacian_blend: A is uint8 [bs, 3, H, W] and B a float tensor of the same shape")
    h, w = a_u8.shape[-2:]
    if (h >> num_levels) < 1 or (w >> num_levels) < 1 or h % (1 << (num_levels - 1)) or w % (1 << (num_levels - 1)):
        raise ValueError(f"laplacian_blend: {h}x{w} cannot carry {num_levels} pyramid levels (the reference runs 1024x1024 with 10)")
    ga, gb = a_u8.float().contiguous(), _c(b, "B")
    gm = _c(mask.expand(-1, 3, -1, -1) if mask.shape[1] == 1 else mask, "mask")
    gpa, gpb, gpm = [ga], [gb], [gm]
    for _ in range(num_levels - 1):              # (the reference's last pyrDown, level num_levels, is never used)
        ga, gb, gm = pyr_down(ga, True), pyr_down(gb), pyr_down(gm)
        gpa.append(ga); gpb.append(gb); gpm.append(gm)
    out = torch.lerp(gpb[-1], gpa[-1], gpm[-1])                                    # la*gm + lb*(1-gm) at the coarsest level
    for i in range(num_levels - 1, 0, -1):
        # Laplacian levels of A and B, their masked mix and the reconstruction step in one pass (10 -> 4 plane sets of traffic per level)
        hi, lo = gpa[i - 1], gpa[i]
        nxt = torch.empty_like(hi)
        lib().call("e4s_pyr_blend_level", _p(nxt), _p(out), _p(hi), _p(lo), _p(gpb[i - 1]), _p(gpb[i]), _p(gpm[i - 1]),
                   lo.numel() // (lo.shape[-2] * lo.shape[-1]), lo.shape[-2], lo.shape[-1], _stream())
        out = nxt
    return out


def blending(full_img_u8: torch.Tensor, ori_img: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``blending(full_img, ori_img, mask)`` (multi_band_blending.py:51-74) for 1024 x 1024 frames (its resizes are then identities):
    uint8 ``[bs, 3, H, W]`` = the clipped, truncated ten-level blend."""
    if tuple(full_img_u8.shape[-2:]) != (1024, 1024):
        raise NotImplementedError("blending: the reference resizes to 1024x1024 first; pass 1024x1024 frames")
    return laplacian_blend(full_img_u8, ori_img, mask, 10).clamp_(0, 255).to(torch.uint8)


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
