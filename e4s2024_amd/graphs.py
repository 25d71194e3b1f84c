"""hipGraph capture of fixed-shape calls of the hot path.

At batch 1 (the reference's frame-by-frame video loop, face_swap_video_pipeline.py:406) one full swap is ~900 short launches and
the host cannot issue them as fast as the GPU retires them; capturing the whole call once and replaying it removes the Python /
launch overhead (MI355X: ~3.5 us per eager launch vs ~10-16 us per whole-graph replay).  Everything the path launches goes to
``torch.cuda.current_stream()``, which is the capture stream during capture, so the ctypes-launched HIP kernels are recorded like
torch's own; outputs of ``torch.empty`` inside the call come from the graph's private pool.

Requirements on ``fn``: fixed shapes, no host synchronisation (``ops.STRICT_MASK`` is switched off inside), no data-dependent
control flow, and — for ``randomize_noise=True`` — torch's graph-safe RNG (works: the noise draws are torch ops).
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch

from . import ops


class GraphedCall:
    """``g = GraphedCall(fn, example_inputs); out = g(*new_inputs)`` — inputs are copied into static buffers, the captured graph is
    replayed, the (static) output tensors are returned: clone them if they must survive the next call."""

    def __init__(self, fn: Callable, example_inputs: Sequence[torch.Tensor], warmup: int = 2):
        self.fn = fn
        self.static_in = [t.clone() for t in example_inputs]
        self._strict = ops.STRICT_MASK
        ops.STRICT_MASK = False                       # the one-hot check reads a flag back to the host
        try:
            # warm-up AND capture run on this object's own stream: the per-stream host state of ops (region-map cache, split-K workspace)
            # is then created eagerly, before the capture, and the graph bakes in pointers that outlive its private pool
            self.stream = torch.cuda.Stream()
            ops.prepare_stream_context(self.stream)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream), torch.no_grad():
                for _ in range(warmup):                # builds the weight caches outside the capture
                    self.fn(*self.static_in)
            torch.cuda.current_stream().wait_stream(self.stream)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream), torch.no_grad():
                self.static_out = self.fn(*self.static_in)
        finally:
            ops.STRICT_MASK = self._strict

    def __call__(self, *inputs: torch.Tensor, guard: bool = False):
        """Copy the inputs in and replay.  ``guard=True`` brackets the replay with an f16 range guard (two 4-byte stream-ordered copies into pinned memory,
        outside the graph) and leaves it armed in ``self.guard`` for a caller that checks it where it synchronises anyway; the default replays bare — a
        latency-bound batch-1 caller that never looks at the guard does not pay for it (``checked()`` = replay + check + self-healing)."""
        if len(inputs) != len(self.static_in):
            raise ValueError(f"expected {len(self.static_in)} inputs")
        for dst, src in zip(self.static_in, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"graphed call was captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        # f16 range guard around the replay (the two snapshots are ordinary stream-ordered copies, outside the graph): ``self.guard.tripped()`` tells
        # the caller — where it synchronises anyway — whether this replay's frames can be trusted; ``checked()`` does that and heals by itself
        self.guard = ops.MxGuard() if guard else _Unguarded()
        self.graph.replay()
        if guard:
            self.guard.arm()
        return self.static_out

    def checked(self, *inputs: torch.Tensor):
        """``__call__`` + the f16 range check (one host synchronisation): a replay whose arithmetic left the f16 range is repeated EAGERLY in the
        split-bf16 arithmetic (``ops.mx_exact``) and those results are returned instead of the static outputs."""
        out = self(*inputs, guard=True)
        if self.guard.tripped():
            ops.mx_fallbacks += 1
            with torch.no_grad(), ops.mx_exact(), ops.mx_guard_scope():
                out = self.fn(*self.static_in)
        return out


class _Unguarded:
    """``GraphedCall.guard`` after a bare replay: there is nothing to ask — say so instead of an ``AttributeError`` on ``None`` (or a wrong ``False``)."""

    def arm(self):
        return self

    def tripped(self):
        raise RuntimeError("this replay ran without an f16 range guard: replay with g(..., guard=True) and then ask g.guard.tripped(), or use g.checked(...)")


def graphed_gen_img(net, codes: torch.Tensor, labels: torch.Tensor, randomize_noise: bool = False) -> GraphedCall:
    """``g(codes, labels) -> image`` for a fixed batch size; ``labels`` = uint8 region maps ``[bs, 512, 512]`` or one-hot masks.
    A plain ``g(...)`` replays WITHOUT the f16 range guard (``ops.guarded`` is bypassed inside a capture): with trained weights call ``g.checked(...)``
    or ``g(..., guard=True)`` + ``g.guard.tripped()``."""
    return GraphedCall(lambda c, m: net.gen_img(None, c, m, randomize_noise=randomize_noise)[0], [codes, labels])


def graphed_swap(net, parser, driven: torch.Tensor, target: torch.Tensor, randomize_noise: bool = False) -> GraphedCall:
    """``g(driven, target) -> (uint8 frames [bs,1024,1024,3], target region maps)`` — the whole full-swap unit as one graph.
    A plain ``g(...)`` replays WITHOUT the f16 range guard: with trained weights call ``g.checked(...)`` or ``g(..., guard=True)`` + ``g.guard.tripped()``."""
    from . import pipeline
    return GraphedCall(lambda d, t: pipeline.swap_batch(net, parser, d, t, randomize_noise=randomize_noise), [driven, target])
