"""Frame-sharded multi-GPU execution of the swap hot path.

The reference processes a clip one frame at a time on one GPU (``face_swap_video_pipeline.py:337, 406``); frames are
independent units, so they shard with no data-path dependency.  One process per GPU (``torch.distributed``, backend
``nccl`` == RCCL over xGMI on ROCm; ``gloo`` in the CPU tests):

* partition  — contiguous block of the frame index range per rank (frame ``i`` → rank ``i * N // n_frames``);
* broadcast  — what is shared by the whole clip (the source identity's style vectors / W+ codes, 61–442 KB) goes from
  rank 0 to every rank once per clip;
* gather     — finished frames travel to rank 0 as **uint8 HWC** (``tensor2im`` semantics, 3 MB per 1024² frame instead of
  12.6 MB fp32); with 8 GPUs that is ~1 GB/s into rank 0 over 7 dedicated links (≈153 GB/s each), i.e. negligible, so a
  plain ``gather`` is used — per batch and asynchronously in ``run_clip_streamed`` (frames leave while the next batch is computed; what
  the clip benchmark times), or once at the end in ``run_clip`` (shards padded to the largest shard, at most one frame per rank).

Weights are replicated (each rank loads the same checkpoint / seed); there is no collective inside a frame.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the contiguous block owned by ``rank``: item i belongs to rank ``i * world // n_items``."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    # smallest i with i*world//n >= rank  ==  ceil(rank*n/world)
    start = -((-rank * n_items) // world)
    stop = -((-(rank + 1) * n_items) // world)
    return start, stop


class FrameShardRunner:
    """Runs ``synth_fn`` over this rank's block of frames in batches and gathers the uint8 frames on rank 0.

    ``synth_fn(shared, frame_inputs) -> uint8 [n, H, W, 3]`` is the per-batch compute (on the GPU box:
    ``gen_img`` + ``ops.tensor2im_u8``); everything else here is host logic + collectives and is covered by the
    world_size-2 gloo tests on CPU."""

    def __init__(self, device: Optional[torch.device] = None, group=None):
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.device = device if device is not None else torch.device("cpu")
        self._pipes = {}          # stream count -> the StreamPipeline run_clip_streamed reuses (close() drops them)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:          # (interpreter shutdown: the streams are gone already)
            pass

    # ---- collectives ---------------------------------------------------------------------------------------------
    def broadcast_shared(self, tensor: Optional[torch.Tensor], shape: Sequence[int], dtype=torch.float32, src: int = 0) -> torch.Tensor:
        """Rank ``src`` provides ``tensor``; every rank returns its copy on ``self.device``."""
        if self.rank == src:
            if tensor is None or tuple(tensor.shape) != tuple(shape):
                raise ValueError(f"source rank must provide a tensor of shape {tuple(shape)}")
            buf = tensor.to(device=self.device, dtype=dtype).contiguous()
        else:
            buf = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        if self.distributed and self.world > 1:
            dist.broadcast(buf, src=src, group=self.group)
        return buf

    def gather_frames(self, local: torch.Tensor, n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
        """``local`` = this rank's frames ``[n_local, ...]`` (block ``shard_range``); rank ``dst`` gets ``[n_total, ...]``
        in frame order, the others ``None``."""
        start, stop = shard_range(n_total, self.rank, self.world)
        if local.shape[0] != stop - start:
            raise ValueError(f"rank {self.rank} owns {stop - start} frames, got {local.shape[0]}")
        if not self.distributed or self.world == 1:
            return local
        n_max = max(shard_range(n_total, r, self.world)[1] - shard_range(n_total, r, self.world)[0] for r in range(self.world))
        pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        if self.rank == dst:
            bufs = [torch.empty_like(pad) for _ in range(self.world)]
            dist.gather(pad, bufs, dst=dst, group=self.group)
            parts = []
            for r, b in enumerate(bufs):
                s, e = shard_range(n_total, r, self.world)
                parts.append(b[: e - s])
            return torch.cat(parts, dim=0)
        dist.gather(pad, None, dst=dst, group=self.group)
        return None

    def max_over_ranks(self, value: float) -> float:
        if not self.distributed or self.world == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.distributed and self.world > 1:
            dist.barrier(group=self.group)

    # ---- the clip loop -------------------------------------------------------------------------------------------
    def run_clip_streamed(self, n_frames: int, shared, frame_inputs: Callable[[int, int], object],
                          synth_fn: Callable[[object, object], torch.Tensor], batch: int = 4, dst: int = 0,
                          out: Optional[torch.Tensor] = None, streams: int = 2, full_batches: bool = True) -> Optional[torch.Tensor]:
        """Like ``run_clip`` but the finished frames travel to rank ``dst`` batch by batch while the next batch is being computed, instead
        of in one padded gather at the end (face_swap_video_pipeline.py:404-486 writes each frame out as soon as it is done).

        Round ``k`` = every rank's ``k``-th batch of its own block.  All ranks issue the same sequence of ``gather`` calls (one per round,
        ``async_op``: the collective runs on the backend's own stream); a rank whose block has no ``k``-th batch, or a short last one,
        sends padding.  Rank ``dst`` scatters each round's buffers into ``out [n_frames, H, W, 3]`` (allocated on first use; pass a
        preallocated one to keep the allocation out of a timed region) after the round's gather has completed — at the latest when the
        round after the next two is issued: one computed round waiting to be sent plus up to three sent rounds whose gathers have not been consumed, i.e. at
        most FOUR rounds of frames (4 x batch x frame bytes per rank, world x that on rank ``dst``) are alive at a time — size ``out=`` / memory against that.

        Batch composition: the encoder picks its convolution kernels from the launch size (``ops.winograd_route`` / ``ops.mx_conv_eligible``), so the same face
        could differ between a short and a full batch by up to 5e-5 of the largest style-vector entry (tests/test_gpu_encoder.py).  The loop therefore never
        launches a short batch when its block holds a full one (``full_batches``, default): a block's last, short round is computed on the block's LAST ``batch``
        frames (re-computing up to ``batch - 1`` of them) and only the new ones are sent — every frame of a clip goes through the same kernels whatever the clip
        length and the world size.  (``E4S_ENC_ROUTE_BY_IMAGE=1`` makes the route choice per image instead: bit-identical at ANY batch, 5-10 % of the encoder's throughput.)

        ``streams`` (GPU only): consecutive rounds run on that many alternating HIP streams (``StreamPipeline``): the latency-bound small layers
        of one batch overlap the large ones of the batch before (+4–6 % swaps/s at batch 8, ``tools/time_swap_pipeline.py``); a round's gather is
        issued from the round's own stream, so it waits for that round's frames only."""
        start, stop = shard_range(n_frames, self.rank, self.world)
        blocks = [shard_range(n_frames, r, self.world) for r in range(self.world)]
        rounds = max(-(-(e - s) // batch) for s, e in blocks) if n_frames > 0 else 0
        pending = []          # (work, send buffer, receive buffers, round, event recorded on the round's stream)

        def finish(item):
            nonlocal out
            work, snd0, bufs, k, ev = item
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)     # the round ran on its own stream
            if work is not None:
                work.wait()
            if self.rank != dst:
                return
            for r, (s, e) in enumerate(blocks):
                lo = s + k * batch
                hi = min(lo + batch, e)
                if hi > lo:
                    src = bufs[r] if bufs is not None else snd0
                    if out is None:
                        out = torch.empty((n_frames,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
                    out[lo:hi].copy_(src[: hi - lo])
                    if src.is_cuda:
                        src.record_stream(torch.cuda.current_stream())     # (allocated on the round's stream, read here)

        shape = None
        on_gpu = torch.device(self.device).type == "cuda"
        if on_gpu:
            from . import ops
        # A round is computed (``compute``) one loop turn before its frames are sent (``send``): the host looks at the round's f16 range guard
        # (``ops.MxGuard``; a wait for that round) while the NEXT round is already queued on the other stream, so the wait costs no GPU time, and a
        # round whose arithmetic overflowed is recomputed in split-bf16 before its frames leave.  Every rank issues the same sequence of gathers.

        def compute(k, exact=False):
            nonlocal shape
            lo = start + k * batch
            hi = min(lo + batch, stop)
            frames = guard = None

            def synth(a, b):
                if not on_gpu:
                    return synth_fn(shared, frame_inputs(a, b)), None
                with ops.mx_guard_scope() as g:
                    if exact:
                        with ops.mx_exact():
                            f = synth_fn(shared, frame_inputs(a, b))
                    else:
                        f = synth_fn(shared, frame_inputs(a, b))
                    g.arm()
                return f, g
            if hi > lo:
                lo2 = max(start, hi - batch) if (full_batches and hi - lo < batch) else lo      # (a short last round: the block's last `batch` frames, see the docstring)
                frames, guard = synth(lo2, hi)
                if frames.dtype != torch.uint8 or frames.shape[0] != hi - lo2:
                    raise ValueError("synth_fn must return uint8 frames [n, H, W, 3] for the requested block")
                if lo2 != lo:
                    frames = frames[lo - lo2:]
                shape = tuple(frames.shape[1:])
            elif shape is None:     # this rank has no frames at all: learn the frame shape from a probe, send padding
                shape = tuple(synth(0, min(1, n_frames))[0].shape[1:])
            return frames, guard, k, (torch.cuda.current_stream() if on_gpu else None)

        def send(item):
            frames, guard, k, st = item
            ctx = torch.cuda.stream(st) if st is not None else _null_context()
            with ctx:
                if guard is not None and guard.tripped():
                    ops.mx_fallbacks += 1
                    frames = compute(k, exact=True)[0]
                if frames is not None and frames.shape[0] == batch:
                    snd = frames.contiguous()
                else:
                    snd = torch.zeros((batch,) + shape, dtype=torch.uint8, device=self.device)
                    if frames is not None:
                        snd[: frames.shape[0]] = frames
                if self.distributed and self.world > 1:
                    bufs = [torch.empty_like(snd) for _ in range(self.world)] if self.rank == dst else None
                    work = dist.gather(snd, bufs, dst=dst, group=self.group, async_op=True)       # (ordered after this round's stream)
                else:
                    bufs, work = None, None
                ev = None
                if snd.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record()
            return work, snd, bufs, k, ev

        # The runner keeps ONE pipeline per stream count for its lifetime.  (A new pipeline per clip drew new streams from torch's pool of 32 every time, and the caching
        # allocator keeps a pool of freed blocks per stream: eight clips in one process left 65 GB reserved for 2.7 GB in use.  ``close()`` gives the streams' host
        # contexts back.)
        key = streams if on_gpu else 1
        pipe = self._pipes.get(key)
        if pipe is None:
            pipe = self._pipes[key] = StreamPipeline(key, device=self.device if on_gpu else None)
            pipe.own_guards = False          # (every round takes its own guard: `compute`)
        with pipe:
            computed = None
            for k in range(rounds):
                nxt = pipe.submit(compute, k)
                if computed is not None:
                    pending.append(send(computed))
                    if len(pending) > 2:
                        finish(pending.pop(0))
                computed = nxt
            if computed is not None:
                pending.append(send(computed))
        while pending:                          # (after the pipeline's exit the current stream has waited for every round's stream)
            finish(pending.pop(0))
        return out if self.rank == dst else None

    def close(self):
        """Drop the streamed loop's pipelines: their side streams give their host contexts (128 MB of split-K workspace each) back."""
        for pipe in self._pipes.values():
            pipe.close()
        self._pipes.clear()

    def run_clip(self, n_frames: int, shared: torch.Tensor, frame_inputs: Callable[[int, int], object],
                 synth_fn: Callable[[torch.Tensor, object], torch.Tensor], batch: int = 4, dst: int = 0) -> Optional[torch.Tensor]:
        """Synthesise frames ``[0, n_frames)``: this rank takes its block, walks it in batches of ``batch`` calling
        ``synth_fn(shared, frame_inputs(lo, hi))`` and the uint8 results are gathered on rank ``dst``."""
        start, stop = shard_range(n_frames, self.rank, self.world)
        outs = []
        for lo in range(start, stop, batch):
            hi = min(lo + batch, stop)
            frames = synth_fn(shared, frame_inputs(lo, hi))
            if frames.dtype != torch.uint8 or frames.shape[0] != hi - lo:
                raise ValueError("synth_fn must return uint8 frames [n, H, W, 3] for the requested block")
            outs.append(frames)
        if outs:
            local = torch.cat(outs, dim=0)
        else:  # a rank can own zero frames when n_frames < world: it still takes part in the gather
            probe = synth_fn(shared, frame_inputs(0, min(1, n_frames)))
            local = probe[:0]
        return self.gather_frames(local, n_frames, dst=dst)


class _null_context:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _Verdict:
    """What is left of a checked ``ops.MxGuard`` in ``StreamPipeline.guards``: the answer, without the guard's pinned words and event."""
    __slots__ = ("moved",)

    def __init__(self, moved: bool):
        self.moved = bool(moved)

    def tripped(self) -> bool:
        return self.moved

    def arm(self):
        return self


class StreamPipeline:
    """Independent batches on alternating HIP streams of one GPU.

    A batch through ``gen_img`` starts with the 4²–32² layers: a dozen small launches that depend on one another and leave most of the chip
    idle (≈ 0.4 of 3.5 ms at batch 4).  Batches are independent units (the reference loops over frames, ``face_swap_video_pipeline.py:406``),
    so the next batch's latency-bound head can run under the previous batch's large layers: ``submit`` puts consecutive calls on ``n`` streams
    round-robin — kernels of one call stay in order on their stream, calls on different streams overlap.  Measured (``tools/time_pipeline.py``,
    1024² synthesis): 1 127 → 1 221 faces/s at batch 4 with two streams (1 233 with three), 1 169 → 1 248 at batch 8; outputs bit-identical.

    ``with StreamPipeline(2) as sp: outs = [sp.submit(fn, x) for x in batches]`` — on entry the side streams wait for the work already queued
    on the current stream (inputs are ready), on exit the current stream waits for all of them (outputs are ready for whatever follows).  The
    host state of the ops (split-K workspace, control words) is per stream (``ops._StreamCtx``).  ``n <= 1``: plain calls on the current stream.
    Tensors returned by ``submit`` (directly or in a tuple / list) are marked as used by the entering stream (``record_stream``), so they can be read
    and dropped there after the ``with`` block like any other tensor."""

    def __init__(self, n: int = 2, device=None):
        self.n = int(n)
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.n)] if self.n > 1 else []
        self._i = 0
        self._main = None
        self.guards = []          # one ops.MxGuard / verdict (or None) per submitted call since the last __enter__
        self.healed = []          # indices of the calls this pipeline re-ran in the exact arithmetic (results copied into the returned tensors)
        self._pending = []        # (index, guard, stream, fn, args, kwargs, out) of the calls whose guard nobody has looked at yet
        self.own_guards = True    # False: the submitted functions bracket their own work (run_clip_streamed's rounds do)

    def __enter__(self):
        self._main = None
        self.guards = []
        self.healed = []
        self._pending = []
        if self.streams:
            self._main = main = torch.cuda.current_stream()
            for st in self.streams:
                st.wait_stream(main)
        return self

    def _hand_over(self, out, depth=0):
        # results are read on the entering stream after the block: tell the allocator, or a block freed there could be handed out again on its
        # own stream while that read is still queued
        if isinstance(out, torch.Tensor):
            if out.is_cuda and self._main is not None:
                out.record_stream(self._main)
        elif isinstance(out, (tuple, list)) and depth < 2:
            for o in out:
                self._hand_over(o, depth + 1)

    MAX_PENDING = 32          # unchecked guards kept alive at most (each holds two pinned words, an event and the call's arguments)

    def submit(self, fn, *args, **kwargs):
        """f16 range (``ops.MxGuard``): a drop-in module called without a guard scope checks its own pass — one host synchronisation per call, which would
        serialise the batches this pipeline is there to overlap.  ``submit`` therefore owns the scope of every call that is not inside one already, and the
        pipeline stays SELF-HEALING by default: a call whose f16 arithmetic left its range is re-run under ``ops.mx_exact()`` on its own stream and the exact
        results are copied INTO the tensors ``submit`` returned (same shapes: the caller's references stay valid) — lazily, for calls whose guard has already
        landed when a later ``submit`` looks (no host wait), at the latest in ``__exit__``.  ``self.guards`` holds one entry per submitted call (``None`` where
        the caller's own scope was in force); a caller that prefers to do the re-run itself checks ``sp.guards[i].tripped()`` / ``tripped_calls()`` before the
        block ends — a guard the caller has looked at is the caller's, the pipeline leaves that call alone."""
        if not self.streams:
            return fn(*args, **kwargs)
        from . import ops
        self._reap(block=len(self._pending) >= self.MAX_PENDING)
        st = self.streams[self._i % self.n]
        self._i += 1
        g = None
        with torch.cuda.stream(st):
            if not self.own_guards or ops.mx_guard_owned():
                out = fn(*args, **kwargs)
                self.guards.append(None)
            else:
                with ops.mx_guard_scope() as g:
                    out = fn(*args, **kwargs)
                    g.arm()
                self.guards.append(g)
        if g is not None and g._live:
            self._pending.append((len(self.guards) - 1, g, st, fn, args, kwargs, out))
        self._hand_over(out)
        return out

    @staticmethod
    def _copy_into(dst, src, depth=0):
        if isinstance(dst, torch.Tensor):
            dst.copy_(src)
        elif isinstance(dst, (tuple, list)) and depth < 3:
            for d, s_ in zip(dst, src):
                StreamPipeline._copy_into(d, s_, depth + 1)
        elif isinstance(dst, dict) and depth < 3:
            for k in dst:
                StreamPipeline._copy_into(dst[k], src[k], depth + 1)

    def _reap(self, block: bool = False, everything: bool = False):
        """Check the pending guards in submission order: those whose "after" snapshot has landed (all of them when ``everything``; the oldest one in any case
        when ``block``), heal the calls that tripped, and replace each checked guard by its verdict so that its pinned words and event go away."""
        from . import ops
        while self._pending:
            idx, g, st, fn, args, kwargs, out = self._pending[0]
            if g._done:                                    # the caller looked at it: the caller's call
                self._pending.pop(0)
                continue
            if not (everything or block or g._ev.query()):
                break
            block = False
            self._pending.pop(0)
            moved = g.tripped()
            if moved:
                ops.mx_fallbacks += 1
                with torch.cuda.stream(st), torch.no_grad(), ops.mx_exact(), ops.mx_guard_scope():
                    self._copy_into(out, fn(*args, **kwargs))
                self.healed.append(idx)
            self.guards[idx] = _Verdict(moved)

    def tripped_calls(self):
        """Indices (in submission order since the pipeline was entered) of the calls whose f16 arithmetic left its range and which the pipeline has NOT healed
        (``self.healed`` lists the ones it has): their results must be recomputed under ``ops.mx_exact()`` by the caller.  Waits for each call's guard (a host
        synchronisation with that call's stream)."""
        return [i for i, g in enumerate(self.guards) if g is not None and i not in self.healed and g.tripped()]

    def __exit__(self, *exc):
        if self.streams:
            if exc[0] is None:
                self._reap(everything=True)                # (waits for each unchecked guard's snapshot: the block's calls are complete behind it)
            self._pending.clear()
            main = torch.cuda.current_stream()
            for st in self.streams:
                main.wait_stream(st)
        return False

    def close(self):
        """Give back what the side streams hold on the host side (each stream's context pins a 128 MB split-K workspace): waits for them, then drops
        their contexts.  For long-running processes that create pipelines repeatedly; a pipeline can be re-entered after ``close`` (contexts are
        re-created on demand).  Do not call while a hipGraph captured on one of these streams is still in use."""
        from . import ops
        for st in self.streams:
            st.synchronize()
            ops.release_stream_context(st)


def gen_img_frames(net, codes: torch.Tensor, labels: torch.Tensor, randomize_noise: bool = False) -> torch.Tensor:
    """Per-batch compute for the MI355X path: ``Net3.gen_img`` on uint8 region maps ``[n, 512, 512]`` (or one-hot masks)
    followed by the device-side ``tensor2im``.  ``codes`` is ``[1 or n, 12, 18, 512]`` (shared codes are expanded)."""
    from . import ops
    n = labels.shape[0]
    if codes.shape[0] == 1 and n > 1:
        codes = codes.expand(n, -1, -1, -1)
    with torch.no_grad():
        img, _, _ = net.gen_img(None, codes, labels, randomize_noise=randomize_noise)
    return ops.tensor2im_u8(img)
