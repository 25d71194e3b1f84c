"""PTI fine-tuning step on the drop-in ``Net3`` (SURVEY §8 row f1 / BASELINE configs[3]).

Mirrors one inner iteration of ``VideoSwapPTICoach.train_e4s`` (training/video_swap_ft_coach.py:253-299) for the part that lives on the
hot path: ``cal_style_codes`` -> ``gen_img`` -> pixel loss -> ``backward`` -> optimiser step, with the style vectors and region map of a
frame as fixed inputs.  The forward runs on the fused HIP kernels; the backward of the synthesis layers differentiates from their
outputs with the gradient kernels of ``csrc/modconv_bwd.hip`` + ``csrc/gemm_sb.hip`` (``ops._MaskedStyledConvGrad`` /
``ops._SingleStyledConvGrad``), only the small per-layer style tables go through autograd (``torch_ref.py``).  The perceptual /
identity / parsing losses of ``calc_loss`` (:176-223) are separate networks outside the path (LPIPS-alex, ArcFace, a UNet parser) and
plug in through ``extra_loss``.

Several GPUs: one process per GPU, each on its own frame; the one exchange step is the gradient average before the
optimiser step (``sync_gradients``: a few large flat all-reduces over RCCL, not one per tensor).  That is a batch-of-N Adam step, not
the reference's N sequential batch-1 steps, so result parity with the reference is not claimed for the multi-GPU mode.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops


def trainable_parameters(net):
    """``configure_optimizer`` (video_swap_ft_coach.py:171-177): every parameter the constructor left ``requires_grad=True``."""
    return [p for p in net.parameters() if p.requires_grad]


def _loss(net, style_vectors, mask, target, foreground_mask, l2_lambda, extra_loss, randomize_noise):
    codes = net.cal_style_codes(style_vectors)
    recon, _, _ = net.gen_img(None, codes, mask, randomize_noise=randomize_noise)
    a, b = (recon, target) if foreground_mask is None else (recon * foreground_mask, target * foreground_mask)
    loss = l2_lambda * F.mse_loss(a, b)                                      # calc_loss :196-199 (loss_l2)
    if extra_loss is not None:
        loss = loss + extra_loss(recon, target)
    return loss, recon


def sync_gradients(params, group=None, bucket_bytes: int = 256 << 20, active_ranks: Optional[int] = None) -> int:
    """Average ``p.grad`` over the ranks of ``group`` in place: gradients are packed into flat buckets of about ``bucket_bytes``
    (xGMI rings are per-link bound, so few large all-reduces: SURVEY §8e), one ``all_reduce`` each, launched back to back and waited
    for together.  Which parameters take part is agreed on first — one small MAX all-reduce of a has-gradient flag per parameter: a
    parameter that NO rank has a gradient for (the optimiser's list follows the reference and includes the whole encoder, which the PTI
    loss never reaches) is skipped exactly as a single-GPU step skips it — no zero gradient, no Adam state, no xGMI traffic; one that only
    some ranks have a gradient for contributes zeros from the others (every rank must issue the same collectives).
    ``active_ranks``: divide the summed gradients by this number instead of the world size (a round of ``tune_clip`` in which only some
    ranks still have a frame).  Returns the number of gradient all-reduces issued; a no-op (0) outside a process group or at world size 1."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    flags = torch.tensor([0 if p.grad is None else 1 for p in params], dtype=torch.int32, device=params[0].device)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    params = [p for p, f in zip(params, flags.tolist()) if f]
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nbytes = p.numel() * p.element_size()
        if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    pending = []
    denom = float(world if active_ranks is None else max(int(active_ranks), 1))
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, bucket))
    for work, flat, bucket in pending:
        work.wait()
        flat.div_(denom)
        off = 0
        for p in bucket:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
    return len(buckets)


class GraphedPTIStep:
    """The whole optimiser step (forward, backward, Adam update) captured once as a hipGraph and replayed per frame: at batch 1 the
    eager step issues several thousand short launches and is bound by the host (~0.12 s) rather than by the GPU.

    ``optimizer`` must be capture-safe (``torch.optim.Adam(..., capturable=True, fused=True)``: the fused multi-tensor update is 6 ms
    per step cheaper than the default one over the generator's 263 tensors); shapes are fixed by the example inputs;
    ``mask`` must be a uint8 region map ``[bs, 512, 512]`` (the one-hot check of a float mask reads a flag back to the host).
    The ``warmup`` eager steps that precede the capture are real optimiser steps on the example frame.  Weight re-layout kernels are
    part of the captured step (the parameters change under them), so every replay prepares its weights from their current values."""

    def __init__(self, net, optimizer, style_vectors, mask, target, foreground_mask=None, l2_lambda: float = 1.0, extra_loss=None,
                 randomize_noise: bool = True, warmup: int = 3, warm_inputs=None):
        """``warm_inputs``: the frames of the eager steps that precede the capture, as ``(style_vectors, mask, target[, foreground_mask])``
        tuples (default: the example frame ``warmup`` times).  They are real optimiser steps: a loop passes its own first frames here
        (``tune_clip``) and continues with the replayed step from the next one; ``self.warm_losses`` holds their losses."""
        if mask.dtype != torch.uint8:
            raise TypeError("GraphedPTIStep needs the uint8 region map (ops.mask_to_labels(onehot)), not a float mask")
        self.net = net
        self.static = [style_vectors.clone(), mask.clone(), target.clone()] + ([foreground_mask.clone()] if foreground_mask is not None else [])
        fg = self.static[3] if foreground_mask is not None else None
        args = (net, self.static[0], self.static[1], self.static[2], fg, l2_lambda, extra_loss, randomize_noise)
        self.stream = torch.cuda.Stream()                 # warm-up and capture on one stream of our own (see graphs.GraphedCall)
        ops.prepare_stream_context(self.stream)
        self.stream.wait_stream(torch.cuda.current_stream())
        self.warm_losses = []
        with torch.cuda.stream(self.stream):
            for w in (warm_inputs if warm_inputs is not None else [None] * warmup):
                if w is not None:
                    for dst, src in zip(self.static, w):
                        dst.copy_(src)
                optimizer.zero_grad(set_to_none=True)
                loss, _ = _loss(*args)
                loss.backward()
                optimizer.step()
                self.warm_losses.append(loss.detach().clone())
            if warm_inputs is not None:                       # the capture's example inputs back in the static buffers
                for dst, src in zip(self.static, [style_vectors, mask, target] + ([foreground_mask] if foreground_mask is not None else [])):
                    dst.copy_(src)
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.loss, self.recon = _loss(*args)
            self.loss.backward()
            optimizer.step()
        ops.invalidate_weight_caches(net)

    def __call__(self, style_vectors, mask, target, foreground_mask=None):
        """Copies the frame into the static buffers and replays the step; returns the (static) loss and reconstruction tensors."""
        new = [style_vectors, mask, target] + ([foreground_mask] if foreground_mask is not None else [])
        if len(new) != len(self.static):
            raise ValueError("foreground_mask must be given iff the step was captured with one")
        for dst, src in zip(self.static, new):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # The replay has just moved the parameters (fused Adam inside the graph) without running any Python forward and without bumping a
        # tensor version: a re-laid-out copy cached by an earlier no_grad forward (a preview between steps) would otherwise still match
        # its key and render with the old weights.  Forgetting the copies costs a few attribute writes; the next eval rebuilds them.
        ops.invalidate_weight_caches(self.net)
        return self.loss, self.recon


def pti_step(net, optimizer: torch.optim.Optimizer, style_vectors: torch.Tensor, mask: torch.Tensor, target: torch.Tensor,
             foreground_mask: Optional[torch.Tensor] = None, l2_lambda: float = 1.0,
             extra_loss: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None, group=None):
    """One optimiser step.  ``style_vectors [bs, 12, 1280]``, ``mask`` one-hot ``[bs, 12, 512, 512]`` (or uint8 labels),
    ``target [bs, 3, 1024, 1024]`` in [-1, 1]; ``foreground_mask [bs, 1, 1024, 1024]`` restricts the loss as at :283-288.
    Inside a ``torch.distributed`` process group every rank passes its own frame and the gradients are averaged before the update.
    Returns ``(loss value, reconstruction)``."""
    loss, recon = _loss(net, style_vectors, mask, target, foreground_mask, l2_lambda, extra_loss, True)   # the coach calls gen_img with fresh noise
    optimizer.zero_grad()
    loss.backward()
    sync_gradients([p for g in optimizer.param_groups for p in g["params"]], group)
    optimizer.step()
    return loss.detach(), recon.detach()


def style_vector_step(net, optimizer: torch.optim.Optimizer, latent: torch.Tensor, mask: torch.Tensor, target: torch.Tensor,
                      l2_lambda: float = 1.0, extra_loss: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
                      randomize_noise: bool = True):
    """One step of the reference's W-optimisation (``Optimizer.optim_W_online``, optimization.py:321-349): the per-region style
    vectors ``latent [bs, 12, 1280]`` (``requires_grad``, held by ``optimizer``) are tuned so that ``gen_img(cal_style_codes(latent))``
    matches ``target``; the network's own parameters are left alone (they are not in ``optimizer``)."""
    optimizer.zero_grad()
    loss, recon = _loss(net, latent, mask, target, None, l2_lambda, extra_loss, randomize_noise)
    loss.backward()
    optimizer.step()
    return loss.detach(), recon.detach()


# ------------------------------------------------------------------------------------------------ the PTI loop over a clip (BASELINE configs[3])
def prepare_clip(labels: torch.Tensor, erode_radius: Optional[int] = None, size=(1024, 1024)):
    """Per-frame masks of the PTI loop (training/video_swap_ft_coach.py:257-279), computed once for the clip on the device:
    the region map the synthesis runs on — ``erode_mask(driven_m, radius)`` when the coach erodes (:259-263), else the map itself — and the
    loss's foreground weight, ``not {background, hair, ear-rings}`` of THAT map, bilinearly resized to 1024 x 1024 (:277-280).
    ``labels``: uint8 ``[n, 512, 512]`` 12-class maps; ``size``: the images' (H, W).  Returns ``(maps uint8 [n, 512, 512], foreground float [n, 1, H, W])``."""
    maps = ops.erode_labels(labels, erode_radius) if erode_radius else ops._labels_u8(labels, "labels")
    fg = torch.ones_like(maps, dtype=torch.float32)
    for c in ops.PTI_BG_CLASSES:
        fg = fg * (maps != c)
    fg = ops.bilinear_resize(fg[:, None].contiguous(), tuple(size), align_corners=False)
    return maps, fg


def tune_clip(net, optimizer, images: torch.Tensor, labels: torch.Tensor, style_vectors: torch.Tensor, steps: int,
              erode_radius: Optional[int] = None, l2_lambda: float = 1.0, extra_loss=None, group=None, graphed: Optional[bool] = None,
              randomize_noise: bool = True, step_fn=None, on_epoch=None, local_only: bool = False):
    """The fine-tuning loop of ``VideoSwapPTICoach.train_e4s`` (training/video_swap_ft_coach.py:242-317) for the part on the hot path:
    ``steps`` passes over the clip's frames, one optimiser step per frame — ``cal_style_codes`` -> ``gen_img`` on the (eroded) region map ->
    L2 against the frame under the foreground weight (+ ``extra_loss`` for the perceptual / identity / parsing nets of ``calc_loss``) ->
    backward -> Adam.  ``images [n, 3, 1024, 1024]`` in [-1, 1], ``labels`` uint8 ``[n, 512, 512]``, ``style_vectors [n, 12, 1280]``.

    Several GPUs (BASELINE configs[3]: 32 frames on 4 GPUs): every rank holds the whole clip's inputs or at least its own block; rank ``r``
    tunes on frames ``shard_range(n, r, world)`` and round ``i`` of a pass = every rank's ``i``-th frame, gradients averaged over the ranks
    that still have one (``sync_gradients``) — a batch-of-N step, so the parameters stay identical on all ranks (not the reference's N
    sequential steps: no parity claim for this mode).  A rank whose block is shorter takes part in the remaining rounds with no gradient.

    Single GPU: the step runs as one replayed hipGraph (``GraphedPTIStep``) unless ``graphed=False``.
    ``local_only``: ignore an initialised process group (this rank tunes on the frames it is given, no collective) — the pre-flight step of
    ``bench.py``'s multi-GPU section.
    ``step_fn(net, optimizer, vec, map, image, fg, group, active) -> loss`` replaces the step (tests).  Returns the mean loss of each pass."""
    from .runner import shard_range
    distributed = (not local_only) and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    n = images.shape[0]
    if labels.shape[0] != n or style_vectors.shape[0] != n:
        raise ValueError("tune_clip: images, labels and style vectors must describe the same frames")
    lo, hi = shard_range(n, rank, world)
    rounds = max(shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world))
    if step_fn is None:
        maps, fgs = prepare_clip(labels[lo:hi].contiguous(), erode_radius, images.shape[-2:])
    else:
        maps, fgs = labels[lo:hi], [None] * (hi - lo)
    if graphed is None:
        graphed = not distributed and step_fn is None and hi > lo
    params = [p for g in optimizer.param_groups for p in g["params"]]
    step = None
    if graphed and distributed:
        raise ValueError("tune_clip: the graph-captured step has no gradient exchange; use graphed=False with several ranks")
    # A capture needs an initialised optimiser state, library handles and allocator pools on its stream: GraphedPTIStep runs warm-up steps
    # eagerly on the capture stream first.  Here those ARE the loop's first two steps (frames 0, 1 of the first pass), not extra ones.
    EAGER_FIRST = min(2, steps * rounds - 1) if graphed else 0
    sched = [(e, i) for e in range(steps) for i in range(rounds)]
    warm_losses = []
    if graphed:
        frame = lambda i: (style_vectors[lo + i:lo + i + 1], maps[i:i + 1], images[lo + i:lo + i + 1], fgs[i:i + 1])   # noqa: E731
        ex = frame(sched[EAGER_FIRST][1])
        step = GraphedPTIStep(net, optimizer, ex[0], ex[1], ex[2], ex[3], l2_lambda, extra_loss, randomize_noise,
                              warm_inputs=[frame(i) for _, i in sched[:EAGER_FIRST]])
        warm_losses = list(step.warm_losses)
    done = 0
    history = []
    for epoch in range(steps):
        losses = []
        for i in range(rounds):
            f = lo + i
            have = f < hi
            active = sum(1 for r in range(world) if shard_range(n, r, world)[0] + i < shard_range(n, r, world)[1])
            done += 1
            if graphed and done <= EAGER_FIRST:                # already taken (eagerly, on the capture stream)
                losses.append(warm_losses[done - 1])
                continue
            if step_fn is not None:
                loss = step_fn(net, optimizer, style_vectors[f:f + 1] if have else None, maps[i:i + 1] if have else None,
                               images[f:f + 1] if have else None, None, group, active)
            elif step is not None:
                loss = step(style_vectors[f:f + 1], maps[i:i + 1], images[f:f + 1], fgs[i:i + 1])[0].clone()
            else:
                optimizer.zero_grad(set_to_none=True)
                loss = None
                if have:
                    loss, _ = _loss(net, style_vectors[f:f + 1], maps[i:i + 1], images[f:f + 1], fgs[i:i + 1], l2_lambda, extra_loss, randomize_noise)
                    loss.backward()
                if distributed:
                    sync_gradients(params, group, active_ranks=active)
                optimizer.step()
            if loss is not None:
                losses.append(loss.detach() if isinstance(loss, torch.Tensor) else torch.tensor(float(loss)))
        mean = torch.stack([l.float().reshape(()) for l in losses]).mean().item() if losses else float("nan")
        history.append(mean)
        if on_epoch is not None:
            on_epoch(epoch, mean)
    return history
