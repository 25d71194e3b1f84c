"""Batched full-swap unit on the device: per face
    2 x parse (BiSeNet, 19 -> 12 classes) + 2 x get_style_vectors + style-vector mix + cal_style_codes + gen_img (+ tensor2im)
which is what ``face_swap_video_pipeline.py`` does per frame between its CPU stages (:212-219 parsing, :332-354 style vectors,
:429-443 mix + synthesis), here for a whole batch without leaving the GPU.  With ``mask_surgery=True`` the synthesis is driven by
``swap_head_mask_hole_first(driven_map, target_map)`` as at face_swap_video_pipeline.py:420 (row f2, ``ops.swap_head_mask``) and the
paste-back masks of :456-463 (row f3, ``ops.foreground_masks``) are returned too; the default keeps BASELINE configs[2]'s unit of
work, where the target's own region map drives the synthesis.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch

from . import ops

DEFAULT_COMP_INDICES = tuple(sorted(set(range(12)) - {0, 4, 11}))      # face_swap_video_pipeline.py:436: keep target background, hair, ear-rings


TWO_STREAMS = True               # (module attributes; ``swap_batch`` also takes them as arguments)
_sel_cache = {}
_side = {}


SWAP_CHAINS = 2     # 2 = driven | target; 4 = each additionally split into half batches
# Driven and target faces as ONE batch of 2*bs through parser and encoder (default): the 32x32 and 16x16 stages of the encoder and the
# launch-bound glue between its convolutions fill the chip better at twice the batch than as two concurrent chains (measured at bs 8:
# parse + encode 4.3 + 13.7 ms for 16 images against 2 x (2.5 + 8.1) ms on one stream and 21.0 ms on two streams).
SWAP_BATCHED = True
PARSE_BESIDE_ENCODE = True    # (batched route) the face parser on a side stream next to the encoder body
ENCODE_ISSUED_FIRST = os.environ.get("E4S_SWAP_ENC_FIRST", "1") != "0"     # (with PARSE_BESIDE_ENCODE) host issue order: encoder body, then the parser


def _side_stream(device, idx=0):
    key = (str(device), idx)
    st = _side.get(key)
    if st is None:
        st = _side[key] = torch.cuda.Stream(device=device)
    return st


def _selector(device, comp_indices, n):
    """bool [n] on ``device`` marking the components taken from the driven face; built once per (device, indices) so that no
    host-to-device copy happens inside a hipGraph capture."""
    key = (str(device), tuple(comp_indices), n)
    sel = _sel_cache.get(key)
    if sel is None:
        m = torch.zeros(n, dtype=torch.bool)
        m[list(comp_indices)] = True
        sel = _sel_cache[key] = m.to(device)
    return sel


def mix_style_vectors(target_vec: torch.Tensor, driven_vec: torch.Tensor, comp_indices: Sequence[int] = DEFAULT_COMP_INDICES,
                      below_face_interpolation: bool = False) -> torch.Tensor:
    """``swap_comp_style_vector`` (swap_face_fine/swap_face_mask.py:336-367) for a batch, without host synchronisation:
    take the listed components from the driven face; ears (7) = mean of both; ear-rings (11) from the target; neck (8) optionally the
    mean; teeth (9) from the target when the driven face has none (its style vector sums to exactly 0)."""
    sel = _selector(target_vec.device, comp_indices, target_vec.shape[1])
    out = torch.where(sel[None, :, None], driven_vec, target_vec)
    out[:, 7, :] = (target_vec[:, 7, :] + driven_vec[:, 7, :]) / 2
    out[:, 11, :] = target_vec[:, 11, :]
    if below_face_interpolation:
        out[:, 8, :] = (target_vec[:, 8, :] + driven_vec[:, 8, :]) / 2
    no_teeth = (driven_vec[:, 9, :].sum(dim=1, keepdim=True) == 0)
    out[:, 9, :] = torch.where(no_teeth, target_vec[:, 9, :], out[:, 9, :])
    return out


@torch.no_grad()
def swap_batch(net, parser, driven: torch.Tensor, target: torch.Tensor, comp_indices: Sequence[int] = DEFAULT_COMP_INDICES,
               randomize_noise: bool = False, to_uint8: bool = True, timings: Optional[dict] = None, mask_surgery: bool = False,
               paste_radius: int = 5, two_streams: Optional[bool] = None, batched: Optional[bool] = None, guard: Optional[list] = None):
    """``driven`` / ``target``: ``[bs, 3, 1024, 1024]`` in [-1, 1] on the device.  Returns uint8 ``[bs, 1024, 1024, 3]`` frames
    (or the float image) and the 12-class region maps the synthesis used; with ``mask_surgery`` a third value
    ``{"hole_mask", "hole_map", "lines", "content", "border", "full"}`` (the reference's paste-back inputs, :456-463).

    f16 range (``ops.MxGuard``): parser, encoder and the masked synthesis layers run in f16-based split arithmetic.  The batch is bracketed by ONE
    guard; by default the call waits for it at its end and re-runs the whole batch in the split-bf16 arithmetic if a value left the f16 range
    (``ops.mx_fallbacks`` counts those).  Callers that keep several batches in flight pass ``guard=[]``: the armed guard is appended instead of
    awaited, and the caller checks ``guard[-1].tripped()`` where it synchronises anyway, repeating the call under ``with ops.mx_exact():``."""
    args = (net, parser, driven, target, comp_indices, randomize_noise, to_uint8, timings, mask_surgery, paste_radius, two_streams, batched)
    if guard is None and ops.mx_guard_owned():        # a caller up the stack brackets this batch with its own guard (runner.run_clip_streamed, bench.py)
        return _swap_batch_once(*args)
    with ops.mx_guard_scope() as g:
        out = _swap_batch_once(*args)
        g.arm()
    if guard is not None:
        guard.append(g)
        return out
    if g.tripped():
        ops.mx_fallbacks += 1
        with ops.mx_exact():        # (the exact re-run leaves the caller's `timings` alone: its stage marks belong to the first pass)
            out = _swap_batch_once(net, parser, driven, target, comp_indices, randomize_noise, to_uint8, None, mask_surgery, paste_radius, two_streams, batched)
    return out


def _swap_batch_once(net, parser, driven, target, comp_indices, randomize_noise, to_uint8, timings, mask_surgery, paste_radius, two_streams, batched):
    def mark(name):
        if timings is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            timings.setdefault("_events", []).append((name, ev))
    mark("start")
    if batched is None:
        batched = SWAP_BATCHED and two_streams is None                 # an explicit two_streams= argument (True or False) selects the unbatched routes
    if two_streams is None:
        two_streams = TWO_STREAMS
    if batched:
        bs = driven.shape[0]
        enc = getattr(net, "encoder", None)
        if PARSE_BESIDE_ENCODE and hasattr(enc, "features"):
            # the encoder needs the region maps only for its last step (masked average pooling): the parser runs on a side stream next to
            # the encoder's body.  Driven and target faces go through both as ONE batch, but the 2 x bs full-size images are never concatenated,
            # shifted to [0, 1] or copied: the two down-sampling kernels in front of parser and encoder read them where they are.
            main, side = torch.cuda.current_stream(), _side_stream(driven.device, 0)

            def parse_on_side():
                with torch.cuda.stream(side):
                    out = parser.parse_batch((driven, target), seg12=True, pm1=True)   # uint8 [2 bs, 512, 512]
                driven.record_stream(side)
                target.record_stream(side)
                return out
            if ENCODE_ISSUED_FIRST:
                # the encoder body is the critical path of a batch (9 ms against the parser's 3): its launches are issued first, the parser's ~40 behind them —
                # the other way round the encoder's first kernel waits for the HOST to have issued the whole parser
                inputs_ready = torch.cuda.Event()
                inputs_ready.record(main)
            else:
                side.wait_stream(main)
                lab = parse_on_side()
            small = torch.empty((2 * bs, driven.shape[1], 256, 256), dtype=torch.float32, device=driven.device)
            ops.bilinear_resize(driven, (256, 256), align_corners=False, out=small[:bs])             # Net3._encode (networks.py:217)
            ops.bilinear_resize(target, (256, 256), align_corners=False, out=small[bs:])
            taps = enc.features(small)
            if ENCODE_ISSUED_FIRST:
                side.wait_event(inputs_ready)
                lab = parse_on_side()
            main.wait_stream(side)
            lab.record_stream(main)
            vec, _ = enc.codes(taps, lab)
        else:
            both = torch.cat([driven, target])
            lab = parser.parse_batch(both, seg12=True, pm1=True)          # uint8 [2 bs, 512, 512]
            vec, _ = net.get_style_vectors(both, lab)
        lab_d, lab_t, vec_d, vec_t = lab[:bs], lab[bs:], vec[:bs], vec[bs:]
        mark("parse+encode_x2")
    elif two_streams:
        # The driven and the target face are independent until the style-vector mix: run the parse -> encode chains on separate HIP
        # streams so that the launch-bound glue kernels and short-K convolutions of one fill the idle CUs of the others.
        main = torch.cuda.current_stream()
        bs = driven.shape[0]
        halves = SWAP_CHAINS // 2 if (SWAP_CHAINS >= 4 and bs % (SWAP_CHAINS // 2) == 0) else 1
        step = bs // halves
        jobs = [(img, i * step, (i + 1) * step) for img in (driven, target) for i in range(halves)]
        outs = []
        for j, (img, lo, hi) in enumerate(jobs):
            st = main if j == len(jobs) - 1 else _side_stream(driven.device, j)
            if st is not main:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                part = img[lo:hi]
                lab = parser.parse_batch((part + 1) / 2, seg12=True)
                vec, _ = net.get_style_vectors(part, lab)
            outs.append((lab, vec, st))
        for lab, vec, st in outs:
            if st is not main:
                main.wait_stream(st)
                lab.record_stream(main)
                vec.record_stream(main)
        lab_d = torch.cat([o[0] for o in outs[:halves]]) if halves > 1 else outs[0][0]
        vec_d = torch.cat([o[1] for o in outs[:halves]]) if halves > 1 else outs[0][1]
        lab_t = torch.cat([o[0] for o in outs[halves:]]) if halves > 1 else outs[1][0]
        vec_t = torch.cat([o[1] for o in outs[halves:]]) if halves > 1 else outs[1][1]
        mark("parse+encode_x2")
    else:
        lab_d = parser.parse_batch((driven + 1) / 2, seg12=True)          # uint8 [bs, 512, 512]
        lab_t = parser.parse_batch((target + 1) / 2, seg12=True)
        mark("parse_x2")
        vec_d, _ = net.get_style_vectors(driven, lab_d)
        vec_t, _ = net.get_style_vectors(target, lab_t)
        mark("encode_x2")
    codes = net.cal_style_codes(mix_style_vectors(vec_t, vec_d, comp_indices))
    mark("mix+mlps")
    extra = None
    lab = lab_t
    if mask_surgery:
        lab, hole, hole_map, lines = ops.swap_head_mask(lab_d, lab_t)
        content, border, full = ops.foreground_masks(lab, hole, paste_radius)
        extra = {"hole_mask": hole, "hole_map": hole_map, "lines": lines, "content": content, "border": border, "full": full}
        mark("mask_surgery")
    img, _, _ = net.gen_img(None, codes, lab, randomize_noise=randomize_noise)
    mark("gen_img")
    out = ops.tensor2im_u8(img) if to_uint8 else img
    mark("tensor2im")
    return (out, lab) if extra is None else (out, lab, extra)


@torch.no_grad()
def paste_back(swapped_u8: torch.Tensor, target_u8: torch.Tensor, content: torch.Tensor, border: torch.Tensor, soften: bool = True) -> torch.Tensor:
    """The reference's "past back" of a swapped face into its target crop (face_swap_video_pipeline.py:447, 464-473), on the device:

        swapped -> PIL resize to 512 x 512 and back to 1024 x 1024 (default BICUBIC; ``soften``, :447), bit-exact with Pillow
        content, border -> bilinear to the frame size (align_corners=False)
        pasted = swapped * content + T * (1 - content)
        out    = blending(T, pasted, mask=border)             (swap_face_fine/multi_band_blending.py:51-74, ten pyramid levels)

    ``swapped_u8`` / ``target_u8``: uint8 ``[bs, 1024, 1024, 3]`` (what ``swap_batch`` returns / the aligned target crop); ``content`` /
    ``border``: float ``[bs, 1, h, w]`` from ``swap_batch(..., mask_surgery=True)``.  Returns uint8 ``[bs, 1024, 1024, 3]``.
    """
    if swapped_u8.dtype != torch.uint8 or target_u8.dtype != torch.uint8 or swapped_u8.shape != target_u8.shape or swapped_u8.shape[-1] != 3:
        raise ValueError("paste_back: swapped and target frames are uint8 [bs, H, W, 3] of the same shape")
    bs, h, w, _ = swapped_u8.shape
    if soften:
        swapped_u8 = ops.pil_resize(ops.pil_resize(swapped_u8, (512, 512)), (w, h))
    t = target_u8.permute(0, 3, 1, 2).contiguous()
    sw = swapped_u8.permute(0, 3, 1, 2).float()
    cm = ops.bilinear_resize(content.float().contiguous(), (h, w), align_corners=False)
    bm = ops.bilinear_resize(border.float().contiguous(), (h, w), align_corners=False)
    pasted = torch.lerp(t.float(), sw, cm)                       # swapped * content + T * (1 - content)
    return ops.blending(t, pasted, bm).permute(0, 2, 3, 1).contiguous()

