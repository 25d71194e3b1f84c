"""Batched full-swap unit on the device: per face
    2 x parse (BiSeNet, 19 -> 12 classes) + 2 x get_style_vectors + style-vector mix + cal_style_codes + gen_img (+ tensor2im)
which is what ``face_swap_video_pipeline.py`` does per frame between its CPU stages (:212-219 parsing, :332-354 style vectors,
:429-443 mix + synthesis), here for a whole batch without leaving the GPU.  With ``mask_surgery=True`` the synthesis is driven by
``swap_head_mask_hole_first(driven_map, target_map)`` as at face_swap_video_pipeline.py:420 (row f2, ``ops.swap_head_mask``) and the
paste-back masks of :456-463 (row f3, ``ops.foreground_masks``) are returned too; the default keeps BASELINE configs[2]'s unit of
work, where the target's own region map drives the synthesis.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops

DEFAULT_COMP_INDICES = tuple(sorted(set(range(12)) - {0, 4, 11}))      # face_swap_video_pipeline.py:436: keep target background, hair, ear-rings


_sel_cache = {}


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
               paste_radius: int = 5):
    """``driven`` / ``target``: ``[bs, 3, 1024, 1024]`` in [-1, 1] on the device.  Returns uint8 ``[bs, 1024, 1024, 3]`` frames
    (or the float image) and the 12-class region maps the synthesis used; with ``mask_surgery`` a third value
    ``{"hole_mask", "hole_map", "lines", "content", "border", "full"}`` (the reference's paste-back inputs, :456-463)."""
    def mark(name):
        if timings is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            timings.setdefault("_events", []).append((name, ev))
    mark("start")
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
