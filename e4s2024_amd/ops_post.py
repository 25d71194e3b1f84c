"""Rows f2 / f3 of the scope table: what the video pipeline does to a frame around the swap — mask surgery, paste-back masks, uint8 <-> float frames,
Pillow's bicubic resize and the multi-band blend (``csrc/maskops.hip``).  Reference: ``swap_face_fine/swap_face_mask.py:194-367``,
``face_swap_video_pipeline.py:447-473``, ``swap_face_fine/multi_band_blending.py:5-74``.  Re-exported by ``ops``.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

from ._lib import lib
from .ops import _c, _p, _stream

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


__all__ = ['_labels_u8', 'swap_head_mask', 'foreground_masks', 'frames_to_tensor', 'PTI_BG_CLASSES', 'erode_labels', '_pil_tables', '_pil_bicubic_tables', 'pil_resize', 'pyr_down', 'pyr_up', 'laplacian_blend', 'blending']
