"""Stage hand-off of the reference pipeline, in memory, with the reference's files as an optional side output.

``FaceSwapVideoPipeline`` passes frames, region maps and style vectors between its stages through an experiment directory
(``face_swap_video_pipeline.py:221-231`` imgs/ + mask/ PNGs, ``:351-354`` styleVec/*.pt, ``:407-435`` read back per frame).
Here a clip is a ``ClipBatch`` of device tensors that goes from parsing to synthesis without touching the disk; ``dump`` writes the
same files with the same names and encodings (uint8 label PNGs, RGB PNGs, ``[1, 12, 1280]`` float tensors saved with
``torch.save``) for debugging or for handing a clip to the unmodified reference, and ``load`` reads a directory the reference wrote.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch


@dataclass
class ClipBatch:
    """``n`` frames of a clip.  Images are float ``[n, 3, 1024, 1024]`` in [-1, 1] (``ToTensor`` + ``Normalize(.5, .5)``,
    face_swap_video_pipeline.py:338-339); masks are uint8 12-class maps ``[n, 512, 512]``; style vectors ``[n, 12, 1280]``."""
    driven: Optional[torch.Tensor] = None
    target: Optional[torch.Tensor] = None
    driven_mask: Optional[torch.Tensor] = None
    target_mask: Optional[torch.Tensor] = None
    driven_style: Optional[torch.Tensor] = None
    target_style: Optional[torch.Tensor] = None

    def __len__(self):
        for t in (self.target, self.driven, self.target_mask, self.driven_mask, self.target_style, self.driven_style):
            if t is not None:
                return int(t.shape[0])
        return 0

    def to(self, device):
        return ClipBatch(*[None if t is None else t.to(device, non_blocking=True) for t in
                           (self.driven, self.target, self.driven_mask, self.target_mask, self.driven_style, self.target_style)])


def _to_u8_hwc(img: torch.Tensor) -> np.ndarray:
    """[-1, 1] float CHW -> uint8 HWC with the reference's ``tensor2im`` arithmetic (utils/torch_utils.py:64-76: truncating cast)."""
    x = ((img.detach().float().cpu().clamp(-1, 1) + 1) / 2 * 255).numpy()
    return np.transpose(x, (1, 2, 0)).astype(np.uint8)


def dump(clip: ClipBatch, exp_dir: str, first_index: int = 0) -> None:
    """Write ``imgs/{D,T}_%04d.png``, ``mask/{D,T}_mask_%04d.png`` and ``styleVec/{D,T}_style_vec_%04d.pt`` for the frames of ``clip``."""
    from PIL import Image
    for sub in ("imgs", "mask", "styleVec"):
        os.makedirs(os.path.join(exp_dir, sub), exist_ok=True)
    for k in range(len(clip)):
        i = first_index + k
        for tag, img, msk, vec in (("D", clip.driven, clip.driven_mask, clip.driven_style), ("T", clip.target, clip.target_mask, clip.target_style)):
            if img is not None:
                Image.fromarray(_to_u8_hwc(img[k])).save(os.path.join(exp_dir, "imgs", f"{tag}_{i:04d}.png"))
            if msk is not None:
                Image.fromarray(msk[k].detach().cpu().numpy().astype(np.uint8)).save(os.path.join(exp_dir, "mask", f"{tag}_mask_{i:04d}.png"))
            if vec is not None:
                torch.save(vec[k:k + 1].detach().cpu().float(), os.path.join(exp_dir, "styleVec", f"{tag}_style_vec_{i:04d}.pt"))


def load(exp_dir: str, first_index: int = 0, count: Optional[int] = None, device="cpu", size: int = 1024) -> ClipBatch:
    """Read back what ``dump`` (or the reference) wrote; missing kinds stay ``None``.  Images are resized to ``size`` like the
    reference does on load (``Image.open(...).convert("RGB").resize((1024, 1024))``, :408-411)."""
    from PIL import Image

    def frames(pattern):
        out, i = [], first_index
        while (count is None or len(out) < count) and os.path.exists(pattern % i):
            out.append(pattern % i)
            i += 1
        return out

    def images(tag):
        fs = frames(os.path.join(exp_dir, "imgs", tag + "_%04d.png"))
        if not fs:
            return None
        arr = np.stack([np.asarray(Image.open(f).convert("RGB").resize((size, size))) for f in fs])
        return (torch.from_numpy(arr).permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5            # ToTensor + Normalize(.5, .5)

    def masks(tag):
        fs = frames(os.path.join(exp_dir, "mask", tag + "_mask_%04d.png"))
        return torch.from_numpy(np.stack([np.asarray(Image.open(f)) for f in fs]).astype(np.uint8)) if fs else None

    def styles(tag):
        fs = frames(os.path.join(exp_dir, "styleVec", tag + "_style_vec_%04d.pt"))
        return torch.cat([torch.load(f, map_location="cpu").float() for f in fs]) if fs else None

    return ClipBatch(images("D"), images("T"), masks("D"), masks("T"), styles("D"), styles("T")).to(device)
