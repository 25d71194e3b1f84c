"""PTI fine-tuning step on the drop-in ``Net3`` (SURVEY §8 row f1 / BASELINE configs[3]).

Mirrors one inner iteration of ``VideoSwapPTICoach.train_e4s`` (training/video_swap_ft_coach.py:253-299) for the part that lives on the
hot path: ``cal_style_codes`` -> ``gen_img`` -> pixel loss -> ``backward`` -> optimiser step, with the style vectors and region map of a
frame as fixed inputs.  The forward runs on the fused HIP kernels; the backward currently goes through the stock-PyTorch forms of
``torch_ref.py``.  The perceptual / identity / parsing losses of ``calc_loss`` (:176-223) are separate networks outside the path
(LPIPS-alex, ArcFace, a UNet parser) and plug in through ``extra_loss``.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F


def trainable_parameters(net):
    """``configure_optimizer`` (video_swap_ft_coach.py:171-177): every parameter the constructor left ``requires_grad=True``."""
    return [p for p in net.parameters() if p.requires_grad]


def pti_step(net, optimizer: torch.optim.Optimizer, style_vectors: torch.Tensor, mask: torch.Tensor, target: torch.Tensor,
             foreground_mask: Optional[torch.Tensor] = None, l2_lambda: float = 1.0,
             extra_loss: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None):
    """One optimiser step.  ``style_vectors [bs, 12, 1280]``, ``mask`` one-hot ``[bs, 12, 512, 512]`` (or uint8 labels),
    ``target [bs, 3, 1024, 1024]`` in [-1, 1]; ``foreground_mask [bs, 1, 1024, 1024]`` restricts the loss as at :283-288.
    Returns ``(loss value, reconstruction)``."""
    codes = net.cal_style_codes(style_vectors)
    recon, _, _ = net.gen_img(None, codes, mask, randomize_noise=True)      # the coach calls gen_img with its default, fresh noise
    a, b = (recon, target) if foreground_mask is None else (recon * foreground_mask, target * foreground_mask)
    loss = l2_lambda * F.mse_loss(a, b)                                      # calc_loss :196-199 (loss_l2)
    if extra_loss is not None:
        loss = loss + extra_loss(recon, target)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach(), recon.detach()
