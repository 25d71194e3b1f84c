"""Deterministic weights / inputs for parity work.

No checkpoint of the reference exists in this environment (reference
``.gitignore:2-4`` excludes ``pretrained/``), so every parity test, the bench and
the golden fixtures are driven by *seeds*: the same ``(seed, key, shape)`` gives
the same fp32 tensor in the container that generated the fixtures and on the GPU
box.  ``numpy.random.RandomState`` (the legacy MT19937 stream) is frozen across
numpy versions, which is why it is used instead of ``torch.manual_seed``.

The distributions follow the reference constructors where they matter
(``models/stylegan2/model.py:103-105, 141, 227-231, 342`` use ``randn``;
modulation bias is initialised to 1, ``:231``) and are otherwise chosen to look
like a trained network (non-zero biases, positive BatchNorm variances) so that no
code path is silently multiplied by zero.
"""
from __future__ import annotations

import re
import zlib
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch

__all__ = [
    "seeded_array",
    "seeded_state_dict",
    "apply_seeded",
    "blocky_labels",
    "iid_labels",
    "labels_to_onehot",
    "seeded_codes",
    "seeded_latent_avg",
    "seeded_image",
]


def _rs(seed: int, key: str) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode("utf-8")) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


_SQRT3 = float(np.sqrt(3.0))


def seeded_array(seed: int, key: str, shape, mean: float = 0.0, std: float = 1.0, dist: str = "uniform") -> np.ndarray:
    """fp32 array with the given mean/std that depends only on (seed, key, shape).

    ``dist="uniform"`` (default, used for weights: 164 M of them must be regenerated on
    the GPU box in well under a second per 10 M) draws uint32 words from the legacy
    stream and maps them to U(-sqrt3, sqrt3)·std + mean; ``dist="normal"`` uses the
    legacy ``standard_normal`` (used for inputs: codes, noise maps, images)."""
    n = int(np.prod(shape)) if len(shape) else 1
    rs = _rs(seed, key)
    if dist == "normal":
        a = rs.standard_normal(n).astype(np.float32)
    else:
        u = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
        a = u.astype(np.float32)
        a *= np.float32(2.0 * _SQRT3 / 4294967296.0)
        a -= np.float32(_SQRT3)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return a.reshape(tuple(shape))


# (regex, rule) — first match wins.  rule = (mean, std) | callable(seed, key, shape)
def _fan_in_std(gain: float):
    def rule(seed, key, shape):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        return seeded_array(seed, key, shape, 0.0, gain / np.sqrt(max(fan_in, 1)))
    return rule


def _positive(mean: float, std: float):
    def rule(seed, key, shape):
        return np.abs(seeded_array(seed, key, shape, 0.0, std)) + np.float32(mean)
    return rule


def _const_kernel(seed, key, shape):
    # Blur / Upsample FIR buffers are constants of the architecture
    # (models/stylegan2/model.py:23-31, 39, 82-87): outer([1,3,3,1]) / 64 * 4.
    k = np.array([1.0, 3.0, 3.0, 1.0], dtype=np.float32)
    k2 = np.outer(k, k)
    k2 = k2 / k2.sum() * 4.0
    assert tuple(shape) == (4, 4), shape
    return k2.astype(np.float32)


_RULES_NET3 = [
    (r"\.blur\.kernel$|\.upsample\.kernel$", _const_kernel),
    (r"^G\.noises\.", (0.0, 1.0, "normal")),
    (r"^G\.input\.input$", (0.0, 1.0)),
    (r"\.modulation\.weight$", (0.0, 1.0)),
    (r"\.modulation\.bias$", (1.0, 0.1)),
    (r"\.noise\.weight$", (0.0, 0.1)),
    (r"\.activate\.bias$", (0.0, 0.1)),
    (r"^G\.to_rgb.*\.bias$", (0.0, 0.1)),
    (r"^G\.style\.\d+\.weight$", (0.0, 100.0)),   # randn / lr_mul, lr_mul = 0.01 (model.py:141)
    (r"^G\.style\.\d+\.bias$", (0.0, 0.1)),
    (r"^G\..*conv\.weight$", (0.0, 1.0)),
    (r"^MLPs\.\d+\.mlp\.\d+\.weight$", (0.0, 1.0)),
    (r"^MLPs\.\d+\.mlp\.\d+\.bias$", (0.0, 0.1)),
    # encoder: PReLU slopes (res_layer.2 / input_layer.2) are 1-D, convs are 4-D
    (r"^encoder\..*\.weight$", None),  # resolved by ndim below
]

_RULES_BISENET = [
    (r"num_batches_tracked$", "zero_long"),
    (r"running_var$", _positive(0.5, 0.5)),
    (r"running_mean$", (0.0, 0.2)),
    (r"\.bn\d*\.weight$|\.bn\.weight$|bn_atten\.weight$|downsample\.1\.weight$", _positive(0.8, 0.3)),
    (r"\.bn\d*\.bias$|\.bn\.bias$|bn_atten\.bias$|downsample\.1\.bias$", (0.0, 0.2)),
    (r"\.weight$", _fan_in_std(1.4)),
]


def _resolve(family: str, seed: int, key: str, shape, dtype) -> torch.Tensor:
    rules = _RULES_NET3 if family == "net3" else _RULES_BISENET
    for pat, rule in rules:
        if re.search(pat, key):
            break
    else:
        raise KeyError(f"no seeded rule for {family} key {key!r} shape {tuple(shape)}")
    if rule == "zero_long":
        return torch.zeros(tuple(shape), dtype=torch.long)
    if rule is None:  # encoder weights
        if len(shape) == 1:
            arr = seeded_array(seed, key, shape, 0.25, 0.05)       # PReLU slopes
        else:
            arr = _fan_in_std(1.4)(seed, key, shape)                # conv kernels
    elif callable(rule):
        arr = rule(seed, key, shape)
    else:
        arr = seeded_array(seed, key, shape, *rule)
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(dtype)


def seeded_state_dict(template: Mapping[str, torch.Tensor], seed: int, family: str) -> Dict[str, torch.Tensor]:
    """Build a state_dict with the template's keys/shapes/dtypes and seeded values.

    ``family`` is ``"net3"`` (keys of ``models.networks.Net3`` or any sub-tree of it
    when the caller prefixes keys accordingly) or ``"bisenet"``.
    """
    out = {}
    for key, t in template.items():
        out[key] = _resolve(family, seed, key, tuple(t.shape), t.dtype)
    return out


def apply_seeded(module: torch.nn.Module, seed: int, family: str, prefix: str = "") -> torch.nn.Module:
    """Load seeded values into ``module`` in place.  ``prefix`` is prepended to the
    module's own keys before the rule lookup (e.g. ``"G."`` for a bare Generator)."""
    sd = module.state_dict()
    tmpl = {prefix + k: v for k, v in sd.items()}
    vals = seeded_state_dict(tmpl, seed, family)
    module.load_state_dict({k[len(prefix):]: v for k, v in vals.items()})
    return module


# --------------------------------------------------------------------------- inputs
def blocky_labels(seed: int, bs: int, n_cls: int = 12, size: int = 512, cells: int = 16) -> np.ndarray:
    """uint8 [bs, size, size] label maps made of ``cells``×``cells`` constant blocks
    (the benign case — most GEMM tiles see one region)."""
    small = np.random.RandomState(seed).randint(0, n_cls, (bs, cells, cells)).astype(np.uint8)
    rep = size // cells
    return np.repeat(np.repeat(small, rep, axis=1), rep, axis=2)


def facelike_labels(seed: int, bs: int, size: int = 512) -> np.ndarray:
    """uint8 [bs, size, size] region maps shaped like a parsed portrait (12 classes as in CelebAMask-HQ's merged set: 0 background, 1 lips,
    2 eyebrows, 3 eyes, 4 hair, 5 nose, 6 skin, 7 ears, 8 neck, 9 mouth, 10 eye glasses unused, 11 ear rings): ellipses with a few pixels of
    per-sample jitter — large coherent regions with curved borders, the kind of map the face parser produces on real photographs."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size

    def ell(cx, cy, rx, ry):
        return ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0

    out = np.zeros((bs, size, size), dtype=np.uint8)
    for b in range(bs):
        j = lambda s=0.015: float(rs.uniform(-s, s))          # noqa: E731
        cx, cy = 0.5 + j(), 0.5 + j()
        m = out[b]
        m[ell(cx, cy + 0.42, 0.34, 0.22)] = 8                                   # neck / shoulders
        m[ell(cx, cy - 0.06, 0.36 + j(), 0.44 + j())] = 4                       # hair
        m[ell(cx - 0.30, cy + 0.02, 0.035, 0.07)] = 7; m[ell(cx + 0.30, cy + 0.02, 0.035, 0.07)] = 7   # ears
        m[ell(cx - 0.30, cy + 0.10, 0.012, 0.02)] = 11                          # ear ring
        m[ell(cx, cy + 0.04, 0.27 + j(), 0.36 + j())] = 6                       # skin
        m[ell(cx - 0.11, cy - 0.07, 0.06, 0.012)] = 2; m[ell(cx + 0.11, cy - 0.07, 0.06, 0.012)] = 2   # eyebrows
        m[ell(cx - 0.11, cy - 0.02, 0.045, 0.02)] = 3; m[ell(cx + 0.11, cy - 0.02, 0.045, 0.02)] = 3   # eyes
        m[ell(cx, cy + 0.08, 0.04, 0.07)] = 5                                   # nose
        m[ell(cx, cy + 0.21, 0.085, 0.035)] = 1                                 # lips
        m[ell(cx, cy + 0.21, 0.055, 0.012)] = 9                                 # mouth
    return out


def iid_labels(seed: int, bs: int, n_cls: int = 12, size: int = 512) -> np.ndarray:
    """uint8 [bs, size, size] i.i.d. per-pixel labels — the adversarial case where
    every tile sees all regions."""
    return np.random.RandomState(seed).randint(0, n_cls, (bs, size, size)).astype(np.uint8)


def labels_to_onehot(labels: np.ndarray, n_cls: int = 12) -> torch.Tensor:
    """Same result as the reference's ``labelMap2OneHot`` (utils/torch_utils.py:207-213)
    on a [bs,H,W] uint8 label array: float32 [bs, n_cls, H, W]."""
    lab = torch.from_numpy(labels.astype(np.int64))
    oh = torch.zeros(lab.shape[0], n_cls, lab.shape[1], lab.shape[2], dtype=torch.float32)
    return oh.scatter_(1, lab[:, None], 1.0)


def seeded_latent_avg(seed: int = 2, n_styles: int = 18) -> torch.Tensor:
    return torch.from_numpy(seeded_array(seed, "latent_avg", (n_styles, 512), 0.0, 0.1, "normal"))


def seeded_codes(seed: int, bs: int, n_cls: int = 12, n_styles: int = 18, latent_avg: torch.Tensor | None = None) -> torch.Tensor:
    """W+ codes [bs, n_cls, n_styles, 512] = latent_avg + 0.5·N(0,1)."""
    if latent_avg is None:
        latent_avg = seeded_latent_avg(2, n_styles)
    c = torch.from_numpy(seeded_array(seed, "codes", (bs, n_cls, n_styles, 512), 0.0, 0.5, "normal"))
    return c + latent_avg[None, None]


def seeded_image(seed: int, bs: int, size: int = 1024) -> torch.Tensor:
    """[-1,1] images [bs,3,size,size]: tanh of smooth-ish noise."""
    a = seeded_array(seed, "image", (bs, 3, size, size), 0.0, 1.0, "normal")
    return torch.tanh(torch.from_numpy(a))
