"""e4s2024_amd — MI355X-native engine for the E4S regional-GAN-inversion hot path.

Layout
    csrc/        hand-written HIP kernels for gfx950 + the C ABI (include/e4s_hip.h) -> lib/libe4s_hip.so
    _lib.py      ctypes binding (fails loudly if the .so is missing: there is no CPU fallback)
    ops.py       tensor-level wrappers (torch = device memory + stream plumbing only)
    dropin/      files with the reference's module paths (``models/networks.py``, ``models/stylegan2/model.py``,
                 ``models/stylegan2/op/``, ``models/encoders/psp_encoders.py``, ``swap_face_fine/face_parsing/*.py``)
                 whose forward passes call the kernels
    runner.py    one-process-per-GPU frame sharding over torch.distributed (RCCL)
    seeded.py    seed-only weights/inputs used by tests, fixtures and the bench

``install()`` redirects exactly those module names to the drop-in files, leaving every other module of the reference
tree (``utils.*``, ``models.encoders.model_irse``, ``swap_face_fine.gpen`` …) to resolve as before::

    import e4s2024_amd; e4s2024_amd.install()
    from models.networks import Net3                      # the MI355X implementation
"""
from __future__ import annotations

import importlib.abc
import importlib.util
import os
import sys

__version__ = "0.1.0"

DROPIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")

# module name -> path under dropin/ (packages end with /__init__.py)
OVERRIDES = {
    "models.networks": "models/networks.py",
    "models.stylegan2.model": "models/stylegan2/model.py",
    "models.stylegan2.op": "models/stylegan2/op/__init__.py",
    "models.stylegan2.op.fused_act": "models/stylegan2/op/fused_act.py",
    "models.stylegan2.op.upfirdn2d": "models/stylegan2/op/upfirdn2d.py",
    "models.stylegan2.op.conv2d_gradfix": "models/stylegan2/op/conv2d_gradfix.py",
    "models.encoders.psp_encoders": "models/encoders/psp_encoders.py",
    "swap_face_fine.face_parsing.model": "swap_face_fine/face_parsing/model.py",
    "swap_face_fine.face_parsing.resnet": "swap_face_fine/face_parsing/resnet.py",
    "swap_face_fine.face_parsing.face_parsing_demo": "swap_face_fine/face_parsing/face_parsing_demo.py",
}


class _DropinFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        rel = OVERRIDES.get(fullname)
        if rel is None:
            return None
        file = os.path.join(DROPIN_DIR, rel)
        is_pkg = rel.endswith("__init__.py")
        return importlib.util.spec_from_file_location(fullname, file, submodule_search_locations=[os.path.dirname(file)] if is_pkg else None)


_finder = None


def install(force: bool = False) -> str:
    """Redirect the hot-path module names (``OVERRIDES``) to the drop-in files.

    Parent packages (``models``, ``models.encoders``, ``swap_face_fine`` …) resolve to whatever is first on ``sys.path`` —
    the reference tree when the engine is used inside it, otherwise the empty packages under ``dropin/`` (appended at the
    END of ``sys.path``).  If an overridden module was already imported from elsewhere, raise unless ``force`` (then it is
    purged so the next import takes the drop-in)."""
    global _finder
    stale = [m for m in OVERRIDES if m in sys.modules
             and not (getattr(sys.modules[m], "__file__", None) or "").startswith(DROPIN_DIR)]
    if stale:
        if not force:
            raise RuntimeError(f"{stale} already imported from elsewhere; call e4s2024_amd.install() first, or install(force=True)")
        for m in list(sys.modules):
            if any(m == s or m.startswith(s + ".") for s in stale):
                del sys.modules[m]
    if _finder is None:
        _finder = _DropinFinder()
        sys.meta_path.insert(0, _finder)
    if DROPIN_DIR not in sys.path:
        sys.path.append(DROPIN_DIR)
    return DROPIN_DIR


def invalidate_weight_caches(module) -> int:
    """See ``ops.invalidate_weight_caches``: forget the prepared weight copies under ``module`` after raw ``p.data`` writes."""
    from . import ops
    return ops.invalidate_weight_caches(module)


def uninstall() -> None:
    global _finder
    if _finder is not None and _finder in sys.meta_path:
        sys.meta_path.remove(_finder)
    _finder = None
    if DROPIN_DIR in sys.path:
        sys.path.remove(DROPIN_DIR)
    for m in list(sys.modules):
        if any(m == s or m.startswith(s + ".") for s in OVERRIDES):
            del sys.modules[m]
