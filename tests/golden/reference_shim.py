"""Import the reference (``/root/reference``) on CPU.  USED ONLY IN THE BUILD CONTAINER
by ``make_golden.py`` to generate the fixtures in this directory — nothing under
``tests/test_*.py``, ``bench.py`` or ``__graft_entry__.py`` imports this module, and the
GPU box never sees ``/root/reference``.

What is wired, no reference source is copied:

1. ``torchvision`` is imported by ``models/networks.py:9`` and
   ``swap_face_fine/face_parsing/model.py:8`` but never used on the path; it is not
   installed here, so an empty module object is registered under that name.
2. ``models.stylegan2.op`` JIT-compiles CUDA at import (``op/fused_act.py:8-15``).  The
   reference ships its *own* CPU implementation of the same two ops in
   ``swap_face_fine/gpen/face_model/op/{fused_act,upfirdn2d}.py`` (``fused_act.py:92-96``,
   ``upfirdn2d.py:149-194``); those two files are loaded from where they lie and
   exposed as ``models.stylegan2.op`` together with the real ``conv2d_gradfix.py``.
3. ``swap_face_fine/face_parsing/model.py:15-16`` calls ``.cuda()`` at import and
   ``resnet.py:84`` downloads weights; both are neutralised for the duration of the
   import (``Tensor.cuda`` → identity, ``modelzoo.load_url`` → ``{}``).
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REF = os.environ.get("E4S_REFERENCE", "/root/reference")


def _load_file(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}; goldens can only be made in the build container")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "torchvision" not in sys.modules:
        try:
            importlib.import_module("torchvision")
        except Exception:
            tv = types.ModuleType("torchvision")
            tv.transforms = types.ModuleType("torchvision.transforms")
            sys.modules["torchvision"] = tv
            sys.modules["torchvision.transforms"] = tv.transforms
    if "cv2" not in sys.modules:
        try:
            importlib.import_module("cv2")
        except Exception:
            sys.modules["cv2"] = types.ModuleType("cv2")  # face_parsing_demo.py:7 imports it for visualisation only

    # (2) the reference's own CPU ops as models.stylegan2.op
    import models  # noqa: F401  (reference package)
    import models.stylegan2  # noqa: F401
    gp = os.path.join(REF, "swap_face_fine", "gpen", "face_model", "op")
    fa = _load_file("_ref_gpen_fused_act", os.path.join(gp, "fused_act.py"))
    up = _load_file("_ref_gpen_upfirdn2d", os.path.join(gp, "upfirdn2d.py"))
    op = types.ModuleType("models.stylegan2.op")
    op.__path__ = [os.path.join(REF, "models", "stylegan2", "op")]
    op.FusedLeakyReLU = fa.FusedLeakyReLU
    op.fused_leaky_relu = fa.fused_leaky_relu
    op.upfirdn2d = up.upfirdn2d
    op.upfirdn2d_native = up.upfirdn2d_native
    sys.modules["models.stylegan2.op"] = op
    op.conv2d_gradfix = _load_file(
        "models.stylegan2.op.conv2d_gradfix",
        os.path.join(REF, "models", "stylegan2", "op", "conv2d_gradfix.py"))
    return op


def import_net3():
    install()
    from models.networks import Net3  # reference
    from models.stylegan2 import model as sg2  # reference
    from models.encoders import psp_encoders, helpers
    return Net3, sg2, psp_encoders, helpers


def import_bisenet():
    install()
    import torch
    import torch.utils.model_zoo as modelzoo
    orig_cuda = torch.Tensor.cuda
    orig_load = modelzoo.load_url
    torch.Tensor.cuda = lambda self, *a, **k: self
    modelzoo.load_url = lambda *a, **k: {}
    try:
        from swap_face_fine.face_parsing import model as bis
        from swap_face_fine.face_parsing import resnet as rn
    finally:
        torch.Tensor.cuda = orig_cuda
    # keep load_url neutralised: Resnet18.__init__ calls it on every construction
    return bis, rn, (modelzoo, orig_load)


def import_face_parsing_demo():
    bis, rn, _ = import_bisenet()
    install()
    import torch
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        # datasets/dataset.py imports torchvision.transforms etc. at module scope; only the
        # 19->12 remap function is needed, so give the names it touches at import time.
        tv = sys.modules["torchvision"]
        if not hasattr(tv.transforms, "Compose"):
            class _T:  # placeholders, never called on this path
                def __init__(self, *a, **k):
                    pass
            tv.transforms.__getattr__ = lambda name: _T
            tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
            tv.transforms.functional.__getattr__ = lambda name: _T
            sys.modules["torchvision.transforms.functional"] = tv.transforms.functional
            tv.__getattr__ = lambda name: _T
        from swap_face_fine.face_parsing import face_parsing_demo as fpd
    finally:
        torch.Tensor.cuda = orig_cuda
    return fpd
