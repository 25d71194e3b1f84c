"""Times the masked same-resolution layers of a synthesis step (batch 4) on the entry kernel (csrc/modconv_mxe.hip) against the round-3 kernel, per kind of region map.
usage: python tools/time_mxe.py [bs] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import e4s2024_amd
from e4s2024_amd import ops, seeded

e4s2024_amd.install()
from models.stylegan2 import model as sg2  # noqa: E402

DEV = "cuda:0"
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
only_layer = int(sys.argv[3]) if len(sys.argv) > 3 else -1
only_maps = sys.argv[4].split(",") if len(sys.argv) > 4 else None
LAYERS = [(512, 512, 32), (512, 512, 64), (256, 256, 128), (128, 128, 256)]
MAPS = {"blocky16": lambda: seeded.blocky_labels(3, bs, 12, 512, 16), "coarse4": lambda: seeded.blocky_labels(3, bs, 12, 512, 4),
        "portrait": lambda: seeded.facelike_labels(3, bs, 512), "iid": lambda: seeded.iid_labels(3, bs, 12, 512)}


def time_layer(m, x, st, lab, nz):
    with torch.no_grad():
        for _ in range(5):
            m(x, st, lab, noise=nz)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            m(x, st, lab, noise=nz)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


torch.manual_seed(0)
for li, (cin, cout, w) in enumerate(LAYERS):
    if only_layer >= 0 and li != only_layer:
        continue
    m = sg2.StyledConv(cin, cout, 3, 512, upsample=False, mask_op=True).to(DEV).eval()
    x = torch.randn(bs, cin, w, w, device=DEV)
    st = torch.randn(bs, 12, 512, device=DEV)
    nz = torch.randn(bs, 1, w, w, device=DEV)
    gf = 2.0 * cin * cout * 9 * w * w * bs / 1e9
    for name, mk in MAPS.items():
        if only_maps and name not in only_maps:
            continue
        lab = torch.from_numpy(mk()).to(DEV)
        ts = {}
        for on in (False, True):
            ops.MXE = on
            ts[on] = time_layer(m, x, st, lab, nz)
        print(f"{cin}->{cout} @{w} bs {bs} {name:9s}: round-3 kernel {ts[False]*1e3:7.1f} us ({gf/ts[False]:6.1f} TF)   entry kernel {ts[True]*1e3:7.1f} us ({gf/ts[True]:6.1f} TF)   ratio {ts[True]/ts[False]:.3f}", flush=True)
