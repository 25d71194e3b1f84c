"""Times the masked up layers (batch 4) on maps made of uniform cells: region-uniform blocks on csrc/modconv_upblock_mx.hip vs the whole layer in the composed form.
usage: python tools/time_upblock.py [bs] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import ops, seeded
e4s2024_amd.install()
from models.stylegan2 import model as sg2  # noqa: E402
DEV = "cuda:0"
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(0)
for cin, cout, w, cells in ((256, 128, 128, 16), (512, 256, 64, 8), (512, 512, 32, 4)):
    m = sg2.StyledConv(cin, cout, 3, 512, upsample=True, mask_op=True).to(DEV).eval()
    x = torch.randn(bs, cin, w, w, device=DEV); st = torch.randn(bs, 12, 512, device=DEV); nz = torch.randn(bs, 1, 2 * w, 2 * w, device=DEV)
    lab = torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, cells)).to(DEV)
    keep = ops.UP_BLOCKS_MIN_WIDTH
    ops.UP_BLOCKS_MIN_WIDTH = 32
    for on in (False, True, False, True):
        ops.UP_BLOCKS = on
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
        print(f"{cin}->{cout} @{w} up, {cells} x {cells} cells, bs {bs}: {'uniform blocks on the block kernel' if on else 'composed form only'}: layer {e0.elapsed_time(e1) / reps * 1e3:7.1f} us", flush=True)
    ops.UP_BLOCKS_MIN_WIDTH = keep
