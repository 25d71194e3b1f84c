"""A/B of the masked up layers under portrait-shaped maps (batch 4): four-parity kernel (UP_MX4) on / off, block path on / off.  usage: python tools/time_up_portrait.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import ops, seeded
e4s2024_amd.install()
from models.stylegan2 import model as sg2  # noqa: E402
DEV = "cuda:0"
bs = 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
lab = torch.from_numpy(seeded.facelike_labels(3, bs, 512)).to(DEV)
for cin, cout, w in ((256, 128, 128), (512, 256, 64)):
    m = sg2.StyledConv(cin, cout, 3, 512, upsample=True, mask_op=True).to(DEV).eval()
    x = torch.randn(bs, cin, w, w, device=DEV); st = torch.randn(bs, 12, 512, device=DEV); nz = torch.randn(bs, 1, 2 * w, 2 * w, device=DEV)
    for blocks_on, mx4_on in ((True, True), (True, False), (False, True), (False, False), (True, True), (True, False)):
        ops.UP_BLOCKS, ops.UP_MX4 = blocks_on, mx4_on
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
        print(f"{cin}->{cout} @{w} up, portrait maps, bs {bs}: UP_BLOCKS={int(blocks_on)} UP_MX4={int(mx4_on)}: layer {e0.elapsed_time(e1) / reps * 1e3:7.1f} us", flush=True)
ops.UP_BLOCKS, ops.UP_MX4 = True, True
