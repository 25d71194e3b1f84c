"""Per-shape launch times of the face parser's convolutions (16 images at 1024 x 1024 -> 512 x 512 -> 19-class maps), nothing beside it."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import install_dropin
install_dropin()
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
from e4s2024_amd import ops, seeded
dev = "cuda:0"
parser = FaceParser(seg_ckpt=None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
img = seeded.seeded_image(5, 16, 1024).to(dev)
with torch.no_grad():
    for _ in range(2):
        parser.parse_batch(img, seg12=True, pm1=True)
    torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        parser.parse_batch(img, seg12=True, pm1=True)
    tot = 0.0
    for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1][1]):
        print(f"{k:40s} calls {v[0]:3d}  {v[1]:7.3f} ms"); tot += v[1]
    print(f"timed launches {tot:.3f} ms")
    for name in [k for k in kt.summary() if k.startswith("conv2d")]:
        for d, (c, t) in sorted(kt.by_detail(name).items(), key=lambda kv: -kv[1][1]):
            print(f"{name:24s} {d:18s} calls {c:3d}  {1e3 * t / c:7.1f} us each")
