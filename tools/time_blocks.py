"""Masked up layers under parser-made region maps (seeded random images): per-kernel time of gen_img with / without the uniform-block path,
and how many 16 x 16 blocks qualify.  python tools/time_blocks.py [bs]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, ops
e4s2024_amd.install()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev); net = net.to(dev)
parser = FaceParser(None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
img = seeded.seeded_image(6, bs, 1024).to(dev)
ops.STRICT_MASK = False
with torch.no_grad():
    lab = parser.parse_batch((img + 1) / 2, seg12=True)
    if len(sys.argv) > 2 and sys.argv[2] == "half":        # left half: 4 x 4 coarse cells, right half: i.i.d.
        import numpy as np
        l1 = seeded.blocky_labels(3, bs, 12, 512, 4); l2 = seeded.iid_labels(9, bs, 12, 512)
        l1[:, :, 256:] = l2[:, :, 256:]
        lab = torch.from_numpy(l1).to(dev).to(torch.uint8)
    if len(sys.argv) > 2 and sys.argv[2] == "tophalf":     # top half coarse, bottom half i.i.d.
        l1 = seeded.blocky_labels(3, bs, 12, 512, 4); l2 = seeded.iid_labels(9, bs, 12, 512)
        l1[:, 256:, :] = l2[:, 256:, :]
        lab = torch.from_numpy(l1).to(dev).to(torch.uint8)
    codes = seeded.seeded_codes(1, bs, 12, 18, net.latent_avg.cpu()).to(dev)
    for res in (64, 128, 256):
        ub, _ = ops.uniform_blocks(lab, res, res, 12)
        print(f"output {res}: {float((ub != 255).float().mean()):.3f} of the 16 x 16 blocks go to the block kernel; labels present: {sorted(set(lab.unique().tolist()))}")
    for flag in (True, False, True, False):
        ops.UP_BLOCKS = flag
        for _ in range(2):
            net.gen_img(None, codes, lab, randomize_noise=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        with ops.KernelTimer() as kt:
            for _ in range(5):
                net.gen_img(None, codes, lab, randomize_noise=False)
        b.record(); torch.cuda.synchronize()
        ks = kt.summary()
        keys = [k for k in ks if "4,1,1,8,5" in k or "blocks" in k]
        print(f"UP_BLOCKS={flag}: {a.elapsed_time(b) / 5:.3f} ms per gen_img;", {k: round(ks[k][1] / 5, 3) for k in keys},
              {d: round(v[1] / 5, 3) for d, v in kt.by_detail("masked_upconv_blocks").items()},
              {d: round(v[1] / 5, 3) for d, v in kt.by_detail(keys[0]).items() if "up" in d})
