"""Breakdown of the full-swap unit (BASELINE config 3: bs=8) on the GPU.  python tools/time_swap.py [bs] [iters]"""
import os, sys, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, pipeline, ops
e4s2024_amd.install()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev); net = net.to(dev)
parser = FaceParser(None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
d = seeded.seeded_image(5, bs, 1024).to(dev); t = seeded.seeded_image(6, bs, 1024).to(dev)
ops.STRICT_MASK = False
for _ in range(2): pipeline.swap_batch(net, parser, d, t)
torch.cuda.synchronize()
acc = {}
for _ in range(iters):
    tm = {}
    pipeline.swap_batch(net, parser, d, t, timings=tm)
    torch.cuda.synchronize()
    ev = tm["_events"]
    for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
        acc.setdefault(n1, []).append(e0.elapsed_time(e1))
tot = 0
for k, v in acc.items():
    m = sorted(v)[len(v) // 2]; tot += m
    print(f"{k:12s} {m:8.3f} ms / batch of {bs}   ({m / bs:.3f} ms per face)")
print(f"total        {tot:8.3f} ms / batch -> {bs / tot * 1e3:.1f} swaps/s, {tot / bs:.3f} ms per face")
