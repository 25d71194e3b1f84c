"""Stand-alone timing of the two per-image stages in front of the synthesis: FaceParser.parse_batch and Net3.get_style_vectors.
python tools/time_parts.py [parse|encode|both] [bs] [iters]   (one stream; run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, ops
e4s2024_amd.install()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
what = sys.argv[1] if len(sys.argv) > 1 else "both"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev); net = net.to(dev)
parser = FaceParser(None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
img = seeded.seeded_image(5, bs, 1024).to(dev)
ops.STRICT_MASK = False
def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): r = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters, r
with torch.no_grad():
    lab = parser.parse_batch(img, seg12=True, pm1=True)
    if what in ("parse", "both"):
        ms, lab = timed(lambda: parser.parse_batch(img, seg12=True, pm1=True))
        print(f"parse_batch        bs={bs}: {ms:7.3f} ms  ({ms / bs:.3f} ms per image)")
    if what in ("encode", "both"):
        ms, _ = timed(lambda: net.get_style_vectors(img, lab))
        print(f"get_style_vectors  bs={bs}: {ms:7.3f} ms  ({ms / bs:.3f} ms per image)")
