"""A/B: the parser on a side stream beside the encoder body (pipeline.PARSE_BESIDE_ENCODE) against parser, then encoder, on one stream."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import default_opts, install_dropin
install_dropin()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
from e4s2024_amd import ops, seeded, pipeline
dev = torch.device("cuda", 0)
net = Net3(default_opts()); seeded.apply_seeded(net, 4, "net3"); net = net.to(dev).eval(); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev)
parser = FaceParser(seg_ckpt=None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
ops.STRICT_MASK = False
d = seeded.seeded_image(50, 8, 1024).to(dev); t = seeded.seeded_image(60, 8, 1024).to(dev)
def bench(n=20):
    with torch.no_grad():
        for _ in range(3): pipeline.swap_batch(net, parser, d, t, mask_surgery=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): pipeline.swap_batch(net, parser, d, t, mask_surgery=True)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for flag in (True, False):
        pipeline.PARSE_BESIDE_ENCODE = flag
        print(f"PARSE_BESIDE_ENCODE={flag}: {bench():.3f} ms per batch of 8")
