"""Consecutive full-swap batches (pipeline.swap_batch, BASELINE configs[2] / [4]) on alternating HIP streams against one stream.
python tools/time_swap_pipeline.py [bs] [batches]"""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, pipeline, ops
from e4s2024_amd.runner import StreamPipeline
e4s2024_amd.install()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev); net = net.to(dev)
parser = FaceParser(None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
d = seeded.seeded_image(5, bs, 1024).to(dev); t = seeded.seeded_image(6, bs, 1024).to(dev)
ops.STRICT_MASK = False
with torch.no_grad():
    ref = pipeline.swap_batch(net, parser, d, t, mask_surgery=True)[0]
    for ns in (1, 2, 3, 1, 2):
        with StreamPipeline(ns, device=dev) as sp:
            outs = [sp.submit(pipeline.swap_batch, net, parser, d, t, mask_surgery=True)[0] for _ in range(2 * ns)]
        torch.cuda.synchronize()
        same = all(torch.equal(o, ref) for o in outs)
        del outs
        t0 = time.perf_counter()
        with StreamPipeline(ns, device=dev) as sp:
            for _ in range(nb):
                sp.submit(pipeline.swap_batch, net, parser, d, t, mask_surgery=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nb
        print(f"bs {bs}: {ns} stream(s): {dt * 1e3:.3f} ms / batch = {bs / dt:.1f} swaps/s   frames equal to one stream: {same}", flush=True)
