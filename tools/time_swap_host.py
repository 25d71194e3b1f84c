"""Full swap (BASELINE configs[2], batch 8): how long does the HOST take to issue a batch, and what does the issue order of parser and encoder
cost its latency?  python tools/time_swap_host.py [bs] [iters]"""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, pipeline, ops
e4s2024_amd.install()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 13
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev); net = net.to(dev)
parser = FaceParser(None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
d = seeded.seeded_image(5, bs, 1024).to(dev); t = seeded.seeded_image(6, bs, 1024).to(dev)
ops.STRICT_MASK = False


def p50(first):
    pipeline.ENCODE_ISSUED_FIRST = first
    for _ in range(2):
        pipeline.swap_batch(net, parser, d, t)
    torch.cuda.synchronize()
    gpu, host = [], []
    for _ in range(iters):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        g = []
        t0 = time.perf_counter()
        a.record(); fr, _ = pipeline.swap_batch(net, parser, d, t, guard=g); b.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        gpu.append(a.elapsed_time(b)); host.append(1e3 * (t1 - t0))
    gpu.sort(); host.sort()
    return gpu[len(gpu) // 2], host[len(host) // 2], fr


ref = None
for rep in range(3):
    for first in (False, True):
        g, h, fr = p50(first)
        if ref is None:
            ref = fr.clone()
        print(f"encoder issued first = {int(first)}: p50 {g:7.3f} ms / batch of {bs} ({g / bs:.3f} ms per frame), host issue {h:6.3f} ms, frames equal {bool(torch.equal(fr, ref))}", flush=True)
