"""Eager vs hipGraph replay of the full swap and of gen_img at small batch."""
import os, sys, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, pipeline, ops, graphs
e4s2024_amd.install()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev); net = net.to(dev)
parser = FaceParser(None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
ops.STRICT_MASK = False
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for bs in ([int(a) for a in sys.argv[1:]] or [1, 2]):
    d = seeded.seeded_image(5, bs, 1024).to(dev); t = seeded.seeded_image(6, bs, 1024).to(dev)
    eager = timeit(lambda: pipeline.swap_batch(net, parser, d, t))
    g = graphs.graphed_swap(net, parser, d, t)
    graphed = timeit(lambda: g(d, t))
    print(f"full swap bs={bs}: eager p50 {eager:.2f} ms  hipGraph replay p50 {graphed:.2f} ms  ({eager / graphed:.2f}x)")
    codes = seeded.seeded_codes(1, bs, 12, 18, seeded.seeded_latent_avg(2, 18)).to(dev)
    lab = torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, 16)).to(dev)
    with torch.no_grad():
        e2 = timeit(lambda: net.gen_img(None, codes, lab, randomize_noise=False))
    g2 = graphs.graphed_gen_img(net, codes, lab)
    gr2 = timeit(lambda: g2(codes, lab))
    print(f"gen_img  bs={bs}: eager p50 {e2:.2f} ms  hipGraph replay p50 {gr2:.2f} ms  ({e2 / gr2:.2f}x)")
