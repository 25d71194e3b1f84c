"""The bench step (gen_img, batch 4, blocky masks) eager against one hipGraph replay per step.  python tools/time_graph_step.py [bs]"""
import os, sys, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, ops, graphs
e4s2024_amd.install()
from models.networks import Net3
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); la = seeded.seeded_latent_avg(2, 18); net.latent_avg = la.to(dev); net = net.to(dev)
codes = seeded.seeded_codes(1, bs, 12, 18, la).to(dev)
mask = seeded.labels_to_onehot(seeded.blocky_labels(3, bs, 12, 512, 16), 12).to(dev)
ops.STRICT_MASK = False
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    e = timed(lambda: net.gen_img(None, codes, mask.view_as(mask), randomize_noise=False)[0])
    g = graphs.graphed_gen_img(net, codes, mask)
    m2 = mask.clone()
    r = timed(lambda: g(codes, m2))           # m2 is a different tensor: copied into the static input every step
print(f"eager {e:.3f} ms/step = {bs / e * 1e3:.1f} faces/s; graph replay (mask copied in per step) {r:.3f} ms/step = {bs / r * 1e3:.1f} faces/s")
