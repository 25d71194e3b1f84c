"""How long does the HOST take to enqueue one synthesis step (bench.py's workload)?  If that approaches the GPU's time per step, kernel work stops being the bound."""
import argparse as _ap, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import e4s2024_amd
from e4s2024_amd import ops, seeded
e4s2024_amd.install()
from models.networks import Net3
from e4s2024_amd.runner import StreamPipeline
dev = torch.device("cuda", 0)
opts = _ap.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net.G, 4, "net3", prefix="G.")
la = seeded.seeded_latent_avg(2, 18); net.latent_avg = la.to(dev); net = net.to(dev)
bs = 4
codes = seeded.seeded_codes(1, bs, 12, 18, la).to(dev)
mask = seeded.labels_to_onehot(seeded.blocky_labels(3, bs, 12, 512, 16), 12).to(dev)
ops.STRICT_MASK = False
scope = ops.mx_guard_scope(); scope.__enter__()
def step():
    with torch.no_grad():
        return net.gen_img(None, codes, mask.view_as(mask), randomize_noise=False)[0]
pipe = StreamPipeline(2, device=dev)
with pipe:
    for _ in range(6): pipe.submit(step)
torch.cuda.synchronize()
for n in (20, 60):
    t0 = time.perf_counter()
    with pipe:
        for _ in range(n): pipe.submit(step)
        t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, until the GPU is done {1e3 * (t2 - t0) / n:.3f} ms/step")
