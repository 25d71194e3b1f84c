"""Where do the small torch kernels and memcpys of one swap batch (pipeline.swap_batch, batch 8, mask surgery on) come from?  One warm batch under
torch.profiler with Python stacks; every device activity that is not one of the library's own kernels is listed with the innermost frame of this
package that issued it.   python tools/find_copies.py [gen]   ("gen": the synthesis step of bench.py alone)"""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import default_opts, install_dropin
install_dropin()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
from e4s2024_amd import ops, seeded, pipeline
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
net = Net3(default_opts()); seeded.apply_seeded(net, 4, "net3"); net = net.to(dev).eval()
la = seeded.seeded_latent_avg(2, 18); net.latent_avg = la.to(dev)
parser = FaceParser(seg_ckpt=None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
ops.STRICT_MASK = False
what = sys.argv[1] if len(sys.argv) > 1 else "swap"
if what == "gen":
    bs = 4
    codes = seeded.seeded_codes(1, bs, 12, 18, la).to(dev)
    mask = seeded.labels_to_onehot(seeded.blocky_labels(3, bs, 12, 512, 16), 12).to(dev)
    def run():
        with torch.no_grad():
            return net.gen_img(None, codes, mask, randomize_noise=False)[0]
else:
    d = seeded.seeded_image(50, 8, 1024).to(dev); t = seeded.seeded_image(60, 8, 1024).to(dev)
    def run():
        with torch.no_grad():
            return pipeline.swap_batch(net, parser, d, t, mask_surgery=True)[0]
for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    run()
    torch.cuda.synchronize()
ours = os.path.join(ROOT, "e4s2024_amd")
by = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue                                            # top-level aten ops only
    kt = sum(k.duration for k in ev.kernels)
    if not ev.kernels:
        continue
    frame = next((s for s in ev.stack if "e4s2024_amd" in s or "tools/" in s), ev.stack[0] if ev.stack else "?")
    key = (ev.name, frame.replace(ROOT + "/", "")[:110], ",".join(sorted({k.name[:40] for k in ev.kernels})))
    by[key] += 1; dur[key] += kt
print(f"{'calls':>5} {'us':>8}  op / kernel / frame")
for key, n in sorted(by.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{n:5d} {dur[key]:8.1f}  {key[0]:24s} {key[2]:44s} {key[1]}")
print("total", sum(by.values()), "ops", f"{sum(dur.values()):.0f} us")
