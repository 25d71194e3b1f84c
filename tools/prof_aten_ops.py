"""Which stock-PyTorch (aten) device operations does a full-swap batch / a synthesis step still launch, and from where?  torch.profiler with Python stacks:
every aten op that launches device work, grouped by its innermost frame inside this repository."""
import argparse as _ap, collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import e4s2024_amd
from e4s2024_amd import ops, seeded, pipeline
e4s2024_amd.install()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
from torch.profiler import profile, ProfilerActivity

what = sys.argv[1] if len(sys.argv) > 1 else "swap"
dev = torch.device("cuda", 0)
opts = _ap.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3")
la = seeded.seeded_latent_avg(2, 18); net.latent_avg = la.to(dev); net = net.to(dev)
parser = FaceParser(None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
ops.STRICT_MASK = False
if what == "swap":
    drv, tgt = seeded.seeded_image(5, 8, 1024).to(dev), seeded.seeded_image(6, 8, 1024).to(dev)
    fn = lambda: pipeline.swap_batch(net, parser, drv, tgt)
else:
    codes = seeded.seeded_codes(1, 4, 12, 18, la).to(dev)
    mask = seeded.labels_to_onehot(seeded.blocky_labels(3, 4, 12, 512, 16), 12).to(dev)
    def fn():
        with torch.no_grad():
            return net.gen_img(None, codes, mask.view_as(mask), randomize_noise=False)[0]
for _ in range(3):
    fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fn()
    torch.cuda.synchronize()
groups = collections.Counter()
dev_us = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    frame = next((s for s in ev.stack if "/e4s2024_amd/" in s or "/tools/" in s), ev.stack[0] if ev.stack else "?")
    key = (ev.name, frame.replace(ROOT + "/", "")[:110])
    groups[key] += 1
    dev_us[key] += ev.device_time_total
print(f"{what}: aten ops that launch device work in ONE call ({sum(groups.values())} ops, {sum(dev_us.values()):.0f} us of device time)")
for key, n in sorted(groups.items(), key=lambda kv: -dev_us[kv[0]]):
    print(f"{n:4d} x {dev_us[key]:8.1f} us  {key[0]:28s} {key[1]}")
