"""Which stock-PyTorch (aten) device operations does one PTI optimiser step launch, and from where?  torch.profiler with Python stacks, forward and backward:
every aten op that launches device work, grouped by op and by its innermost frame inside this repository (autograd's own nodes show up without one)."""
import collections, os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import e4s2024_amd
from e4s2024_amd import seeded, pti, ops
e4s2024_amd.install()
from models.networks import Net3
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
opts = types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=True, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts); seeded.apply_seeded(net, 4, "net3"); net = net.to(dev).train()
net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev)
opt = torch.optim.Adam(pti.trainable_parameters(net), lr=1e-3, capturable=True, fused=True)
vec = torch.from_numpy(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal")).to(dev)
lab = torch.from_numpy(seeded.blocky_labels(3, 1, 12, 512, 16)).to(dev).to(torch.uint8)
target = torch.tanh(torch.from_numpy(seeded.seeded_array(5, "img", (1, 3, 1024, 1024), dist="normal"))).to(dev)
for _ in range(3):
    pti.pti_step(net, opt, vec, lab, target)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    pti.pti_step(net, opt, vec, lab, target)
    torch.cuda.synchronize()
groups, dev_us = collections.Counter(), collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or (ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::")):
        continue
    frame = next((s for s in ev.stack if "/e4s2024_amd/" in s), None)
    if frame is None:
        par = ev.cpu_parent
        while par is not None and not ("Backward" in par.name or par.name.startswith("autograd::")):
            par = par.cpu_parent
        frame = f"<{par.name}>" if par is not None else (ev.stack[0] if ev.stack else "?")
    key = (ev.name, frame.replace(ROOT + "/", "")[:120])
    groups[key] += 1
    dev_us[key] += ev.device_time_total
print(f"one PTI step: {sum(groups.values())} aten ops launch device work, {sum(dev_us.values()):.0f} us of device time")
for key, n in sorted(groups.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d} x {dev_us[key]:8.1f} us  {key[0]:30s} {key[1]}")
