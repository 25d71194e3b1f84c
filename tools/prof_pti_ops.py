"""Which ATen ops of one eager PTI step launch the copy / elementwise kernels (torch.profiler, grouped by op and input shapes)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, pti
e4s2024_amd.install()
from models.networks import Net3
dev = torch.device("cuda:0")
opts = types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=True, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts); seeded.apply_seeded(net, 4, "net3"); net = net.to(dev).train()
net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev)
opt = torch.optim.Adam(pti.trainable_parameters(net), lr=1e-3, fused=True)
vec = torch.from_numpy(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal")).to(dev)
lab = torch.from_numpy(seeded.blocky_labels(3, 1, 12, 512, 16)).to(dev).to(torch.uint8)
target = torch.tanh(torch.from_numpy(seeded.seeded_array(5, "img", (1, 3, 1024, 1024), dist="normal"))).to(dev)
for _ in range(3):
    pti.pti_step(net, opt, vec, lab, target)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    pti.pti_step(net, opt, vec, lab, target)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::mul", "aten::add", "aten::sum", "aten::fill_", "aten::zero_", "aten::mm", "aten::bmm", "aten::addmm", "aten::baddbmm", "aten::matmul")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:28]:
    print(f"{e.key:18s} x{e.count:3d} device {e.self_device_time_total:8.1f} us  shapes {str(e.input_shapes)[:110]}")
