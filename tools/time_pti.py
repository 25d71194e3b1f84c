"""BASELINE configs[3] (PTI): seconds per optimiser step at 1024x1024, batch 1, L2 loss, Adam."""
import argparse, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, pti

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=5); args = ap.parse_args()
e4s2024_amd.install()
from models.networks import Net3
dev = torch.device("cuda:0")
opts = types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=True, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts)
seeded.apply_seeded(net, 4, "net3")
net = net.to(dev).train()
net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev)
opt = torch.optim.Adam(pti.trainable_parameters(net), lr=1e-3)
vec = torch.from_numpy(seeded.seeded_array(41, "vec", (1, 12, 1280), dist="normal")).to(dev)
mask = seeded.labels_to_onehot(seeded.blocky_labels(3, 1, 12, 512, 16)).to(dev)
target = torch.tanh(torch.from_numpy(seeded.seeded_array(5, "img", (1, 3, 1024, 1024), dist="normal"))).to(dev)
print("trainable tensors:", len(pti.trainable_parameters(net)), "params:", sum(p.numel() for p in pti.trainable_parameters(net)))
losses = []
for i in range(args.steps + 1):
    if i == 1:
        torch.cuda.synchronize(); t0 = time.time()
    loss, _ = pti.pti_step(net, opt, vec, mask, target)
    losses.append(loss.item())
torch.cuda.synchronize()
eager_s = (time.time() - t0) / args.steps
from e4s2024_amd import ops
with ops.KernelTimer() as kt:
    pti.pti_step(net, opt, vec, mask, target)
torch.cuda.synchronize()
rows = sorted(kt.summary().items(), key=lambda kv: -kv[1][1])
print("per-call-site time of one step (HIP events; forward kernels and backward re-evaluations):")
for name, (calls, ms) in rows[:24]:
    print(f"   {ms:8.3f} ms  x{calls:3d}  {name}")
print(f"   total timed {sum(v[1] for v in kt.summary().values()):.2f} ms")
# the same step as one hipGraph
opt2 = torch.optim.Adam(pti.trainable_parameters(net), lr=1e-3, capturable=True, fused=os.environ.get("E4S_PTI_FUSED_ADAM", "1") != "0")
lab = ops.mask_to_labels(mask)
g = pti.GraphedPTIStep(net, opt2, vec, lab, target)
torch.cuda.synchronize(); tg = time.time()
for i in range(args.steps):
    gl, _ = g(vec, lab, target)
torch.cuda.synchronize()
print(f"PTI step as one hipGraph: {(time.time() - tg) / args.steps * 1e3:.2f} ms/iter, loss now {gl.item():.4f}")
print(f"PTI step, eager (fused HIP forward, native gradient kernels + split-bf16 MFMA GEMMs (csrc/gemm_sb.hip)): {eager_s:.3f} s/iter, loss {losses[0]:.4f} -> {losses[-1]:.4f}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
