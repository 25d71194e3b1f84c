"""Do torch's Adam variants agree on this build?  (plain foreach vs capturable vs fused, eager and inside a captured graph)"""
import torch
dev = "cuda:0"
torch.manual_seed(0)
p0 = torch.randn(1000, device=dev)
grads = [torch.randn(1000, device=dev) * (10.0 ** -i) for i in range(4)]


def run(graph=False, **kw):
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-3, **kw)
    if not graph:
        for g in grads:
            p.grad = g.clone()
            opt.step()
        return p.detach().clone()
    static_g = grads[0].clone()
    p.grad = static_g
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        opt.step()                       # warm-up = real step 1
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        opt.step()
    for g in grads[1:]:
        static_g.copy_(g)
        gr.replay()
    torch.cuda.synchronize()
    return p.detach().clone()


ref = run()
for name, kw, graph in [("capturable", dict(capturable=True), False), ("fused", dict(fused=True), False), ("fused+capturable", dict(fused=True, capturable=True), False),
                        ("capturable, graphed", dict(capturable=True), True), ("fused+capturable, graphed", dict(fused=True, capturable=True), True)]:
    out = run(graph, **kw)
    print(f"{name:28s} max |dp| vs plain Adam: {(out - ref).abs().max().item():.3e}   (total movement {(ref - p0).abs().max().item():.3e})")

# the generator's own parameter list (263 tensors, 162 M elements, 1-element noise weights up to 8.5 M-element MLP matrices)
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import e4s2024_amd
e4s2024_amd.install()
from models.networks import Net3
net = Net3(types.SimpleNamespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=True, start_from_latent_avg=True,
                                 learn_in_w=False))
shapes = [tuple(p.shape) for p in net.parameters() if p.requires_grad]
del net
torch.manual_seed(1)
base = [torch.randn(s, device=dev) for s in shapes]
gs = [[torch.randn(s, device=dev) * 10.0 ** -k for s in shapes] for k in range(3)]


def run_many(**kw):
    ps = [b.clone().requires_grad_(True) for b in base]
    opt = torch.optim.Adam(ps, lr=1e-3, **kw)
    for step in gs:
        for p, g in zip(ps, step):
            p.grad = g.clone()
        opt.step()
    return [p.detach() for p in ps]


ref = run_many()
for name, kw in [("capturable", dict(capturable=True)), ("fused+capturable", dict(fused=True, capturable=True))]:
    out = run_many(**kw)
    worst = max(((a - b).abs().max().item(), i) for i, (a, b) in enumerate(zip(out, ref)))
    print(f"{len(shapes)} tensors, {name:18s}: worst max |dp| vs plain Adam {worst[0]:.3e} (tensor {worst[1]}, shape {shapes[worst[1]]})")
