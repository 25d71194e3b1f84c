"""conv2d_winograd_pre on a side stream while OTHER work runs on the main stream: which concurrent neighbour breaks it?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from e4s2024_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
bs, cin, cout, h = 8, 512, 512, 32
x = torch.randn(bs, cin, h, h, device=dev); x2 = torch.randn(bs, cin, h, h, device=dev)
w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
st = ops.plane_stats(x, 1e-5); st2 = ops.plane_stats(x2, 1e-5)
slope = torch.rand(cout, device=dev)
ps, pc = ops.PreparedWinogradSplit().get(w), ops.PreparedConv().get(w)
side = torch.cuda.Stream()
big = torch.randn(64, 1024, 1024, device=dev)
neighbours = {
    "nothing": lambda: None,
    "direct conv": lambda: [ops.conv2d(x2, pc, 1, 1, in_norm=st2, prelu=slope) for _ in range(6)],
    "winograd pre": lambda: [ops.conv2d_winograd_pre(x2, ps, in_norm=st2, prelu=slope) for _ in range(6)],
    "plane_stats": lambda: [ops.plane_stats(x2, 1e-5) for _ in range(40)],
    "torch elementwise": lambda: [big.mul_(1.0001) for _ in range(10)],
    "torch.empty + fill": lambda: [torch.zeros(16, 512, 2048, device=dev) for _ in range(10)],
}
with torch.no_grad():
    ref = ops.conv2d_winograd_pre(x, ps, in_norm=st, prelu=slope)
    torch.cuda.synchronize()
    for name, fn in neighbours.items():
        worst = 0.0
        for it in range(10):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                outs = [ops.conv2d_winograd_pre(x, ps, in_norm=st, prelu=slope) for _ in range(6)]
            keep = fn()
            main.wait_stream(side)
            torch.cuda.synchronize()
            worst = max(worst, max((o - ref).abs().max().item() for o in outs))
        print(f"neighbour on the main stream: {name:20s} worst |diff| of the side-stream results {worst:.3e}", flush=True)
