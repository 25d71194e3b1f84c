import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/e4s2024_amd") else os.getcwd())
import torch
from e4s2024_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
bs, cin, cout, h = 8, 512, 512, 32
xs = [torch.randn(bs, cin, h, h, device=dev) for _ in range(2)]
w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
sts = [ops.plane_stats(x, 1e-5) for x in xs]
slope = torch.rand(cout, device=dev)
ps = ops.PreparedWinogradSplit()
with torch.no_grad():
    ref = [ops.conv2d_winograd_pre(x, ps.get(w), in_norm=st, prelu=slope) for x, st in zip(xs, sts)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad = 0
    for it in range(20):
        outs = []
        for i, st_ in enumerate(streams):
            st_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st_):
                o = xs[i]
                for _ in range(3):
                    o2 = ops.conv2d_winograd_pre(xs[i], ps.get(w), in_norm=sts[i], prelu=slope)
                outs.append(o2)
        for st_ in streams:
            torch.cuda.current_stream().wait_stream(st_)
        torch.cuda.synchronize()
        for i in range(2):
            if not torch.equal(outs[i], ref[i]):
                bad += 1
                print("iter", it, "stream", i, "max diff", (outs[i] - ref[i]).abs().max().item())
    print("mismatches:", bad)
