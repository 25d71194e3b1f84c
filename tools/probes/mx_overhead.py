"""Fixed cost per workgroup of the masked mx kernel: the same geometry (cout 128, 256^2 map, batch 4, blocky labels) at 1, 2, 4, 8 chunks of input channels;
the intercept of time over chunks is prologue + epilogue + launch."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from e4s2024_amd import ops, seeded
dev = "cuda:0"
bs, cout, h = 4, 128, 256
labels = torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, 16)).to(dev)
for cin in (16, 32, 64, 128, 256, 512):
    g = torch.Generator(device=dev).manual_seed(cin)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(1, cout, cin, 3, 3, device=dev, generator=g)
    s = 1.0 + 0.3 * torch.randn(bs, 12, cin, device=dev, generator=g)
    d = torch.rand(bs, 12, cout, device=dev, generator=g) + 0.5
    nz = torch.randn(bs, 1, h, h, device=dev, generator=g); nw = torch.tensor([0.1], device=dev); ab = torch.zeros(cout, device=dev)
    wt, _ = ops.PreparedWeights().get(w, None, False, True)
    mx = (ops.PreparedMx().get(w, None, False, 1), 1)
    call = lambda: ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, False, mx=mx)
    for _ in range(3): call()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): call()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
    print(f"cin {cin:3d} ({cin // 16:2d} chunks): {statistics.median(ts) * 1e3:.1f} us")
