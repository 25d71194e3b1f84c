"""The masked layers below 32 x 32 of a synthesis step (batch 4, the benchmark's maps): ms per launch (split-K finalize included), median of 7 rounds x 10."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s2024_amd import ops, seeded

dev = "cuda:0"
bs = 4
labels = torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, 16)).to(dev)
blur = torch.tensor([1., 3., 3., 1.], device=dev); blur = blur[:, None] * blur[None, :]; blur = blur / blur.sum() * 4
res = []
for cin, cout, h, up in [(512, 512, 4, False), (512, 512, 4, True), (512, 512, 8, False), (512, 512, 8, True), (512, 512, 16, False), (512, 512, 16, True)]:
    g = torch.Generator(device=dev).manual_seed(cin + h + up)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(1, cout, cin, 3, 3, device=dev, generator=g)
    s = 1.0 + 0.3 * torch.randn(bs, 12, cin, device=dev, generator=g)
    d = torch.rand(bs, 12, cout, device=dev, generator=g) + 0.5
    ho = 2 * h if up else h
    nz = torch.randn(bs, 1, ho, ho, device=dev, generator=g); nw = torch.tensor([0.1], device=dev); ab = torch.zeros(cout, device=dev)
    wt, _ = ops.PreparedWeights().get(w, blur if up else None, up, True)
    call = lambda: ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, up)
    ref = call()
    ts = []
    for rnd in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            call()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    res.append(f"{h}{'up' if up else ''} {statistics.median(ts) * 1e3:.1f}us (sum {ref.double().sum().item():.6e})")
print(os.environ.get("E4S_SB_W16_BIG", "-"), "  ".join(res), flush=True)
