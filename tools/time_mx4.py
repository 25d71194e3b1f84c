"""Masked up layers (batch 4) with and without the four-parity kernel (csrc/modconv_mx4.hip), interleaved rounds, median ms per call."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s2024_amd import ops, seeded

dev = "cuda:0"
bs = 4
blur = torch.tensor([1., 3., 3., 1.], device=dev); blur = blur[:, None] * blur[None, :]; blur = blur / blur.sum() * 4
maps = {"blocky": torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, 16)).to(dev), "coarse": torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, 4)).to(dev),
        "portrait": torch.from_numpy(seeded.facelike_labels(5, bs, 512)).to(dev).to(torch.uint8), "iid": torch.from_numpy(seeded.iid_labels(9, bs, 12, 512)).to(dev).to(torch.uint8)}
for cin, cout, h in [(512, 256, 64), (512, 512, 32), (256, 128, 128)]:
    g = torch.Generator(device=dev).manual_seed(cin + h)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(1, cout, cin, 3, 3, device=dev, generator=g)
    s = 1.0 + 0.3 * torch.randn(bs, 12, cin, device=dev, generator=g)
    d = torch.rand(bs, 12, cout, device=dev, generator=g) + 0.5
    nz = torch.randn(bs, 1, 2 * h, 2 * h, device=dev, generator=g); nw = torch.tensor([0.1], device=dev); ab = torch.zeros(cout, device=dev)
    wt, _ = ops.PreparedWeights().get(w, blur, True, True)
    wmx = ops.PreparedMx().get(w, blur, True, 1); wmx4 = ops.PreparedMx().get(w, blur, True, 4)
    for name, labels in maps.items():
        call = {0: lambda: ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, True, mx=(wmx, 1)),
                1: lambda: ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, True, mx=(wmx, 1), mx4=wmx4)}
        same = torch.equal(call[0](), call[1]())
        ts = {0: [], 1: []}
        for rnd in range(7):
            for m in (0, 1):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    call[m]()
                b.record(); torch.cuda.synchronize()
                ts[m].append(a.elapsed_time(b) / 5)
        print(f"{cin}->{cout} @{h} up, {name:8s}: composed {statistics.median(ts[0]):.4f} ms   with mx4 {statistics.median(ts[1]):.4f} ms   equal {same}", flush=True)
