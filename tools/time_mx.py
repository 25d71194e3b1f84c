"""The masked layers of a synthesis step (batch 4, the benchmark's blocky maps) on the three routes of E4S_MX: 0 = register-staged kernel
(modconv_sb.hip), 1 = DMA pipeline with the split-bf16 arithmetic (bit-identical), 2 = DMA pipeline with f16 + 2 x MX fp6.  Interleaved rounds,
median ms per launch; the error of each against an fp64 evaluation of the same layer on a sample of pixels."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s2024_amd import ops, seeded

dev = "cuda:0"
LAYERS = [(512, 512, 32, False), (512, 512, 32, True), (512, 512, 64, False), (512, 256, 64, True), (256, 256, 128, False), (256, 128, 128, True), (128, 128, 256, False)]
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
labels = torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, 16)).to(dev)
blur = torch.tensor([1., 3., 3., 1.], device=dev); blur = blur[:, None] * blur[None, :]; blur = blur / blur.sum() * 4
total = {0: 0.0, 1: 0.0, 2: 0.0}
for cin, cout, h, up in LAYERS:
    g = torch.Generator(device=dev).manual_seed(cin + h + up)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(1, cout, cin, 3, 3, device=dev, generator=g)
    s = 1.0 + 0.3 * torch.randn(bs, 12, cin, device=dev, generator=g)
    d = torch.rand(bs, 12, cout, device=dev, generator=g) + 0.5
    ho = 2 * h if up else h
    nz = torch.randn(bs, 1, ho, ho, device=dev, generator=g); nw = torch.tensor([0.1], device=dev); ab = torch.zeros(cout, device=dev)
    wt, _ = ops.PreparedWeights().get(w, blur if up else None, up, True)
    mx = {0: None, 1: (ops.PreparedMx().get(w, blur if up else None, up, 0), 0), 2: (ops.PreparedMx().get(w, blur if up else None, up, 1), 1)}
    call = lambda m: ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, up, mx=mx[m])
    outs = {m: call(m) for m in (0, 1, 2)}
    times = {m: [] for m in (0, 1, 2)}
    for rnd in range(7):
        for m in (0, 1, 2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                call(m)
            b.record(); torch.cuda.synchronize()
            times[m].append(a.elapsed_time(b) / 5)
    med = {m: statistics.median(times[m]) for m in times}
    for m in med:
        total[m] += med[m]
    gf = 2.0 * cin * cout * 9 * h * h * bs / 1e9
    scale = outs[0].abs().max().item()
    print(f"{cin:3d}->{cout:3d} @{h:3d}{' up' if up else '   '}: sb {med[0]:.4f} ms ({gf / med[0]:.0f} alg TF/s)  mx/bf16x3 {med[1]:.4f} ms ({gf / med[1]:.0f})  mx/f16+fp6 {med[2]:.4f} ms ({gf / med[2]:.0f})"
          f"   bit-identical(1 vs 0): {torch.equal(outs[0], outs[1])}   |2 - 0|max {(outs[2] - outs[0]).abs().max().item():.2e} of {scale:.1f}", flush=True)
print(f"sum of the seven launches: sb {total[0]:.3f} ms   mx/bf16x3 {total[1]:.3f} ms   mx/f16+fp6 {total[2]:.3f} ms    f16 overflow flag: {ops.mx_overflowed()}")
