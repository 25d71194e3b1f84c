"""The encoder's stride-1 3x3 convolutions at the full swap's batch (16 images): direct split-bf16 kernel against e4s_conv3x3_mx (both arithmetics)
and the two-phase kernel e4s_conv3x3_mx3 (f16 + fp6)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s2024_amd import ops
dev = "cuda:0"
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for cin, cout, h, count in [(512, 512, 32, 27), (256, 256, 64, 6), (256, 512, 64, 1), (128, 128, 128, 4), (128, 256, 128, 1), (64, 128, 256, 1)]:
    g = torch.Generator(device=dev).manual_seed(cin + h)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / (cin * 9) ** 0.5
    mean, rstd = x.mean((2, 3)), 1.0 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
    slope = torch.rand(cout, device=dev, generator=g)
    pc = ops.PreparedConv().get(w)
    wm = {a: ops.PreparedMx().get(w, None, False, a) for a in (0, 1, 3)}
    calls = {"direct": lambda: ops.conv2d(x, pc, 1, 1, in_norm=(mean, rstd), prelu=slope),
             "mx/bf16x3": lambda: ops.conv3x3_mx(x, wm[0], 0, cout, in_norm=(mean, rstd), prelu=slope),
             "mx/f16+fp6": lambda: ops.conv3x3_mx(x, wm[1], 1, cout, in_norm=(mean, rstd), prelu=slope),
             "mx3": lambda: ops.conv3x3_mx(x, wm[3], 3, cout, in_norm=(mean, rstd), prelu=slope)}
    outs = {k: f() for k, f in calls.items()}
    t = {k: [] for k in calls}
    for rnd in range(7):
        for k, f in calls.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                f()
            b.record(); torch.cuda.synchronize()
            t[k].append(a.elapsed_time(b) / 5)
    med = {k: statistics.median(v) for k, v in t.items()}
    gf = 2.0 * cin * cout * 9 * h * h * bs / 1e9
    sc = outs["direct"].abs().max().item()
    print(f"{cin:3d}->{cout:3d} @{h:3d} x{count:2d}: " + "  ".join(f"{k} {med[k]:.4f} ms ({gf / med[k]:.0f} TF/s)" for k in med)
          + f"   |mx0 - direct| {(outs['mx/bf16x3'] - outs['direct']).abs().max().item() / sc:.1e}  |mx1 - direct| {(outs['mx/f16+fp6'] - outs['direct']).abs().max().item() / sc:.1e}  |mx3 - direct| {(outs['mx3'] - outs['direct']).abs().max().item() / sc:.1e}", flush=True)
