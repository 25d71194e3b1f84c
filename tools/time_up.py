"""Times the single-region up layers (256->512, 512->1024) at batch 4: fused launch vs tconv + blur epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s2024_amd import ops

dev = "cuda:0"
def run(cin, cout, h, bs=4, iters=20):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(1, cout, cin, 3, 3, device=dev, generator=g)
    blur = torch.tensor([1., 3., 3., 1.], device=dev); blur = (blur[:, None] * blur[None, :]); blur = blur / blur.sum() * 4
    s = torch.randn(bs, 1, cin, device=dev, generator=g); d = torch.rand(bs, 1, cout, device=dev, generator=g)
    nz = torch.randn(bs, 1, 2 * h, 2 * h, device=dev, generator=g); nw = torch.tensor([0.1], device=dev); ab = torch.zeros(cout, device=dev)
    pw = ops.PreparedWeights()
    wt, _ = pw.get(w, None, False, True, tconv=True)
    res = {}
    for fused in (True, False):
        ops.UP_FUSED = fused
        for _ in range(3):
            o = ops.modconv_up_single(x, wt, s, d, blur, nz, nw, ab, True, cout)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            o = ops.modconv_up_single(x, wt, s, d, blur, nz, nw, ab, True, cout)
        b.record(); torch.cuda.synchronize()
        res[fused] = (a.elapsed_time(b) / iters, o)
    err = (res[True][1] - res[False][1]).abs().max().item()
    gb = bs * (cin * h * h + cout * 4 * h * h) * 4 / 1e9
    print(f"cin={cin} cout={cout} {h}->{2*h}: fused {res[True][0]:.3f} ms ({gb / res[True][0] * 1e3:.0f} GB/s alg)  two-stage {res[False][0]:.3f} ms  maxdiff {err:.2e}")

run(128, 64, 256)
run(64, 32, 512)
