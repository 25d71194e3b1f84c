"""Long soak of e4s_conv3x3_mx3 under two-stream pressure (tests/test_gpu_mx.py runs the bounded version): every result must equal the single-stream one."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from e4s2024_amd import ops
dev = "cuda:0"
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
g = torch.Generator(device=dev).manual_seed(3)
shapes = [(16, 512, 512, 32), (16, 256, 256, 64), (16, 64, 128, 256), (5, 96, 136, 40)]
cases = []
with torch.no_grad():
    for bs, cin, cout, h in shapes:
        x = torch.randn(bs, cin, h, h, device=dev, generator=g); w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * 0.03
        mean, rstd = x.mean((2, 3)), 1.0 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
        w3 = ops.PreparedMx().get(w, None, False, 3)
        ref = ops.conv3x3_mx(x, w3, 3, cout, in_norm=(mean, rstd)).clone()
        cases.append((x, w3, cout, mean, rstd, ref))
    hog_a = torch.randn(64 << 20, device=dev); hog_b = torch.empty_like(hog_a)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    t0, launches, bad = time.time(), 0, 0
    while time.time() - t0 < secs:
        outs = []
        with torch.cuda.stream(s1):
            for x, w3, cout, mean, rstd, ref in cases:
                for _ in range(3):
                    outs.append((ops.conv3x3_mx(x, w3, 3, cout, in_norm=(mean, rstd)), ref))
        with torch.cuda.stream(s2):
            for _ in range(3):
                hog_b.copy_(hog_a)
                x, w3, cout, mean, rstd, ref = cases[launches % len(cases)]
                outs.append((ops.conv3x3_mx(x, w3, 3, cout, in_norm=(mean, rstd)), ref))
        torch.cuda.synchronize()
        bad += sum(not torch.equal(o, r) for o, r in outs)
        launches += len(outs)
print(f"{launches} launches in {secs:.0f} s beside a second stream: {bad} results differ from the single-stream reference; overflow flag {ops.mx_overflowed()}")
