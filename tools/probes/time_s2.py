"""The encoder's four stride-2 3x3 convolutions at 16 images on the direct split-bf16 kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from e4s2024_amd import ops
dev = "cuda:0"
for c, h in [(64, 256), (128, 128), (256, 64), (512, 32)]:
    x = torch.randn(16, c, h, h, device=dev); w = torch.randn(c, c, 3, 3, device=dev) * 0.02
    pc = ops.PreparedConv().get(w)
    with torch.no_grad():
        for _ in range(3): ops.conv2d(x, pc, 2, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): ops.conv2d(x, pc, 2, 1)
        b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    gf = 2.0 * c * c * 9 * (h // 2) ** 2 * 16 / 1e9
    print(f"{c}->{c} s2 @{h}->{h//2}: {ms:.4f} ms  {gf/ms:.0f} TF/s")
