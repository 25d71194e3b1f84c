"""se_gate launch time at the encoder's shapes (16 images)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from e4s2024_amd import ops
dev = "cuda:0"
for C in (64, 128, 256, 512):
    pooled = torch.randn(16, C, device=dev); w1 = torch.randn(C // 16, C, 1, 1, device=dev); w2 = torch.randn(C, C // 16, 1, 1, device=dev)
    for _ in range(3): ops.se_gate(pooled, w1, w2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): ops.se_gate(pooled, w1, w2)
    b.record(); torch.cuda.synchronize()
    print(f"C={C}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per launch")
