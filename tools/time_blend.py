"""Row f3 measurement: the paste-back (mask resize + alpha paste + ten-level multi-band blend) per frame at batch 1 and 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e4s2024_amd import ops, pipeline, seeded

dev = "cuda:0"
for bs in (1, 8):
    g = torch.Generator(device=dev).manual_seed(0)
    sw = torch.randint(0, 256, (bs, 1024, 1024, 3), device=dev, generator=g, dtype=torch.uint8)
    tg = torch.randint(0, 256, (bs, 1024, 1024, 3), device=dev, generator=g, dtype=torch.uint8)
    lab = torch.from_numpy(seeded.blocky_labels(5, bs, 12, 512, 16)).to(dev)
    content, border, _ = ops.foreground_masks(lab, None, 5)
    for _ in range(3):
        pipeline.paste_back(sw, tg, content, border)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        out = pipeline.paste_back(sw, tg, content, border)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    # algorithmic bytes per frame: the three fp32 pyramids (A, B, mask: 3 planes each, 4/3 of the base level) are written once and read twice
    # (Laplacian level, blend), the reconstruction chain once more, uint8 frames in / out
    algo = 3 * 3 * 1024 * 1024 * 4 * (4 / 3) * 3 + 3 * 1024 * 1024 * 4 * (4 / 3) * 2 + 3 * 3 * 1024 * 1024
    print(f"paste_back batch {bs}: {ms:.3f} ms per call, {ms / bs:.3f} ms per frame, {algo * bs / (ms * 1e-3) / 1e12:.2f} TB/s of algorithmic bytes ({algo / 1e6:.0f} MB per frame)")
