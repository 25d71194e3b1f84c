"""The regional-style encoder alone (no parser beside it): 16 faces at 1024 x 1024 -> style vectors, wall time per call; under rocprofv3 (tools/prof_encoder.sh)
the kernel table shows what each of its launches costs when nothing shares the chip."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import default_opts, install_dropin
install_dropin()
from models.networks import Net3
from e4s2024_amd import ops, seeded
dev = "cuda:0"
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = Net3(default_opts()); seeded.apply_seeded(net.encoder, 4, "net3", prefix="encoder."); net = net.to(dev).eval()
img = seeded.seeded_image(5, bs, 1024).to(dev)
lab = torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, 16)).to(dev).to(torch.uint8)
def timed(n=8):
    for _ in range(3):
        net.get_style_vectors(img, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net.get_style_vectors(img, lab)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad(), ops.mx_guard_scope() as g:
    print(f"get_style_vectors x {bs} faces: {timed():.3f} ms per call")
