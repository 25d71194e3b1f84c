import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import default_opts, install_dropin
install_dropin()
from models.networks import Net3
from e4s2024_amd import ops, seeded
dev="cuda:0"
net = Net3(default_opts()); seeded.apply_seeded(net.encoder, 4, "net3", prefix="encoder."); net = net.to(dev).eval()
img = seeded.seeded_image(5, 16, 1024).to(dev)
lab = torch.from_numpy(seeded.blocky_labels(3, 16, 12, 512, 16)).to(dev).to(torch.uint8)
with torch.no_grad():
    net.get_style_vectors(img, lab)
    with ops.KernelTimer() as kt:
        net.get_style_vectors(img, lab)
    tot = 0
    for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]["ms"] if isinstance(kv[1], dict) else 0)[:25]:
        print(k, v)
    for name in ("conv3x3_mx<3>", "conv3x3_s2_mx<3>"):
        for d, (c, t) in sorted(kt.by_detail(name).items()):
            print(f"{name:20s} {d:18s} calls {c:3d}  {1e3 * t / c:7.1f} us each")
