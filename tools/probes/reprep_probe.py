"""Which prepared weight copies are rebuilt on every parser / encoder call?  (A rebuild in steady state = a cache that is being shared or invalidated.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import default_opts, install_dropin
install_dropin()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
from e4s2024_amd import ops, seeded, pipeline
dev = torch.device("cuda", 0)
net = Net3(default_opts()); seeded.apply_seeded(net, 4, "net3"); net = net.to(dev).eval(); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev)
parser = FaceParser(seg_ckpt=None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
ops.STRICT_MASK = False
d = seeded.seeded_image(50, 8, 1024).to(dev); t = seeded.seeded_image(60, 8, 1024).to(dev)
with torch.no_grad():
    for _ in range(2):
        pipeline.swap_batch(net, parser, d, t, mask_surgery=True)
    torch.cuda.synchronize()
    import traceback
    orig = ops._Prepared._publish
    def spy(self, key, payload):
        fr = [f for f in traceback.extract_stack()[:-1] if "e4s2024_amd" in f.filename][-3:]
        print("REBUILD", type(self).__name__, [f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr], str(key)[:160])
        return orig(self, key, payload)
    ops._Prepared._publish = spy
    pipeline.swap_batch(net, parser, d, t, mask_surgery=True)
    torch.cuda.synchronize()
print("done")
