"""Which host calls issue device copies inside one encoder call (torch.profiler, with stacks)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, ops
e4s2024_amd.install()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev); net = net.to(dev)
parser = FaceParser(None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
img = seeded.seeded_image(5, 2, 1024).to(dev)
ops.STRICT_MASK = False
with torch.no_grad():
    lab = parser.parse_batch((img + 1) / 2, seg12=True)
    for _ in range(2): net.get_style_vectors(img, lab)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        net.get_style_vectors(img, lab)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=25, max_name_column_width=60, max_src_column_width=110))
