"""Two parse -> encode chains on two streams (as pipeline.swap_batch(two_streams=True)): are the style vectors run-to-run identical?"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, ops
e4s2024_amd.install()
from models.networks import Net3
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net, 4, "net3"); net.latent_avg = seeded.seeded_latent_avg(2, 18).to(dev); net = net.to(dev)
ops.STRICT_MASK = False
imgs = [seeded.seeded_image(5, 8, 1024).to(dev), seeded.seeded_image(6, 8, 1024).to(dev)]
labs = [torch.from_numpy(seeded.blocky_labels(3 + i, 8, 12, 512, 16)).to(dev).to(torch.uint8) for i in range(2)]
side = torch.cuda.Stream()
with torch.no_grad():
    ref = [net.get_style_vectors(imgs[i], labs[i])[0].clone() for i in range(2)]
    torch.cuda.synchronize()
    for it in range(int(os.environ.get("ITERS", "6"))):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            v0 = net.get_style_vectors(imgs[0], labs[0])[0]
        v1 = net.get_style_vectors(imgs[1], labs[1])[0] if os.environ.get('ALONE') != '1' else ref[1]
        main.wait_stream(side)
        torch.cuda.synchronize()
        print(it, "side-stream chain max|diff| vs sequential:", (v0 - ref[0]).abs().max().item(), " main-stream chain:", (v1 - ref[1]).abs().max().item(), flush=True)
