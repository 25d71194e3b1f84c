"""Does splitting a gen_img batch over two HIP streams (half batches, per-stream host state) beat one stream?  Tuning probe."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, ops
e4s2024_amd.install()
from models.networks import Net3
dev = "cuda:0"
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net.G, 4, "net3", prefix="G."); la = seeded.seeded_latent_avg(2, 18); net.latent_avg = la.to(dev); net = net.to(dev)
ops.STRICT_MASK = False
for bs in (4, 8):
    codes = seeded.seeded_codes(1, bs, 12, 18, la).to(dev)
    mask = seeded.labels_to_onehot(seeded.blocky_labels(3, bs, 12, 512, 16), 12).to(dev)
    streams = [torch.cuda.Stream() for _ in range(4)]

    def run(nsplit):
        main = torch.cuda.current_stream()
        if nsplit == 1:
            return net.gen_img(None, codes, mask.view_as(mask), randomize_noise=False)[0]
        outs = []
        step = bs // nsplit
        for i in range(nsplit):
            st = streams[i]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(net.gen_img(None, codes[i * step:(i + 1) * step], mask[i * step:(i + 1) * step], randomize_noise=False)[0])
        for st in streams[:nsplit]:
            main.wait_stream(st)
        return torch.cat(outs)

    with torch.no_grad():
        ref = run(1)
        for ns in (1, 2, 4):
            if bs % ns:
                continue
            out = run(ns); torch.cuda.synchronize()
            d = (out - ref).abs().max().item()
            for _ in range(3):
                run(ns)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                run(ns)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print(f"bs {bs}: {ns} stream(s): {dt * 1e3:.3f} ms/step = {bs / dt:.0f} faces/s   max|diff| vs one stream {d:.1e}", flush=True)
