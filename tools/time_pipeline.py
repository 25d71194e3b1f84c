"""Consecutive gen_img batches on alternating HIP streams (whole batches, per-stream host state): does the latency-bound 4^2-32^2 stack of one
batch hide under the large layers of the previous one?   python tools/time_pipeline.py [bs]"""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import e4s2024_amd
from e4s2024_amd import seeded, ops
e4s2024_amd.install()
from models.networks import Net3
dev = "cuda:0"
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opts = argparse.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net.G, 4, "net3", prefix="G."); la = seeded.seeded_latent_avg(2, 18); net.latent_avg = la.to(dev); net = net.to(dev)
ops.STRICT_MASK = False
codes = seeded.seeded_codes(1, bs, 12, 18, la).to(dev)
mask = seeded.labels_to_onehot(seeded.blocky_labels(3, bs, 12, 512, 16), 12).to(dev)
streams = [torch.cuda.Stream() for _ in range(3)]


def run(nstream, n):
    main = torch.cuda.current_stream()
    outs = []
    if nstream == 1:
        for _ in range(n):
            outs.append(net.gen_img(None, codes, mask.clone(), randomize_noise=False)[0])
        return outs
    for st in streams[:nstream]:
        st.wait_stream(main)
    for i in range(n):
        with torch.cuda.stream(streams[i % nstream]):
            outs.append(net.gen_img(None, codes, mask.clone(), randomize_noise=False)[0])
    for st in streams[:nstream]:
        main.wait_stream(st)
    return outs


with torch.no_grad():
    ref = run(1, 1)[0]
    for ns in (1, 2, 3, 1, 2):
        outs = run(ns, 4); torch.cuda.synchronize()
        d = max((o - ref).abs().max().item() for o in outs)
        del outs
        run(ns, 6); torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 40
        run(ns, n)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"bs {bs}: {ns} stream(s): {dt * 1e3:.3f} ms/step = {bs / dt:.0f} faces/s   max|diff| vs one stream {d:.1e}", flush=True)
