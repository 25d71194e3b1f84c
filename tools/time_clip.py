"""The clip measurement of bench.py (BASELINE configs[4] on one GPU: runner.run_clip_streamed over pipeline.swap_batch(mask_surgery=True)) several times in ONE process:
how much does it vary from run to run?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import default_opts, install_dropin
install_dropin()
from models.networks import Net3
from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
from e4s2024_amd import ops, seeded, pipeline, runner as _runner
dev = torch.device("cuda", 0)
net = Net3(default_opts()); seeded.apply_seeded(net, 4, "net3"); net = net.to(dev).eval()
la = seeded.seeded_latent_avg(2, 18); net.latent_avg = la.to(dev)
parser = FaceParser(seg_ckpt=None, device=dev); seeded.apply_seeded(parser.seg, 7, "bisenet"); parser.seg.eval()
ops.STRICT_MASK = False
rn = _runner.FrameShardRunner(device=dev)
POOL, cb, n_frames = 16, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 256
pool_d = seeded.seeded_image(50, POOL, 1024).to(dev); pool_t = seeded.seeded_image(60, POOL, 1024).to(dev)
def frame_inputs(lo, hi):
    idx = torch.arange(lo, hi, device=dev) % POOL
    return pool_d.index_select(0, idx), pool_t.index_select(0, idx)
def synth(shared, fi):
    net.latent_avg = shared
    return pipeline.swap_batch(net, parser, fi[0], fi[1], mask_surgery=True)[0]
out_buf = torch.empty((n_frames, 1024, 1024, 3), dtype=torch.uint8, device=dev)
with torch.no_grad():
    synth(la.to(dev), frame_inputs(0, cb))
torch.cuda.synchronize()
pause = float(os.environ.get("E4S_CLIP_PAUSE", "0"))
for rep in range(8):
    if pause:
        time.sleep(pause)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rn.run_clip_streamed(n_frames, la.to(dev), frame_inputs, synth, batch=cb, out=out_buf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"run {rep}: {n_frames / dt:.1f} frames/s ({dt * 1e3:.0f} ms)   stream contexts {len(ops._ctxs)}  mem {torch.cuda.memory_allocated() >> 20} MiB reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
