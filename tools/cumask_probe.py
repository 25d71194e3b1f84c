"""Does the synthesis step need all 256 CUs?  At the board's power limit fewer active CUs run at a higher clock: the one-stream step on a stream whose CU mask
leaves N CUs of every XCD out (hipExtStreamCreateWithCUMask), against the full chip.  usage: python tools/cumask_probe.py
Measured (round 6): 3.02 ms per step on 256 CUs, 4.24 ms on 248 / 240 / 224 alike — the step's grids are cut for 256 CUs (the persistent chain kernels launch one
workgroup per CU, the 64^2 masked layer is exactly 256 workgroups at batch 4): on fewer CUs those launches take a second round.  Reserving CUs for the other
stream's latency-bound head is therefore not an option at this batch size."""
import argparse as _ap, ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import e4s2024_amd
from e4s2024_amd import ops, seeded
e4s2024_amd.install()
from models.networks import Net3
dev = torch.device("cuda", 0)
opts = _ap.Namespace(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False, start_from_latent_avg=True, learn_in_w=False)
net = Net3(opts).eval(); seeded.apply_seeded(net.G, 4, "net3", prefix="G.")
la = seeded.seeded_latent_avg(2, 18); net.latent_avg = la.to(dev); net = net.to(dev)
codes = seeded.seeded_codes(1, 4, 12, 18, la).to(dev)
mask = seeded.labels_to_onehot(seeded.blocky_labels(3, 4, 12, 512, 16), 12).to(dev)
ops.STRICT_MASK = False
scope = ops.mx_guard_scope(); scope.__enter__()
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(drop_per_xcd):
    """A stream that may use every CU except the first `drop_per_xcd` of each XCD.  Bit i of the mask = CU i; consecutive CU ids go round-robin over the 8 XCDs
    (CU id = xcd + 8 * index-in-xcd on this part, as the dispatch order suggests): dropping ids 0 .. 8 * n - 1 takes n CUs from every XCD."""
    words = (ctypes.c_uint32 * 8)(*([0xffffffff] * 8))
    for cu in range(8 * drop_per_xcd):
        words[cu // 32] &= ~(1 << (cu % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def step():
    with torch.no_grad():
        return net.gen_img(None, codes, mask.view_as(mask), randomize_noise=False)[0]


for drop in (0, 0, 1, 2, 4, 0, 2):
    st = masked_stream(drop) if drop else torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
    print(f"CUs left out per XCD {drop} ({256 - 8 * drop} CUs): {dt * 1e3:.3f} ms per step, {4 / dt:.1f} faces/s", flush=True)
