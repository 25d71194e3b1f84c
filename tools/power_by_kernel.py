"""Which kernels of the synthesis step run at the board's power limit?  Each kernel class is launched back to back for SECONDS while a thread samples
`rocm-smi --showpower --showclocks` (package power, shader clock): a kernel below the cap at the full 2.4 GHz is bound by something else (latency, LDS, HBM)."""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e4s2024_amd import ops, seeded

dev = "cuda:0"
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
bs = 4


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
            ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            if pw and ck:
                out.append((float(pw.group(1)), int(ck.group(1))))
        except Exception:
            pass
        time.sleep(0.25)


def run(name, fn, flop=None):
    fn(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    out = out[len(out) // 3:] or out                      # (settled part)
    pw = sum(o[0] for o in out) / max(1, len(out)); ck = sum(o[1] for o in out) / max(1, len(out))
    extra = f"  {flop / (dt / n) / 1e12:7.0f} alg TF/s" if flop else ""
    print(f"{name:42s} {dt / n * 1e3:8.4f} ms  {pw:7.0f} W  {ck:6.0f} MHz  ({len(out)} samples){extra}", flush=True)


labels = torch.from_numpy(seeded.blocky_labels(3, bs, 12, 512, 16)).to(dev)
blur = torch.tensor([1., 3., 3., 1.], device=dev); blur = blur[:, None] * blur[None, :]; blur = blur / blur.sum() * 4
# ---- masked layers (the default f16 + fp6 route)
for cin, cout, h, up in [(512, 512, 64, False), (256, 256, 128, False), (128, 128, 256, False), (512, 256, 64, True), (512, 512, 32, False)]:
    g = torch.Generator(device=dev).manual_seed(cin + h + up)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(1, cout, cin, 3, 3, device=dev, generator=g)
    s = 1.0 + 0.3 * torch.randn(bs, 12, cin, device=dev, generator=g)
    d = torch.rand(bs, 12, cout, device=dev, generator=g) + 0.5
    ho = 2 * h if up else h
    nz = torch.randn(bs, 1, ho, ho, device=dev, generator=g); nw = torch.tensor([0.1], device=dev); ab = torch.zeros(cout, device=dev)
    wt, _ = ops.PreparedWeights().get(w, blur if up else None, up, True)
    for arith, nm in ((1, "f16+fp6"), (0, "bf16x3")):
        mx = (ops.PreparedMx().get(w, blur if up else None, up, arith), arith)
        run(f"masked {cin}->{cout} @{h}{' up' if up else ''} {nm}", lambda: ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, up, mx=mx),
            2.0 * cin * cout * 9 * h * h * bs)
    del x, w


# ---- the single-region chain
def chain_conv(c, res, want_out):
    torch.manual_seed(0)
    x = torch.randn(bs, c, res, res, device=dev)
    w = torch.randn(1, c, c, 3, 3, device=dev)
    styles = torch.randn(bs, 1, 512, device=dev)
    mw, mb = torch.randn(c, 512, device=dev), torch.ones(c, device=dev)
    with torch.no_grad():
        wt, wsq = ops.PreparedWeights().get(w, None, False, True)
        s, d = ops.style_demod(styles, mw, mb, wsq, c)
    noise = torch.randn(1, 1, res, res, device=dev)
    nw, ab = torch.tensor([0.1], device=dev), torch.randn(c, device=dev)
    rw = torch.randn(1, 3, c, 1, 1, device=dev)
    with torch.no_grad():
        r_wt, _ = ops.PreparedWeights().get(rw, None, False, False)
    rgb = (r_wt, torch.randn(bs, 1, c, device=dev), torch.randn(1, 3, 1, 1, device=dev), torch.randn(bs, 3, res // 2, res // 2, device=dev), blur)
    s_next = torch.randn(bs, 1, c, device=dev)
    xsp = ops.to_split_planes(x, s)
    del x
    run(f"chain conv {c}->{c} @{res}", lambda: ops.chain_conv3x3(xsp, wt, d, noise, nw, ab, True, c, s_next=s_next if want_out else None, rgb=rgb), 2.0 * c * c * 9 * res * res * bs)


def chain_up(cin, cout, res):
    torch.manual_seed(1)
    x = torch.randn(bs, cin, res, res, device=dev)
    w = torch.randn(1, cout, cin, 3, 3, device=dev)
    styles = torch.randn(bs, 1, 512, device=dev)
    mw, mb = torch.randn(cin, 512, device=dev), torch.ones(cin, device=dev)
    with torch.no_grad():
        wt, wsq = ops.PreparedWeights().get(w, None, False, True, tconv=True)
        s, d = ops.style_demod(styles, mw, mb, wsq, cout)
    noise = torch.randn(1, 1, 2 * res, 2 * res, device=dev)
    nw, ab = torch.tensor([0.1], device=dev), torch.randn(cout, device=dev)
    s_next = torch.randn(bs, 1, cout, device=dev)
    xsp = ops.to_split_planes(x, s)
    hc = ops.PreparedHc().get(w, blur)
    del x
    run(f"chain up {cin}->{cout} @{res}->{2 * res}", lambda: ops.modconv_up_single(xsp, wt, s, d, blur, noise, nw, ab, True, cout, s_next=s_next, hc=hc), 2.0 * cin * cout * 9 * res * res * bs)


chain_conv(32, 1024, False)
chain_conv(64, 512, True)
chain_up(64, 32, 512)
chain_up(128, 64, 256)
# ---- a copy kernel for scale: HBM-bound, no matrix work
a = torch.empty(1 << 28, dtype=torch.float32, device=dev); b_ = torch.empty_like(a)
run("torch copy 1 GiB -> 1 GiB", lambda: b_.copy_(a))
