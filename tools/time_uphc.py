"""The chain's up layers: half-composed kernel (csrc/modconv_uphc.hip) against the fused LDS-DMA kernel (csrc/modconv_upfused.hip), interleaved in one run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e4s2024_amd import ops

dev = "cuda:0"


def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def case(cin, cout, res, bs=4):
    torch.manual_seed(1)
    x = torch.randn(bs, cin, res, res, device=dev)
    w = torch.randn(1, cout, cin, 3, 3, device=dev)
    styles = torch.randn(bs, 1, 512, device=dev)
    mw, mb = torch.randn(cin, 512, device=dev), torch.ones(cin, device=dev)
    with torch.no_grad():
        wt, wsq = ops.PreparedWeights().get(w, None, False, True, tconv=True)
        s, d = ops.style_demod(styles, mw, mb, wsq, cout)
    k1 = torch.tensor([1., 3., 3., 1.], device=dev)
    blur = k1[:, None] * k1[None, :] / k1.sum() ** 2 * 4
    noise = torch.randn(1, 1, 2 * res, 2 * res, device=dev)
    nw, ab = torch.tensor([0.1], device=dev), torch.randn(cout, device=dev)
    s_next = torch.randn(bs, 1, cout, device=dev)
    hc = ops.PreparedHc().get(w, blur)
    xsp = ops.to_split_planes(x, s)
    old = lambda: ops.modconv_up_single(xsp, wt, s, d, blur, noise, nw, ab, True, cout, s_next=s_next)
    new = lambda: ops.modconv_up_single(xsp, wt, s, d, blur, noise, nw, ab, True, cout, s_next=s_next, hc=hc)
    a, b = ops.from_split_planes(old()), ops.from_split_planes(new())
    print(f"up {cin}->{cout} @ {res}->{2 * res} bs {bs}: max|diff| {(a - b).abs().max().item():.3e} (|v|max {a.abs().max().item():.1f})", flush=True)
    rounds = [(t(old), t(new)) for _ in range(3)]
    gb = bs * (cin * res * res + cout * 4 * res * res) * 4 / 1e9
    for o, n in rounds:
        print(f"     fused-dma {o:.3f} ms | half-composed {n:.3f} ms ({gb / n * 1e3:.0f} GB/s algorithmic) | ratio {n / o:.3f}", flush=True)


if __name__ == "__main__":
    case(64, 32, 512)
    case(128, 64, 256)
