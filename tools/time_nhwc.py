"""Channel-blocked ([bs, C/8, H, W, 8]) vs channels-first activations on the single-region layers (512x512 / 1024x1024): how much faster?"""


def to_blocked(t):
    b, c, h, w = t.shape
    return t.view(b, c // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous()


def from_blocked(t):
    b, cb, h, w, _ = t.shape
    return t.permute(0, 1, 4, 2, 3).reshape(b, cb * 8, h, w)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e4s2024_amd import ops

dev = "cuda:0"


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for c, res, masked in [(32, 1024, False), (64, 512, False)]:
    bs = 4
    torch.manual_seed(0)
    x = torch.randn(bs, c, res, res, device=dev)
    w = torch.randn(1, c, c, 3, 3, device=dev)
    nreg = 12 if masked else 1
    styles = torch.randn(bs, nreg, 512, device=dev)
    mw, mb = torch.randn(c, 512, device=dev), torch.ones(c, device=dev)
    pw = ops.PreparedWeights()
    with torch.no_grad():
        wt, wsq = pw.get(w, None, False, True)
        s, d = ops.style_demod(styles, mw, mb, wsq, c)
    lab = torch.randint(0, 12, (bs, 32, 32), device=dev, dtype=torch.uint8).repeat_interleave(16, 1).repeat_interleave(16, 2) if masked else None
    noise = torch.randn(1, 1, res, res, device=dev)
    nw, ab = torch.tensor([0.1], device=dev), torch.randn(c, device=dev)
    xn = to_blocked(x)
    run = lambda xi, a, b_: ops.region_modconv3x3(xi, wt, s, d, lab, noise, nw, ab, True, c, False, x_nhwc=a, out_nhwc=b_)
    ref = run(x, False, False)
    for a, b_ in [(True, False), (False, True), (True, True)]:
        o = run(xn if a else x, a, b_)
        o = from_blocked(o) if b_ else o
        print('   max |diff| vs channels-first', a, b_, (o - ref).abs().max().item())
    print(f"{c:3d} ch @ {res}^2 bs {bs} masked={masked}: NCHW {t(lambda: run(x, False, False)):.3f} ms | C8 in {t(lambda: run(xn, True, False)):.3f} | "
          f"C8 out {t(lambda: run(x, False, True)):.3f} | C8 in+out {t(lambda: run(xn, True, True)):.3f} ms", flush=True)

# the fused up-sampling layers (transposed conv + blur in one launch): which side of the channels-last layout costs / gains what
for cin, cout, res in [(64, 32, 512), (128, 64, 256)]:
    bs = 4
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
    xn = to_blocked(x)
    run = lambda xi, a, b_: ops.modconv_up_single(xi, wt, s, d, blur, noise, nw, ab, True, cout, x_nhwc=a, out_nhwc=b_)
    ref = run(x, False, False)
    res_t = {}
    for a, b_ in [(False, False), (True, False), (False, True), (True, True)]:
        o = run(xn if a else x, a, b_)
        o = from_blocked(o) if b_ else o
        res_t[(a, b_)] = (t(lambda: run(xn if a else x, a, b_)), (o - ref).abs().max().item())
    print(f"fused up {cin}->{cout} @ {res}->{2 * res} bs {bs}: " + " | ".join(f"in {'C8' if a else 'NCHW'} out {'C8' if b_ else 'NCHW'} {v[0]:.3f} ms (diff {v[1]:.1e})"
                                                                              for (a, b_), v in res_t.items()), flush=True)
