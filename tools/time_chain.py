"""Split-plane chain kernels (csrc/modconv_chain.hip) against the kernels they replace: same bits?  how much faster?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e4s2024_amd import ops

dev = "cuda:0"


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def to_blocked(t_):
    b, c, h, w = t_.shape
    return t_.view(b, c // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous()


def conv_case(c, res, want_out, bs=4):
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
    # fused ToRGB operands
    rw = torch.randn(1, 3, c, 1, 1, device=dev)
    with torch.no_grad():
        r_wt, _ = ops.PreparedWeights().get(rw, None, False, False)
    r_s = torch.randn(bs, 1, c, device=dev)
    r_bias = torch.randn(1, 3, 1, 1, device=dev)
    skip = torch.randn(bs, 3, res // 2, res // 2, device=dev)
    k1 = torch.tensor([1., 3., 3., 1.], device=dev)
    upk = k1[:, None] * k1[None, :] / k1.sum() ** 2 * 4
    rgb = (r_wt, r_s, r_bias, skip, upk)
    s_next = torch.randn(bs, 1, c, device=dev)
    xn = to_blocked(x)
    old = lambda: ops.region_modconv3x3(xn, wt, s, d, None, noise, nw, ab, True, c, False, rgb=rgb, want_out=want_out, x_nhwc=True, out_nhwc=want_out)
    xsp = ops.to_split_planes(x, s)
    new = lambda: ops.chain_conv3x3(xsp, wt, d, noise, nw, ab, True, c, s_next=s_next if want_out else None, rgb=rgb)
    o_old, rgb_old = old()
    o_new, rgb_new = new()
    torch.cuda.synchronize()
    print(f"{c}->{c} @ {res}^2 bs {bs} want_out={want_out}: rgb max|diff| {(rgb_old - rgb_new).abs().max().item():.3e} (|rgb|max {rgb_old.abs().max().item():.2f}) equal={torch.equal(rgb_old, rgb_new)}", flush=True)
    if want_out:
        ref_sp = ops.to_split_planes(o_old, s_next, x_nhwc=True)
        print(f"     out planes equal={torch.equal(ref_sp, o_new)}  max|diff| of hi+lo {(ops.from_split_planes(ref_sp) - ops.from_split_planes(o_new)).abs().max().item():.3e}", flush=True)
    for _ in range(3 if not os.environ.get("E4S_CHAIN_EXP") else 0):   # determinism / races
        o2, r2 = new()
        assert torch.equal(r2, rgb_new) and (o2 is None or torch.equal(o2, o_new)), "run-to-run difference"
    print(f"     old (channel-blocked in{'/out' if want_out else ''}, fused rgb) {t(old):.3f} ms | chain {t(new):.3f} ms | to_split_planes {t(lambda: ops.to_split_planes(x, s)):.3f} ms", flush=True)


def up_case(cin, cout, res, bs=4):
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
    xn = to_blocked(x)
    old = lambda: ops.modconv_up_single(xn, wt, s, d, blur, noise, nw, ab, True, cout, x_nhwc=True, out_nhwc=True)
    xsp = ops.to_split_planes(x, s)
    hc = ops.PreparedHc().get(w, blur)
    new = lambda: ops.modconv_up_single(xsp, wt, s, d, blur, noise, nw, ab, True, cout, s_next=s_next, hc=hc)
    o_old, o_new = old(), new()
    ref_sp = ops.to_split_planes(o_old, s_next, x_nhwc=True)
    a_, b_ = ops.from_split_planes(ref_sp), ops.from_split_planes(o_new)
    print(f"up {cin}->{cout} @ {res}->{2 * res} bs {bs}: planes equal={torch.equal(ref_sp, o_new)} max|diff| of hi+lo {(a_ - b_).abs().max().item():.3e} (|v|max {a_.abs().max().item():.1f})", flush=True)
    for _ in range(3 if not os.environ.get("E4S_CHAIN_EXP") else 0):
        assert torch.equal(new(), o_new), "run-to-run difference"
    print(f"     old fused up (channel-blocked in/out) {t(old):.3f} ms | chain {t(new):.3f} ms", flush=True)


if __name__ == "__main__":
    up_case(64, 32, 512)
    up_case(128, 64, 256)
    if not os.environ.get("E4S_CHAIN_EXP"):
        up_case(64, 32, 37, bs=2)
        up_case(128, 64, 16, bs=1)
    conv_case(32, 1024, False)
    conv_case(64, 512, True)
    if not os.environ.get("E4S_CHAIN_EXP"):
        conv_case(32, 64, False, bs=1)
        conv_case(64, 64, True, bs=3)
