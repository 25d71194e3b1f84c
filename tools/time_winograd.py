"""Direct split-bf16 3x3 convolution against the Winograd F(2x2, 3x3) route on the encoder's shapes (batch 16 = the two faces of 8 swaps)."""
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


with torch.no_grad():
    for bs, cin, cout, h in [(16, 512, 512, 32), (16, 256, 256, 64), (16, 256, 512, 64), (16, 128, 128, 128), (16, 512, 512, 16), (8, 512, 512, 32), (4, 512, 512, 32)]:
        x = torch.randn(bs, cin, h, h, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        st = ops.plane_stats(x, 1e-5)
        slope = torch.rand(cout, device=dev)
        pc, pw = ops.PreparedConv().get(w), ops.PreparedWinograd().get(w)
        d = t(lambda: ops.conv2d(x, pc, 1, 1, in_norm=st, prelu=slope))
        wg = t(lambda: ops.conv2d_winograd(x, pw, in_norm=st, prelu=slope))
        T = bs * h * h // 4
        V = torch.empty(16, cin, T, device=dev)
        g = t(lambda: ops.gemm_sb(pw, V, True, False))
        ps = ops.PreparedWinogradSplit().get(w)
        wp = t(lambda: ops.conv2d_winograd_pre(x, ps, in_norm=st, prelu=slope))
        vh = torch.empty(16, cin // 8, T, 8, dtype=torch.int16, device=dev); vl = torch.empty_like(vh); M = torch.empty(16, cout, T, device=dev)
        from e4s2024_amd._lib import lib
        from e4s2024_amd.ops import _p, _stream
        gp = t(lambda: lib().call("e4s_gemm_pre", _p(M), _p(ps[0]), _p(ps[1]), _p(vh), _p(vl), cout, T, cin, cout * cin, cin * T, cout * T, 16, _stream()))
        ti = t(lambda: lib().call("e4s_wino_input_pre", _p(vh), _p(vl), _p(x), _p(st[0]), _p(st[1]), bs, cin, h, h, _stream()))
        y = torch.empty(bs, cout, h, h, device=dev)
        to = t(lambda: lib().call("e4s_wino_output", _p(y), _p(M), _p(slope), bs, cout, h, h, _stream()))
        err = (ops.conv2d_winograd_pre(x, ps, in_norm=st, prelu=slope) - ops.conv2d(x, pc, 1, 1, in_norm=st, prelu=slope)).abs().max().item()
        print(f"bs {bs} {cin}->{cout} @{h}: direct {d:.3f} ms   winograd {wg:.3f} ms (its 16 GEMMs alone {g:.3f})   pre-split {wp:.3f} ms "
              f"(input {ti:.3f} + GEMMs {gp:.3f} + output {to:.3f}; max |diff| vs direct {err:.1e})", flush=True)
