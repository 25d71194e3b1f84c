"""Time e4s_mconv_dgrad (the fused data / style gradient) against e4s_gemm_sb + e4s_mconv_fold on the PTI layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e4s2024_amd import ops, seeded

dev = "cuda:0"


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for cin, cout, h, up in [(128, 128, 256, 1), (256, 128, 128, 2), (256, 256, 128, 1), (512, 512, 64, 1)]:
    G = up * up
    x = torch.randn(1, cin, h, h, device=dev)
    s = torch.randn(1, 12, cin, device=dev)
    lab = torch.from_numpy(seeded.blocky_labels(3, 1, 12, up * h, max(1, up * h // 32))).to(dev).to(torch.uint8)
    wg = torch.randn(G, cout, cin, 3, 3, device=dev)
    gz = torch.randn(G, 1, cout, h * h, device=dev)
    ops.DGRAD_FUSED = True
    f = t(lambda: ops._mconv_input_grads(gz, wg, x, s, lab, up, True, True, False))
    from e4s2024_amd._lib import lib
    from e4s2024_amd.ops import _p, _stream
    dx = torch.empty_like(x); nt = lib().cdll.e4s_mconv_dgrad_tiles(h, h); part = torch.empty(nt, 1, 12, cin, device=dev)
    k = t(lambda: lib().call("e4s_mconv_dgrad", _p(dx), _p(part), _p(gz), _p(wg), _p(x), _p(s), _p(lab), 1, cin, cout, h, h, 12, up, _stream()))
    ops.DGRAD_FUSED = False
    u = t(lambda: ops._mconv_input_grads(gz, wg, x, s, lab, up, True, True, False))
    print(f"cin {cin} cout {cout} h {h} up {up}: fused {f:.3f} ms (kernel alone {k:.3f})   gemm + fold {u:.3f} ms", flush=True)
