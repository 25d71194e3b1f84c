"""Where the native backward's time goes, layer shape by layer shape (PTI, batch 1): the library GEMMs vs the kernels around them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e4s2024_amd import ops, seeded

dev = "cuda:0"


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


# (cin, cout, h(out of plain / in of up), up)
for cin, cout, h, up in [(128, 128, 256, 1), (256, 128, 128, 2), (256, 256, 128, 1), (512, 256, 64, 2), (512, 512, 64, 1), (512, 512, 32, 2), (512, 512, 32, 1), (32, 32, 1024, 1), (64, 64, 512, 1)]:
    G, P = up * up, h * h
    x = torch.randn(1, cin, h, h, device=dev)
    s = torch.randn(1, 12, cin, device=dev)
    d = torch.rand(1, 12, cout, device=dev) + 0.5
    lab = torch.from_numpy(seeded.blocky_labels(3, 1, 12, up * h, max(1, up * h // 32))).to(dev).to(torch.uint8)
    wg = torch.randn(G, cout, cin, 3, 3, device=dev)
    gy = torch.randn(1, cout, up * h, up * h, device=dev)
    out = torch.randn(1, cout, up * h, up * h, device=dev)
    gz = ops._mconv_scale(gy, out, d, lab, 12, up, want_q=True, act=True, want_sums=True)[0]
    cols = ops._mconv_unfold(x, s, lab, 3, up)
    wt = wg.reshape(G, 1, cout, cin * 9).transpose(2, 3)
    u = torch.matmul(wt, gz)
    res = {
        "scale": t(lambda: ops._mconv_scale(gy, out, d, lab, 12, up, want_q=True, act=True, want_sums=True)),
        "unfold": t(lambda: ops._mconv_unfold(x, s, lab, 3, up)),
        "U gemm (lib)": t(lambda: torch.matmul(wt, gz)),
        "U gemm_sb": t(lambda: ops.gemm_sb(wg.reshape(G, cout, cin * 9), gz.view(G, cout, -1), False, False)),
        "U+fold": t(lambda: (setattr(ops, "DGRAD_FUSED", False), ops._mconv_input_grads(gz, wg, x, s, lab, up, True, True, False))),
        "dgrad fused": t(lambda: (setattr(ops, "DGRAD_FUSED", True), ops._mconv_input_grads(gz, wg, x, s, lab, up, True, True, False))),
        "dW gemm (lib)": t(lambda: torch.matmul(gz, cols.transpose(2, 3))),
        "dW gemm_sb": t(lambda: ops._gemm_nt(gz, cols)),
        "dW implicit": t(lambda: ops.mconv_wgrad(gz, x, s, lab, cout, 3, up)),
    }
    for kc in (2048, 8192):
        if P % kc == 0 and P > kc:
            nb = P // kc
            a = gz.view(G, cout, nb, kc).permute(0, 2, 1, 3).reshape(G * nb, cout, kc) if False else gz.view(G, cout, nb, kc).permute(0, 2, 1, 3)
            bm = cols.view(G, cin * 9, nb, kc).permute(0, 2, 3, 1)
            res[f"dW bmm kc={kc}"] = t(lambda: torch.matmul(a, bm).sum(1))
    gf = 2 * cin * cout * 9 * P * G / 1e9
    print(f"cin {cin} cout {cout} h {h} up {up}: {gf:.1f} GFLOP per GEMM | " + " | ".join(f"{k} {v:.3f} ms" for k, v in res.items()), flush=True)
    del x, cols, u, gz
    torch.cuda.empty_cache()
