"""Which stage of the pre-split Winograd chain goes wrong first under two-stream concurrency?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from e4s2024_amd import ops
from e4s2024_amd._lib import lib
from e4s2024_amd.ops import _p, _stream
dev = "cuda:0"
torch.manual_seed(0)
bs, c, h = 8, 512, 32
T = bs * h * h // 4
xs = [torch.randn(bs, c, h, h, device=dev) for _ in range(2)]
ws = [torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5) for _ in range(6)]
slope = torch.rand(c, device=dev)
caches = [ops.PreparedWinogradSplit() for _ in ws]


def chain(x, keep):
    for w, ca in zip(ws, caches):
        st = ops.plane_stats(x, 1e-5)
        uh, ul = ca.get(w)
        vh = torch.empty((16, c // 8, T, 8), dtype=torch.int16, device=dev); vl = torch.empty_like(vh)
        lib().call("e4s_wino_input_pre", _p(vh), _p(vl), _p(x), _p(st[0]), _p(st[1]), bs, c, h, h, _stream())
        M = torch.empty((16, c, T), device=dev)
        lib().call("e4s_gemm_pre", _p(M), _p(uh), _p(ul), _p(vh), _p(vl), c, T, c, c * c, c * T, c * T, 16, _stream())
        out = torch.empty((bs, c, h, h), device=dev)
        lib().call("e4s_wino_output", _p(out), _p(M), _p(slope), bs, c, h, h, _stream())
        keep.append({"mean": st[0], "rstd": st[1], "vh": vh, "vl": vl, "M": M, "out": out})      # kept alive: no reuse inside a chain
        x = out
    return x


side = torch.cuda.Stream()
with torch.no_grad():
    refs = [[], []]
    chain(xs[0], refs[0]); chain(xs[1], refs[1])
    torch.cuda.synchronize()
    found = 0
    for it in range(int(os.environ.get("ITERS", "60"))):
        k0, k1 = [], []
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            chain(xs[0], k0)
        chain(xs[1], k1)
        main.wait_stream(side)
        torch.cuda.synchronize()
        for row, kk in ((0, k0), (1, k1)):
            for li, (a, r) in enumerate(zip(kk, refs[row])):
                bad = [n for n in ("mean", "rstd", "vh", "vl", "M", "out") if not torch.equal(a[n], r[n])]
                if bad:
                    n = bad[0]
                    diff = (a[n] != r[n])
                    idx = diff.nonzero()
                    print(f"iter {it} chain {row} layer {li}: first wrong = {n} ({int(diff.sum())} elements; first at {idx[0].tolist()}, last at {idx[-1].tolist()}); also {bad[1:]}", flush=True)
                    found += 1
                    break
        if found >= 6:
            break
    print("failures:", found)
