"""A CHAIN of pre-split Winograd convolutions (each consuming the previous output) on a side stream, another chain on the main stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from e4s2024_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
bs, c, h = 8, 512, 32
xs = [torch.randn(bs, c, h, h, device=dev) for _ in range(2)]
ws = [torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5) for _ in range(6)]
slope = torch.rand(c, device=dev)
caches = [(ops.PreparedConv(), ops.PreparedWinograd(), ops.PreparedWinogradSplit()) for _ in ws]
mode = sys.argv[1] if len(sys.argv) > 1 else "pre"


def chain(x):
    for w, ca in zip(ws, caches):
        st = ops.plane_stats(x, 1e-5)
        if mode in ("preA", "preB"):
            from e4s2024_amd._lib import lib
            from e4s2024_amd.ops import _p, _stream
            b_, c_, h_, w_ = x.shape
            T = b_ * (h_ // 2) * (w_ // 2)
            uh, ul = ca[2].get(w)
            M = torch.empty((16, c_, T), device=dev)
            if mode == "preA":           # V by the fp32 transform + torch re-layout + split kernel, GEMM = e4s_gemm_pre
                V = torch.empty((16, c_, T), device=dev)
                lib().call("e4s_wino_input", _p(V), _p(x), _p(st[0]), _p(st[1]), b_, c_, h_, w_, _stream())
                Vb = V.view(16, c_ // 8, 8, T).permute(0, 1, 3, 2).contiguous()
                vh = torch.empty((16, c_ // 8, T, 8), dtype=torch.int16, device=dev); vl = torch.empty_like(vh)
                lib().call("e4s_split_bf16", _p(vh), _p(vl), _p(Vb), Vb.numel(), _stream())
                lib().call("e4s_gemm_pre", _p(M), _p(uh), _p(ul), _p(vh), _p(vl), c_, T, c_, c_ * c_, c_ * T, c_ * T, 16, _stream())
            else:                        # V by e4s_wino_input_pre, rebuilt as fp32, GEMM = e4s_gemm_sb
                vh = torch.empty((16, c_ // 8, T, 8), dtype=torch.int16, device=dev); vl = torch.empty_like(vh)
                lib().call("e4s_wino_input_pre", _p(vh), _p(vl), _p(x), _p(st[0]), _p(st[1]), b_, c_, h_, w_, _stream())
                Vf = (vh.view(torch.bfloat16).float() + vl.view(torch.bfloat16).float()).permute(0, 1, 3, 2).reshape(16, c_, T).contiguous()
                M = ops.gemm_sb(ca[1].get(w), Vf, True, False, split_k=False)
            out = torch.empty((b_, c_, h_, w_), device=dev)
            lib().call("e4s_wino_output", _p(out), _p(M), _p(slope), b_, c_, h_, w_, _stream())
            x = out
        elif mode == "pre":
            x = ops.conv2d_winograd_pre(x, ca[2].get(w), in_norm=st, prelu=slope)
        elif mode == "f32":
            x = ops.conv2d_winograd(x, ca[1].get(w), in_norm=st, prelu=slope)
        else:
            x = ops.conv2d(x, ca[0].get(w), 1, 1, in_norm=st, prelu=slope)
    return x


side = torch.cuda.Stream()
big = torch.randn(64, 1024, 1024, device=dev)
with torch.no_grad():
    ref = [chain(x).clone() for x in xs]
    torch.cuda.synchronize()
    nbad = 0
    for it in range(int(os.environ.get('ITERS', '8'))):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            o0 = chain(xs[0])
        nb = os.environ.get("NEIGHBOUR", "chain")
        if nb == "chain":
            o1 = chain(xs[1])
        elif nb == "busy":                       # GPU work on the main stream, no allocation
            for _ in range(60):
                big.mul_(1.0001)
            o1 = ref[1]
        elif nb == "alloc":                      # allocator traffic on the main stream, (almost) no GPU work
            junk = [torch.empty(16, 64, 2048, 8, dtype=torch.int16, device=dev) for _ in range(40)]
            del junk
            o1 = ref[1]
        elif nb == "directchain":
            mode_saved = mode
            globals()["mode"] = "direct"; o1 = chain(xs[1]); globals()["mode"] = mode_saved
            o1 = ref[1]
        main.wait_stream(side)
        torch.cuda.synchronize()
        d0, d1 = (o0 - ref[0]).abs().max().item(), (o1 - ref[1]).abs().max().item()
        nbad += d0 != 0 or d1 != 0
        if d0 or d1:
            print(mode, it, "side", d0, "main", d1, flush=True)
    print(mode, "bad iterations:", nbad, flush=True)
