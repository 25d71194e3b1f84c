"""Only e4s_wino_input_pre, on two streams at once, different inputs per call: is its output what it is on one stream?"""
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
xs = [[torch.randn(bs, c, h, h, device=dev) for _ in range(6)] for _ in range(2)]
sts = [[ops.plane_stats(x, 1e-5) for x in row] for row in xs]
fresh = os.environ.get("FRESH", "1") == "1"


def run(row):
    outs = []
    for x, st in zip(xs[row], sts[row]):
        vh = torch.empty((16, c // 8, T, 8), dtype=torch.int16, device=dev); vl = torch.empty_like(vh)
        lib().call("e4s_wino_input_pre", _p(vh), _p(vl), _p(x), _p(st[0]), _p(st[1]), bs, c, h, h, _stream())
        outs.append((vh, vl) if not fresh else (vh.clone(), vl.clone()))
    return outs


side = torch.cuda.Stream()
ref = [run(0), run(1)]
torch.cuda.synchronize()
bad = 0
for it in range(int(os.environ.get("ITERS", "60"))):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        o0 = run(0)
    o1 = run(1)
    main.wait_stream(side)
    torch.cuda.synchronize()
    for row, o in ((0, o0), (1, o1)):
        for i, (a, b) in enumerate(o):
            if not (torch.equal(a, ref[row][i][0]) and torch.equal(b, ref[row][i][1])):
                bad += 1
                n = int((a != ref[row][i][0]).sum().item())
                print("iter", it, "row", row, "call", i, "differing hi elements:", n, flush=True)
print("bad outputs:", bad)
