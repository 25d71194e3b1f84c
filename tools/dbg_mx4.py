import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from e4s2024_amd import ops
import test_gpu_mx4 as M
DEV = "cuda:0"
bs, cin, cout, h, w, nreg, lh, lw = M.SHAPES[0]
kind = sys.argv[1] if len(sys.argv) > 1 else "cells8"
rs = np.random.RandomState(1)
lab = M._labels(kind, rs, bs, nreg, lh, lw, 2 * h, 2 * w)
g = torch.Generator(device=DEV).manual_seed(1)
x = torch.randn(bs, cin, h, w, device=DEV, generator=g); wgt = torch.randn(1, cout, cin, 3, 3, device=DEV, generator=g)
s = 1.0 + 0.3 * torch.randn(bs, nreg, cin, device=DEV, generator=g); d = torch.rand(bs, nreg, cout, device=DEV, generator=g) + 0.5
nz = torch.randn(bs, 1, 2 * h, 2 * w, device=DEV, generator=g); nw, ab = torch.tensor([0.17], device=DEV), torch.zeros(cout, device=DEV)
blur = torch.tensor([1., 3., 3., 1.], device=DEV); blur = blur[:, None] * blur[None, :]; blur = blur / blur.sum() * 4
labels = torch.from_numpy(lab).to(DEV)
wt, _ = ops.PreparedWeights().get(wgt, blur, True, True)
wmx = ops.PreparedMx().get(wgt, blur, True, 1); wmx4 = ops.PreparedMx().get(wgt, blur, True, 4)
ref = ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, True, mx=(wmx, 1))
# each kernel alone into a NaN-filled tensor
from e4s2024_amd._lib import lib
import ctypes
_p = ops._p
def run(q, c):
    out = torch.full((bs, cout, 2 * h, 2 * w), float("nan"), device=DEV)
    if q:
        lib().call("e4s_region_upconv_mx4", _p(out), _p(x), _p(wmx4), _p(ops.mx_flags(x.device)), _p(s), _p(d), _p(labels), lh, lw, _p(nz), bs, _p(nw), _p(ab), 1, bs, cin, cout, h, w, nreg, ops._stream())
    if c:
        lib().call("e4s_region_modconv3x3_mx", _p(out), _p(x), _p(wmx), 1, _p(ops.mx_flags(x.device)), _p(s), _p(d), _p(labels), lh, lw, _p(nz), bs, _p(nw), _p(ab), 1, bs, cin, cout, h, w, nreg,
                   1 | 32, None, 0, *([None] * 6), None, None, None, ops._stream())
    torch.cuda.synchronize()
    return out
for name, (q, c) in {"mx4 only": (1, 0), "composed(skip) only": (0, 1), "both": (1, 1)}.items():
    o = run(q, c)
    nan = torch.isnan(o)
    print(name, "NaN share", float(nan.float().mean()), "rows with NaN (b0, c0):", torch.nonzero(nan[0, 0].any(1)).flatten().tolist()[:40], "cols:", torch.nonzero(nan[0, 0].any(0)).flatten().tolist()[:8],
          " max|diff| where written:", float(torch.nan_to_num(o - ref, nan=0.0).abs().max()))
    per_c = nan.float().mean((0, 2, 3))
    print("   NaN share per channel block of 32:", [round(float(per_c[i:i + 32].mean()), 3) for i in range(0, cout, 32)], " per batch:", nan.float().mean((1, 2, 3)).tolist())
o = run(1, 0)
dif = torch.nan_to_num(o - ref, nan=0.0, posinf=1e9, neginf=-1e9).abs()
for cb in range(0, cout, 32):
    blk = dif[:, cb:cb + 32]
    print("co", cb, "max diff", float(blk.max()), "per parity", [[float(blk[:, :, pa::2, pb::2].max()) for pb in (0, 1)] for pa in (0, 1)],
          "nan per parity", [[float(torch.isnan(o[:, cb:cb + 32, pa::2, pb::2]).float().mean()) for pb in (0, 1)] for pa in (0, 1)])
# which co inside block 1 are bad
bad = torch.isnan(o[:, 32:64]).float().mean((0, 2, 3))
print("nan share per co 32..63:", [round(float(v), 2) for v in bad])
# ---- the two prepared weight copies must hold the same values
a4 = wmx4.cpu().numpy(); a1 = wmx.cpu().numpy()
nchunk, ncot4, ncot1 = cin // 16, -(-cout // 64), -(-cout // 128)
ROWB1, PARB, ROWB4 = 25600, 13312, 53248
bad = 0
for par in range(4):
    for chunk in (0, nchunk - 1):
        for row in range(3):
            for co in (0, 31, 32, 47, 48, 63, 64, 100, cout - 1):
                t1, n1 = co // 128, co % 128
                t4, n4 = co // 64, co % 64
                s1 = (((par * nchunk + chunk) * ncot1 + t1) * 3 + row) * ROWB1
                s4 = ((chunk * ncot4 + t4) * 3 + row) * ROWB4 + par * PARB
                for tap in range(3):
                    for half in range(2):
                        w1a = a1[s1 + ((tap * 2 + half) * 128 + n1) * 16: s1 + ((tap * 2 + half) * 128 + n1) * 16 + 16]
                        w1b = a4[s4 + ((tap * 2 + half) * 64 + n4) * 16: s4 + ((tap * 2 + half) * 64 + n4) * 16 + 16]
                        bad += int((w1a != w1b).any())
                for term in range(2):
                    for half in range(2):
                        la = a1[s1 + 12288 + ((term * 2 + half) * 128 + n1) * 16:][:16]; lb = a4[s4 + 6144 + ((term * 2 + half) * 64 + n4) * 16:][:16]
                        ha = a1[s1 + 12288 + 8192 + ((term * 2 + half) * 128 + n1) * 8:][:8]; hb = a4[s4 + 6144 + 4096 + ((term * 2 + half) * 64 + n4) * 8:][:8]
                        bad += int((la != lb).any()) + int((ha != hb).any())
                for half in range(2):
                    sa = a1[s1 + 12288 + 8192 + 4096 + (half * 128 + n1) * 4:][:4]; sb_ = a4[s4 + 6144 + 4096 + 2048 + (half * 64 + n4) * 4:][:4]
                    bad += int((sa != sb_).any())
print("weight copies: mismatching fields", bad)
