import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from e4s2024_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
cin, cout, res, bs = 16, 32, 20, 1
x = torch.randn(bs, cin, res, res, device=dev)
w = torch.randn(1, cout, cin, 3, 3, device=dev)
s = torch.ones(bs, 1, cin, device=dev); d = torch.ones(bs, 1, cout, device=dev); sn = torch.ones(bs, 1, cout, device=dev)
k1 = torch.tensor([1., 3., 3., 1.], device=dev)
blur = k1[:, None] * k1[None, :] / k1.sum() ** 2 * 4
with torch.no_grad():
    wt, _ = ops.PreparedWeights().get(w, None, False, True, tconv=True)
    hc = ops.PreparedHc().get(w, blur)
    xsp = ops.to_split_planes(x, s)
    out = ops.from_split_planes(ops.modconv_up_single(xsp, wt, s, d, blur, None, None, None, False, cout, s_next=sn, hc=hc))
    old = ops.from_split_planes(ops.modconv_up_single(xsp, wt, s, d, blur, None, None, None, False, cout, s_next=sn))
wd = w.double()[0] / (cin * 9) ** 0.5
z = F.conv_transpose2d(x.double(), wd.transpose(0, 1), stride=2)
ref = F.conv2d(F.pad(z, (1, 1, 1, 1)), torch.flip(blur.double(), [0, 1])[None, None].expand(cout, 1, 4, 4), groups=cout)
print("old vs ref", (old.double() - ref).abs().max().item(), "hc vs ref", (out.double() - ref).abs().max().item(), "ref max", ref.abs().max().item())
e = (out.double() - ref).abs()[0]
print("err by channel%8:", [round(e[c::8].max().item(), 3) for c in range(8)])
print("err by channel//8:", [round(e[8 * g:8 * g + 8].max().item(), 3) for g in range(cout // 8)])
print("err by row parity:", [round(e[:, p::2].max().item(), 3) for p in range(2)], "col parity:", [round(e[:, :, p::2].max().item(), 3) for p in range(2)])
print("err by col (first 40):", [round(e[:, :, c].max().item(), 2) for c in range(40)])
print("err by row (first 40):", [round(e[:, r].max().item(), 2) for r in range(40)])
# is it a permutation?  compare sorted values of one pixel's channels
print("pixel (4,4) out", out[0, :8, 4, 4].tolist()); print("pixel (4,4) ref", ref[0, :8, 4, 4].tolist())
print("pixel (4,5) out", out[0, :8, 4, 5].tolist()); print("pixel (4,5) ref", ref[0, :8, 4, 5].tolist())
print("pixel (5,4) out", out[0, :8, 5, 4].tolist()); print("pixel (5,4) ref", ref[0, :8, 5, 4].tolist())
