import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from e4s2024_amd import ops
torch.manual_seed(0)
cin, cout, h, ks = 32, 64, 32, 3
x = torch.randn(1, cin, h, h); w = torch.randn(cout, cin, ks, ks) * (cin * ks * ks) ** -0.5
def run(x, w):
    return ops.conv2d(x.cuda(), ops.PreparedConv().get(w.cuda()), 1, ks // 2).cpu()
for name, xm, wm in (("x[16:]=0", lambda t: torch.cat([t[:, :16], torch.zeros_like(t[:, 16:])], 1), None),
                     ("x[:16]=0", lambda t: torch.cat([torch.zeros_like(t[:, :16]), t[:, 16:]], 1), None),
                     ("w[16:]=0", None, lambda t: torch.cat([t[:, :16], torch.zeros_like(t[:, 16:])], 1)),
                     ("w[:16]=0", None, lambda t: torch.cat([torch.zeros_like(t[:, :16]), t[:, 16:]], 1))):
    xx = xm(x) if xm else x; ww = wm(w) if wm else w
    ref = F.conv2d(xx, ww, padding=1); out = run(xx, ww)
    # alternative hypotheses
    ref_c0 = F.conv2d(x[:, :16], w[:, :16], padding=1)
    print(name, "diff", (out - ref).abs().max().item(), "| out max", out.abs().max().item(), "ref max", ref.abs().max().item())
