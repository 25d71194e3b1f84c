import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import e4s2024_amd
from e4s2024_amd import ops, seeded
e4s2024_amd.install()
from models.stylegan2 import model as sg2
DEV="cuda:0"
cfg = sys.argv[1] if len(sys.argv) > 1 else "s"
sync = len(sys.argv) > 2 and sys.argv[2] == "sync"
if cfg == "s":
    bs,cin,cout,w,nreg=2,128,128,64,12
    rs = np.random.RandomState(3)
    lab = np.repeat(np.repeat(rs.randint(0,nreg,(bs,8,8)).astype(np.uint8),8,1),8,2); lab[:, :9, :12] = 255
elif cfg == "a":
    bs,cin,cout,w,nreg=4,512,512,32,12
    lab = seeded.blocky_labels(3, bs, 12, 512, 4)
torch.manual_seed(0)
m = sg2.StyledConv(cin, cout, 3, 512, upsample=False, mask_op=True).to(DEV).eval()
x = torch.randn(bs, cin, w, w, device=DEV); st = torch.randn(bs, nreg, 512, device=DEV); nz = torch.randn(bs,1,w,w,device=DEV)
lab = torch.from_numpy(lab).to(DEV)
with torch.no_grad():
    ops.MXE=False; y0 = m(x, st, lab, noise=nz); torch.cuda.synchronize()
    ops.MXE=True
    ys = []
    for i in range(12):
        ys.append(m(x, st, lab, noise=nz))
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
for i, y in enumerate(ys):
    d = (y - ys[0]).abs()
    nbad = int((d > 0).sum())
    msg = ""
    if nbad:
        idx = (d > 0).nonzero()
        msg = f" first {idx[0].tolist()} last {idx[-1].tolist()} chans {sorted(set(idx[:,1].tolist()))[:8]} rows {sorted(set(idx[:,2].tolist()))[:12]}"
    print(f"cfg {cfg} sync {sync} run {i}: vs old {float((y-y0).abs().max()):.3e} vs run0 max {float(d.max()):.3e} n {nbad}{msg}", flush=True)
