import os, sys, torch
sys.path.insert(0, os.getcwd())
from e4s2024_amd import ops
dev="cuda:0"
g=torch.Generator(device=dev).manual_seed(1)
cin,cout,h=128,128,32
x=torch.randn(1,cin,h,h,device=dev,generator=g); w=torch.randn(cout,cin,3,3,device=dev,generator=g)/(cin*9)**0.5
pc=ops.PreparedConv().get(w); w3=ops.PreparedMx().get(w,None,False,3)
ref=ops.conv2d(x,pc,1,1)
y=ops.conv3x3_mx(x,w3,3,cout)
torch.cuda.synchronize()
err=(y-ref).abs().amax(dim=(0,1))   # [h,w]
print("max err", err.max().item())
bad=(err>1e-3)
print("bad rows:", bad.any(dim=1).nonzero().flatten().tolist())
print("bad cols:", bad.any(dim=0).nonzero().flatten().tolist()[:40])
errc=(y-ref).abs().amax(dim=(0,2,3)); print("bad channels:", (errc>1e-3).nonzero().flatten().tolist()[:40])
