"""One launch of the plain-convolution mx kernel (512 -> 512 @32^2, batch 16): with the -DMX_ABL=32 tuning library (tools/build_abl.sh 32, E4S_HIP_LIB=...) the
kernel prints the cycle counts of its phases for workgroup 0."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from e4s2024_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(16, 512, 32, 32, device=dev, generator=g)
w = torch.randn(512, 512, 3, 3, device=dev, generator=g) * 0.02
wmx = ops.PreparedMx().get(w, None, False, 1)
for _ in range(2):
    y = ops.conv3x3_mx(x, wmx, 1, 512)
    torch.cuda.synchronize()
