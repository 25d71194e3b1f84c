"""One launch of a plain-convolution mx kernel (512 -> 512 @32^2, batch 16): with a profiling build of the library (E4S_HIP_LIB=...: -DMX_ABL=32 for
modconv_mx.hip via tools/build_abl.sh, -DMX3_PROF for conv_mx3.hip) the kernel prints the cycle counts of its phases for workgroup 0."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from e4s2024_amd import ops
dev = "cuda:0"
arith = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(16, 512, 32, 32, device=dev, generator=g)
w = torch.randn(512, 512, 3, 3, device=dev, generator=g) * 0.02
wmx = ops.PreparedMx().get(w, None, False, arith)
for _ in range(2):
    y = ops.conv3x3_mx(x, wmx, arith, 512)
    torch.cuda.synchronize()
