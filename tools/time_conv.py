"""Per-shape timing of the encoder's 3x3 convolutions (IR-SE-50 body of FSEncoder_PSP at 256x256 input, batch 8)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s2024_amd import ops
dev = "cuda:0"
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [  # (cin, cout, h_in, stride, count per encoder call)
    (64, 64, 256, 1, 1), (64, 64, 256, 2, 1), (64, 64, 128, 1, 4),
    (64, 128, 128, 1, 1), (128, 128, 128, 2, 1), (128, 128, 64, 1, 6),
    (128, 256, 64, 1, 1), (256, 256, 64, 2, 1), (256, 256, 32, 1, 26),
    (256, 512, 32, 1, 1), (512, 512, 32, 2, 1), (512, 512, 16, 1, 4)]
tot = 0.0
for cin, cout, h, s, n in shapes:
    x = torch.randn(bs, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    mean = torch.zeros(bs, cin, device=dev); rstd = torch.ones(bs, cin, device=dev); slope = torch.full((cout,), 0.25, device=dev)
    pc = ops.PreparedConv().get(w)
    for _ in range(3): y = ops.conv2d(x, pc, stride=s, pad=1, in_norm=(mean, rstd), prelu=slope)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): y = ops.conv2d(x, pc, stride=s, pad=1, in_norm=(mean, rstd), prelu=slope)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    ho = y.shape[-1]
    gf = 2 * cin * cout * 9 * ho * ho * bs / 1e9
    tot += ms * n
    print(f"{cin:4d}->{cout:4d} @{h:3d} s{s}: {ms*1e3:7.1f} us  {gf/ms:7.1f} TFLOP/s (alg)   x{n:2d} = {ms*n:6.3f} ms")
print(f"3x3 convs of one encoder call at batch {bs}: {tot:.3f} ms")
