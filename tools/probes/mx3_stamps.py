"""One launch each of the encoder's two-phase convolution kernel in its four forms, under the -DMX3_PROF build (tools/prof_mx3_stamps.sh): workgroup 0 prints per wave
the cycles of its read / MFMA phases, the barrier waits behind them and the store phases."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from e4s2024_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
cin = cout = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h = int(sys.argv[2]) if len(sys.argv) > 2 else 128
bs = 16
x = torch.randn(bs, cin, h, h, device=dev, generator=g)
w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / (cin * 9) ** 0.5
w3, w5 = ops.PreparedMx().get(w, None, False, 3), ops.PreparedMx().get(w, None, False, 5)
with torch.no_grad():
    for name, fn in (("stride 1, planes in / planes out", lambda: ops.conv3x3_mx(x, w3, 3, cout)),
                     ("stride 1, planes in / phased + blocked out", lambda: ops.conv3x3_mx(x, w3, 3, cout, out_phased=True, out_c4=True))):
        fn(); torch.cuda.synchronize()
        print("==", name, flush=True)
        fn(); torch.cuda.synchronize()
    r6 = ops.conv3x3_mx(x, w3, 3, cout, out_phased=True)
    r7 = ops.conv3x3_mx(x, w3, 3, cout, out_phased=True, out_c4=True)
    r5 = ops.conv3x3_mx(x, w3, 3, cout, out_c4=True)
    ro = ops.conv3x3_mx(x, w3, 3, cout, out_prep=True)
    rop = ops.conv3x3_mx(x, w3, 3, cout, out_phased=True, out_prep=True)
    torch.cuda.synchronize()
    which = sys.argv[3] if len(sys.argv) > 3 else "all"
    for name, fn in [t for t in (("stride 1, blocked in", lambda: ops.conv3x3_mx(r5, w3, 3, cout)),
                     ("stride 1, prepared in", lambda: ops.conv3x3_mx(ro, w3, 3, cout)),
                     ("stride 2, phase planes in", lambda: ops.conv3x3_s2_mx(r6, w5, cout)),
                     ("stride 2, phased + blocked in", lambda: ops.conv3x3_s2_mx(r7, w5, cout)),
                     ("stride 2, prepared in", lambda: ops.conv3x3_s2_mx(rop, w5, cout))) if which == "all" or which in t[0]]:
        print("==", name, flush=True)
        fn(); torch.cuda.synchronize()
