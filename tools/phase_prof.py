"""Per-workgroup phase timeline of the split-bf16 modulated-conv kernel (tuning aid).

Needs the instrumented build:  python -m e4s2024_amd.build --phase-prof ;  E4S_HIP_LIB=e4s2024_amd/lib/libe4s_hip_prof.so
Marks (100 MHz wall clock): 0 start, 1 first chunk staged, 2 K loop done, 3 epilogue tables staged, 4 stores issued, 5 stores drained.
"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s2024_amd import ops
from e4s2024_amd._lib import lib

dev = "cuda:0"
SLOTS = 8


def read(n_blocks, which="sb"):
    buf = np.zeros(n_blocks * SLOTS, dtype=np.int64)
    fn = getattr(lib().cdll, f"e4s_prof_read_{which}")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    rc = fn(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    assert rc == 0, rc
    return buf.reshape(n_blocks, SLOTS)


def clear(which="sb"):
    torch.cuda.synchronize()
    assert getattr(lib().cdll, f"e4s_prof_clear_{which}")() == 0


def report(name, t, kernel_ms):
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0          # 100 MHz -> us
    life = us[:, 5] - us[:, 0]
    ph = [us[:, i + 1] - us[:, i] for i in range(5)]
    span = us[:, 5].max()
    conc = life.sum() / span
    print(f"{name}: {len(t)} workgroups, kernel {kernel_ms*1e3:.0f} us (marks span {span:.0f} us), mean lifetime {life.mean():.1f} us, "
          f"mean concurrency {conc:.0f} workgroups ({conc/256:.2f}/CU)")
    hw = t[:, 7]
    cu_key = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xff)        # XCC id, (se, sh, cu) bits of HW_ID
    keys, counts = np.unique(cu_key, return_counts=True)
    # per-CU concurrency: sum of lifetimes on a CU / span
    per_cu = np.array([life[cu_key == k].sum() / span for k in keys])
    inst = []
    for k in keys[:32]:
        m = cu_key == k
        ev = sorted([(x, 1) for x in us[m, 0]] + [(x, -1) for x in us[m, 5]])
        cur = mx = 0
        for _, dlt in ev:
            cur += dlt; mx = max(mx, cur)
        inst.append(mx)
    print(f"    instantaneous max resident per CU (first 32 CUs): {max(inst)}")
    print(f"    distinct CU keys {len(keys)}, workgroups per CU min/mean/max {counts.min()}/{counts.mean():.1f}/{counts.max()}, "
          f"resident per CU mean {per_cu.mean():.2f} max {per_cu.max():.2f}")
    for lbl, p in zip(("first chunk load+stage", "K loop (rest)", "epilogue table stage", "epilogue compute+store issue", "store drain"), ph):
        print(f"    {lbl:30s} mean {p.mean():7.2f} us   p50 {np.median(p):7.2f}   p90 {np.percentile(p, 90):7.2f}")


def run_same(cin, cout, h, bs=4, masked=False, up=False):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(1, cout, cin, 3, 3, device=dev, generator=g)
    nreg = 12 if masked else 1
    s = torch.randn(bs, nreg, cin, device=dev, generator=g); d = torch.rand(bs, nreg, cout, device=dev, generator=g)
    labels = None
    if masked:
        lab = torch.randint(0, 12, (bs, 16, 16), device=dev, generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2)
        labels = lab.to(torch.uint8).contiguous()
    nz = torch.randn(bs, 1, h, h, device=dev, generator=g); nw = torch.tensor([0.1], device=dev); ab = torch.zeros(cout, device=dev)
    pw = ops.PreparedWeights()
    blur = torch.tensor([1., 3., 3., 1.], device=dev); blur = (blur[:, None] * blur[None, :]); blur = blur / blur.sum() * 4
    wt, _ = pw.get(w, blur if up else None, up, True)
    if up:
        nz = torch.randn(bs, 1, 2 * h, 2 * h, device=dev, generator=g)
    for _ in range(3):
        ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, up)
    clear("sb")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, up); b.record(); torch.cuda.synchronize()
    n = 1 << 17
    report(f"{'up' if up else 'same'} conv {cin}->{cout} @{h}^2 bs{bs} {'masked' if masked else 'single-region'}", read(n), a.elapsed_time(b))


def run_up_fused(cin, cout, h, bs=4, sp=False):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(bs, cin, h, h, device=dev, generator=g)
    w = torch.randn(1, cout, cin, 3, 3, device=dev, generator=g)
    blur = torch.tensor([1., 3., 3., 1.], device=dev); blur = (blur[:, None] * blur[None, :]); blur = blur / blur.sum() * 4
    s = torch.randn(bs, 1, cin, device=dev, generator=g); d = torch.rand(bs, 1, cout, device=dev, generator=g)
    nz = torch.randn(bs, 1, 2 * h, 2 * h, device=dev, generator=g); nw = torch.tensor([0.1], device=dev); ab = torch.zeros(cout, device=dev)
    wt, _ = ops.PreparedWeights().get(w, None, False, True, tconv=True)
    ops.UP_FUSED = True
    if sp:
        xs, sn = ops.to_split_planes(x, s), torch.randn(bs, 1, cout, device=dev, generator=g)
        call = lambda: ops.modconv_up_single(xs, wt, s, d, blur, nz, nw, ab, True, cout, s_next=sn)
    else:
        call = lambda: ops.modconv_up_single(x, wt, s, d, blur, nz, nw, ab, True, cout)
    for _ in range(3):
        call()
    clear("up")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); call(); b.record(); torch.cuda.synchronize()
    report(f"fused up {cin}->{cout} @{h}^2 bs{bs}{' split planes' if sp else ''}", read(1 << 17, "up"), a.elapsed_time(b))


def run_conv(cin, cout, h, bs=8):
    x = torch.randn(bs, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    mean = torch.zeros(bs, cin, device=dev); rstd = torch.ones(bs, cin, device=dev); slope = torch.full((cout,), 0.25, device=dev)
    pc = ops.PreparedConv().get(w)
    for _ in range(3):
        ops.conv2d(x, pc, stride=1, pad=1, in_norm=(mean, rstd), prelu=slope)
    clear("conv")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.conv2d(x, pc, stride=1, pad=1, in_norm=(mean, rstd), prelu=slope); b.record(); torch.cuda.synchronize()
    report(f"encoder conv {cin}->{cout} @{h}^2 bs{bs}", read(1 << 17, "conv"), a.elapsed_time(b))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        for h in (4, 8, 16, 32):
            run_same(512, 512, h, masked=True)
            run_same(512, 512, h, masked=True, up=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "conv":
        run_conv(256, 256, 64); run_conv(128, 128, 128); run_conv(64, 64, 256); run_conv(512, 512, 32)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "up":
        run_up_fused(128, 64, 256, sp=True); run_up_fused(64, 32, 512, sp=True)
        sys.exit(0)
    run_same(512, 512, 64, masked=True)
    run_same(256, 256, 128, masked=True)
    run_same(128, 128, 256, masked=True)
    run_same(512, 256, 64, masked=True, up=True)
    run_same(256, 128, 128, masked=True, up=True)
    run_up_fused(128, 64, 256)
    run_up_fused(64, 32, 512)
    run_up_fused(128, 64, 256, sp=True)
    run_up_fused(64, 32, 512, sp=True)
