import os, sys, torch
sys.path.insert(0, os.getcwd())
from e4s2024_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
bs, cin, depth, h, w = 1, 32, 32, 32, 32
x = torch.randn(bs, cin, h, w, device=dev, generator=g)
w1 = torch.randn(depth, cin, 3, 3, device=dev, generator=g) / (cin * 9) ** 0.5
w2 = torch.randn(depth, depth, 3, 3, device=dev, generator=g) / (depth * 9) ** 0.5
with torch.no_grad():
    w13, w23 = ops.PreparedMx().get(w1, None, False, 3), ops.PreparedMx().get(w2, None, False, 3)
    r = ops.conv3x3_mx(x, w13, 3, depth)
    torch.cuda.synchronize(); print("plain ok", flush=True)
    ro = ops.conv3x3_mx(x, w13, 3, depth, out_prep=True)
    torch.cuda.synchronize(); print("producer ok", flush=True)
    # decode the f16 part of the map and compare with r
    d = ro.data.view(torch.uint8).reshape(bs, depth // 32, 116 * h * w)
    a1 = d[:, :, : 64 * h * w].reshape(bs, depth // 32, 4, h * w, 16).contiguous().view(torch.float16).reshape(bs, depth // 32, 4, h * w, 8)
    a1 = a1.permute(0, 1, 2, 4, 3).reshape(bs, depth, h, w).float()
    print("f16 part vs plain: max abs diff", (a1 - r).abs().max().item(), "scale", r.abs().max().item(), flush=True)
    y = ops.conv3x3_mx(r, w23, 3, depth)
    torch.cuda.synchronize()
    y2 = ops.conv3x3_mx(ro, w23, 3, depth)
    torch.cuda.synchronize(); print("consumer ok", torch.equal(y, y2), (y - y2).abs().max().item(), flush=True)
