import torch
dev = "cuda:0"
for kw in (dict(), dict(capturable=True), dict(fused=True), dict(fused=True, capturable=True)):
    p = torch.randn(10, device=dev, requires_grad=True)
    opt = torch.optim.Adam([p], lr=1e-3, **kw)
    v0 = p._version
    p.grad = torch.randn(10, device=dev)
    opt.step()
    print(kw, "version", v0, "->", p._version)
