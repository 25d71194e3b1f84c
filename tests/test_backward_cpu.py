"""Row f1 (interim): the stock-PyTorch forms used for the backward pass (e4s2024_amd/torch_ref.py) against the oracle — values and
gradients, on CPU.  The oracle is pinned to the reference by tests/test_oracle_golden.py; here it also serves as the autograd truth."""
import numpy as np
import torch

from e4s2024_amd import torch_ref
from oracle import e4s_oracle as O

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


def _leaf(rs, *shape, scale=1.0):
    return T((scale * rs.standard_normal(shape)).astype(np.float32)).requires_grad_(True)


def _grads(out, wrt, seed=0):
    g = T(np.random.RandomState(seed).standard_normal(tuple(out.shape)).astype(np.float32))
    return torch.autograd.grad(out, wrt, g, allow_unused=True)


def test_fir_resample_matches_oracle_upfirdn2d():
    rs = np.random.RandomState(1)
    x = _leaf(rs, 2, 3, 7, 9)
    k = O.make_blur_kernel((1, 3, 3, 1), gain=4.0)
    for up, pad in ((1, (1, 1)), (2, (2, 1)), (1, (2, 1))):
        a = torch_ref.fir_resample(x, k, up=up, pad=pad)
        b = O.upfirdn2d(x, k, up=up, pad=pad)
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-6
        ga, gb = _grads(a, [x]), _grads(b, [x])
        assert (ga[0] - gb[0]).abs().max().item() <= 1e-5


def test_styled_conv_and_to_rgb_values_and_gradients():
    rs = np.random.RandomState(2)
    bs, cin, cout, h, nreg = 2, 6, 5, 8, 4
    labels = T(rs.randint(0, nreg + 1, (bs, 16, 16)).astype(np.uint8))          # class nreg = a pixel outside every region
    mask = torch.stack([(labels == c) for c in range(nreg)], 1).float()
    for upsample in (False, True):
        for masked in (True, False):
            sd = {"conv.weight": _leaf(rs, 1, cout, cin, 3, 3), "conv.modulation.weight": _leaf(rs, cin, 512), "conv.modulation.bias": _leaf(rs, cin, scale=0.1),
                  "noise.weight": _leaf(rs, 1, scale=0.3), "activate.bias": _leaf(rs, cout, scale=0.1)}
            if upsample:
                sd["conv.blur.kernel"] = O.make_blur_kernel((1, 3, 3, 1), gain=4.0)
            x = _leaf(rs, bs, cin, h, h)
            st = _leaf(rs, bs, nreg, 512) if masked else _leaf(rs, bs, 512)
            ho = 2 * h if upsample else h
            nz = T(rs.standard_normal((bs, 1, ho, ho)).astype(np.float32))
            ref = O.styled_conv(sd, "", x, st, mask if masked else None, nz, masked=masked, upsample=upsample)
            got = torch_ref.styled_conv(x, st if masked else st[:, None, :], sd["conv.weight"], sd["conv.modulation.weight"], sd["conv.modulation.bias"],
                                        sd["noise.weight"], sd["activate.bias"], labels=labels if masked else None, noise=nz, act=True, upsample=upsample,
                                        blur=sd.get("conv.blur.kernel"), demodulate=True, mod_scale=1 / np.sqrt(512), mod_lr=1.0)
            assert (ref - got).abs().max().item() <= 2e-5, (upsample, masked)
            wrt = [x, st] + [sd[k] for k in ("conv.weight", "conv.modulation.weight", "conv.modulation.bias", "noise.weight", "activate.bias")]
            for a, b in zip(_grads(ref, wrt), _grads(got, wrt)):
                assert (a - b).abs().max().item() <= 2e-4 * max(1.0, a.abs().max().item()), (upsample, masked)
    # ToRGB with a skip connection
    sd = {"conv.weight": _leaf(rs, 1, 3, cin, 1, 1), "conv.modulation.weight": _leaf(rs, cin, 512), "conv.modulation.bias": _leaf(rs, cin, scale=0.1),
          "bias": _leaf(rs, 1, 3, 1, 1, scale=0.1), "upsample.kernel": O.make_blur_kernel((1, 3, 3, 1), gain=4.0)}
    x, st, skip = _leaf(rs, bs, cin, h, h), _leaf(rs, bs, nreg, 512), _leaf(rs, bs, 3, h // 2, h // 2)
    ref = O.to_rgb(sd, "", x, st, mask, skip, masked=True)
    got = torch_ref.to_rgb(x, st, skip, sd["conv.weight"], sd["conv.modulation.weight"], sd["conv.modulation.bias"], sd["bias"], labels=labels,
                           up_kernel=sd["upsample.kernel"], mod_scale=1 / np.sqrt(512), mod_lr=1.0)
    assert (ref - got).abs().max().item() <= 2e-5
    wrt = [x, st, skip, sd["conv.weight"], sd["conv.modulation.weight"], sd["bias"]]
    for a, b in zip(_grads(ref, wrt), _grads(got, wrt)):
        assert (a - b).abs().max().item() <= 2e-4 * max(1.0, a.abs().max().item())


def test_local_mlps_and_equal_linear():
    rs = np.random.RandomState(3)
    x = _leaf(rs, 3, 2, 10)
    w0, b0 = [_leaf(rs, 7, 10) for _ in range(2)], [_leaf(rs, 7, scale=0.1) for _ in range(2)]
    w2, b2 = [_leaf(rs, 12, 7) for _ in range(2)], [_leaf(rs, 12, scale=0.1) for _ in range(2)]
    got = torch_ref.local_mlps(x, w0, b0, w2, b2, 1 / np.sqrt(10), 1 / np.sqrt(7), 1.0, 1.0, 0.01, None)
    for g in range(2):
        h = torch.nn.functional.leaky_relu(O.equal_linear(x[:, g], w0[g], b0[g]), 0.01)
        assert (got[:, g] - O.equal_linear(h, w2[g], b2[g])).abs().max().item() <= 1e-5
    y = torch_ref.equal_linear(x[:, 0], w0[0], b0[0], 0.01 / np.sqrt(10), 0.01, True)
    assert (y - O.equal_linear(x[:, 0], w0[0], b0[0], lr_mul=0.01, activation=True)).abs().max().item() <= 1e-5


def test_hand_written_table_gradient_parity_weights_and_batched_mlps_equal_the_op_by_op_forms():
    """The launch-count optimisations of the PTI backward are algebra, not approximations: in fp64 they equal the plain forms —
    ``_StyleTables`` vs autograd through ``_tables_autograd``; the one-contraction parity weights vs slicing + flipping the 6x6 composed
    kernel; the two batched GEMMs of the per-region MLPs vs the per-region loop."""
    import torch.nn.functional as F
    from e4s2024_amd import torch_ref as R
    torch.manual_seed(0)
    f64 = dict(dtype=torch.float64)
    for demod in (True, False):
        x = torch.randn(2, 6, 4, 4, **f64)
        leaves = [torch.randn(2, 3, 8, **f64).requires_grad_(True), torch.randn(1, 5, 6, 3, 3, **f64).requires_grad_(True),
                  torch.randn(6, 8, **f64).requires_grad_(True), torch.randn(6, **f64).requires_grad_(True)]

        def run(fn):
            st, w, mw, mb = leaves
            s, ws, d = fn(x, st, w, mw, mb, 0.3, 0.7, demod)
            loss = (s * torch.arange(s.numel(), **f64).view_as(s)).sum() + (ws ** 2).sum() + ((d ** 3).sum() if d is not None else 0)
            return [s, ws, d], torch.autograd.grad(loss, leaves)
        (o1, g1), (o2, g2) = run(R._tables), run(R._tables_autograd)
        assert all((a is None and b is None) or torch.allclose(a, b, rtol=1e-12) for a, b in zip(o1, o2))
        assert all(torch.allclose(a, b, rtol=1e-10, atol=1e-12) for a, b in zip(g1, g2))
    blur = torch.tensor([1., 3., 3., 1.], **f64)
    blur = blur[:, None] * blur[None, :] / blur.sum() ** 2 * 4
    ws = torch.randn(5, 6, 3, 3, **f64)
    c2 = R._composed_up_weights(ws, blur, torch.float64)
    wg = R._parity_weights(ws, blur, torch.float64)
    for a in (0, 1):
        for b in (0, 1):
            assert torch.allclose(wg[2 * a + b], c2[:, :, a::2, b::2].flip(2, 3), rtol=1e-12)
    n, bs, dim, hid, out = 4, 3, 7, 9, 5
    w0, b0 = [torch.randn(hid, dim, **f64) for _ in range(n)], [torch.randn(hid, **f64) for _ in range(n)]
    w2, b2 = [torch.randn(out, hid, **f64) for _ in range(n)], [torch.randn(out, **f64) for _ in range(n)]
    xx, addend = torch.randn(bs, n, dim, **f64), torch.randn(out, **f64)
    got = R.local_mlps(xx, w0, b0, w2, b2, 0.5, 0.25, 1.5, 0.75, 0.01, addend)
    exp = torch.stack([F.linear(F.leaky_relu(F.linear(xx[:, g], w0[g] * 0.5, b0[g] * 1.5), 0.01), w2[g] * 0.25, b2[g] * 0.75) + addend
                       for g in range(n)], dim=1)
    assert got.shape == exp.shape and torch.allclose(got, exp, rtol=1e-12)
