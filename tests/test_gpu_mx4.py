"""GPU parity of the four-parity masked up kernel (csrc/modconv_mx4.hip, round 4): the reference's up-sampling StyledConv (models/stylegan2/model.py:287-300 per
region, mixed per output pixel :385-400).  The tiles whose positions' 2 x 2 outputs share a region run as four-parity tiles, the others — in the same launch — as the
composed kernel's tiles (csrc/modconv_mx_tile.h): the result must equal the composed kernel's own launch BIT FOR BIT (same products, same order) — on maps where every
tile qualifies, where some do, where none does, with region-less pixels and ragged sizes — and meet the layer bar against the faithful CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import install_dropin, record_parity
from e4s2024_amd import ops
from oracle import e4s_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(ops.MODCONV_MODE != "sb", reason="these kernels are the split-arithmetic routes (E4S_MODCONV=f32 switches them off)")]
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
MX_LAYER_TOL = 2e-4


@pytest.fixture(scope="module")
def sg2():
    install_dropin()
    from models.stylegan2 import model
    return model


def _labels(kind, rs, bs, nreg, lh, lw, ho, wo):
    """Region maps at the label resolution; what matters is the map sampled 'nearest' at the output resolution (ho, wo)."""
    if kind == "iid":                                   # no position has four equal outputs (almost surely): every tile stays with the composed kernel
        return rs.randint(0, nreg, (bs, lh, lw)).astype(np.uint8)
    cell = {"cells8": 8, "cells2": 2, "mixed": 8}[kind] * (lh // ho)          # cells of 8 / 2 OUTPUT pixels, aligned to even output coordinates
    gy, gx = -(-lh // cell), -(-lw // cell)
    lab = np.repeat(np.repeat(rs.randint(0, nreg, (bs, gy, gx)), cell, 1), cell, 2)[:, :lh, :lw].astype(np.uint8)
    if kind == "mixed":                                 # the right half: borders on ODD output columns -> those tiles do not qualify; plus a region-less block
        sh = lh // ho
        lab[:, :, lw // 2:] = np.roll(lab, sh, axis=2)[:, :, lw // 2:]
        lab[:, : lh // 4, : lw // 4] = 255
    return lab


#         bs cin cout  h   w  nreg lh   lw
SHAPES = [(2, 64, 128, 16, 64, 12, 128, 512),      # two tile rows, labels at 4x the output resolution
          (1, 48, 128, 20, 40, 5, 40, 80),         # ragged: partial tiles, a channel tail in the last chunk, labels at the output resolution
          (3, 32, 384, 8, 32, 7, 64, 256),         # three 128-channel tiles (six 64-channel ones: no XCD grouping), one tile per image
          (4, 512, 256, 64, 64, 12, 512, 512)]     # the generator's 64 -> 128 layer at the benchmark's batch


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("kind", ["cells8", "cells2", "mixed", "iid"])
def test_four_parity_launch_equals_the_composed_kernel(sg2, shape, kind):
    bs, cin, cout, h, w, nreg, lh, lw = shape
    rs = np.random.RandomState(17 * cin + h + len(kind))
    lab = _labels(kind, rs, bs, nreg, lh, lw, 2 * h, 2 * w)
    g = torch.Generator(device=DEV).manual_seed(cin + h)
    x = torch.randn(bs, cin, h, w, device=DEV, generator=g)
    wgt = torch.randn(1, cout, cin, 3, 3, device=DEV, generator=g)
    s = 1.0 + 0.3 * torch.randn(bs, nreg, cin, device=DEV, generator=g)
    d = torch.rand(bs, nreg, cout, device=DEV, generator=g) + 0.5
    nz = torch.randn(bs, 1, 2 * h, 2 * w, device=DEV, generator=g)
    nw, ab = torch.tensor([0.17], device=DEV), 0.1 * torch.randn(cout, device=DEV, generator=g)
    blur = torch.tensor([1., 3., 3., 1.], device=DEV)
    blur = blur[:, None] * blur[None, :]
    blur = blur / blur.sum() * 4
    labels = T(lab).to(DEV)
    wt, _ = ops.PreparedWeights().get(wgt, blur, True, True)
    wmx = ops.PreparedMx().get(wgt, blur, True, 1)
    wmx4 = ops.PreparedMx().get(wgt, blur, True, 4)
    ops.mx_overflowed()
    keep = ops.SPLITK_MAX_OUT_FLOATS
    try:
        ops.SPLITK_MAX_OUT_FLOATS = 0                    # (no split-K for the small test layers: its partial sums are added in another order)
        composed = ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, True, mx=(wmx, 1))
        poison = torch.full_like(composed, float("nan"))     # the next allocation of this size: every output pixel must be written
        torch.cuda.synchronize()
        del poison
        out = ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, True, mx=(wmx, 1), mx4=wmx4)
    finally:
        ops.SPLITK_MAX_OUT_FLOATS = keep
    assert torch.isfinite(out).all()
    assert torch.equal(out, composed), (shape, kind, float((out - composed).abs().max()))
    assert not ops.mx_overflowed()


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[1]])
def test_four_parity_kernel_against_the_oracle(sg2, shape):
    """``StyledConv(upsample=True, mask_op=True)`` through the module (model.py routes the layer itself) on a map made of 8-pixel cells: every tile on the new kernel."""
    bs, cin, cout, h, w, nreg, lh, lw = shape
    rs = np.random.RandomState(5 * cin + w)
    lab = _labels("cells8", rs, bs, nreg, lh, lw, 2 * h, 2 * w)
    onehot = torch.zeros(bs, nreg, lh, lw)
    for c in range(nreg):
        onehot[:, c] = T((lab == c).astype(np.float32))
    m = sg2.StyledConv(cin, cout, 3, 512, upsample=True, mask_op=True)
    with torch.no_grad():
        m.conv.weight.copy_(T(rs.standard_normal(m.conv.weight.shape).astype(np.float32)))
        m.conv.modulation.weight.copy_(T(rs.standard_normal(m.conv.modulation.weight.shape).astype(np.float32)))
        m.noise.weight.fill_(0.21)
        m.activate.bias.copy_(T(0.1 * rs.standard_normal(cout).astype(np.float32)))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = T(rs.standard_normal((bs, cin, h, w)).astype(np.float32))
    st = T(rs.standard_normal((bs, nreg, 512)).astype(np.float32))
    nz = T(rs.standard_normal((bs, 1, 2 * h, 2 * w)).astype(np.float32))
    m = m.to(DEV)
    keep = (ops.UP_MX4, ops.mx4_eligible)
    try:
        ops.mx4_eligible = lambda *a: ops.UP_MX4          # (the module asks this; small test layers would not fill the chip)
        with torch.no_grad():
            ops.UP_MX4 = True
            y4 = m(x.to(DEV), st.to(DEV), T(lab).to(DEV), noise=nz.to(DEV)).cpu()
            ops.UP_MX4 = False
            y1 = m(x.to(DEV), st.to(DEV), T(lab).to(DEV), noise=nz.to(DEV)).cpu()
    finally:
        ops.UP_MX4, ops.mx4_eligible = keep
    assert float((y4 - y1).abs().max()) <= 2e-5 * max(1.0, float(y1.abs().max()))      # (the composed kernel splits K on a layer this small: another summation order)
    ref = O.styled_conv(sd, "", x, st, onehot, nz, masked=True, upsample=True)
    scale = max(1.0, float(ref.abs().max()))
    e = float((y4 - ref).abs().max()) / scale
    record_parity(f"mx4_layer_up_{cin}to{cout}_{h}x{w}", e, MX_LAYER_TOL, note=f"four-parity up kernel against the oracle, relative to the output scale {scale:.1f}")
    assert e <= MX_LAYER_TOL

def test_four_parity_launch_is_bit_stable_across_repeats_and_streams(sg2):
    """The launch's LDS protocol (two-slot weight ring refilled by LDS-DMA behind a row's first MFMAs, double-buffered patch, role chosen per workgroup) has no
    data-dependent control flow: 24 launches of the generator's 64 -> 128 layer at the benchmark's batch, alternating over two streams, give 24 identical tensors."""
    bs, cin, cout, h, w, nreg, lh, lw = SHAPES[2]
    rs = np.random.RandomState(3)
    lab = _labels("mixed", rs, bs, nreg, lh, lw, 2 * h, 2 * w)
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(bs, cin, h, w, device=DEV, generator=g)
    wgt = torch.randn(1, cout, cin, 3, 3, device=DEV, generator=g)
    s = 1.0 + 0.3 * torch.randn(bs, nreg, cin, device=DEV, generator=g)
    d = torch.rand(bs, nreg, cout, device=DEV, generator=g) + 0.5
    nz = torch.randn(bs, 1, 2 * h, 2 * w, device=DEV, generator=g)
    nw, ab = torch.tensor([0.17], device=DEV), 0.1 * torch.randn(cout, device=DEV, generator=g)
    blur = torch.tensor([1., 3., 3., 1.], device=DEV)
    blur = blur[:, None] * blur[None, :]
    blur = blur / blur.sum() * 4
    labels = T(lab).to(DEV)
    wt, _ = ops.PreparedWeights().get(wgt, blur, True, True)
    wmx = ops.PreparedMx().get(wgt, blur, True, 1)
    wmx4 = ops.PreparedMx().get(wgt, blur, True, 4)
    ref = ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, True, mx=(wmx, 1), mx4=wmx4)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    outs = []
    for i in range(24):
        with torch.cuda.stream(streams[i & 1]):
            outs.append(ops.region_modconv3x3(x, wt, s, d, labels, nz, nw, ab, True, cout, True, mx=(wmx, 1), mx4=wmx4))
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    for st in streams:
        ops.release_stream_context(st)
