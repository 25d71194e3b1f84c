"""GPU parity of the region-uniform block kernel of the masked up layers (csrc/modconv_upblock_mx.hip, round 5: f16 + 2 x MX-fp6) against the faithful CPU oracle
(models/stylegan2/model.py:385-400 restated in oracle/e4s_oracle.py) and against the all-composed route, and a two-stream back-to-back stress of the DMA-fed
kernels (the asm-issued request hazard of round 5, csrc/sb_common.h).  (The entry kernel csrc/modconv_mxe.hip these tests also covered in round 5 tied with
the kernel it was to replace; round 6's loop probe — profiles/r06_tile_probe.txt — says why, and it was deleted.)"""
import numpy as np
import pytest
import torch

from conftest import install_dropin, record_parity
from e4s2024_amd import ops, seeded
from oracle import e4s_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(ops.MODCONV_MODE != "sb" or ops.MX_MODE < 2, reason="the block kernel is the f16 + fp6 route of the split-arithmetic kernels")]
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
UB_LAYER_TOL = 2e-4      # of the layer's output scale: the single-layer bar of tests/test_gpu_mx.py


@pytest.fixture(scope="module")
def sg2():
    install_dropin()
    from models.stylegan2 import model
    return model


class _Calls:
    """Counts the library entry points a block of code calls."""

    def __enter__(self):
        self.names = []
        self.lib = ops.lib()
        self.orig = self.lib.call

        def call(name, *a):
            self.names.append(name)
            return self.orig(name, *a)
        self.lib.call = call
        return self

    def __exit__(self, *exc):
        self.lib.call = self.orig


def _labels(kind, rs, bs, nreg, lh, lw):
    if kind.startswith("cells"):
        c = int(kind[5:])
        small = rs.randint(0, nreg, (bs, -(-lh // c), -(-lw // c))).astype(np.uint8)
        return np.repeat(np.repeat(small, c, axis=1), c, axis=2)[:, :lh, :lw].copy()
    if kind == "one":
        return np.full((bs, lh, lw), nreg - 1, np.uint8)
    if kind == "iid":
        return rs.randint(0, nreg, (bs, lh, lw)).astype(np.uint8)
    if kind == "portrait":
        lab = seeded.facelike_labels(5, bs, 512)
        ys = (np.arange(lh) * 512 // lh)[:, None]
        xs = (np.arange(lw) * 512 // lw)[None, :]
        return np.minimum(lab[:, ys, xs], nreg - 1).astype(np.uint8)
    raise ValueError(kind)


def _layer(sg2, shape, kind, seed, hole=True):
    bs, cin, cout, h, w, nreg, lh, lw = shape
    rs = np.random.RandomState(seed)
    lab = _labels(kind, rs, bs, nreg, lh, lw)
    if hole:
        lab[:, : max(1, lh // 7), : max(1, lw // 5)] = 255                       # a corner that belongs to no region
    onehot = torch.zeros(bs, nreg, lh, lw)
    for c in range(nreg):
        onehot[:, c] = T((lab == c).astype(np.float32))
    m = sg2.StyledConv(cin, cout, 3, 512, upsample=False, mask_op=True)
    with torch.no_grad():
        m.conv.weight.copy_(T(rs.standard_normal(m.conv.weight.shape).astype(np.float32)))
        m.conv.modulation.weight.copy_(T(rs.standard_normal(m.conv.modulation.weight.shape).astype(np.float32)))
        m.noise.weight.fill_(0.21)
        m.activate.bias.copy_(T(0.1 * rs.standard_normal(cout).astype(np.float32)))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = T(rs.standard_normal((bs, cin, h, w)).astype(np.float32))
    st = T(rs.standard_normal((bs, nreg, 512)).astype(np.float32))
    nz = T(rs.standard_normal((bs, 1, h, w)).astype(np.float32))
    return m.to(DEV), sd, x, st, lab, onehot, nz


# ---------------------------------------------------------------------------------------------- region-uniform blocks of the masked up layers on f16 + fp6
#             bs cin cout  h   w  nreg lh  lw
UB_SHAPES = [(2, 32, 128, 32, 32, 5, 64, 64),          # one chunk
             (1, 64, 128, 32, 32, 5, 64, 64),          # two chunks
             (2, 128, 136, 32, 48, 12, 64, 96),        # an output-channel tail (three 64-channel tiles), several block columns
             (1, 256, 128, 64, 64, 12, 512, 512)]      # the 128 -> 256 layer's channel plan at half its size, labels at the mask resolution


@pytest.mark.parametrize("shape", UB_SHAPES)
def test_uniform_block_kernel_on_f16_fp6_against_the_oracle_and_the_composed_form(sg2, shape):
    """csrc/modconv_upblock_mx.hip: maps made of 16 x 16 output blocks of one region each, one row of blocks with per-pixel noise (mixed: composed form) and a corner
    without a region — every block written by exactly one kernel, the block kernel's f16 + fp6 output against the oracle and against the all-composed route."""
    bs, cin, cout, h, w, nreg, lh, lw = shape
    rs = np.random.RandomState(13 * cin + h)
    ho, wo = 2 * h, 2 * w
    cy, cx = 16 * lh // ho, 16 * lw // wo
    cells = rs.randint(0, nreg, (bs, ho // 16, wo // 16)).astype(np.uint8)
    lab = np.repeat(np.repeat(cells, cy, axis=1), cx, axis=2)
    lab[:, cy:2 * cy, : lw // 2] = rs.randint(0, nreg, (bs, cy, lw // 2))       # second row of blocks: noise in its left half
    lab[:, lh - cy // 2:, lw - cx // 2:] = 255                                  # no region
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
    nz = T(rs.standard_normal((bs, 1, ho, wo)).astype(np.float32))
    m = m.to(DEV)
    labd = T(lab).to(DEV)
    keep = (ops.UP_BLOCKS, ops.UP_BLOCKS_MIN_WIDTH, ops.UP_BLOCKS_MIN_PERCENT, ops.UP_BLOCKS_MIN_PERCENT_SMALL)
    ops.UP_BLOCKS_MIN_WIDTH, ops.UP_BLOCKS_MIN_PERCENT, ops.UP_BLOCKS_MIN_PERCENT_SMALL = 32, 1, 1
    ys = {}
    try:
        with torch.no_grad():
            for on in (False, True):
                ops.UP_BLOCKS = on
                with _Calls() as calls:
                    ys[on] = m(x.to(DEV), st.to(DEV), labd, noise=nz.to(DEV)).cpu()
                assert ("e4s_masked_upconv_blocks_mx" in calls.names) == on, calls.names
            again = m(x.to(DEV), st.to(DEV), labd, noise=nz.to(DEV)).cpu()
    finally:
        ops.UP_BLOCKS, ops.UP_BLOCKS_MIN_WIDTH, ops.UP_BLOCKS_MIN_PERCENT, ops.UP_BLOCKS_MIN_PERCENT_SMALL = keep
    assert torch.equal(again, ys[True])
    ref = O.styled_conv(sd, "", x, st, onehot, nz, masked=True, upsample=True)
    scale = max(1.0, float(ref.abs().max()))
    e_new = float((ys[True] - ref).abs().max()) / scale
    e_old = float((ys[False] - ref).abs().max()) / scale
    record_parity(f"upblock_mx_{cin}to{cout}_{h}x{w}", e_new, UB_LAYER_TOL, note=f"f16 + fp6 block kernel against the oracle, relative to the output scale {scale:.1f}; the all-composed route: {e_old:.2e}")
    assert e_new <= UB_LAYER_TOL and torch.isfinite(ys[True]).all(), (shape, e_new, e_old)
    assert not ops.mx_overflowed()


def test_dma_fed_kernels_back_to_back_on_two_streams_keep_their_bits(sg2):
    """The hazard of round 5 (a VALU-written SGPR read by an asm-issued vector-memory request five wait states too early, csrc/sb_common.h) showed only under
    back-to-back launches: the masked same-resolution kernel and the region-uniform block kernel, 60 launches each, alternating over two HIP streams with nothing between them —
    every output equals the first one's bits, nothing faults, the f16 flag stays down."""
    shape = (2, 128, 128, 64, 64, 12, 64, 64)
    m1, _, x1, st1, lab1, _, nz1 = _layer(sg2, shape, "cells8", 21)
    bs, cin, cout, h, w, nreg = 2, 128, 128, 32, 32, 12
    rs = np.random.RandomState(22)
    lab2 = np.repeat(np.repeat(rs.randint(0, nreg, (bs, 4, 4)).astype(np.uint8), 16, axis=1), 16, axis=2)      # 64 x 64: every 16 x 16 output block under one region
    m2 = sg2.StyledConv(cin, cout, 3, 512, upsample=True, mask_op=True)
    with torch.no_grad():
        m2.conv.weight.copy_(T(rs.standard_normal(m2.conv.weight.shape).astype(np.float32)))
        m2.conv.modulation.weight.copy_(T(rs.standard_normal(m2.conv.modulation.weight.shape).astype(np.float32)))
    m2 = m2.to(DEV)
    x2 = T(rs.standard_normal((bs, cin, h, w)).astype(np.float32)).to(DEV)
    st2 = T(rs.standard_normal((bs, nreg, 512)).astype(np.float32)).to(DEV)
    nz2 = T(rs.standard_normal((bs, 1, 2 * h, 2 * w)).astype(np.float32)).to(DEV)
    a1 = (x1.to(DEV), st1.to(DEV), T(lab1).to(DEV), nz1.to(DEV))
    a2 = (x2, st2, T(lab2).to(DEV), nz2)
    keep = (ops.UP_BLOCKS, ops.UP_BLOCKS_MIN_WIDTH, ops.UP_BLOCKS_MIN_PERCENT, ops.UP_BLOCKS_MIN_PERCENT_SMALL)
    ops.UP_BLOCKS, ops.UP_BLOCKS_MIN_WIDTH, ops.UP_BLOCKS_MIN_PERCENT, ops.UP_BLOCKS_MIN_PERCENT_SMALL = True, 32, 1, 1
    ops.mx_overflowed()
    streams = [torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)]
    try:
        with torch.no_grad():
            with _Calls() as calls:
                r1 = m1(a1[0], a1[1], a1[2], noise=a1[3])
                r2 = m2(a2[0], a2[1], a2[2], noise=a2[3])
            assert "e4s_region_modconv3x3_mx" in calls.names and "e4s_masked_upconv_blocks_mx" in calls.names, calls.names
            torch.cuda.synchronize()
            outs = []
            for i in range(60):
                for j, (m, a) in enumerate(((m1, a1), (m2, a2))):
                    with torch.cuda.stream(streams[(i + j) & 1]):
                        outs.append((j, m(a[0], a[1], a[2], noise=a[3])))
            torch.cuda.synchronize()
    finally:
        ops.UP_BLOCKS, ops.UP_BLOCKS_MIN_WIDTH, ops.UP_BLOCKS_MIN_PERCENT, ops.UP_BLOCKS_MIN_PERCENT_SMALL = keep
    for j, y in outs:
        assert torch.equal(y, r1 if j == 0 else r2)
    assert not ops.mx_overflowed()
