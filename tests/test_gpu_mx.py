"""GPU parity of the DMA-fed masked kernel (csrc/modconv_mx.hip, round 3): its split-bf16 arithmetic must reproduce the register-staged kernel
bit for bit (same products, same accumulation order), its f16 + 2 x MX-fp6 arithmetic must stay within a few 1e-5 of a layer's output scale of the
faithful CPU oracle; ragged sizes, channel tails, region-less pixels, up layers, the split-K route, the fused ToRGB epilogue."""
import numpy as np
import pytest
import torch

from conftest import install_dropin, record_parity
from e4s2024_amd import ops
from oracle import e4s_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(ops.MODCONV_MODE != "sb", reason="these kernels are the split-arithmetic routes (E4S_MODCONV=f32 switches them off)")]
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
MX_LAYER_TOL = 2e-4      # of the layer's output scale, as the split-bf16 kernels' single-layer bar (measured: see profiles/r03_parity.json)


@pytest.fixture(scope="module")
def sg2():
    install_dropin()
    from models.stylegan2 import model
    return model


@pytest.fixture
def mx_mode():
    keep = ops.MX_MODE

    def set_mode(m):
        ops.MX_MODE = m
    yield set_mode
    ops.MX_MODE = keep


#        bs cin cout  h   w  nreg lh  lw
SHAPES = [(2, 64, 128, 32, 32, 5, 64, 64),        # one tile row of workgroups, one output-channel tile
          (1, 48, 136, 40, 36, 12, 80, 72),       # ragged: partial tiles, an output-channel tail (two co tiles), labels at another resolution
          (1, 512, 512, 32, 32, 12, 512, 512),    # the 32 x 32 layer of the generator: long K, split over workgroups
          (3, 32, 256, 64, 64, 7, 64, 64)]


def _layer(sg2, shape, upsample, seed):
    bs, cin, cout, h, w, nreg, lh, lw = shape
    rs = np.random.RandomState(seed)
    lab = rs.randint(0, nreg, (bs, lh, lw)).astype(np.uint8)
    lab[:, : max(1, lh // 7), : max(1, lw // 5)] = 255                       # a corner that belongs to no region
    onehot = torch.zeros(bs, nreg, lh, lw)
    for c in range(nreg):
        onehot[:, c] = T((lab == c).astype(np.float32))
    m = sg2.StyledConv(cin, cout, 3, 512, upsample=upsample, mask_op=True)
    with torch.no_grad():
        m.conv.weight.copy_(T(rs.standard_normal(m.conv.weight.shape).astype(np.float32)))
        m.conv.modulation.weight.copy_(T(rs.standard_normal(m.conv.modulation.weight.shape).astype(np.float32)))
        m.noise.weight.fill_(0.21)
        m.activate.bias.copy_(T(0.1 * rs.standard_normal(cout).astype(np.float32)))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = T(rs.standard_normal((bs, cin, h, w)).astype(np.float32))
    st = T(rs.standard_normal((bs, nreg, 512)).astype(np.float32))
    ho, wo = (2 * h, 2 * w) if upsample else (h, w)
    nz = T(rs.standard_normal((bs, 1, ho, wo)).astype(np.float32))
    return m.to(DEV), sd, x, st, lab, onehot, nz


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("upsample", [False, True])
def test_mx_pipeline_with_split_bf16_is_bit_identical_and_f16_fp6_meets_the_layer_bar(sg2, mx_mode, shape, upsample):
    m, sd, x, st, lab, onehot, nz = _layer(sg2, shape, upsample, 31 * shape[1] + shape[3] + (5 if upsample else 0))
    ys = {}
    with torch.no_grad():
        for mode in (0, 1, 2):
            mx_mode(mode)
            ys[mode] = m(x.to(DEV), st.to(DEV), T(lab).to(DEV), noise=nz.to(DEV)).cpu()
    assert torch.equal(ys[0], ys[1]), "the DMA pipeline with the split-bf16 arithmetic must equal the register-staged kernel bit for bit"
    ref = O.styled_conv(sd, "", x, st, onehot, nz, masked=True, upsample=upsample)
    scale = max(1.0, float(ref.abs().max()))
    e_sb = float((ys[1] - ref).abs().max()) / scale
    e_mx = float((ys[2] - ref).abs().max()) / scale
    record_parity(f"mx_layer_{'up' if upsample else 'same'}_{shape[1]}to{shape[2]}_{shape[3]}x{shape[4]}", e_mx, MX_LAYER_TOL,
                  note=f"f16 + 2 x MX fp6 against the oracle, relative to the output scale {scale:.1f}; split-bf16 on the same layer: {e_sb:.2e}")
    assert e_mx <= MX_LAYER_TOL, (shape, upsample, e_mx, e_sb)
    assert not ops.mx_overflowed()


def test_mx_raises_its_flag_when_an_activation_leaves_the_f16_range(sg2, mx_mode):
    m, sd, x, st, lab, onehot, nz = _layer(sg2, SHAPES[0], False, 9)
    mx_mode(2)
    with torch.no_grad():
        m(x.to(DEV), st.to(DEV), T(lab).to(DEV), noise=nz.to(DEV))
        assert not ops.mx_overflowed()
        m((x * 3.0e4).to(DEV), st.to(DEV), T(lab).to(DEV), noise=nz.to(DEV))       # |x * s| well past 65504
        assert ops.mx_overflowed()
        assert not ops.mx_overflowed()                                             # reading resets it


def test_mx_fused_torgb_and_split_plane_handover_equal_the_register_staged_kernel(sg2, mx_mode):
    """The last masked layer of Generator(1024) (128 -> 128 @ 256, fused single-region ToRGB, split-plane output for the chain) through
    ``Generator.forward`` at size 256: identical images from the two pipelines under the split-bf16 arithmetic."""
    from e4s2024_amd import seeded
    torch.manual_seed(3)
    g = sg2.Generator(256, 512, 8, remaining_layer_idx=9).to(DEV).eval()
    codes = seeded.seeded_codes(1, 2, 12, g.n_latent, seeded.seeded_latent_avg(2, g.n_latent)).to(DEV)
    mask = seeded.labels_to_onehot(seeded.blocky_labels(3, 2, 12, 512, 16), 12).to(DEV)
    imgs = {}
    with torch.no_grad():
        for mode in (0, 1, 2):
            mx_mode(mode)
            imgs[mode] = g([codes], None, mask, input_is_latent=True, randomize_noise=False)[0].cpu()
    assert torch.equal(imgs[0], imgs[1])
    d = float((imgs[2] - imgs[0]).abs().max())
    record_parity("mx_generator256_vs_split_bf16", d, 5e-4, note="f16 + 2 x MX fp6 against the split-bf16 arithmetic, pixels")
    assert d <= 5e-4


# ---------------------------------------------------------------------------------------------- plain-convolution mode (the encoder's 3x3 convs)
#             bs cin cout  h   w
CONV_SHAPES = [(16, 64, 128, 32, 32), (2, 48, 136, 40, 36), (4, 128, 256, 64, 64)]


@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("norm_prelu", [True, False])
def test_conv3x3_mx_against_fp64_and_the_direct_kernel(shape, norm_prelu):
    """``e4s_conv3x3_mx`` (instance norm on load, PReLU epilogue — the two convolutions of ``bottleneck_IR_SE_Ours``, helpers.py:128-139) against a float64
    convolution on the host and against the direct split-bf16 kernel: ragged sizes, an output-channel tail, padding that must stay exactly zero."""
    bs, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(7 * cin + h)
    x = torch.randn(bs, cin, h, w, generator=g) * 2.0 + 0.5
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9.0)
    slope = torch.rand(cout, generator=g) * 0.5
    xd, wd = x.to(DEV), wgt.to(DEV)
    mean = rstd = None
    xn = x.double()
    if norm_prelu:
        mean = x.mean((2, 3))
        rstd = 1.0 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
        xn = (x.double() - mean.double()[:, :, None, None]) * rstd.double()[:, :, None, None]
    ref = torch.nn.functional.conv2d(xn, wgt.double(), padding=1)
    if norm_prelu:
        ref = torch.where(ref > 0, ref, ref * slope.double()[None, :, None, None])
    scale = float(ref.abs().max())
    in_norm = (mean.to(DEV), rstd.to(DEV)) if norm_prelu else None
    pr = slope.to(DEV) if norm_prelu else None
    with torch.no_grad():
        y_dir = ops.conv2d(xd, ops.PreparedConv().get(wd), 1, 1, in_norm=in_norm, prelu=pr).cpu()
        ys = {a: ops.conv3x3_mx(xd, ops.PreparedMx().get(wd, None, False, a), a, cout, in_norm=in_norm, prelu=pr).cpu() for a in (0, 1)}
    e_dir = float((y_dir.double() - ref).abs().max()) / scale
    e0 = float((ys[0].double() - ref).abs().max()) / scale
    e1 = float((ys[1].double() - ref).abs().max()) / scale
    record_parity(f"conv3x3_mx_{cin}to{cout}_{h}x{w}_{'norm_prelu' if norm_prelu else 'plain'}", e1, MX_LAYER_TOL,
                  note=f"f16 + 2 x MX fp6 against float64, relative to the output scale; split-bf16 in the same kernel {e0:.2e}, direct kernel {e_dir:.2e}")
    assert e0 <= 2e-5 and e1 <= MX_LAYER_TOL, (shape, e0, e1, e_dir)
    assert float((ys[0] - y_dir).abs().max()) / scale <= 2e-5
    assert not ops.mx_overflowed()


#              bs cin cout  h   w        (cin % 32 == 0; ragged maps, an output-channel tail, one chunk / several, a map smaller than a tile)
CONV3_SHAPES = [(16, 64, 128, 32, 32), (2, 32, 136, 40, 36), (4, 128, 256, 64, 64), (1, 96, 128, 7, 45), (3, 512, 512, 32, 32)]


@pytest.mark.parametrize("shape", CONV3_SHAPES)
@pytest.mark.parametrize("norm_prelu", [True, False])
def test_conv3x3_mx3_against_fp64_and_the_direct_kernel(shape, norm_prelu):
    """``e4s_conv3x3_mx3`` — the same operator and arithmetic as ``e4s_conv3x3_mx`` (f16 + 2 x MX fp6; helpers.py:128-139) on the two-phase kernel with
    32-channel chunks and activations converted once per staged value — against float64 and against the direct split-bf16 kernel."""
    bs, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(11 * cin + h)
    x = torch.randn(bs, cin, h, w, generator=g) * 2.0 + 0.5
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9.0)
    slope = torch.rand(cout, generator=g) * 0.5
    xd, wd = x.to(DEV), wgt.to(DEV)
    mean = rstd = None
    xn = x.double()
    if norm_prelu:
        mean = x.mean((2, 3))
        rstd = 1.0 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
        xn = (x.double() - mean.double()[:, :, None, None]) * rstd.double()[:, :, None, None]
    ref = torch.nn.functional.conv2d(xn, wgt.double(), padding=1)
    if norm_prelu:
        ref = torch.where(ref > 0, ref, ref * slope.double()[None, :, None, None])
    scale = float(ref.abs().max())
    in_norm = (mean.to(DEV), rstd.to(DEV)) if norm_prelu else None
    pr = slope.to(DEV) if norm_prelu else None
    with torch.no_grad():
        y_dir = ops.conv2d(xd, ops.PreparedConv().get(wd), 1, 1, in_norm=in_norm, prelu=pr).cpu()
        w3 = ops.PreparedMx().get(wd, None, False, 3)
        y3 = ops.conv3x3_mx(xd, w3, 3, cout, in_norm=in_norm, prelu=pr).cpu()
        y3b = ops.conv3x3_mx(xd, w3, 3, cout, in_norm=in_norm, prelu=pr).cpu()
    e3 = float((y3.double() - ref).abs().max()) / scale
    record_parity(f"conv3x3_mx3_{cin}to{cout}_{h}x{w}_{'norm_prelu' if norm_prelu else 'plain'}", e3, MX_LAYER_TOL,
                  note=f"two-phase kernel, f16 + 2 x MX fp6 against float64, relative to the output scale; direct kernel {float((y_dir.double() - ref).abs().max()) / scale:.2e}")
    assert e3 <= MX_LAYER_TOL, (shape, e3)
    assert torch.equal(y3, y3b)                        # run-to-run identical (no race between the wave groups, the DMA ring and the single patch buffer)
    assert not ops.mx_overflowed()


#                 bs cin cout  h    w      (input size, even; ragged output maps, an output-channel tail, one chunk / several, an output smaller than a tile, the encoder's three)
CONV3_S2_SHAPES = [(2, 32, 136, 80, 72), (1, 96, 128, 14, 90), (3, 64, 128, 64, 64), (4, 128, 128, 128, 128), (2, 256, 256, 64, 64), (3, 512, 512, 64, 64)]


@pytest.mark.parametrize("shape", CONV3_S2_SHAPES)
@pytest.mark.parametrize("norm_prelu", [True, False])
def test_conv3x3_s2_mx3_against_fp64_and_the_direct_kernel(shape, norm_prelu):
    """``e4s_conv3x3_s2_mx3`` — the stride-2 3x3 convolution of a stage's first encoder unit (helpers.py:128-139 with stride 2) on the two-phase kernel, the input read
    as four phase planes — against float64 and the direct split-bf16 kernel; run-to-run identical; the default encoder route takes it."""
    bs, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(13 * cin + h)
    x = torch.randn(bs, cin, h, w, generator=g) * 2.0 + 0.5
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9.0)
    slope = torch.rand(cout, generator=g) * 0.5
    xd, wd = x.to(DEV), wgt.to(DEV)
    mean = rstd = None
    xn = x.double()
    if norm_prelu:
        mean = x.mean((2, 3))
        rstd = 1.0 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
        xn = (x.double() - mean.double()[:, :, None, None]) * rstd.double()[:, :, None, None]
    ref = torch.nn.functional.conv2d(xn, wgt.double(), stride=2, padding=1)
    if norm_prelu:
        ref = torch.where(ref > 0, ref, ref * slope.double()[None, :, None, None])
    scale = float(ref.abs().max())
    in_norm = (mean.to(DEV), rstd.to(DEV)) if norm_prelu else None
    pr = slope.to(DEV) if norm_prelu else None
    with torch.no_grad():
        y_dir = ops.conv2d(xd, ops.PreparedConv().get(wd), 2, 1, in_norm=in_norm, prelu=pr).cpu()
        w5 = ops.PreparedMx().get(wd, None, False, 5)
        y = ops.conv3x3_s2_mx(xd, w5, cout, in_norm=in_norm, prelu=pr).cpu()
        yb = ops.conv3x3_s2_mx(xd, w5, cout, in_norm=in_norm, prelu=pr).cpu()
    assert tuple(y.shape) == tuple(ref.shape)
    e = float((y.double() - ref).abs().max()) / scale
    record_parity(f"conv3x3_s2_mx3_{cin}to{cout}_{h}x{w}_{'norm_prelu' if norm_prelu else 'plain'}", e, MX_LAYER_TOL,
                  note=f"stride 2, f16 + 2 x MX fp6 against float64, relative to the output scale; direct kernel {float((y_dir.double() - ref).abs().max()) / scale:.2e}")
    assert e <= MX_LAYER_TOL, (shape, e)
    assert torch.equal(y, yb)
    assert not ops.mx_overflowed()
    # the phase-plane hand-over: the same input as four half-resolution planes per channel gives the same bits ...
    x6 = torch.stack([torch.stack([xd[:, :, py::2, px::2] for px in (0, 1)], 2) for py in (0, 1)], 2).contiguous()
    with torch.no_grad():
        assert torch.equal(ops.conv3x3_s2_mx(x6, w5, cout, in_norm=in_norm, prelu=pr).cpu(), y)
        # ... and the stride-1 kernel writes that layout itself (its values unchanged)
        w3 = ops.PreparedMx().get(wd, None, False, 3)
        plain = ops.conv3x3_mx(xd, w3, 3, cout, in_norm=in_norm, prelu=pr)
        ph = ops.conv3x3_mx(xd, w3, 3, cout, in_norm=in_norm, prelu=pr, out_phased=True)
        assert tuple(ph.shape) == (bs, cout, 2, 2, h // 2, w // 2)
        for py in (0, 1):
            for px in (0, 1):
                assert torch.equal(ph[:, :, py, px], plain[:, :, py::2, px::2])
    if not norm_prelu and ops.mx_arith() == 1 and ops.S2_MX3:
        caches = (ops.PreparedConv(), ops.PreparedWinograd(), ops.PreparedMx())
        with torch.no_grad():
            routed = ops.conv3x3_s2(xd, wd, caches).cpu()
            took = ops.mx_conv_eligible(xd[:, :, ::2, ::2], cout)
            assert torch.equal(routed, y if took else y_dir)
            with ops.mx_exact():
                assert torch.equal(ops.conv3x3_s2(xd, wd, caches).cpu(), y_dir)         # the guard's re-run takes the split-bf16 kernel
    with pytest.raises(ValueError):
        ops.conv3x3_s2_mx(xd[:, :, 1:], w5, cout)


def test_conv3x3_mx3_is_bit_stable_beside_another_stream():
    """The two-phase kernel counts its own vector-memory requests (weight DMA and activation prefetch are issued from asm, its ``vmcnt`` waits leave the younger
    ones in flight): a miscount would show as a changing value when memory gets slower — with another stream's kernels beside it.  Bounded (~8 s): the
    512 -> 512 @32^2 x 16 launch on one stream, a stream of 256 -> 256 @64^2 launches of the same kernel and of the direct kernel on the other."""
    import time
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(16, 512, 32, 32, device=DEV, generator=g)
    w = torch.randn(512, 512, 3, 3, device=DEV, generator=g) * 0.02
    x2 = torch.randn(8, 256, 64, 64, device=DEV, generator=g)
    w2 = torch.randn(256, 256, 3, 3, device=DEV, generator=g) * 0.03
    mean, rstd = x.mean((2, 3)), 1.0 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
    with torch.no_grad():
        w3, w23, pc2 = ops.PreparedMx().get(w, None, False, 3), ops.PreparedMx().get(w2, None, False, 3), ops.PreparedConv().get(w2)
        ref = ops.conv3x3_mx(x, w3, 3, 512, in_norm=(mean, rstd)).clone()
        ref2 = ops.conv3x3_mx(x2, w23, 3, 256).clone()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        t0, rounds, bad = time.time(), 0, 0
        while time.time() - t0 < 6.0 and rounds < 400:
            with torch.cuda.stream(s1):
                outs = [ops.conv3x3_mx(x, w3, 3, 512, in_norm=(mean, rstd)) for _ in range(6)]
            with torch.cuda.stream(s2):
                outs2 = []
                for _ in range(4):
                    outs2.append(ops.conv3x3_mx(x2, w23, 3, 256))
                    ops.conv2d(x2, pc2, 1, 1)
            torch.cuda.synchronize()
            bad += sum(not torch.equal(o, ref) for o in outs) + sum(not torch.equal(o, ref2) for o in outs2)
            rounds += 1
    record_parity("conv3x3_mx3.two_stream_mismatches", bad, 0, note=f"{rounds} rounds of 6 + 4 launches")
    assert rounds >= 5 and bad == 0
    assert not ops.mx_overflowed()


def test_conv3x3_mx3_raises_the_overflow_flag():
    """A normalised activation beyond the f16 range raises ``flags[0]`` (the result is then not to be used), as in the one-phase kernel."""
    x = torch.randn(1, 32, 16, 32, device=DEV)
    x[0, 3, 5, 7] = 1.0e5
    wgt = torch.randn(128, 32, 3, 3, device=DEV) * 0.05
    with torch.no_grad():
        ops.conv3x3_mx(x, ops.PreparedMx().get(wgt, None, False, 3), 3, 128)
    assert ops.mx_overflowed()
    assert not ops.mx_overflowed()


def test_conv3x3_s2_mx3_is_bit_stable_beside_another_stream():
    """The stride-2 form shares the stride-1 kernel's hand-counted waits (one more wait per one-unit sub-chunk): its results must not change when memory gets slower.
    Bounded (~4 s): the encoder's 256 -> 256 @128 -> 64 layer, plain and from the phase-plane hand-over, beside a stream of stride-1 launches."""
    import time
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(8, 256, 128, 128, device=DEV, generator=g)
    w = torch.randn(256, 256, 3, 3, device=DEV, generator=g) * 0.03
    x2 = torch.randn(8, 256, 64, 64, device=DEV, generator=g)
    with torch.no_grad():
        w5, w3 = ops.PreparedMx().get(w, None, False, 5), ops.PreparedMx().get(w, None, False, 3)
        ref = ops.conv3x3_s2_mx(x, w5, 256).clone()
        x6 = torch.stack([torch.stack([x[:, :, py::2, px::2] for px in (0, 1)], 2) for py in (0, 1)], 2).contiguous()
        assert torch.equal(ops.conv3x3_s2_mx(x6, w5, 256), ref)
        ref2 = ops.conv3x3_mx(x2, w3, 3, 256).clone()
        # (round 5) the same inputs as channel-blocked maps and as prepared operands: eight / eight requests per patch thread instead of 32, their own wait counts
        x7 = x6.reshape(8, 64, 4, 2, 2, 64, 64).movedim(2, -1).contiguous()
        ident = torch.zeros(256, 256, 3, 3, device=DEV)
        ident[torch.arange(256), torch.arange(256), 1, 1] = 1.0          # the identity convolution writes x itself as prepared operands (f16-exact inputs below)
        xh = x.half().float()
        ref_h = ops.conv3x3_s2_mx(xh, w5, 256).clone()
        xo = ops.conv3x3_mx(xh, ops.PreparedMx().get(ident, None, False, 3), 3, 256, out_phased=True, out_prep=True)
        assert torch.equal(ops.conv3x3_s2_mx(x7, w5, 256), ref) and torch.equal(ops.conv3x3_s2_mx(xo, w5, 256), ref_h)
        x2o = ops.conv3x3_mx(x2.half().float(), ops.PreparedMx().get(ident, None, False, 3), 3, 256, out_prep=True)
        ref2_h = ops.conv3x3_mx(x2.half().float(), w3, 3, 256).clone()
        assert torch.equal(ops.conv3x3_mx(x2o, w3, 3, 256), ref2_h)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        t0, rounds, bad = time.time(), 0, 0
        while time.time() - t0 < 3.0 and rounds < 300:
            with torch.cuda.stream(s1):
                outs = [ops.conv3x3_s2_mx((x, x6, x7)[i % 3], w5, 256) for i in range(6)]
                outs_h = [ops.conv3x3_s2_mx(xo, w5, 256) for _ in range(3)]
            with torch.cuda.stream(s2):
                outs2 = [ops.conv3x3_mx(x2, w3, 3, 256) for _ in range(4)]
                outs2_h = [ops.conv3x3_mx(x2o, w3, 3, 256) for _ in range(4)]
            torch.cuda.synchronize()
            bad += sum(not torch.equal(o, ref) for o in outs) + sum(not torch.equal(o, ref2) for o in outs2)
            bad += sum(not torch.equal(o, ref_h) for o in outs_h) + sum(not torch.equal(o, ref2_h) for o in outs2_h)
            rounds += 1
    record_parity("conv3x3_s2_mx3.two_stream_mismatches", bad, 0, note=f"{rounds} rounds of 9 + 8 launches (plain, phase-plane, channel-blocked, prepared-operand inputs)")
    assert rounds >= 5 and bad == 0
    assert not ops.mx_overflowed()


def test_conv3x3_s2_mx3_raises_the_overflow_flag_and_the_guard_heals_it():
    """The stride-2 form reports an activation beyond the f16 range like the stride-1 kernel, and ``ops.guarded`` re-runs the layer on the split-bf16 kernel."""
    if ops.mx_arith() != 1 or not ops.S2_MX3:
        pytest.skip("the f16 + fp6 arithmetic is off")
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(2, 128, 256, 256, device=DEV, generator=g)              # (the encoder's first stride-2 layer)
    x[1, 5, 11, 13] = 1.0e5
    wgt = torch.randn(128, 128, 3, 3, device=DEV, generator=g) / (128 * 9) ** 0.5
    caches = (ops.PreparedConv(), ops.PreparedWinograd(), ops.PreparedMx())
    with torch.no_grad():
        assert ops.conv3x3_s2_takes_mx(2, 128, 128, 256, 256, x.device)
        ops.mx_overflowed()
        ops.conv3x3_s2(x, wgt, caches)
        assert ops.mx_overflowed()
        before = ops.mx_fallbacks
        out = ops.guarded(lambda: ops.conv3x3_s2(x, wgt, caches))
        assert ops.mx_fallbacks == before + 1
    ops.mx_overflowed()
    ref = torch.nn.functional.conv2d(x.double(), wgt.double(), stride=2, padding=1)
    assert torch.isfinite(out).all() and float((out.double() - ref).abs().max() / ref.abs().max()) <= 2e-5


def test_encoder_at_batch_16_takes_the_mx_route_and_matches_the_direct_route(mx_mode):
    """``FSEncoder_PSP`` on 16 images (what a batch-8 swap feeds it): its stride-1 3x3 convolutions with >= 128 output channels run on
    ``e4s_conv3x3_mx``; the style vectors must equal the all-direct route's (E4S_MX=0) within the encoder's parity bar."""
    from conftest import default_opts
    install_dropin()
    from models.networks import Net3
    from e4s2024_amd import seeded
    net = Net3(default_opts())
    seeded.apply_seeded(net.encoder, 4, "net3", prefix="encoder.")
    net = net.to(DEV).eval()
    img = seeded.seeded_image(5, 16, 1024).to(DEV)
    lab = torch.from_numpy(seeded.blocky_labels(3, 16, 12, 512, 16)).to(DEV).to(torch.uint8)
    out = {}
    with torch.no_grad():
        for mode in (0, 2):
            mx_mode(mode)
            with ops.KernelTimer() as kt:
                out[mode] = net.get_style_vectors(img, lab)[0].cpu()
            names = set(kt.summary())
            assert any(n.startswith("conv3x3_mx") for n in names) == (mode == 2), names
    scale = float(out[0].abs().max())
    d = float((out[2] - out[0]).abs().max()) / scale
    record_parity("encoder_bs16_mx_vs_direct_route.style_vectors", d, 1e-3, note=f"relative to the largest style vector entry {scale:.2f}")
    assert d <= 1e-3
    assert not ops.mx_overflowed()


def _to_c4(t):
    """[bs, c, ...] -> [bs, c / 4, ..., 4]: the channel-blocked hand-over layout."""
    bs, c = t.shape[:2]
    return t.reshape(bs, c // 4, 4, *t.shape[2:]).movedim(2, -1).contiguous()


@pytest.mark.parametrize("shape", [(2, 64, 160, 40, 36), (1, 128, 128, 64, 64), (3, 32, 256, 24, 70)])
def test_channel_blocked_hand_over_between_two_convolutions_keeps_every_bit(shape):
    """Round 5 (csrc/conv_mx3.hip, ``e4s_conv3x3_mx3_ex``): the map between the two 3x3 convolutions of an IR-SE unit (helpers.py:128-139) handed over as
    ``[bs, c / 4, h, w, 4]`` — the producer's blocked output IS the plain output permuted, the consumer's result from it IS its result from the plain map; the same
    for the stride-2 pair (phase planes + channel blocks); ragged maps, an output-channel tail."""
    if ops.mx_arith() != 1 or not ops.MX3:
        pytest.skip("the two-phase f16 + fp6 kernel is off")
    bs, cin, depth, h, w = shape
    g = torch.Generator(device=DEV).manual_seed(7 * cin + w)
    x = torch.randn(bs, cin, h, w, device=DEV, generator=g) * 1.5 + 0.3
    w1 = torch.randn(depth, cin, 3, 3, device=DEV, generator=g) / (cin * 9) ** 0.5
    w2 = torch.randn(depth, depth, 3, 3, device=DEV, generator=g) / (depth * 9) ** 0.5
    slope = torch.rand(depth, device=DEV, generator=g) * 0.5
    mean = x.mean((2, 3))
    rstd = 1.0 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5)
    with torch.no_grad():
        w13, w23, w25 = (ops.PreparedMx().get(w1, None, False, 3), ops.PreparedMx().get(w2, None, False, 3), ops.PreparedMx().get(w2, None, False, 5))
        r = ops.conv3x3_mx(x, w13, 3, depth, in_norm=(mean, rstd), prelu=slope)
        r4 = ops.conv3x3_mx(x, w13, 3, depth, in_norm=(mean, rstd), prelu=slope, out_c4=True)
        assert tuple(r4.shape) == (bs, depth // 4, h, w, 4) and torch.equal(r4, _to_c4(r))
        y = ops.conv3x3_mx(r, w23, 3, depth)
        assert torch.equal(ops.conv3x3_mx(r4, w23, 3, depth), y)
        # blocked in, blocked out (a chain of such links), with the consumer's own normalisation and PReLU
        m2, s2 = r.mean((2, 3)), 1.0 / torch.sqrt(r.var((2, 3), unbiased=False) + 1e-5)
        y2 = ops.conv3x3_mx(r, w23, 3, depth, in_norm=(m2, s2), prelu=slope)
        assert torch.equal(ops.conv3x3_mx(r4, w23, 3, depth, in_norm=(m2, s2), prelu=slope, out_c4=True), _to_c4(y2))
        if h % 2 == 0 and w % 2 == 0:
            rp = ops.conv3x3_mx(x, w13, 3, depth, in_norm=(mean, rstd), prelu=slope, out_phased=True, out_c4=True)
            r6 = torch.stack([torch.stack([r[:, :, py::2, px::2] for px in (0, 1)], 2) for py in (0, 1)], 2).contiguous()
            assert tuple(rp.shape) == (bs, depth // 4, 2, 2, h // 2, w // 2, 4) and torch.equal(rp, _to_c4(r6))
            ys = ops.conv3x3_s2_mx(r, w25, depth)
            assert torch.equal(ops.conv3x3_s2_mx(rp, w25, depth), ys)
        # the map as the consumer's prepared operands (ops.MxOperandMap): conversion once per pixel in the producer's epilogue, the consumer's staging a copy
        if depth % 32 == 0 and (h * w) % 4 == 0:
            ro = ops.conv3x3_mx(x, w13, 3, depth, in_norm=(mean, rstd), prelu=slope, out_prep=True)
            assert isinstance(ro, ops.MxOperandMap) and ro.shape == (bs, depth, h, w) and tuple(ro.data.shape) == (bs, depth // 32, 29 * h * w)
            assert torch.equal(ops.conv3x3_mx(ro, w23, 3, depth), y)
            assert torch.equal(ops.conv3x3_mx(ro, w23, 3, depth, prelu=slope, out_c4=True), _to_c4(ops.conv3x3_mx(r, w23, 3, depth, prelu=slope)))
            if h % 2 == 0 and w % 2 == 0:
                rop = ops.conv3x3_mx(x, w13, 3, depth, in_norm=(mean, rstd), prelu=slope, out_phased=True, out_prep=True)
                assert rop.phased and torch.equal(ops.conv3x3_s2_mx(rop, w25, depth), ys)
    assert not ops.mx_overflowed()


def test_prepared_operand_producer_reports_the_f16_range():
    """The consumer of a prepared-operand map does not see fp32 values any more: the PRODUCER's epilogue raises the f16 flag for an output beyond the range."""
    if ops.mx_arith() != 1 or not ops.MX3:
        pytest.skip("the two-phase f16 + fp6 kernel is off")
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(1, 32, 32, 32, device=DEV, generator=g)
    w1 = torch.randn(64, 32, 3, 3, device=DEV, generator=g) / (32 * 9) ** 0.5
    with torch.no_grad():
        w13 = ops.PreparedMx().get(w1, None, False, 3)
        ops.mx_overflowed()
        ops.conv3x3_mx(x, w13, 3, 64, out_prep=True)
        assert not ops.mx_overflowed()
        big = ops.PreparedMx().get(w1 * 1e5, None, False, 3)
        ops.conv3x3_mx(x, big, 3, 64)                       # plain fp32 output: values of 1e5 are fine there
        assert not ops.mx_overflowed()
        ops.conv3x3_mx(x, big, 3, 64, out_prep=True)        # ... but not as f16 operands
        assert ops.mx_overflowed()


def test_encoder_units_take_the_channel_blocked_link_and_keep_their_values():
    """An IR-SE unit at the full swap's launch size (16 faces) with the hand-over between its two convolutions plain, channel-blocked and as prepared operands:
    the same output bits, and each link IS taken (stride 1 and stride 2)."""
    if ops.mx_arith() != 1 or not ops.MX3:
        pytest.skip("the two-phase f16 + fp6 kernel is off")
    from conftest import install_dropin
    install_dropin()
    from models.encoders.psp_encoders import bottleneck_IR_SE_Ours
    g = torch.Generator(device=DEV).manual_seed(31)
    for cin, depth, stride, h in ((256, 256, 1, 64), (128, 256, 2, 128)):
        unit = bottleneck_IR_SE_Ours(cin, depth, stride).to(DEV).eval()
        with torch.no_grad():
            for p_ in unit.parameters():
                p_.copy_(torch.randn(p_.shape, device=DEV, generator=g) * (0.05 if p_.dim() == 4 else 0.25))
        x = torch.randn(16, cin, h, h, device=DEV, generator=g)
        keep = (ops.ENC_C4_LINK, ops.ENC_PREP_LINK)
        outs = {}
        try:
            for on in (False, "c4", "prep"):
                ops.ENC_C4_LINK, ops.ENC_PREP_LINK = on == "c4", on == "prep"
                names = []
                orig = ops.lib().call

                def call(name, *a, _names=names, _orig=orig):
                    _names.append((name, a[-3], a[-2]) if name == "e4s_conv3x3_mx3_ex" else (name, a[-2]) if name == "e4s_conv3x3_s2_mx3" else (name,))
                    return _orig(name, *a)
                ops.lib().call = call
                try:
                    with torch.no_grad():
                        outs[on] = unit(x).clone()
                finally:
                    ops.lib().call = orig
                bit = {"c4": 2, "prep": 4}.get(on, 6)
                linked_out = [n for n in names if n[0] == "e4s_conv3x3_mx3_ex" and n[2] & bit]
                linked_in = [n for n in names if (n[0] == "e4s_conv3x3_mx3_ex" and n[1] & bit) or (n[0] == "e4s_conv3x3_s2_mx3" and n[1] & bit)]
                assert (len(linked_out), len(linked_in)) == ((1, 1) if on else (0, 0)), (cin, stride, on, names)
        finally:
            ops.ENC_C4_LINK, ops.ENC_PREP_LINK = keep
        assert torch.equal(outs[False], outs["c4"]) and torch.equal(outs[False], outs["prep"])
    assert not ops.mx_overflowed()
