"""GPU parity of the DMA-fed masked kernel (csrc/modconv_mx.hip, round 3): its split-bf16 arithmetic must reproduce the register-staged kernel
bit for bit (same products, same accumulation order), its f16 + 2 x MX-fp6 arithmetic must stay within a few 1e-5 of a layer's output scale of the
faithful CPU oracle; ragged sizes, channel tails, region-less pixels, up layers, the split-K route, the fused ToRGB epilogue."""
import numpy as np
import pytest
import torch

from conftest import install_dropin, record_parity
from e4s2024_amd import ops
from oracle import e4s_oracle as O

pytestmark = pytest.mark.gpu
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
