"""GPU parity of the face parser: BiSeNet + pre/post-processing."""
import numpy as np
import pytest
import torch
from PIL import Image

from conftest import load_golden, install_dropin, record_parity
from e4s2024_amd import ops, seeded
from oracle import e4s_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


def maxdiff(a, b):
    return (a.detach().double().cpu() - torch.as_tensor(b).double()).abs().max().item()


@pytest.fixture(scope="module")
def parser(bisenet_sd):
    install_dropin()
    from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
    p = FaceParser(seg_ckpt=None, device=DEV)
    p.seg.load_state_dict(bisenet_sd)
    p.seg.eval()
    return p


def _golden_image01():
    img01 = (seeded.seeded_image(5, 1, 1024) + 1) / 2
    return (torch.nn.functional.avg_pool2d(img01, 31, 1, 15) * 3 - 1).clamp(0, 1)


def test_module_imports_without_gpu_side_effects():
    install_dropin()
    from swap_face_fine.face_parsing import model
    assert model.seg_mean.device.type == "cpu" and tuple(model.seg_mean.shape) == (1, 3, 1, 1)


def test_g10_bicubic_downsample_and_preprocess(parser):
    g = load_golden("g10_preprocess")
    from swap_face_fine.face_parsing.face_parsing_demo import BicubicDownSample
    ds = BicubicDownSample(factor=2)
    assert maxdiff(ds.taps, g["taps"]) <= 1e-7
    assert maxdiff(ds(T(g["img64"]).to(DEV)), g["down64"]) <= 2e-6
    x = parser.preprocess_tensor(_golden_image01().to(DEV))
    assert tuple(x.shape) == (1, 3, 512, 512)
    assert maxdiff(x.flatten().cpu()[::257], g["x_sample"]) <= 1e-5
    ds4 = BicubicDownSample(factor=4)
    img = seeded.seeded_image(7, 1, 64)
    assert maxdiff(ds4(img.to(DEV)), O.bicubic_downsample(img, 4)) <= 2e-6
    # ragged sizes: partial tiles of the LDS-tiled kernel in both directions, images smaller and larger than one tile
    gen = torch.Generator().manual_seed(3)
    for (hh, ww, f) in ((50, 70, 2), (130, 258, 2), (36, 44, 4), (140, 132, 4)):
        im = torch.rand(2, 3, hh, ww, generator=gen)
        dsf = BicubicDownSample(factor=f)
        assert maxdiff(dsf(im.to(DEV)), O.bicubic_downsample(im, f)) <= 2e-6, (hh, ww, f)


def test_g9_bisenet_logits_and_argmax(parser, bisenet_sd):
    g = load_golden("g9_bisenet")
    x = O.parser_preprocess(_golden_image01())
    with torch.no_grad():
        out, out16, out32 = parser.seg(x.to(DEV))
        seg = parser.seg.parse(x.to(DEV))[0].cpu().numpy()
        seg12 = parser.seg.parse(x.to(DEV), parser._lut12)[0].cpu().numpy()
    assert tuple(out.shape) == (1, 19, 512, 512) and tuple(out16.shape) == (1, 19, 512, 512) and tuple(out32.shape) == (1, 19, 512, 512)
    ref = T(g["logits_sample"])
    scale = ref.abs().max().item()
    got = out[0].reshape(19, -1)[:, torch.from_numpy(g["pix_idx"]).to(DEV)]
    d = maxdiff(got, ref)
    record_parity("g9.bisenet_logits_vs_reference_golden", d, 5e-5 * scale, f"|logit|max = {scale:.1f}")
    assert d <= 5e-5 * scale
    # aux heads against the oracle
    ol, o16, o32 = O.bisenet_forward(bisenet_sd, x, aux=True)
    assert maxdiff(out16, o16) <= 5e-5 * scale and maxdiff(out32, o32) <= 5e-5 * scale
    # segmentation argmax: exact, except where the reference's own top-2 logits are closer than fp32 re-association noise
    assert (np.argmax(out[0].cpu().numpy(), 0) == seg).all()               # fused bilinear+argmax == argmax of materialised logits
    bad = np.argwhere(seg != g["seg"])
    record_parity("g9.bisenet_argmax_flips_vs_reference_golden", len(bad), 0, f"of {seg.size} pixels; reference min top-2 gap {float(g['gap_min']):.2e}")
    assert len(bad) == 0, f"{len(bad)} pixels of the reference's own argmax map moved (north_star: bit-exact segmentation argmax)"
    exp12 = O.remap_19_to_12(seg)
    assert (seg12 == exp12).all()
    same = seg == g["seg"]
    assert (seg12[same] == g["seg12"][same]).all()


def test_face_parsing_demo_pil_entry_both_branches(parser, bisenet_sd):
    """faceParsing_demo on PIL images: the >=512 branch (1024 -> bicubic /2) and the <512 branch of BASELINE config 1
    (256x256 face -> PIL bilinear resize to 512)."""
    from swap_face_fine.face_parsing.face_parsing_demo import faceParsing_demo
    rs = np.random.RandomState(0)
    for size in (1024, 256):
        small = rs.randint(0, 256, (size // 32, size // 32, 3)).astype(np.uint8)
        arr = np.asarray(Image.fromarray(small).resize((size, size), Image.BICUBIC))      # smooth-ish uint8 image
        pil = Image.fromarray(arr)
        seg12 = faceParsing_demo(parser, pil, convert_to_seg12=True)
        seg19 = faceParsing_demo(parser, pil, convert_to_seg12=False)
        assert seg12.dtype == np.uint8 and seg12.shape == (512, 512)
        if size >= 512:
            t = T(arr).permute(2, 0, 1)[None].float() / 255.0
            x = O.parser_preprocess(t)
        else:
            t = T(np.asarray(pil.resize((512, 512), Image.BILINEAR))).permute(2, 0, 1)[None].float() / 255.0
            m = torch.tensor(O.SEG_MEAN).view(1, 3, 1, 1); s = torch.tensor(O.SEG_STD).view(1, 3, 1, 1)
            x = (t.clamp(0, 1) - m) / s
        logits = O.bisenet_forward(bisenet_sd, x)
        ref = torch.argmax(logits, 1)[0].numpy().astype(np.uint8)
        top2 = torch.topk(logits[0], 2, dim=0).values
        gap = (top2[0] - top2[1]).numpy()
        bad = seg19 != ref
        assert bad.sum() <= 16 and (gap[bad] <= 1e-4 * logits.abs().max().item()).all(), (size, int(bad.sum()))
        assert (seg12 == O.remap_19_to_12(seg19)).all()
        lab = parser(pil)
        assert lab.dtype == torch.long and tuple(lab.shape) == (512, 512) and (lab.cpu().numpy() == seg19).all()


def test_parse_batch_matches_single(parser):
    img = ((seeded.seeded_image(9, 2, 1024) + 1) / 2).to(DEV)
    both = parser.parse_batch(img)
    one = parser.parse_batch(img[1:2].contiguous())
    assert both.dtype == torch.uint8 and tuple(both.shape) == (2, 512, 512) and int(both.max()) <= 11
    assert torch.equal(both[1], one[0])


def test_parse_batch_takes_the_pipelines_images_where_they_are(parser):
    """pipeline.swap_batch hands the parser the driven and the target faces as two [-1, 1] tensors: the network input built from them (each part
    down-sampled into its slice, ``(v + 1) * 0.5`` on load) must be the one built from ``(cat + 1) / 2`` bit for bit, and so must the labels."""
    a, b = seeded.seeded_image(9, 2, 1024).to(DEV), seeded.seeded_image(10, 3, 1024).to(DEV)
    ref_in = parser.preprocess_tensor((torch.cat([a, b]) + 1) / 2)
    assert torch.equal(parser.preprocess_tensor(b, pm1=True), ref_in[2:])
    out = torch.full((5, 3, 512, 512), float("nan"), device=DEV)
    parser.preprocess_tensor(a, pm1=True, out=out[:2])
    parser.preprocess_tensor(b, pm1=True, out=out[2:])
    assert torch.equal(out, ref_in)
    assert torch.equal(parser.parse_batch((a, b), pm1=True), parser.parse_batch((torch.cat([a, b]) + 1) / 2))
    with pytest.raises(ValueError):
        parser.preprocess_tensor(a, pm1=True, out=out[:3])
    small = ops.bilinear_resize(torch.cat([a, b]), (256, 256))
    part = torch.empty_like(small)
    ops.bilinear_resize(a, (256, 256), out=part[:2])
    ops.bilinear_resize(b, (256, 256), out=part[2:])
    assert torch.equal(part, small)


@pytest.mark.parametrize("shape", [(2, 5, 256, 256), (1, 3, 64, 36), (2, 2, 33, 47), (1, 1, 8, 4), (1, 4, 30, 62)])
def test_maxpool3x3s2_equals_torch(shape):
    """MaxPool2d(3, 2, 1) of the parser's ResNet stem (swap_face_fine/face_parsing/resnet.py:65): both kernels (two outputs per thread where the width is a
    multiple of four, one otherwise) give torch's maxima exactly."""
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(*shape, device=DEV, generator=g)
    assert torch.equal(ops.maxpool3x3s2(x), torch.nn.functional.max_pool2d(x, 3, 2, 1))


@pytest.mark.parametrize("shape", [(2, 19, 64, 64, 512, 512), (1, 5, 7, 9, 30, 50), (1, 19, 64, 64, 100, 130), (1, 3, 40, 40, 20, 24)])
def test_bilinear_argmax_against_the_same_arithmetic_in_torch(shape):
    """labels = argmax_c of the align_corners bilinear up-sampling of the logits (face_parsing/model.py:257 + face_parsing_demo.py:170), never materialised: the kernel
    (LDS-staged where a block's corner of the logits fits, direct otherwise: the last shape down-samples) against the same fp32 expression evaluated by torch.
    Where they differ (an FMA contraction can move a value by an ulp) the two classes' values must be that close."""
    bs, ncls, ih, iw, oh, ow = shape
    g = torch.Generator(device=DEV).manual_seed(ncls + ih)
    lg = torch.randn(bs, ncls, ih, iw, device=DEV, generator=g)
    got = ops.bilinear_argmax(lg, (oh, ow)).long()
    f32 = torch.float32
    sy = torch.tensor((ih - 1) / (oh - 1) if oh > 1 else 0.0, dtype=f32, device=DEV)
    sx = torch.tensor((iw - 1) / (ow - 1) if ow > 1 else 0.0, dtype=f32, device=DEV)
    fy = torch.arange(oh, device=DEV, dtype=f32) * sy
    fx = torch.arange(ow, device=DEV, dtype=f32) * sx
    y0 = fy.long().clamp(max=ih - 1); x0 = fx.long().clamp(max=iw - 1)
    y1 = (y0 + 1).clamp(max=ih - 1); x1 = (x0 + 1).clamp(max=iw - 1)
    ly = (fy - y0.to(f32))[:, None]; lx = (fx - x0.to(f32))[None, :]
    hy, hx = 1 - ly, 1 - lx
    q = lambda yy, xx: lg[:, :, yy][:, :, :, xx]
    v = hy * (hx * q(y0, x0) + lx * q(y0, x1)) + ly * (hx * q(y1, x0) + lx * q(y1, x1))
    ref = v.argmax(1)
    bad = got != ref
    if bad.any():
        top2 = v.topk(2, dim=1).values
        assert float((top2[:, 0] - top2[:, 1])[bad].max()) <= 1e-5 and int(bad.sum()) <= 8, int(bad.sum())


@pytest.mark.parametrize("shape", [(2, 512, 512), (1, 70, 90), (3, 33, 47), (1, 8, 260)])
def test_stem7_kernel_against_float64(shape):
    """Conv2d(3, 64, 7, 2, 3) + BatchNorm (eval) + ReLU of the parser's ResNet stem (face_parsing/resnet.py:57-58, 66) on csrc/stem7.hip (K = (c, ky, kx) flattened,
    two-term f16 split) against float64, and against the exact-fp32 kernel it replaces; ragged and odd sizes."""
    bs, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(bs, 3, h, w, generator=g) * 1.2
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    bn = torch.nn.BatchNorm2d(64).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(64, 3, 7, 7, generator=g) * 0.08)
        bn.weight.copy_(torch.rand(64, generator=g) + 0.5); bn.bias.copy_(torch.randn(64, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(64, generator=g) * 0.1); bn.running_var.copy_(torch.rand(64, generator=g) + 0.5)
    with torch.no_grad():
        ref = torch.relu(bn.double()(conv.double()(x.double())))
    conv, bn = conv.float().to(DEV), bn.float().to(DEV)
    xd = x.to(DEV)
    assert ops.STEM7 and ops.PARSER_EXACT == "f16x3"
    with torch.no_grad():
        prep = ops.PreparedConv(exact="f16x3").get(conv.weight, bn)
        assert prep[4] == "stem7"
        y = ops.conv2d(xd, prep, 2, 3, relu=True).cpu()
        with ops.mx_exact():
            y_exact = ops.conv2d(xd, ops.PreparedConv(exact="f16x3").get(conv.weight, bn), 2, 3, relu=True).cpu()
    scale = float(ref.abs().max())
    e, e_exact = float((y.double() - ref).abs().max()) / scale, float((y_exact.double() - ref).abs().max()) / scale
    record_parity(f"parser_stem7_{bs}x{h}x{w}.vs_float64", e, 2e-6, note=f"exact-fp32 kernel: {e_exact:.2e}")
    assert tuple(y.shape) == tuple(ref.shape) and e <= 2e-6, (shape, e, e_exact)
    with pytest.raises(ValueError):
        ops.conv2d(xd, prep, 1, 3)


def test_training_mode_is_refused(parser):
    parser.seg.train()
    try:
        with pytest.raises(RuntimeError):
            parser.seg(torch.zeros(1, 3, 64, 64, device=DEV))
    finally:
        parser.seg.eval()


def test_three_way_split_parser_against_exact_fp32_and_oracle(bisenet_sd):
    """The parser's convolutions in the two-term f16 split (3 f16 MFMAs per product, E4S_PARSER_CONV=f16x3: the default since round 3) and in the
    three-way bf16 split (6 bf16 MFMAs, ``sb3``) next to the exact-fp32 MFMA kernel (``f32``) on seeded images with random weights — a near-tie
    generator far harsher than real faces.  Logits agree to fp32 rounding; every pixel on which two argmaxes differ is a near-tie of the CPU
    oracle (top-2 gap below 1e-5 of the logit scale), and each variant disagrees with the oracle on such pixels only."""
    from e4s2024_amd import ops
    install_dropin()
    from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
    old = ops.PARSER_EXACT
    parsers = {}
    try:
        for mode in (True, "sb3", "f16x3"):
            ops.PARSER_EXACT = mode
            p = FaceParser(seg_ckpt=None, device=DEV)
            p.seg.load_state_dict(bisenet_sd)
            p.seg.eval()
            parsers[mode] = p
    finally:
        ops.PARSER_EXACT = old
    imgs = []
    for seed in range(2):
        raw = (seeded.seeded_image(20 + seed, 1, 1024) + 1) / 2
        imgs.append(raw.clamp(0, 1))                                                        # per-pixel noise
        imgs.append((torch.nn.functional.avg_pool2d(raw, 31, 1, 15) * 3 - 1).clamp(0, 1))   # smooth structure
    batch = torch.cat(imgs)
    with torch.no_grad():
        x = parsers[True].preprocess_tensor(batch.to(DEV))
        la = parsers[True].seg(x)[0].cpu()
        lb = parsers["sb3"].seg(x)[0].cpu()
        lc = parsers["f16x3"].seg(x)[0].cpu()
        ref = O.bisenet_forward(bisenet_sd, O.parser_preprocess(batch))
        ref = ref[0] if isinstance(ref, (tuple, list)) else ref
    scale = ref.abs().max().item()
    assert (la - lb).abs().max().item() <= 2e-5 * scale
    record_parity("parser.random_inputs.f16x3_logits_vs_exact_fp32", (la - lc).abs().max().item() / scale, 2e-5, "of the logit scale; three-way bf16: "
                  f"{(la - lb).abs().max().item() / scale:.2e}")
    assert (la - lc).abs().max().item() <= 2e-5 * scale
    top2 = ref.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]) / scale
    am_ref, am_a, am_b, am_c = ref.argmax(1), la.argmax(1), lb.argmax(1), lc.argmax(1)
    for name, am in (("exact fp32 MFMA", am_a), ("three-way split", am_b), ("two-term f16 split", am_c)):
        bad = am != am_ref
        n = int(bad.sum())
        record_parity(f"parser.random_inputs.argmax_flips_vs_oracle[{name}]", n, 64,
                      f"of {am.numel()} px; largest oracle top-2 gap among them {gap[bad].max().item() if n else 0.0:.2e} of the logit scale")
        assert n <= 64 and (n == 0 or gap[bad].max().item() < 1e-5)
    for am in (am_b, am_c):
        diff = am_a != am
        assert int(diff.sum()) <= 16 and (int(diff.sum()) == 0 or gap[diff].max().item() < 1e-5)
