"""Pin the oracle: every golden fixture (outputs of the reference itself, run on CPU in the
build container by tests/golden/make_golden.py) must be reproduced by oracle/e4s_oracle.py
and oracle/native_ops.c from the same seeded inputs.  No GPU, no reference tree needed."""
import numpy as np
import pytest
import torch

from conftest import load_golden, template_from_manifest
from e4s2024_amd import seeded
from oracle import e4s_oracle as O
from oracle import native as ON

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731


def close(a, b, tol):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    assert d <= tol, f"max-abs diff {d:.3e} > {tol:.1e}"


def test_g1_fused_act_python_and_c():
    g = load_golden("g1_fused_act")
    x, b, y = T(g["x"]), T(g["bias"]), T(g["y"])
    close(O.fused_leaky_relu(x, b), y, 0)
    gi, gb = O.fused_leaky_relu_backward(T(g["grad_out"]), y)
    close(gi, g["grad_in"], 1e-6)
    close(gb, g["grad_bias"], 1e-5)
    close(O.fused_leaky_relu(T(g["x2d"]), T(g["b2d"])), g["y2d"], 0)
    # C restatement of fused_bias_act_kernel: act=3, grad=0 / grad=1
    close(ON.fused_bias_act(g["x"], g["bias"], None, 3, 0, 0.2, 2 ** 0.5), y, 1e-6)
    close(ON.fused_bias_act(g["grad_out"], None, g["y"], 3, 1, 0.2, 2 ** 0.5), g["grad_in"], 1e-6)
    close(ON.fused_bias_act(g["x2d"], g["b2d"], None, 3, 0, 0.2, 2 ** 0.5), g["y2d"], 1e-6)


def test_g2_upfirdn2d_python_and_c():
    g = load_golden("g2_upfirdn2d")
    for name in g["names"]:
        x, k, y = g[f"{name}.x"], g[f"{name}.k"], g[f"{name}.y"]
        up, down, p0, p1 = (int(v) for v in g[f"{name}.p"])
        close(O.upfirdn2d(T(x), T(k), up, down, (p0, p1)), y, 1e-6)
        close(ON.upfirdn2d(x, k, (up, up), (down, down), (p0, p1, p0, p1)), y, 2e-6)


def test_upfirdn2d_edge_cases_c_vs_python():
    # empty batch, 1x1 kernel identity, anisotropic up/down/pad: C restatement == torch restatement
    k = np.array([[1.0]], dtype=np.float32)
    x = seeded.seeded_array(1, "edge.x", (2, 3, 4, 5), dist="normal")
    close(ON.upfirdn2d(x, k), x, 0)
    assert ON.upfirdn2d(np.zeros((0, 3, 4, 4), np.float32), k).shape == (0, 3, 4, 4)
    k2 = seeded.seeded_array(1, "edge.k", (4, 2), dist="normal")
    a = ON.upfirdn2d(x, k2, (2, 3), (1, 2), (1, 0, 2, 3))
    b = O.upfirdn2d_xy(T(x), T(k2), 2, 3, 1, 2, 1, 0, 2, 3)
    close(a, b, 2e-6)


def test_g3_modulated_conv():
    g = load_golden("g3_modconv")
    for name in ("same", "up", "rgb"):
        y = O.modulated_conv2d(T(g[f"{name}.x"]), T(g[f"{name}.s"]), T(g[f"{name}.w"]), T(g[f"{name}.mw"]), T(g[f"{name}.mb"]),
                               name != "rgb", name == "up", T(g["up.blur"]) if name == "up" else None)
        close(y, g[f"{name}.y"], 1e-5)


def test_g4_masked_layers():
    g = load_golden("g4_styled")
    mask = seeded.labels_to_onehot(g["labels"], 5)
    assert mask[:, 3].sum() == 0  # the empty region
    for name, up in (("same", False), ("up", True)):
        sd = {"L." + k[len(name) + 4:]: T(g[k]) for k in g.files if k.startswith(name + ".sd.")}
        y = O.styled_conv(sd, "L.", T(g[f"{name}.x"]), T(g[f"{name}.s"]), mask, T(g[f"{name}.nz"]), True, up)
        close(y, g[f"{name}.y"], 1e-5)
    sd = {"L." + k[7:]: T(g[k]) for k in g.files if k.startswith("rgb.sd.")}
    close(O.to_rgb(sd, "L.", T(g["rgb.x"]), T(g["rgb.s"]), mask, T(g["rgb.skip"]), True), g["rgb.y"], 1e-5)


def _gen_inputs(g, tag):
    size, rli, ncls, bs = (int(v) for v in g[tag + ".cfg"])
    lab = seeded.blocky_labels(22, bs, ncls, 64, cells=8)
    lab[:, 5:9, 3:40] = (lab[:, 5:9, 3:40] + 1) % ncls
    assert (lab == g[tag + ".labels"]).all()
    n_latent = int(np.log2(size)) * 2 - 2
    codes = seeded.seeded_codes(23, bs, ncls, n_latent, seeded.seeded_latent_avg(2, n_latent))
    return size, rli, ncls, bs, seeded.labels_to_onehot(lab, ncls), codes


@pytest.mark.parametrize("tag,man", [("s64", "generator_64_rli5"), ("s256", "generator_256_rli13")])
def test_g5_small_generators(manifest, tag, man):
    g = load_golden("g5_generator_small")
    size, rli, ncls, bs, mask, codes = _gen_inputs(g, tag)
    sd = seeded.seeded_state_dict(template_from_manifest(manifest[man]), 21, "net3")
    img, feats = O.generator_forward(sd, codes, mask, None, size=size, remaining_layer_idx=rli)
    close(img, g[tag + ".image"], 2e-5)
    close(feats.flatten()[:: max(1, feats.numel() // 4096)], g[tag + ".feats_sample"], 2e-5)


def test_g7_style_vectors(net3_sd):
    g = load_golden("g7_style_vectors")
    lab = seeded.blocky_labels(3, 1, 12, 512, cells=16)
    lab[lab == 9] = 0
    lab[lab == 11] = 0
    lab[0, 100:131, 57:300] = 5
    vec, struct = O.get_style_vectors(net3_sd, seeded.seeded_image(5, 1, 1024), seeded.labels_to_onehot(lab, 12))
    close(vec, g["vectors"], 2e-5)
    assert vec[0, 9].abs().max() == 0 and vec[0, 11].abs().max() == 0     # empty regions -> zeros
    assert tuple(struct.shape) == tuple(g["struct_shape"]) and struct.abs().max() == 0


def test_g8_style_codes(net3_sd):
    g = load_golden("g8_style_codes")
    codes = O.cal_style_codes(net3_sd, T(g["vectors"]), seeded.seeded_latent_avg(2, 18), 13)
    assert tuple(codes.shape) == tuple(g["shape"])
    close(codes.flatten()[g["idx"]], g["codes_sample"], 1e-5)
    assert abs(codes.double().sum().item() - float(g["codes_sum"])) < 1e-2
    # layers 13..17 are latent_avg for every region (models/networks.py:248)
    close(codes[0, 5, 13:], seeded.seeded_latent_avg(2, 18)[13:], 0)


def test_g6_generator_1024(net3_sd):
    g = load_golden("g6_gen1024")
    la = seeded.seeded_latent_avg(2, 18)
    codes = seeded.seeded_codes(1, 1, 12, 18, la)
    mask = seeded.labels_to_onehot(seeded.blocky_labels(3, 1, 12, 512, cells=16), 12)
    img, feats = O.generator_forward(net3_sd, codes, mask, None)
    close(img.flatten()[g["pix_idx"]], g["pix"], 5e-5)
    close(img[0, :, 480:544, 480:544], g["crop"], 5e-5)
    close(img[0, :, 777, :], g["row"], 5e-5)
    close(feats.flatten()[::32], g["feats_sample"], 5e-5)


def test_g9_g10_parser(bisenet_sd):
    g9, g10 = load_golden("g9_bisenet"), load_golden("g10_preprocess")
    close(O.bicubic_downsample(T(g10["img64"]), 2), g10["down64"], 1e-6)
    close(O.bicubic_taps(2), g10["taps"], 1e-7)
    img01 = (seeded.seeded_image(5, 1, 1024) + 1) / 2
    img01 = (torch.nn.functional.avg_pool2d(img01, 31, 1, 15) * 3 - 1).clamp(0, 1)
    x = O.parser_preprocess(img01)
    close(x.flatten()[::257], g10["x_sample"], 1e-5)
    logits = O.bisenet_forward(bisenet_sd, x)
    ref = T(g9["logits_sample"])
    got = logits[0].reshape(19, -1)[:, g9["pix_idx"]]
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-3
    seg = torch.argmax(logits, 1)[0].numpy().astype(np.uint8)
    bad = seg != g9["seg"]
    # a CPU re-run may re-associate sums: any flipped pixel must be a near-tie in the reference
    assert bad.sum() <= 8
    assert (O.remap_19_to_12(g9["seg"]) == g9["seg12"]).all()
    assert (O.remap_19_to_12(np.arange(19, dtype=np.uint8)) == g9["remap_table"]).all()


def test_g11_helpers():
    g = load_golden("g11_helpers")
    assert (O.tensor2im_array(T(g["img"]).clone()) == g["im"]).all()
    assert (O.label_map_to_onehot(T(g["lab"]), 12).numpy() == g["onehot"]).all()


def test_synthesis_flop_count():
    assert abs(O.synthesis_flops(1024) / 1e9 - 148.52) < 0.05


G12_CASES = ("generic", "ragged", "no_eyes", "no_eyes_no_brows_no_nose", "iid", "target_all_bg", "skin_row0")


def test_g12_mask_surgery():
    """f2/f3: swap_head_mask_hole_first and create_masks('expansion') restated in numpy vs the reference's own outputs (integer, exact)."""
    g = load_golden("g12_mask_surgery")
    for name in G12_CASES:
        res, hole, hole_map, nose_line, eye_line = O.swap_head_mask_hole_first(g[f"{name}.source"], g[f"{name}.target"])
        assert (res == g[f"{name}.res"]).all() and (hole == g[f"{name}.hole"].astype(bool)).all() and (hole_map == g[f"{name}.hole_map"]).all(), name
        assert [eye_line, nose_line] == g[f"{name}.lines"].tolist(), name
        _, border, full = O.create_masks_expansion(O.foreground_mask(res, hole)[None, None], 5)
        assert (border[0, 0] == g[f"{name}.border"]).all() and (full[0, 0] == g[f"{name}.full"]).all(), name
    m = g["morph.mask"].astype(np.float32)
    for r in (0, 1, 3, 7):
        _, border, full = O.create_masks_expansion(m, r)
        assert (border == g[f"morph.r{r}.border"]).all() and (full == g[f"morph.r{r}.full"]).all(), r


def test_pil_bicubic_resize_restatement_equals_pillow():
    """Row f3, face_swap_video_pipeline.py:447: the oracle's integer restatement of ``PIL.Image.resize`` (default BICUBIC, Resample.c) equals
    Pillow itself bit for bit — down, up, ragged, tiny — and the product's coefficient tables (ops._pil_bicubic_tables, sequential sums as
    in the C code) equal the oracle's."""
    PIL = pytest.importorskip("PIL.Image")
    import torch
    from e4s2024_amd import ops
    rs = np.random.RandomState(0)
    for (h, w), (ow, oh) in [((37, 53), (20, 31)), ((64, 64), (32, 32)), ((32, 32), (64, 64)), ((50, 41), (123, 77)), ((9, 7), (3, 2)),
                             ((128, 96), (128, 48)), ((5, 5), (5, 5))]:
        a = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        ref = np.array(PIL.fromarray(a).resize((ow, oh)))
        assert np.array_equal(O.pil_resize_bicubic(a, (ow, oh)), ref), ((h, w), (ow, oh))
    for n_in, n_out in [(1024, 512), (512, 1024), (53, 20), (41, 123)]:
        xmin, cnt, kk = O.pil_resample_coeffs(n_in, n_out)
        t_xmin, t_cnt, t_kk, ksize = ops._pil_bicubic_tables(n_in, n_out, torch.device("cpu"))
        assert ksize == kk.shape[1] and np.array_equal(t_xmin.numpy(), xmin) and np.array_equal(t_cnt.numpy(), cnt) and np.array_equal(t_kk.numpy(), kk)


G13_CASES = ("default", "no_teeth", "teeth_cancel", "below_face", "below_no_teeth", "no_indices", "all_indices")


def test_g13_style_vector_mix():
    """f2: ``swap_comp_style_vector`` (swap_face_fine/swap_face_mask.py:336-367) — the oracle restatement AND the product's batched,
    sync-free ``pipeline.mix_style_vectors`` (pure tensor plumbing, runs on any device) reproduce the reference's own outputs bit for bit,
    incl. a driven face without teeth, a teeth vector that merely sums to zero, neck interpolation on/off, and a batch = per-sample calls."""
    from e4s2024_amd import pipeline
    g = load_golden("g13_style_mix")
    assert tuple(g["names"]) == G13_CASES
    for name in G13_CASES:
        t, s, idx, below = T(g[f"{name}.target"]), T(g[f"{name}.source"]), [int(i) for i in g[f"{name}.idx"]], bool(g[f"{name}.below"])
        assert torch.equal(O.swap_comp_style_vector(t, s, idx, below), T(g[f"{name}.out"])), name
        assert torch.equal(pipeline.mix_style_vectors(t, s, idx, below), T(g[f"{name}.out"])), name
    t, s, idx = T(g["batch.target"]), T(g["batch.source"]), [int(i) for i in g["batch.idx"]]
    assert idx == list(pipeline.DEFAULT_COMP_INDICES)
    for below in (False, True):
        assert torch.equal(O.swap_comp_style_vector(t, s, idx, below), T(g[f"batch.out_below{int(below)}"]))
        assert torch.equal(pipeline.mix_style_vectors(t, s, idx, below), T(g[f"batch.out_below{int(below)}"]))


def test_erode_mask_restatement_against_an_independent_erosion():
    """f1: ``erode_mask`` (training/video_swap_ft_coach.py:64-93).  cv2 is not available, so the flat erosion is checked against
    scipy.ndimage.binary_erosion (box structuring element, border_value 0 = cv2's BORDER_CONSTANT 0) on seeded label maps."""
    ndi = pytest.importorskip("scipy.ndimage")
    rs = np.random.RandomState(3)
    for (h, w), r in (((64, 64), 3), ((40, 57), 1), ((33, 33), 5), ((16, 20), 0)):
        lab = np.repeat(np.repeat(rs.randint(0, 12, (h // 4 + 1, w // 4 + 1)), 4, 0), 4, 1)[:h, :w].astype(np.uint8)
        face = ~np.isin(lab, (0, 4, 11))
        er = ndi.binary_erosion(face, structure=np.ones((2 * r + 1, 2 * r + 1), bool), border_value=0) if r else face
        want = np.where(er, lab, 0).astype(np.uint8)
        assert np.array_equal(O.erode_mask(lab, r), want), ((h, w), r)


def test_frames_to_tensor_restatement():
    a = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, 3)
    t = O.frames_to_tensor(a)
    assert t.shape == (1, 3, 16, 16) and t.min().item() == -1.0 and t.max().item() == 1.0
    assert torch.equal(t[0, 0].flatten(), (torch.arange(256, dtype=torch.float32) / 255 - 0.5) / 0.5)
