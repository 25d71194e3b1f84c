"""GPU parity of the mask surgery between parsing and synthesis through the C ABI: bit-exact against the
golden fixtures (outputs of the reference's swap_head_mask_hole_first / create_masks) and against the numpy oracle at full size."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from e4s2024_amd import ops, seeded
from oracle import e4s_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
CASES = ("generic", "ragged", "no_eyes", "no_eyes_no_brows_no_nose", "iid", "target_all_bg", "skin_row0")


@pytest.mark.parametrize("name", CASES)
def test_swap_head_mask_golden(name):
    g = load_golden("g12_mask_surgery")
    src, tgt = T(g[f"{name}.source"])[None].to(DEV), T(g[f"{name}.target"])[None].to(DEV)
    res, hole, hole_map, lines = ops.swap_head_mask(src, tgt)
    assert (res[0].cpu().numpy() == g[f"{name}.res"]).all()
    assert (hole[0].cpu().numpy() == g[f"{name}.hole"]).all()
    assert (hole_map[0].cpu().numpy() == g[f"{name}.hole_map"]).all()
    assert lines[0].cpu().tolist() == g[f"{name}.lines"].tolist()
    content, border, full = ops.foreground_masks(res, hole, 5)
    assert (content[0, 0].cpu().numpy() == O.foreground_mask(g[f"{name}.res"], g[f"{name}.hole"].astype(bool))).all()
    assert (border[0, 0].cpu().numpy() == g[f"{name}.border"]).all()
    assert (full[0, 0].cpu().numpy() == g[f"{name}.full"]).all()


@pytest.mark.parametrize("radius", [0, 1, 3, 7])
def test_flat_morphology_radius_sweep(radius):
    g = load_golden("g12_mask_surgery")
    m = g["morph.mask"]                                   # [2, 1, 20, 27] in {0,1}
    lab = T(np.where(m[:, 0] > 0, 6, 0).astype(np.uint8)).to(DEV)     # skin = foreground, background = not
    content, border, full = ops.foreground_masks(lab, None, radius)
    assert (content.cpu().numpy() == m).all()
    assert (border.cpu().numpy() == g[f"morph.r{radius}.border"]).all()
    assert (full.cpu().numpy() == g[f"morph.r{radius}.full"]).all()


def test_swap_head_mask_batch_512_against_oracle():
    """Full-size maps (512^2, batch 5: blocky, iid, and a sample without eyes/brows/nose), bit-exact against the oracle."""
    bs = 5
    src = seeded.blocky_labels(31, bs, 12, 512, 16).astype(np.uint8)
    tgt = seeded.blocky_labels(32, bs, 12, 512, 16).astype(np.uint8)
    src[1] = seeded.iid_labels(33, 1, 12, 512)[0]
    tgt[2] = seeded.iid_labels(34, 1, 12, 512)[0]
    src[3][np.isin(src[3], (2, 3, 5))] = 6
    tgt[4][:] = 0
    res, hole, hole_map, lines = ops.swap_head_mask(T(src).to(DEV), T(tgt).to(DEV))
    content, border, full = ops.foreground_masks(res, hole, 5)
    for b in range(bs):
        o = O.swap_head_mask_hole_first(src[b], tgt[b])
        assert (res[b].cpu().numpy() == o[0]).all(), b
        assert (hole[b].cpu().numpy().astype(bool) == o[1]).all(), b
        assert (hole_map[b].cpu().numpy() == o[2]).all(), b
        assert lines[b].cpu().tolist() == [o[4], o[3]], b
        oc, ob, of = O.create_masks_expansion(O.foreground_mask(o[0], o[1])[None, None], 5)
        assert (content[b, 0].cpu().numpy() == oc[0, 0]).all() and (border[b, 0].cpu().numpy() == ob[0, 0]).all() and (full[b, 0].cpu().numpy() == of[0, 0]).all(), b


def test_swap_head_mask_argument_errors():
    a = torch.zeros(1, 8, 8, dtype=torch.uint8, device=DEV)
    with pytest.raises(ValueError):
        ops.swap_head_mask(a, torch.zeros(1, 8, 9, dtype=torch.uint8, device=DEV))
    with pytest.raises(ValueError):
        ops.swap_head_mask(a.float(), a)
    e = torch.zeros(0, 8, 8, dtype=torch.uint8, device=DEV)
    res, hole, hole_map, lines = ops.swap_head_mask(e, e)       # empty batch
    assert res.shape == (0, 8, 8) and lines.shape == (0, 2)


@pytest.mark.parametrize("shape", [(2, 3, 64, 48), (1, 1, 7, 5), (1, 2, 1, 9), (3, 1, 2, 2)])
def test_pyramid_steps_against_the_oracle(shape):
    """e4s_pyr_down / e4s_pyr_up against the numpy restatement of cv2.pyrDown / cv2.pyrUp (oracle; unpinned: no cv2 in this image):
    float and 8-bit rounding, odd sizes, one-pixel rows, and the fused Laplacian / reconstruction forms."""
    from e4s2024_amd import ops
    rs = np.random.RandomState(5)
    x8 = rs.randint(0, 256, shape).astype(np.uint8)
    xf = rs.rand(*shape).astype(np.float32) * 255
    hwc = lambda a: np.moveaxis(a.reshape((-1,) + a.shape[-2:]), 0, -1)          # planes -> [H, W, planes]
    back = lambda a, like: np.moveaxis(a, -1, 0).reshape(like.shape[:-2] + a.shape[:2])
    d8 = ops.pyr_down(torch.from_numpy(x8).float().to(DEV), round_u8=True).cpu().numpy()
    assert np.array_equal(d8, back(O.pyr_down(hwc(x8)), x8).astype(np.float32))
    df = ops.pyr_down(torch.from_numpy(xf).to(DEV)).cpu().numpy()
    assert np.abs(df - back(O.pyr_down(hwc(xf)), xf)).max() <= 1e-4
    up = ops.pyr_up(torch.from_numpy(xf).to(DEV)).cpu().numpy()
    ref = back(O.pyr_up(hwc(xf)), xf)
    assert up.shape == ref.shape and np.abs(up - ref).max() <= 1e-4
    other = rs.rand(*up.shape).astype(np.float32)
    t = torch.from_numpy(other).to(DEV)
    assert np.abs(ops.pyr_up(torch.from_numpy(xf).to(DEV), minuend=t).cpu().numpy() - (other - ref)).max() <= 1e-4
    assert np.abs(ops.pyr_up(torch.from_numpy(xf).to(DEV), addend=t).cpu().numpy() - (ref + other)).max() <= 1e-4


def test_multi_band_blend_1024_against_the_oracle():
    """``blending`` (multi_band_blending.py:51-74) at its call-site size: the ten-level blend of a uint8 frame A and a float frame B under a
    soft border mask, against the oracle's float64 restatement — equal up to one grey level where fp32 vs fp64 lands on the other side of
    the final truncation; mask 1 returns A exactly, mask 0 returns trunc(B)."""
    from e4s2024_amd import ops
    rs = np.random.RandomState(11)
    a = rs.randint(0, 256, (1024, 1024, 3)).astype(np.uint8)
    b = (rs.rand(1024, 1024, 3) * 255).astype(np.float64)
    yy, xx = np.mgrid[0:1024, 0:1024]
    m = np.clip(1.5 - np.hypot(yy - 500, xx - 540) / 200.0, 0, 1).astype(np.float32)[..., None].repeat(3, -1)
    ref = O.blending(a, b, m)
    chw = lambda t: torch.from_numpy(np.ascontiguousarray(np.moveaxis(t, -1, 0)))[None].to(DEV)
    out = ops.blending(chw(a), chw(b).float(), chw(m)).cpu().numpy()[0]
    diff = np.abs(out.astype(np.int32) - np.moveaxis(ref, -1, 0).astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() <= 2e-3, (diff.max(), (diff > 0).mean())
    ones, zeros = torch.ones(1, 1, 1024, 1024, device=DEV), torch.zeros(1, 1, 1024, 1024, device=DEV)
    assert torch.equal(ops.blending(chw(a), chw(b).float(), ones), chw(a))
    tb = ops.blending(chw(a), chw(b).float(), zeros).cpu().numpy()[0].astype(np.int32)
    assert np.abs(tb - np.moveaxis(np.floor(b), -1, 0)).max() <= 1


def test_paste_back_chain_against_the_oracle():
    """pipeline.paste_back = reference :447, :464-473 — the swapped face softened through Pillow's 512 x 512 round trip, masks resized bilinearly (align_corners=False), alpha paste, ten-level multi-band blend —
    against the same chain on the CPU (torch interpolate + the oracle's blend)."""
    import torch.nn.functional as F
    from e4s2024_amd import pipeline
    rs = np.random.RandomState(21)
    sw = rs.randint(0, 256, (1, 1024, 1024, 3)).astype(np.uint8)
    tg = rs.randint(0, 256, (1, 1024, 1024, 3)).astype(np.uint8)
    lab = T(seeded.blocky_labels(5, 1, 12, 512, 16))
    content, border, _ = ops.foreground_masks(lab.to(DEV), None, 5)
    out = pipeline.paste_back(T(sw).to(DEV), T(tg).to(DEV), content, border).cpu().numpy()[0]
    PIL = pytest.importorskip("PIL.Image")
    sw = np.array(PIL.fromarray(sw[0]).resize((512, 512)).resize((1024, 1024)))[None]          # :447, Pillow itself
    cm = F.interpolate(content.cpu(), (1024, 1024), mode="bilinear", align_corners=False)[0, 0, :, :, None].numpy()
    bm = F.interpolate(border.cpu(), (1024, 1024), mode="bilinear", align_corners=False)[0, 0, :, :, None].numpy().repeat(3, -1)
    pasted = sw[0] * cm + tg[0] * (1 - cm)
    ref = O.blending(tg[0], pasted, bm)
    diff = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    assert out.shape == ref.shape and diff.max() <= 1 and (diff > 0).mean() <= 2e-3, (diff.max(), (diff > 0).mean())


def test_pil_resize_on_the_device_equals_pillow_bit_for_bit():
    """ops.pil_resize (e4s_resample_u8) against Pillow itself: the reference's softening round trip 1024 -> 512 -> 1024 of the swapped face
    (face_swap_video_pipeline.py:447) and ragged up / down sizes, batch 2."""
    PIL = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(4)
    a = rs.randint(0, 256, (2, 1024, 1024, 3)).astype(np.uint8)
    out = ops.pil_resize(ops.pil_resize(T(a).to(DEV), (512, 512)), (1024, 1024)).cpu().numpy()
    for b in range(2):
        assert np.array_equal(out[b], np.array(PIL.fromarray(a[b]).resize((512, 512)).resize((1024, 1024))))
    for (h, w), (ow, oh) in [((37, 53), (20, 31)), ((50, 41), (123, 77)), ((9, 7), (3, 2)), ((64, 48), (64, 24))]:
        x = rs.randint(0, 256, (1, h, w, 3)).astype(np.uint8)
        assert np.array_equal(ops.pil_resize(T(x).to(DEV), (ow, oh)).cpu().numpy()[0], np.array(PIL.fromarray(x[0]).resize((ow, oh))))


@pytest.mark.parametrize("radius", [0, 1, 3, 5])
def test_erode_labels_matches_the_oracle(radius):
    """f1: erode_mask of the PTI loop (training/video_swap_ft_coach.py:64-93) on the device, ragged sizes, a batch, exact."""
    rs = np.random.RandomState(11 + radius)
    lab = np.repeat(np.repeat(rs.randint(0, 12, (3, 14, 19)), 5, 1), 5, 2)[:, :67, :93].astype(np.uint8)
    out = ops.erode_labels(T(lab).to(DEV), radius).cpu().numpy()
    for b in range(3):
        assert np.array_equal(out[b], O.erode_mask(lab[b], radius)), (radius, b)
    full = seeded.blocky_labels(7, 2, 12, 512, 16)
    assert np.array_equal(ops.erode_labels(T(full).to(DEV), 3).cpu().numpy()[1], O.erode_mask(full[1], 3))


def test_frames_to_tensor_bit_exact():
    """f3: ToTensor + Normalize(.5, .5) of uint8 frames on the device == the float32 expression torchvision evaluates."""
    rs = np.random.RandomState(5)
    fr = rs.randint(0, 256, (2, 37, 53, 3)).astype(np.uint8)
    fr[0, 0, :, 0] = np.arange(53) * 4 % 256
    got = ops.frames_to_tensor(T(fr).to(DEV)).cpu()
    assert torch.equal(got, O.frames_to_tensor(fr))
    with pytest.raises(ValueError):
        ops.frames_to_tensor(torch.zeros(1, 3, 4, 4, device=DEV))
