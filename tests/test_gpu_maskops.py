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
