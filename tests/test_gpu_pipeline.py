"""GPU parity of the batched full-swap unit, BASELINE configs[2]: ``pipeline.swap_batch`` at batch 8 — two HIP
streams, half-batch chains, ``torch.cat`` of halves, style-vector mix — against the CPU oracle chain

    parse (x2) -> get_style_vectors (x2) -> swap_comp_style_vector -> cal_style_codes -> generator_forward -> tensor2im

(reference: face_swap_video_pipeline.py:212-219, 332-354, 429-443) on faces of the batch that fall into different half-batch chains,
plus run-to-run bit identity (a stream race would show as a changing result), and the style-vector mix against the reference-generated
golden g13."""
import numpy as np
import pytest
import torch

from conftest import load_golden, install_dropin, record_parity
from e4s2024_amd import ops, pipeline, seeded
from oracle import e4s_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
BS = 8
CHECK_FACES = (1, 6)          # one face of each half batch (SWAP_CHAINS=4 runs faces 0-3 and 4-7 on different streams)
PIXEL_TOL = 1e-3              # north_star: <= 1e-3 max-abs fp32 on generated pixels


@pytest.fixture(scope="module")
def parser(bisenet_sd):
    install_dropin()
    from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
    p = FaceParser(seg_ckpt=None, device=DEV)
    p.seg.load_state_dict(bisenet_sd)
    p.seg.eval()
    return p


@pytest.fixture(scope="module")
def faces():
    """Driven / target batches: half per-pixel noise, half smooth structure (the parser then produces large regions as well as speckle)."""
    def batch(seed):
        raw = seeded.seeded_image(seed, BS, 1024)
        smooth = (torch.nn.functional.avg_pool2d((raw + 1) / 2, 31, 1, 15) * 3 - 1).clamp(0, 1) * 2 - 1
        raw[1::2] = smooth[1::2]
        return raw.contiguous()
    return batch(5), batch(6)


@pytest.fixture(scope="module")
def oracle_chain(faces, net3_sd, bisenet_sd):
    """The CPU oracle's full swap for the checked faces: labels, style vectors, codes, float image."""
    drv, tgt = faces
    la = seeded.seeded_latent_avg(2, 18)
    out = {}
    for b in CHECK_FACES:
        ent = {}
        for name, img in (("d", drv[b:b + 1]), ("t", tgt[b:b + 1])):
            logits = O.bisenet_forward(bisenet_sd, O.parser_preprocess((img + 1) / 2))
            top2 = torch.topk(logits[0], 2, dim=0).values
            ent["gap_" + name] = ((top2[0] - top2[1]) / logits.abs().max()).numpy()
            ent["lab19_" + name] = torch.argmax(logits, 1)[0].numpy().astype(np.uint8)
            ent["lab_" + name] = O.remap_19_to_12(ent["lab19_" + name])
        ent["img_d"], ent["img_t"] = drv[b:b + 1], tgt[b:b + 1]
        out[b] = ent
    out["latent_avg"] = la
    return out


_rest_cache = {}


def _oracle_rest(ent, lab_d, lab_t, net3_sd, la, comp_indices=pipeline.DEFAULT_COMP_INDICES, below=False):
    """Everything after the parse, from given 12-class maps (uint8 [512, 512]); memoised on the maps (every mode of the batched call must
    arrive at the same ones, so the ~6 s oracle chain of a face runs once)."""
    key = (id(ent), lab_d.tobytes(), lab_t.tobytes(), tuple(comp_indices), below)
    if key not in _rest_cache:
        _rest_cache[key] = _oracle_rest_uncached(ent, lab_d, lab_t, net3_sd, la, comp_indices, below)
    return _rest_cache[key]


def _oracle_rest_uncached(ent, lab_d, lab_t, net3_sd, la, comp_indices, below):
    m_d = O.label_map_to_onehot(T(lab_d.astype(np.int64))[None, None], 12)
    m_t = O.label_map_to_onehot(T(lab_t.astype(np.int64))[None, None], 12)
    v_d, _ = O.get_style_vectors(net3_sd, ent["img_d"], m_d)
    v_t, _ = O.get_style_vectors(net3_sd, ent["img_t"], m_t)
    mixed = O.swap_comp_style_vector(v_t, v_d, comp_indices, below)
    codes = O.cal_style_codes(net3_sd, mixed, la, 13)
    img, _ = O.generator_forward(net3_sd, codes, m_t, None)
    return v_d, v_t, mixed, codes, img


def _check_labels(tag, got, ent, which):
    """12-class map of the device parser vs the oracle's: equal, except on pixels where the oracle's own top-2 logits tie to within fp32
    re-association noise (the inputs here are seeded noise, not the reference's golden image — that one must match exactly, test_gpu_parser)."""
    ref = ent["lab_" + which]
    bad = got != ref
    n = int(bad.sum())
    record_parity(f"swap_bs8.{tag}.label_flips_vs_oracle", n, 16, "12-class map, 262144 px")
    assert n <= 16 and (n == 0 or ent["gap_" + which][bad].max() < 1e-5), f"{tag}: {n} label pixels differ from the oracle"
    return n


@pytest.mark.parametrize("mode", ["batched", "one_stream", "two_streams", "four_chains"])
def test_swap_batch_bs8_vs_oracle(gpu_net3, parser, faces, oracle_chain, net3_sd, mode):
    drv, tgt = faces[0].to(DEV), faces[1].to(DEV)
    old = pipeline.SWAP_CHAINS
    pipeline.SWAP_CHAINS = 4 if mode == "four_chains" else 2
    # "batched" = the default route (driven and target as one batch of 16 through parser and encoder)
    kw = {"batched": True} if mode == "batched" else {"batched": False, "two_streams": mode != "one_stream"}
    try:
        runs = []
        for _ in range(3):            # race detection: the same call three times must give the same bits
            img, lab = pipeline.swap_batch(gpu_net3, parser, drv, tgt, to_uint8=False, **kw)
            runs.append((img.clone(), lab.clone()))
        frames, lab_u8 = pipeline.swap_batch(gpu_net3, parser, drv, tgt, to_uint8=True, **kw)
        torch.cuda.synchronize()
    finally:
        pipeline.SWAP_CHAINS = old
    for img, lab in runs[1:]:
        assert torch.equal(img, runs[0][0]) and torch.equal(lab, runs[0][1]), f"{mode}: swap_batch is not run-to-run bit-identical (stream race?)"
    img, lab = runs[0]
    assert tuple(img.shape) == (BS, 3, 1024, 1024) and tuple(lab.shape) == (BS, 512, 512) and lab.dtype == torch.uint8
    assert frames.dtype == torch.uint8 and tuple(frames.shape) == (BS, 1024, 1024, 3) and torch.equal(lab_u8, lab)
    assert np.array_equal(frames.cpu().numpy(), np.stack([O.tensor2im_array(img[b].cpu()) for b in range(BS)]))   # device tensor2im == reference arithmetic
    la = oracle_chain["latent_avg"]
    with torch.no_grad():
        lab_d_gpu = parser.parse_batch((drv + 1) / 2, seg12=True).cpu().numpy()
    for b in CHECK_FACES:
        ent = oracle_chain[b]
        got_t = lab[b].cpu().numpy()
        _check_labels(f"{mode}.face{b}.target", got_t, ent, "t")
        _check_labels(f"{mode}.face{b}.driven", lab_d_gpu[b], ent, "d")
        # the rest of the chain from the maps the device used (identical to the oracle's unless a true tie flipped above)
        _, _, _, _, ref = _oracle_rest(ent, lab_d_gpu[b], got_t, net3_sd, la)
        d = (img[b].cpu() - ref[0]).abs().max().item()
        record_parity(f"swap_bs8.{mode}.face{b}.pixels_vs_oracle", d, PIXEL_TOL)
        assert d <= PIXEL_TOL, f"{mode}: face {b} of the batch is {d:.3e} from the oracle chain"
        grey = np.abs(frames[b].cpu().numpy().astype(np.int16) - O.tensor2im_array(ref[0]).astype(np.int16)).max()
        record_parity(f"swap_bs8.{mode}.face{b}.frame_grey_levels_vs_oracle", int(grey), 1)
        assert grey <= 1


def test_swap_batch_with_mask_surgery_bs8_vs_oracle(gpu_net3, parser, faces, oracle_chain, net3_sd):
    """The clip's unit of work (BASELINE configs[4], what ``bench.py --clip`` times): ``swap_batch(mask_surgery=True)`` at batch 8 — the synthesis is
    driven by ``swap_head_mask_hole_first(driven map, target map)`` (face_swap_video_pipeline.py:420, 429-443) — against the oracle chain
    parse x2 -> swap_head_mask_hole_first -> get_style_vectors x2 (on the faces' OWN maps, :332-354) -> mix -> cal_style_codes -> generator_forward
    on the SWAPPED map, for one face of each half batch: region map exact, pixels <= 1e-3, frames <= 1 grey level; three runs bit-identical."""
    drv, tgt = faces[0].to(DEV), faces[1].to(DEV)
    runs = []
    for _ in range(3):
        img, lab, extra = pipeline.swap_batch(gpu_net3, parser, drv, tgt, to_uint8=False, mask_surgery=True)
        runs.append((img.clone(), lab.clone(), extra["hole_mask"].clone()))
    frames, lab_u8, _ = pipeline.swap_batch(gpu_net3, parser, drv, tgt, to_uint8=True, mask_surgery=True)
    torch.cuda.synchronize()
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r, runs[0])), "swap_batch(mask_surgery=True) is not run-to-run bit-identical"
    img, lab, hole = runs[0]
    assert torch.equal(lab_u8, lab) and tuple(lab.shape) == (BS, 512, 512)
    la = oracle_chain["latent_avg"]
    with torch.no_grad():
        both = parser.parse_batch((torch.cat([drv, tgt]) + 1) / 2, seg12=True).cpu().numpy()
    lab_d_gpu, lab_t_gpu = both[:BS], both[BS:]
    for b in CHECK_FACES:
        ent = oracle_chain[b]
        _check_labels(f"surgery.face{b}.target", lab_t_gpu[b], ent, "t")
        _check_labels(f"surgery.face{b}.driven", lab_d_gpu[b], ent, "d")
        res, hole_o, _, _, _ = O.swap_head_mask_hole_first(lab_d_gpu[b], lab_t_gpu[b])          # from the maps the device parsed (ties aside: the oracle's)
        assert np.array_equal(lab[b].cpu().numpy(), res.astype(np.uint8)), f"face {b}: swapped region map differs from the oracle's"
        assert np.array_equal(hole[b].cpu().numpy().astype(bool).reshape(res.shape), np.asarray(hole_o).astype(bool))
        v_d, v_t, mixed, codes, _ = _oracle_rest(ent, lab_d_gpu[b], lab_t_gpu[b], net3_sd, la)   # style vectors from each face's own map
        ref, _ = O.generator_forward(net3_sd, codes, O.label_map_to_onehot(T(res.astype(np.int64))[None, None], 12), None)
        d = (img[b].cpu() - ref[0]).abs().max().item()
        record_parity(f"swap_bs8.mask_surgery.face{b}.pixels_vs_oracle", d, PIXEL_TOL)
        assert d <= PIXEL_TOL, f"face {b} of the batch is {d:.3e} from the oracle chain"
        grey = np.abs(frames[b].cpu().numpy().astype(np.int16) - O.tensor2im_array(ref[0]).astype(np.int16)).max()
        record_parity(f"swap_bs8.mask_surgery.face{b}.frame_grey_levels_vs_oracle", int(grey), 1)
        assert grey <= 1


def test_two_stream_stress_of_the_default_routes_is_bit_stable(gpu_net3, parser, faces):
    """Concurrency canary (bounded, ~15 s): the DEFAULT encoder route (direct + Winograd convolutions, as ``ops.winograd_route`` picks them) and the
    masked synthesis kernel, each hammered on one stream while the other stream runs the other — every result of every round must equal the
    single-stream result bit for bit.  A kernel that loses a wait under concurrency (round 2's first pre-split Winograd transform did, with a
    second stream beside it and never alone) shows up here as a changing value."""
    import time
    drv, tgt = faces[0].to(DEV), faces[1].to(DEV)
    net = gpu_net3
    la = net.latent_avg.cpu()
    codes = seeded.seeded_codes(91, 4, 12, 18, la).to(DEV)
    lab_g = torch.from_numpy(seeded.iid_labels(92, 4, 12, 512)).to(DEV).to(torch.uint8)
    small = [drv[:1].contiguous(), drv[:2].contiguous(), tgt[:4].contiguous()]        # batches 1, 2, 4: the Winograd route's range
    with torch.no_grad():
        labs = [parser.parse_batch((x + 1) / 2, seg12=True) for x in small]
        ref_vec = [net.get_style_vectors(x, l)[0].clone() for x, l in zip(small, labs)]
        ref_img = net.gen_img(None, codes, lab_g, randomize_noise=False)[0].clone()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        t0, rounds, bad = time.time(), 0, []
        while time.time() - t0 < 12.0 and rounds < 40:
            outs_v, outs_i = [], []
            with torch.cuda.stream(s1):
                for x, l in zip(small, labs):
                    outs_v.append(net.get_style_vectors(x, l)[0])
            with torch.cuda.stream(s2):
                for _ in range(2):
                    outs_i.append(net.gen_img(None, codes, lab_g, randomize_noise=False)[0])
            torch.cuda.synchronize()
            for k, (o, r) in enumerate(zip(outs_v, ref_vec)):
                if not torch.equal(o, r):
                    bad.append(("style_vectors", rounds, k, (o - r).abs().max().item()))
            for k, o in enumerate(outs_i):
                if not torch.equal(o, ref_img):
                    bad.append(("gen_img", rounds, k, (o - ref_img).abs().max().item()))
            rounds += 1
    record_parity("two_stream_stress.rounds", rounds, None, "encoder (batches 1, 2, 4) on one stream beside masked synthesis (i.i.d. maps, batch 4) on another")
    record_parity("two_stream_stress.mismatches", len(bad), 0)
    assert rounds >= 5 and not bad, bad[:5]


def test_swap_batch_stages_vs_oracle(gpu_net3, parser, faces, oracle_chain, net3_sd):
    """The intermediate tensors of the same chain on one checked face: style vectors of both faces, the mix, the codes."""
    drv, tgt = faces[0].to(DEV), faces[1].to(DEV)
    b = CHECK_FACES[0]
    ent = oracle_chain[b]
    with torch.no_grad():
        lab_d = parser.parse_batch((drv + 1) / 2, seg12=True)
        lab_t = parser.parse_batch((tgt + 1) / 2, seg12=True)
        vec_d, _ = gpu_net3.get_style_vectors(drv, lab_d)
        vec_t, _ = gpu_net3.get_style_vectors(tgt, lab_t)
        mixed = pipeline.mix_style_vectors(vec_t, vec_d)
        codes = gpu_net3.cal_style_codes(mixed)
    v_d, v_t, o_mixed, o_codes, _ = _oracle_rest(ent, lab_d[b].cpu().numpy(), lab_t[b].cpu().numpy(), net3_sd, oracle_chain["latent_avg"])
    for name, got, ref, tol in (("vec_driven", vec_d[b], v_d[0], 1e-3), ("vec_target", vec_t[b], v_t[0], 1e-3), ("mixed", mixed[b], o_mixed[0], 1e-3),
                                ("codes", codes[b], o_codes[0], 1e-3)):
        d = (got.cpu() - ref).abs().max().item()
        record_parity(f"swap_bs8.stage.{name}_vs_oracle", d, tol)
        assert d <= tol, name


# ---------------------------------------------------------------------------------------------- style-vector mix (row f2)
G13_CASES = ("default", "no_teeth", "teeth_cancel", "below_face", "below_no_teeth", "no_indices", "all_indices")


@pytest.mark.parametrize("name", G13_CASES)
def test_mix_style_vectors_golden(name):
    """``pipeline.mix_style_vectors`` on the device == outputs of the reference's own ``swap_comp_style_vector`` (bit-exact: the mix only
    copies and halves sums of two floats)."""
    g = load_golden("g13_style_mix")
    t, s = T(g[f"{name}.target"]).to(DEV), T(g[f"{name}.source"]).to(DEV)
    out = pipeline.mix_style_vectors(t, s, tuple(int(i) for i in g[f"{name}.idx"]), bool(g[f"{name}.below"]))
    assert torch.equal(out.cpu(), T(g[f"{name}.out"])), name
    assert torch.equal(t.cpu(), T(g[f"{name}.target"])) and torch.equal(s.cpu(), T(g[f"{name}.source"]))     # inputs untouched (the reference deep-copies)


@pytest.mark.parametrize("below", [False, True])
def test_mix_style_vectors_batch_golden(below):
    g = load_golden("g13_style_mix")
    t, s = T(g["batch.target"]).to(DEV), T(g["batch.source"]).to(DEV)
    out = pipeline.mix_style_vectors(t, s, tuple(int(i) for i in g["batch.idx"]), below)
    assert torch.equal(out.cpu(), T(g[f"batch.out_below{int(below)}"]))


def test_mix_style_vectors_inside_graph_capture():
    """No host synchronisation inside the mix (it runs inside the captured full swap): capture + replay gives the eager result."""
    g = load_golden("g13_style_mix")
    t, s = T(g["batch.target"]).to(DEV), T(g["batch.source"]).to(DEV)
    eager = pipeline.mix_style_vectors(t, s)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pipeline.mix_style_vectors(t, s)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = pipeline.mix_style_vectors(t, s)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_clip_from_reference_style_directory_through_the_swap(gpu_net3, parser, tmp_path):
    """Row f4 on the device: a directory written the reference's way (RGB PNG crops, uint8 label PNGs) -> ``handoff.load`` onto the GPU ->
    the batched swap.  The loader's image normalisation equals the device ingest kernel's (``ops.frames_to_tensor``, the ToTensor +
    Normalize of face_swap_video_pipeline.py:338-339) bit for bit, and the swap on the loaded clip equals the swap on the same tensors
    built directly."""
    from PIL import Image
    from e4s2024_amd import handoff
    rs = np.random.RandomState(2)
    os_path = str(tmp_path)
    import os
    os.makedirs(os.path.join(os_path, "imgs"))
    u8 = {}
    for t in "DT":
        u8[t] = rs.randint(0, 256, (2, 1024, 1024, 3)).astype(np.uint8)
        for i in range(2):
            Image.fromarray(u8[t][i]).save(os.path.join(os_path, "imgs", "%s_%04d.png" % (t, i)))
    clip = handoff.load(os_path, device=DEV)
    assert clip.driven_mask is None and clip.target.is_cuda
    for k, t in (("driven", "D"), ("target", "T")):
        assert torch.equal(getattr(clip, k), ops.frames_to_tensor(T(u8[t]).to(DEV)))
    a, lab_a = pipeline.swap_batch(gpu_net3, parser, clip.driven, clip.target)
    b, lab_b = pipeline.swap_batch(gpu_net3, parser, ops.frames_to_tensor(T(u8["D"]).to(DEV)), ops.frames_to_tensor(T(u8["T"]).to(DEV)))
    assert torch.equal(a, b) and torch.equal(lab_a, lab_b) and a.dtype == torch.uint8 and tuple(a.shape) == (2, 1024, 1024, 3)


def test_batches_on_alternating_streams_equal_one_stream(gpu_net3):
    """``runner.StreamPipeline``: six ``gen_img`` batches with DIFFERENT codes and region maps on two and on three alternating streams — every
    image bit-identical to the same call on one stream (a race on the per-stream host state, the shared prepared weights or the region-map
    cache would mix batches up), twice in a row."""
    from e4s2024_amd.runner import StreamPipeline
    net = gpu_net3
    la = net.latent_avg.cpu()
    batches = []
    for i in range(6):
        codes = seeded.seeded_codes(70 + i, 2, 12, 18, la).to(DEV)
        lab = seeded.blocky_labels(80 + i, 2, 12, 512, 16 if i % 2 else 4)
        batches.append((codes, seeded.labels_to_onehot(lab, 12).to(DEV)))
    strict = ops.STRICT_MASK
    ops.STRICT_MASK = False                      # (the one-hot check is a host sync: it would serialise the streams)
    try:
        with torch.no_grad():
            ref = [net.gen_img(None, c, m, randomize_noise=False)[0] for c, m in batches]
            for n in (2, 3, 2):
                with StreamPipeline(n, device=DEV) as sp:
                    outs = [sp.submit(net.gen_img, None, c, m, randomize_noise=False)[0] for c, m in batches]
                torch.cuda.synchronize()
                for i, (o, r) in enumerate(zip(outs, ref)):
                    assert torch.equal(o, r), (n, i, (o - r).abs().max().item())
                del outs
    finally:
        ops.STRICT_MASK = strict
    assert not torch.equal(ref[0], ref[1])
