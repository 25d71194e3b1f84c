"""hipGraph capture of the hot path: replay == eager, and a second replay with new inputs == eager on those inputs."""
import pytest
import torch

from conftest import install_dropin
from e4s2024_amd import seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def parser(bisenet_sd):
    install_dropin()
    from swap_face_fine.face_parsing.face_parsing_demo import FaceParser
    p = FaceParser(seg_ckpt=None, device=DEV)
    p.seg.load_state_dict(bisenet_sd)
    p.seg.eval()
    return p


def test_graphed_gen_img_matches_eager(gpu_net3):
    from e4s2024_amd import graphs
    la = seeded.seeded_latent_avg(2, 18)
    codes = seeded.seeded_codes(1, 1, 12, 18, la).to(DEV)
    lab = torch.from_numpy(seeded.blocky_labels(3, 1, 12, 512, 16)).to(DEV)
    g = graphs.graphed_gen_img(gpu_net3, codes, lab)
    with torch.no_grad():
        ref = gpu_net3.gen_img(None, codes, lab, randomize_noise=False)[0]
    out = g(codes, lab)
    assert torch.equal(out, ref)
    codes2 = seeded.seeded_codes(7, 1, 12, 18, la).to(DEV)
    lab2 = torch.from_numpy(seeded.iid_labels(9, 1, 12, 512)).to(DEV)
    with torch.no_grad():
        ref2 = gpu_net3.gen_img(None, codes2, lab2, randomize_noise=False)[0]
    assert torch.equal(g(codes2, lab2), ref2)
    with pytest.raises(ValueError):
        g(codes2.expand(2, -1, -1, -1).contiguous(), lab2)


def test_graphed_full_swap_matches_eager(gpu_net3, parser):
    from e4s2024_amd import graphs, pipeline
    d = seeded.seeded_image(5, 1, 1024).to(DEV)
    t = seeded.seeded_image(6, 1, 1024).to(DEV)
    g = graphs.graphed_swap(gpu_net3, parser, d, t)
    ref, lab = pipeline.swap_batch(gpu_net3, parser, d, t)
    out, glab = g(d, t)
    assert torch.equal(glab, lab) and torch.equal(out, ref)
    assert out.dtype == torch.uint8 and tuple(out.shape) == (1, 1024, 1024, 3)
    d2 = seeded.seeded_image(8, 1, 1024).to(DEV)
    ref2, _ = pipeline.swap_batch(gpu_net3, parser, d2, t)
    assert torch.equal(g(d2, t)[0], ref2)


def test_two_threads_two_streams_match_the_serial_run(gpu_net3):
    """Host-state hygiene: the Python layer keeps its state per HIP stream, so two host threads that each drive their own
    stream through the same network — different codes, different masks, 4 x 4 .. 32 x 32 layers on split-K workspaces, style-table plans,
    region-map caches all in flight at once — produce exactly the bits of the two calls run one after the other."""
    import threading
    from e4s2024_amd import seeded
    la = seeded.seeded_latent_avg(2, 18)
    jobs = []
    for k in range(2):
        codes = seeded.seeded_codes(61 + k, 2, 12, 18, la).to(DEV)
        lab = torch.from_numpy(seeded.blocky_labels(71 + k, 2, 12, 512, 16 if k == 0 else 64)).to(DEV).to(torch.uint8)
        jobs.append((codes, lab))
    with torch.no_grad():
        serial = [gpu_net3.gen_img(None, c, m, randomize_noise=False)[0].clone() for c, m in jobs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in jobs]
    results = [[None] * 6 for _ in jobs]
    errors = []
    gate = threading.Barrier(len(jobs))

    def work(i):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(streams[i]), torch.no_grad():
                gate.wait()
                for r in range(6):
                    results[i][r] = gpu_net3.gen_img(None, jobs[i][0], jobs[i][1], randomize_noise=False)[0]
            streams[i].synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for i in range(len(jobs)):
        for r in range(6):
            assert torch.equal(results[i][r], serial[i]), f"thread {i}, repetition {r}: differs from the serial run"
