import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def template_from_manifest(man):
    """manifest entry {key: [shape, dtype]} -> {key: meta tensor} usable by seeded_state_dict."""
    return {k: torch.empty(tuple(s), dtype=getattr(torch, d), device="meta") for k, (s, d) in man.items()}


@pytest.fixture(scope="session")
def net3_sd(manifest):
    """Seeded full-size Net3 state_dict (seed 4) on CPU — shared by the oracle tests."""
    from e4s2024_amd import seeded
    return seeded.seeded_state_dict(template_from_manifest(manifest["net3_1024_rli13"]), 4, "net3")


@pytest.fixture(scope="session")
def bisenet_sd(manifest):
    from e4s2024_amd import seeded
    return seeded.seeded_state_dict(template_from_manifest(manifest["bisenet_19"]), 7, "bisenet")


# ------------------------------------------------------------------------------- GPU fixtures
def install_dropin():
    import e4s2024_amd
    e4s2024_amd.install()


def default_opts(**kw):
    import argparse
    d = dict(fsencoder_type="psp", remaining_layer_idx=13, num_seg_cls=12, out_size=1024, train_G=False,
             start_from_latent_avg=True, learn_in_w=False)
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.fixture(scope="session")
def gpu_net3(net3_sd):
    """The MI355X Net3 (drop-in) with the seeded weights, on cuda:0."""
    install_dropin()
    from models.networks import Net3
    from e4s2024_amd import seeded
    net = Net3(default_opts()).eval()
    net.load_state_dict(net3_sd)
    net.latent_avg = seeded.seeded_latent_avg(2, 18)
    net = net.to("cuda:0")
    net.latent_avg = net.latent_avg.to("cuda:0")
    return net
