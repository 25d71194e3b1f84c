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
