import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # -m gpu tests are skipped (not failed) when collected without a GPU and without -m selection
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def golden_blocks():
    return load_golden("blocks.pt")


@pytest.fixture(scope="session")
def golden_samplers():
    return load_golden("samplers.pt")


@pytest.fixture(scope="session")
def golden_vae():
    return load_golden("vae.pt")


@pytest.fixture(scope="session")
def golden_schedule():
    return load_golden("schedule.pt")


@pytest.fixture(scope="session")
def golden_imgcond():
    """Image-conditional variant (tests/golden/make_golden_imgcond.py)."""
    return load_golden("imgcond.pt")


@pytest.fixture(scope="session")
def golden_encoder():
    """VAE Encoder outputs of the reference (tests/golden/make_golden_encoder.py)."""
    return load_golden("encoder.pt")



@pytest.fixture(scope="session")
def golden_configs():
    """BASELINE configs 3 / 4 / 5 as workloads, from the reference (tests/golden/make_golden_configs.py)."""
    return load_golden("configs.pt")


@pytest.fixture(scope="session")
def golden_r3():
    """Round-3 fixtures from the reference (tests/golden/make_golden_r3.py): the LDM YAML shapes (3x16x16 / 3x8x8 latents,
    the ch_mult (1,2,4,8) VQModel), objectives pred_x0 / pred_v, self-conditioning."""
    return load_golden("r3.pt")
