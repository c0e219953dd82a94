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


@pytest.fixture(scope="session")
def golden_train():
    """Loss + parameter-gradient digests of the reference's own p_losses(...).backward()
    (tests/golden/make_golden_train.py)."""
    return load_golden("train.pt")


@pytest.fixture(scope="session")
def golden_r4():
    """Round-4 fixtures from the reference (tests/golden/make_golden_r4.py): the learned / random sinusoidal U-Net forward."""
    return load_golden("r4.pt")


@pytest.fixture(scope="session")
def golden_hybrid():
    """The hybrid (KL) loss branch of p_losses and the ddpm=False loss weights, from the reference
    (tests/golden/make_golden_hybrid.py)."""
    return load_golden("hybrid.pt")


@pytest.fixture(scope="session")
def golden_train_noise():
    """The reference's p_losses(...).backward() with offset noise / with the immiscible noise assignment
    (tests/golden/make_golden_train_noise.py)."""
    return load_golden("train_noise.pt")


@pytest.fixture(scope="session")
def golden_guided():
    """The reference's model_predictions / p_mean_variance / q_posterior / predict_* with per-sample timesteps and its
    ddim_sample_guided (tests/golden/make_golden_guided.py)."""
    return load_golden("guided.pt")


def check_grad_digest(name: str, grad: torch.Tensor, dg: dict, tol: float):
    """A gradient against its golden digest: norm, 8 random projections, the first elements, the whole tensor if small.
    Every check is relative to the golden gradient's norm (a projection of a vector of norm n on a unit-variance random
    direction is ~n, so an error of tol*n in the vector moves it by ~tol*n)."""
    from oracle.train_oracle import directions

    flat = grad.detach().double().cpu().reshape(-1)
    n = max(dg["norm"], 1e-30)
    assert abs(float(flat.norm()) - dg["norm"]) <= tol * n, (name, float(flat.norm()), dg["norm"])
    proj = directions(name, flat.numel()) @ flat
    err = float((proj - dg["proj"]).abs().max())
    assert err <= 4 * tol * n, (name, "projection", err / n)
    head = dg["head"].double()
    scale = max(float(head.norm()), n * (head.numel() / flat.numel()) ** 0.5)
    assert float((flat[: head.numel()] - head).norm()) <= 4 * tol * scale, (name, "head")
    if "full" in dg:
        assert float((flat - dg["full"].double().reshape(-1)).norm()) <= tol * n, (name, "full tensor")
