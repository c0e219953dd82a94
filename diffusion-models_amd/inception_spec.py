"""Static description of InceptionV3 as the reference's evaluators use it (host side, no GPU).

The reference does not contain the network: `FIDEvaluation` takes it from ``pytorch_fid.inception.InceptionV3``
(denoising_diffusion/fid_evaluation.py:8,38) and `InceptionScoreEvaluation` from
``torchvision.models.inception_v3`` (denoising_diffusion/inception_score_evaluation.py:5,41).  Both are pip
dependencies that are neither in the reference tree nor in this image (environment.yml pins torchvision 0.17.2 and
pytorch-fid 0.3.0), and their pretrained weights cannot be fetched.  What follows restates the PUBLISHED architecture
(torchvision/models/inception.py; pytorch_fid/inception.py for the three FID deviations) -- **parity unpinned**:
nothing in the reference's files can confirm it, the tests check the HIP path against the oracle's independent
restatement of the same description.

Parameter names are torchvision's (``Mixed_5b.branch1x1.conv.weight``, ``...bn.running_mean``), which is also what
pytorch_fid's weight file uses.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

# (name, cin, cout, (kh, kw), stride, (ph, pw)) of every BasicConv2d = conv(bias=False) -> BatchNorm(eps=1e-3) -> ReLU
Conv = Tuple[str, int, int, Tuple[int, int], int, Tuple[int, int]]
BN_EPS = 1e-3

STEM: List[Conv] = [
    ("Conv2d_1a_3x3", 3, 32, (3, 3), 2, (0, 0)),
    ("Conv2d_2a_3x3", 32, 32, (3, 3), 1, (0, 0)),
    ("Conv2d_2b_3x3", 32, 64, (3, 3), 1, (1, 1)),
    # max_pool 3 / 2
    ("Conv2d_3b_1x1", 64, 80, (1, 1), 1, (0, 0)),
    ("Conv2d_4a_3x3", 80, 192, (3, 3), 1, (0, 0)),
    # max_pool 3 / 2
]


def inception_a(p: str, cin: int, pool: int) -> List[Conv]:
    return [
        (f"{p}.branch1x1", cin, 64, (1, 1), 1, (0, 0)),
        (f"{p}.branch5x5_1", cin, 48, (1, 1), 1, (0, 0)),
        (f"{p}.branch5x5_2", 48, 64, (5, 5), 1, (2, 2)),
        (f"{p}.branch3x3dbl_1", cin, 64, (1, 1), 1, (0, 0)),
        (f"{p}.branch3x3dbl_2", 64, 96, (3, 3), 1, (1, 1)),
        (f"{p}.branch3x3dbl_3", 96, 96, (3, 3), 1, (1, 1)),
        (f"{p}.branch_pool", cin, pool, (1, 1), 1, (0, 0)),
    ]


def inception_b(p: str, cin: int) -> List[Conv]:
    return [
        (f"{p}.branch3x3", cin, 384, (3, 3), 2, (0, 0)),
        (f"{p}.branch3x3dbl_1", cin, 64, (1, 1), 1, (0, 0)),
        (f"{p}.branch3x3dbl_2", 64, 96, (3, 3), 1, (1, 1)),
        (f"{p}.branch3x3dbl_3", 96, 96, (3, 3), 2, (0, 0)),
    ]


def inception_c(p: str, cin: int, c7: int) -> List[Conv]:
    return [
        (f"{p}.branch1x1", cin, 192, (1, 1), 1, (0, 0)),
        (f"{p}.branch7x7_1", cin, c7, (1, 1), 1, (0, 0)),
        (f"{p}.branch7x7_2", c7, c7, (1, 7), 1, (0, 3)),
        (f"{p}.branch7x7_3", c7, 192, (7, 1), 1, (3, 0)),
        (f"{p}.branch7x7dbl_1", cin, c7, (1, 1), 1, (0, 0)),
        (f"{p}.branch7x7dbl_2", c7, c7, (7, 1), 1, (3, 0)),
        (f"{p}.branch7x7dbl_3", c7, c7, (1, 7), 1, (0, 3)),
        (f"{p}.branch7x7dbl_4", c7, c7, (7, 1), 1, (3, 0)),
        (f"{p}.branch7x7dbl_5", c7, 192, (1, 7), 1, (0, 3)),
        (f"{p}.branch_pool", cin, 192, (1, 1), 1, (0, 0)),
    ]


def inception_d(p: str, cin: int) -> List[Conv]:
    return [
        (f"{p}.branch3x3_1", cin, 192, (1, 1), 1, (0, 0)),
        (f"{p}.branch3x3_2", 192, 320, (3, 3), 2, (0, 0)),
        (f"{p}.branch7x7x3_1", cin, 192, (1, 1), 1, (0, 0)),
        (f"{p}.branch7x7x3_2", 192, 192, (1, 7), 1, (0, 3)),
        (f"{p}.branch7x7x3_3", 192, 192, (7, 1), 1, (3, 0)),
        (f"{p}.branch7x7x3_4", 192, 192, (3, 3), 2, (0, 0)),
    ]


def inception_e(p: str, cin: int) -> List[Conv]:
    return [
        (f"{p}.branch1x1", cin, 320, (1, 1), 1, (0, 0)),
        (f"{p}.branch3x3_1", cin, 384, (1, 1), 1, (0, 0)),
        (f"{p}.branch3x3_2a", 384, 384, (1, 3), 1, (0, 1)),
        (f"{p}.branch3x3_2b", 384, 384, (3, 1), 1, (1, 0)),
        (f"{p}.branch3x3dbl_1", cin, 448, (1, 1), 1, (0, 0)),
        (f"{p}.branch3x3dbl_2", 448, 384, (3, 3), 1, (1, 1)),
        (f"{p}.branch3x3dbl_3a", 384, 384, (1, 3), 1, (0, 1)),
        (f"{p}.branch3x3dbl_3b", 384, 384, (3, 1), 1, (1, 0)),
        (f"{p}.branch_pool", cin, 192, (1, 1), 1, (0, 0)),
    ]


# (block kind, name, constructor arguments): the order of torchvision's Inception3.forward
BLOCKS = [
    ("A", "Mixed_5b", (192, 32)), ("A", "Mixed_5c", (256, 64)), ("A", "Mixed_5d", (288, 64)),
    ("B", "Mixed_6a", (288,)),
    ("C", "Mixed_6b", (768, 128)), ("C", "Mixed_6c", (768, 160)), ("C", "Mixed_6d", (768, 160)),
    ("C", "Mixed_6e", (768, 192)),
    ("D", "Mixed_7a", (768,)),
    ("E", "Mixed_7b", (1280,)), ("E", "Mixed_7c", (2048,)),
]
_BUILD = {"A": inception_a, "B": inception_b, "C": inception_c, "D": inception_d, "E": inception_e}
FEATURE_DIM = 2048
NUM_CLASSES = 1000
# pytorch_fid.InceptionV3.BLOCK_INDEX_BY_DIM
BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}


def all_convs() -> List[Conv]:
    out = list(STEM)
    for kind, name, args in BLOCKS:
        out += _BUILD[kind](name, *args)
    return out


def block_convs(kind: str, name: str, args) -> Dict[str, Conv]:
    """The BasicConv2d layers of one Inception block keyed by their short name (``branch1x1`` ...)."""
    return {c[0].split(".", 1)[1]: c for c in _BUILD[kind](name, *args)}


def inception_param_spec(with_fc: bool = True) -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict() entries on the evaluation path (AuxLogits and num_batches_tracked are not)."""
    spec: List[Tuple[str, Tuple[int, ...]]] = []
    for name, cin, cout, (kh, kw), _, _ in all_convs():
        spec.append((f"{name}.conv.weight", (cout, cin, kh, kw)))
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            spec.append((f"{name}.bn.{leaf}", (cout,)))
    if with_fc:
        spec += [("fc.weight", (NUM_CLASSES, FEATURE_DIM)), ("fc.bias", (NUM_CLASSES,))]
    return spec
