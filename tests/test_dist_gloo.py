"""The N>1 path on CPU: world_size-2 gloo runs of the shard + single all-gather wrappers of
diffusion_models_amd.dist.

* ``test_world2_oracle_sampler_sharded_equals_unsharded``: every rank runs the ORACLE's DDIM sampler (CPU restatement of
  the reference, test infrastructure) on its slice of the batch with noise keyed by the GLOBAL sample index; the
  gathered batch must equal the single-process batch.  This is the sharding contract of SURVEY.md 8(e) with a real
  sampler on both sides (the GPU equivalent, Philox keyed by global element index, is
  tests/test_hip_configs.py::test_sharded_seeded_sampling_equals_unsharded_bitwise).
* ``test_world2_sample_global``: ``dist.sample_global`` hands every rank the same seed (broadcast from rank 0) and its
  own ``sample_offset``; even and ragged splits, a batch smaller than the world (one rank's slice is EMPTY: it must
  still enter the collective), and a per-sample condition tensor (sliced to each rank's rows).
"""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

from conftest import ROOT, rel_l2

SHAPE = (3, 8, 8)
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class GlobalNoise:
    """Draw k of sample i comes from a generator seeded by (k, i): independent of how the batch is split."""

    def __init__(self, lo, hi):
        self.lo, self.hi, self.k = lo, hi, 0

    def __call__(self, shape):
        rows = [torch.randn(tuple(shape[1:]), generator=torch.Generator().manual_seed(7919 * self.k + i + 1))
                for i in range(self.lo, self.hi)]
        self.k += 1
        return torch.stack(rows) if rows else torch.zeros((0,) + tuple(shape[1:]))


def _oracle_sampler():
    sys.path.insert(0, ROOT)
    import diffusion_models_amd as dm
    from diffusion_models_amd.spec import UnetConfig
    from oracle import sampler_oracle as so
    from oracle import unet_oracle as uo

    torch.set_num_threads(1)  # bitwise reproducible summation order whatever the batch
    cfg = UnetConfig(dim=16, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=17)
    sched = dm.make_schedule(1000, "linear")

    def sample(lo, hi):
        if hi == lo:
            return torch.zeros((0,) + SHAPE)
        return so.ddim_sample(lambda x, t: uo.unet_forward(sd, cfg, x, t), sched, (hi - lo,) + SHAPE, GlobalNoise(lo, hi),
                              STEPS, eta=0.5)

    return sample


def _worker_oracle(rank, world, port, batch, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from diffusion_models_amd.dist import sample_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = sample_sharded(_oracle_sampler(), batch)
        torch.save(full, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_world2_oracle_sampler_sharded_equals_unsharded(tmp_path):
    sample = _oracle_sampler()
    for batch in (4, 3):  # even and ragged split
        port = _free_port()
        mp.spawn(_worker_oracle, args=(2, port, batch, str(tmp_path)), nprocs=2, join=True)
        want = sample(0, batch)
        for r in range(2):
            got = torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True)
            assert got.shape == want.shape
            # per-sample arithmetic is independent of the batch it rides in up to the CPU conv's blocking (fp32 rounding)
            assert rel_l2(got, want) < 1e-5, rel_l2(got, want)


class _FakeDiffusion:
    """Stands in for DenoisingDiffusion on CPU: a 'sample' is a function of (seed, global sample index[, its condition
    row]) only.  Like the real classes it refuses an empty batch (dm_sample: "empty run")."""

    image_size = SHAPE[1:]
    channels = SHAPE[0]
    device = "cpu"

    def __init__(self):
        self.calls = []

    def sample(self, batch_size, seed, sample_offset, cond=None):
        assert batch_size > 0, "empty run"
        assert cond is None or cond.shape[0] == batch_size, "condition rows != batch"
        self.calls.append((batch_size, seed, sample_offset))
        rows = [torch.randn(SHAPE, generator=torch.Generator().manual_seed(seed % (2 ** 31) + i))
                for i in range(sample_offset, sample_offset + batch_size)]
        out = torch.stack(rows)
        return out if cond is None else out + cond.reshape(batch_size, 1, 1, 1)


def _worker_global(rank, world, port, batch, out_dir, with_cond):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from diffusion_models_amd.dist import sample_global

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # ranks disagree about the default seed: rank 0's must win
        d = _FakeDiffusion()
        kw = {"cond": torch.arange(batch, dtype=torch.float32) * 10.0} if with_cond else {}
        full = sample_global(d, batch, **kw)
        torch.save({"full": full, "calls": d.calls}, os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_world2_sample_global(tmp_path):
    for batch, with_cond in ((6, False), (5, False), (1, False), (5, True), (1, True)):
        port = _free_port()
        mp.spawn(_worker_global, args=(2, port, batch, str(tmp_path), with_cond), nprocs=2, join=True)
        r0 = torch.load(os.path.join(tmp_path, "g0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(tmp_path, "g1.pt"), weights_only=True)
        (b0, seed0, off0) = r0["calls"][0]
        if batch == 1:
            assert r1["calls"] == [], "the rank with an empty slice must not call the library"
            b1 = 0
        else:
            (b1, seed1, off1) = r1["calls"][0]
            assert seed0 == seed1 and off1 == b0
        assert off0 == 0 and b0 + b1 == batch
        cond = torch.arange(batch, dtype=torch.float32) * 10.0 if with_cond else None
        want = _FakeDiffusion().sample(batch, seed0, 0, cond)
        assert r0["full"].shape == want.shape
        assert torch.equal(r0["full"], want) and torch.equal(r1["full"], want)


def test_sample_global_refuses_text_model_without_embeddings():
    import pytest

    from diffusion_models_amd.dist import sample_global

    class _Text(_FakeDiffusion):
        class model:
            text_condition = True

    with pytest.raises(ValueError, match="text_emb"):
        sample_global(_Text(), 4)
