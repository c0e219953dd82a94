"""The N>1 path on CPU: world_size-2 gloo run of the shard + single all-gather wrapper.
The per-rank "sampler" is the oracle on CPU (test infrastructure); what is under test is the
sharding/gather logic of diffusion_models_amd.dist and that the gathered batch equals the
single-process batch when noise is keyed by global sample index."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _per_sample(lo, hi):
    # deterministic per-GLOBAL-index "sample": what a rank produces for its slice
    rows = []
    for i in range(lo, hi):
        g = torch.Generator().manual_seed(1000 + i)
        rows.append(torch.randn((3, 4, 4), generator=g))
    return torch.stack(rows) if rows else torch.zeros((0, 3, 4, 4))


def _worker(rank, world, port, batch, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from diffusion_models_amd.dist import sample_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = sample_sharded(_per_sample, batch)
        torch.save(full, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_world2_gather_equals_single_process(tmp_path):
    for batch in (8, 5):  # even and ragged split
        port = _free_port()
        mp.spawn(_worker, args=(2, port, batch, str(tmp_path)), nprocs=2, join=True)
        want = _per_sample(0, batch)
        for r in range(2):
            got = torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True)
            assert got.shape == want.shape and torch.equal(got, want)
