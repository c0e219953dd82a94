"""Data-parallel training on the one GPU a test box has: two ranks (gloo, both on cuda:0 -- RCCL refuses two ranks on one
device) each run loss + backward on HALF of a batch, all-reduce the library's flat gradient buffer in place (one
collective for all 245 gradients, a zero-copy torch view over device memory the library owns) and take the optimiser step;
the parameters must equal those of one process training on the whole batch.  The RCCL leg of the same call is covered at
world size 1 (nccl backend).  N > 1 ranks over xGMI remain for the driver's multi-GPU runs."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["DM_ROOT"])
import torch
import torch.distributed as dist
import diffusion_models_amd as dm
from diffusion_models_amd.spec import UnetConfig

rank, world, backend = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ["DM_BACKEND"]
torch.cuda.set_device(0)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=42)
    def model():
        u = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device="cuda:0")
        u.load_state_dict(sd)
        return dm.DenoisingDiffusion(u, image_size=16, timesteps=1000).train()
    g = torch.Generator().manual_seed(7)
    B = 8
    img = torch.rand((B, 3, 16, 16), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    nz = torch.randn((B, 3, 16, 16), generator=g)
    lo, hi = rank * B // world, (rank + 1) * B // world
    d = model()
    bucketed = os.environ.get("DM_TEST_BUCKETED") == "1"   # False: ONE collective over the whole buffer
    if bucketed:
        assert len(d.model.grad_buckets()) >= 2, d.model.grad_buckets()   # DM_TRAIN_BUCKET_MB=1: several buckets on this net
    # sharded: the all-reduce(s) inside; asynchronous, so that a bucket's collective really is enqueued beside the pass
    loss, norm = dm.train_step(d, [img[lo:hi]], lr=1e-3, t=[t[lo:hi]], noise=[nz[lo:hi]], sync=False, bucketed=bucketed)
    assert d.model._bucketed == bool(bucketed) and float(loss) > 0 and float(norm) > 0
    got = d.model.state_dict()
    if rank == 0:
        dist_was = dist.is_initialized()
        ref = model()
        import diffusion_models_amd.train as tr
        # the same iteration on the whole batch without a process group in the way
        loss = ref.p_losses(ref.normalize(img.to("cuda:0")), t, noise=nz)
        ref.model.optimizer_step(lr=1e-3)
        want = ref.model.state_dict()
        worst = max(float((got[k] - want[k]).norm() / want[k].norm().clamp_min(1e-30)) for k in want)
        print("WORST", worst)
        assert worst < 2e-5, worst
    dist.barrier()
    print("DP_OK", rank)
finally:
    dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, backend, bucketed=False):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, DM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(r), DM_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if bucketed:  # 1 MB buckets: the two-stage test net (15 MB of gradients) then has several
            env.update(DM_TEST_BUCKETED="1", DM_TRAIN_BUCKET_MB="1")
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and "DP_OK" in o, o[-2000:] + e[-3000:]
    return outs


def test_two_ranks_gloo_equal_one_process_on_the_whole_batch():
    outs = _run(2, "gloo")
    print(outs[0][0])


def test_world1_nccl_all_reduce_of_the_flat_gradient_buffer():
    _run(1, "nccl")


def test_two_ranks_gloo_bucketed_all_reduce_overlapping_the_backward_pass():
    """The same equality with the gradient buffer all-reduced bucket by bucket on a second stream, each bucket as soon as the
    backward pass has completed it (torch DDP's overlap; Unet.grad_buckets / dm_unet_train_bucket)."""
    outs = _run(2, "gloo", bucketed=True)
    print(outs[0][0])


def test_world1_nccl_bucketed_all_reduce():
    _run(1, "nccl", bucketed=True)
