"""RCCL on the one GPU a test box has: a fresh child process joins a world-size-1 process group on the **nccl** backend
(= RCCL on ROCm) and runs ``dist.sample_global`` / ``dist.gather_shards``, so RCCL initialisation, the seed broadcast and
``all_gather_into_tensor`` -- the only collective of the multi-GPU path -- execute once on real
hardware.  N > 1 ranks over xGMI remain unmeasured here (the driver runs them); the N > 1 logic is covered by the
world-size-2 gloo tests in test_dist_gloo.py.

The child is started before it touches the GPU (a new interpreter, never a re-exec of this process)."""
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

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    assert dist.get_backend() == "nccl"
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=3)
    u = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device="cuda:0")
    u.load_state_dict(sd)
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000, sampling_timesteps=3)
    full = dm.sample_global(d, 4, seed=1234)                 # shared_seed -> sample -> all_gather_into_tensor (RCCL)
    want = d.sample(batch_size=4, seed=1234, sample_offset=0)
    assert full.shape == (4, 3, 16, 16) and full.is_cuda
    assert torch.equal(full, want), float((full - want).abs().max())
    seeded = dm.shared_seed(None)                            # broadcast of rank 0's draw over RCCL
    assert isinstance(seeded, int)
    x = torch.arange(24, dtype=torch.float32, device="cuda:0").reshape(4, 6)
    assert torch.equal(dm.gather_shards(x, 4), x)
    torch.cuda.synchronize()
    print("NCCL_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
finally:
    dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world1_nccl_sample_global_runs_rccl():
    env = dict(os.environ, DM_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
