"""2 ranks (gloo, one GPU): per-iteration host times of bench.py's data-parallel train leg (dropout, EMA, timing events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import diffusion_models_amd as dm
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, dropout=float(os.environ.get("P", "0.1")), device="cuda:0")
u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
d = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000).train()
ema = dm.EMA(d, beta=0.995, update_every=10) if os.environ.get("EMA", "1") == "1" else None
img = torch.rand(64, 3, 32, 32, device="cuda:0")
for _ in range(3):
    dm.train_step(d, [img], lr=2e-4, ema=ema)
torch.cuda.synchronize(); dist.barrier()
timing = {} if os.environ.get("TIMING", "1") == "1" else None
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    dm.train_step(d, [img], lr=2e-4, ema=ema, sync=False, timing=timing)
    ts.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
if rank == 0:
    print("host ms per iteration:", [round(v, 1) for v in ts], flush=True)
dist.destroy_process_group()
