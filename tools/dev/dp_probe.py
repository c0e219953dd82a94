"""2 ranks (gloo, one GPU): host time of each step of train_step's bucketed / unbucketed gradient averaging."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import diffusion_models_amd as dm
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, device="cuda:0")
u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
d = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000).train()
img = torch.rand(64, 3, 32, 32, device="cuda:0")
for mode in (False, True, False, True):
    for _ in range(2):
        dm.train_step(d, [img], lr=2e-4, bucketed=mode)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(4):
        dm.train_step(d, [img], lr=2e-4, sync=False, bucketed=mode)
    torch.cuda.synchronize()
    if rank == 0:
        print(f"bucketed={mode}: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms per iteration", flush=True)
# per-collective host times in bucketed mode
flat = u.grads_flat()
side = torch.cuda.Stream()
for rep in range(2):
    d.p_losses(d.normalize(img), torch.randint(0, 1000, (64,)), sync=False)
    for b, (off, n) in enumerate(u.grad_buckets(enable=True)):
        t0 = time.perf_counter(); u.bucket_wait(b, side); t1 = time.perf_counter()
        with torch.cuda.stream(side):
            dist.all_reduce(flat[off:off + n]); t2 = time.perf_counter()
        if rank == 0:
            print(f"rep {rep} bucket {b} ({4 * n / 2**20:.0f} MB): wait {1e3 * (t1 - t0):.2f} ms, all_reduce call {1e3 * (t2 - t1):.1f} ms", flush=True)
    torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
dist.destroy_process_group()
