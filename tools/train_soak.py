import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusion_models_amd as dm
u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, dropout=0.1, device="cuda:0")
u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
d = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000).train()
ema = dm.EMA(d, beta=0.995, update_every=10)
img = torch.rand(64, 3, 32, 32, device="cuda:0")
for i in range(5): dm.train_step(d, [img], lr=2e-4, ema=ema)
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter(); marks = []
for i in range(1500):
    loss, norm = dm.train_step(d, [img], lr=2e-4, ema=ema)
    if i % 500 == 499:
        torch.cuda.synchronize(); marks.append((i + 1, time.perf_counter() - t0, loss, torch.cuda.mem_get_info()[0]))
for m in marks: print(m)
print("free memory change (MB):", (torch.cuda.mem_get_info()[0] - free0) / 1e6)
out = ema.ema_model.ddim_sample((8, 3, 32, 32), sampling_timesteps=10)
print("ema sample finite:", bool(torch.isfinite(out).all()), float(out.min()), float(out.max()))
