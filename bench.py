#!/usr/bin/env python3
"""Headline benchmark: sampled images/sec of the 32x32 denoising U-Net, DDPM-1000, on MI355X.

    python bench.py --gpus N --steps K --warmup W

BASELINE.json's metric: "sampled images/sec (and denoise-steps/sec) at 32x32 DDPM-1000, 1/2/4/8 GPU".
One "step" = one pass of the hot path over one batch = one complete ``sample()`` call: 32x32 U-Net (dim 64, mults
(1,2,4,8)), ``p_sample_loop`` with T = 1000 reverse steps, 256 images per GPU, the denoise step replayed as a hipGraph
captured once, device Philox noise, name-seeded random-init weights (no network for checkpoints).  Everything is resident
in HBM when the timed region starts; the boundary takes device pointers, so there is no PCIe leg.

N > 1: ``python bench.py --gpus N`` starts N ranks itself (``python -m torch.distributed.run`` as a child process, before
this process touches the GPU); under a launcher (WORLD_SIZE set) it is a rank.  Every rank samples its own 256 images of
the global batch (weak scaling; Philox counters are global element indices, so the gathered batch equals the batch one
GPU would produce) and ONE all-gather over RCCL assembles the (256*N) batch inside the timed region.
``--scaling strong --global-batch G`` fixes the global batch instead (G/N per GPU).

Rank 0 prints ONE JSON line.  Besides the contract's fields: ``roofline`` of the dominant kernel (timed live by HIP
events on its launch stream), ``cpu_baseline`` (the oracle on this box's host cores on BASELINE config 1, bounded
sample), per-kernel rows and the DDIM-50 rate of the same step (``--workload ddim50`` measures that one directly).
"""
import argparse
import glob
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3  # MI355X f32 vector == f32-input MFMA peak (MI355X_MICROARCH.md)
IMAGE, CHANNELS, T = 32, 3, 1000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per sample() call (weak scaling)")
    ap.add_argument("--workload", default="ddpm1000", choices=["ddpm1000", "ddim50"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--global-batch", type=int, default=256, help="global batch of --scaling strong")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=20, help="DDPM steps of the CPU baseline sample (config 1)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` typed as is: start N fresh ranks as a child process.  Nothing in this process has
    touched the GPU (torch is not even imported yet), and the parent only waits and forwards the exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def csrc_sha() -> str:
    """Hash of the kernel sources: a PMC traffic file only applies to the kernels it was collected on."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "diffusion-models_amd", "csrc", "*"))):
        if path.endswith((".hip", ".h", ".inc")):
            with open(path, "rb") as f:
                h.update(os.path.basename(path).encode())
                h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, FETCH_SIZE doubled per the gfx950 correction; tools/pmc_traffic.py).  PMC cannot run inside this
    process, so the figure is an offline pass of the same command on the same kernels: it is used only when the file's
    source hash equals the hash of the sources this library was built from."""
    import re

    files = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json"))
    if not files:
        return None, "no profiles/r*_pmc_hbm_traffic.json"
    # newest = highest (round, step) in the name: r2_step10 is later than r2_step9
    newest = max(files, key=lambda p: tuple(int(v) for v in re.findall(r"\d+", os.path.basename(p))))
    with open(newest) as f:
        doc = json.load(f)
    sha = doc.get("_csrc_sha")
    if sha != csrc_sha():
        return None, f"{os.path.basename(newest)} was collected on other kernel sources ({sha} != {csrc_sha()})"
    rows = {k.replace(" ", ""): v for k, v in doc.items() if isinstance(v, dict)}
    want = kernel.replace(" ", "")
    row = rows.get(want)
    if not row and want.endswith(">"):
        # the library's profile name may carry fewer template arguments than the symbol (wino_mfma_kernel<R> times every
        # <R, Q> instantiation): take the instantiation(s) of that prefix -- one row, or the launch-weighted mean of several
        cand = [v for k, v in rows.items() if k.startswith(want[:-1] + ",")]
        if len(cand) == 1:
            row = cand[0]
        elif cand and all("launches" in v for v in cand):
            n = sum(v["launches"] for v in cand)
            row = {k: sum(v[k] * v["launches"] for v in cand) / n for k in cand[0]
                   if isinstance(cand[0][k], (int, float)) and k != "launches" and all(k in v for v in cand)}
            row["launches"] = n
    if not row:
        return None, f"{os.path.basename(newest)} has no row for {kernel}"
    return row["fetch_bytes_per_launch_corrected"] + row["write_bytes_per_launch"], os.path.basename(newest)


def pmc_file():
    """(doc, name) of the newest committed PMC pass when it was collected on THESE kernel sources, else (None, why)."""
    import re

    files = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json"))
    if not files:
        return None, "no profiles/r*_pmc_hbm_traffic.json"
    newest = max(files, key=lambda p: tuple(int(v) for v in re.findall(r"\d+", os.path.basename(p))))
    with open(newest) as f:
        doc = json.load(f)
    if doc.get("_csrc_sha") != csrc_sha():
        return None, f"{os.path.basename(newest)} was collected on other kernel sources"
    return doc, os.path.basename(newest)


FAMILIES = (("wino4", "wino4_mfma_kernel"), ("wino", "wino_mfma_kernel"), ("upwino", "upwino_mfma_kernel"),
            ("pw", "pw_mfma_kernel"), ("linattn", "linattn_"), ("attn16", "attn16_fused_kernel"),
            ("landing", "norm_act"), ("init7", "init7_mfma_kernel"), ("direct", "conv_mfma_kernel"))


def family_of(kernel: str) -> str:
    for fam, prefix in FAMILIES:
        if kernel.startswith(prefix):
            return fam
    return "other"


def step_breakdown(ms_per_denoise_step: float, pmc_steps: int = 50):
    """Whole-step and per-family MFMA utilisation by EXECUTED FLOPs: SQ_INSTS_VALU_MFMA_MOPS_F32 (x512 FLOPs) per launch
    x launches per denoise step, from the committed PMC pass of `bench.py --workload ddim50 --steps 1` (50 steps of the
    same denoise step), over this run's measured step time; a family's share of the step is its kernel time in the
    committed rocprofv3 --kernel-trace --stats summary of the headline command over that summary's total."""
    import csv
    import re

    doc, src = pmc_file()
    if doc is None:
        return None, src
    flops = {}
    for k, row in doc.items():
        if not isinstance(row, dict) or "mfma_mops_f32_per_launch" not in row:
            continue
        name = k.replace("dm::", "")
        fam = family_of(name)
        flops[fam] = flops.get(fam, 0.0) + row["mfma_mops_f32_per_launch"] * 512.0 * row["launches"] / pmc_steps
    total = sum(flops.values())
    out = {"step_executed_tflops": total / (ms_per_denoise_step * 1e-3) / 1e12,
           "step_executed_frac": total / (ms_per_denoise_step * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
           "source": src}
    stats = glob.glob(os.path.join(ROOT, "profiles", src.replace("_pmc_hbm_traffic.json", "_kernel_stats.csv")))
    fam_ns, all_ns = {}, 0.0
    if stats:
        for r in csv.DictReader(open(stats[0])):
            name = re.sub(r"^void\s+", "", r["Name"]).replace("dm::", "")
            ns = float(r["TotalDurationNs"])
            all_ns += ns
            fam_ns[family_of(name)] = fam_ns.get(family_of(name), 0.0) + ns
    fams = {}
    for fam in sorted(set(flops) | set(fam_ns), key=lambda f: -fam_ns.get(f, 0.0)):
        share = fam_ns.get(fam, 0.0) / all_ns if all_ns else None
        row = {"share_of_step": share}
        if share and flops.get(fam):
            row["executed_tflops"] = flops[fam] / (share * ms_per_denoise_step * 1e-3) / 1e12
            row["executed_frac"] = row["executed_tflops"] / PEAK_F32_TFLOPS
        fams[fam] = row
    out["families"] = fams
    return out, src


TRAIN_FAMILIES = (("wgrad", ("wgrad_", "thin_out_bwd")), ("wino4", "wino4_mfma_kernel"), ("wino", "wino_mfma_kernel"),
                  ("upwino", "upwino_mfma_kernel"), ("pw", "pw_mfma_kernel"), ("norm_bwd", "norm_act_bwd"),
                  ("norm_fwd", "norm_act"), ("linattn_bwd", "linattn_bwd"), ("linattn", "linattn_"), ("attn", "att"),
                  ("optimiser", ("adam_ema", "pack_", "scatter_copy", "sumsq", "clip_coef", "lerp", "rot_transpose")),
                  ("linear", ("linear_", "rows_gemm", "mlp_rows", "colsum", "rowgrad", "act_")), ("init7", "init7_mfma_kernel"))


def train_family_of(kernel: str) -> str:
    for fam, prefix in TRAIN_FAMILIES:
        if kernel.startswith(prefix):
            return fam
    return "other"


def train_roofline(ms_per_iteration: float):
    """The training iteration against the f32-MFMA peak by EXECUTED FLOPs, as the sampling step is priced: the committed
    PMC pass of `tools/train_time.py --batch 64 --dropout 0.1 --full-only` (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOPs per
    launch x launches / iterations, HBM bytes from FETCH_SIZE x2 + WRITE_SIZE) over THIS run's measured iteration time; a
    family's share is its kernel time in the committed rocprofv3 --kernel-trace --stats summary of the same command.
    Accepted only when the file's hash of csrc/ equals the sources'."""
    import csv
    import re

    files = glob.glob(os.path.join(ROOT, "profiles", "r*_train_pmc.json"))
    if not files:
        return {"executed_frac": None, "source": "no profiles/r*_train_pmc.json"}
    newest = max(files, key=lambda p: tuple(int(v) for v in re.findall(r"\d+", os.path.basename(p))))
    with open(newest) as f:
        doc = json.load(f)
    src = os.path.basename(newest)
    if doc.get("_csrc_sha") != csrc_sha():
        return {"executed_frac": None, "source": f"{src} was collected on other kernel sources"}
    iters = doc.get("_iterations")
    if not iters:
        return {"executed_frac": None, "source": f"{src} does not record its iteration count"}
    flops, launches, hbm = {}, 0.0, 0.0
    for k, row in doc.items():
        if not isinstance(row, dict) or "launches" not in row:
            continue
        fam = train_family_of(k.replace("dm::", ""))
        launches += row["launches"] / iters
        hbm += (row["fetch_bytes_per_launch_corrected"] + row["write_bytes_per_launch"]) * row["launches"] / iters
        if "mfma_mops_f32_per_launch" in row:
            flops[fam] = flops.get(fam, 0.0) + row["mfma_mops_f32_per_launch"] * 512.0 * row["launches"] / iters
    total = sum(flops.values())
    sec = ms_per_iteration * 1e-3
    out = {"bound": "mfma", "executed_gflop_per_iteration": total / 1e9, "executed_tflops": total / sec / 1e12,
           "executed_frac": total / sec / 1e12 / PEAK_F32_TFLOPS, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
           "launches_per_iteration": round(launches, 1), "hbm_bytes_per_iteration": hbm, "hbm_TBps": hbm / sec / 1e12,
           "source": src}
    stats = os.path.join(ROOT, "profiles", src.replace("_train_pmc.json", "_train_kernel_stats.csv"))
    fam_ns, all_ns = {}, 0.0
    if os.path.exists(stats):
        for r in csv.DictReader(open(stats)):
            name = re.sub(r"^void\s+", "", r["Name"]).replace("dm::", "")
            ns = float(r["TotalDurationNs"])
            all_ns += ns
            fam_ns[train_family_of(name)] = fam_ns.get(train_family_of(name), 0.0) + ns
    fams = {}
    for fam in sorted(set(flops) | set(fam_ns), key=lambda f: -fam_ns.get(f, 0.0)):
        share = fam_ns.get(fam, 0.0) / all_ns if all_ns else None
        row = {"share_of_iteration": share}
        if share and flops.get(fam):
            row["executed_tflops"] = flops[fam] / (share * sec) / 1e12
            row["executed_frac"] = row["executed_tflops"] / PEAK_F32_TFLOPS
        fams[fam] = row
    out["families"] = fams
    return out


def other_configs(dev, replays: int = 100):
    """BASELINE configs 3, 4 and 5 on this GPU, timed in this process after the headline region: graph-replayed sampler,
    synthetic weights, `replays` denoise steps per timing (the per-step cost does not depend on the schedule length)."""
    import torch

    import diffusion_models_amd as dm
    from diffusion_models_amd.spec import DecoderConfig

    def unet(**kw):
        u = dm.Unet(device=dev, **kw)
        u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
        return u

    def timed(fn):
        fn()  # capture + warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    S = replays
    out = {}
    u = unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3)
    d = dm.DenoisingDiffusion(u, image_size=64, timesteps=1000)  # config 3 is DDPM: the p_sample step with its Philox draw
    for B in (32, 8):
        dt = timed(lambda: d.p_sample_loop((B, 3, 64, 64), seed=1, max_steps=S))
        out[f"config3_64x64_ddpm1000_b{B}_per_gpu"] = {"ms_per_denoise_step": 1e3 * dt / S, "replays": S,
                                                      "sampler": "p_sample_loop, first %d of 1000 steps" % S,
                                                      "images_per_s_per_gpu": B / (1000 * dt / S)}
    del d, u
    u4 = unet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
    dcfg = DecoderConfig(ch=64, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64, z_channels=4,
                         embed_dim=4)
    vae = dm.VQDecoder(dict(ch=64, out_ch=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64,
                            z_channels=4), embed_dim=4, device=dev)
    vae.load_state_dict(dm.synth_state_dict(dm.decoder_param_spec(dcfg), salt=4))
    ld = dm.LatentDiffusion(u4, vae, latent_shape=(4, 32, 32), timesteps=1000, sampling_timesteps=S)
    B = 128
    dt_loop = timed(lambda: ld.ddim_sample((B, 4, 32, 32), seed=1))
    z = ld.ddim_sample((B, 4, 32, 32), seed=1)
    dt_dec = timed(lambda: vae.decode(z))
    out["config4_latent4x32x32_ddim200_b128_with_decode"] = {
        "ms_per_denoise_step": 1e3 * dt_loop / S, "decode_ms": 1e3 * dt_dec, "replays": S,
        "images_per_s": B / (200 * dt_loop / S + dt_dec)}
    del ld, u4, vae
    ut = unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, text_condition=True, use_cross_attn=True)
    d5 = dm.TextConditionalDenoisingDiffusion(model=ut, image_size=64, timesteps=1000, sampling_timesteps=S)
    B = 32
    emb = torch.randn(B, 512, device=dev)
    dt = timed(lambda: d5.sample(batch_size=B, text_emb=emb, seed=1))
    out["config5_text64x64_ddim100_b32_per_gpu"] = {"ms_per_denoise_step": 1e3 * dt / S, "replays": S,
                                                    "images_per_s_per_gpu": B / (100 * dt / S)}
    return out


def train_leg(dev, with_cpu: bool, batch: int = 64, iters: int = 10, world: int = 1, rank: int = 0, rehearse: bool = False):
    """The training half of the caller of record, measured beside the headline: one iteration of
    Trainer.train (DD/denoising_diffusion.py:1164-1190: p_losses + backward, clip_grad_norm_, Adam, ema.update) at the shipped
    ddpm_cifar.yaml shape (32x32 U-Net dim 64, train_batch_size 64, dropout 0.1), all on the GPU with device-resident
    parameters; and the same iteration through torch autograd on this box's host cores (the oracle, one iteration).
    world > 1 (bench.py --gpus N): data-parallel as accelerate runs the reference's Trainer -- `batch` images per rank (weak
    scaling), ONE all-reduce of the flat gradient buffer over RCCL inside every timed iteration, every rank steps its own
    replica; the time is the maximum over ranks and the all-reduce's share comes from HIP events around it."""
    import torch
    import torch.distributed as dist

    import diffusion_models_amd as dm
    from diffusion_models_amd.spec import UnetConfig

    cfg = UnetConfig(dim=64, dim_mults=(1, 2, 4, 8), channels=CHANNELS)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=0)
    u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=CHANNELS, dropout=0.1, device=dev)
    u.load_state_dict(sd)
    d = dm.DenoisingDiffusion(u, image_size=IMAGE, timesteps=T).train()
    ema = dm.EMA(d, beta=0.995, update_every=10)
    img = torch.rand(batch, CHANNELS, IMAGE, IMAGE, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))
    for _ in range(3):  # warm-up: workspace sizing, then the first lazily re-packed iteration, then the steady state
        dm.train_step(d, [img], lr=2e-4, ema=ema)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timing = {} if world > 1 else None
    barrier()
    t0 = time.perf_counter()
    host_ms = []
    for _ in range(iters):  # enqueued back to back; loss / norm stay device tensors (the reference's loss.item() is the caller's)
        th = time.perf_counter()
        loss, norm = dm.train_step(d, [img], lr=2e-4, ema=ema, sync=False, timing=timing)
        host_ms.append(round(1e3 * (time.perf_counter() - th), 2))
    barrier()
    if os.environ.get("DM_BENCH_DEBUG"):
        print(f"[rank {rank}] train_step host ms per iteration: {host_ms}", file=sys.stderr)
    dt = (time.perf_counter() - t0) / iters
    if world > 1:
        tmax = torch.tensor([dt], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss, norm = float(loss), float(norm)
    assert math.isfinite(loss) and math.isfinite(norm) and loss > 0 and norm > 0, (loss, norm)
    out = {"workload": f"Trainer.train iteration, 32x32 U-Net dim 64, batch {batch} per GPU, dropout 0.1, Adam + clip + EMA",
           "ms_per_iteration": 1e3 * dt, "images_per_s": world * batch / dt, "iterations_timed": iters, "loss": loss,
           "grad_norm": norm, "n_gpus": world, "global_batch": world * batch}
    if world > 1:
        ar = [a.elapsed_time(b) for a, b in timing["allreduce_events"]]
        out["allreduce_ms"] = sum(ar) / len(ar)
        out["allreduce_share"] = out["allreduce_ms"] / (1e3 * dt)
        out["allreduce_bytes"] = int(d.model.grads_flat().numel()) * 4
        out["allreduce"] = ("bucket by bucket on a second stream, each bucket as soon as the backward pass has completed it "
                            "(allreduce_ms = what is left exposed behind the pass)" if getattr(d.model, "_bucketed", False)
                            else "one in-place all-reduce of the flat gradient buffer per iteration, behind the backward pass")
        out["gradient_buckets_MB"] = [round(4 * n / 2 ** 20, 1) for _, n in d.model.grad_buckets()]
    out["roofline"] = train_roofline(1e3 * dt) if world == 1 else None
    del d, u, ema
    if with_cpu:
        from oracle import train_oracle as to

        sched = dm.make_schedule(T, "linear")
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = min(16, avail)
        torch.set_num_threads(threads)
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        opt = torch.optim.Adam(list(params.values()), lr=2e-4, betas=(0.9, 0.99))
        x = img.cpu() * 2 - 1
        tt = torch.randint(0, T, (batch,))
        nz = torch.randn(x.shape)

        def one():
            opt.zero_grad()
            l = to.p_losses(params, cfg, sched, x, tt, nz)
            l.backward()
            torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
            opt.step()

        one()  # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        one()
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"ms_per_iteration": 1e3 * cdt, "images_per_s": batch / cdt, "cores": threads, "kind": "port",
                               "sample": "one iteration after one warm-up iteration, torch autograd through the oracle U-Net "
                                         "(no dropout), clip_grad_norm_, torch.optim.Adam"}
        out["gpu_over_cpu"] = cdt / dt
    return out


def cpu_baseline(n_steps):
    """BASELINE config 1 exactly: Unet(dim 64, mults (1,2,4,8), channels 3), 32x32, linear beta,
    T = 1000, p_sample_loop, B = 64, fp32 -- the oracle (CPU restatement pinned to the reference's outputs) on this
    box's host cores.  The loop is per-step homogeneous: `n_steps` DDPM steps are timed after one warm-up step and
    extrapolated linearly to 1000.  The thread count is swept over {8, 16, 32, 64} (3 steps each); the two best run the
    full sample and the faster one is reported."""
    import torch

    import diffusion_models_amd as dm
    from diffusion_models_amd.spec import UnetConfig
    from oracle import sampler_oracle as so
    from oracle import unet_oracle as uo

    B = 64
    cfg = UnetConfig(dim=64, dim_mults=(1, 2, 4, 8), channels=CHANNELS)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=0)
    sched = dm.make_schedule(T, "linear")
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    model = lambda x, t: uo.unet_forward(sd, cfg, x, t)  # noqa: E731
    shape = (B, CHANNELS, IMAGE, IMAGE)

    def run(steps, seed=0):
        stream = so.NoiseStream(seed)
        with torch.inference_mode():
            x = stream(shape)
            x, _ = so.p_sample(model, sched, x, T - 1, stream(shape))  # warm-up step (thread pool, allocator)
            t0 = time.perf_counter()
            for i in range(steps):
                x, _ = so.p_sample(model, sched, x, T - 2 - i, stream(shape))
            return time.perf_counter() - t0

    sweep = {}
    for n in sorted({min(c, avail) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        sweep[n] = run(3) / 3
    # the short sweep is noisy on a shared host: the two best thread counts both run the full sample, the faster one counts
    finalists = sorted(sweep, key=sweep.get)[:2]
    timed = {}
    for n in finalists:
        torch.set_num_threads(n)
        timed[n] = run(n_steps)
    best = min(timed, key=timed.get)
    dt = timed[best]
    step_s = dt / n_steps
    return {
        "value": B / (step_s * T),
        "unit": "images/s",
        "cores": best,
        "kind": "port",
        "image_steps_per_s": B / step_s,
        "sample": f"BASELINE config 1: B=64, 32x32, p_sample (DDPM), {n_steps} of 1000 steps timed ({dt:.1f} s) after 1 "
                  f"warm-up step, linear extrapolation to 1000; threads swept {{"
                  + ", ".join(f"{k}: {1e3 * v:.0f} ms/step" for k, v in sweep.items())
                  + f"}} of {avail} available, full sample at {finalists}, best = {best}; torch {torch.__version__} CPU ops, fp32",
    }


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # DM_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend -- exercises the N > 1 code path on a one-GPU box
    rehearse = os.environ.get("DM_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import diffusion_models_amd as dm
    from diffusion_models_amd import _lib
    from diffusion_models_amd.spec import UnetConfig

    cfg = UnetConfig(dim=64, dim_mults=(1, 2, 4, 8), channels=CHANNELS)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=0)
    unet = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=CHANNELS, device=dev)
    unet.load_state_dict(sd)
    S = 50 if args.workload == "ddim50" else T
    diff = dm.DenoisingDiffusion(unet, image_size=IMAGE, timesteps=T,
                                 sampling_timesteps=S if S < T else None, use_graph=not args.no_graph)
    if args.scaling == "weak":
        B_global = args.batch * world
    else:
        B_global = args.global_batch
    lo, hi = dm.shard_bounds(B_global, world, rank)
    B = hi - lo

    def step(i):
        # one seed per call for every rank; this rank draws the noise of ITS global sample indices [lo, hi)
        local = diff.sample(batch_size=B, seed=1000 + i, sample_offset=lo)
        return dm.gather_shards(local, B_global) if world > 1 else local

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(-1 - i)
    barrier()
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        out = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert out.shape == (B_global, CHANNELS, IMAGE, IMAGE) and bool(torch.isfinite(out).all())
    assert unet.graph_captures <= 1, "the step graph must be captured once, not per sample() call"

    images = B_global * args.steps
    value = images / elapsed
    ms_denoise = 1e3 * elapsed / args.steps / S
    result = {
        "metric": "sampled images/sec, 32x32 U-Net (dim 64, mults 1-2-4-8), DDPM-1000" if S == T
                  else "sampled images/sec, 32x32 U-Net (dim 64, mults 1-2-4-8), DDIM-50",
        "value": value,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (name-seeded random-init weights, device Philox noise keyed by global sample index)",
        "config": {
            "workload": f"{args.workload}_32x32_unet64_b{args.batch if args.scaling == 'weak' else B}_per_gpu",
            "sampler": "p_sample_loop (DDPM)" if S == T else "ddim_sample (eta 0)",
            "sampler_steps": S,
            "batch_per_gpu": B,
            "global_batch": B_global,
            "hip_graph": not args.no_graph,
            "graph_captures": unet.graph_captures,
            "workspace_MB": round(unet.workspace_bytes / 2 ** 20, 1),
            "parallelism": f"batch-shard x{world}, one all-gather per sample()",
        },
        "denoise_image_steps_per_s": value * S,
        "ms_per_denoise_step": ms_denoise,
        "ddim50_equiv_images_per_s": value * S / 50.0,
    }

    if rank == 0 and not args.no_roofline:
        # roofline leg: the same denoise step, eager launches, every conv bracketed by HIP events on its stream
        diff_e = dm.DenoisingDiffusion(unet, image_size=IMAGE, timesteps=T, use_graph=False)
        _lib.profile_enable(True)
        diff_e.p_sample_loop((B, CHANNELS, IMAGE, IMAGE), seed=3, max_steps=4)
        rows = _lib.profile_read()
        _lib.profile_enable(False)
        rows.sort(key=lambda r: -r["total_ms"])
        kern = []
        for r in rows:
            avg_ms = r["total_ms"] / r["launches"]
            # the rows are priced as the reference's direct convolution; a Winograd-type kernel's own matrix work is a
            # fraction of that count.  Multiply-adds the kernel executes per multiply-add of the direct 3x3 convolution it is priced as:
            # F(4x4,3x3) 36/144, upsample algorithm 9/36 (per output pixel), F(2x2,3x3) 16/36
            k = r["kernel"]
            executed = 0.25 if k.startswith(("wino4", "upwino")) else (16.0 / 36.0 if k.startswith("wino") else 1.0)
            tf_direct = r["total_flops"] / (r["total_ms"] * 1e-3) / 1e12
            kern.append({"kernel": r["kernel"], "launches_per_unet_fwd": r["launches"] // 4,
                         "avg_ms": avg_ms, "mfma_tflops": tf_direct * executed,
                         "frac_f32_peak": tf_direct * executed / PEAK_F32_TFLOPS,
                         "direct_conv_equiv_tflops": tf_direct,
                         "algorithmic_bytes_per_launch": r["total_bytes"] / r["launches"],
                         "algorithmic_GBps": r["total_bytes"] / (r["total_ms"] * 1e-3) / 1e9})
        top = kern[0]
        traffic, traffic_src = pmc_traffic(top["kernel"])
        result["roofline"] = {
            "bound": "mfma",
            "kernel": top["kernel"],
            "achieved": top["mfma_tflops"],
            "peak": PEAK_F32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": top["mfma_tflops"] / PEAK_F32_TFLOPS,
            "algorithmic_equiv": top["direct_conv_equiv_tflops"],
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, offline pass of this command)",
            "algorithmic_bytes_per_launch": top["algorithmic_bytes_per_launch"],
            "avg_launch_ms": top["avg_ms"],
            "timing": "HIP events on the launch stream around every launch of an eager run behind a parked GPU, minus the "
                      "calibrated interval of an empty-kernel bracket (dispatch + event packets, ~9 us): kernel "
                      "execution time, comparable with rocprofv3 --kernel-trace",
            "note": "achieved = f32-MFMA FLOPs the kernel executes per launch (Winograd F(4x4,3x3): 2*36*Cin*Cout per 4x4 "
                    "output pixels; F(2x2,3x3): 2*16 per 2x2) / its average launch time; frac = achieved / f32 MFMA peak.  algorithmic_equiv prices "
                    "the same launches as the reference's direct 3x3 convolution (2*9*Cin*Cout per pixel, SURVEY 8(d)) "
                    "and can exceed the peak; it is not a roofline fraction",
        }
        result["kernels"] = kern
        result["conv_ms_per_unet_fwd"] = sum(r["total_ms"] for r in rows) / 4
        # the unflattering view: the whole step and every kernel family by EXECUTED MFMA FLOPs (PMC) over measured time
        brk, brk_src = step_breakdown(ms_denoise)
        if brk is not None:
            result["roofline"]["step_executed_frac"] = brk["step_executed_frac"]
            result["roofline"]["step_executed_tflops"] = brk["step_executed_tflops"]
            result["roofline"]["families"] = brk["families"]
            result["roofline"]["families_source"] = brk["source"]
        else:
            result["roofline"]["step_executed_frac"] = None
            result["roofline"]["families_source"] = brk_src

    if rank == 0 and world == 1 and not args.no_other_configs:
        result["other_configs"] = other_configs(dev)
    if not args.no_train:  # every rank: under world > 1 the iteration holds a collective
        try:
            tr = train_leg(dev, with_cpu=world == 1 and not args.no_cpu_baseline, world=world, rank=rank, rehearse=rehearse)
        except Exception as e:  # noqa: BLE001 -- the headline above must still be reported (every rank takes this branch
            # together: the leg is the same code on every rank, a failure here is not a reason to lose the sampling result)
            if world == 1:
                raise
            tr = {"error": f"{type(e).__name__}: {e}"[:400]}
        if rank == 0:
            result["train_step"] = tr
        if world == 1 and "error" not in tr:  # the small-batch point VERDICT r3 asks about (an 8-GPU shard of a global batch of 128)
            t16 = train_leg(dev, with_cpu=False, batch=16)
            result["train_step"]["batch16"] = {k: t16[k] for k in ("ms_per_iteration", "images_per_s", "iterations_timed")}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.cpu_steps)
        gpu_ddpm = value if S == T else value * S / T
        result["gpu_over_cpu"] = gpu_ddpm / result["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
