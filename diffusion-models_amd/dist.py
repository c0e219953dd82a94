"""Batch-sharded sampling across the GPUs of one node: one process per GPU, every
rank samples its slice of the batch with the full weight replica, and ONE
collective (all-gather over RCCL/xGMI; gloo in CPU tests) assembles the result.

The reference never shards sampling (it runs on the main process only,
denoising_diffusion.py:1188-1198); samples are independent through the whole
loop, so there is no exchange step inside it."""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch_size: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the global batch owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(batch_size, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(batch_size: int, world_size: int) -> List[int]:
    return [shard_bounds(batch_size, world_size, r)[1] - shard_bounds(batch_size, world_size, r)[0]
            for r in range(world_size)]


def gather_shards(local: torch.Tensor, batch_size: int, group=None) -> torch.Tensor:
    """All-gather per-rank shards (dim 0) into the (batch_size, ...) tensor on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(batch_size, world)
    mx = max(sizes)
    if min(sizes) == mx and local.is_cuda and dist.get_backend(group) == "nccl":
        # equal shards: one collective straight into the result tensor
        out = local.new_empty((batch_size,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = local
    if local.shape[0] < mx:  # ragged tail: pad to the common size for the collective
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def sample_sharded(sample_fn: Callable[[int, int], torch.Tensor], batch_size: int, group=None) -> torch.Tensor:
    """`sample_fn(lo, hi)` produces samples [lo, hi) of the global batch on this rank's GPU
    (key any randomness by the GLOBAL sample index so the result does not depend on the world size:
    ``diffusion.sample(batch_size=hi - lo, seed=seed, sample_offset=lo)`` does, see :func:`sample_global`).

    A rank whose slice is empty (batch_size < world size, e.g. the remainder group of an FID run) must return a
    ``(0, ...)`` tensor from `sample_fn` -- it still enters the collective, so the other ranks do not hang."""
    if not (dist.is_available() and dist.is_initialized()):
        return sample_fn(0, batch_size)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(batch_size, world, rank)
    return gather_shards(sample_fn(lo, hi), batch_size, group)


def shared_seed(seed=None, group=None) -> int:
    """One Philox key for every rank: `seed`, or a draw from rank 0's torch generator broadcast to the group."""
    if seed is not None:
        return int(seed)
    t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
    if dist.is_available() and dist.is_initialized():
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else t.device
        t = t.to(dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.item())


def _slice_batch_kwargs(kw: dict, batch_size: int, lo: int, hi: int) -> dict:
    """Per-sample keyword tensors (dim 0 == the GLOBAL batch: ``text_emb=``, ``cond=``) cut to this rank's rows."""
    out = {}
    for k, v in kw.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == batch_size:
            v = v[lo:hi]
        out[k] = v
    return out


def _sample_shape(diffusion) -> Tuple[int, ...]:
    """(C, H, W) of what ``diffusion.sample`` returns per sample, without running it (the empty-shard result)."""
    probe = getattr(diffusion, "sample_shape", None)
    if callable(probe):
        return tuple(probe())
    (h, w), c = diffusion.image_size, diffusion.channels
    return (c, h, w)


def sample_global(diffusion, batch_size: int, seed=None, group=None, **sample_kw) -> torch.Tensor:
    """``diffusion.sample(batch_size)`` with the batch sharded over the ranks of `group` and ONE all-gather at the
    end.  Every rank draws the noise of ITS global sample indices (Philox counter = global element index), so the
    result is the same tensor for any world size, including 1.

    Conditions are per-sample data and are sharded like the batch: a keyword tensor whose dim 0 equals `batch_size`
    (``text_emb=``, ``cond=``) is sliced to the rank's rows.  A text-conditional model called WITHOUT ``text_emb=``
    would draw different random captions on every rank and for every world size, so that combination is refused here:
    draw the embeddings once (``diffusion.get_random_text_condition(batch_size, device)`` on rank 0 + broadcast, or any
    fixed tensor) and pass them in.  Ranks whose slice is empty (batch_size < world size) skip the library call and
    contribute a (0, C, H, W) tensor, so every rank still enters the collective."""
    if getattr(getattr(diffusion, "model", None), "text_condition", False) and sample_kw.get("text_emb") is None:
        raise ValueError("sample_global on a text-conditional model needs text_emb= (B, E): every rank would otherwise "
                         "draw its own random captions")
    seed = shared_seed(seed, group)

    def shard(lo: int, hi: int) -> torch.Tensor:
        if hi == lo:
            dev = getattr(diffusion, "device", "cpu")
            return torch.zeros((0,) + _sample_shape(diffusion), dtype=torch.float32, device=dev)
        return diffusion.sample(batch_size=hi - lo, seed=seed, sample_offset=lo,
                                **_slice_batch_kwargs(sample_kw, batch_size, lo, hi))

    return sample_sharded(shard, batch_size, group)
