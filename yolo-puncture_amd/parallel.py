"""Frame sharding across the GPUs of one node: one process per GPU, frames split contiguously, weights replicated, one
all-gather of the [B/G,300,6] detections per batch (RCCL over xGMI when the backend is "nccl"; 230,400 B per rank at
B/G=32 - latency-bound, so ONE collective, never per-frame). The reference has no multi-GPU path at all
(SURVEY.md 2.1); frames are independent inside `predict` (reference yolo_seg/app.py:85-91), which is what makes the
path shard. Masks are not gathered."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split: rank g gets frames [lo, hi). Remainder frames go to the lowest ranks."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_detections(det_local: torch.Tensor, out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """det_local [b,300,6] (equal b on every rank) -> [world*b,300,6] in rank order, one collective."""
    if not dist.is_available() or not dist.is_initialized():
        return det_local
    world = dist.get_world_size(group)
    if world == 1:
        return det_local
    det_local = det_local.contiguous()
    if out is None:
        out = det_local.new_empty((world * det_local.shape[0],) + tuple(det_local.shape[1:]))
    if dist.get_backend(group) == "gloo":            # CPU tests: gloo has no all_gather_into_tensor for every dtype path
        parts = list(out.chunk(world, 0))
        dist.all_gather(parts, det_local, group=group)
    else:
        dist.all_gather_into_tensor(out, det_local, group=group)
    return out


def sync_tuning(engine, shape: Tuple[int, int, int], frames: Optional[torch.Tensor] = None, group=None, src: int = 0) -> None:
    """Make every rank run the SAME conv tile configurations for `shape` = (B,H,W): rank `src` tunes (one forward on `frames`,
    [B,H,W,3] uint8 on its device), its per-op configuration ids are broadcast, the other ranks install them before their first
    forward. bf16 outputs depend on the tile configuration in the last bit (fp32 summation order), and each rank's autotuner
    times its own GPU, so without this a frame's detections could depend on which rank it was sharded to. One broadcast of a
    few hundred int32 per plan, outside the data path. No-op without an initialised process group."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    B, H, W = (int(v) for v in shape)
    rank = dist.get_rank(group)
    dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
    if rank == src:
        if frames is None:
            raise ValueError("sync_tuning: the source rank needs frames to tune on")
        engine.forward(frames)
        cfgs = engine.tuning_export()
        n = torch.tensor([len(cfgs)], dtype=torch.int32, device=dev)
    else:
        n = torch.zeros(1, dtype=torch.int32, device=dev)
    dist.broadcast(n, src=src, group=group)
    buf = torch.tensor(cfgs, dtype=torch.int32, device=dev) if rank == src else torch.zeros(int(n.item()), dtype=torch.int32, device=dev)
    dist.broadcast(buf, src=src, group=group)
    if rank != src:
        engine.tuning_import(B, H, W, buf.cpu().tolist())
