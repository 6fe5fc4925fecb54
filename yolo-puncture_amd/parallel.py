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


class EngineRing:
    """Several batches in flight on ONE GPU: `n` engines (each with its own activation arena, hipGraph and stream; the same weights
    and the same tile configurations) take the submitted batches round-robin. A forward of a batch ends in kernels that cannot fill
    the chip - the 20x20 layers (100-400 workgroups on 256 CUs) and the top-k (one workgroup per frame) - and starts with ones that
    can; with a second batch in flight those phases run beside each other (YOLOv10-S, 32 frames per batch, same box:
    17.9 k img/s with one batch in flight, 20.7 k with two, 21.6 k with three). The reference calls `predict` frame by frame
    (yolo_seg/app.py:85-91) and has no counterpart; a caller that has the next batch ready uses this instead of one Engine.

    submit() orders the batch after the caller's current stream, runs it on the ring's stream and returns (outputs, event): the
    outputs are complete once the event has completed (`ring.wait(event)` makes the current stream wait for it). The frames and the
    output tensors of a submit must stay untouched until then. Results do not depend on `n` or on which engine ran a batch."""

    def __init__(self, engines):
        self.engines = list(engines)
        if not self.engines:
            raise ValueError("EngineRing needs at least one engine")
        self.streams = None
        self._k = 0
        n = len(self.engines)
        self._stage = [None] * n         # the tensor engine j's graph reads (the caller's, or _own[j])
        self._own = [None] * n           # ring-owned staging tensors, made on first need
        self._outs = [None] * n          # ring-owned outputs for submit(out=None)

    @classmethod
    def create(cls, make_engine, n: int = 2) -> "EngineRing":
        """`make_engine()` is called n times (e.g. lambda: Engine("s", 80, False, "bf16", 0, state=state))."""
        if n < 1:
            raise ValueError("EngineRing needs at least one engine")
        return cls([make_engine() for _ in range(n)])

    def prepare(self, frames: torch.Tensor) -> None:
        """Plan + tune engine 0 on `frames` ([B,H,W,3] uint8 on the GPU), hand its tile configurations to the others, capture every graph."""
        B, H, W = (int(v) for v in frames.shape[:3])
        self.engines[0].forward(frames)
        torch.cuda.synchronize(frames.device)
        cfgs = self.engines[0].tuning_export()
        for e in self.engines[1:]:
            e.tuning_import(B, H, W, cfgs)
            e.forward(frames)
        torch.cuda.synchronize(frames.device)
        # The runtime maps streams onto a few hardware queues and two graphs on ONE queue cannot overlap (tools/ring_queue_probe.py, same
        # box, same engines: default-priority streams 0 and 1 of torch's pool 1.864 ms per step = no overlap at all, streams 0 and 2
        # 1.522 ms; which pairs collide depends on what else the process has created). Queues of different PRIORITY come from different
        # pools, so neighbouring engines of the ring get streams of alternating priority: 1.524 ms whatever the pool's state.
        self.streams = [torch.cuda.Stream(frames.device, priority=-(j & 1)) for j in range(len(self.engines))]
        for e, s in zip(self.engines, self.streams):
            # with more than one batch in flight every batch's graph is ONE chain on one stream (yp_set_graph mode 2): the other batch
            # fills the CUs that a lone graph fills with its concurrent head lanes, and a batch then occupies exactly one of the runtime's
            # hardware queues - with lanes the branches of two graphs compete for the same four queues (20.7 k img/s on a good mapping,
            # 17.6 k beside an RCCL stream; without lanes 21.1 k / 20.8 k, same box)
            e.set_graph(2 if len(self.engines) > 1 else True)
            with torch.cuda.stream(s):
                e.forward(frames)
        torch.cuda.synchronize(frames.device)

    def submit(self, frames: torch.Tensor, out=None, resident: bool = False):
        """A captured graph is specialised on its input and output POINTERS (yp_forward re-captures when they change, which blocks the
        host on the engine's previous batch). So every engine of the ring reads from a staging tensor of its own: the frames are copied
        into it on the ring's stream (39 MB at 32 x 640 x 640: ~15 us). `resident=True` says that `frames` is a long-lived buffer the
        caller will hand to this ring again (one buffer per engine, in the ring's round-robin order): it is read in place and the graph
        is captured on it. Either way an engine re-captures only when the tensor it reads actually changes. Results go to `out` when
        given (keep one `out` per engine from call to call and the graph writes it directly; changing ones switch the engine to
        engine-owned results + a 0.3 MB copy-out) or, with out=None, to a per-engine output set that is returned and overwritten by that
        engine's next batch."""
        if self.streams is None:
            self.prepare(frames)
        j = self._k % len(self.engines)
        self._k += 1
        s = self.streams[j]
        s.wait_stream(torch.cuda.current_stream(frames.device))
        with torch.cuda.stream(s):
            src = frames.contiguous()
            if resident:
                self._stage[j] = src
            else:
                if self._own[j] is None or self._own[j].shape != src.shape:
                    self._own[j] = torch.empty_like(src)
                self._own[j].copy_(src, non_blocking=True)
                self._stage[j] = self._own[j]
            if out is None:
                if self._outs[j] is None or self._outs[j]["det"].shape[0] != src.shape[0]:
                    self._outs[j] = None
                res = self.engines[j].forward(self._stage[j], self._outs[j])
                self._outs[j] = res
            else:
                res = self.engines[j].forward(self._stage[j], out)
            ev = torch.cuda.Event()
            ev.record(s)
        return res, ev

    @staticmethod
    def wait(event) -> None:
        torch.cuda.current_stream().wait_event(event)

    def synchronize(self) -> None:
        for s in self.streams or []:
            s.synchronize()

    def close(self) -> None:
        self.synchronize()
        for e in self.engines:
            e.close()
        self.engines = []
