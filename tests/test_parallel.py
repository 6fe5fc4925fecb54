"""The N>1 path on CPU: two gloo ranks shard a batch of frames and all-gather their detections in rank order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolo_puncture_amd.parallel import gather_detections, shard_range


def test_shard_range_covers_everything():
    for n, w in ((256, 8), (10, 4), (3, 8), (32, 1)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = torch.arange(8 * 300 * 6, dtype=torch.float32).view(8, 300, 6)      # "detections" of 8 frames
    lo, hi = shard_range(8, rank, world)
    out = gather_detections(frames[lo:hi].clone())
    q.put((rank, bool(torch.equal(out, frames))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_restores_frame_order():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


class _FakeEngine:
    """Stands in for Engine in the world_size-2 CPU test: records what sync_tuning asks of it."""
    def __init__(self, rank):
        self.rank, self.imported, self.forwards = rank, None, 0

    def forward(self, frames):
        self.forwards += 1

    def tuning_export(self):
        return [-1, 304, 412, 700 + self.rank, 601]

    def tuning_import(self, B, H, W, cfgs):
        self.imported = (B, H, W, list(cfgs))


def _tune_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolo_puncture_amd.parallel import sync_tuning
    e = _FakeEngine(rank)
    sync_tuning(e, (4, 64, 96), torch.zeros(4, 64, 96, 3, dtype=torch.uint8))
    q.put((rank, e.forwards, e.imported))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tuning_broadcast():
    """rank 0 tunes (one forward) and its per-op configuration ids reach rank 1 unchanged; rank 1 neither tunes nor exports."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tune_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res[0] == (0, 1, None)
    assert res[1] == (1, 0, (4, 64, 96, [-1, 304, 412, 700, 601]))
