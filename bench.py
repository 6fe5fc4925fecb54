#!/usr/bin/env python3
"""bench.py - images/sec of the YOLOv10-S 640x640 bs=32 predict hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of 32 synthetic 640x640 uint8 BGR frames per GPU that are
already resident in HBM: stem (u8->bf16, /255, BGR->RGB) -> backbone -> neck -> one-to-one head -> DFL decode ->
two-stage top-k -> [32,300,6]; at N>1 followed by the single RCCL all-gather of the detections. Frames are
sharded across ranks (weak scaling: 32 frames per GPU). Rank 0 prints the result as ONE JSON line - the LAST line of stdout; when
the roofline / cpu_baseline legs are on, the bare measurement is printed first (marked "partial") so that a fault in a diagnostic
leg cannot cost it.
"""
from __future__ import annotations

import argparse
import faulthandler
import json
import os
import sys
import time

faulthandler.enable()       # a native fault prints the interpreter stack (libyolop adds the native backtrace in front of it)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step (BASELINE config: 32)")
    ap.add_argument("--variant", default="s")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--seg", action="store_true", help="config 5: S-seg trunk + proto (mask tail not in the step)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-lanes", action="store_true", help="hipGraph without the concurrent head-branch lanes (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the all-gather even with one rank (rehearsal of the N>1 path on a 1-GPU box; launch under torch.distributed.run)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=0, help="batches in flight per GPU (parallel.EngineRing: that many engines take the steps round-robin); 1 = one engine, one step after the other; "
                    "0 (default) = 2 if an untimed trial before the warm-up finds it faster on this box than 1, else 1")
    ap.add_argument("--no-one-in-flight", action="store_true", help="skip the comparison leg that repeats the steps with one batch in flight")
    ap.add_argument("--no-dense-head", action="store_true", help="skip the comparison leg that times the same step with every head branch dense")
    ap.add_argument("--no-steady", action="store_true", help="skip the >= 200-step repetition of the timed loop (spread of the step time)")
    ap.add_argument("--no-spread", action="store_true", help="skip the leg with equal class biases on the three levels (winners on P3 / P4 / P5)")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames in the bounded CPU-baseline sample")
    return ap.parse_args()


def op_traffic() -> dict:
    """HBM bytes per launch, keyed by OP NAME (profiles/op_traffic.json, written by tools/op_traffic.py from two rocprofv3 --pmc passes of
    this workload - FETCH_SIZE and WRITE_SIZE collected separately, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
    coalesced reads on gfx950; every op's dispatches are bracketed by marker kernels, so the rows do not depend on which tile
    configuration - i.e. which device symbol - the tuner picked on the profiled box). {} when the file is absent."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "op_traffic.json")))
        return {k: int(v) for k, v in t["ops"].items()}
    except Exception:
        return {}


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a 1-GPU box gets a share
    of a 256-thread host; oversubscribing torch's thread pool there is 20x slower than matching the share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("YOLOP_CPU_THREADS", "32"))))


def cpu_baseline(variant: str, seg: bool, imgsz: int, frames: int):
    """The oracle (torch-CPU fp32 restatement of the reference's path; the reference's own `ultralytics` path
    cannot be installed) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle.yolov10_oracle import Oracle
    from yolo_puncture_amd.weights import synthetic_state
    cores = usable_cores()
    torch.set_num_threads(cores)
    st = synthetic_state(variant, 80, seg, seed=0)
    orc = Oracle(st, variant, 80, seg, "fp32")
    g = torch.Generator().manual_seed(0)
    im = torch.randint(0, 256, (frames, imgsz, imgsz, 3), dtype=torch.uint8, generator=g)
    with torch.no_grad():
        orc.forward(im[:2])                       # warm-up (oneDNN primitive creation)
        reps, t0 = 0, time.perf_counter()
        while True:
            orc.forward(im)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > 10.0 or reps >= 5:
                break
    return dict(value=round(frames * reps / dt, 2), unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle fp32 (torch-CPU), YOLOv10-{variant.upper()} {imgsz}x{imgsz}, {frames} frames x {reps} passes, "
                       f"one-to-one head only; host reports {os.cpu_count()} cpus, {cores} usable")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    from yolo_puncture_amd.engine import Engine
    from yolo_puncture_amd.weights import synthetic_state

    B, S = a.batch, a.imgsz
    st = synthetic_state(a.variant, 80, a.seg, seed=0)         # seeded synthetic weights (no checkpoint offline)
    eng = Engine(a.variant, 80, a.seg, a.dtype, local, state=st)
    g = torch.Generator().manual_seed(rank)                    # rank r holds frames [r*B, (r+1)*B)
    frames = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).to(dev)
    out = dict(det=torch.empty((B, 300, 6), dtype=torch.float32, device=dev),
               idx=torch.empty((B, 300), dtype=torch.int32, device=dev),
               coeff=torch.empty((B, 300, 32), dtype=torch.float32, device=dev) if a.seg else None)
    # N>1: two detection buffers so that the all-gather of step i (RCCL stream) overlaps the forward of step i+1
    # (the 230 KB of detections are copied to alternating send buffers)
    send = [torch.empty_like(out["det"]) for _ in range(2)] if use_dist else None
    gath = [torch.empty((world * B, 300, 6), dtype=torch.float32, device=dev) for _ in range(2)] if use_dist else None
    pending = [None, None]
    nstep = [0]
    if use_dist:
        # every rank runs rank 0's tile configurations, so a frame's bf16 detections do not depend on the rank it was sharded to
        from yolo_puncture_amd.parallel import sync_tuning
        sync_tuning(eng, (B, S, S), frames)
    eng.set_graph(not a.no_graph)
    if a.no_lanes and not a.no_graph:
        eng._chk(eng.lib.yp_set_graph(eng._h, 2))
    # several batches in flight (parallel.EngineRing): engine j takes steps j, j+n, ... on its own stream, so that the low-occupancy end of
    # one step (20x20 layers, top-k) runs beside the start of the next. Every engine has its own arena, graph and output buffers and the
    # same tile configurations; every step is a complete forward + post-process of its 32 frames.
    auto_fl = a.in_flight <= 0 and not a.no_graph
    n_fl = (2 if auto_fl else max(1, a.in_flight)) if not a.no_graph else 1
    from yolo_puncture_amd.parallel import EngineRing
    ring_note = None
    try:
        ring = EngineRing([eng] + [Engine(a.variant, 80, a.seg, a.dtype, local, state=st) for _ in range(n_fl - 1)])
        if n_fl > 1:
            ring.prepare(frames)                           # engine 0's tile configurations to the others, every graph captured (without lanes)
    except Exception as ex:
        if not auto_fl or use_dist:                        # (asked for explicitly, or other ranks would go on without this one)
            raise
        ring_note = f"second engine unavailable ({type(ex).__name__}: {ex}): one batch in flight"
        ring, n_fl, auto_fl = EngineRing([eng]), 1, False
        eng.set_graph(2 if a.no_lanes else True)
    outs = [out] + [{k: (torch.empty_like(v) if v is not None else None) for k, v in out.items()} for _ in range(n_fl - 1)]
    lanes = ring.streams if n_fl > 1 else [None]

    def step_on(j):
        ring.engines[j].forward(frames, outs[j])
        if use_dist:
            i = nstep[0] & 1
            if pending[i] is not None:
                pending[i].wait()                   # the gather that last used this pair of buffers has finished
            send[i].copy_(outs[j]["det"])
            pending[i] = dist.all_gather_into_tensor(gath[i], send[i], async_op=True)

    def lanes_mode(on):                                    # engine 0 alone replays a graph with concurrent head lanes, in a ring one chain
        if not a.no_graph:
            eng.set_graph(1 if (on and not a.no_lanes) else 2)

    def step():
        j = nstep[0] % n_fl
        if n_fl > 1:
            with torch.cuda.stream(lanes[j]):
                step_on(j)
        else:
            step_on(0)
        nstep[0] += 1

    def sync():
        for h in pending:
            if h is not None:
                h.wait()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # the steps run under a side stream: the engine replays its graph on the caller's stream when that is not the legacy null stream
    # (no hop to a stream of its own and back between steps: -25 us per step)
    side = torch.cuda.Stream(dev)
    fl_trial = None
    if auto_fl:
        # how well two batches overlap depends on how the runtime maps the streams in play onto its hardware queues (with 8 queues instead
        # of the default 4 two in flight are SLOWER than one on the same box): a short untimed trial decides, all ranks together
        fl_trial = {}
        with torch.cuda.stream(side):
            for cand in (2, 1):
                n_fl = cand
                lanes_mode(cand == 1)
                nstep[0] = 0
                for _ in range(4):
                    step()
                sync()
                t0 = time.perf_counter()
                for _ in range(16):
                    step()
                sync()
                tt = time.perf_counter() - t0
                if use_dist:
                    t = torch.tensor([tt], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    tt = float(t.item())
                fl_trial[cand] = round(tt / 16 * 1e3, 4)
        n_fl = 2 if fl_trial[2] < fl_trial[1] else 1
        lanes_mode(n_fl == 1)
        nstep[0] = 0
    with torch.cuda.stream(side):
        for _ in range(a.warmup):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync()
        dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    value = world * B * a.steps / dt
    one_fl = None
    if n_fl > 1 and rank == 0 and world == 1 and not a.no_one_in_flight:
        # the same K steps with ONE batch in flight (engine 0 alone, one step after the other): what a caller without a next batch ready sees
        lanes_mode(True)
        with torch.cuda.stream(side):
            for _ in range(max(a.warmup, 2)):
                eng.forward(frames, out)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                eng.forward(frames, out)
            torch.cuda.synchronize(dev)
            dt1 = time.perf_counter() - t0
        one_fl = {"ms_per_step": round(dt1 / a.steps * 1e3, 4), "value": round(B * a.steps / dt1, 1)}

    head_note = {"box_branch": "dense (YOLOP_DENSE_HEAD=1)" if os.environ.get("YOLOP_DENSE_HEAD") == "1" else
                 "evaluated at the positions the top-k winners' 3x3 neighbourhoods cover (head_branch.hip): the rows v10postprocess gathers, "
                 "same values as the dense maps up to fp32 summation order; dense_head = the same step with every branch dense"}

    failures = []                                              # cross-checks that did not hold: reported in the line, non-zero exit at the end
    steady = None

    def line(roof, cpu, partial):
        d = {
            "metric": "images/sec", "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"YOLOv10-{a.variant.upper()}{'-seg' if a.seg else ''} {S}x{S} bs={B}/GPU {a.dtype}, "
                                   f"u8 frames resident in HBM -> [B,300,6] detections"
                                   + (f", {n_fl} batches in flight" if n_fl > 1 else "")
                                   + (", RCCL all-gather of detections" if world > 1 else ""),
                       "global_batch": world * B, "imgsz": S, "parallelism": f"frame-shard x{world}",
                       "weights": "seeded synthetic (SURVEY 8d)", "hipgraph": not a.no_graph,
                       "tile_configurations": eng.tuning_source(),
                       "in_flight": n_fl, "in_flight_trial_ms": fl_trial if ring_note is None else ring_note, "one_in_flight": one_fl,
                       "steady": steady, "head": head_note, "checks": "ok" if not failures else failures},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if partial:
            d["partial"] = "measured line; the same line with roofline and cpu_baseline follows"
        return json.dumps(d)

    want_roof = rank == 0 and not a.no_roofline
    want_cpu = rank == 0 and world == 1 and not a.no_cpu_baseline
    if rank == 0 and (want_roof or want_cpu):
        # the measurement is on stdout before any diagnostic leg runs: a fault in the per-op profile or the CPU baseline cannot cost it.
        # The LAST line printed is the complete one.
        print(line(None, None, True), flush=True)

    if rank == 0 and world == 1 and not a.no_steady:
        # the contract's K steps are a short region (20 steps = 30 ms): the same loop again in blocks, >= 200 steps in all, for the spread
        blocks, per = 10, max(20, a.steps)
        ts = []
        lanes_mode(n_fl == 1)                                  # (the one-in-flight leg above left engine 0 in lanes mode)
        with torch.cuda.stream(side):
            for _ in range(blocks):
                sync()
                t0 = time.perf_counter()
                for _ in range(per):
                    step()
                sync()
                ts.append((time.perf_counter() - t0) / per * 1e3)
        steady = {"steps": blocks * per, "blocks": blocks, "ms_per_step_mean": round(sum(ts) / len(ts), 4), "ms_per_step_min": round(min(ts), 4),
                  "ms_per_step_max": round(max(ts), 4), "value_mean": round(B * 1e3 / (sum(ts) / len(ts)), 1)}

    roof = None
    if want_roof:
        try:
            # per-op HIP event pairs on the launch stream (eager replay of the same plan), after the timed region
            hp = eng.head_positions()                     # the last timed forward's winners-only head: positions / winners per level
            eng.set_graph(False)
            prof = eng.profile(frames, iters=5)
            peak_tf = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
            ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
            traffic_by_op = op_traffic()
            if hp is not None:
                # the head op's algorithmic work from the run's own counters (the plan's figure assumes a third of the winners per level)
                cin = {int(o["name"].split(".")[3]): o["in"][2] for o in prof if o["name"].startswith("model.23.one2one_cv2.") and o["name"].endswith(".0")}
                cmid, cout = 64, 64
                hfl = sum(2.0 * (hp["positions"][l] * 9.0 * cin[l] * cmid + hp["winners"][l] * (9.0 * cmid * cmid + cmid * cout)) for l in range(3))
                hby = sum(hp["positions"][l] * (9.0 * cin[l] * 2 + 2.0 * cmid * 2) + hp["winners"][l] * cout * 4 + (9.0 * cin[l] * cmid + 9.0 * cmid * cmid + cmid * cout) * 2 for l in range(3))
                for o in prof:
                    if o["name"] == "model.23.postprocess":
                        base_by = sum(q["bytes"] for q in prof if q["name"].startswith("model.23.amax.")) + B * 300 * (80 + 64 + 6 + 1) * 4
                        o["flops"], o["bytes"] = hfl, hby + base_by
            layers, by_kernel = [], {}
            for o in prof:
                if o["kernel"] == "-":       # work done by a fused consumer: no launch of its own
                    continue
                sec = o["ms"] * 1e-3
                ai = o["flops"] / max(o["bytes"], 1.0)
                bound_tf = min(peak_tf, ai * PEAK_HBM_GBS * 1e-3)              # the north star's definition: min(P, AI x BW), TFLOP/s
                if o["flops"] > 0:
                    frac = (o["flops"] / sec / 1e12) / bound_tf
                else:                                                           # pure data movement (pools, class-max): bytes against HBM
                    frac = (o["bytes"] / sec / 1e9) / PEAK_HBM_GBS
                tr = traffic_by_op.get(o["name"])
                layers.append(dict(op=o["name"], kernel=o["kernel"].split("<")[0], us=round(o["ms"] * 1e3, 2), gflop=round(o["flops"] / 1e9, 3),
                                   mb=round(o["bytes"] / 1e6, 2), bound="mfma" if ai >= ridge else "hbm", frac=round(frac, 3),
                                   traffic_mb=None if tr is None else round(tr / 1e6, 2)))
                k = by_kernel.setdefault(o["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, n=0, traffic=0, tn=0))
                k["ms"] += o["ms"]; k["flops"] += o["flops"]; k["bytes"] += o["bytes"]; k["n"] += 1
                if tr is not None:
                    k["traffic"] += tr; k["tn"] += 1
            total_ms = sum(k["ms"] for k in by_kernel.values())
            # whole-step fraction: the time the step would take with every layer AT its bound, over the time it takes
            bound_ms = sum(max(o["flops"] / (peak_tf * 1e12), o["bytes"] / (PEAK_HBM_GBS * 1e9)) * 1e3 for o in prof if o["kernel"] != "-")
            dom = max(by_kernel, key=lambda n: by_kernel[n]["ms"])      # the single device symbol with the most time
            d = by_kernel[dom]
            ai = d["flops"] / max(d["bytes"], 1.0)
            avg_ms = d["ms"] / d["n"]
            traffic = round(d["traffic"] / d["tn"]) if d["tn"] == d["n"] else None      # mean over this symbol's ops, when every one was measured
            if ai >= ridge:
                ach = d["flops"] / d["n"] / (avg_ms * 1e-3) / 1e12
                roof = dict(bound="mfma", achieved=round(ach, 2), peak=peak_tf, unit="TFLOP/s", frac=round(ach / peak_tf, 4), traffic=traffic)
            else:
                ach = d["bytes"] / d["n"] / (avg_ms * 1e-3) / 1e9
                roof = dict(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4), traffic=traffic)
            # the conv family as a whole (all instantiations of the LDS-DMA MFMA conv kernels), for orientation
            fam = [v for n, v in by_kernel.items() if n.startswith("conv_")]
            fam_ms, fam_fl, fam_by = sum(v["ms"] for v in fam), sum(v["flops"] for v in fam), sum(v["bytes"] for v in fam)
            roof.update(kernel=dom, ops=[o["name"] for o in prof if o["kernel"] == dom], launches_per_step=d["n"], avg_launch_ms=round(avg_ms, 5),
                        alg_bytes_per_launch=round(d["bytes"] / d["n"]), alg_flops_per_launch=round(d["flops"] / d["n"]),
                        flop_per_byte=round(ai, 1), tflops=round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2),
                        share_of_step=round(d["ms"] / total_ms, 3), eager_step_ms=round(total_ms, 3),
                        step_frac=round(bound_ms / total_ms, 4),
                        traffic_source="profiles/op_traffic.json (by op name)" if traffic_by_op else None,
                        head_positions=hp,
                        conv_family=dict(ms=round(fam_ms, 4), share_of_step=round(fam_ms / total_ms, 3),
                                         tflops=round(fam_fl / max(fam_ms, 1e-9) / 1e9, 1), gbs=round(fam_by / max(fam_ms, 1e-9) / 1e6, 0)),
                        per_layer=layers)
        except Exception as ex:      # (a failed diagnostic pass must not cost the measured line)
            roof = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0 and world == 1 and not a.no_dense_head and not a.no_graph and a.dtype == "bf16" and os.environ.get("YOLOP_DENSE_HEAD") != "1":
        # comparison leg, outside the timed region: the same step with the box branch dense over all 8400 anchors (what round 2 measured),
        # and the check that the winners-only head returns the dense head's detections on the TIMED network and frames. The dense engine
        # runs the timed engine's tile configurations (bf16 scores depend on the fp32 summation order of the class branch; with its own
        # autotune the dense engine's top-300 sets differ - that is what BENCH_r03's `false / 610 px` was). A mismatch fails the run.
        try:
            cfgs = eng.tuning_export()
            os.environ["YOLOP_DENSE_HEAD"] = "1"                # read by yp_create
            try:
                de = Engine(a.variant, 80, a.seg, a.dtype, local, state=st)
            finally:
                del os.environ["YOLOP_DENSE_HEAD"]
            de.tuning_import(B, S, S, cfgs)
            out_d = {k: (torch.empty_like(v) if v is not None else None) for k, v in out.items()}
            out_w = {k: (torch.empty_like(v) if v is not None else None) for k, v in out.items()}
            de.set_graph(True)
            eng.set_graph(True)
            with torch.cuda.stream(side):
                eng.forward(frames, out_w)                      # the timed engine once more, into buffers nothing else writes
                for _ in range(max(a.warmup, 3)):
                    de.forward(frames, out_d)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    de.forward(frames, out_d)
                torch.cuda.synchronize(dev)
                dt_d = time.perf_counter() - t0
            same = bool(torch.equal(out_w["idx"], out_d["idx"]) and torch.equal(out_w["det"][..., 4:], out_d["det"][..., 4:]))
            dbox = float((out_w["det"][..., :4] - out_d["det"][..., :4]).abs().max())
            same_timed = bool(torch.equal(out_w["det"], out["det"]) and torch.equal(out_w["idx"], out["idx"]))   # what the timed steps left in `out`
            head_note["dense_head"] = {"ms_per_step": round(dt_d / a.steps * 1e3, 4), "value": round(B * a.steps / dt_d, 1),
                                       "tile_configurations": "imported from the timed engine",
                                       "same_anchors_classes_scores": same, "max_box_difference_px": round(dbox, 4),
                                       "timed_steps_left_the_same_detections": same_timed}
            if not same or dbox > 2.0 or not same_timed:
                failures.append("winners-only head differs from the dense head on the timed workload")
            de.close()
        except Exception as ex:
            head_note["dense_head"] = {"error": f"{type(ex).__name__}: {ex}"}
            failures.append("dense-head cross-check did not run")

    if rank == 0 and world == 1 and not a.no_spread and not a.no_graph and a.dtype == "bf16":
        # the bench network (ultralytics' bias_init: class bias log(5/nc/(640/stride)^2)) puts all 300 winners of a frame on P5, the cheapest
        # case for the winners-only head (work = min(9 x winners, H x W) positions per level). Same network with EQUAL class biases on the
        # three levels: winners spread over P3 / P4 / P5 as a trained detector's do. One batch in flight, graph with lanes.
        try:
            st2 = synthetic_state(a.variant, 80, a.seg, seed=0, cls_bias=-6.0)
            e2 = Engine(a.variant, 80, a.seg, a.dtype, local, state=st2)
            e2.tuning_import(B, S, S, eng.tuning_export())
            out2 = {k: (torch.empty_like(v) if v is not None else None) for k, v in out.items()}
            e2.set_graph(True)
            with torch.cuda.stream(side):
                for _ in range(max(a.warmup, 3)):
                    e2.forward(frames, out2)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    e2.forward(frames, out2)
                torch.cuda.synchronize(dev)
                dt2 = time.perf_counter() - t0
            head_note["spread_winners"] = {"ms_per_step": round(dt2 / a.steps * 1e3, 4), "value": round(B * a.steps / dt2, 1), "in_flight": 1,
                                           "class_bias": "-6.0 on every level", "head": e2.head_positions(),
                                           "compare_with": "one_in_flight (same mode, the bench network)"}
            e2.close()
        except Exception as ex:
            head_note["spread_winners"] = {"error": f"{type(ex).__name__}: {ex}"}

    cpu = None
    if want_cpu:
        try:
            cpu = cpu_baseline(a.variant, a.seg, S, a.cpu_frames)
        except Exception as ex:          # the timed result above stands on its own: report the failure instead of losing the line
            cpu = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0:
        print(line(roof, cpu, False), flush=True)
    if use_dist:
        if rank == 0 and not (torch.equal(gath[0][:B], out["det"]) and torch.equal(gath[1][:B], out["det"])):
            raise SystemExit("all-gather returned different detections than the local shard")
        dist.barrier()
        dist.destroy_process_group()
    for e2 in ring.engines:
        e2.close()
    if failures:
        raise SystemExit("bench.py: " + "; ".join(failures) + " (the measured line above stands; this exit status marks the failed cross-check)")


if __name__ == "__main__":
    main()
