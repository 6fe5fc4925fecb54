"""parallel.EngineRing: several batches in flight on one GPU give the detections one engine gives, batch by batch."""
import pytest
import torch

from helpers import make_case, rand_image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant,seg,shape,n", [("n", False, (2, 96, 128), 3), ("s", True, (3, 96, 160), 2)])
def test_engine_ring_matches_one_engine(variant, seg, shape, n):
    from yolo_puncture_amd.engine import Engine
    from yolo_puncture_amd.parallel import EngineRing
    st, im = make_case(variant, 80, seg, 0, shape)
    dev = torch.device("cuda", 0)
    batches = [im.to(dev)] + [rand_image((shape[0], shape[1], shape[2], 3), seed=100 + i).to(dev) for i in range(2 * n + 1)]
    one = Engine(variant, 80, seg, "bf16", 0, state=st)
    one.forward(batches[0])
    cfgs = one.tuning_export()
    want = [{k: v.clone() for k, v in one.forward(b).items() if v is not None} for b in batches]
    torch.cuda.synchronize()
    ring = EngineRing.create(lambda: Engine(variant, 80, seg, "bf16", 0, state=st), n)
    ring.engines[0].tuning_import(shape[0], shape[1], shape[2], cfgs)       # the same tile configurations as `one`: bit-equal results
    ring.prepare(batches[0])
    outs = [dict(det=torch.empty((shape[0], 300, 6), device=dev), idx=torch.empty((shape[0], 300), dtype=torch.int32, device=dev),
                 coeff=torch.empty((shape[0], 300, 32), device=dev) if seg else None) for _ in batches]
    handles = [ring.submit(b, o) for b, o in zip(batches, outs)]           # all in flight before anything is read
    for (res, ev), w in zip(handles, want):
        ring.wait(ev)
        torch.cuda.current_stream().synchronize()
        for k, v in w.items():
            assert torch.equal(res[k], v), k
    ring.close()
    one.close()


def test_ring_does_not_recapture_with_fresh_frame_tensors():
    """a caller that brings a NEW frames tensor per submit (the advertised use) must not pay capture + instantiate per submit.
    submit() stages the frames in a ring-owned tensor, so after the first round the capture counters stand still; results stay those of
    one engine."""
    from yolo_puncture_amd.engine import Engine
    from yolo_puncture_amd.parallel import EngineRing
    variant, seg, shape, n = "n", False, (2, 96, 128), 2
    st, im = make_case(variant, 80, seg, 0, shape)
    dev = torch.device("cuda", 0)
    one = Engine(variant, 80, seg, "bf16", 0, state=st)
    one.forward(im.to(dev))
    cfgs = one.tuning_export()
    ring = EngineRing.create(lambda: Engine(variant, 80, seg, "bf16", 0, state=st), n)
    ring.engines[0].tuning_import(shape[0], shape[1], shape[2], cfgs)
    ring.prepare(im.to(dev))
    counts = []
    for i in range(5 * n):
        fresh = rand_image((shape[0], shape[1], shape[2], 3), seed=200 + i).to(dev)      # a new tensor every time
        res, ev = ring.submit(fresh)
        ring.wait(ev)
        torch.cuda.current_stream().synchronize()
        want = one.forward(fresh)
        torch.cuda.synchronize()
        assert torch.equal(res["det"], want["det"]) and torch.equal(res["idx"], want["idx"])
        counts.append([e.graph_info()["captures"] for e in ring.engines])
    assert counts[-1] == counts[n], f"captures kept growing: {counts}"                   # steady after every engine's first staged batch
    # resident buffers (one per engine, handed to the same engine every time, declared as such) are read in place: one more capture each
    res_bufs = [rand_image((shape[0], shape[1], shape[2], 3), seed=300 + j).to(dev) for j in range(n)]
    for rep in range(4):
        for j in range(n):
            res, ev = ring.submit(res_bufs[j], resident=True)
            ring.wait(ev)
    torch.cuda.synchronize()
    after = [e.graph_info()["captures"] for e in ring.engines]
    for rep in range(3):
        for j in range(n):
            ring.submit(res_bufs[j], resident=True)
    ring.synchronize()
    assert [e.graph_info()["captures"] for e in ring.engines] == after
    ring.close()
    one.close()


def test_graph_is_a_dag_without_events_and_stream_changes_are_ordered():
    """The multi-lane forward is captured on one stream with explicit dependencies (engine.hip capture_dag): the graph must hold the
    lane schedule's edges (parallel branches: more edges into op heads than a chain has), replay bit-identically to eager launches,
    survive re-captures (new output pointers), and forwards issued on alternating streams must not overlap (same results)."""
    from yolo_puncture_amd.engine import Engine
    st, im = make_case("s", 80, True, 0, (2, 96, 160))
    dev = torch.device("cuda", 0)
    im = im.to(dev)
    eng = Engine("s", 80, True, "bf16", 0, state=st)
    want = {k: v.clone() for k, v in eng.forward(im).items() if v is not None}          # eager
    torch.cuda.synchronize()
    eng.set_graph(True)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.cuda.stream(s1):
        got = eng.forward(im)
    s1.synchronize()
    gi = eng.graph_info()
    assert gi["live"] and gi["lanes"] > 1 and gi["nodes"] > 50
    # a chain of N nodes has N-1 edges; parallel branches add edges. Exactly: the schedule's edges into the first kernel of every op +
    # the inner chains of ops that launch several kernels (nodes - launching ops)
    nops = sum(1 for o in eng.plan(2, 96, 160) if o["kernel"] != "-")
    assert gi["edges"] > gi["nodes"] - 1, gi
    assert gi["edges"] == gi["schedule_edges"] + (gi["nodes"] - nops), (gi, nops)
    for k, v in want.items():
        assert torch.equal(got[k], v), k
    # re-captures (fresh outputs each call -> two changes of pointers, then engine-owned results), alternating streams, the NULL stream
    for it in range(8):
        st_ = (s1, s2, None)[it % 3]
        if st_ is None:
            again = eng.forward(im)
        else:
            with torch.cuda.stream(st_):
                again = eng.forward(im)
        torch.cuda.synchronize()
        for k, v in want.items():
            assert torch.equal(again[k], v), (it, k)
    assert eng.graph_info()["captures"] <= 4
    # without lanes: a chain
    eng.set_graph(2)
    with torch.cuda.stream(s2):
        chain = eng.forward(im)
    torch.cuda.synchronize()
    gi2 = eng.graph_info()
    assert gi2["edges"] == gi2["nodes"] - 1, gi2
    for k, v in want.items():
        assert torch.equal(chain[k], v), k
    # back-to-back forwards on two streams WITHOUT a sync in between: the second must wait for the first (shared arena)
    eng.set_graph(True)
    outs = []
    for it in range(6):
        with torch.cuda.stream((s1, s2)[it & 1]):
            outs.append(eng.forward(im))
    torch.cuda.synchronize()
    for o in outs:
        for k, v in want.items():
            assert torch.equal(o[k], v), k
    eng.close()
