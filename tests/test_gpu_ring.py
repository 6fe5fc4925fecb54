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
