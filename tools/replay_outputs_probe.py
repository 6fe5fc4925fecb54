import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from yolo_puncture_amd.engine import Engine
from yolo_puncture_amd.weights import synthetic_state
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = Engine("s", 80, False, "bf16", 0, state=synthetic_state("s", 80, False))
im = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8).cuda()
out = eng.forward(im)
torch.cuda.synchronize()
ref = out["det"].clone()
eng.set_graph(True)
for it in range(12):
    again = eng.forward(im)
    torch.cuda.synchronize()
    print(it, bool(torch.equal(again["det"], ref)), flush=True)
print("done")
