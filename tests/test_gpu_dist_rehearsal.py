"""The N > 1 code path of bench.py rehearsed on the one GPU a test box has: `--force-dist` initialises RCCL with one
rank under torch.distributed.run, runs sync_tuning, the async all-gather on alternating buffers beside the forward of the next step, and
exits non-zero when the gathered tensor is not the local shard (bench.py, end of main). No 8-GPU node is available to this repo: no
scaling curve exists (README)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_force_dist_one_rank():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "6", "--warmup", "2", "--batch", "8", "--imgsz", "320",
           "--no-cpu-baseline", "--no-roofline", "--no-dense-head", "--no-spread", "--no-steady"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])          # (a gathered tensor that differs from the local shard exits non-zero)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["global_batch"] == 8
