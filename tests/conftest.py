import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fatal signal inside the native library leaves its C backtrace here (pytest captures fd 2 and loses it with the process)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        os.environ.setdefault("YOLOP_SEGV_FILE", os.path.join(out, "native_backtrace.txt"))
    except OSError:
        pass


import os

import pytest


@pytest.fixture(autouse=True)
def _side_stream():
    """YOLOP_TEST_STREAM=1: run every test under a non-default torch stream (the engine launches its graph on the caller's stream
    when that is not the legacy null stream)."""
    if os.environ.get("YOLOP_TEST_STREAM") != "1":
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        yield
    torch.cuda.synchronize()
