"""The MulRelin pipeline is capturable into a HIP graph: after a warm-up call (pools at their size, tables built) a call enqueues
kernels and memsets only, on the stream the contexts are set to, so torch.cuda.CUDAGraph can record and replay it
(tests/_graph_capture_worker.py, in its own process because torch's HIP runtime has to come up before the library's)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mulrelin_replays_from_a_hip_graph(gpu_pkg):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_graph_capture_worker.py")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert "graph replay ok" in res.stdout
