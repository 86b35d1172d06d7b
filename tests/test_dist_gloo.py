"""The multi-GPU path of bench.py shards independent polynomials across ranks with no data-path collective
(SURVEY.md 8(e)).  Exercised here with world_size 2 and 3 on the CPU (gloo)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,total", [(2, 7), (3, 8)])
def test_sharded_batch_matches_single_process(tmp_path, oracle, pkg, world, total):
    out = tmp_path / "out.json"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), str(total)]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    data = json.load(open(out))
    assert data["world"] == world
    merged = {}
    covered = 0
    times = set()
    for rank, start, count, res, t in data["parts"]:
        assert len(res) == count
        merged.update({int(k): v for k, v in res.items()})
        covered += count
        times.add(round(t, 9))
    assert covered == total and sorted(merged) == list(range(total))
    assert len(times) == 1                      # every rank reports the same max-over-ranks time
    # single-process reference on the same units
    N, moduli = 1 << 10, list(pkg.params.Qi60()[-2:])
    oc = oracle.Context(N, moduli)
    for g in range(total):
        x = pkg.sampling.uniform_poly(moduli, N, 1, seed=1000 + g)[0]
        assert merged[g] == int(oc.ntt(x).sum(dtype=np.uint64))
