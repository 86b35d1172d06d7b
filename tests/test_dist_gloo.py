"""The multi-GPU path shards independent units across ranks in contiguous blocks and gathers the results to rank 0
(BASELINE.json config 5).  Exercised here with world_size 2 and 3 on the gloo backend: partition, timing
harness, and that the gathered result of global unit g sits at index g whatever rank computed it."""
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


def test_partition_covers_every_unit_once(pkg):
    sh = pkg.sharding
    for total in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = sh.shard_units(total, r, world)
                seen += list(range(s, s + c))
                for g in range(s, s + c):
                    assert sh.unit_owner(g, total, world) == r
            assert seen == list(range(total))
    # config 5: 1024 products on 8 GPUs = 128 each, rank r owns [128 r, 128 (r + 1))
    assert [pkg.sharding.shard_units(1024, r, 8) for r in (0, 7)] == [(0, 128), (896, 128)]


@pytest.mark.parametrize("world,total", [(2, 7), (2, 8), (3, 8)])
def test_sharded_batch_gathers_in_global_order(tmp_path, oracle, pkg, world, total):
    out = tmp_path / "out.json"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), str(total)]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    data = json.load(open(out))
    assert data["world"] == world
    covered = 0
    times = set()
    for rank, start, count, t in data["parts"]:
        assert (start, count) == pkg.sharding.shard_units(total, rank, world)
        covered += count
        times.add(round(t, 9))
    assert covered == total
    assert len(times) == 1                      # every rank reports the same max-over-ranks time
    # result placement: index g of the gathered array is the transform of global unit g (single-process reference)
    gathered = np.load(str(out) + ".npy")
    N, moduli = 1 << 10, list(pkg.params.Qi60()[-2:])
    oc = oracle.Context(N, moduli)
    assert gathered.shape == (total, len(moduli), N)
    for g in range(total):
        x = pkg.sampling.uniform_poly(moduli, N, 1, seed=1000 + g)[0]
        assert np.array_equal(gathered[g], oc.ntt(x)), g


@pytest.mark.gpu
def test_sharded_batch_on_device(tmp_path, oracle, gpu_pkg):
    """the same two-rank run with the product doing the per-unit work (both ranks on the box's one device)"""
    out = tmp_path / "out.json"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), "9"]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    data = json.load(open(out))
    assert data["on_device"] is True
    gathered = np.load(str(out) + ".npy")
    N, moduli = 1 << 10, list(gpu_pkg.params.Qi60()[-2:])
    oc = oracle.Context(N, moduli)
    for g in range(9):
        x = gpu_pkg.sampling.uniform_poly(moduli, N, 1, seed=1000 + g)[0]
        assert np.array_equal(gathered[g], oc.ntt(x)), g
