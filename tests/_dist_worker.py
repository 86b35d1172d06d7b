"""Worker for test_dist_gloo.py: the N>1 harness of bench.py (sharding, barrier, max-over-ranks timing) on the
CPU with the gloo backend.  The per-unit work is the CPU oracle (the HIP path cannot run without a GPU)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def main():
    out_path = sys.argv[1]
    total_units = int(sys.argv[2])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    N, moduli = 1 << 10, list(pkg.params.Qi60()[-2:])
    oc = oracle.Context(N, moduli)
    start, count = bench.shard_units(total_units, rank, world)
    # unit g (global index) has its own seeded input: the shards of all ranks tile the single-process batch
    results = {}

    def step():
        for g in range(start, start + count):
            x = pkg.sampling.uniform_poly(moduli, N, 1, seed=1000 + g)[0]
            results[g] = int(oc.ntt(x).sum(dtype=np.uint64))

    def barrier():
        dist.barrier()

    def all_max(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    seconds, _ = bench.timed_region(step, steps=2, warmup=1, sync=lambda: None, barrier=barrier, all_max=all_max)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, start, count, results, seconds))
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"world": world, "parts": [[r, s, c, {str(k): v for k, v in res.items()}, t] for r, s, c, res, t in gathered]}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
