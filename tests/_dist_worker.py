"""Worker for test_dist_gloo.py: the N>1 data path of bench.py -- contiguous blocks of independent units per rank, the
max-over-ranks timing harness, and the gather of results to rank 0 (sharding.gather_blocks, the same function bench.py's
config-5 leg calls on the nccl backend) -- with the gloo backend.

Per-unit work: the product (HIP path through the C ABI) when this process sees a device; in the build container, which has
none, the CPU oracle stands in so that partition, timing and result placement are still exercised end to end."""
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
    sharding = pkg.sharding
    N, moduli = 1 << 10, list(pkg.params.Qi60()[-2:])
    L = len(moduli)
    on_device = pkg._native.device_count() > 0 and os.environ.get("LR_DIST_FORCE_CPU") != "1"
    start, count = sharding.shard_units(total_units, rank, world)
    assert (start, count) == bench.shard_units(total_units, rank, world)
    # unit g (global index) has its own seeded input: the shards of all ranks tile the single-process batch
    x = np.stack([pkg.sampling.uniform_poly(moduli, N, 1, seed=1000 + g)[0] for g in range(start, start + count)]) \
        if count else np.zeros((0, L, N), dtype=np.uint64)
    result = [None]
    if on_device:
        ctx = pkg.ring.NewContextWithParams(N, moduli, device=0)
        src, dst = ctx.NewPoly(max(count, 1)), ctx.NewPoly(max(count, 1))
        if count:
            src.set(x)

        def step():
            ctx.NTT(src, dst)
            result[0] = dst.get().reshape(-1, L, N)[:count]
    else:
        oc = graft.load_oracle().Context(N, moduli)

        def step():
            result[0] = np.stack([oc.ntt(x[j]) for j in range(count)]) if count else np.zeros((0, L, N), dtype=np.uint64)

    def barrier():
        dist.barrier()

    def all_max(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    seconds, _ = bench.timed_region(step, steps=2, warmup=1, sync=lambda: None, barrier=barrier, all_max=all_max)
    local = torch.from_numpy(np.ascontiguousarray(result[0]).view(np.int64).reshape(count, L * N))
    gathered = sharding.gather_blocks(local, total_units, rank, world, dst=0)
    if total_units % world == 0 and count >= 2:
        # the same gather in chunks handed over as they become final (bench.py's config-5 leg overlaps it with the next chunk)
        cg = sharding.ChunkedGather(local, total_units, rank, world, dst=0)
        half = count // 2
        cg.submit(0, half)
        cg.submit(half, count - half)
        chunked = cg.wait()
        if rank == 0:
            assert torch.equal(chunked, gathered)
        else:
            assert chunked is None
    everywhere = sharding.all_gather_blocks(local, total_units, rank, world)
    assert everywhere.shape == (total_units, L * N)
    if rank == 0:
        assert torch.equal(everywhere, gathered)
    else:
        # every rank holds its own block at its place of the result
        assert torch.equal(everywhere[start:start + count], local)
    times = [None] * world
    dist.all_gather_object(times, (rank, start, count, seconds))
    if rank == 0:
        assert gathered is not None and gathered.shape == (total_units, L * N)
        np.save(out_path + ".npy", gathered.numpy().view(np.uint64).reshape(total_units, L, N))
        with open(out_path, "w") as f:
            json.dump({"world": world, "on_device": bool(on_device), "parts": times}, f)
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
