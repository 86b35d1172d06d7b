"""Batch sharding of independent units (polynomials, ciphertexts) across one process per GPU, and the one
collective the path has: the gather of results (BASELINE.json config 5, "sharded 8 x MI355X with RCCL gather").

The reference's model is a pool of workers over independent ciphertext products (examples/dbfv/psi/psi.go:215-233):
no arithmetic crosses units, so the only data that moves between devices is finished results.  Partitioning is by
contiguous blocks of the global unit index; ``gather_blocks`` brings the blocks to one rank in global order with
``torch.distributed.gather`` -- on the ``nccl`` backend (RCCL) that is one grouped send / receive per rank pair, i.e.
direct peer-to-peer copies into the root over xGMI rather than a ring (a ring all-gather is bound by one link and moves
every block through every rank).

Host logic only; ``torch`` is imported lazily so the module loads wherever the package does.  The same functions run under
``gloo`` on CPU tensors (tests/test_dist_gloo.py).
"""


def shard_units(total, rank, world):
    """Contiguous block partition of `total` independent units: (start, count) for `rank`."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def unit_owner(unit, total, world):
    """The rank whose block holds global unit index `unit`."""
    base, extra = divmod(total, world)
    edge = extra * (base + 1)
    if unit < edge:
        return unit // (base + 1)
    return extra + (unit - edge) // base


def gather_blocks(local, total, rank, world, dst=0):
    """Gather the ranks' blocks of per-unit results to rank `dst` in global unit order.

    local: tensor [count_of_this_rank, ...] (device tensor under nccl, CPU tensor under gloo), contiguous.
    Returns a tensor [total, ...] on `dst` and None elsewhere.  With world == 1 and no process group it is the
    identity.  Blocks are padded to the largest block for the collective (at most one unit per rank) and trimmed on arrival.
    """
    import torch
    import torch.distributed as dist

    start, count = shard_units(total, rank, world)
    if local.shape[0] != count:
        raise ValueError("rank %d holds %d units, its block has %d" % (rank, local.shape[0], count))
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local
    biggest = -(-total // world)
    send = local.contiguous()
    if count < biggest:
        pad = torch.zeros((biggest - count,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], dim=0)
    if rank == dst:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if total % world == 0:
            # equal blocks: the peers write straight into their place of the result (no staging copy on the root)
            dist.gather(send, gather_list=[out[r * biggest:(r + 1) * biggest] for r in range(world)], dst=dst)
            return out
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=recv, dst=dst)
        for r in range(world):
            s, c = shard_units(total, r, world)
            out[s:s + c] = recv[r][:c]
        return out
    dist.gather(send, gather_list=None, dst=dst)
    return None
