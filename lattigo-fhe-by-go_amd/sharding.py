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


def all_gather_blocks(local, total, rank, world):
    """Every rank receives every rank's block in global unit order ("ncclAllGather if every rank needs all outputs").
    Equal blocks go through one ``all_gather_into_tensor``; unequal ones are padded to the largest block and trimmed on arrival.
    Note the cost on xGMI: every block crosses every rank's links, seven times the root gather's traffic per link."""
    import torch
    import torch.distributed as dist

    start, count = shard_units(total, rank, world)
    if local.shape[0] != count:
        raise ValueError("rank %d holds %d units, its block has %d" % (rank, local.shape[0], count))
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local
    biggest = -(-total // world)
    send = local.contiguous()
    if total % world == 0:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, send)
        return out
    if count < biggest:
        pad = torch.zeros((biggest - count,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], dim=0)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        s, c = shard_units(total, r, world)
        out[s:s + c] = recv[r][:c]
    return out


class ChunkedGather:
    """The gather of results overlapped with the computation that produces them ("overlapped with the tail of
    compute").  A rank's block is produced in chunks; as soon as a chunk is final on the current stream it is handed to
    ``submit``, which starts an asynchronous ``torch.distributed.gather`` of that chunk from every rank straight into its place of
    the result on `dst` (zero-copy views, so the blocks must be equal: total % world == 0).  On the nccl backend (RCCL) the
    collective runs on the communicator's stream behind an event of the current stream, i.e. concurrently with the next chunk's
    kernels; ``wait`` joins the outstanding collectives into the current stream and returns the [total, ...] result on `dst`
    (None elsewhere).  Every rank must submit the same chunk boundaries in the same order.
    """

    def __init__(self, local, total, rank, world, dst=0, out=None):
        import torch
        if total % world != 0:
            raise ValueError("ChunkedGather needs equal blocks (total %d, world %d); use gather_blocks" % (total, world))
        self.local, self.total, self.rank, self.world, self.dst = local, total, rank, world, dst
        self.per = total // world
        if local.shape[0] != self.per:
            raise ValueError("rank %d holds %d units, its block has %d" % (rank, local.shape[0], self.per))
        self.out = None
        if rank == dst:
            self.out = out if out is not None else torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            if tuple(self.out.shape) != (total,) + tuple(local.shape[1:]):
                raise ValueError("result buffer has the wrong shape")
        self.works = []

    def submit(self, u0, nb):
        """units [u0, u0 + nb) of this rank's block are final (on the current stream under nccl)"""
        import torch.distributed as dist
        if u0 < 0 or nb <= 0 or u0 + nb > self.per:
            raise ValueError("chunk [%d, %d) outside the block of %d units" % (u0, u0 + nb, self.per))
        send = self.local[u0:u0 + nb]
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            if self.out.data_ptr() != self.local.data_ptr():
                self.out[u0:u0 + nb].copy_(send)
            return
        if self.rank == self.dst:
            slots = [self.out[r * self.per + u0:r * self.per + u0 + nb] for r in range(self.world)]
            self.works.append(dist.gather(send, gather_list=slots, dst=self.dst, async_op=True))
        else:
            self.works.append(dist.gather(send, gather_list=None, dst=self.dst, async_op=True))

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        return self.out
