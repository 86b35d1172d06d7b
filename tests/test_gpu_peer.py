"""Single-process multi-device exchange through the C ABI (lr_poly_copy_peer, lr_context_wait_peer_copies, lr_gather_blocks).
A GPU box has one device: producer and consumer are two contexts of that device on DIFFERENT streams, which is where the ordering the entry
points promise can go wrong -- the copies must wait for the producer's kernels and the consumer's stream for the copies -- and the placement
of the blocks is the same whatever devices the handles live on (the copy itself is hipMemcpyPeerAsync between devices, hipMemcpyAsync
within one).  tools/multi_gpu_bench.cpp is the same thing with one host thread per device."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def streams():
    hip = ctypes.CDLL("libamdhip64.so")
    made = []

    def make():
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0        # hipStreamNonBlocking
        made.append(st)
        return st.value
    yield make
    assert hip.hipDeviceSynchronize() == 0
    for st in made:
        hip.hipStreamDestroy(st)


def test_gather_places_blocks_in_order_behind_their_producers(gpu_pkg, oracle, streams):
    ring, params, sampling = gpu_pkg.ring, gpu_pkg.params, gpu_pkg.sampling
    N, Q = params.DefaultParamsQi(14)
    Q = list(Q[:4])
    root = ring.NewContextWithParams(N, Q)
    workers = [ring.NewContextWithParams(N, Q) for _ in range(3)]
    keep = []
    for w in workers:
        w.SetStream(streams())                                    # every "device" on its own stream
    counts = [5, 3, 4]
    x = [sampling.uniform_poly(Q, N, c, seed=30 + i).reshape(c, len(Q), N) for i, c in enumerate(counts)]
    oc = oracle.Context(N, Q)
    total = sum(counts)
    dst = root.NewPoly(total + 2).set(np.full((total + 2, len(Q), N), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64))     # poisoned
    root.Sync()
    blocks = []
    for w, xi, c in zip(workers, x, counts):
        src, res = w.NewPoly(c).set(xi), w.NewPoly(c)
        for _ in range(20):                                           # a queue of kernels in front of the result: the copy must wait for them
            w.NTT(src, res)
            w.InvNTT(res, res)
        w.NTT(src, res)
        blocks.append((w, res, c))
        keep.append(src)
    root.GatherBlocks(dst, blocks)                                    # (no Sync on any worker in between)
    got = dst.get()
    slot = 0
    for xi, c in zip(x, counts):
        for b in range(c):
            assert np.array_equal(got[slot + b], oc.ntt(xi[b])), (slot, b)
        slot += c
    assert (got[total:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()      # the slots behind the blocks are untouched
    for w in workers:
        w.Sync()
        w.SetStream(None)                                             # off the caller's stream before it is destroyed


def test_chunked_copies_overlap_and_the_consumer_waits_on_the_device(gpu_pkg, oracle, streams):
    """the producer hands chunks over as they are final (lr_poly_copy_peer per chunk, the next chunk's kernels behind it), into slots of a
    strided destination (a component of [ciphertext][component][limb][N]); the consumer transforms the gathered polys on ITS stream after
    lr_context_wait_peer_copies, with no host synchronisation in between"""
    ring, params, sampling = gpu_pkg.ring, gpu_pkg.params, gpu_pkg.sampling
    N, Q = params.DefaultParamsQi(13)
    Q = list(Q[:3])
    L = len(Q)
    root, worker = ring.NewContextWithParams(N, Q), ring.NewContextWithParams(N, Q)
    root.SetStream(streams())
    worker.SetStream(streams())
    units, chunk = 8, 2
    x = sampling.uniform_poly(Q, N, units, seed=77).reshape(units, L, N)
    src, res = worker.NewPoly(units).set(x), worker.NewPoly(units)
    backing = root.NewPoly(2 * units)                                 # [unit][2 components][L][N]: component 1 of every unit is the destination
    view = ring.Poly.wrap_strided(root, backing.device_ptr + L * N * 8, L, units, 2 * L)
    for u0 in range(0, units, chunk):
        sv = ring.Poly.wrap(worker, src.device_ptr + u0 * L * N * 8, L, chunk)
        rv = ring.Poly.wrap(worker, res.device_ptr + u0 * L * N * 8, L, chunk)
        worker.NTT(sv, rv)
        root.CopyPeer(view, u0, worker, res, u0, chunk)
    root.WaitPeerCopies()
    root.InvNTT(view, view)                                           # on the root's stream, behind the copies
    got = backing.get().reshape(units, 2, L, N)
    assert np.array_equal(got[:, 1], x)
    assert not got[:, 0].any()
    worker.Sync()
    for c in (root, worker):
        c.SetStream(None)


def test_peer_copy_argument_checks(gpu_pkg):
    ring, params = gpu_pkg.ring, gpu_pkg.params
    N, Q = params.DefaultParamsQi(14)
    a, b = ring.NewContextWithParams(N, Q[:4]), ring.NewContextWithParams(N, Q[:2])
    pa, pb = a.NewPoly(4), b.NewPoly(4)
    err = gpu_pkg._native.LatticeRingError
    with pytest.raises(err):
        a.CopyPeer(pa, 0, b, pb, 0, 1)                                # limb counts differ
    with pytest.raises(err):
        a.CopyPeer(pa, 3, a, a.NewPoly(4), 0, 2)                      # runs past the destination
    with pytest.raises(err):
        a.GatherBlocks(pa, [(a, a.NewPoly(3), 3), (a, a.NewPoly(3), 2)])    # the blocks do not fit
    a.CopyPeer(pa, 0, a, a.NewPoly(4), 0, 0)                          # an empty range is fine
    a.WaitPeerCopies()
    a.Sync()


def test_single_process_harness_shards_and_gathers(gpu_pkg):
    """tools/multi_gpu_bench.cpp -- a plain C++ host, one thread per device, the C ABI only -- on this box's device(s): the PN16QP1761 limb
    structure on a 2^12 ring, a block of 6 units in chunks of 4 (a short last chunk), poisoned root, placement of every block checked by
    the harness itself; the JSON line carries the keys of bench.py's config5 object"""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("multi_gpu_bench", os.path.join(root, "tools", "dbg", "multi_gpu_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(["--gpus", 8, "--units", 6, "--chunk", 4, "--steps", 2, "--warmup", 1, "--logn", 12])
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert d["placement_ok"] is True and d["placement_units_checked"] >= 4 * d["n_gpus"]
    assert d["n_gpus"] == min(8, d["devices_visible"]) and d["units_total"] == 6 * d["n_gpus"] and d["chunk"] == 4
    for key in ("value", "compute_only_value", "ms_per_step_compute_and_gather", "ms_per_step_compute_only", "gather", "gather_bytes_to_root", "roofline"):
        assert key in d, key
    assert d["value"] > 0 and d["compute_only_value"] >= d["value"] * 0.5
