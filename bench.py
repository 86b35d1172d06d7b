#!/usr/bin/env python3
"""Headline benchmark: batched forward NTT on the reference's benchmark ring R15.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "NTT/s ... at N=2^15, L=16"): ring.DefaultParamsQi[15]
(ring/params.go:14: N = 2^15, 16 x 60-bit limbs), a batch of B uniform polynomials resident in HBM
(synthetic, splitmix64 -- the reference's BenchmarkRing uses NewUniformPoly, ring_benchmark_test.go:160).
One step = Context.NTT (ring/ntt.go:4) over the whole batch = ONE kernel launch of B*16 limb-NTTs.
Units are independent polynomials, so N GPUs shard the batch with no data-path collective
(weak scaling: B polys per GPU); the only collectives are the timing barrier and a max over ranks.

Prints one JSON line (rank 0).  `value` = limb-NTTs per second over all GPUs; `roofline.achieved` =
algorithmic bytes (16*N per limb-NTT) / kernel time measured with HIP events on the
launch stream.  `cpu_baseline` times the CPU oracle (C restatement of the Go algorithm; Go itself is
not installable here) on the host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def shard_units(total, rank, world):
    """Contiguous block partition of `total` independent units: (start, count) for `rank`."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def ntt_bytes(N, limbs, polys=1):
    """Algorithmic HBM bytes of `polys` poly-NTTs: read 8N + write 8N per limb."""
    return 16 * N * limbs * polys


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def timed_region(step, steps, warmup, sync, barrier, all_max, ev_start=None, ev_stop=None):
    """W untimed warm-ups, then exactly K steps bracketed by barrier + sync; returns (max-over-ranks seconds,
    this rank's device milliseconds between HIP events recorded on the launch stream around the same K steps)."""
    for _ in range(warmup):
        step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    if ev_start:
        ev_start()
    for _ in range(steps):
        step()
    dev_ms = ev_stop() if ev_stop else None
    sync()
    dt = time.perf_counter() - t0
    barrier()
    return all_max(dt), dev_ms


def cpu_baseline_ntt(N, moduli, target_seconds=12.0):
    """Times the CPU oracle's Context.NTT (serial over limbs, as ring/ntt.go:4-8) with one thread per host
    core, each on its own polynomials (the reference's goroutine-per-evaluator model,
    examples/dbfv/psi/psi.go:219-233), on a bounded sample."""
    import concurrent.futures as cf

    import numpy as np

    import __graft_entry__ as graft
    oracle = graft.load_oracle()
    pkg = graft.load_package()
    oc = oracle.Context(N, moduli)
    cores = max(1, min(os.cpu_count() or 1, 32))
    x = pkg.sampling.uniform_poly(moduli, N, 1, seed=1)[0]
    bufs = [(x.copy(), np.empty_like(x)) for _ in range(cores)]
    lib = oracle.lib()
    level = len(moduli) - 1

    def one(i):
        a, b = bufs[i]
        lib.oc_ntt_lvl(oc.h, level, a.ctypes.data, b.ctypes.data)

    t0 = time.perf_counter()
    one(0)
    t_one = time.perf_counter() - t0
    per_thread = max(1, int(target_seconds / max(t_one, 1e-6)))
    per_thread = min(per_thread, 8192)

    def work(i):
        for _ in range(per_thread):
            one(i)

    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(work, range(cores)))
        dt = time.perf_counter() - t0
    polys = cores * per_thread
    return {
        "value": polys * len(moduli) / dt,
        "unit": "limb-NTT/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d poly-NTTs (N=2^%d, %d limbs) = %d limb-NTTs, %d threads x %d polys, %.1f s; 1-thread poly-NTT %.2f ms"
                  % (polys, N.bit_length() - 1, len(moduli), polys * len(moduli), cores, per_thread, dt, t_one * 1e3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="polynomials per GPU")
    ap.add_argument("--logn", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the InvNTT / MulCoeffsMontgomery / ModUp timings")
    ap.add_argument("--no-ckks", action="store_true", help="skip the CKKS MulRelin leg")
    ap.add_argument("--ckks-batch", type=int, default=128)
    args = ap.parse_args()

    import numpy as np
    import torch

    import __graft_entry__ as graft

    rank, world, local = dist_env()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    pkg = graft.load_package()
    pkg._native.lib()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("LR_BENCH_FORCE_DIST") == "1"   # the latter: rehearse the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        tok = torch.zeros(1, device="cuda")

        def barrier():
            dist.all_reduce(tok)
            torch.cuda.synchronize()

        def all_max(v):
            t = torch.tensor([v], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
    else:
        def barrier():
            pass

        def all_max(v):
            return v

    ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
    N, moduli = params.DefaultParamsQi(args.logn)
    L = len(moduli)
    B = args.batch
    _, my_polys = shard_units(B * world, rank, world)   # weak scaling: B per GPU
    ctx = ring.NewContextWithParams(N, moduli, device=local)
    # synthetic operands: a few distinct polys tiled over the batch (generation cost), resident before timing
    base = sampling.uniform_poly(moduli, N, min(my_polys, 8), seed=0x4C415454 ^ rank)
    host = np.concatenate([base] * (-(-my_polys // base.shape[0])))[:my_polys]
    src, dst = ctx.NewPoly(my_polys).set(host), ctx.NewPoly(my_polys)
    del host

    def step():
        ctx.NTT(src, dst)

    # the device clock needs some tens of milliseconds of load to leave its idle state (the first launches after
    # start-up run ~12 % slower); bring it up during set-up so that short --steps/--warmup runs measure steady state
    t_up = time.perf_counter()
    while time.perf_counter() - t_up < 0.25:
        for _ in range(20):
            step()
        ctx.Sync()

    sync = torch.cuda.synchronize
    # the K timed launches are bracketed by HIP events on the launch stream as well (device-side duration)
    seconds, dev_ms = timed_region(step, args.steps, args.warmup, sync, barrier, all_max, ctx.TimerStart, ctx.TimerStop)
    kernel_ms = dev_ms / args.steps

    # parity spot-check inside the bench: first poly against the oracle (checker only)
    bit_exact = None
    if rank == 0:
        oracle = graft.load_oracle()
        oc = oracle.Context(N, moduli)
        got = np.empty((L, N), dtype=np.uint64)
        full = dst.get().reshape(my_polys, L, N)
        got[:] = full[0]
        bit_exact = bool(np.array_equal(got, oc.ntt(base[0])))
        del full

    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_final", "pmc_hbm.json")
    if os.path.exists(pmc_path) and args.logn == 15 and B == 256:
        try:
            traffic = json.load(open(pmc_path))["hbm_bytes_per_launch"]   # rocprofv3 --pmc, separate passes (see file)
        except Exception:
            traffic = None
    limb_ntts_total = B * world * L
    value = limb_ntts_total * args.steps / seconds
    achieved = ntt_bytes(N, L, my_polys) / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "NTT/s at N=2^%d, L=%d (batched forward negacyclic NTT, bit-exact vs reference arithmetic)" % (args.logn, L),
        "value": value,
        "unit": "limb-NTT/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": seconds / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "ring.DefaultParamsQi[%d]: Context.NTT on N=2^%d, %d x 60-bit limbs" % (args.logn, args.logn, L),
                   "polys_per_gpu": B, "limbs": L, "N": N, "sharding": "batch of independent polys, no collective"},
        "poly_ntt_per_s": value / L,
        "bit_exact": bit_exact,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": ("lr_ntt_fwd%d%s_m1" % (args.logn, "" if args.logn == 15 else "x" if args.logn < 16 else "s")) if 12 <= args.logn <= 16 and not os.environ.get("LR_NO_ASM") else "ntt_fwd_kernel<%d>" % args.logn, "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": ntt_bytes(N, L, my_polys),
                     # the north star says "HBM-read roofline"; `achieved` counts read + write, this is the read half
                     "achieved_read_only": achieved / 2},
    }

    if rank == 0 and not args.no_ckks:
        # second half of BASELINE.json's metric: CKKS MulRelin (ckks/evaluator.go:1016) at DefaultParams[PN15QP880],
        # device-resident batch of independent ciphertexts, synthetic operands and evaluation key
        cN, cQm, cPm = params.ckks_moduli("PN15QP880")
        cB = args.ckks_batch
        ccQ, ccP = ring.NewContextWithParams(cN, cQm, device=local), ring.NewContextWithParams(cN, cPm, device=local)
        plan = ring.CkksPlan(ccQ, ccP, cB)
        clevel = len(cQm) - 1
        cbeta = -(-len(cQm) // len(cPm))
        evk = plan.NewSwitchingKey().set(sampling.uniform_poly(cQm + cPm, cN, 2 * cbeta, seed=9))
        cbase = sampling.uniform_poly(cQm, cN, 2, seed=3)
        chost = np.concatenate([cbase] * (-(-cB // 2)))[:cB]
        mkc = lambda: ccQ.NewPoly(cB).set(chost)
        ct0, ct1, cto = (mkc(), mkc()), (mkc(), mkc()), (ccQ.NewPoly(cB), ccQ.NewPoly(cB))
        for _ in range(3):
            plan.MulRelin(clevel, ct0, ct1, evk, cto)
        ccQ.Sync()
        ccQ.TimerStart()
        for _ in range(10):
            plan.MulRelin(clevel, ct0, ct1, evk, cto)
        cms = ccQ.TimerStop() / 10
        out["ckks_mulrelin"] = {"value": cB / (cms * 1e-3), "unit": "MulRelin/s", "batch": cB, "ms_per_batch": cms,
                                "params": "PN15QP880 (N=2^15, 18 Q limbs + 3 P limbs, beta=6), level 17",
                                "algorithmic_GBs": cB * 8 * cN * 360 / (cms * 1e-3) / 1e9}
        del plan, ct0, ct1, cto, evk

    if rank == 0 and not args.no_ckks:
        # BASELINE.json config 4: BFV DefaultParams[PN14QP438] Evaluator.Mul (tensorAndRescale, bfv/evaluator.go:278),
        # degree-1 x degree-1 -> degree-2, coefficient-domain operands, device-resident batch
        bN, bQ, _, bQMul = params.bfv_moduli("PN14QP438")
        bB = args.ckks_batch
        bcQ, bcM = ring.NewContextWithParams(bN, bQ, device=local), ring.NewContextWithParams(bN, bQMul, device=local)
        bplan = ring.BfvPlan(bcQ, bcM, 65537, bB)
        bbase = sampling.uniform_poly(bQ, bN, 2, seed=5)
        bhost = np.concatenate([bbase] * (-(-bB // 2)))[:bB]
        mkb = lambda: bcQ.NewPoly(bB).set(bhost)
        b0, b1, bo = (mkb(), mkb()), (mkb(), mkb()), (bcQ.NewPoly(bB), bcQ.NewPoly(bB), bcQ.NewPoly(bB))
        for _ in range(3):
            bplan.Mul(b0, b1, bo)
        bcQ.Sync()
        bcQ.TimerStart()
        for _ in range(10):
            bplan.Mul(b0, b1, bo)
        bms = bcQ.TimerStop() / 10
        out["bfv_mul"] = {"value": bB / (bms * 1e-3), "unit": "Mul/s", "batch": bB, "ms_per_batch": bms,
                          "params": "PN14QP438 (N=2^14, 6 Q limbs, 6 QMul limbs, t=65537)"}
        del bplan, b0, b1, bo

    if not args.no_extras and rank == 0:
        # the other kernels BASELINE.json's north_star asks throughput for, same ring, same resident batch
        extras = {}
        reps = max(10, min(args.steps, 50))

        def timed(fn):
            for _ in range(3):
                fn()
            ctx.Sync()
            ctx.TimerStart()
            for _ in range(reps):
                fn()
            return ctx.TimerStop() / reps

        ms = timed(lambda: ctx.InvNTT(src, dst))
        extras["intt"] = {"limb_ntt_per_s": my_polys * L / (ms * 1e-3), "ms": ms,
                          "frac_hbm": ntt_bytes(N, L, my_polys) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        ms = timed(lambda: ctx.MulCoeffsMontgomery(src, dst, dst))
        extras["mulcoeffs_montgomery"] = {"poly_per_s": my_polys / (ms * 1e-3), "ms": ms,
                                          "frac_hbm": 24 * N * L * my_polys / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        _, pmod = params.DefaultParamsPi(args.logn)
        ctxP = ring.NewContextWithParams(N, pmod, device=local)
        bext = ring.NewFastBasisExtender(ctx, ctxP)
        if True:
            outP = ctxP.NewPoly(my_polys)
            ms = timed(lambda: bext.ModUpSplitQP(L - 1, src, outP))
            extras["modup_split_qp"] = {"poly_per_s": my_polys / (ms * 1e-3), "ms": ms,
                                        "frac_hbm": 8 * N * (L + len(pmod)) * my_polys / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "note": "%d -> %d limbs, %d modular multiplies per coefficient (compute-bound)" %
                                                (L, len(pmod), L + L * len(pmod))}
            del outP, bext
        if 12 <= args.logn <= 15:
            # the same transform on CKKS-size moduli (DefaultParams[PN15QP880]'s first limbs at N = 2^15: one of 50 bits, the rest
            # 40): limbs below 2^46 run on the FP64 body of the dual kernels, the others on the integer body beside it
            cq = list(params.ckks_moduli("PN15QP880")[1][:L]) if args.logn == 15 else params.GenerateNTTPrimes(40, args.logn, L)
            ctxC = ring.NewContextWithParams(N, cq, device=local)
            cb = sampling.uniform_poly(cq, N, min(my_polys, 2), seed=11)
            csrc = ctxC.NewPoly(my_polys).set(np.concatenate([cb] * (-(-my_polys // cb.shape[0])))[:my_polys])
            cdst = ctxC.NewPoly(my_polys)
            for name, fn in (("ntt", lambda: ctxC.NTT(csrc, cdst)), ("intt", lambda: ctxC.InvNTT(csrc, cdst))):
                ms = timed(fn)
                extras[name + "_ckks_moduli"] = {"limb_ntt_per_s": my_polys * L / (ms * 1e-3), "ms": ms,
                                                 "frac_hbm": ntt_bytes(N, L, my_polys) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "moduli_bits": [int(q).bit_length() for q in cq],
                                                 "asm_variants": list(ctxC.ntt_variants())}
            del csrc, cdst, ctxC
        out["extras"] = extras

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_ntt(N, moduli)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
