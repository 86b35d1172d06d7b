#!/usr/bin/env python3
"""Headline benchmark: batched forward NTT on the reference's benchmark ring R15, plus the CKKS / BFV caller sequences.

    python bench.py --gpus N --steps K --warmup W [--workload ntt|ckks16]

Workload "ntt" (default; BASELINE.json metric "NTT/s ... at N=2^15, L=16"): ring.DefaultParamsQi[15]
(ring/params.go:14: N = 2^15, 16 x 60-bit limbs), a batch of B uniform polynomials resident in HBM
(synthetic, splitmix64 -- the reference's BenchmarkRing uses NewUniformPoly, ring_benchmark_test.go:160).
One step = Context.NTT (ring/ntt.go:4) over the whole batch = ONE kernel launch of B*16 limb-NTTs.
Units are independent polynomials, so N GPUs shard the batch with no data-path collective
(weak scaling: B polys per GPU); the collectives of that leg are the timing barrier and a max over ranks.

Workload "ckks16" (BASELINE.json config 5): independent CKKS MulRelin at DefaultParams[PN16QP1761] (N = 2^16, 34 Q + 4 P
limbs, ckks/params.go:78-86), --config5-units ciphertext products per GPU (128 = config 5's share of one GPU of eight), each rank
on its contiguous block, then ONE gather of the results to rank 0 (torch.distributed.gather on the nccl backend = RCCL
send/recv over xGMI; lattigo-fhe-by-go_amd/sharding.py).  One step = the whole block + the gather; compute-only and
compute+gather times are reported separately.  The default run carries this leg as the `config5` object.

Prints one JSON line (rank 0).  `value` = units per second over all GPUs; `roofline.achieved` = algorithmic bytes
 / device time measured with HIP events on the launch stream inside this run; `roofline.kernel` is the
kernel the library reports it dispatched.  `roofline.traffic` (HBM bytes per launch, and per product on the MulRelin leg) is counted in
the run at N = 1: PMC counters need their own rocprofv3 passes, so four short child runs of `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE, one
counter each) execute the same kernel / pipeline on the same shape at the end; null with the reason in `traffic_source` if the profiler
cannot run (--no-traffic skips it; tools/collect_profiles.sh writes the committed figures named in `traffic_profile`).  `cpu_baseline` objects time the CPU oracle
(C restatement of the Go algorithm, rebuilt -O2 -march=native on this machine; Go itself is not installable here) on the
host cores, rank 0 at N=1 only, on bounded samples.

Beside the headline the line carries (N = 1): `extras` / `rings` / `mulrelin_sets` (the other ring operations and sizes), `ckks_mulrelin`, `bfv_mul`
(BASELINE config 4) and `pipelines` -- one timed call per entry point of the reference's ring, ckks-evaluator and bfv-evaluator benchmark lists
(tools/bench_legs.py), each with roofline, in-run traffic, vector issue, kernel split, bit_exact and cpu_baseline --, `evaluator_threads`,
`evaluator_threads_bfv` (one evaluator per host thread, direct and through the batchers, Python threads), `evaluator_threads_native` (the same
from a C++ host, tools/batcher_bench.cpp), `config5` / `config5_single_process`, and last `summary`: the short form a reader of the line's tail needs.
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
    """Contiguous block partition of `total` independent units: (start, count) for `rank` (package: sharding.shard_units)."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def ntt_bytes(N, limbs, polys=1):
    """Algorithmic HBM bytes of `polys` poly-NTTs: read 8N + write 8N per limb."""
    return 16 * N * limbs * polys


def mulrelin_bytes(N, nq, np_, products=1):
    """Algorithmic HBM bytes of CKKS MulRelin at full level: two input ciphertexts 2*2*nq limbs, the
    evaluation key beta*2*(nq+np) limbs, the output ciphertext 2*nq limbs; PN15QP880: 8N*360 per product."""
    beta = -(-nq // np_)
    return 8 * N * (4 * nq + beta * 2 * (nq + np_) + 2 * nq) * products


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


# ------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (kind "port") on the host cores of this machine
# ------------------------------------------------------------------------------------------------------------------------
def host_cores():
    """Threads worth starting on this machine: the CPUs this process may run on, capped by the cgroup CPU quota (a GPU box
    hands one GPU's share of the host, e.g. 16 of 256 hardware threads, to the job), overridable with LR_BENCH_CPU_THREADS."""
    if os.environ.get("LR_BENCH_CPU_THREADS"):
        return max(1, int(os.environ["LR_BENCH_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(-(-int(txt[0]) // int(txt[1])))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, -(-quota // period)))
            break
        except Exception:
            continue
    return max(1, n)


_EFFECTIVE = [None]


def effective_cores():
    """host_cores(), checked against what the machine really grants: when more than 32 CPUs are visible and no cgroup quota is
    published, a short calibration (the oracle's NTT on a small ring, T threads against one) measures the parallel speed-up
    and the thread count is cut to it -- starting 256 threads on a 16-core share only lengthens the sample."""
    if _EFFECTIVE[0] is not None:
        return _EFFECTIVE[0]
    n = host_cores()
    if n > 32 and not os.environ.get("LR_BENCH_CPU_THREADS"):
        import concurrent.futures as cf

        import numpy as np

        import __graft_entry__ as graft
        oracle = graft.load_oracle()
        oc = oracle.Context(1 << 12, [1152921504606584833])
        lib = oracle.lib()
        bufs = [(np.ones((1, 1 << 12), dtype=np.uint64), np.empty((1, 1 << 12), dtype=np.uint64)) for _ in range(n)]

        def spin(i, calls):
            a, b = bufs[i]
            for _ in range(calls):
                lib.oc_ntt_lvl(oc.h, 0, a.ctypes.data, b.ctypes.data)

        t0 = time.perf_counter()
        spin(0, 200)
        t1 = (time.perf_counter() - t0) / 200
        calls = max(50, int(0.4 / t1))
        with cf.ThreadPoolExecutor(max_workers=n) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda i: spin(i, calls), range(n)))
            tn = (time.perf_counter() - t0) / calls
        speedup = n * t1 / tn
        n = max(1, min(n, int(round(speedup))))
    _EFFECTIVE[0] = n
    return n


NOMINAL_SCLK_HZ = 2.4e9     # MI355X peak engine clock; what the chip sustains under these kernels is reported beside it
SIMDS = 256 * 4


def _under_profiler():
    return any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def _pmc_pass(counters, script_args, timeout=180):
    """One child run of `rocprofv3 --pmc <counters>` (no trace domains) over a script under tools/dbg: the rows of its counter CSV
    and its stdout.  Raises RuntimeError with the reason when the profiler cannot run."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    if _under_profiler():
        raise RuntimeError("this run is itself under a profiler (nested rocprofv3 passes are not attempted)")
    out_dir = tempfile.mkdtemp(prefix="lr_pmc_", dir="/tmp")
    try:
        # a clean environment for the child: no preload / tool variables of an outer profiler
        env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "ROCPROF")) and k != "LD_PRELOAD"}
        env["TMPDIR"] = "/tmp"
        # the profiled program goes directly after `--`: the interpreter binary itself, no wrapper
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out_dir, "-o", "p", "--", os.path.realpath(sys.executable)] + script_args
        res = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        rows = []
        for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
            rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] in counters]
        if res.returncode != 0 or not rows:
            raise RuntimeError("%s pass: rc %d, %d counter rows" % ("+".join(counters), res.returncode, len(rows)))
        return rows, res.stdout.decode(errors="replace")
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"))


def _valu_object(insts, waves, shader_cycles, profiled_seconds, run_seconds):
    """The other roof (VALU utilisation beside GB/s).  One vector instruction occupies its SIMD for 4 clocks (the cost the
    SQ's own VALU-busy counter charges; plain 32-bit instructions can issue in 2, so a kernel dense in those reads above its true share):
    issue_frac = instructions x 4 / (1024 SIMDs x shader clocks of the launch).  `nominal` prices the launch at the 2.4 GHz peak clock and
    the un-profiled duration of this run, `sustained` at the shader clocks the profiled pass counted (GRBM_GUI_ACTIVE / 8 XCDs): the chip
    runs these kernels at its 1.4 kW package power limit and lowers its clock to stay there."""
    return {"instr_per_wave": insts / waves if waves else None, "waves": waves, "instructions": insts, "clocks_per_instruction_assumed": 4,
            "issue_frac_nominal": insts * 4 / (SIMDS * NOMINAL_SCLK_HZ * run_seconds),
            "issue_frac_sustained": insts * 4 / (SIMDS * shader_cycles) if shader_cycles else None,
            "sclk_MHz": shader_cycles / profiled_seconds / 1e6 if profiled_seconds else None}


def measure_kernel(script_args, kernel_name, run_ms):
    """HBM bytes and vector-instruction issue of ONE kernel, counted now: three child passes (PMC_PASSES) over a script that launches it on
    the bench shape; per-launch medians over the dispatches whose name contains `kernel_name`; FETCH_SIZE doubled per the gfx950
    correction of the microarch guide.  Returns (traffic bytes, traffic detail, valu object) or (None, reason, None)."""
    import statistics
    med = {}
    dur = None
    try:
        for counters in PMC_PASSES:
            rows, _ = _pmc_pass(counters, script_args)
            mine = [r for r in rows if kernel_name in r["Kernel_Name"]]
            for c in counters:
                vals = [float(r["Counter_Value"]) for r in mine if r["Counter_Name"] == c]
                if not vals:
                    raise RuntimeError("no %s rows for %s" % (c, kernel_name))
                med[c] = statistics.median(vals)
            if "GRBM_GUI_ACTIVE" in counters:
                dur = statistics.median([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in mine if r["Counter_Name"] == "GRBM_GUI_ACTIVE"])
    except Exception as ex:     # noqa: BLE001 -- the measurement is optional, the reason is reported
        return None, str(ex), None
    traffic = 2 * med["FETCH_SIZE"] * 1024 + med["WRITE_SIZE"] * 1024
    valu = _valu_object(med["SQ_INSTS_VALU"], med["SQ_WAVES"], med["GRBM_GUI_ACTIVE"] / 8, dur, run_ms * 1e-3)
    return traffic, {"FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"]}, valu


def measure_pipeline(script_args, run_ms_per_product):
    """The same for a pipeline of launches (tools/dbg/mulrelin_pmc.py): every kernel summed (the runtime's own fill / copy kernels of
    the set-up excluded) and divided by the products the script reports."""
    tot = {}
    dur = 0.0
    products = None
    try:
        for counters in PMC_PASSES:
            rows, stdout = _pmc_pass(counters, script_args)
            marks = [ln for ln in stdout.splitlines() if ln.startswith("PRODUCTS")]
            if not marks:
                raise RuntimeError("the script did not report its product count")
            products = int(marks[-1].split()[1])
            mine = [r for r in rows if "rocclr" not in r["Kernel_Name"]]
            for c in counters:
                tot[c] = sum(float(r["Counter_Value"]) for r in mine if r["Counter_Name"] == c) / products
            if "GRBM_GUI_ACTIVE" in counters:
                dur = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in mine if r["Counter_Name"] == "GRBM_GUI_ACTIVE") / products
    except Exception as ex:     # noqa: BLE001
        return None, str(ex), None
    traffic = 2 * tot["FETCH_SIZE"] * 1024 + tot["WRITE_SIZE"] * 1024
    valu = _valu_object(tot["SQ_INSTS_VALU"], tot["SQ_WAVES"], tot["GRBM_GUI_ACTIVE"] / 8, dur, run_ms_per_product * 1e-3)
    valu["per"] = "product"
    return traffic, {"FETCH_SIZE_KB_per_product": tot["FETCH_SIZE"], "WRITE_SIZE_KB_per_product": tot["WRITE_SIZE"]}, valu


def sample_power(step, sync, seconds=1.6):
    """package power and shader clock while `step` loops for `seconds`: rocm-smi polled from a thread (each poll is a process start, a few
    per second); -> {"package_W": max seen, "sclk_MHz": median seen, "samples": n} or {"error": ...}.  The kernels of this path run the
    package into its 1.4 kW limit and the clock comes down to stay there: these two numbers are that statement, per run."""
    import re
    import shutil
    import statistics
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"error": "rocm-smi not found"}
    seen, stop = [], [False]

    def poll():
        while not stop[0]:
            try:
                txt = subprocess.run([exe, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
                card = next(iter(json.loads(txt).values()))
                w = [float(v) for k, v in card.items() if "ower" in k and "(W)" in k and re.match(r"^[0-9.]+$", str(v))]
                mhz = [int(m.group(1)) for k, v in card.items() if k.startswith("sclk") for m in [re.search(r"\((\d+)Mhz\)", str(v))] if m]
                if w and mhz:
                    seen.append((max(w), mhz[0]))
            except Exception:     # noqa: BLE001 -- a failed poll is a missing sample
                pass
            time.sleep(0.05)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            step()
        sync()
    stop[0] = True
    th.join()
    seen = seen[1:] if len(seen) > 2 else seen          # the first poll starts before the load has built up
    if not seen:
        return {"error": "rocm-smi gave no power / clock reading"}
    return {"package_W": max(w for w, _ in seen), "sclk_MHz_smi": statistics.median(m for _, m in seen), "samples": len(seen),
            "how": "rocm-smi --showpower --showclocks polled while the headline launch loops for %.1f s" % seconds}


def load_bench_legs():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_legs", os.path.join(ROOT, "tools", "bench_legs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def measure_legs(names, run_ms):
    """Counters of the pipeline legs, all in three child passes: tools/dbg/legs_pmc.py runs every leg between marker kernels
    (div_selftest_kernel: marker, call, marker, call, marker per leg), the dispatch list is cut at the markers and leg i's measured call is
    segment 3i + 2.  -> {leg: (traffic bytes per call, detail, valu object, [(kernel, launches, microseconds)])} or (None, reason)."""
    per = {}
    order = None
    try:
        for counters in PMC_PASSES:
            rows, stdout = _pmc_pass(counters, [_tool("legs_pmc.py")] + list(names), timeout=600)
            legs = [ln.split()[1] for ln in stdout.splitlines() if ln.startswith("LEG ")]
            if not legs or "END" not in stdout:
                raise RuntimeError("the leg script did not finish: %s" % stdout[-300:])
            order = legs
            disp = {}
            for r in rows:
                disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
            segs, cur = [], []
            for did in sorted(disp):
                if "div_selftest" in disp[did][0]["Kernel_Name"]:
                    segs.append(cur)
                    cur = []
                else:
                    cur.append(disp[did])
            segs.append(cur)
            if len(segs) < 3 * len(legs) + 1:
                raise RuntimeError("%d marker segments for %d legs" % (len(segs), len(legs)))
            for i, name in enumerate(legs):
                mine = [d for d in segs[3 * i + 2] if "rocclr" not in d[0]["Kernel_Name"]]
                acc = per.setdefault(name, {})
                for c in counters:
                    acc[c] = sum(float(r["Counter_Value"]) for d in mine for r in d if r["Counter_Name"] == c)
                if "GRBM_GUI_ACTIVE" in counters:
                    ks = {}
                    for d in mine:
                        r = d[0]
                        k = ks.setdefault(r["Kernel_Name"].split("(")[0][-72:], [0, 0.0])
                        k[0] += 1
                        k[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                    acc["dur"] = sum(v[1] for v in ks.values()) * 1e-6
                    acc["kernels"] = sorted(([k, v[0], round(v[1], 1)] for k, v in ks.items()), key=lambda t: -t[2])[:6]
    except Exception as ex:     # noqa: BLE001 -- the measurement is optional, the reason is reported
        return None, str(ex)
    out = {}
    for name in order:
        a = per[name]
        traffic = 2 * a["FETCH_SIZE"] * 1024 + a["WRITE_SIZE"] * 1024
        valu = _valu_object(a["SQ_INSTS_VALU"], a["SQ_WAVES"], a["GRBM_GUI_ACTIVE"] / 8, a["dur"], run_ms.get(name, 0.0) * 1e-3 or a["dur"])
        valu["per"] = "call"
        out[name] = (traffic, {"FETCH_SIZE_KB_per_call": a["FETCH_SIZE"], "WRITE_SIZE_KB_per_call": a["WRITE_SIZE"]}, valu, a["kernels"])
    return out, None


def finish_roofline(r, traffic, detail, valu, how, algorithmic):
    """fills traffic / traffic_source / valu of a roofline object and names the binding roof: whichever of the HBM fraction and the
    vector-issue fraction (at the clock the chip sustained) is nearer 1"""
    r["traffic"] = traffic
    r["traffic_source"] = ({"how": how, **detail, "ratio_to_algorithmic": traffic / algorithmic} if traffic is not None
                           else {"how": "not measured in this run: %s" % detail})
    r["valu"] = valu
    r["frac_is"] = "hbm: algorithmic bytes / duration / 8 TB/s"
    if valu and valu.get("issue_frac_sustained") is not None:
        r["bound"] = "valu" if valu["issue_frac_sustained"] > r["frac"] else "hbm"
        r["bound_note"] = ("vector-instruction issue at the sustained clock is %.2f of its ceiling, HBM %.2f of 8 TB/s; every kernel of this path runs at the "
                           "1.4 kW package power limit (clock %.0f MHz of 2400), see DESIGN.md 3.1" % (valu["issue_frac_sustained"], r["frac"], valu["sclk_MHz"] or 0))


def _tool(name):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "dbg", name)


def progress(msg):
    print("[bench %6.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline(make_worker, units_per_call, unit, what, target_seconds):
    """make_worker(i) -> a no-argument callable doing ONE call of the reference's method on thread i's own operands
    (FastBasisExtender and the evaluators own scratch: one object per thread, like one per goroutine,
    examples/dbfv/psi/psi.go:219-233).  Times one call on one thread, then one call on each of T threads at once (this
    calibrates the sample: the machine may grant fewer cores than it shows), then T threads x k calls for about
    `target_seconds`."""
    import concurrent.futures as cf

    import __graft_entry__ as graft
    oracle = graft.load_oracle()
    cores = effective_cores()
    workers = [make_worker(i) for i in range(cores)]
    t0 = time.perf_counter()
    workers[0]()
    t_one = time.perf_counter() - t0

    def run(per_thread):
        def work(i):
            for _ in range(per_thread):
                workers[i]()
        with cf.ThreadPoolExecutor(max_workers=cores) as ex:
            t0 = time.perf_counter()
            list(ex.map(work, range(cores)))
            return time.perf_counter() - t0

    t_round = run(1)                                    # one call on every thread, concurrently
    per_thread = max(1, min(8192, int(target_seconds / max(t_round, 1e-6))))
    dt = run(per_thread) if per_thread > 1 else t_round  # (a call that already fills the sample is the sample)
    calls = cores * per_thread
    progress("cpu baseline: %s done" % what)
    return {
        "value": calls * units_per_call / dt,
        "unit": unit,
        "cores": cores,
        "kind": "port",
        "one_thread_value": units_per_call / t_one,
        "march_native": bool(oracle.native_loaded),
        "sample": "%s: %d threads x %d calls, %.1f s; one call on one thread %.2f ms" % (what, cores, per_thread, dt, t_one * 1e3),
    }


def agreed_leg(name, prepare, run, all_max, world, rank, log=lambda m: None):
    """A leg after the headline, in two phases.  prepare(): set-up WITHOUT collectives (contexts, allocations, uploads) -> state;
    run(state): the timing, whose barrier / max-over-ranks are collectives.  All ranks agree on the outcome of prepare (max of an
    error flag: one all-reduce) before any of them enters run, so a rank that failed in set-up never leaves the others waiting in
    a collective it will not join; the leg is then reported as failed on every rank.  An exception inside run on a multi-rank job
    ends this rank with a non-zero code (the launcher stops the others) instead of a 300 s collective timeout."""
    err, state = None, None
    try:
        state = prepare()
    except Exception as ex:     # noqa: BLE001 -- reported, not swallowed
        err = "%s: %s" % (type(ex).__name__, ex)
    if all_max(1.0 if err else 0.0) > 0:
        res = {"error": err or "set-up failed on another rank"}
        log("%s leg skipped on every rank: %s" % (name, res["error"]))
        return res
    try:
        return run(state)
    except Exception as ex:     # noqa: BLE001
        if world > 1:
            log("%s leg failed inside its timed part on rank %d: %s: %s -- ending the job" % (name, rank, type(ex).__name__, ex))
            sys.stderr.flush()
            os._exit(3)
        log("%s leg failed: %s: %s" % (name, type(ex).__name__, ex))
        return {"error": "%s: %s" % (type(ex).__name__, ex)}


def launcher_argv(gpus, port, passthrough):
    """The command the parent of a multi-GPU run starts as a child process: one rank per GPU of this node under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [os.path.realpath(sys.executable), "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(passthrough)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process -- which has not imported torch nor touched a
    GPU -- starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line (the
    child's stdout) and exits with the child's code.  Under a launcher (WORLD_SIZE set) bench.py runs as a rank instead."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT") or free_port())
    cmd = launcher_argv(args.gpus, port, argv)
    if args.dry_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    lines = 0
    for raw in child.stdout:
        line = raw.decode(errors="replace").rstrip("\n")
        try:
            json.loads(line)
        except ValueError:
            print(line, file=sys.stderr, flush=True)      # anything else a rank printed on stdout
            continue
        print(line, flush=True)
        lines += 1
    rc = child.wait()
    if rc == 0 and lines != 1:
        print("[bench] the ranks exited cleanly but printed %d JSON lines" % lines, file=sys.stderr)
        return 4
    return rc


def launch_check():
    """--launch-check: what a rank does in the self-launch rehearsal (tests/test_bench_launch.py, no GPU): rendezvous over gloo with the
    environment the launcher gave it, the barrier + max-over-ranks of the timing harness, the error-flag agreement of the legs, one JSON
    line from rank 0."""
    import datetime

    import torch
    import torch.distributed as dist
    rank, world, _ = dist_env()
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    def all_max(v):
        x = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        return float(x.item())

    def prepare_maybe_failing():
        if os.environ.get("LR_BENCH_CHECK_FAIL_RANK") == str(rank):
            raise RuntimeError("set-up failure injected on rank %d" % rank)
        return rank

    # the two-phase legs of the real run, with gloo collectives: one leg whose set-up fails on one rank, one that runs
    leg_a = agreed_leg("check-a", prepare_maybe_failing, lambda st: {"ran_on": st, "max": all_max(float(st))}, all_max, world, rank)
    leg_b = agreed_leg("check-b", lambda: 7, lambda st: {"max": all_max(float(st + rank))}, all_max, world, rank)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "max_rank": int(t.item()), "a_rank_failed": "error" in leg_a,
                          "leg_a": leg_a, "leg_b": leg_b,
                          "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["ntt", "ckks16"], default="ntt")
    ap.add_argument("--batch", type=int, default=256, help="polynomials per GPU (workload ntt)")
    ap.add_argument("--logn", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child runs that fill roofline.traffic (N = 1, default shape)")
    ap.add_argument("--no-rings", action="store_true", help="skip the R13..R16 table (NTT / InvNTT / MulCoeffsMontgomery / ModUpSplitQP on every rank)")
    ap.add_argument("--rings-bytes", type=int, default=1 << 30, help="bytes per operand of the R13..R16 table")
    ap.add_argument("--no-extras", action="store_true", help="skip the InvNTT / MulCoeffsMontgomery / ModUp timings")
    ap.add_argument("--no-ckks", action="store_true", help="skip the CKKS MulRelin / BFV Mul legs")
    ap.add_argument("--no-threads", action="store_true", help="skip the evaluator-per-host-thread MulRelin leg (T = 1, 4, 16 threads, batch 1 each)")
    ap.add_argument("--no-config5", action="store_true", help="skip the PN16QP1761 sharded MulRelin + gather leg")
    ap.add_argument("--no-pipelines", action="store_true", help="skip the legs of tools/bench_legs.py other than bfv_mul (ModDown, DivFloor / DivRound, Rescale, Relin, "
                    "rotations, hoisted rotations, pk-encrypt, decrypt, BFV Relin / rotations, SimpleScaler, marshal ingest)")
    ap.add_argument("--pipelines", default="", help="comma-separated subset of the legs to run")
    ap.add_argument("--hw-queues", type=int, default=16, help="GPU_MAX_HW_QUEUES for this process (0: leave the runtime's default of 4 per priority class); "
                    "only the evaluator_threads rows have more than one busy stream")
    ap.add_argument("--ckks-batch", type=int, default=128)
    ap.add_argument("--config5-units", type=int, default=128, help="PN16QP1761 ciphertext products per GPU")
    ap.add_argument("--config5-chunk", type=int, default=32, help="products per lr_ckks_mulrelin call")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no launcher: print the child command as JSON and exit")
    ap.add_argument("--launch-check", action="store_true", help="ranks only rendezvous over gloo and print one line (CPU rehearsal of the self-launch path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: become the launcher (before torch is imported or a GPU is touched)
        raise SystemExit(self_launch(args, [a for a in sys.argv[1:] if a != "--dry-launch"]))
    if args.launch_check:
        launch_check()
        return

    # the contract is ONE JSON line on stdout: libraries that print there (RCCL's version banner at communicator creation)
    # are sent to stderr for the whole run, the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    # Streams are bound to hardware queues when they are created, four per priority class by default; two streams on one queue run one
    # after the other, and which ones collide depends on every stream the process has created (profiles/r03/hw_queues.txt).
    # Only the evaluator-per-thread rows have more than one stream with work; the setting is reported there.  The runtime reads it once,
    # at its initialisation, hence here.
    if args.hw_queues > 0:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(args.hw_queues))
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
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        import datetime
        # a collective that a peer never joins ends the run after five minutes instead of holding the node
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
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

    ring, params, sampling, sharding, nat = pkg.ring, pkg.params, pkg.sampling, pkg.sharding, pkg._native
    sync = torch.cuda.synchronize
    oracle = graft.load_oracle() if rank == 0 else None
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if want_cpu:
        oracle.use_native()

    def warm_clock(step, ctx):
        # the device clock needs some tens of milliseconds of load to leave its idle state (the first launches after
        # start-up run ~12 % slower); bring it up during set-up so that short --steps/--warmup runs measure steady state
        t_up = time.perf_counter()
        while time.perf_counter() - t_up < 0.25:
            for _ in range(20):
                step()
            ctx.Sync()

    # --------------------------------------------------------------------------------------------------------------------
    # config 5: PN16QP1761 MulRelin on this rank's block of independent ciphertext products, then the gather to rank 0
    # --------------------------------------------------------------------------------------------------------------------
    def checked(fn, default):
        """rank-0-only oracle comparisons: an exception there must not separate rank 0 from the ranks waiting at the next agreement"""
        try:
            return fn(), None
        except Exception as ex:     # noqa: BLE001 -- reported in the leg's object
            return default, "%s: %s" % (type(ex).__name__, ex)

    def secondary(name, prepare, run, store=None):
        res = agreed_leg(name, prepare, run, all_max, world, rank, progress)
        if rank == 0:
            if store is not None:
                store.append(res)
            else:
                out[name] = res

    def config5_prepare():
        cN, cQm, cPm = params.ckks_moduli("PN16QP1761")
        nq, np_ = len(cQm), len(cPm)
        level, beta = nq - 1, -(-nq // np_)
        per_gpu = args.config5_units
        total = per_gpu * world
        start, count = sharding.shard_units(total, rank, world)
        chunk = min(args.config5_chunk, count)
        cQ, cP = ring.NewContextWithParams(cN, cQm, device=local), ring.NewContextWithParams(cN, cPm, device=local)
        # Every launch of this leg goes on ONE explicit torch stream: the products (both contexts moved to it) and, under
        # torch.cuda.stream(s), the point ProcessGroupNCCL orders its collectives behind -- so a gather of a chunk starts after the
        # kernels that wrote it, without a host synchronisation.  torch's default stream has handle 0, which lr_context_set_stream
        # reads as "the library's own stream": that stream does not synchronise with the legacy default stream, so the default
        # stream must not be used here (tests/test_gpu_bench_contract.py::test_config5_leg_under_rccl_is_ordered poisons the
        # outputs to catch exactly that).
        st = torch.cuda.Stream()
        cQ.SetStream(st.cuda_stream)
        cP.SetStream(st.cuda_stream)
        plan = ring.CkksPlan(cQ, cP, chunk)
        evk_h = sampling.uniform_poly(cQm + cPm, cN, 2 * beta, seed=9)       # replicated on every rank
        evk = plan.NewSwitchingKey().set(evk_h)
        # unit g of the global batch has its own seeded operands; a few distinct ones are uploaded and tiled over the block ON THE
        # DEVICE (the host never holds more than the distinct units: 4 x 4 x 34 x 65536 words = 285 MB, not 9 GB per rank)
        distinct = min(count, 4)
        base = np.stack([sampling.uniform_poly(cQm, cN, 4, seed=0xC5 * 1000 + ((start + j) % 64)).reshape(4, nq, cN) for j in range(distinct)])
        with torch.cuda.stream(st):
            dev_base = torch.from_numpy(base.view(np.int64)).to("cuda")                                     # [distinct, 4, nq, N]
            dev_in = dev_base[torch.arange(count, device="cuda") % distinct].contiguous()                   # [count, 4 (a0 a1 b0 b1), nq, N]
            dev_out = torch.empty((count, 2, nq, cN), dtype=torch.int64, device="cuda")
        st.synchronize()
        del dev_base, base
        esz = 8

        def comp(t, u0, nb, k):
            # component k of units u0..u0+nb as an lr_poly over the tensor's memory: poly stride = t.shape[1] * nq limbs
            return ring.Poly.wrap_strided(cQ, t.data_ptr() + ((u0 * t.shape[1] + k) * nq * cN) * esz, nq, nb, t.shape[1] * nq)

        calls, spans = [], []
        for u0 in range(0, count, chunk):
            nb = min(chunk, count - u0)
            spans.append((u0, nb))
            calls.append(((comp(dev_in, u0, nb, 0), comp(dev_in, u0, nb, 1)), (comp(dev_in, u0, nb, 2), comp(dev_in, u0, nb, 3)),
                          (comp(dev_out, u0, nb, 0), comp(dev_out, u0, nb, 1))))
        overlapped = use_dist and total % world == 0
        with torch.cuda.stream(st):
            root_out = torch.empty((total, 2, nq, cN), dtype=torch.int64, device="cuda") if overlapped and rank == 0 else None
        return dict(cN=cN, cQm=cQm, cPm=cPm, nq=nq, np_=np_, level=level, beta=beta, per_gpu=per_gpu, total=total, start=start, count=count,
                    chunk=chunk, cQ=cQ, cP=cP, st=st, plan=plan, evk_h=evk_h, evk=evk, dev_in=dev_in, dev_out=dev_out, calls=calls,
                    spans=spans, overlapped=overlapped, root_out=root_out)

    def config5_run(c, steps, warmup):
        cN, cQm, cPm, nq, np_, level, beta = c["cN"], c["cQm"], c["cPm"], c["nq"], c["np_"], c["level"], c["beta"]
        total, count, chunk, cQ, st, plan, evk = c["total"], c["count"], c["chunk"], c["cQ"], c["st"], c["plan"], c["evk"]
        dev_out, calls, spans, overlapped, root_out = c["dev_out"], c["calls"], c["spans"], c["overlapped"], c["root_out"]

        def compute():
            for ct0, ct1, out in calls:
                plan.MulRelin(level, ct0, ct1, evk, out)

        gathered = [None]

        # the gather runs chunk by chunk behind the products: chunk k moves over xGMI (RCCL, its own stream, behind an event of `st`)
        # while chunk k + 1 is computed (sharding.ChunkedGather); rank 0's result buffer is allocated once
        def step():
            with torch.cuda.stream(st):
                if overlapped:
                    cg = sharding.ChunkedGather(dev_out, total, rank, world, dst=0, out=root_out)
                    for (ct0, ct1, out), (u0, nb) in zip(calls, spans):
                        plan.MulRelin(level, ct0, ct1, evk, out)
                        cg.submit(u0, nb)
                    gathered[0] = cg.wait()
                    return
                compute()
                gathered[0] = sharding.gather_blocks(dev_out, total, rank, world, dst=0) if use_dist else dev_out

        warm_clock(compute, cQ)
        seconds, dev_ms = timed_region(step, steps, warmup, sync, barrier, all_max, cQ.TimerStart, cQ.TimerStop)
        # compute only, same block, no collective
        comp_seconds, comp_ms = timed_region(compute, steps, 1, sync, barrier, all_max, cQ.TimerStart, cQ.TimerStop)
        # the checked result comes from one more step on POISONED buffers: every warm-up and timed step wrote the same values, so a
        # gather that ran ahead of the kernels would otherwise find yesterday's (identical) data and pass
        with torch.cuda.stream(st):
            dev_out.fill_(-1)
            if root_out is not None:
                root_out.fill_(-1)
        step()
        sync()
        barrier()
        res = {
            "value": total * steps / seconds, "unit": "MulRelin/s", "params": "PN16QP1761 (N=2^16, %d Q + %d P limbs, beta=%d), level %d" % (nq, np_, beta, level),
            "units_total": total, "units_per_gpu": c["per_gpu"], "chunk": chunk, "n_gpus": world,
            "ms_per_step_compute_and_gather": seconds / steps * 1e3, "ms_per_step_compute_only": comp_seconds / steps * 1e3,
            "compute_only_value": total * steps / comp_seconds,
            "stream": "explicit torch stream shared by the products and the collective's ordering point",
            "gather": ("torch.distributed.gather(nccl=RCCL) of %d x %.1f MiB to rank 0%s" % (total, 2 * nq * cN * 8 / 2**20,
                       ", in chunks of %d products per rank overlapped with the next chunk's kernels" % chunk if overlapped else "")) if use_dist else "none (single process)",
            "gather_bytes_to_root": (total - count) * 2 * nq * cN * 8 if use_dist else 0,
            "roofline": {"bound": "hbm", "achieved": mulrelin_bytes(cN, nq, np_, count) / (comp_ms / steps * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "traffic": None, "pipeline_ms": comp_ms / steps,
                         "algorithmic_bytes_per_product": mulrelin_bytes(cN, nq, np_)},
        }
        res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
        if rank == 0:
            # placement + parity on the poisoned-then-recomputed buffers: global unit 0 (this rank's), the last unit of this rank's
            # block and, when gathered, the first and last unit of the last rank's block
            def check():
                oplan = oracle.CkksPlan(oracle.Context(cN, cQm), oracle.Context(cN, cPm))
                full = gathered[0] if gathered[0] is not None else dev_out
                checks = {0, count - 1}
                if use_dist and world > 1:
                    s_last, c_last = sharding.shard_units(total, world - 1, world)
                    checks |= {s_last, s_last + c_last - 1}
                ok = True
                for g in sorted(checks):
                    owner = sharding.unit_owner(g, total, world)
                    s_owner, c_owner = sharding.shard_units(total, owner, world)
                    d_owner = min(c_owner, 4)
                    opnd = sampling.uniform_poly(cQm, cN, 4, seed=0xC5 * 1000 + ((s_owner + (g - s_owner) % d_owner) % 64)).reshape(4, nq, cN)
                    want = oplan.mulrelin(level, opnd[0:2], opnd[2:4], c["evk_h"].reshape(beta, 2, nq + np_, cN))
                    got = full[g].cpu().numpy().view(np.uint64)
                    ok = ok and bool(np.array_equal(got, want))
                return ok, sorted(checks)
            (res["bit_exact"], res["checked_units"]), err = checked(check, (None, []))
            res["checked_after"] = "outputs poisoned (-1) on every rank and on the root, then one more step"
            if err:
                res["check_error"] = err
        return res

    if args.workload == "ckks16":
        c5 = config5_run(config5_prepare(), max(1, min(args.steps, 10)), max(1, min(args.warmup, 2)))
        if rank == 0:
            steps = max(1, min(args.steps, 10))
            out = {"metric": "CKKS homomorphic-mul/s at N=2^16 (MulRelin, DefaultParams[PN16QP1761]), sharded batch + gather to rank 0",
                   "value": c5["value"], "unit": "MulRelin/s", "n_gpus": world, "steps": steps, "warmup": max(1, min(args.warmup, 2)),
                   "ms_per_step": c5["ms_per_step_compute_and_gather"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": "u64", "data": "synthetic",
                   "config": {"workload": "BASELINE config 5: %d independent CKKS MulRelin per GPU at PN16QP1761, contiguous blocks, RCCL gather" % args.config5_units,
                              "units_per_gpu": args.config5_units},
                   "roofline": c5.pop("roofline"), "config5": c5}
            emit(out)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # --------------------------------------------------------------------------------------------------------------------
    # the reference's benchmark rings R13..R16 (ring/params.go:10-25, ring/ring_test.go:30-36): NTT, InvNTT, MulCoeffsMontgomery and
    # ModUpSplitQP, 1 GiB per operand on every rank (weak scaling), whole-job throughput = units of all ranks / slowest rank's time
    # --------------------------------------------------------------------------------------------------------------------
    def ring_prepare(logn):
        Nr, Qr = params.DefaultParamsQi(logn)
        _, Pr = params.DefaultParamsPi(logn)
        Lr, Kr = len(Qr), len(Pr)
        Br = max(2, args.rings_bytes // (8 * Nr * Lr)) & ~1
        cq, cp = ring.NewContextWithParams(Nr, Qr, device=local), ring.NewContextWithParams(Nr, Pr, device=local)
        pair = sampling.uniform_poly(Qr, Nr, 2, seed=(logn << 8) ^ rank)
        a = cq.NewPoly(Br).set(np.concatenate([pair] * (Br // 2)))
        b, c, pp = cq.NewPoly(Br), cq.NewPoly(Br), cp.NewPoly(Br)
        cq.Copy(a, b)
        be = ring.NewFastBasisExtender(cq, cp)
        return dict(logn=logn, Nr=Nr, Qr=Qr, Pr=Pr, Lr=Lr, Kr=Kr, Br=Br, cq=cq, cp=cp, pair=pair, a=a, b=b, c=c, pp=pp, be=be)

    def ring_run(r):
        logn, Nr, Qr, Pr, Lr, Kr, Br, cq, pair, a, b, c, pp, be = (r[k] for k in ("logn", "Nr", "Qr", "Pr", "Lr", "Kr", "Br", "cq", "pair", "a", "b", "c", "pp", "be"))
        row = {"ring": "R%d" % logn, "N": Nr, "limbs": Lr, "polys_per_gpu": Br, "n_gpus": world}
        checks = {}

        def rescale():
            # Context.DivRoundByLastModulusNTT (ring/ring_scaling.go:72) works in place and drops the last limb: the operand is the
            # copy in c with its limb count put back up (the values differ from call to call; the checked call starts from a)
            nat.check(nat.lib().lr_poly_set_limbs(c.h, Lr))
            cq.DivRoundByLastModulusNTT(c)

        for name, fn, nbytes, reps in (("ntt", lambda: cq.NTT(a, c), 16 * Nr * Lr * Br, 20),
                                       ("intt", lambda: cq.InvNTT(a, c), 16 * Nr * Lr * Br, 20),
                                       ("mulcoeffs_montgomery", lambda: cq.MulCoeffsMontgomery(a, b, c), 24 * Nr * Lr * Br, 20),
                                       ("modup_split_qp", lambda: be.ModUpSplitQP(Lr - 1, a, pp), 8 * Nr * (Lr + Kr) * Br, 10),
                                       ("div_round_by_last_modulus_ntt", rescale, 8 * Nr * (2 * Lr - 1) * Br, 10)):
            t_up = time.perf_counter()        # bring the device clock up (see warm_clock): ~0.1 s of the same launches
            while time.perf_counter() - t_up < 0.1:
                for _ in range(reps):
                    fn()
                cq.Sync()
            barrier()
            cq.TimerStart()
            for _ in range(reps):
                fn()
            ms = all_max(cq.TimerStop() / reps)
            row[name] = {"poly_per_s": Br * world / (ms * 1e-3), "ms": ms, "frac_hbm_per_gpu": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if name in ("ntt", "intt"):
                row[name]["limb_ntt_per_s"] = Br * world * Lr / (ms * 1e-3)
                row[name]["kernel"] = cq.last_ntt_kernel()
            if rank == 0:      # the last poly of the output against the oracle (checker only)
                def check(name=name):
                    ocr = oracle.Context(Nr, Qr)
                    x = pair[(Br - 1) % 2]
                    if name == "modup_split_qp":
                        want = oracle.BasisExtender(ocr, oracle.Context(Nr, Pr)).modup_split_qp(Lr - 1, x)
                        got = np.stack(pp.get_limb_slices(Br - 1))
                    elif name == "div_round_by_last_modulus_ntt":
                        nat.check(nat.lib().lr_poly_set_limbs(c.h, Lr))
                        cq.Copy(a, c)
                        cq.DivRoundByLastModulusNTT(c)
                        want = ocr.rescale_op("oc_div_round_by_last_modulus_ntt", x)
                        got = np.stack(c.get_limb_slices(Br - 1))[:Lr - 1]
                        nat.check(nat.lib().lr_poly_set_limbs(c.h, Lr))
                    else:
                        want = {"ntt": ocr.ntt, "intt": ocr.intt, "mulcoeffs_montgomery": lambda v: ocr.ewise("MUL_MONT", v, v)}[name](x)
                        got = np.stack(c.get_limb_slices(Br - 1))
                    return bool(np.array_equal(got, want))
                checks[name], err = checked(check, None)
                if err:
                    row.setdefault("check_errors", {})[name] = err
        if rank == 0:
            row["bit_exact"] = all(v is True for v in checks.values())
            if want_cpu:
                # the CPU restatement beside every ring ("CPU (1 / T threads) next to every ring R13...R16";
                # ring/ring_benchmark_test.go:154-186 NTT, :349-404 DivRoundByLastModulusNTT): short samples, ~1 s each
                ocr = oracle.Context(Nr, Qr)
                x0 = pair[0]

                def mk_ntt(i):
                    xi, yi = x0.copy(), np.empty_like(x0)
                    olib = oracle.lib()
                    return lambda: olib.oc_ntt_lvl(ocr.h, Lr - 1, xi.ctypes.data, yi.ctypes.data)

                def mk_rescale(i):
                    xi = x0.copy()
                    return lambda: ocr.rescale_op("oc_div_round_by_last_modulus_ntt", xi)
                row["ntt"]["cpu_baseline"] = cpu_baseline(mk_ntt, Lr, "limb-NTT/s", "oracle Context.NTT on R%d, one poly per call" % logn, 1.0)
                row["div_round_by_last_modulus_ntt"]["cpu_baseline"] = cpu_baseline(
                    mk_rescale, 1, "poly/s", "oracle DivRoundByLastModulusNTT on R%d, one poly per call" % logn, 1.0)
        return row

    # CKKS MulRelin on the default parameter sets of the same degrees (ckks/params.go:36-76; N = 2^16 is the config-5 leg), every rank
    # on its own batch, whole-job products per second
    def mulrelin_set_prepare(name, mb):
        mN, mQ, mP = params.ckks_moduli(name)
        nq, np_ = len(mQ), len(mP)
        mcQ, mcP = ring.NewContextWithParams(mN, mQ, device=local), ring.NewContextWithParams(mN, mP, device=local)
        mplan = ring.CkksPlan(mcQ, mcP, mb)
        mlevel, mbeta = nq - 1, -(-nq // np_)
        key_h = sampling.uniform_poly(mQ + mP, mN, 2 * mbeta, seed=9)
        key = mplan.NewSwitchingKey().set(key_h)
        ops = [sampling.uniform_poly(mQ, mN, 2, seed=(40 + k) ^ (rank << 8)) for k in range(4)]
        tile = lambda x: np.concatenate([x] * (mb // 2))
        c0 = (mcQ.NewPoly(mb).set(tile(ops[0])), mcQ.NewPoly(mb).set(tile(ops[1])))
        c1 = (mcQ.NewPoly(mb).set(tile(ops[2])), mcQ.NewPoly(mb).set(tile(ops[3])))
        co = (mcQ.NewPoly(mb), mcQ.NewPoly(mb))
        return dict(name=name, mb=mb, mN=mN, mQ=mQ, mP=mP, nq=nq, np_=np_, mcQ=mcQ, mcP=mcP, mplan=mplan, mlevel=mlevel, mbeta=mbeta,
                    key_h=key_h, key=key, ops=ops, c0=c0, c1=c1, co=co)

    def mulrelin_set_run(m):
        name, mb, mN, mQ, mP, nq, np_, mcQ, mplan, mlevel, mbeta, key_h, key, ops, c0, c1, co = (m[k] for k in (
            "name", "mb", "mN", "mQ", "mP", "nq", "np_", "mcQ", "mplan", "mlevel", "mbeta", "key_h", "key", "ops", "c0", "c1", "co"))
        fn = lambda: mplan.MulRelin(mlevel, c0, c1, key, co)
        t_up = time.perf_counter()
        while time.perf_counter() - t_up < 0.1:
            for _ in range(3):
                fn()
            mcQ.Sync()
        barrier()
        mcQ.TimerStart()
        for _ in range(5):
            fn()
        ms = all_max(mcQ.TimerStop() / 5)
        row = {"params": name, "N": mN, "limbs_Q": nq, "limbs_P": np_, "batch_per_gpu": mb, "n_gpus": world, "ms_per_batch": ms,
               "mulrelin_per_s": mb * world / (ms * 1e-3),
               "frac_hbm_per_gpu": mulrelin_bytes(mN, nq, np_, mb) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if rank == 0:
            def check():
                op = oracle.CkksPlan(oracle.Context(mN, mQ), oracle.Context(mN, mP))
                j = (mb - 1) % 2
                want = op.mulrelin(mlevel, np.stack([ops[0][j], ops[1][j]]), np.stack([ops[2][j], ops[3][j]]), key_h.reshape(mbeta, 2, nq + np_, mN))
                return bool(np.array_equal(np.stack(co[0].get_limb_slices(mb - 1)), want[0]) and
                            np.array_equal(np.stack(co[1].get_limb_slices(mb - 1)), want[1]))
            row["bit_exact"], err = checked(check, None)
            if err:
                row["check_error"] = err
        return row

    # --------------------------------------------------------------------------------------------------------------------
    # headline: forward NTT on R15
    # --------------------------------------------------------------------------------------------------------------------
    N, moduli = params.DefaultParamsQi(args.logn)
    L = len(moduli)
    B = args.batch
    _, my_polys = shard_units(B * world, rank, world)   # weak scaling: B per GPU
    ctx = ring.NewContextWithParams(N, moduli, device=local)
    fwd_variant, inv_variant = ctx.ntt_variants()
    if 12 <= args.logn <= 16 and fwd_variant < 0 and not os.environ.get("LR_NO_ASM"):
        raise SystemExit("bench.py: the context did not select an assembly NTT kernel (variant -1); refusing to report the C++ fallback as the headline")
    # synthetic operands: a few distinct polys tiled over the batch (generation cost), resident before timing
    base = sampling.uniform_poly(moduli, N, min(my_polys, 8), seed=0x4C415454 ^ rank)
    host = np.concatenate([base] * (-(-my_polys // base.shape[0])))[:my_polys]
    src, dst = ctx.NewPoly(my_polys).set(host), ctx.NewPoly(my_polys)
    del host

    def step():
        ctx.NTT(src, dst)

    warm_clock(step, ctx)
    # the K timed launches are bracketed by HIP events on the launch stream as well (device-side duration)
    seconds, dev_ms = timed_region(step, args.steps, args.warmup, sync, barrier, all_max, ctx.TimerStart, ctx.TimerStop)
    kernel_ms = dev_ms / args.steps
    kernel_name = ctx.last_ntt_kernel()
    progress("headline NTT timed: %.4f ms per launch, kernel %s" % (kernel_ms, kernel_name))

    power = sample_power(step, ctx.Sync) if (rank == 0 and world == 1 and not args.no_traffic) else None
    # parity spot-check inside the bench: first and last poly against the oracle (checker only)
    bit_exact = None
    oc = None
    if rank == 0:
        oc = oracle.Context(N, moduli)
        full = dst.get().reshape(my_polys, L, N)
        bit_exact = bool(np.array_equal(full[0], oc.ntt(base[0])) and
                         np.array_equal(full[my_polys - 1], oc.ntt(base[(my_polys - 1) % base.shape[0]])))
        del full

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
                     "frac": achieved / HBM_PEAK_GBS,
                     # PMC counters need their own rocprofv3 passes: filled at the end of the run by two child runs on the same
                     # kernel and shape (measure_hbm_traffic); null if that is skipped or fails, the committed figure is in the file
                     "traffic": None, "traffic_profile": "profiles/r04/pmc_hbm.json",
                     "kernel": kernel_name, "asm_variant": fwd_variant, "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": ntt_bytes(N, L, my_polys),
                     # the north star says "HBM-read roofline"; `achieved` counts read + write, this is the read half
                     "achieved_read_only": achieved / 2},
    }

    def timed_on(c, fn, reps):
        # the CPU baselines before a leg leave the device idle for seconds: bring its clock up first (see warm_clock; three launches were
        # not enough -- the same kernel read 7-15 % slower here than in the R13..R16 table of the same run)
        t_up = time.perf_counter()
        while time.perf_counter() - t_up < 0.15:
            for _ in range(max(3, reps // 2)):
                fn()
            c.Sync()
        c.TimerStart()
        for _ in range(reps):
            fn()
        return c.TimerStop() / reps

    pipeline_ms = {}
    if rank == 0 and not args.no_ckks:
        # second half of BASELINE.json's metric: CKKS MulRelin (ckks/evaluator.go:1016) at DefaultParams[PN15QP880],
        # device-resident batch of independent ciphertexts, synthetic operands and evaluation key
        cN, cQm, cPm = params.ckks_moduli("PN15QP880")
        cB = args.ckks_batch
        nq, np_ = len(cQm), len(cPm)
        ccQ, ccP = ring.NewContextWithParams(cN, cQm, device=local), ring.NewContextWithParams(cN, cPm, device=local)
        plan = ring.CkksPlan(ccQ, ccP, cB)
        clevel = nq - 1
        cbeta = -(-nq // np_)
        evk_h = sampling.uniform_poly(cQm + cPm, cN, 2 * cbeta, seed=9)
        evk = plan.NewSwitchingKey().set(evk_h)
        cbase = [sampling.uniform_poly(cQm, cN, 2, seed=3 + k) for k in range(4)]          # a0, a1, b0, b1 for two distinct products
        tile = lambda x: np.concatenate([x] * (-(-cB // 2)))[:cB]
        ct0 = (ccQ.NewPoly(cB).set(tile(cbase[0])), ccQ.NewPoly(cB).set(tile(cbase[1])))
        ct1 = (ccQ.NewPoly(cB).set(tile(cbase[2])), ccQ.NewPoly(cB).set(tile(cbase[3])))
        cto = (ccQ.NewPoly(cB), ccQ.NewPoly(cB))
        cms = timed_on(ccQ, lambda: plan.MulRelin(clevel, ct0, ct1, evk, cto), 10)
        progress("ckks MulRelin PN15QP880 timed: %.3f ms per batch of %d" % (cms, cB))
        oplan = oracle.CkksPlan(oracle.Context(cN, cQm), oracle.Context(cN, cPm))
        idx = cB - 1
        want = oplan.mulrelin(clevel, np.stack([cbase[0][idx % 2], cbase[1][idx % 2]]), np.stack([cbase[2][idx % 2], cbase[3][idx % 2]]),
                              evk_h.reshape(cbeta, 2, nq + np_, cN))
        got0, got1 = cto[0].get().reshape(cB, nq, cN)[idx], cto[1].get().reshape(cB, nq, cN)[idx]
        alg = mulrelin_bytes(cN, nq, np_, cB) / (cms * 1e-3) / 1e9
        out["ckks_mulrelin"] = {"value": cB / (cms * 1e-3), "unit": "MulRelin/s", "batch": cB, "ms_per_batch": cms,
                                "params": "PN15QP880 (N=2^15, 18 Q limbs + 3 P limbs, beta=6), level 17",
                                "bit_exact": bool(np.array_equal(got0, want[0]) and np.array_equal(got1, want[1])),
                                "roofline": {"bound": "hbm", "achieved": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / HBM_PEAK_GBS,
                                             "traffic": None, "traffic_profile": "profiles/r04/mulrelin_pmc_hbm.json",
                                             "kernel": "pipeline of launches; per-kernel split in profiles/r04/mulrelin_kernel_stats.csv",
                                             "pipeline_ms": cms, "algorithmic_bytes_per_product": mulrelin_bytes(cN, nq, np_)}}
        if want_cpu:
            def mk_mulrelin(i):
                op = oracle.CkksPlan(oracle.Context(cN, cQm), oracle.Context(cN, cPm))
                a, b, k = np.stack([cbase[0][0], cbase[1][0]]), np.stack([cbase[2][0], cbase[3][0]]), evk_h.reshape(cbeta, 2, nq + np_, cN)
                return lambda: op.mulrelin(clevel, a, b, k)
            out["ckks_mulrelin"]["cpu_baseline"] = cpu_baseline(mk_mulrelin, 1, "MulRelin/s", "oracle MulRelin PN15QP880 level 17", 4.0)
        del plan, ct0, ct1, cto, evk

        # BASELINE.json config 4 (BFV DefaultParams[PN14QP438] Evaluator.Mul) and every other pipeline entry point the reference benchmarks
        # (tools/bench_legs.py: ring/ring_benchmark_test.go:310-404, ckks/ckks_benchmarks_test.go:79-240, bfv/bfv_benchmark_test.go:133-162):
        # one timed call over a device-resident batch each, oracle check of the last unit, CPU oracle beside it
        bench_legs = load_bench_legs()
        only = {"bfv_mul"} if (world > 1 or args.no_pipelines) else (set(x for x in args.pipelines.split(",") if x) or None)
        kits, groups = bench_legs.build_legs(pkg, oracle, device=local, only=only)
        pipes = {}
        for gname, make in groups:
            for leg in make():
                ms = timed_on(leg.sync, leg.run, leg.reps)
                leg.after()
                ok, err = checked(leg.check, None)
                leg.after()
                units = leg.units_per_call
                ach = leg.bytes_per_unit * units / (ms * 1e-3) / 1e9
                obj = {"value": units / (ms * 1e-3), "unit": leg.unit, "batch": leg.batch, "ms_per_batch": ms, "params": leg.params,
                       "reference_benchmark": leg.ref, "bit_exact": ok,
                       "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                    "kernel": "pipeline of launches (per-kernel split in `kernels` once the counters ran)", "pipeline_ms": ms,
                                    "algorithmic_bytes_per_unit": leg.bytes_per_unit, "units_per_call": units}}
                if err:
                    obj["check_error"] = err
                if leg.note:
                    obj["note"] = leg.note
                if leg.host_bytes_per_unit:
                    obj["pcie_inclusive"] = {"host_bytes_per_unit": leg.host_bytes_per_unit, "GB_per_s_through_the_entry_point": leg.host_bytes_per_unit * units / (ms * 1e-3) / 1e9}
                if want_cpu:
                    obj["cpu_baseline"] = cpu_baseline(leg.cpu, leg.cpu_units, leg.unit, "oracle %s, one call = %d unit(s)" % (leg.name, leg.cpu_units), 1.0)
                pipes[leg.name] = obj
                pipeline_ms[leg.name] = ms
                progress("%s: %.3f ms per call of %d, %.0f %s, %.3f of the HBM roofline, bit_exact %s" % (leg.name, ms, units, obj["value"], leg.unit, obj["roofline"]["frac"], ok))
            kits.drop(gname)
        if "bfv_mul" in pipes:
            out["bfv_mul"] = pipes.pop("bfv_mul")
        if pipes:
            out["pipelines"] = pipes
        del kits, groups

    if rank == 0 and world == 1 and not args.no_ckks and not args.no_threads:
        # The reference's concurrency model (examples/dbfv/psi/psi.go:215-233): one evaluator per goroutine, one ciphertext each.  Here: T
        # host threads, each with its OWN contexts, plan, operands and stream (a context carries its stream, so per-thread streams mean
        # per-thread contexts; the evaluation key is uploaded once per thread like every evaluator holds its own reference), batch 1
        # PN15QP880 MulRelin in a loop, no synchronisation inside the loop.  ctypes releases the GIL during a call, so the threads'
        # launches really overlap on the host.  Aggregate products per second beside the batch-128 figure of one call.
        import threading
        tN, tQ, tP = params.ckks_moduli("PN15QP880")
        tnq, tnp = len(tQ), len(tP)
        tlevel, tbeta = tnq - 1, -(-tnq // tnp)
        tkey_h = sampling.uniform_poly(tQ + tP, tN, 2 * tbeta, seed=9)
        tops = [sampling.uniform_poly(tQ, tN, 1, seed=60 + k) for k in range(4)]
        rows = []
        want_t = None
        for T in (1, 4, 16):
            workers = []
            for i in range(T):
                st_i = torch.cuda.Stream()
                wq, wp = ring.NewContextWithParams(tN, tQ, device=local), ring.NewContextWithParams(tN, tP, device=local)
                wq.SetStream(st_i.cuda_stream)
                wp.SetStream(st_i.cuda_stream)
                wplan = ring.CkksPlan(wq, wp, 1)
                wkey = wplan.NewSwitchingKey().set(tkey_h)
                wc0 = (wq.NewPoly(1).set(tops[0]), wq.NewPoly(1).set(tops[1]))
                wc1 = (wq.NewPoly(1).set(tops[2]), wq.NewPoly(1).set(tops[3]))
                wout = (wq.NewPoly(1), wq.NewPoly(1))
                wplan.MulRelin(tlevel, wc0, wc1, wkey, wout)          # warm-up: pools and scratch exist before the loop
                wq.Sync()
                workers.append((st_i, wq, wp, wplan, wkey, wc0, wc1, wout))
            iters = 200 if T == 1 else 100
            gate = threading.Barrier(T + 1)

            def loop(w):
                _, wq, _, wplan, wkey, wc0, wc1, wout = w
                gate.wait()
                for _ in range(iters):
                    wplan.MulRelin(tlevel, wc0, wc1, wkey, wout)
                wq.Sync()
            ths = [threading.Thread(target=loop, args=(w,)) for w in workers]
            for th in ths:
                th.start()
            gate.wait()
            t_w = time.perf_counter()
            for th in ths:
                th.join()
            dt_w = time.perf_counter() - t_w
            if want_t is None:
                want_t = oracle.CkksPlan(oracle.Context(tN, tQ), oracle.Context(tN, tP)).mulrelin(
                    tlevel, np.stack([tops[0][0], tops[1][0]]), np.stack([tops[2][0], tops[3][0]]), tkey_h.reshape(tbeta, 2, tnq + tnp, tN))
            ok_t = all(np.array_equal(w[7][0].get().reshape(tnq, tN), want_t[0]) and np.array_equal(w[7][1].get().reshape(tnq, tN), want_t[1]) for w in workers)
            rows.append({"threads": T, "streams": T, "batch_per_call": 1, "calls_per_thread": iters, "mulrelin_per_s": T * iters / dt_w,
                         "us_per_product_per_thread": dt_w / iters * 1e6, "bit_exact": bool(ok_t),
                         # (lr_ckks_plan_stats; forks are for a lone plan at N = 2^16 only: 0 at this parameter set)
                         "forks_per_call": workers[0][3].Stats()["forks"] / (iters + 1)})
            del workers, ths
        # the same callers through the batcher (lr_ckks_batcher_*): concurrent batch-1 calls are merged into batched launches on two lanes
        brows = []
        bat = ring.CkksBatcher(tN, tQ, tP, max_batch=64, lanes=2, device=local)
        bkey = bat.NewSwitchingKey().set(tkey_h)
        callers_ctx = ring.NewContextWithParams(tN, tQ, device=local)
        for T in (16, 64):
            callers = []
            for i in range(T):
                mk1 = lambda k: callers_ctx.NewPoly(1).set(tops[k])
                callers.append(((mk1(0), mk1(1)), (mk1(2), mk1(3)), (callers_ctx.NewPoly(1), callers_ctx.NewPoly(1))))
            bat.MulRelin(tlevel, callers[0][0], callers[0][1], bkey, callers[0][2])      # warm-up: the lanes' pools exist
            callers_ctx.Sync()
            before = bat.Stats()
            iters = 100
            gate = threading.Barrier(T + 1)

            def bloop(w):
                gate.wait()
                for _ in range(iters):
                    bat.MulRelin(tlevel, w[0], w[1], bkey, w[2])
            ths = [threading.Thread(target=bloop, args=(w,)) for w in callers]
            for th in ths:
                th.start()
            gate.wait()
            t_w = time.perf_counter()
            for th in ths:
                th.join()
            dt_w = time.perf_counter() - t_w
            after = bat.Stats()
            ok_t = all(np.array_equal(w[2][0].get().reshape(tnq, tN), want_t[0]) and np.array_equal(w[2][1].get().reshape(tnq, tN), want_t[1]) for w in callers)
            brows.append({"threads": T, "lanes": 2, "batch_per_call": 1, "calls_per_thread": iters, "mulrelin_per_s": T * iters / dt_w,
                          "launches": after["batches"] - before["batches"], "mean_batch": (after["products"] - before["products"]) / max(1, after["batches"] - before["batches"]),
                          "largest_batch": after["largest"], "bit_exact": bool(ok_t)})
            del callers, ths
        del bat
        out["evaluator_threads"] = {"params": "PN15QP880, level 17, batch 1 per call, one plan + contexts + stream per host thread", "rows": rows,
                                    "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                                    "batcher_rows": brows,
                                    "batcher": "the same batch-1 calls through lr_ckks_batcher_mulrelin: queued requests run as one batched MulRelin per free lane",
                                    "reference_model": "one evaluator per goroutine (examples/dbfv/psi/psi.go:215-233)"}
        progress("evaluator-per-thread MulRelin: " + ", ".join("T=%d %.0f/s" % (r["threads"], r["mulrelin_per_s"]) for r in rows) +
                 "; through the batcher: " + ", ".join("T=%d %.0f/s (mean batch %.1f)" % (r["threads"], r["mulrelin_per_s"], r["mean_batch"]) for r in brows))

    if rank == 0 and world == 1 and not args.no_ckks and not args.no_threads:
        # The same for the workload the reference itself pools: every task of examples/dbfv/psi/psi.go:219-228 runs evaluator.Mul and
        # evaluator.Relinearize on one BFV ciphertext pair.  T host threads, each with its own contexts, plans and stream, one pair per call
        # (direct), and the same callers through lr_bfv_batcher_* (two lanes).  PN14QP438 (BASELINE config 4's set), pairs per second.
        import threading
        fN, fQ, fP, fM = params.bfv_moduli("PN14QP438")
        fQ, fP, fM = list(fQ), list(fP), list(fM)
        fnq, fnp = len(fQ), len(fP)
        fbeta = -(-fnq // fnp)
        fkey_h = sampling.uniform_poly(fQ + fP, fN, 2 * fbeta, seed=19)
        fops = [sampling.uniform_poly(fQ, fN, 1, seed=80 + k) for k in range(4)]
        want_mul = oracle.BfvPlan(oracle.Context(fN, fQ), oracle.Context(fN, fM), 65537).mul(np.stack([fops[0][0], fops[1][0]]), np.stack([fops[2][0], fops[3][0]]))
        want_lin = oracle.CkksPlan(oracle.Context(fN, fQ), oracle.Context(fN, fP)).bfv_relinearize(want_mul, fkey_h.reshape(fbeta, 2, fnq + fnp, fN))

        def run_threads(T, make, iters):
            workers = [make(i) for i in range(T)]
            gate = threading.Barrier(T + 1)

            def loop(w):
                gate.wait()
                for _ in range(iters):
                    w["call"]()
                w["ctx"].Sync()
            ths = [threading.Thread(target=loop, args=(w,)) for w in workers]
            for th in ths:
                th.start()
            gate.wait()
            t_w = time.perf_counter()
            for th in ths:
                th.join()
            dt_w = time.perf_counter() - t_w
            ok = all(np.array_equal(w["lin"][k].get().reshape(fnq, fN), want_lin[k]) for w in workers for k in range(2))
            return T * iters / dt_w, bool(ok)

        def make_direct(i):
            st_i = torch.cuda.Stream()
            cq, cp, cm = (ring.NewContextWithParams(fN, m, device=local) for m in (fQ, fP, fM))
            for c in (cq, cp, cm):
                c.SetStream(st_i.cuda_stream)
            mul, ks = ring.BfvPlan(cq, cm, 65537, 1), ring.CkksPlan(cq, cp, 1)
            key = ks.NewSwitchingKey().set(fkey_h)
            c0, c1 = (cq.NewPoly(1).set(fops[0]), cq.NewPoly(1).set(fops[1])), (cq.NewPoly(1).set(fops[2]), cq.NewPoly(1).set(fops[3]))
            d2, lin = (cq.NewPoly(1), cq.NewPoly(1), cq.NewPoly(1)), (cq.NewPoly(1), cq.NewPoly(1))

            def call():
                mul.Mul(c0, c1, d2)
                ks.BfvRelinearize(d2, key, lin)
            call()
            cq.Sync()
            return {"call": call, "ctx": cq, "lin": lin, "keep": (st_i, cp, cm, mul, ks, key, c0, c1, d2)}
        frows = []
        for T in (1, 4, 16):
            rate, ok = run_threads(T, make_direct, 200 if T == 1 else 100)
            frows.append({"threads": T, "streams": T, "pairs_per_call": 1, "mul_relin_per_s": rate, "bit_exact": ok})
        fbat = ring.BfvBatcher(fN, fQ, fP, fM, 65537, max_batch=64, lanes=2, device=local)
        fbkey = fbat.NewSwitchingKey().set(fkey_h)
        fctx = ring.NewContextWithParams(fN, fQ, device=local)

        def make_batched(i):
            c0, c1 = (fctx.NewPoly(1).set(fops[0]), fctx.NewPoly(1).set(fops[1])), (fctx.NewPoly(1).set(fops[2]), fctx.NewPoly(1).set(fops[3]))
            d2, lin = (fctx.NewPoly(1), fctx.NewPoly(1), fctx.NewPoly(1)), (fctx.NewPoly(1), fctx.NewPoly(1))

            def call():
                fbat.Mul(c0, c1, d2)
                fbat.Relinearize(d2, fbkey, lin)
            call()
            return {"call": call, "ctx": fctx, "lin": lin, "keep": (c0, c1, d2)}
        fbrows = []
        for T in (16, 64):
            before = fbat.Stats()
            rate, ok = run_threads(T, make_batched, 100)
            after = fbat.Stats()
            fbrows.append({"threads": T, "lanes": 2, "pairs_per_call": 1, "mul_relin_per_s": rate, "bit_exact": ok,
                           "mean_batch": (after["products"] - before["products"]) / max(1, after["batches"] - before["batches"]), "largest_batch": after["largest"]})
        del fbat
        out["evaluator_threads_bfv"] = {"params": "bfv PN14QP438, one ciphertext pair per call: evaluator.Mul then evaluator.Relinearize", "rows": frows, "batcher_rows": fbrows,
                                        "batched_call_of_256": "bfv_mul + pipelines.bfv_relinearize of this line: one call over 256 pairs",
                                        "reference_model": "the pooled task of examples/dbfv/psi/psi.go:219-228"}
        progress("evaluator-per-thread BFV Mul + Relinearize: " + ", ".join("T=%d %.0f/s" % (r["threads"], r["mul_relin_per_s"]) for r in frows) +
                 "; through the batcher: " + ", ".join("T=%d %.0f/s (mean batch %.1f)" % (r["threads"], r["mul_relin_per_s"], r["mean_batch"]) for r in fbrows))

    if rank == 0 and world == 1 and not args.no_ckks and not args.no_threads:
        # The same calling shape from a plain C++ host (tools/batcher_bench.cpp over include/lattigo_ring.h: no interpreter, no GIL in the
        # loop): the Python threads above lose throughput from T = 4 to T = 16, a C++ host does not.  Built with g++ at run time, run as a
        # child process; skipped with its reason if the toolchain is missing.
        try:
            import subprocess
            tool = _tool("batcher_bench.py")
            subprocess.run([sys.executable, tool, "--build"], check=True, timeout=120, capture_output=True)
            res = subprocess.run([sys.executable, tool, "PN15QP880", "1,4,16,64", "60", "2", "64"], check=True, timeout=240, capture_output=True, text=True)
            nrows = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
            out["evaluator_threads_native"] = {"params": "PN15QP880, level 17, batch 1 per call; C++ host threads (tools/batcher_bench.cpp), 60 calls per thread",
                                               "rows": nrows, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
            progress("the same from a C++ host: " + ", ".join("%s T=%d %.0f/s" % (r["how"], r["threads"], r["calls_per_s"]) for r in nrows))
        except Exception as ex:     # noqa: BLE001 -- optional leg
            out["evaluator_threads_native"] = {"skipped": str(ex)[-300:]}

    if not args.no_extras and rank == 0:
        # the other kernels BASELINE.json's north_star asks throughput for, same ring, same resident batch; after timing,
        # the last poly of every output is compared with the oracle
        extras = {}
        reps = max(10, min(args.steps, 50))
        last = my_polys - 1
        x_last = base[last % base.shape[0]]
        timed = lambda fn: timed_on(ctx, fn, reps)

        def leg_roofline(nbytes, ms, kernel):
            ach = nbytes / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "kernel": kernel, "kernel_ms": ms, "algorithmic_bytes_per_launch": nbytes}

        ms = timed(lambda: ctx.InvNTT(src, dst))
        extras["intt"] = {"limb_ntt_per_s": my_polys * L / (ms * 1e-3), "ms": ms, "kernel": ctx.last_ntt_kernel(),
                          "frac_hbm": ntt_bytes(N, L, my_polys) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "roofline": leg_roofline(ntt_bytes(N, L, my_polys), ms, ctx.last_ntt_kernel()),
                          "bit_exact": bool(np.array_equal(dst.get().reshape(my_polys, L, N)[last], oc.intt(x_last)))}
        if want_cpu:
            def mk_intt(i):
                a, b = x_last.copy(), np.empty_like(x_last)
                lib = oracle.lib()
                return lambda: lib.oc_intt_lvl(oc.h, L - 1, a.ctypes.data, b.ctypes.data)
            extras["intt"]["cpu_baseline"] = cpu_baseline(mk_intt, L, "limb-NTT/s", "oracle Context.InvNTT on R15, one poly per call", 1.0)
        ctx.Copy(src, dst)
        ms = timed(lambda: ctx.MulCoeffsMontgomery(src, dst, dst))
        # the timed calls chain dst <- MRed(src, dst) an unknown number of times (clock warm-up by time): the check starts over from
        # dst = src and replays three calls on the oracle for the checked poly
        ctx.Copy(src, dst)
        for _ in range(3):
            ctx.MulCoeffsMontgomery(src, dst, dst)
        chain = x_last.copy()
        for _ in range(3):
            chain = oc.ewise("MUL_MONT", x_last, chain)
        extras["mulcoeffs_montgomery"] = {"poly_per_s": my_polys / (ms * 1e-3), "ms": ms,
                                          "frac_hbm": 24 * N * L * my_polys / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "bit_exact": bool(np.array_equal(dst.get().reshape(my_polys, L, N)[last], chain))}
        if want_cpu:
            def mk_mul(i):
                a, b, o = x_last.copy(), x_last.copy(), np.empty_like(x_last)
                lib = oracle.lib()
                return lambda: lib.oc_ewise(oc.h, oracle.OP["MUL_MONT"], L - 1, a.ctypes.data, b.ctypes.data, o.ctypes.data, None)
            extras["mulcoeffs_montgomery"]["cpu_baseline"] = cpu_baseline(mk_mul, 1, "poly/s", "oracle MulCoeffsMontgomery R15", 2.0)
        # Context.DivRoundByLastModulusNTT (ring/ring_scaling.go:72; the reference benches it at ring/ring_benchmark_test.go:349-404):
        # in place, drops the last limb; the limb count goes back up before every call (values differ from call to call, the checked
        # call starts from src)
        def rescale_once():
            nat.check(nat.lib().lr_poly_set_limbs(dst.h, L))
            ctx.DivRoundByLastModulusNTT(dst)
        ctx.Copy(src, dst)
        ms = timed(rescale_once)
        nat.check(nat.lib().lr_poly_set_limbs(dst.h, L))
        ctx.Copy(src, dst)
        rescale_once()
        got_rs = dst.get_limb_slices(last)[:L - 1]
        nat.check(nat.lib().lr_poly_set_limbs(dst.h, L))
        extras["div_round_by_last_modulus_ntt"] = {
            "poly_per_s": my_polys / (ms * 1e-3), "ms": ms, "frac_hbm": 8 * N * (2 * L - 1) * my_polys / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_poly": 8 * N * (2 * L - 1),
            "bit_exact": bool(np.array_equal(np.stack(got_rs), oc.rescale_op("oc_div_round_by_last_modulus_ntt", x_last)))}
        if want_cpu:
            def mk_rescale(i):
                xi = x_last.copy()
                return lambda: oc.rescale_op("oc_div_round_by_last_modulus_ntt", xi)
            extras["div_round_by_last_modulus_ntt"]["cpu_baseline"] = cpu_baseline(mk_rescale, 1, "poly/s", "oracle DivRoundByLastModulusNTT R15", 2.0)
        _, pmod = params.DefaultParamsPi(args.logn)
        ctxP = ring.NewContextWithParams(N, pmod, device=local)
        bext = ring.NewFastBasisExtender(ctx, ctxP)
        outP = ctxP.NewPoly(my_polys)
        ms = timed(lambda: bext.ModUpSplitQP(L - 1, src, outP))
        obe = oracle.BasisExtender(oc, oracle.Context(N, pmod))
        extras["modup_split_qp"] = {"poly_per_s": my_polys / (ms * 1e-3), "ms": ms,
                                    "frac_hbm": 8 * N * (L + len(pmod)) * my_polys / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "%d -> %d limbs, %d modular multiplies per coefficient (compute-bound)" %
                                            (L, len(pmod), L + L * len(pmod)),
                                    "roofline": leg_roofline(8 * N * (L + len(pmod)) * my_polys, ms, "ext_wide_kernel / ext_sum_kernel (lr_bext.hip)"),
                                    "bit_exact": bool(np.array_equal(outP.get().reshape(my_polys, len(pmod), N)[last], obe.modup_split_qp(L - 1, x_last)))}
        if want_cpu:
            def mk_modup(i):
                be = oracle.BasisExtender(oracle.Context(N, moduli), oracle.Context(N, pmod))
                return lambda: be.modup_split_qp(L - 1, x_last)
            extras["modup_split_qp"]["cpu_baseline"] = cpu_baseline(mk_modup, 1, "poly/s", "oracle ModUpSplitQP R15 16->16", 3.0)
        del outP, bext
        if 12 <= args.logn <= 15:
            # the same transform on CKKS-size moduli (DefaultParams[PN15QP880]'s first limbs at N = 2^15: one of 50 bits, the rest
            # 40): limbs below 2^46 run on the FP64 body of the dual kernels, the others on the integer body beside it
            cq = list(params.ckks_moduli("PN15QP880")[1][:L]) if args.logn == 15 else params.GenerateNTTPrimes(40, args.logn, L)
            ctxC = ring.NewContextWithParams(N, cq, device=local)
            occ = oracle.Context(N, cq)
            cb = sampling.uniform_poly(cq, N, min(my_polys, 2), seed=11)
            csrc = ctxC.NewPoly(my_polys).set(np.concatenate([cb] * (-(-my_polys // cb.shape[0])))[:my_polys])
            cdst = ctxC.NewPoly(my_polys)
            for name, fn, ofn in (("ntt", lambda: ctxC.NTT(csrc, cdst), occ.ntt), ("intt", lambda: ctxC.InvNTT(csrc, cdst), occ.intt)):
                ms = timed_on(ctxC, fn, reps)
                extras[name + "_ckks_moduli"] = {"limb_ntt_per_s": my_polys * L / (ms * 1e-3), "ms": ms,
                                                 "frac_hbm": ntt_bytes(N, L, my_polys) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "moduli_bits": [int(q).bit_length() for q in cq],
                                                 "asm_variants": list(ctxC.ntt_variants()), "kernel": ctxC.last_ntt_kernel(),
                                                 "roofline": leg_roofline(ntt_bytes(N, L, my_polys), ms, ctxC.last_ntt_kernel()),
                                                 "bit_exact": bool(np.array_equal(cdst.get().reshape(my_polys, L, N)[last], ofn(cb[last % cb.shape[0]])))}
                if want_cpu:
                    def mk_c(i, name=name):
                        a, b = cb[0].copy(), np.empty_like(cb[0])
                        lib = oracle.lib()
                        f = lib.oc_ntt_lvl if name == "ntt" else lib.oc_intt_lvl
                        return lambda: f(occ.h, L - 1, a.ctypes.data, b.ctypes.data)
                    extras[name + "_ckks_moduli"]["cpu_baseline"] = cpu_baseline(mk_c, L, "limb-NTT/s", "oracle Context.%s on the CKKS moduli, one poly per call" % ("NTT" if name == "ntt" else "InvNTT"), 1.0)
            del csrc, cdst, ctxC
        out["extras"] = extras
        progress("extras timed")

    del src, dst

    if not args.no_rings:
        progress("rings R13..R16: NTT / InvNTT / MulCoeffsMontgomery / ModUpSplitQP / DivRoundByLastModulusNTT on every rank")
        rows = []
        for lg in (13, 14, 15, 16):
            secondary("rings", lambda lg=lg: ring_prepare(lg), ring_run, store=rows)
        if rank == 0:
            out["rings"] = rows
        progress("rings timed")
        if not args.no_ckks:
            rows = []
            for sname, mb in (("PN13QP218", 512), ("PN14QP438", 256), ("PN15QP880", 128)):
                secondary("mulrelin_sets", lambda sname=sname, mb=mb: mulrelin_set_prepare(sname, mb), mulrelin_set_run, store=rows)
            if rank == 0:
                out["mulrelin_sets"] = rows
            progress("MulRelin on PN13QP218 / PN14QP438 / PN15QP880 timed on every rank")
    if not args.no_config5 and not args.no_ckks:
        progress("config 5 leg: PN16QP1761, %d products per GPU" % args.config5_units)
        secondary("config5", config5_prepare, lambda c: config5_run(c, 3, 1))
        progress("config 5 leg done")

    if rank == 0 and world == 1 and not use_dist and not args.no_config5 and not args.no_ckks:
        # the same leg from a plain C++ host: ONE process, one host thread per device, the C ABI only, lr_poly_copy_peer for the gather
        # (tools/multi_gpu_bench.cpp; SURVEY 8(e), the reference's goroutine-per-evaluator model).  G = min(8, devices visible to this process).
        def single_process_leg():
            import importlib.util
            spec = importlib.util.spec_from_file_location("multi_gpu_bench", _tool("multi_gpu_bench.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            res = mod.run(["--gpus", 8, "--units", args.config5_units, "--chunk", args.config5_chunk, "--steps", 3, "--warmup", 1], timeout=600)
            if res.returncode != 0:
                raise RuntimeError("multi_gpu_bench exited with %d: %s" % (res.returncode, res.stderr[-300:]))
            return json.loads(res.stdout.strip().splitlines()[-1])
        out["config5_single_process"], err = checked(single_process_leg, None)
        if err:
            out["config5_single_process"] = {"error": err}
        progress("config 5 from one process through the C ABI: %s" % (out["config5_single_process"].get("value") or out["config5_single_process"].get("error")))

    if rank == 0 and world == 1 and not use_dist and not args.no_traffic and B == (1 << 30) // (8 * N * L):
        # counters need their own rocprofv3 passes: three child runs per kernel / pipeline (FETCH_SIZE, WRITE_SIZE, and the vector-issue
        # set SQ_INSTS_VALU + SQ_WAVES + GRBM_GUI_ACTIVE) on the same shapes, after all timing
        progress("counters of one %s launch: three rocprofv3 --pmc child runs" % kernel_name)
        how = "rocprofv3 --pmc child passes over tools/dbg/%s in this run; 2 x FETCH_SIZE + WRITE_SIZE (KB) per launch, median"
        traffic, detail, valu = measure_kernel([_tool("pmc_run.py"), str(args.logn), "qi60", "ntt"], kernel_name, kernel_ms)
        finish_roofline(out["roofline"], traffic, detail, valu, how % ("pmc_run.py %d qi60 ntt" % args.logn), ntt_bytes(N, L, my_polys))
        progress("traffic: %s" % (("%.4f x algorithmic, vector issue %.2f of its ceiling at %.0f MHz" % (
            traffic / ntt_bytes(N, L, my_polys), valu["issue_frac_sustained"], valu["sclk_MHz"])) if traffic is not None else detail))
        ex = out.get("extras", {})
        for key, op, match in (("intt", "intt", None), ("modup_split_qp", "modup", "ext_")):
            if key in ex and "roofline" in ex[key]:
                r = ex[key]["roofline"]
                t, d, v = measure_kernel([_tool("pmc_run.py"), str(args.logn), "qi60", op], match or r["kernel"], r["kernel_ms"])
                finish_roofline(r, t, d, v, how % ("pmc_run.py %d qi60 %s" % (args.logn, op)), r["algorithmic_bytes_per_launch"])
        for key, op in (("ntt_ckks_moduli", "ntt"), ("intt_ckks_moduli", "intt")):
            if key in ex and "roofline" in ex[key]:
                r = ex[key]["roofline"]
                t, d, v = measure_kernel([_tool("pmc_run.py"), str(args.logn), "ckks", op], r["kernel"], r["kernel_ms"])
                finish_roofline(r, t, d, v, how % ("pmc_run.py %d ckks %s" % (args.logn, op)), r["algorithmic_bytes_per_launch"])
        progress("counters of InvNTT / ModUpSplitQP / the CKKS-moduli transforms collected")
        for key, pname, pb, pk in (("ckks_mulrelin", "PN15QP880", 64, 4), ("config5", "PN16QP1761", 32, 2)):
            if isinstance(out.get(key), dict) and "roofline" in out[key]:
                mr = out[key]["roofline"]
                per_product_ms = mr["pipeline_ms"] / (out[key].get("batch") or out[key].get("units_per_gpu") or 1)
                mt, md, mv = measure_pipeline([_tool("mulrelin_pmc.py"), pname, str(pb), str(pk)], per_product_ms)
                finish_roofline(mr, mt, md, mv, "rocprofv3 --pmc child passes over tools/dbg/mulrelin_pmc.py %s %d %d in this run, all kernels summed, per product" % (pname, pb, pk),
                                mr["algorithmic_bytes_per_product"])
                progress("%s traffic: %s" % (pname, ("%.1f MB per product = %.3f x algorithmic" % (mt / 1e6, mt / mr["algorithmic_bytes_per_product"])) if mt is not None else md))
        if pipeline_ms:
            progress("counters of the %d pipeline legs: three rocprofv3 --pmc child runs over tools/dbg/legs_pmc.py" % len(pipeline_ms))
            res, why = measure_legs(list(pipeline_ms), pipeline_ms)
            for name in pipeline_ms:
                obj = out["bfv_mul"] if name == "bfv_mul" else out["pipelines"][name]
                r = obj["roofline"]
                per_call = r["algorithmic_bytes_per_unit"] * r["units_per_call"]
                if res and name in res:
                    t, d, v, ks = res[name]
                    finish_roofline(r, t, d, v, "rocprofv3 --pmc child passes over tools/dbg/legs_pmc.py in this run: the leg's measured call between marker kernels, "
                                    "all its kernels summed; 2 x FETCH_SIZE + WRITE_SIZE (KB)", per_call)
                    r["kernels"] = [{"kernel": k, "launches": n, "us_under_the_profiler": us} for k, n, us in ks]
                else:
                    finish_roofline(r, None, why or "the leg is missing from the counter run", None, "", per_call)
            progress("pipeline legs counted" if res else "pipeline legs not counted: %s" % why)

    if want_cpu:
        def mk_ntt(i):
            a, b = base[0].copy(), np.empty_like(base[0])
            lib = oracle.lib()
            return lambda: lib.oc_ntt_lvl(oc.h, L - 1, a.ctypes.data, b.ctypes.data)
        out["cpu_baseline"] = cpu_baseline(mk_ntt, L, "limb-NTT/s", "oracle Context.NTT on R15 (N=2^%d, %d limbs), one poly per call" % (args.logn, L), 10.0)
    if rank == 0:
        r = out["roofline"]
        v = r.get("valu") or {}
        if v.get("instr_per_wave") and v.get("issue_frac_sustained"):
            # why the 60-bit transform sits at 0.40: the four numbers it follows from
            cpi_mix = 5.2         # clocks per instruction of an alternating v_mad_u64_u32 / v_add_u32 stream, one wave per SIMD (profiles/r02/asm_energy.txt)
            per_simd = v["instructions"] / SIMDS
            cpi_run = 4.0 / v["issue_frac_sustained"]                      # shader clocks of the launch x SIMDs / instructions: independent of the clock
            smi = (power or {}).get("sclk_MHz_smi")
            sclk_run = (smi or v["sclk_MHz"]) * 1e6                        # the clock of the TIMED launches (rocm-smi beside them); the profiled pass runs lower
            ms = lambda cpi, hz: per_simd * cpi / hz * 1e3
            fr = lambda t_ms: r["algorithmic_bytes_per_launch"] / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            r["floor"] = {"valu_instructions_per_wave": v["instr_per_wave"], "waves": v["waves"], "simds": SIMDS,
                          "clocks_per_instruction": {"in_this_run": cpi_run, "of_the_multiply_mix_alone": cpi_mix, "issue_minimum": 4.0},
                          "clocks_per_instruction_in_run": cpi_run,
                          "sclk_MHz": {"timed_launches_rocm_smi": smi, "under_the_profiler": v["sclk_MHz"], "peak": NOMINAL_SCLK_HZ / 1e6},
                          "sclk_MHz_sustained": smi or v["sclk_MHz"],
                          "package_power": power,
                          "ms": {"measured": r["kernel_ms"], "instructions_x_cpi_in_this_run_at_the_timed_clock": ms(cpi_run, sclk_run),
                                 "instructions_x_cpi_of_the_multiply_mix_at_the_timed_clock": ms(cpi_mix, sclk_run),
                                 "instructions_x_4_clocks_at_the_timed_clock": ms(4.0, sclk_run), "instructions_x_4_clocks_at_the_peak_clock": ms(4.0, NOMINAL_SCLK_HZ)},
                          "model_ms": ms(cpi_run, sclk_run), "measured_ms": r["kernel_ms"],
                          "frac": {"measured": r["frac"], "if_every_instruction_issued_in_4_clocks_at_the_timed_clock": fr(ms(4.0, sclk_run)),
                                   "if_every_instruction_issued_in_4_clocks_at_the_peak_clock": fr(ms(4.0, NOMINAL_SCLK_HZ))},
                          "reading": "four numbers give the launch time: vector instructions per wave (x waves / 1024 SIMDs), clocks per instruction (the butterfly's "
                                     "multiply mix costs 5.2 alone; 9 of its 14 instructions are the 32-bit multiplies a 60-bit Shoup product needs), the clock the "
                                     "1.4 kW package power limit leaves, and nothing else -- HBM traffic is 1.00 x algorithmic and far from its roof.  0.50 of the HBM "
                                     "roofline would take 4-clock issue of every instruction at more than the sustained clock (DESIGN.md 3.1)"}
        elif power:
            r["floor"] = {"package_power": power, "note": "counters not collected in this run"}
        # the figures a reader of the LAST kilobytes of this line needs (the driver keeps a 10 KB tail): fractions of the HBM roofline
        ex = out.get("extras", {})
        frac = lambda o: round(o["roofline"]["frac"], 4) if isinstance(o, dict) and "roofline" in o else None
        ratio = lambda o: (round(o["roofline"]["traffic_source"]["ratio_to_algorithmic"], 3)
                           if isinstance(o, dict) and isinstance(o.get("roofline", {}).get("traffic_source"), dict) and "ratio_to_algorithmic" in o["roofline"]["traffic_source"] else None)
        summary = {"ntt_fwd_R15": {"limb_ntt_per_s": round(out["value"]), "frac": round(r["frac"], 4), "traffic_ratio": ratio(out), "bit_exact": out["bit_exact"]}}
        for key, name in (("intt", "ntt_inv_R15"), ("ntt_ckks_moduli", "ntt_fwd_ckks_moduli"), ("intt_ckks_moduli", "ntt_inv_ckks_moduli"), ("modup_split_qp", "modup_R15")):
            if key in ex:
                summary[name] = {"frac": frac(ex[key]), "traffic_ratio": ratio(ex[key]), "bit_exact": ex[key].get("bit_exact")}
        for key in ("mulcoeffs_montgomery", "div_round_by_last_modulus_ntt"):
            if key in ex:
                summary[key + "_R15"] = {"frac": round(ex[key]["frac_hbm"], 4), "bit_exact": ex[key].get("bit_exact")}
        if isinstance(out.get("config5_single_process"), dict) and "value" in out["config5_single_process"]:
            c5s = out["config5_single_process"]
            summary["config5_single_process"] = {"per_s": round(c5s["value"], 1), "n_gpus": c5s["n_gpus"], "placement_ok": c5s["placement_ok"],
                                                 "frac": round(c5s["roofline"]["frac"], 4)}
        for key in ("ckks_mulrelin", "bfv_mul", "config5"):
            if isinstance(out.get(key), dict) and "roofline" in out[key]:
                summary[key] = {"per_s": round(out[key]["value"], 1), "frac": frac(out[key]), "traffic_ratio": ratio(out[key]), "bit_exact": out[key].get("bit_exact")}
        for key, o in out.get("pipelines", {}).items():
            summary[key] = {"per_s": round(o["value"], 1), "frac": frac(o), "traffic_ratio": ratio(o), "bit_exact": o.get("bit_exact"),
                            "cpu_per_s": round(o["cpu_baseline"]["value"], 1) if "cpu_baseline" in o else None}
        et, en = out.get("evaluator_threads"), out.get("evaluator_threads_native")
        if isinstance(et, dict) and "rows" in et:
            th = {"python_threads_direct": {str(x["threads"]): round(x["mulrelin_per_s"]) for x in et["rows"]},
                  "python_threads_batcher": {str(x["threads"]): round(x["mulrelin_per_s"]) for x in et["batcher_rows"]}}
            if isinstance(en, dict) and "rows" in en:
                for how in ("direct", "batcher"):
                    th["cpp_threads_" + how] = {str(x["threads"]): round(x["calls_per_s"]) for x in en["rows"] if x["how"] == how}
            summary["mulrelin_batch1_per_s_by_threads"] = th
        summary["runtime"] = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "note": "set by bench.py before HIP initialises (--hw-queues); only the "
                              "evaluator_threads rows have more than one busy stream"}
        r["companions"] = {k: v for k, v in summary.items() if k not in ("runtime", "mulrelin_batch1_per_s_by_threads")}      # (the driver's parsed record keeps the roofline object whole)
        out["summary"] = summary
        emit(out)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
