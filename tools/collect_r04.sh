#!/bin/bash
# Collects round 4's rocprofv3 evidence on the GPU box (through gpurun); output: gpurun_out/prof_r04/* (copy what is to be judged into profiles/r04/).
#   1. the default bench line (its own --pmc child passes fill every roofline object: headline, extras, MulRelin, the pipeline legs)
#   2. kernel-trace statistics of the same command (--no-traffic: no profiler nested in the profiler) -> kernel_stats.csv
#   3. kernel-trace statistics of the pipeline legs with the lowest fractions, one leg per trace -> legs_<name>_kernel_stats.csv
#   4. FETCH_SIZE / WRITE_SIZE of the headline launch and of one MulRelin product in separate --pmc passes -> pmc_hbm.json, mulrelin_pmc_hbm.json
set -e
OUT=/root/repo/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 /root/repo/bench.py > $OUT/bench.json 2> $OUT/bench.stderr.txt
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 /root/repo/bench.py --no-cpu-baseline --no-traffic --no-threads --steps 50 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $OUT/trace/*kernel_stats.csv $OUT/kernel_stats.csv
echo "bench trace done"
for leg in bfv_mul moddown_ntt bfv_rotate_columns ckks_encrypt_pk ckks_rotate_hoisted; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$leg -o leg -- python3 /root/repo/tools/dbg/legs_pmc.py $leg > $OUT/legs_$leg.log 2>&1
  cp $OUT/trace_$leg/*kernel_stats.csv $OUT/legs_${leg}_kernel_stats.csv
done
echo "leg traces done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 /root/repo/tools/dbg/pmc_run.py 15 > $OUT/pmc_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/mr_pmc_$c -o p -- python3 /root/repo/tools/dbg/mulrelin_pmc.py PN15QP880 64 4 > $OUT/mr_pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mr -o mulrelin -- python3 /root/repo/tools/dbg/mulrelin_pmc.py PN15QP880 64 8 > $OUT/mulrelin.log 2>&1
cp $OUT/trace_mr/*kernel_stats.csv $OUT/mulrelin_kernel_stats.csv
echo "pmc done"
python3 - <<'PY'
import csv, collections, json, glob, os
out = "/root/repo/gpurun_out/prof_r04"
def med(dirname, counter, keep):
    v = []
    for f in glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True):
        v += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and keep(r["Kernel_Name"])]
    v.sort()
    return v[len(v) // 2]
f, w = med("pmc_FETCH_SIZE", "FETCH_SIZE", lambda n: "ntt_fwd15" in n), med("pmc_WRITE_SIZE", "WRITE_SIZE", lambda n: "ntt_fwd15" in n)
alg = 16 * 32768 * 16 * 256
json.dump({"FETCH_SIZE_KB_per_launch_median": f, "WRITE_SIZE_KB_per_launch_median": w, "kernel": "lr_ntt_fwd15_m1, 256 polys x 16 limbs per launch (tools/dbg/pmc_run.py 15)",
           "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024, "ratio_to_algorithmic": (2 * f * 1024 + w * 1024) / alg,
           "note": "rocprofv3 --pmc, one counter per pass (tools/collect_r04.sh); FETCH_SIZE doubled per the gfx950 correction of the microarch guide"},
          open(os.path.join(out, "pmc_hbm.json"), "w"), indent=1)
tot, per_kernel = {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    log = open(os.path.join(out, "mr_pmc_%s.log" % c)).read()
    products = int([l for l in log.splitlines() if l.startswith("PRODUCTS")][-1].split()[1])
    acc = collections.defaultdict(float)
    for fn in glob.glob(os.path.join(out, "mr_pmc_%s" % c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if "rocclr" not in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]] += float(r["Counter_Value"])
    tot[c] = sum(acc.values()) / products
    for k, v in acc.items():
        per_kernel.setdefault(k, {})[c] = v / products
nq, np_, N = 18, 3, 32768
beta = -(-nq // np_)
alg = 8 * N * (4 * nq + beta * 2 * (nq + np_) + 2 * nq)
hbm = 2 * tot["FETCH_SIZE"] * 1024 + tot["WRITE_SIZE"] * 1024
json.dump({"params": "PN15QP880", "algorithmic_bytes_per_product": alg, "hbm_bytes_per_product": hbm, "ratio_to_algorithmic": hbm / alg,
           "FETCH_SIZE_KB_per_product": tot["FETCH_SIZE"], "WRITE_SIZE_KB_per_product": tot["WRITE_SIZE"], "per_kernel_KB_per_product": per_kernel,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/dbg/mulrelin_pmc.py, every kernel summed and divided by the products executed; reads doubled per the gfx950 correction"},
          open(os.path.join(out, "mulrelin_pmc_hbm.json"), "w"), indent=1)
print("summaries written")
PY
head -12 $OUT/kernel_stats.csv
