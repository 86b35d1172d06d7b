#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box through gpurun):
#   kernel-trace statistics of the bench command, and PMC passes (separate runs, no trace domains) for HBM bytes
#   and VALU utilisation of the forward NTT launch.  Output: gpurun_out/prof_final/*
set -e
OUT=/root/repo/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 /root/repo/tools/dbg/pmc_run.py 15 > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- python3 /root/repo/tools/dbg/pmc_run.py 15 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -o p -- python3 /root/repo/tools/dbg/pmc_run.py 15 > $OUT/pmc_sq2.log 2>&1
python3 - <<'PY'
import csv, collections, json, glob, os
out = "/root/repo/gpurun_out/prof_final"
res = {}
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq", "pmc_sq2"):
    for f in glob.glob(os.path.join(out, d, "*counter_collection.csv")):
        acc, n = collections.defaultdict(list), 0
        for r in csv.DictReader(open(f)):
            if "ntt_fwd15" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v.sort()
            res[k] = v[len(v) // 2]
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res))
PY
cp $OUT/trace/*kernel_stats.csv $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
tail -1 $OUT/bench_under_rocprof.log | cut -c1-400
