#!/bin/bash
# rocprofv3 evidence for the FP64-bodied kernels and the MulRelin pipeline (run on the GPU box through gpurun):
#   kernel-trace statistics of one CKKS MulRelin batch loop, PMC passes (separate runs, no trace domains) of the
#   forward NTT on 40-bit moduli (lr_ntt_fwd15_m3).  Output: gpurun_out/prof_fp/*
set -e
OUT=/root/repo/gpurun_out/prof_fp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o mulrelin -- python3 /root/repo/tools/dbg/ckksbench.py PN15QP880 64 > $OUT/mulrelin.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 /root/repo/tools/dbg/pmc_run.py 15 40 > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- python3 /root/repo/tools/dbg/pmc_run.py 15 40 > $OUT/pmc_sq.log 2>&1
python3 - <<'PY'
import csv, collections, json, glob, os
out = "/root/repo/gpurun_out/prof_fp"
res = {}
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq"):
    for f in glob.glob(os.path.join(out, d, "*counter_collection.csv")):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "ntt_fwd15_m3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v.sort()
            res[k] = v[len(v) // 2]
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res))
PY
cp $OUT/trace/*kernel_stats.csv $OUT/mulrelin_kernel_stats.csv
grep -v "^[EWI]2026" $OUT/mulrelin.log
