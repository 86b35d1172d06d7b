#!/bin/bash
# Round-3 counter evidence (run on the GPU box through gpurun): SQ / LDS / HBM counters, each set in its own rocprofv3 --pmc pass
# (no trace domains), for the kernels round 2 left without any: the FP64-bodied transforms (lr_ntt_fwd15_m3, lr_ntt_inv15_m3, and
# lr_ntt_fwd15_m4 inside MulRelin), the 60-bit inverse (lr_ntt_inv15_m1), the basis-extension kernels (ext_wide_kernel on R15 / R16,
# ext_sum_kernel inside MulRelin) and the rounding rescale.  Output: gpurun_out/prof_r03/<run>.json = {kernel: {counter: median}}.
set -e
OUT=/root/repo/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run() {  # tag script args...
  local tag=$1; shift
  rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/${tag}_sq1 -o p -- python3 "$@" > $OUT/${tag}_sq1.log 2>&1
  rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/${tag}_sq2 -o p -- python3 "$@" > $OUT/${tag}_sq2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_fetch -o p -- python3 "$@" > $OUT/${tag}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_write -o p -- python3 "$@" > $OUT/${tag}_write.log 2>&1
  python3 - $OUT $tag <<'PY'
import collections, csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "sq2", "fetch", "write"):
    for f in glob.glob(os.path.join(out, "%s_%s" % (tag, d), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "rocclr" in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    row = {"dispatches": max(len(v) for v in cs.values())}
    for c, v in cs.items():
        v.sort()
        row[c] = v[len(v) // 2]
    if "SQ_INSTS_VALU" in row and row.get("SQ_WAVES"):
        row["valu_instructions_per_wave"] = row["SQ_INSTS_VALU"] / row["SQ_WAVES"]
    if "SQ_ACTIVE_INST_VALU" in row and row.get("SQ_BUSY_CYCLES"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles over all SIMDs; SQ_BUSY_CYCLES is summed over the SQs (one per SE x XCD)
        row["valu_busy_note"] = "see profiles/r03/README.md for the normalisation"
    if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
        row["hbm_bytes_per_launch"] = 2 * row["FETCH_SIZE"] * 1024 + row["WRITE_SIZE"] * 1024
    res[k] = row
json.dump(res, open(os.path.join(out, tag + ".json"), "w"), indent=1)
print(tag, json.dumps({k: {c: v for c, v in row.items() if c in ("valu_instructions_per_wave", "SQ_WAVES", "GRBM_GUI_ACTIVE", "hbm_bytes_per_launch")} for k, row in res.items()}))
PY
  rm -rf $OUT/${tag}_sq1 $OUT/${tag}_sq2 $OUT/${tag}_fetch $OUT/${tag}_write
}
T=/root/repo/tools/dbg
run ntt15_ckks_fwd $T/pmc_run.py 15 ckks ntt
run ntt15_ckks_inv $T/pmc_run.py 15 ckks intt
run ntt15_qi60_inv $T/pmc_run.py 15 qi60 intt
run modup15 $T/pmc_run.py 15 qi60 modup
run modup16 $T/pmc_run.py 16 qi60 modup
run rescale15 $T/pmc_run.py 15 qi60 rescale
run mulrelin15 $T/mulrelin_pmc.py PN15QP880 64 2
echo "r03 pmc done"
