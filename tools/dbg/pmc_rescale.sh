set -e
OUT=/root/repo/gpurun_out/pmc_rs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/${1}_$c -o p -- python3 /root/repo/tools/dbg/pmc_run.py 15 qi60 rescale > $OUT/${1}_$c.log 2>&1
done
python3 - $OUT $1 <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, tag, c), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "m5" in k:
        f = sorted(cs["FETCH_SIZE"])[len(cs["FETCH_SIZE"]) // 2]; w = sorted(cs["WRITE_SIZE"])[len(cs["WRITE_SIZE"]) // 2]
        print(tag, k, "FETCH KB", f, "WRITE KB", w, "2F+W GB", (2 * f + w) * 1024 / 1e9)
PY
