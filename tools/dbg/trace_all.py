"""The last `n` kernels of a rocprofv3 --kernel-trace csv: start relative to the first of them, duration, gap, grid, name.   trace_all.py <dir> [n]"""
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -12:]
t0, last = int(rows[0]["Start_Timestamp"]), int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  dur %6.1f  after-last-end %6.1f  grid %sx%sx%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Kernel_Name"][:60]))
    last = max(last, e)
