"""Kernel timeline of one MulRelin out of a rocprofv3 --kernel-trace csv (the last complete product of the run): start, duration, gap to the
previous kernel's end, grid.  usage: trace_b1.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "tensor_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
last_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  dur %6.1f  after-last-end %6.1f  grid %sx%sx%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3, r.get("Grid_Size_X", ""),
                                                                         r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r["Kernel_Name"][:48]))
    last_end = max(last_end, e)
print("product to product: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
