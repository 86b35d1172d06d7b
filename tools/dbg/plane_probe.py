"""Runs tools/build/plane_probe (tools/plane_probe.hip) and samples rocm-smi beside it: milliseconds, GB/s and package power per variant.
    python tools/dbg/plane_probe.py            (builds the probe with hipcc if it is missing)"""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "tools", "build", "plane_probe")
if not os.path.exists(EXE):
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", EXE, os.path.join(ROOT, "tools", "plane_probe.hip")])
samples, stop = [], [False]


def poll():
    while not stop[0]:
        try:
            txt = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            card = next(iter(json.loads(txt).values()))
            w = [float(v) for k, v in card.items() if "ower" in k and "(W)" in k and re.match(r"^[0-9.]+$", str(v))]
            mhz = [int(m.group(1)) for k, v in card.items() if k.startswith("sclk") for m in [re.search(r"\((\d+)Mhz\)", str(v))] if m]
            if w:
                samples.append((time.time(), max(w), mhz[0] if mhz else 0))
        except Exception:
            pass
        time.sleep(0.05)


th = threading.Thread(target=poll)
th.start()
proc = subprocess.Popen([EXE, "10", "2.5"], stdout=subprocess.PIPE, text=True)
marks, cur = [], None
for line in proc.stdout:
    line = line.rstrip("\n")
    now = time.time()
    if line.startswith("BEGIN "):
        cur = [line[6:], now, None, None]
    elif line.startswith("END ") and cur:
        cur[2] = now
        marks.append(cur)
        cur = None
    elif cur is not None:
        cur[3] = line
proc.wait()
stop[0] = True
th.join()
for name, t0, t1, result in marks:
    mid = [(w, m) for t, w, m in samples if t0 + 0.8 <= t <= t1]
    if mid:
        print("%s   | package %.0f W (max of %d samples), sclk %d MHz" % (result, max(w for w, _ in mid), len(mid), sorted(m for _, m in mid)[len(mid) // 2]))
    else:
        print("%s   | no power sample" % result)
sys.exit(proc.returncode)
