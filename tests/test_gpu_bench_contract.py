"""bench.py's one-line contract, on the device: exactly one JSON line on stdout with the keys the driver reads, the roofline object of
the dominant kernel (measured in the run: kernel name and duration from the library, fraction = algorithmic bytes / duration / peak)
and the cpu_baseline object."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_json_line_with_roofline_and_cpu_baseline():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-config5", "--no-rings", "--no-extras", "--no-ckks"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "bit_exact"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "u64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["bit_exact"] is True
    # value = limb-NTTs of the batch per second over the timed region
    units = d["config"]["polys_per_gpu"] * d["config"]["limbs"]
    assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["kernel"] == "lr_ntt_fwd15_m1"
    # HBM bytes of one launch, counted in this run by rocprofv3 --pmc child passes (null only if the profiler could not run)
    assert r["traffic"] is not None, r.get("traffic_source")
    assert 0.98 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10 and "FETCH_SIZE_KB" in r["traffic_source"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert r["algorithmic_bytes_per_launch"] == 16 * d["config"]["N"] * units
    assert 0.2 < r["frac"] < 1.0 and r["kernel_ms"] <= d["ms_per_step"] * 1.05
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
