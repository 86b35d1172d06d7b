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
    assert r["bound"] in ("hbm", "valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0
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
    # the other roof, counted in the same run: vector instructions per wave (the generator's static count), waves of the launch, issue
    # fraction at the nominal clock and at the clock the profiled pass sustained
    v = r["valu"]
    for key in ("instr_per_wave", "waves", "issue_frac_nominal", "issue_frac_sustained", "sclk_MHz"):
        assert v[key] is not None and v[key] > 0, key
    assert v["waves"] == units * 16 and 4300 < v["instr_per_wave"] < 4450
    assert 0.4 < v["issue_frac_nominal"] < 1.0 and v["issue_frac_nominal"] <= v["issue_frac_sustained"] * 1.02 and 1000 < v["sclk_MHz"] <= 2500
    assert r["bound"] == ("valu" if v["issue_frac_sustained"] > r["frac"] else "hbm")


def test_every_benchmarked_pipeline_has_a_roofline_and_a_cpu_baseline():
    """every pipeline entry point the reference benchmarks carries `roofline` (in-run traffic, bound, vector issue, kernel
    split) and `cpu_baseline` in the line, bfv_mul (BASELINE config 4) included; the short `summary` object is the LAST key (it survives a
    tail of the line) and the headline roofline carries `floor` and `companions`"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-config5", "--no-rings", "--no-extras", "--no-threads"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_legs", os.path.join(ROOT, "tools", "bench_legs.py"))
    bl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bl)
    legs = dict(d["pipelines"], bfv_mul=d["bfv_mul"], ckks_mulrelin=d["ckks_mulrelin"])
    assert set(bl.LEG_NAMES) <= set(legs), sorted(set(bl.LEG_NAMES) - set(legs))
    for name, o in legs.items():
        assert o["bit_exact"] is True, (name, o.get("check_error"))
        r, c = o["roofline"], o["cpu_baseline"]
        assert r["bound"] in ("hbm", "valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1.0, name
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12, name
        assert r["traffic"] is not None and r["traffic_source"]["ratio_to_algorithmic"] > 0.5, (name, r.get("traffic_source"))
        assert r["valu"]["issue_frac_sustained"] > 0, name
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == o["unit"], name
        if name != "ckks_mulrelin":
            assert r["kernels"] and "reference_benchmark" in o, name
            assert abs(r["achieved"] - r["algorithmic_bytes_per_unit"] * r["units_per_call"] / (r["pipeline_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9, name
    assert list(d)[-1] == "summary"
    s = d["summary"]
    for name in list(bl.LEG_NAMES) + ["ckks_mulrelin", "ntt_fwd_R15"]:
        assert s[name]["frac"] is not None and s[name]["bit_exact"] is True, name
    assert len(json.dumps(s)) < 6000                       # it has to fit the tail of the line
    f = d["roofline"]["floor"]
    assert 4300 < f["valu_instructions_per_wave"] < 4450 and 3.9 < f["clocks_per_instruction_in_run"] < 8 and 1000 < f["sclk_MHz_sustained"] <= 2500
    assert 0.8 < f["model_ms"] / f["measured_ms"] < 1.25                    # instructions x clocks per instruction / clock IS the launch time
    assert f["frac"]["if_every_instruction_issued_in_4_clocks_at_the_peak_clock"] > f["frac"]["measured"] and f["package_power"]["package_W"] > 500
    assert d["roofline"]["companions"]["ckks_mulrelin"]["frac"] == s["ckks_mulrelin"]["frac"]
    assert "traffic_profile" not in d["roofline"] or "r04" in d["roofline"]["traffic_profile"]


def test_secondary_legs_carry_their_roofs_and_the_rescale_leg():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-config5", "--no-rings", "--no-ckks", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    ex = d["extras"]
    for key in ("intt", "modup_split_qp", "ntt_ckks_moduli", "intt_ckks_moduli"):
        r = ex[key]["roofline"]
        assert ex[key]["bit_exact"] is True
        assert r["traffic"] is not None and r["valu"]["issue_frac_sustained"] > 0 and r["bound"] in ("hbm", "valu"), (key, r.get("traffic_source"))
        assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    assert ex["mulcoeffs_montgomery"]["bit_exact"] is True
    rs = ex["div_round_by_last_modulus_ntt"]
    assert rs["bit_exact"] is True and rs["poly_per_s"] > 0 and rs["algorithmic_bytes_per_poly"] == 8 * d["config"]["N"] * (2 * d["config"]["limbs"] - 1)


def test_config5_leg_under_rccl_is_ordered():
    """The config-5 leg on a real RCCL communicator (world size 1: LR_BENCH_FORCE_DIST=1): products and the chunked gather on one
    explicit stream; the leg poisons its outputs and the root buffer before the step it checks, so a gather that ran ahead of the
    kernels would be caught."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "ckks16", "--steps", "2", "--warmup", "1", "--config5-units", "8", "--config5-chunk", "4",
           "--no-cpu-baseline", "--no-traffic"]
    env = dict(os.environ, LR_BENCH_FORCE_DIST="1", MASTER_PORT="29577")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        env.pop(k, None)
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    c5 = d["config5"]
    assert c5["bit_exact"] is True and c5["checked_units"] == [0, 7] and "poisoned" in c5["checked_after"]
    assert "RCCL" in c5["gather"] and "chunks of 4" in c5["gather"] and "explicit torch stream" in c5["stream"]
