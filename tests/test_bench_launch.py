"""bench.py --gpus N without a launcher around it: the parent must start torch.distributed.run as a CHILD process (before it imports
torch or touches a GPU), relay rank 0's JSON line and exit with the child's code.  No GPU here: --dry-launch shows the command,
--launch-check lets the ranks rendezvous over gloo and run the timing harness's collectives and the legs' error-flag agreement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_dry_launch_prints_the_child_command():
    res = run(["--gpus", "8", "--steps", "7", "--warmup", "3", "--dry-launch"])
    assert res.returncode == 0, res.stderr[-2000:]
    cmd = json.loads(res.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7", "--warmup", "3"]        # the ranks get the same arguments, minus --dry-launch


def test_parent_does_not_import_torch_before_launching():
    # the launcher path must run before anything initialises a GPU: torch is imported only below it
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    assert main.index("raise SystemExit(self_launch(") < main.index("    import torch\n")
    code = "import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--dry-launch']\ntry:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit:\n    pass\nprint('torch' in sys.modules)" % BENCH
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert res.returncode == 0, res.stderr[-2000:]
    assert res.stdout.strip().splitlines()[-1] == "False"


def test_self_launch_relays_one_line_and_the_exit_code():
    res = run(["--gpus", "2", "--launch-check"])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["launch_check"] is True and d["world"] == 2 and d["max_rank"] == 1 and d["a_rank_failed"] is False
    assert d["leg_a"] == {"ran_on": 0, "max": 1.0} and d["leg_b"] == {"max": 8.0}
    assert d["master"].startswith("127.0.0.1:")
    # a set-up failure on rank 1 only: the agreement all-reduce makes EVERY rank skip that leg (rank 0 reports it although its own
    # set-up succeeded), nobody is left waiting in the leg's collectives, and the next leg runs on both ranks
    res = run(["--gpus", "2", "--launch-check"], env={"LR_BENCH_CHECK_FAIL_RANK": "1"})
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert d["a_rank_failed"] is True and d["leg_a"] == {"error": "set-up failed on another rank"} and d["leg_b"] == {"max": 8.0}
    res = run(["--gpus", "2", "--launch-check"], env={"LR_BENCH_CHECK_FAIL_RANK": "0"})
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert "injected on rank 0" in d["leg_a"]["error"] and d["leg_b"] == {"max": 8.0}


def test_under_a_launcher_bench_runs_as_a_rank():
    # WORLD_SIZE set: no second launcher; with --launch-check and world size 1 the rank path is exercised directly
    res = run(["--gpus", "1", "--launch-check"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29641"})
    assert res.returncode == 0, res.stderr[-2000:]
    assert json.loads(res.stdout.strip().splitlines()[-1])["world"] == 1
