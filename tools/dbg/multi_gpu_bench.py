"""Builds and runs tools/multi_gpu_bench.cpp: BASELINE config 5 from one process, one host thread per device, through the C ABI only
(lr_poly_copy_peer / lr_context_wait_peer_copies for the gather).
usage: multi_gpu_bench.py --build | multi_gpu_bench.py [--gpus G] [--units U] [--chunk C] [--steps K] [--warmup W] [--set NAME] [--logn n]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "tools", "build", "multi_gpu_bench")
SRC = os.path.join(ROOT, "tools", "multi_gpu_bench.cpp")


def build(force=False):
    lib = os.path.join(ROOT, "lattigo-fhe-by-go_amd")
    if not force and os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "lattigo_ring.h"))):
        return EXE
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", SRC, "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-llattigo_ring_hip",
                           "-Wl,-rpath," + lib, "-o", EXE])
    return EXE


def run(args, timeout=900):
    exe = build()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        print(build(force=True))
    else:
        res = run(sys.argv[1:])
        sys.stderr.write(res.stderr)
        sys.stdout.write(res.stdout)
        raise SystemExit(res.returncode)
