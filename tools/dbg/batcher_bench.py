"""Builds and runs tools/batcher_bench.cpp (the MulRelin batcher seen from a C++ host, no interpreter in the loop).
usage: batcher_bench.py --build | batcher_bench.py [PN15QP880] [T,T,...] [iters] [lanes] [max_batch]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "tools", "build", "batcher_bench")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib = os.path.join(ROOT, "lattigo-fhe-by-go_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "batcher_bench.cpp"), "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", "-L" + lib, "-L/opt/rocm/lib", "-llattigo_ring_hip", "-lamdhip64",
                           "-Wl,-rpath," + lib + ":/opt/rocm/lib", "-o", EXE])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        build()
        return
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("lr_params", os.path.join(ROOT, "lattigo-fhe-by-go_amd", "params.py"))
    params = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(params)
    name = sys.argv[1] if len(sys.argv) > 1 else "PN15QP880"
    Ts = (sys.argv[2] if len(sys.argv) > 2 else "1,4,16,64").split(",")
    iters = sys.argv[3] if len(sys.argv) > 3 else "100"
    lanes = sys.argv[4] if len(sys.argv) > 4 else "2"
    max_batch = sys.argv[5] if len(sys.argv) > 5 else "64"
    N, Q, P = params.ckks_moduli(name)
    path = "/tmp/batcher_bench_%s.txt" % name
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (N.bit_length() - 1, len(Q), len(P)))
        f.write(" ".join(str(q) for q in list(Q) + list(P)) + "\n")
    if not os.path.exists(EXE):
        build()
    for T in Ts:
        subprocess.check_call([EXE, path, T, iters, lanes, max_batch])


if __name__ == "__main__":
    main()
