"""Host-side logic that needs no device: parameter generation, sampling, sharding."""
import numpy as np

import bench


def test_sampling_is_reproducible_and_in_range(pkg):
    mod = list(pkg.params.Qi60()[-2:]) + [1099512938497]
    a = pkg.sampling.uniform_poly(mod, 512, batch=3, seed=7)
    b = pkg.sampling.uniform_poly(mod, 512, batch=3, seed=7)
    assert np.array_equal(a, b)
    for i, q in enumerate(mod):
        assert int(a[:, i].max()) < q
    assert not np.array_equal(a[0], a[1])
    # uses the top bits too: mean close to q/2
    assert abs(float(a[:, 0].astype(np.float64).mean()) / mod[0] - 0.5) < 0.05


def test_ring_presets(pkg):
    N, Q = pkg.params.DefaultParamsQi(15)
    assert N == 1 << 15 and len(Q) == 16 and Q[-1] == 1152921504050839553
    N, P = pkg.params.DefaultParamsPi(14)
    assert N == 1 << 14 and len(P) == 8


def test_shard_units_partitions_the_batch():
    for total in (1, 7, 256, 1024):
        for world in (1, 2, 3, 8):
            parts = [bench.shard_units(total, r, world) for r in range(world)]
            assert sum(n for _, n in parts) == total
            pos = 0
            for start, n in parts:
                assert start == pos
                pos += n
            assert max(n for _, n in parts) - min(n for _, n in parts) <= 1


def test_algorithmic_bytes():
    # limb-NTT = 16*N bytes; R15 poly-NTT = 8 MiB
    assert bench.ntt_bytes(1 << 15, 16) == 8 << 20


def test_precompute_under_sanitizers(tmp_path):
    """the host-side table generation (constants, primitive roots with the reference's factorisation quirks, psi tables,
    basis-extension tables) compiled with AddressSanitizer + UndefinedBehaviorSanitizer and run on the CPU"""
    import os
    import subprocess

    from conftest import ROOT
    src = os.path.join(ROOT, "tests", "cpp", "host_precompute_sanitize.cpp")
    csrc = os.path.join(ROOT, "lattigo-fhe-by-go_amd", "csrc")
    exe = str(tmp_path / "host_san")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + csrc,
                           "-I" + os.path.join(ROOT, "include"), src, os.path.join(csrc, "lr_precompute.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


def test_kernel_argument_block_size_is_one_number():
    """The assembly code objects declare the size of their kernel-argument block (.amdhsa_kernarg_size and the metadata's
    .kernarg_segment_size); the host passes sizeof(NttLaunch).  A launch whose block is smaller than the declared size makes
    the runtime hand the GPU a short buffer: the kernel reads past it and the process dies with a GPU memory fault / abort
    (reproduced in round 2 with tools/asm_ubench's runner, and the likely cause of round 1's abort while NttLaunch was being
    grown for the dual kernels, DESIGN.md "Incidents").  Pin the three numbers to each other."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hpp = open(os.path.join(root, "lattigo-fhe-by-go_amd", "csrc", "lr_device.hpp")).read()
    host = int(re.search(r"static_assert\(sizeof\(NttLaunch\) == (\d+)", hpp).group(1))
    gen = open(os.path.join(root, "lattigo-fhe-by-go_amd", "csrc", "asmgen", "gen_ntt.py")).read()
    declared = set(int(x) for x in re.findall(r"\.amdhsa_kernarg_size (\d+)", gen)) | set(int(x) for x in re.findall(r"\.kernarg_segment_size: (\d+)", gen))
    assert declared == {host}, (declared, host)
    runner = open(os.path.join(root, "tools", "asm_ubench", "run.cpp")).read()
    assert "sizeof(args) == %d" % host in runner
