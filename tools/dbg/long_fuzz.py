"""One-off extended fuzz on the GPU box: the bodies of tests/test_gpu_fuzz.py with seeds far beyond the committed ranges.
    python tools/dbg/long_fuzz.py [first_seed] [count]"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

pkg, oracle = g.load_package(), g.load_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fails = 0
t0 = time.time()
for name in ("test_ntt_and_elementwise_fuzz", "test_basis_extension_and_rescale_fuzz", "test_key_switch_fuzz", "test_dual_kernel_fuzz", "test_mulrelin_rescale_fuzz",
             "test_rotation_encrypt_decrypt_fuzz", "test_moddown_divfloor_permute_fuzz", "test_bfv_pipelines_fuzz"):
    fn = getattr(F, name)
    done = 0
    for seed in range(first, first + count):
        try:
            fn(pkg, oracle, seed)
            done += 1
        except Exception:      # noqa: BLE001
            fails += 1
            print("FAIL", name, seed)
            traceback.print_exc(limit=3)
    print("%s: %d seeds ok (%.0f s)" % (name, done, time.time() - t0), flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
