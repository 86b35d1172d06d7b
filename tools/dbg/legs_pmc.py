"""Launch pattern for counter collection over the pipeline legs of bench.py (tools/bench_legs.py): every leg runs twice (a warm-up call, then
the measured call), each call between marker kernels (div_selftest_kernel, which no pipeline launches): marker, call, marker, call, marker
per leg, so that a collector can cut the dispatch list of `rocprofv3 --pmc` into legs by order (leg i's measured call is segment 3i + 2):

    python tools/dbg/legs_pmc.py [leg ...]

Prints `LEG <name> <units per call>` before each leg and `END` after the last one."""
import importlib.util
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

pkg = g.load_package()
spec = importlib.util.spec_from_file_location("bench_legs", os.path.join(ROOT, "tools", "bench_legs.py"))
bench_legs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench_legs)
only = set(sys.argv[1:]) or None
kits, groups = bench_legs.build_legs(pkg, None, device=0, only=only)
mark_ctx = pkg.ring.NewContextWithParams(16, [pkg.params.Qi60()[-1]])
for gname, make in groups:
    legs = make()
    for leg in legs:
        leg.run()                                  # pools and scratch exist before the marked calls
        leg.sync.Sync()
        print("LEG %s %d" % (leg.name, leg.units_per_call), flush=True)
        for _ in range(2):
            mark_ctx.selftest_division(256, 1)     # marker (synchronises)
            leg.run()
            leg.sync.Sync()
        mark_ctx.selftest_division(256, 1)         # closes the measured call: what follows (the next leg's set-up) is nobody's
        leg.after()
    del legs
    kits.drop(gname)
print("END", flush=True)
