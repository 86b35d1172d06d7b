"""PCIe-inclusive rate of the host-pointer entry point: lr_ntt_host on R15, one poly per call, per-limb host slices
in and out as the Go boundary hands them over."""
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
N, Q = params.DefaultParamsQi(15)
ctx = ring.NewContextWithParams(N, Q)
x = sampling.uniform_poly(Q, N, 1, seed=1)[0]
limbs = [np.ascontiguousarray(x[i]) for i in range(len(Q))]
for _ in range(5):
    ctx.NTTHost(limbs)
t0 = time.perf_counter()
K = 200
for _ in range(K):
    out = ctx.NTTHost(limbs)
dt = (time.perf_counter() - t0) / K
print("lr_ntt_host R15: %.1f us per poly-NTT = %.0f poly-NTT/s = %.0f limb-NTT/s (%.2f GB/s each way)" % (dt * 1e6, 1 / dt, len(Q) / dt, 8 * N * len(Q) / dt / 1e9))
