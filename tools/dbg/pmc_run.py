"""One launch pattern for counter collection: a few launches of one ring operation on the bench shape (1 GiB per operand).

    pmc_run.py <logn> [qi60 | ckks | <bits>] [ntt | intt | modup | mulcoeffs | rescale]

qi60 = ring.DefaultParamsQi[logn] (default); ckks = the first limbs of DefaultParams[PN15QP880] at that degree (dual kernels: FP64 body);
<bits> = generated NTT primes of that size.  modup = ModUpSplitQP onto DefaultParamsPi[logn] (ext_wide_kernel / ext_sum_kernel).
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
ring, params, sampling = pkg.ring, pkg.params, pkg.sampling
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 15
mset = sys.argv[2] if len(sys.argv) > 2 else "qi60"
op = sys.argv[3] if len(sys.argv) > 3 else "ntt"
N, moduli = params.DefaultParamsQi(logn)
L = len(moduli)
if mset == "ckks":
    moduli = list(params.ckks_moduli("PN15QP880")[1][:L]) if logn == 15 else params.GenerateNTTPrimes(40, logn, L)
elif mset != "qi60":                       # pmc_run.py 15 40: moduli of that many bits (below 2^46: the FP64 body)
    moduli = params.GenerateNTTPrimes(int(mset), logn, L)
B = (1 << 30) // (8 * N * L)
ctx = ring.NewContextWithParams(N, moduli)
base = sampling.uniform_poly(moduli, N, 2, seed=1)
src, dst = ctx.NewPoly(B).set(np.concatenate([base] * (B // 2))), ctx.NewPoly(B)
if op == "ntt":
    fn = lambda: ctx.NTT(src, dst)
elif op == "intt":
    fn = lambda: ctx.InvNTT(src, dst)
elif op == "mulcoeffs":
    fn = lambda: ctx.MulCoeffsMontgomery(src, src, dst)
elif op == "rescale":
    def fn():                                   # in place; the limb count goes back up for the next call
        pkg._native.check(pkg._native.lib().lr_poly_set_limbs(src.h, L))
        ctx.DivRoundByLastModulusNTT(src)
elif op == "modup":
    _, pmod = params.DefaultParamsPi(logn)
    ctxP = ring.NewContextWithParams(N, pmod)
    be = ring.NewFastBasisExtender(ctx, ctxP)
    outP = ctxP.NewPoly(B)
    fn = lambda: be.ModUpSplitQP(L - 1, src, outP)
else:
    raise SystemExit("unknown op " + op)
for _ in range(5):
    fn()
ctx.Sync()
print("LAUNCHES 5 polys %d limbs %d N %d" % (B, L, N))
