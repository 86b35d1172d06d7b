// lr_arith.hpp -- 64-bit modular primitives shared by the host precompute and the gfx950 kernels.
//
// Semantics follow Lattigo v1.3.1 ring/modular_reduction.go (cited per function); the
// results of the canonical forms are mathematically unique, the *_constant / lazy forms
// reproduce the reference's exact intermediate (they are observable through the
// "...Constant" / "...NoMod" ring.Context methods).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LR_HD __host__ __device__ __forceinline__
#define LR_D __device__ __forceinline__
#else
#define LR_HD inline
#endif

namespace lr {

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

LR_HD u64 mul_hi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((u128)a * b) >> 64);
#endif
}

LR_HD void mul_wide64(u64 a, u64 b, u64 &hi, u64 &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    u128 p = (u128)a * b;
    hi = (u64)(p >> 64);
    lo = (u64)p;
#endif
}

// CRed, modular_reduction.go:211
LR_HD u64 cred(u64 a, u64 q) { return a >= q ? a - q : a; }

// MRedConstant, modular_reduction.go:83 -- result in [0, 2q)
LR_HD u64 mred_constant(u64 x, u64 y, u64 q, u64 qinv) {
    u64 ahi, alo;
    mul_wide64(x, y, ahi, alo);
    u64 H = mul_hi64(alo * qinv, q);
    return ahi - H + q;
}

// MRed, modular_reduction.go:70
LR_HD u64 mred(u64 x, u64 y, u64 q, u64 qinv) { return cred(mred_constant(x, y, q, qinv), q); }

// MFormConstant / MForm, modular_reduction.go:26,15 (u = {hi, lo} of floor(2^128/q))
LR_HD u64 mform_constant(u64 a, u64 q, u64 u_hi, u64 u_lo) {
    u64 mhi = mul_hi64(a, u_lo);
    return (u64)(0 - (a * u_hi + mhi)) * q;
}
LR_HD u64 mform(u64 a, u64 q, u64 u_hi, u64 u_lo) { return cred(mform_constant(a, q, u_hi, u_lo), q); }

// InvMForm, modular_reduction.go:34
LR_HD u64 inv_mform(u64 a, u64 q, u64 qinv) {
    u64 r = q - mul_hi64(a * qinv, q);
    return cred(r, q);
}

// BRedAddConstant / BRedAdd, modular_reduction.go:123,112 -- exact x mod q for any 64-bit x
LR_HD u64 bred_add_constant(u64 x, u64 q, u64 u_hi) { return x - mul_hi64(x, u_hi) * q; }
LR_HD u64 bred_add(u64 x, u64 q, u64 u_hi) { return cred(bred_add_constant(x, q, u_hi), q); }

// BRedConstant / BRed, modular_reduction.go:172,133 (same carry chain as the reference)
LR_HD u64 bred_constant(u64 x, u64 y, u64 q, u64 u_hi, u64 u_lo) {
    u64 ahi, alo, mhi, mlo;
    mul_wide64(x, y, ahi, alo);
    u64 lhi = mul_hi64(alo, u_lo);
    mul_wide64(alo, u_hi, mhi, mlo);
    u64 s0 = mlo + lhi;
    u64 s1 = mhi + (u64)(s0 < mlo);
    mul_wide64(ahi, u_lo, mhi, mlo);
    u64 t = mlo + s0;
    lhi = mhi + (u64)(t < mlo);
    s0 = ahi * u_hi + s1 + lhi;
    return alo - s0 * q;
}
LR_HD u64 bred(u64 x, u64 y, u64 q, u64 u_hi, u64 u_lo) { return cred(bred_constant(x, y, q, u_hi, u_lo), q); }

// PowerOf2, ring/utils.go:8 (Go shift semantics: a count >= 64 yields 0)
LR_HD u64 power_of_2(u64 x, u64 n, u64 q, u64 qinv) {
    u64 ahi = (n == 0 || n > 64) ? 0 : (x >> (64 - n));
    u64 alo = (n >= 64) ? 0 : (x << n);
    u64 H = mul_hi64(alo * qinv, q);
    return cred(ahi - H + q, q);
}

// ---------------------------------------------------------------------------------------
// Kernel-internal lazy multiplication by a precomputed constant (Shoup/Harvey form).
//   w  : the constant, plain domain, < q
//   ws : floor(w * 2^64 / q)
// mul_shoup_lazy(v) == v*w - qhat*q with qhat = an under-estimate of floor(v*ws/2^64) by at
// most 2, hence the result is congruent to v*w (mod q) and lies in [0, 4q) for ANY 64-bit v
// (exact Shoup gives [0, 2q); each unit of quotient deficit adds q).  Only congruence is
// relied on: every kernel ends in an exact canonical reduction, so outputs equal the
// reference's bit for bit.  9 32-bit multiplies instead of MRedConstant's 11.
// ---------------------------------------------------------------------------------------
LR_HD u64 mul_shoup_lazy(u64 v, u64 w, u64 ws, u64 q) {
    const u32 v0 = (u32)v, v1 = (u32)(v >> 32), s0 = (u32)ws, s1 = (u32)(ws >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 qhat = (u64)v1 * s1 + (u64)__umulhi(v1, s0) + (u64)__umulhi(v0, s1);
    // v*w - qhat*q (mod 2^64) as multiply-accumulate chains on the negated modulus: the 64-bit
    // accumulate of v_mad_u64_u32 absorbs the additions (2 fewer VALU ops than mul/mul/sub)
    const u64 nq = 0 - q;
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32), n0 = (u32)nq, n1 = (u32)(nq >> 32);
    const u32 h0 = (u32)qhat, h1 = (u32)(qhat >> 32);
    u64 acc = (u64)v0 * w0;
    acc = (u64)h0 * n0 + acc;
    u32 hi = (u32)(acc >> 32);
    hi = (u32)((u64)v0 * w1 + hi);
    hi = (u32)((u64)v1 * w0 + hi);
    hi = (u32)((u64)h0 * n1 + hi);
    hi = (u32)((u64)h1 * n0 + hi);
    return ((u64)hi << 32) | (u32)acc;
#else
    const u64 qhat = (u64)v1 * s1 + (((u64)v1 * s0) >> 32) + (((u64)v0 * s1) >> 32);
    return v * w - qhat * q;
#endif
}

// same contract, written as two low products and a subtraction: two more VALU ops but fewer live
// 64-bit temporaries (used where the kernel is register-starved)
LR_HD u64 mul_shoup_lazy_lowreg(u64 v, u64 w, u64 ws, u64 q) {
    const u32 v0 = (u32)v, v1 = (u32)(v >> 32), s0 = (u32)ws, s1 = (u32)(ws >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 qhat = (u64)v1 * s1 + (u64)__umulhi(v1, s0) + (u64)__umulhi(v0, s1);
#else
    const u64 qhat = (u64)v1 * s1 + (((u64)v1 * s0) >> 32) + (((u64)v0 * s1) >> 32);
#endif
    return v * w - qhat * q;
}

// exact Shoup multiplication: result in [0, 2q) for any 64-bit v
LR_HD u64 mul_shoup_exact(u64 v, u64 w, u64 ws, u64 q) { return v * w - mul_hi64(v, ws) * q; }

// floor(w * 2^64 / q), host only (u128 division)
inline u64 shoup_companion(u64 w, u64 q) { return (u64)((((u128)w) << 64) / q); }

}  // namespace lr
