// lr_float128.hpp -- the double-double arithmetic of ring/float128.go, operation for operation (host + device).
// Every function is a fixed sequence of IEEE-754 double operations in the reference's order; this translation unit is
// compiled with -ffp-contract=off and without fast-math (a fused multiply-add would change the error terms).
#pragma once
#include <cmath>

#include "lr_arith.hpp"

namespace lr {

struct F128 { double hi, lo; };

LR_HD F128 f128_set_uint53(u64 i) { return F128{(double)i, 0.0}; }                                   // float128.go:16
LR_HD F128 f128_set_uint64(u64 i) { return F128{(double)(i >> 12), (double)(i & 0xfff) / 4096.0}; }   // :22
LR_HD u64 f128_to_uint53(F128 f) { return (u64)f.hi; }                                               // :44
LR_HD u64 f128_to_uint64(F128 f) {                                                                    // :48
    const double s = f.hi * 4096.0;
    const u64 t = (u64)s;
    // a negative rounded value wraps modulo 2^64, as Go's amd64 float->uint64 conversion (through int64) does
    return t + (u64)(long long)round((s - (double)t) + f.lo * 4096.0);
}
LR_HD void two_sum(double a, double b, double &s, double &err) {      // :52
    s = a + b;
    const double bb = s - a;
    err = (a - (s - bb)) + (b - bb);
}
LR_HD void quick_two_sum(double a, double b, double &s, double &err) {   // :59
    s = a + b;
    err = b - (s - a);
}
LR_HD F128 f128_add(F128 a, F128 b) {                                  // :65
    double s1, s2, t1, t2;
    two_sum(a.hi, b.hi, s1, s2);
    two_sum(a.lo, b.lo, t1, t2);
    s2 += t1;
    quick_two_sum(s1, s2, s1, s2);
    s2 += t2;
    F128 f;
    quick_two_sum(s1, s2, f.hi, f.lo);
    return f;
}
LR_HD void two_diff(double a, double b, double &s, double &err) {     // :75
    s = a - b;
    const double bb = s - a;
    err = (a - (s - bb)) - (b + bb);
}
LR_HD void split(double a, double &hi, double &lo) {                  // :94
    const double temp = 134217729.0 * a;
    hi = temp - (temp - a);
    lo = a - hi;
}
LR_HD void two_prod(double a, double b, double &p, double &err) {     // :101
    p = a * b;
    double a_hi, a_lo, b_hi, b_lo;
    split(a, a_hi, a_lo);
    split(b, b_hi, b_lo);
    err = ((a_hi * b_hi - p) + a_hi * b_lo + a_lo * b_hi) + a_lo * b_lo;
}
LR_HD F128 f128_mul(F128 a, F128 b) {                                  // :109
    double p1, p2;
    two_prod(a.hi, b.hi, p1, p2);
    p2 += a.hi * b.lo + a.lo * b.hi;
    F128 f;
    quick_two_sum(p1, p2, f.hi, f.lo);
    return f;
}
LR_HD F128 f128_div(F128 a, F128 b) {                                  // :116
    const double q1 = a.hi / b.hi;
    double p1, p2;
    two_prod(q1, b.hi, p1, p2);
    p2 += q1 * b.lo;
    const double t0 = p1 + p2;
    const double t1 = p2 - (t0 - p1);
    double p3, p4, v1, v2;
    two_diff(a.hi, t0, p3, p4);
    two_diff(a.lo, t1, v1, v2);
    p4 += v1;
    quick_two_sum(p3, p4, p3, p4);
    p4 += v2;
    const double r = (p3 + p4) / b.hi;
    F128 f;
    f.hi = q1 + r;
    f.lo = r - (f.hi - q1);
    return f;
}

}  // namespace lr
