// lr_bext.hip -- RNS fast basis extension (HPS, eprint 2018/117) for gfx950.
//
// One kernel serves modUpExact (ring/ring_basis_extension.go:352-393) and the three copies of
// its body inside Decompose / DecomposeAndSplit (:547-594, :663-710): one thread per
// coefficient column computes y_i = x_i * (Q/q_i)^-1 mod q_i and the float64 correction index
// v = floor(sum y_i / q_i) ONCE, then walks the output limbs.  The float path is the
// reference's, operation for operation: uint64->double (round to nearest even), IEEE
// division (div_by_const below: the same quotient bit for bit, computed with the host's
// reciprocal of the constant divisor), left-to-right accumulation, truncation;
// this file is compiled with -ffp-contract=off and without fast-math.  All table constants are wave-uniform and
// come in through scalar loads.
#include "lr_device.hpp"

namespace lr {

// a / b, correctly rounded, for a divisor that is a table constant: r = RN(1 / b) comes from the host.  q0 = RN(a r) is within
// two ulps of a / b; one residual step makes it faithful; a second one, on a faithful quotient with the exactly representable
// residual a - b q1, rounds to RN(a / b) (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, 4.7).  No scaling
// or fix-up is needed: 0 <= a < 2^64 and 2 <= b < 2^64 keep every intermediate far from the overflow and subnormal ranges.  The
// theorem's precondition on the divisor -- its significand is not all ones -- is checked on the host for every modulus of a table
// (ExtTables::fast_div_ok, DevModup::init); a table that fails it runs on the reference-shaped kernel and its plain division.
// Five full-rate instructions instead of the eleven of the generic IEEE expansion (v_div_scale x2, v_rcp, four Newton FMAs, multiply,
// residual, v_div_fmas, v_div_fixup), bit for bit the same quotient (lr_selftest_division compares the two on the device).
__device__ __forceinline__ double div_by_const(double a, double b, double r) {
    const double q0 = a * r;
    const double q1 = __builtin_fma(__builtin_fma(-b, q0, a), r, q0);
    return __builtin_fma(__builtin_fma(-b, q1, a), r, q1);
}




template <int NIN>
__global__ __launch_bounds__(256) void ext_kernel(ExtLaunch L) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= L.n) return;
    const long long b = blockIdx.y;
    const u64 *in = L.in + b * L.in_stride + (long long)L.in_limb0 * L.n + x;
    u64 y[NIN];
    double vi = 0.0;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const u64 qi = L.t.Q[i];
        y[i] = mred(in[(long long)i * L.n], L.t.qib_mont[i], qi, L.t.mredQ[i]);
        vi += (double)y[i] / (double)qi;
    }
    const u64 v = (u64)vi;
#pragma unroll
    for (int s = 0; s < kExtSegments; ++s) {
        const ExtSegment sg = L.seg[s];
        u64 *out = sg.out + b * sg.stride + (long long)sg.limb0 * L.n + x;
        for (int jj = 0; jj < sg.count; ++jj) {
            const int col = sg.col0 + jj;
            const u64 pj = L.t.P[col], pinv = L.t.mredP[col], bh = L.t.bredP_hi[col];
            u64 acc = 0;
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                acc += mred(y[i], L.t.qispj_mont[(long long)i * L.t.nP + col], pj, pinv);
                if ((i & 7) == 6) acc = bred_add(acc, pj, bh);
            }
            out[(long long)jj * L.n] = bred_add(acc + L.t.qpj_inv[(long long)col * (L.t.nQ + 1) + v], pj, bh);
        }
    }
}

// Same result through Shoup-form constants: every output is the canonical residue of
// sum_i y_i * (Q/q_i) + qpjInv[v] (mod p_j), which does not depend on how the sum is carried, so the terms are
// taken lazily ([0,4p) with 9 multiplies, or [0,2p) when the moduli leave less room) and Barrett-reduced only
// when `chunk` of them have accumulated -- for moduli below 2^57 never before the final BRedAdd.  Two
// coefficients per thread: 16-byte accesses and two independent dependency chains.
// EVERY = 0: lazy [0,4p) terms, no intermediate reduction (host guarantees 4p * NIN + p < 2^64);
// EVERY = 3 / 7: [0,2p) terms, BRedAdd after every 3rd (any p < 2^61) / 7th (p <= 2^60) term.
template <int NIN, int EVERY, int W>
__global__ __launch_bounds__(256) void ext_shoup_kernel(ExtLaunch L) {
    constexpr bool EXACT = EVERY != 0;
    const int xw = blockIdx.x * 256 + threadIdx.x;
    if (W * xw >= L.n) return;
    const long long b = blockIdx.y;
    const u64 *in = L.in + b * L.in_stride + (long long)L.in_limb0 * L.n + W * xw;
    u64 y[W][NIN];
    double vf[W];
#pragma unroll
    for (int w = 0; w < W; ++w) vf[w] = 0.0;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const u64 qi = L.t.Q[i];
        const double qr = L.t.Qrcp[i];
        u64 v[W];
        if (W == 2) {
            const ulonglong2 t = ld_stream(reinterpret_cast<const ulonglong2 *>(in + (long long)i * L.n));
            v[0] = t.x;
            v[W - 1] = t.y;
        } else {
            v[0] = ld_stream(in + (long long)i * L.n);
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            y[w][i] = mred(v[w], L.t.qib_mont[i], qi, L.t.mredQ[i]);
            vf[w] += div_by_const((double)y[w][i], (double)qi, qr);
        }
    }
    u64 vi[W];
#pragma unroll
    for (int w = 0; w < W; ++w) vi[w] = (u64)vf[w];
#pragma unroll
    for (int s = 0; s < kExtSegments; ++s) {
        const ExtSegment sg = L.seg[s];
        u64 *out = sg.out + b * sg.stride + (long long)sg.limb0 * L.n + W * xw;
        for (int jj = 0; jj < sg.count; ++jj) {
            const int col = sg.col0 + jj;
            const u64 pj = ld_const(L.t.P + col), bh = ld_const(L.t.bredP_hi + col);
            u64 acc[W];
#pragma unroll
            for (int w = 0; w < W; ++w) acc[w] = 0;
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const ulonglong2 c = ld_const(L.t.qispj_shoup + (long long)i * L.t.nP + col);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    acc[w] += EXACT ? mul_shoup_exact(y[w][i], c.x, c.y, pj) : mul_shoup_lazy(y[w][i], c.x, c.y, pj);
                    if (EXACT && (i % (EXACT ? EVERY : 1)) == EVERY - 1 && i + 1 < NIN) acc[w] = bred_add(acc[w], pj, bh);
                }
            }
            const u64 *corr = L.t.qpj_inv + (long long)col * (L.t.nQ + 1);
            if (EXACT) {
#pragma unroll
                for (int w = 0; w < W; ++w) acc[w] = bred_add(acc[w] + corr[vi[w]], pj, bh);
            } else {
                // qpjInv[j][v] = v * qpjInv[j][1] mod p_j (ring_basis_extension.go:134-138): the product instead of a
                // per-lane table look-up; the host admits this path only if NIN lazy terms plus NIN * p fit in 64 bits
                const u64 nq = ld_const(corr + 1);
#pragma unroll
                for (int w = 0; w < W; ++w) acc[w] = bred_add(acc[w] + vi[w] * nq, pj, bh);
            }
            if (W == 2) st_stream(reinterpret_cast<ulonglong2 *>(out + (long long)jj * L.n), make_ulonglong2(acc[0], acc[W - 1]));
            else st_stream(out + (long long)jj * L.n, acc[0]);
        }
    }
}

// The lazy case once more, with the quotient taken once per output instead of once per term:
//   sum_i (y_i*c_ij - qhat_i*p_j) + v*qpjInv[j][1]  ==  sum_i y_i*c_ij  -  (sum_i qhat_i)*p_j  +  v*qpjInv[j][1]   (mod 2^64)
// and the left side is below 2^64 (the host admits the lazy path only then), so the value -- and with it the canonical
// residue bred_add returns -- is the one ext_shoup_kernel<NIN, 0, W> computes.  The low 64 bits of every product are
// gathered in two multiply-accumulate chains (`lo`: the 2^0 column with its carries; `hi`: the 2^32 column, of which only
// the low word is kept), the quotient estimates in a third and a fourth: 7 VALU instructions per term instead of about 28.
// The quotient sum falls short of sum_i floor(y_i*c_ij / p_j) by fewer than 3*NIN + NIN/8 + 2, so the value stays below
// (4*NIN + NIN/8 + 2) * p_j <= 5*NIN*p_j for NIN >= 2, the room the host checks (ExtTables::lazy_terms).
// multiply-accumulate steps spelled out, so that every one is a single v_mad_u64_u32 whatever part of the result is used
// later (left to itself the compiler narrows the `hi` chain to v_mul_lo_u32 + v_add3_u32 and widens the quotient sums to
// add / add-with-carry pairs): c = wave-uniform table word (SGPR), y = per-lane word
__device__ __forceinline__ u64 mad_word(u32 c, u32 y, u64 acc) {
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "s"(c), "v"(y), "v"(acc));
    return d;
}
__device__ __forceinline__ u64 mad_word_vv(u32 a, u32 y, u64 acc) {
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(y), "v"(acc));
    return d;
}
// acc + floor(c * y / 2^32): v_mul_hi_u32, then x * 1 + acc
__device__ __forceinline__ u64 mad_hi_word(u32 c, u32 y, u64 acc) {
    const u32 h = __umulhi(c, y);
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(carry) : "v"(h), "v"(acc));
    return d;
}

// x - c if that does not borrow, else x (the borrow of the subtraction is the select mask: 4 instructions)
__device__ __forceinline__ u64 csub_borrow(u64 x, u64 c) {
    u64 t;
    const bool borrow = __builtin_usubl_overflow(x, c, &t);
    return borrow ? x : t;
}

// canonical residue of x modulo p for p > 2^32 and floor(x / p) < 2^32, with a one-word quotient estimate: u = floor(2^64 / p)
// fits a word, qe = floor((x >> 32) * u / 2^32) is below floor(x / p) by at most 3 (the dropped low word of x is worth less than
// one p, u is short of 2^64 / p by less than one, the floor by one more), so x - qe * p lies in [0, 4p)
__device__ __forceinline__ u64 bred_word(u64 x, u64 p, u64 p2, u32 u) {
    const u32 qe = __umulhi((u32)(x >> 32), u);
    const u64 r = x - (u64)qe * p;
    return csub_borrow(csub_borrow(r, p2), p);
}

// TOP (N = 2^16 key switch): a thread takes the coefficient pairs at j and at j + N/2 and writes, instead of the extension x of the
// two halves, the first forward stage over them -- X = x[j] + psi[1] * x[j + N/2], Y = x[j] - psi[1] * x[j + N/2] (lazy, below 3p),
// psi[1] from the target limb's forward table -- so that the 2^15 sub-block transforms that follow read their own half only
// (the "p" kernels) instead of both halves (the fused "s" kernels, 1.5 x the traffic).  The transform is the canonical one either
// way: its first stage has merely moved into the kernel that produces its input.

// ExtSegment::epi_mode on one canonical extension value
__device__ __forceinline__ u64 ext_epilogue(int mode, u64 e, u64 x, u64 p, u64 pinv, u64 c, u64 sc) {
    if (mode == 1) return cred(mred(x + (p - e), c, p, pinv) + sc, p);
    return mred(cred(e + (p - sc), p), c, p, pinv);
}

// constants of one target limb (column of the extension tables), wave-uniform
template <int NIN>
struct ExtColumn {
    u64 pj, bh, nq;
    ulonglong2 c[NIN], tw;
    __device__ __forceinline__ void load(const ExtTables &t, int col, const Twiddle *top_tw, int top_mod, int n) {
        pj = ld_const(t.P + col);
        bh = ld_const(t.bredP_hi + col);
        nq = ld_const(t.qpj_inv + (long long)col * (t.nQ + 1) + 1);
#pragma unroll
        for (int i = 0; i < NIN; ++i) c[i] = ld_const(t.qispj_shoup + (long long)i * t.nP + col);
        // twiddle of the stage over index bit logN - 1: heap entry 1 of the target limb's forward table {w, floor(w 2^64 / p)}
        tw = top_tw ? ld_const(reinterpret_cast<const ulonglong2 *>(top_tw) + ((long long)top_mod * n + 1)) : make_ulonglong2(0, 0);
    }
};

template <int NIN, int W, bool TOP>
__device__ __forceinline__ void ext_sum_body(const ExtLaunch &L) {
    constexpr int C = TOP ? 2 * W : W;                         // coefficient columns per thread
    const int xw = blockIdx.x * 256 + threadIdx.x;
    const int span = TOP ? L.n >> 1 : L.n;
    if (W * xw >= span) return;
    const long long b = blockIdx.y;
    const u64 *in = L.in + b * L.in_stride + (long long)L.in_limb0 * L.n + W * xw;
    u32 y0[C][NIN], y1[C][NIN];
    double vf[C];
#pragma unroll
    for (int w = 0; w < C; ++w) vf[w] = 0.0;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const u64 qi = L.t.Q[i];
        const double qr = L.t.Qrcp[i];
        u64 v[C];
#pragma unroll
        for (int h = 0; h < C / W; ++h) {
            const u64 *src = in + (long long)i * L.n + (long long)h * span;
            if (W == 2) {
                const ulonglong2 t = ld_stream(reinterpret_cast<const ulonglong2 *>(src));
                v[h * W] = t.x;
                v[h * W + W - 1] = t.y;
            } else {
                v[h * W] = ld_stream(src);
            }
        }
        if constexpr (TOP) {
            if (L.inv_top) {
                // v[w] / v[W + w] = the halves U, V (below 8q, or 4q for a modulus above 2^60) of the limb before its last inverse stage:
                // the finished coefficients are (U + V) N^-1 and (U - V) psi_inv[1] N^-1, and both constants ride in the Montgomery
                // multiplier that turns a coefficient into y_i anyway
                const u64 bound = (qi >> 60) ? qi << 2 : qi << 3;
                const u64 k0 = L.t.invtop0[i], k1 = L.t.invtop1[i], qinv = L.t.mredQ[i];
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const u64 U = v[w], V = v[W + w];
                    const u64 ya = mred(U + V, k0, qi, qinv), yb = mred(U + bound - V, k1, qi, qinv);
                    vf[w] += div_by_const((double)ya, (double)qi, qr);
                    vf[W + w] += div_by_const((double)yb, (double)qi, qr);
                    y0[w][i] = (u32)ya;
                    y1[w][i] = (u32)(ya >> 32);
                    y0[W + w][i] = (u32)yb;
                    y1[W + w][i] = (u32)(yb >> 32);
                }
                continue;
            }
        }
#pragma unroll
        for (int w = 0; w < C; ++w) {
            const u64 y = mred(v[w], L.t.qib_mont[i], qi, L.t.mredQ[i]);
            vf[w] += div_by_const((double)y, (double)qi, qr);
            y0[w][i] = (u32)y;
            y1[w][i] = (u32)(y >> 32);
        }
    }
    u32 vi[C];
#pragma unroll
    for (int w = 0; w < C; ++w) vi[w] = (u32)(u64)vf[w];
#pragma unroll
    for (int s = 0; s < kExtSegments; ++s) {
        const ExtSegment sg = L.seg[s];
        u64 *out = sg.out + b * sg.stride + (long long)sg.limb0 * L.n + W * xw;
        // the per-target constants of column jj + 1 are requested (scalar loads) before column jj's arithmetic starts, so that their
        // latency runs under ~200 vector instructions instead of in front of them
        ExtColumn<NIN> next;
        if (sg.count > 0) next.load(L.t, sg.col0, TOP ? sg.top_tw : nullptr, sg.top_mod0, L.n);
        for (int jj = 0; jj < sg.count; ++jj) {
            const ExtColumn<NIN> cur = next;
            if (jj + 1 < sg.count) next.load(L.t, sg.col0 + jj + 1, TOP ? sg.top_tw : nullptr, sg.top_mod0 + jj + 1, L.n);
            const u64 pj = cur.pj, bh = cur.bh, nq = cur.nq;
            u64 lo[C], hi[C], qs[C], xs[C];
#pragma unroll
            for (int w = 0; w < C; ++w) lo[w] = hi[w] = qs[w] = xs[w] = 0;
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const ulonglong2 c = cur.c[i];
                const u32 w0 = (u32)c.x, w1 = (u32)(c.x >> 32), s0 = (u32)c.y, s1 = (u32)(c.y >> 32);
#pragma unroll
                for (int w = 0; w < C; ++w) {
                    lo[w] = mad_word(w0, y0[w][i], lo[w]);
                    hi[w] = mad_word(w1, y0[w][i], hi[w]);
                    hi[w] = mad_word(w0, y1[w][i], hi[w]);
                    qs[w] = mad_word(s1, y1[w][i], qs[w]);
                    // y1 < 2^29 (every q_i < 2^61): eight of these products fit one 64-bit sum, whose high word joins the quotient
                    xs[w] = mad_word(s0, y1[w][i], xs[w]);
                    if ((i & 7) == 7 || i == NIN - 1) {
                        qs[w] += xs[w] >> 32;
                        xs[w] = 0;
                    }
                    qs[w] = mad_hi_word(s1, y0[w][i], qs[w]);
                }
            }
            const u64 np = 0 - pj;
            const u32 n0 = (u32)np, n1 = (u32)(np >> 32), c0 = (u32)nq, c1 = (u32)(nq >> 32);
            u64 r[C];
#pragma unroll
            for (int w = 0; w < C; ++w) {
                const u32 h0 = (u32)qs[w], h1 = (u32)(qs[w] >> 32);
                lo[w] = mad_word(n0, h0, lo[w]);
                hi[w] = mad_word(n1, h0, hi[w]);
                hi[w] = mad_word(n0, h1, hi[w]);
                lo[w] = mad_word(c0, vi[w], lo[w]);
                hi[w] = mad_word(c1, vi[w], hi[w]);
                r[w] = bred_word(lo[w] + (hi[w] << 32), pj, pj << 1, (u32)bh);
            }
            if constexpr (TOP) {
                const ulonglong2 tw = cur.tw;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const u64 t = mul_shoup_exact(r[W + w], tw.x, tw.y, pj);       // in [0, 2p)
                    r[W + w] = r[w] + (pj << 1) - t;                                  // Y in (0, 3p)
                    r[w] = r[w] + t;                                                  // X in [0, 3p)
                }
            }
            if (!TOP && sg.epi_mode) {
                const int col = sg.col0 + jj;
                const u64 pinv = ld_const(L.t.mredP + col), ec = ld_const(sg.epi_c + col), es = sg.epi_s ? ld_const(sg.epi_s + col) : 0;
                const u64 *px = sg.epi_x + b * sg.epi_x_stride + (long long)(sg.limb0 + jj) * L.n + W * xw;
                const u64 *px2 = sg.epi_x2 ? sg.epi_x2 + b * sg.epi_x2_stride + (long long)(sg.limb0 + jj) * L.n + W * xw : nullptr;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    u64 x = sg.epi_mode == 1 ? ld_stream(px + w) : 0;
                    if (sg.epi_mode == 1 && px2) x = cred(x + ld_stream(px2 + w), pj);
                    r[w] = ext_epilogue(sg.epi_mode, r[w], x, pj, pinv, ec, es);
                }
            }
#pragma unroll
            for (int h = 0; h < C / W; ++h) {
                u64 *dst = out + (long long)jj * L.n + (long long)h * span;
                if (W == 2) st_stream(reinterpret_cast<ulonglong2 *>(dst), make_ulonglong2(r[h * W], r[h * W + W - 1]));
                else st_stream(dst, r[h * W]);
            }
        }
    }
}

// Moduli that leave no room for lazy terms (60-bit QMul / P): the products are summed exactly in 128 bits and reduced ONCE,
// by a Montgomery reduction -- sum_i MRed(y_i, qispj_mont[i][j]) and MRed-of-the-sum are the same residue modulo p_j, and the
// reduction's precondition (sum < p_j * 2^64) holds because NIN * max q_i < 2^64 (ExtTables::wide_ok, checked by the host).
// With y = y1 * 2^32 + y0 and c = c1 * 2^32 + c0 (y1, c1 < 2^29: every modulus is below 2^61) the sum is gathered by columns:
//   lo  += y0 * c0            one v_mad_u64_u32 whose carry-out is counted (+1 instruction),
//   mid += y0 * c1 + y1 * c0  two v_mad_u64_u32 (each product < 2^61; folded into lo / hi every four terms),
//   hi  += y1 * c1            one v_mad_u64_u32 (each product < 2^58).
// About 6 VALU instructions per term instead of the ~23 of an exact Shoup product with its share of the Barrett steps.
__device__ __forceinline__ void mad_carry(u32 c, u32 y, u64 &acc, u32 &carries) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(carries) : "s"(c), "v"(y) : "vcc");
}

template <int NIN, int W, bool TOP>
__global__ __launch_bounds__(256) void ext_sum_kernel(ExtLaunch L) {
    ext_sum_body<NIN, W, TOP>(L);
}

// the same on several extensions at once (grid z): the digits of one key switch are independent launches of the same shape, and at a
// small batch each alone fills a quarter of the chip.  The launch records travel in the kernel-argument block and are read by index.
template <int NIN, int W, bool TOP>
__global__ __launch_bounds__(256) void ext_sum_group_kernel(ExtGroupLaunch G) {
    ext_sum_body<NIN, W, TOP>(G.L[blockIdx.z]);
}

// G = terms per Montgomery reduction (G * max q_i < 2^64); NIN > G: the partial residues (each below p_j) are added up.
template <int NIN, int W, int G>
__device__ __forceinline__ void ext_wide_body(const ExtLaunch &L);

template <int NIN, int W, int G>
__global__ __launch_bounds__(256) void ext_wide_kernel(ExtLaunch L) {
    ext_wide_body<NIN, W, G>(L);
}

// grouped form (grid z = record), as ext_sum_group_kernel: the column ranges of a small launch
template <int NIN, int W, int G>
__global__ __launch_bounds__(256) void ext_wide_group_kernel(ExtGroupLaunch Gr) {
    ext_wide_body<NIN, W, G>(Gr.L[blockIdx.z]);
}

template <int NIN, int W, int G>
__device__ __forceinline__ void ext_wide_body(const ExtLaunch &L) {
    const int xw = blockIdx.x * 256 + threadIdx.x;
    if (W * xw >= L.n) return;
    const long long b = blockIdx.y;
    const u64 *in = L.in + b * L.in_stride + (long long)L.in_limb0 * L.n + W * xw;
    u32 y0[W][NIN], y1[W][NIN];
    double vf[W];
#pragma unroll
    for (int w = 0; w < W; ++w) vf[w] = 0.0;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const u64 qi = L.t.Q[i];
        const double qr = L.t.Qrcp[i];
        u64 v[W];
        if (W == 2) {
            const ulonglong2 t = ld_stream(reinterpret_cast<const ulonglong2 *>(in + (long long)i * L.n));
            v[0] = t.x;
            v[W - 1] = t.y;
        } else {
            v[0] = ld_stream(in + (long long)i * L.n);
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const u64 y = mred(v[w], L.t.qib_mont[i], qi, L.t.mredQ[i]);
            vf[w] += div_by_const((double)y, (double)qi, qr);
            y0[w][i] = (u32)y;
            y1[w][i] = (u32)(y >> 32);
        }
        // long inputs: keep the scheduler from pulling every limb's load (and its 64-bit temporaries) to the top -- with 32 limbs
        // that cost 131 instead of 98 VGPRs, i.e. three instead of five waves per SIMD
        if (NIN > 16 && (i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    u32 vi[W];
#pragma unroll
    for (int w = 0; w < W; ++w) vi[w] = (u32)(u64)vf[w];
#pragma unroll
    for (int s = 0; s < kExtSegments; ++s) {
        const ExtSegment sg = L.seg[s];
        u64 *out = sg.out + b * sg.stride + (long long)sg.limb0 * L.n + W * xw;
        for (int jj = 0; jj < sg.count; ++jj) {
            const int col = sg.col0 + jj;
            const u64 pj = ld_const(L.t.P + col), pinv = ld_const(L.t.mredP + col);
            const u64 *corr = L.t.qpj_inv + (long long)col * (L.t.nQ + 1);
            u64 lo[W], mid[W], hi[W], r[W];
            u32 cy[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                lo[w] = mid[w] = hi[w] = 0;
                cy[w] = 0;
                r[w] = corr[vi[w]];                                           // qpjInv[j][v], canonical
            }
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const u64 c = ld_const(L.t.qispj_mont + (long long)i * L.t.nP + col);
                const u32 c0 = (u32)c, c1 = (u32)(c >> 32);
                const bool group_end = i == NIN - 1 || i % G == G - 1;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    mad_carry(c0, y0[w][i], lo[w], cy[w]);
                    mid[w] = mad_word(c1, y0[w][i], mid[w]);
                    mid[w] = mad_word(c0, y1[w][i], mid[w]);
                    hi[w] = mad_word(c1, y1[w][i], hi[w]);
                    if ((i & 3) == 3 || group_end) {
                        const u64 low_part = mid[w] << 32;
                        lo[w] += low_part;
                        hi[w] += (mid[w] >> 32) + (lo[w] < low_part ? 1 : 0);
                        mid[w] = 0;
                    }
                    if (group_end) {
                        const u64 th = hi[w] + cy[w];                         // the group's sum is th * 2^64 + lo, th < p_j
                        const u64 H = mul_hi64(lo[w] * pinv, pj);             // MRed on the 128-bit sum (modular_reduction.go:70)
                        r[w] = cred(r[w] + cred(th - H + pj, pj), pj);
                        lo[w] = hi[w] = 0;
                        cy[w] = 0;
                    }
                }
            }
            if (sg.epi_mode) {
                const u64 ec = ld_const(sg.epi_c + col), es = sg.epi_s ? ld_const(sg.epi_s + col) : 0;
                const u64 *px = sg.epi_x + b * sg.epi_x_stride + (long long)(sg.limb0 + jj) * L.n + W * xw;
                const u64 *px2 = sg.epi_x2 ? sg.epi_x2 + b * sg.epi_x2_stride + (long long)(sg.limb0 + jj) * L.n + W * xw : nullptr;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    u64 x = sg.epi_mode == 1 ? ld_stream(px + w) : 0;
                    if (sg.epi_mode == 1 && px2) x = cred(x + ld_stream(px2 + w), pj);
                    r[w] = ext_epilogue(sg.epi_mode, r[w], x, pj, pinv, ec, es);
                }
            }
            if (W == 2) st_stream(reinterpret_cast<ulonglong2 *>(out + (long long)jj * L.n), make_ulonglong2(r[0], r[W - 1]));
            else st_stream(out + (long long)jj * L.n, r[0]);
        }
    }
}

template <int NIN>
static hipError_t launch_n(const ExtLaunch &L, int batch, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale (non-sticky) errors of unrelated earlier calls
    if ((L.n & 1) == 0 && L.t.exact_terms >= 4 && L.t.fast_div_ok) {   // exact_terms >= 4 <=> every p < 2^61
        // two coefficients per thread while their y_i fit comfortably in registers
        constexpr int W = NIN <= 20 ? 2 : 1;
        const dim3 grid((unsigned)((L.n / W + 255) / 256), (unsigned)batch), block(256);
        if (L.seg[0].top_tw != nullptr) {
            // top-stage variant (ext_top_supported() has vetted the tables): half the threads, four columns each
            if constexpr (NIN <= 8) {
                // one column per half and thread beyond three input limbs: with two, four digits' worth of y_i push the kernel to
                // 85 VGPRs (5 waves per SIMD instead of 8)
                constexpr int WT = NIN <= 3 ? 2 : 1;
                const dim3 gtop((unsigned)((L.n / 2 / WT + 255) / 256), (unsigned)batch);
                hipLaunchKernelGGL((ext_sum_kernel<NIN, WT, true>), gtop, block, 0, stream, L);
                return hipGetLastError();
            } else {
                return hipErrorInvalidValue;
            }
        }
        if (L.t.lazy_terms >= (NIN < 2 ? 2 : NIN) && L.t.word_barrett) hipLaunchKernelGGL((ext_sum_kernel<NIN, W, false>), grid, block, 0, stream, L);
        else if (L.t.lazy_terms >= NIN) hipLaunchKernelGGL((ext_shoup_kernel<NIN, 0, W>), grid, block, 0, stream, L);
        else if (L.t.wide_ok >= NIN) hipLaunchKernelGGL((ext_wide_kernel<NIN, W, NIN>), grid, block, 0, stream, L);
        else if (L.t.wide_ok >= 16 && NIN > 16) hipLaunchKernelGGL((ext_wide_kernel<NIN, W, (NIN > 16 ? 16 : NIN)>), grid, block, 0, stream, L);
        else if (L.t.wide_ok >= 8 && NIN > 8) hipLaunchKernelGGL((ext_wide_kernel<NIN, W, (NIN > 8 ? 8 : NIN)>), grid, block, 0, stream, L);
        else if (L.t.exact_terms >= 8) hipLaunchKernelGGL((ext_shoup_kernel<NIN, 7, W>), grid, block, 0, stream, L);
        else hipLaunchKernelGGL((ext_shoup_kernel<NIN, 3, W>), grid, block, 0, stream, L);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((L.n + 255) / 256), (unsigned)batch), block(256);
    hipLaunchKernelGGL(ext_kernel<NIN>, grid, block, 0, stream, L);
    return hipGetLastError();
}

// the top-stage variant exists for the sum-form kernel with at most eight input limbs (the key-switch digits have alpha <= 8)
bool ext_top_supported(const ExtTables &t, int n_in, int n) {
    return (n & 3) == 0 && n_in >= 1 && n_in <= 8 && t.exact_terms >= 4 && t.fast_div_ok && t.word_barrett && t.lazy_terms >= (n_in < 2 ? 2 : n_in);
}

// ExtSegment::epi_mode is implemented by the sum-form and the 128-bit-sum kernels: does launch_n pick one of them for this shape?
bool ext_epilogue_supported(const ExtTables &t, int n_in, int n) {
    if ((n & 1) != 0 || t.exact_terms < 4 || !t.fast_div_ok) return false;
    if (t.lazy_terms >= (n_in < 2 ? 2 : n_in) && t.word_barrett) return true;
    if (t.lazy_terms >= n_in) return false;
    return t.wide_ok >= n_in || (t.wide_ok >= 16 && n_in > 16) || (t.wide_ok >= 8 && n_in > 8);
}

// ---- diagnostics: div_by_const against the IEEE division it replaces ---------------------------------------------------------
// Every thread walks `per_thread` (a, b) pairs from a splitmix64 stream: divisors of every size the extension tables can hold
// (30 .. 61 bits, odd and even, plus powers of two and their neighbours), dividends uniform over 64 bits, below the divisor (the
// product's case: y_i < q_i), multiples of the divisor +- 1 and values around 2^53 where the uint64 -> double conversion rounds.
__global__ __launch_bounds__(256) void div_selftest_kernel(u64 seed, int per_thread, unsigned long long *mismatches) {
    u64 state = seed + ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        u64 z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    unsigned bad = 0;
    for (int k = 0; k < per_thread; ++k) {
        const u64 r0 = next(), r1 = next();
        const int bits = 30 + (int)(r0 % 32);                       // 30 .. 61
        u64 q = ((u64)1 << (bits - 1)) | (r1 >> (65 - bits));
        switch ((r0 >> 8) & 7) {
        case 0: q |= 1; break;                                       // odd, like every modulus
        case 1: q = (u64)1 << (bits - 1); break;                     // power of two
        case 2: q = ((u64)1 << bits) - 1; break;                     // all ones
        case 3: q = ((u64)1 << (bits - 1)) + 1; break;
        default: q |= 1; break;
        }
        u64 y = next();
        switch ((r0 >> 16) & 7) {
        case 0: break;                                               // any 64-bit value
        case 1: y = (y >> 11) % 64 * q + ((r0 >> 20) & 3) - 1; break;   // small multiples of q, +- 1
        case 2: y = ((u64)1 << 53) + (y & 0xFFF) - 0x800; break;     // where the conversion to double starts rounding
        case 3: y = q - 1 - (y & 3); break;
        default: y %= q; break;                                      // the product's case
        }
        const double a = (double)y, b = (double)q;
        const double want = a / b, got = div_by_const(a, b, 1.0 / b);
        bad += __double_as_longlong(want) != __double_as_longlong(got) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

hipError_t launch_div_selftest(u64 seed, int blocks, int per_thread, unsigned long long *d_mismatches, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(div_selftest_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, per_thread, d_mismatches);
    return hipGetLastError();
}

hipError_t launch_ext(const ExtLaunch &L, int n_in, int batch, hipStream_t stream) {
    if (batch <= 0) return hipSuccess;
    switch (n_in) {
#define LR_EXT(K) \
    case K: return launch_n<K>(L, batch, stream);
        LR_EXT(1) LR_EXT(2) LR_EXT(3) LR_EXT(4) LR_EXT(5) LR_EXT(6) LR_EXT(7) LR_EXT(8)
        LR_EXT(9) LR_EXT(10) LR_EXT(11) LR_EXT(12) LR_EXT(13) LR_EXT(14) LR_EXT(15) LR_EXT(16)
        LR_EXT(17) LR_EXT(18) LR_EXT(19) LR_EXT(20) LR_EXT(21) LR_EXT(22) LR_EXT(23) LR_EXT(24)
        LR_EXT(25) LR_EXT(26) LR_EXT(27) LR_EXT(28) LR_EXT(29) LR_EXT(30) LR_EXT(31) LR_EXT(32)
        LR_EXT(33) LR_EXT(34) LR_EXT(35) LR_EXT(36) LR_EXT(37) LR_EXT(38) LR_EXT(39) LR_EXT(40)
#undef LR_EXT
    default: return hipErrorInvalidValue;
    }
}

// Several extensions of one shape in one launch.  Only the sum-form kernel (what every key-switch digit of the default parameter sets
// takes) has the grouped form; hipErrorNotSupported tells the caller to launch the records one by one.
template <int NIN>
static hipError_t launch_group_n(const ExtLaunch *Ls, int count, int batch, hipStream_t stream) {
    if constexpr (NIN > 8) {
        return hipErrorNotSupported;
    } else {
        const ExtLaunch &L0 = Ls[0];
        const bool top = L0.seg[0].top_tw != nullptr;
        bool sum_form = true, wide_form = !top;     // (launch_n's order of preference: sum-form, per-term Shoup, 128-bit sums over all terms)
        for (int k = 0; k < count; ++k) {
            const ExtLaunch &L = Ls[k];
            if ((L.n & 1) != 0 || L.n != L0.n || L.t.exact_terms < 4 || !L.t.fast_div_ok || (L.seg[0].top_tw != nullptr) != top)
                return hipErrorNotSupported;
            const bool sum_ok = L.t.lazy_terms >= (NIN < 2 ? 2 : NIN) && L.t.word_barrett;
            sum_form = sum_form && sum_ok;
            wide_form = wide_form && !sum_ok && !(L.t.lazy_terms >= NIN) && L.t.wide_ok >= NIN && !L.inv_top;
        }
        if (!sum_form && !wide_form) return hipErrorNotSupported;
        ExtGroupLaunch G;
        for (int k = 0; k < count; ++k) G.L[k] = Ls[k];
        for (int k = count; k < kExtGroupMax; ++k) G.L[k] = Ls[0];
        (void)hipGetLastError();
        const dim3 block(256);
        if (top) {
            constexpr int WT = NIN <= 3 ? 2 : 1;
            const dim3 grid((unsigned)((L0.n / 2 / WT + 255) / 256), (unsigned)batch, (unsigned)count);
            hipLaunchKernelGGL((ext_sum_group_kernel<NIN, WT, true>), grid, block, 0, stream, G);
        } else {
            constexpr int W = 2;
            const dim3 grid((unsigned)((L0.n / W + 255) / 256), (unsigned)batch, (unsigned)count);
            if (sum_form) hipLaunchKernelGGL((ext_sum_group_kernel<NIN, W, false>), grid, block, 0, stream, G);
            else hipLaunchKernelGGL((ext_wide_group_kernel<NIN, W, NIN>), grid, block, 0, stream, G);
        }
        return hipGetLastError();
    }
}

hipError_t launch_ext_group(const ExtLaunch *Ls, int count, int n_in, int batch, hipStream_t stream) {
    if (batch <= 0 || count <= 0) return hipSuccess;
    if (count > kExtGroupMax) return hipErrorNotSupported;
    switch (n_in) {
    case 1: return launch_group_n<1>(Ls, count, batch, stream);
    case 2: return launch_group_n<2>(Ls, count, batch, stream);
    case 3: return launch_group_n<3>(Ls, count, batch, stream);
    case 4: return launch_group_n<4>(Ls, count, batch, stream);
    case 5: return launch_group_n<5>(Ls, count, batch, stream);
    case 6: return launch_group_n<6>(Ls, count, batch, stream);
    case 7: return launch_group_n<7>(Ls, count, batch, stream);
    case 8: return launch_group_n<8>(Ls, count, batch, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace lr
