// lr_ntt.hip -- negacyclic NTT / InvNTT kernels for gfx950 (MI355X).
//
// Replaces the per-limb loops of ring/ntt.go:4-29 (Context.NTT/NTTLvl/InvNTT/InvNTTLvl) and
// the cores NTT (:53) / InvNTT (:89).  Same transform, same psi tables, canonical outputs, so
// results are bit-identical; the internal butterfly is our own (Shoup/Harvey lazy form,
// lr_arith.hpp) because only the canonical output is observable.
//
// Design (one workgroup = one limb of one polynomial, HBM traffic = 16*N bytes per limb):
//   index bit p of a coefficient is consumed by stage logN-1-p (forward: high bits first).
//   pass A   : each thread owns the 2^A coefficients {k*S + t} (S = N / 2^A = blockDim), loads
//              them coalesced straight from HBM and runs the top A stages in registers with
//              wave-uniform twiddles (scalar loads).
//   LDS pass : the remaining bits are consumed 3-4 at a time; the limb (or, for N = 2^15, one
//              half of it while the other half stays parked in registers) lives in LDS,
//              padded by 16 B per 16 coefficients so that every pass is bank-conflict free.
//   copy-out : canonical reduction fused with a coalesced 16 B/lane store.
// The inverse transform is the mirror image (copy-in, LDS passes low bits first, pass A last,
// scaling by N^-1 fused into the final store).
#include <atomic>
#include "lr_device.hpp"

namespace lr {

// ------------------------------------------------------------------------------------------
// butterflies.
//
// Forward (Cooley-Tukey): X = U + V*w, Y = U - V*w + 4q with V*w in [0,4q) for ANY 64-bit V
// (mul_shoup_lazy), so every stage grows the bound of a value by at most 4q and only U ever
// needs correcting.  How often depends on the head-room above q (MODE, chosen per context):
//   MODE 0  q < 2^61 : cond-subtract 4q before every stage          values stay in [0, 8q)
//   MODE 1  q <= 2^60: cond-subtract 8q before every second stage   values stay in [0, 16q)
//   MODE 2  q < 2^57 : never (at most 4*logN+1 <= 65 multiples of q accumulate < 2^64)
//   MODE 3  any mix of sizes below 2^61: as MODE 0 with plain 64-bit compares and Barrett reductions
//   (MODE 0/1 additionally assume q >= 2^57 for their one-multiply quotient estimates)
// Inverse (Gentleman-Sande): X = U + V doubles the bound, so it is corrected every stage:
// values stay in [0, 4q).
// ------------------------------------------------------------------------------------------
// x in [0, 2m) -> x mod m, for 2^59 <= m <= 2^63: after d = x - m the two candidates differ in their
// high words (m >= 2^32) and d wraps above 2^63 when x < m, so one 32-bit compare of the high words
// decides (add + cmp + 2 cndmask instead of the compiler's cmp64 + 2 cndmask + sub + subb).
LR_D u64 csub_hi(u64 x, u64 m) {
    const u64 d = x - m;
    return (u32)(d >> 32) < (u32)(x >> 32) ? d : x;
}

// floor(x / q) under-estimated by at most 1, for any 64-bit x and 2^57 <= q < 2^61
LR_D u32 est_quotient(u64 x, const LimbParams &lp) { return __umulhi((u32)(x >> 32), lp.red_m) >> lp.red_g; }

// x - k*q with k < 2^8
LR_D u64 sub_kq(u64 x, u32 k, u64 q) {
    u64 kq = (u64)k * (u32)q;
    kq += (u64)(k * (u32)(q >> 32)) << 32;
    return x - kq;
}

template <int MODE, bool LOWREG = false>
LR_D void fwd_bfly(u64 &U, u64 &V, u64 w, u64 ws, u64 q, u64 q4, bool correct) {
    u64 u = U;
    if (MODE == 0) {
        u = csub_hi(u, q4);                        // [0,8q) -> [0,4q)
    } else if (MODE == 1) {
        if (correct) u = csub_hi(u, q4 << 1);      // [0,16q) -> [0,8q)
    } else if (MODE == 3) {
        u = u >= q4 ? u - q4 : u;                  // any q < 2^61 (mixed-size contexts)
    }
    const u64 v = LOWREG ? mul_shoup_lazy_lowreg(V, w, ws, q) : mul_shoup_lazy(V, w, ws, q);  // [0,4q) for any 64-bit V
    U = u + v;
    V = u + q4 - v;
}

template <bool LOWREG = false, bool BIGQ = false>
LR_D void inv_bfly(u64 &U, u64 &V, u64 w, u64 ws, u64 q, u64 q4) {
    const u64 s = U + V;                           // [0,8q)
    const u64 t = U + q4 - V;                      // (0,8q)
    U = BIGQ ? csub_hi(s, q4) : (s >= q4 ? s - q4 : s);   // [0,4q); csub_hi needs 2^32 <= 4q < 2^63
    V = LOWREG ? mul_shoup_lazy_lowreg(t, w, ws, q) : mul_shoup_lazy(t, w, ws, q);  // [0,4q)
}

LR_D u64 canon_from_4q(u64 x, u64 q) {
    const u64 q2 = q << 1;
    x = x >= q2 ? x - q2 : x;
    return x >= q ? x - q : x;
}

// canonical representative after the last forward stage
template <int MODE>
LR_D u64 fwd_canon(u64 x, const LimbParams &lp) {
    const u64 q = lp.q;
    if (MODE == 2) return bred_add(x, q, lp.bred_hi);   // exact for any 64-bit x
    if (MODE == 3) {
        const u64 q4 = q << 2;
        x = x >= q4 ? x - q4 : x;
        return canon_from_4q(x, q);
    }
    // q >= 2^57: one multiply estimates the quotient to within 1, then a single conditional subtract
    const u64 r = sub_kq(x, est_quotient(x, lp), q);     // [0, 2q)
    const u64 d = r - q;
    return (long long)d < 0 ? r : d;
}

// first-stage U operands: any 64-bit value -> congruent value the lazy invariant accepts
template <int MODE>
LR_D u64 fwd_input(u64 x, const LimbParams &lp) {
    if (MODE == 2 || MODE == 3) return bred_add_constant(x, lp.q, lp.bred_hi);   // [0, 2q)
    return sub_kq(x, est_quotient(x, lp), lp.q);                   // [0, 2q)
}

// R stages over the R index bits [plo, plo+R) held in registers: x[k] has bit pattern k there.
// H = 2^(logN - plo - R) + (index bits above plo+R): the heap position of the block's twiddle;
// the stage over bit plo+b uses twiddles (H << (R-1-b)) + j, j = k >> (b+1).
// S0 = global index of the first of these stages (stage s consumes index bit logN-1-s); in
// MODE 1 the even stages s >= 2 correct.  FENCE keeps the compiler from hoisting every
// stage's per-lane twiddle loads to the top (register pressure).
// FIN: twiddles come from the lane-transposed table of the last four stages,
// entry [slot = 2^c - 1 + j][block] with fin_stride = N/16 blocks and H = the block index:
// consecutive lanes read consecutive 16-byte entries.
template <int R, int MODE, int S0, bool FENCE, int GROUP = 0, bool FIN = false, bool LOWREG = false>
LR_D void fwd_stages(u64 (&x)[1 << R], const Twiddle *__restrict__ tw, u32 H, u64 q, u64 q4, int fin_stride = 0) {
#pragma unroll
    for (int b = R - 1; b >= 0; --b) {
        const int c = R - 1 - b;
        const int s = S0 + c;
        const bool correct = (s >= 2) && ((s & 1) == 0);
#pragma unroll
        for (int j = 0; j < (1 << c); ++j) {
            const Twiddle w = FIN ? tw[((1 << c) - 1 + j) * fin_stride + H] : tw[(H << c) + j];
#pragma unroll
            for (int i = 0; i < (1 << b); ++i) {
                const int k0 = (j << (b + 1)) | i;
                fwd_bfly<MODE, LOWREG>(x[k0], x[k0 | (1 << b)], w.x, w.y, q, q4, correct);
                // GROUP > 0: let at most GROUP butterflies interleave (bounds the live temporaries)
                if constexpr (GROUP > 0) {
                    if (((j * (1 << b) + i + 1) % GROUP) == 0) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (FENCE && b > 0) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int R, bool FIN = false, bool LOWREG = false, bool BIGQ = false>
LR_D void inv_stages(u64 (&x)[1 << R], const Twiddle *__restrict__ tw, u32 H, u64 q, u64 q4, int fin_stride = 0) {
#pragma unroll
    for (int b = 0; b < R; ++b) {
        const int c = R - 1 - b;
#pragma unroll
        for (int j = 0; j < (1 << c); ++j) {
            const Twiddle w = FIN ? tw[((1 << c) - 1 + j) * fin_stride + H] : tw[(H << c) + j];
#pragma unroll
            for (int i = 0; i < (1 << b); ++i) {
                const int k0 = (j << (b + 1)) | i;
                inv_bfly<LOWREG, BIGQ>(x[k0], x[k0 | (1 << b)], w.x, w.y, q, q4);
            }
        }
    }
}

// LDS image: 16 B of padding after every 16 coefficients (144-B rows)
LR_D int lds_slot(int i) { return i + ((i >> 4) << 1); }
constexpr int lds_words(int m) { return m + ((m >> 4) << 1); }

// per-size plan: LOGT = log2(blockDim), A = bits consumed by pass A, HALVES = LDS residency
// rounds, then the LDS passes (forward order, high bits first) consuming the remaining bits.
template <int LOGN> struct Plan;
template <> struct Plan<15> { static constexpr int LOGT = 10, A = 5, HALVES = 2, P0 = 3, P1 = 3, P2 = 4; };
template <> struct Plan<14> { static constexpr int LOGT = 10, A = 4, HALVES = 1, P0 = 3, P1 = 3, P2 = 4; };
template <> struct Plan<13> { static constexpr int LOGT = 9,  A = 4, HALVES = 1, P0 = 3, P1 = 2, P2 = 4; };
template <> struct Plan<12> { static constexpr int LOGT = 8,  A = 4, HALVES = 1, P0 = 0, P1 = 4, P2 = 4; };

// one forward LDS pass over bits [PLO, PLO+R) of the M resident coefficients
// A workgroup may transform a 2^LOGN sub-block of a longer transform (N = 2^16 runs its top stage in a
// separate kernel): hroot = 2^d + i is the heap position of sub-block i at depth d (1 for a whole limb),
// fin_stride/fin_block0 locate the sub-block in the lane-transposed table of the full transform.
struct Sub {
    u32 hroot;
    int fin_stride;
    int fin_block0;
};

template <int LOGN, int M, int T, int R, int PLO, int MODE, int GROUP = 0, bool LOWREG = false>
LR_D void fwd_lds_pass(u64 *lds, const Twiddle *__restrict__ tw, const Twiddle *__restrict__ tw_fin, int res_base, int t,
                       u64 q, u64 q4, Sub sub) {
    if constexpr (R > 0) {
        constexpr int NT = M >> R;
#pragma unroll
        for (int u = t; u < NT; u += T) {
            const int u_lo = u & ((1 << PLO) - 1), u_hi = u >> PLO;
            const int base = (u_hi << (PLO + R)) | u_lo;
            u32 H = (sub.hroot << (LOGN - PLO - R)) + (u32)(res_base >> (PLO + R)) + (u32)u_hi;
            if constexpr (PLO >= 6) H = __builtin_amdgcn_readfirstlane(H);  // wave-uniform: scalar twiddle loads
            u64 y[1 << R];
            // the padded image is linear in k: slot(base + k*2^PLO) = slot(base) + k*(2^PLO + 2^(PLO-3)) for PLO >= 4
            u64 *const row = lds + lds_slot(base);
            constexpr int KSTRIDE = PLO >= 4 ? (1 << PLO) + (1 << (PLO - 3)) : 0;
            static_assert(PLO == 0 || PLO >= 4, "pass layout");
            if constexpr (PLO == 0) {
                const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(row);
#pragma unroll
                for (int k = 0; k < (1 << R); k += 2) {
                    const ulonglong2 v = p[(k >> 1) + (k >> 4)];
                    y[k] = v.x;
                    y[k + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < (1 << R); ++k) y[k] = row[k * KSTRIDE];
            }
            if constexpr (PLO == 0 && R == 4)
                fwd_stages<R, MODE, LOGN - PLO - R, true, GROUP, true, LOWREG>(y, tw_fin, (u32)(sub.fin_block0 + (res_base >> 4) + u), q, q4, sub.fin_stride);
            else
                fwd_stages<R, MODE, LOGN - PLO - R, (PLO < 6 && R >= 4), GROUP, false, LOWREG>(y, tw, H, q, q4);
            if constexpr (PLO == 0) {
                ulonglong2 *p = reinterpret_cast<ulonglong2 *>(row);
#pragma unroll
                for (int k = 0; k < (1 << R); k += 2) p[(k >> 1) + (k >> 4)] = make_ulonglong2(y[k], y[k + 1]);
            } else {
#pragma unroll
                for (int k = 0; k < (1 << R); ++k) row[k * KSTRIDE] = y[k];
            }
        }
        __syncthreads();
    }
}

template <int LOGN, int M, int T, int R, int PLO, bool LOWREG = false, bool BIGQ = false>
LR_D void inv_lds_pass(u64 *lds, const Twiddle *__restrict__ tw, const Twiddle *__restrict__ tw_fin, int res_base, int t,
                       u64 q, u64 q4, Sub sub) {
    if constexpr (R > 0) {
        constexpr int NT = M >> R;
#pragma unroll
        for (int u = t; u < NT; u += T) {
            const int u_lo = u & ((1 << PLO) - 1), u_hi = u >> PLO;
            const int base = (u_hi << (PLO + R)) | u_lo;
            u32 H = (sub.hroot << (LOGN - PLO - R)) + (u32)(res_base >> (PLO + R)) + (u32)u_hi;
            if constexpr (PLO >= 6) H = __builtin_amdgcn_readfirstlane(H);
            u64 y[1 << R];
            // the padded image is linear in k: slot(base + k*2^PLO) = slot(base) + k*(2^PLO + 2^(PLO-3)) for PLO >= 4
            u64 *const row = lds + lds_slot(base);
            constexpr int KSTRIDE = PLO >= 4 ? (1 << PLO) + (1 << (PLO - 3)) : 0;
            static_assert(PLO == 0 || PLO >= 4, "pass layout");
            if constexpr (PLO == 0) {
                const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(row);
#pragma unroll
                for (int k = 0; k < (1 << R); k += 2) {
                    const ulonglong2 v = p[(k >> 1) + (k >> 4)];
                    y[k] = v.x;
                    y[k + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < (1 << R); ++k) y[k] = row[k * KSTRIDE];
            }
            if constexpr (PLO == 0 && R == 4)
                inv_stages<R, true, LOWREG, BIGQ>(y, tw_fin, (u32)(sub.fin_block0 + (res_base >> 4) + u), q, q4, sub.fin_stride);
            else
                inv_stages<R, false, LOWREG, BIGQ>(y, tw, H, q, q4);
            if constexpr (PLO == 0) {
                ulonglong2 *p = reinterpret_cast<ulonglong2 *>(row);
#pragma unroll
                for (int k = 0; k < (1 << R); k += 2) p[(k >> 1) + (k >> 4)] = make_ulonglong2(y[k], y[k + 1]);
            } else {
#pragma unroll
                for (int k = 0; k < (1 << R); ++k) row[k * KSTRIDE] = y[k];
            }
        }
        __syncthreads();
    }
}

struct Item {
    const u64 *src;
    u64 *dst;
    const Twiddle *tw;
    const Twiddle *tw_fin;
    Sub sub;
    LimbParams lp;
};

// n = coefficients per workgroup; the launch carries sub_log = log2(sub-blocks per limb) (0 except N = 2^16)
LR_D Item locate(const NttLaunch &a, int n) {
    const int blocks_per_limb = 1 << a.sub_log;
    const int wg = blockIdx.x % (a.n_items * blocks_per_limb), b = blockIdx.x / (a.n_items * blocks_per_limb);
    int item = wg >> a.sub_log;
    const int blk = wg & (blocks_per_limb - 1);
    if (a.hole > 0 && item >= (b / a.group) * a.hole) item += a.hole;
    const int mod = a.mod0 + item * a.mod_step;
    const long long n_full = (long long)n << a.sub_log;
    Item it;
    it.src = a.in + (long long)b * a.in_poly_stride + (long long)(a.in_limb0 + item * a.in_limb_step) * n_full + (long long)blk * n;
    it.dst = a.out + (long long)b * a.out_poly_stride + (long long)(a.out_limb0 + item * a.out_limb_step) * n_full + (long long)blk * n;
    it.tw = a.tw + (long long)mod * n_full;
    it.tw_fin = a.tw_fin ? a.tw_fin + (long long)mod * (15 * (n_full >> 4)) : nullptr;
    it.lp = a.lp[mod];
    it.sub.hroot = (u32)(blocks_per_limb + blk);
    it.sub.fin_stride = (int)(n_full >> 4);
    it.sub.fin_block0 = blk * (n >> 4);
    return it;
}

// ------------------------------------------------------------------------------------------
// forward, 2^12 <= N <= 2^15
// ------------------------------------------------------------------------------------------
template <int LOGN, int MODE>
__global__ __launch_bounds__(1 << Plan<LOGN>::LOGT, 4) void ntt_fwd_kernel(NttLaunch a) {
    using P = Plan<LOGN>;
    constexpr int N = 1 << LOGN, T = 1 << P::LOGT, A = P::A, RA = 1 << A, S = N >> A;
    constexpr int HALVES = P::HALVES, M = N / HALVES, RH = RA / HALVES;
    static_assert(S == T, "one column per thread");
    static_assert(A + P::P0 + P::P1 + P::P2 == LOGN, "bit budget");
    extern __shared__ __align__(16) u64 lds[];

    const Item it = locate(a, N);
    const u64 q = it.lp.q, q4 = q << 2;
    const int t = threadIdx.x;

    u64 x[RA];
    if (LOGN == 15 && a.sub_log == 1 && a.fuse_top) {
        // N = 2^16, out of place: sub-block 0 continues with X = U + V*psi[1], sub-block 1 with Y = U - V*psi[1] of the
        // stage over bit 15 (U from the low half of the limb, V from the high half); both blocks read both halves
        const int blk = (int)it.sub.hroot - 2;
        const u64 *lo = it.src - (long long)blk * N, *hi = lo + N;
        const Twiddle w1 = it.tw[1];
#pragma unroll
        for (int k = 0; k < RA; ++k) {
            const u64 U = bred_add(ld_stream(lo + k * S + t), q, it.lp.bred_hi);
            const u64 r = mul_shoup_lazy(ld_stream(hi + k * S + t), w1.x, w1.y, q);   // [0,4q) for any 64-bit V
            x[k] = blk ? U + q4 - r : U + r;                                             // < 5q
        }
    } else {
#pragma unroll
        for (int k = 0; k < RA; ++k) x[k] = ld_stream(it.src + k * S + t);   // uniform row base (SGPR) + one lane offset
    }
    // The first stage needs U < 8q; V may be any 64-bit value.  The reference accepts inputs
    // >= q (ring/ring_scaling.go:19,102), so the U operands are reduced exactly.
#pragma unroll
    for (int k = 0; k < RA / 2; ++k) x[k] = fwd_input<MODE>(x[k], it.lp);
    fwd_stages<A, MODE, 0, false, (HALVES > 1 ? 4 : 0), false, (HALVES > 1)>(x, it.tw, it.sub.hroot, q, q4);

#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        {
            u64 *const col = lds + lds_slot(t);
#pragma unroll
            for (int kk = 0; kk < RH; ++kk) col[kk * (S + S / 8)] = x[half * RH + kk];
        }
        __syncthreads();
        const int res_base = half * M;
        fwd_lds_pass<LOGN, M, T, P::P0, P::P1 + P::P2, MODE, 0, (HALVES > 1)>(lds, it.tw, it.tw_fin, res_base, t, q, q4, it.sub);
        fwd_lds_pass<LOGN, M, T, P::P1, P::P2, MODE, 0, (HALVES > 1)>(lds, it.tw, it.tw_fin, res_base, t, q, q4, it.sub);
        fwd_lds_pass<LOGN, M, T, P::P2, 0, MODE, (HALVES > 1 ? 4 : 0), (HALVES > 1)>(lds, it.tw, it.tw_fin, res_base, t, q, q4, it.sub);
        // copy-out: canonical reduction + coalesced 16-B stores
        ulonglong2 *dst2 = reinterpret_cast<ulonglong2 *>(it.dst + res_base);
        const u64 *const pair = lds + lds_slot(2 * t);
#pragma unroll
        for (int i = 0; i < M / (2 * T); ++i) {   // element pair e = t + i*T; slot(2e) = slot(2t) + i*(2T + 2T/8)
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(pair + i * (2 * T + T / 4));
            st_stream(dst2 + i * T + t, make_ulonglong2(fwd_canon<MODE>(v.x, it.lp), fwd_canon<MODE>(v.y, it.lp)));
        }
        if (half + 1 < HALVES) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// inverse, 2^12 <= N <= 2^15.  Inputs must be < 4q (the reference's own InvButterfly,
// ring/ntt.go:43-50, is only congruence-preserving for inputs <= 2q: U+2Q-V must not wrap).
// ------------------------------------------------------------------------------------------
template <int LOGN, bool BIGQ>
__global__ __launch_bounds__(1 << Plan<LOGN>::LOGT, 4) void ntt_inv_kernel(NttLaunch a) {
    using P = Plan<LOGN>;
    constexpr int N = 1 << LOGN, T = 1 << P::LOGT, A = P::A, RA = 1 << A, S = N >> A;
    constexpr int HALVES = P::HALVES, M = N / HALVES, RH = RA / HALVES;
    extern __shared__ __align__(16) u64 lds[];

    const Item it = locate(a, N);
    const u64 q = it.lp.q, q4 = q << 2;
    const int t = threadIdx.x;

    u64 x[RA];
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        const int res_base = half * M;
        const ulonglong2 *src2 = reinterpret_cast<const ulonglong2 *>(it.src + res_base);
        u64 *const pair = lds + lds_slot(2 * t);
#pragma unroll
        for (int i = 0; i < M / (2 * T); ++i)
            *reinterpret_cast<ulonglong2 *>(pair + i * (2 * T + T / 4)) = ld_stream(src2 + i * T + t);
        __syncthreads();
        inv_lds_pass<LOGN, M, T, P::P2, 0, (HALVES > 1), BIGQ>(lds, it.tw, it.tw_fin, res_base, t, q, q4, it.sub);
        inv_lds_pass<LOGN, M, T, P::P1, P::P2, (HALVES > 1), BIGQ>(lds, it.tw, it.tw_fin, res_base, t, q, q4, it.sub);
        inv_lds_pass<LOGN, M, T, P::P0, P::P1 + P::P2, (HALVES > 1), BIGQ>(lds, it.tw, it.tw_fin, res_base, t, q, q4, it.sub);
        {
            const u64 *const col = lds + lds_slot(t);
#pragma unroll
            for (int kk = 0; kk < RH; ++kk) x[half * RH + kk] = col[kk * (S + S / 8)];
        }
        if (half + 1 < HALVES) __syncthreads();
    }
    inv_stages<A, false, (HALVES > 1), BIGQ>(x, it.tw, it.sub.hroot, q, q4);
    // MRed(x, nttNInv) of ring/ntt.go:136-138 == x * N^-1 mod q, canonical
    if (a.sub_log == 0) {
#pragma unroll
        for (int k = 0; k < RA; ++k)
            st_stream(it.dst + k * S + t, canon_from_4q(mul_shoup_lazy(x[k], it.lp.n_inv, it.lp.n_inv_shoup, q), q));
    } else {
        // sub-block of a longer transform: the last stage and the scaling follow in ntt_top_kernel; values < 4q
#pragma unroll
        for (int k = 0; k < RA; ++k) (it.dst + k * S)[t] = x[k];
    }
}

// ------------------------------------------------------------------------------------------
// small degrees (2 <= N <= 2^11): whole limb in LDS, one radix-2 stage per barrier.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ntt_small_kernel(NttLaunch a, int logn, int inverse) {
    extern __shared__ __align__(16) u64 lds[];
    const int n = 1 << logn;
    const Item it = locate(a, n);
    const u64 q = it.lp.q, q4 = q << 2;
    const int t = threadIdx.x;
    if (!inverse) {
        for (int e = t; e < n; e += 256) lds[e] = bred_add(it.src[e], q, it.lp.bred_hi);
        __syncthreads();
        for (int p = logn - 1; p >= 0; --p) {
            for (int bf = t; bf < n / 2; bf += 256) {
                const int i = bf >> p, jj = bf & ((1 << p) - 1);
                const int j = (i << (p + 1)) | jj;
                const Twiddle w = it.tw[(1 << (logn - 1 - p)) + i];
                u64 U = lds[j], V = lds[j + (1 << p)];
                fwd_bfly<3>(U, V, w.x, w.y, q, q4, true);
                lds[j] = U;
                lds[j + (1 << p)] = V;
            }
            __syncthreads();
        }
        for (int e = t; e < n; e += 256) it.dst[e] = fwd_canon<3>(lds[e], it.lp);
    } else {
        for (int e = t; e < n; e += 256) lds[e] = it.src[e];
        __syncthreads();
        for (int p = 0; p < logn; ++p) {
            for (int bf = t; bf < n / 2; bf += 256) {
                const int i = bf >> p, jj = bf & ((1 << p) - 1);
                const int j = (i << (p + 1)) | jj;
                const Twiddle w = it.tw[(1 << (logn - 1 - p)) + i];
                u64 U = lds[j], V = lds[j + (1 << p)];
                inv_bfly<false, false>(U, V, w.x, w.y, q, q4);
                lds[j] = U;
                lds[j + (1 << p)] = V;
            }
            __syncthreads();
        }
        for (int e = t; e < n; e += 256)
            it.dst[e] = canon_from_4q(mul_shoup_lazy(lds[e], it.lp.n_inv, it.lp.n_inv_shoup, q), q);
    }
}

// ------------------------------------------------------------------------------------------
// N = 2^16: the stage over index bit 15 as a streaming kernel (two passes over HBM in total).
//   forward: (x[j], x[j + N/2]) -> butterfly with psi heap entry 1, then the halves are two independent
//            2^15 transforms (sub-blocks with hroot 2 and 3) handled by ntt_fwd_kernel<15>.
//   inverse: after the two 2^15 inverse sub-transforms: last Gentleman-Sande stage, scaling by N^-1, canonical.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ntt_top_kernel(NttLaunch a, int logn, int inverse) {
    int item = blockIdx.y % a.n_items;
    const int b = blockIdx.y / a.n_items;
    if (a.hole > 0 && item >= (b / a.group) * a.hole) item += a.hole;
    const int mod = a.mod0 + item * a.mod_step;
    const long long n = 1ll << logn, h = n >> 1;
    const LimbParams lp = a.lp[mod];
    const Twiddle w = (a.tw + (long long)mod * n)[inverse ? 0 : 1];   // inverse: entry 0 = psi_inv[1] * N^-1
    const u64 q = lp.q, q4 = q << 2;
    // the inverse sub-transforms leave values below 8q (q <= 2^60) or below 4q (larger moduli)
    const u64 bound = (q >> 60) ? q4 : q4 << 1;
    const u64 *src = a.in + (long long)b * a.in_poly_stride + (long long)(a.in_limb0 + item * a.in_limb_step) * n;
    u64 *dst = a.out + (long long)b * a.out_poly_stride + (long long)(a.out_limb0 + item * a.out_limb_step) * n;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < h; j += (long long)gridDim.x * 256) {
        u64 U = ld_stream(src + j), V = ld_stream(src + j + h);
        if (!inverse) {
            U = bred_add(U, q, lp.bred_hi);
            fwd_bfly<3>(U, V, w.x, w.y, q, q4, true);                 // X, Y < 8q: the sub-transforms reduce their inputs
            dst[j] = U;
            dst[j + h] = V;
        } else {
            const u64 s = U + V, t = U + bound - V;                   // any 64-bit value is a valid multiplicand
            st_stream(dst + j, canon_from_4q(mul_shoup_lazy(s, lp.n_inv, lp.n_inv_shoup, q), q));
            st_stream(dst + j + h, canon_from_4q(mul_shoup_lazy(t, w.x, w.y, q), q));
        }
    }
}

// Rounding / flooring rescale on the sub-block kernels (small launches at N = 2^15): between the lazy inverse sub-blocks of the last limb
// (halves U, V below 8q_L) and the forward sub-blocks of every target limb sit the inverse transform's last stage with the scaling, the
// + pHalf of ring_scaling.go:87-89 and the forward transforms' first stage -- three streaming passes (ntt_top_kernel, rowadd_kernel,
// ntt_top_kernel) as one: block (x, item + n_items * poly) reads the last limb's pair (j, j + N/2), finishes it under q_L and writes the
// pair after the forward top stage under the target limb's modulus to that limb's scratch row.
// a: forward-side addressing (in = the last limb's row with in_limb_step 0, out = one row per target limb, mod0 / mod_step = targets,
// tw = the forward table); tw_inv: the inverse table; last_mod: the last limb's modulus index.
__global__ __launch_bounds__(256) void rescale_mid_kernel(NttLaunch a, const Twiddle *tw_inv, int last_mod, u64 phalf, int logn) {
    const int item = blockIdx.y % a.n_items;
    const int b = blockIdx.y / a.n_items;
    const int mod = a.mod0 + item * a.mod_step;
    const long long n = 1ll << logn, h = n >> 1;
    const LimbParams lpL = a.lp[last_mod], lp = a.lp[mod];
    const Twiddle wL = (tw_inv + (long long)last_mod * n)[0];          // psi_inv[1] * N^-1 mod q_L
    const Twiddle w = (a.tw + (long long)mod * n)[1];
    const u64 qL = lpL.q, q = lp.q, q4 = q << 2;
    const u64 bound = (qL >> 60) ? qL << 2 : qL << 3;
    const u64 *src = a.in + (long long)b * a.in_poly_stride + (long long)a.in_limb0 * n;
    u64 *dst = a.out + (long long)b * a.out_poly_stride + (long long)(a.out_limb0 + item * a.out_limb_step) * n;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < h; j += (long long)gridDim.x * 256) {
        const u64 U0 = src[j], V0 = src[j + h];                          // (read by every target limb's blocks: default cache policy)
        u64 x0 = canon_from_4q(mul_shoup_lazy(U0 + V0, lpL.n_inv, lpL.n_inv_shoup, qL), qL);
        u64 x1 = canon_from_4q(mul_shoup_lazy(U0 + bound - V0, wL.x, wL.y, qL), qL);
        x0 = cred(x0 + phalf, qL);
        x1 = cred(x1 + phalf, qL);
        u64 U = bred_add(x0, q, lp.bred_hi), V = x1;
        fwd_bfly<3>(U, V, w.x, w.y, q, q4, true);
        st_stream(dst + j, U);
        st_stream(dst + j + h, V);
    }
}

hipError_t launch_rescale_mid(const NttLaunch &a, const Twiddle *tw_inv, int last_mod, u64 phalf, int logn, hipStream_t stream) {
    if (a.n_items <= 0 || a.batch <= 0) return hipSuccess;
    const dim3 grid(32, (unsigned)(a.n_items * a.batch)), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(rescale_mid_kernel, grid, block, 0, stream, a, tw_inv, last_mod, phalf, logn);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
template <int LOGN, int MODE>
static hipError_t launch_fwd(const NttLaunch &a, hipStream_t stream) {
    using P = Plan<LOGN>;
    constexpr int M = (1 << LOGN) / P::HALVES;
    constexpr size_t lds_bytes = (size_t)lds_words(M) * sizeof(u64);
    auto fn = ntt_fwd_kernel<LOGN, MODE>;
    {
        // the dynamic-LDS limit is an attribute of the function on the CURRENT device: set once per device of the process (one process may
        // drive several devices, one host thread each)
        static std::atomic<bool> configured[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<bool> &done = configured[dev >= 0 && dev < 64 ? dev : 0];
        if (!done.load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
            done.store(true, std::memory_order_release);
        }
    }
    const dim3 grid((unsigned)((a.n_items << a.sub_log) * a.batch)), block(1u << P::LOGT);
    (void)hipGetLastError();  // drop stale (non-sticky) errors of unrelated earlier calls
    hipLaunchKernelGGL(fn, grid, block, lds_bytes, stream, a);
    return hipGetLastError();
}

template <int LOGN, bool BIGQ>
static hipError_t launch_inv(const NttLaunch &a, hipStream_t stream) {
    using P = Plan<LOGN>;
    constexpr int M = (1 << LOGN) / P::HALVES;
    constexpr size_t lds_bytes = (size_t)lds_words(M) * sizeof(u64);
    auto fn = ntt_inv_kernel<LOGN, BIGQ>;
    {
        // the dynamic-LDS limit is an attribute of the function on the CURRENT device: set once per device of the process (one process may
        // drive several devices, one host thread each)
        static std::atomic<bool> configured[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<bool> &done = configured[dev >= 0 && dev < 64 ? dev : 0];
        if (!done.load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
            done.store(true, std::memory_order_release);
        }
    }
    const dim3 grid((unsigned)((a.n_items << a.sub_log) * a.batch)), block(1u << P::LOGT);
    (void)hipGetLastError();
    hipLaunchKernelGGL(fn, grid, block, lds_bytes, stream, a);
    return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_big(const NttLaunch &a, bool inverse, int mode, hipStream_t stream) {
    // mode bit 8: every modulus is at least 2^32 (the inverse butterflies may use csub_hi)
    if (inverse) return (mode & 256) ? launch_inv<LOGN, true>(a, stream) : launch_inv<LOGN, false>(a, stream);
    switch (mode & 255) {
    case 3: return launch_fwd<LOGN, 3>(a, stream);
    case 2: return launch_fwd<LOGN, 2>(a, stream);
    case 1: return launch_fwd<LOGN, 1>(a, stream);
    default: return launch_fwd<LOGN, 0>(a, stream);
    }
}

hipError_t launch_ntt_top(const NttLaunch &a, int inverse, hipStream_t stream, int logn) {
    if (a.n_items <= 0 || a.batch <= 0) return hipSuccess;
    const dim3 tgrid(logn == 16 ? 64 : 32, (unsigned)(a.n_items * a.batch)), tblock(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(ntt_top_kernel, tgrid, tblock, 0, stream, a, logn, inverse);
    return hipGetLastError();
}

bool ntt_rows_disjoint(const NttLaunch &a, int logn) {
    const long long n_full = 1ll << logn;
    const long long last = (long long)(a.n_items - 1 + (a.hole > 0 ? a.hole : 0));
    const u64 *in_lo = a.in + (long long)a.in_limb0 * n_full;
    const u64 *in_hi = a.in + (long long)(a.batch - 1) * a.in_poly_stride + ((long long)a.in_limb0 + last * a.in_limb_step + 1) * n_full;
    const u64 *out_lo = a.out + (long long)a.out_limb0 * n_full;
    const u64 *out_hi = a.out + (long long)(a.batch - 1) * a.out_poly_stride + ((long long)a.out_limb0 + last * a.out_limb_step + 1) * n_full;
    const bool positive = a.in_poly_stride >= 0 && a.out_poly_stride >= 0 && a.in_limb_step >= 0 && a.out_limb_step >= 0;
    if (!positive) return false;
    if (in_hi <= out_lo || out_hi <= in_lo) return true;
    // rows of the same polys (Rescale: the last limb's row into the rows below it): compare the row sets inside one poly
    if (a.in != a.out || a.in_poly_stride != a.out_poly_stride || a.n_items > 256) return false;
    const long long top_row = std::max((long long)a.in_limb0 + last * a.in_limb_step, (long long)a.out_limb0 + last * a.out_limb_step);
    if (a.batch > 1 && a.in_poly_stride < (top_row + 1) * n_full) return false;
    for (long long i = 0; i <= last; ++i)
        for (long long j = 0; j <= last; ++j)
            if (a.in_limb0 + i * a.in_limb_step == a.out_limb0 + j * a.out_limb_step) return false;
    return true;
}

hipError_t launch_ntt(const NttLaunch &a, int logn, bool inverse, int mode, hipStream_t stream) {
    if (a.n_items <= 0 || a.batch <= 0) return hipSuccess;
    switch (logn) {
    case 15: return launch_big<15>(a, inverse, mode, stream);
    case 14: return launch_big<14>(a, inverse, mode, stream);
    case 13: return launch_big<13>(a, inverse, mode, stream);
    case 12: return launch_big<12>(a, inverse, mode, stream);
    default: break;
    }
    if (logn == 16) {
        // two kernels: the stage over bit 15 streams through HBM, the rest runs as 2^15 sub-transforms
        NttLaunch top = a, sub = a;
        const dim3 tgrid(64, (unsigned)(a.n_items * a.batch)), tblock(256);
        sub.sub_log = 1;
        if (!inverse) {
            // out of place (no byte of the input rows is an output row): one pass, the sub-transforms do the top stage
            if (ntt_rows_disjoint(a, 16)) {
                sub.fuse_top = 1;
                return launch_big<15>(sub, false, mode, stream);
            }
            (void)hipGetLastError();
            hipLaunchKernelGGL(ntt_top_kernel, tgrid, tblock, 0, stream, top, 16, 0);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            sub.in = a.out;                      // continue in place on the output rows
            sub.in_poly_stride = a.out_poly_stride;
            sub.in_limb0 = a.out_limb0;
            sub.in_limb_step = a.out_limb_step;
            return launch_big<15>(sub, false, mode, stream);
        }
        hipError_t e = launch_big<15>(sub, true, mode, stream);
        if (e != hipSuccess) return e;
        top.in = a.out;
        top.in_poly_stride = a.out_poly_stride;
        top.in_limb0 = a.out_limb0;
        top.in_limb_step = a.out_limb_step;
        (void)hipGetLastError();
        hipLaunchKernelGGL(ntt_top_kernel, tgrid, tblock, 0, stream, top, 16, 1);
        return hipGetLastError();
    }
    if (logn >= 1 && logn <= 11) {
        const dim3 grid((unsigned)(a.n_items * a.batch)), block(256);
        (void)hipGetLastError();  // drop stale (non-sticky) errors of unrelated earlier calls
    hipLaunchKernelGGL(ntt_small_kernel, grid, block, sizeof(u64) << logn, stream, a, logn, inverse ? 1 : 0);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

}  // namespace lr
