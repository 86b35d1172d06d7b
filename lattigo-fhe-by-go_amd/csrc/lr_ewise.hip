// lr_ewise.hip -- the coefficient-wise family of ring/ring.go as one templated streaming kernel.
//
// HBM-bound (16-24 B per coefficient): 16 B per lane per access, limb index on blockIdx.y so the
// per-modulus constants are wave-uniform (SGPRs), batch on blockIdx.z.  An operand whose batch
// is 1 is broadcast (poly stride 0).
#include <atomic>

#include "lattigo_ring.h"
#include "lr_device.hpp"

namespace lr {


template <int OP>
LR_D u64 apply(u64 x, u64 y, u64 z, const LimbParams &lp, u64 s) {
    const u64 q = lp.q;
    if constexpr (OP == LR_ADD) return cred(x + y, q);
    if constexpr (OP == LR_ADD_NOMOD) return x + y;
    if constexpr (OP == LR_SUB) return cred((x + q) - y, q);
    if constexpr (OP == LR_SUB_NOMOD) return (x + q) - y;
    if constexpr (OP == LR_NEG) return q - x;
    if constexpr (OP == LR_REDUCE) return bred_add(x, q, lp.bred_hi);
    if constexpr (OP == LR_MUL_COEFFS) return bred(x, y, q, lp.bred_hi, lp.bred_lo);
    if constexpr (OP == LR_MUL_COEFFS_AND_ADD) return cred(z + bred(x, y, q, lp.bred_hi, lp.bred_lo), q);
    if constexpr (OP == LR_MUL_COEFFS_AND_ADD_NOMOD) return z + bred(x, y, q, lp.bred_hi, lp.bred_lo);
    if constexpr (OP == LR_MUL_COEFFS_CONSTANT) return bred_constant(x, y, q, lp.bred_hi, lp.bred_lo);
    if constexpr (OP == LR_MUL_MONT) return mred(x, y, q, lp.qinv);
    if constexpr (OP == LR_MUL_MONT_AND_ADD) return cred(z + mred(x, y, q, lp.qinv), q);
    if constexpr (OP == LR_MUL_MONT_AND_ADD_NOMOD) return z + mred(x, y, q, lp.qinv);
    if constexpr (OP == LR_MUL_MONT_CONSTANT_AND_ADD_NOMOD) return z + mred_constant(x, y, q, lp.qinv);
    if constexpr (OP == LR_MUL_MONT_AND_SUB) return cred(z + (q - mred(x, y, q, lp.qinv)), q);
    if constexpr (OP == LR_MUL_MONT_AND_SUB_NOMOD) return z + (q - mred(x, y, q, lp.qinv));
    if constexpr (OP == LR_MUL_MONT_CONSTANT) return mred_constant(x, y, q, lp.qinv);
    if constexpr (OP == LR_MFORM) return mform(x, q, lp.bred_hi, lp.bred_lo);
    if constexpr (OP == LR_INV_MFORM) return inv_mform(x, q, lp.qinv);
    if constexpr (OP == LR_MUL_SCALAR || OP == LR_MUL_SCALAR_LIMBS) return mred(x, s, q, lp.qinv);
    if constexpr (OP == LR_ADD_SCALAR_LIMBS) return cred(x + s, q);
    if constexpr (OP == LR_SUB_SCALAR_LIMBS) return cred(x + (q - s), q);
    if constexpr (OP == LR_COPY) return x;
    if constexpr (OP == LR_MUL_BY_POW2) return power_of_2(x, s, q, lp.qinv);
    return 0;
}

constexpr bool reads_b(int op) {
    return op == LR_ADD || op == LR_ADD_NOMOD || op == LR_SUB || op == LR_SUB_NOMOD ||
           (op >= LR_MUL_COEFFS && op <= LR_MUL_MONT_CONSTANT);
}
constexpr bool reads_out(int op) {
    return op == LR_MUL_COEFFS_AND_ADD || op == LR_MUL_COEFFS_AND_ADD_NOMOD || op == LR_MUL_MONT_AND_ADD ||
           op == LR_MUL_MONT_AND_ADD_NOMOD || op == LR_MUL_MONT_CONSTANT_AND_ADD_NOMOD ||
           op == LR_MUL_MONT_AND_SUB || op == LR_MUL_MONT_AND_SUB_NOMOD;
}

// grid: x = coefficient pairs / 256, y = limb, z = batch
template <int OP>
__global__ __launch_bounds__(256) void ewise_kernel(EwiseLaunch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const LimbParams lp = L.lp[limb];
    const u64 s = L.has_scalars ? L.scalars.v[limb] : 0;
    const long long row = (long long)limb * L.n;
    const ulonglong2 *pa = reinterpret_cast<const ulonglong2 *>(L.a + b * L.a_stride + row);
    const ulonglong2 *pb = reads_b(OP) ? reinterpret_cast<const ulonglong2 *>(L.b + b * L.b_stride + row) : nullptr;
    ulonglong2 *po = reinterpret_cast<ulonglong2 *>(L.out + b * L.out_stride + row);
    const int pairs = L.n >> 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) {
        const ulonglong2 x = ld_stream(pa + e);
        ulonglong2 y = make_ulonglong2(0, 0), z = make_ulonglong2(0, 0);
        if constexpr (reads_b(OP)) y = ld_stream(pb + e);
        if constexpr (reads_out(OP)) z = ld_stream(po + e);
        st_stream(po + e, make_ulonglong2(apply<OP>(x.x, y.x, z.x, lp, s), apply<OP>(x.y, y.y, z.y, lp, s)));
    }
}

// odd degree fallback is impossible (N is a power of two >= 2)

template <int OP>
static hipError_t launch_one(const EwiseLaunch &L, int limbs, int batch, hipStream_t stream) {
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();  // drop stale (non-sticky) errors of unrelated earlier calls
    hipLaunchKernelGGL(ewise_kernel<OP>, grid, block, 0, stream, L);
    return hipGetLastError();
}

hipError_t launch_ewise(int op, const EwiseLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    switch (op) {
#define LR_CASE(OPNAME) \
    case OPNAME: return launch_one<OPNAME>(L, limbs, batch, stream);
        LR_CASE(LR_ADD)
        LR_CASE(LR_ADD_NOMOD)
        LR_CASE(LR_SUB)
        LR_CASE(LR_SUB_NOMOD)
        LR_CASE(LR_NEG)
        LR_CASE(LR_REDUCE)
        LR_CASE(LR_MUL_COEFFS)
        LR_CASE(LR_MUL_COEFFS_AND_ADD)
        LR_CASE(LR_MUL_COEFFS_AND_ADD_NOMOD)
        LR_CASE(LR_MUL_COEFFS_CONSTANT)
        LR_CASE(LR_MUL_MONT)
        LR_CASE(LR_MUL_MONT_AND_ADD)
        LR_CASE(LR_MUL_MONT_AND_ADD_NOMOD)
        LR_CASE(LR_MUL_MONT_CONSTANT_AND_ADD_NOMOD)
        LR_CASE(LR_MUL_MONT_AND_SUB)
        LR_CASE(LR_MUL_MONT_AND_SUB_NOMOD)
        LR_CASE(LR_MUL_MONT_CONSTANT)
        LR_CASE(LR_MFORM)
        LR_CASE(LR_INV_MFORM)
        LR_CASE(LR_MUL_SCALAR)
        LR_CASE(LR_MUL_SCALAR_LIMBS)
        LR_CASE(LR_ADD_SCALAR_LIMBS)
        LR_CASE(LR_SUB_SCALAR_LIMBS)
        LR_CASE(LR_COPY)
        LR_CASE(LR_MUL_BY_POW2)
#undef LR_CASE
    default: return hipErrorInvalidValue;
    }
}

// ---- fused tails used by ModDown / rescale: out = MRed(a + (q - b), c[limb]) --------------
// ring/ring_basis_extension.go:196-199,237-239,270-272 and ring/ring_scaling.go:28-30,108-110

__global__ __launch_bounds__(256) void submul_kernel(SubMulLaunch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const LimbParams lp = L.lp[limb];
    const u64 c = L.consts[limb];
    const u64 add = L.addend.v[limb];
    const long long row = (long long)limb * L.n;
    const ulonglong2 *pa = reinterpret_cast<const ulonglong2 *>(L.a + b * L.a_stride + row);
    const ulonglong2 *pb = reinterpret_cast<const ulonglong2 *>(L.b + b * L.b_stride + (long long)limb * L.b_row_stride);
    ulonglong2 *po = reinterpret_cast<ulonglong2 *>(L.out + b * L.out_stride + row);
    const int pairs = L.n >> 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) {
        const ulonglong2 x = ld_stream(pa + e);
        ulonglong2 y = ld_stream(pb + e);
        if (L.reduce_b) {
            y.x = bred_add(y.x + add, lp.q, lp.bred_hi);
            y.y = bred_add(y.y + add, lp.q, lp.bred_hi);
        }
        ulonglong2 r = make_ulonglong2(mred(x.x + (lp.q - y.x), c, lp.q, lp.qinv), mred(x.y + (lp.q - y.y), c, lp.q, lp.qinv));
        if (L.plus) {
            const ulonglong2 p = ld_stream(reinterpret_cast<const ulonglong2 *>(L.plus + b * L.plus_stride + row) + e);
            r.x = cred(p.x + r.x, lp.q);
            r.y = cred(p.y + r.y, lp.q);
        }
        if (L.has_post) {
            const u64 s = L.post.v[limb];
            r.x = cred(r.x + s, lp.q);
            r.y = cred(r.y + s, lp.q);
        }
        st_stream(po + e, r);
    }
}

// c0 = MRed(MForm(a0), b0), c1 = MRed(MForm(a0), b1) (+)= MRed(MForm(a1), b0), c2 = MRed(MForm(a1), b1):
// the six Context calls of ckks/evaluator.go:1080-1095 with the two Montgomery-form temporaries kept in
// registers (56 B per coefficient instead of 128).
__global__ __launch_bounds__(256) void tensor_kernel(TensorLaunch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const LimbParams lp = L.lp[limb];
    const u64 q = lp.q;
    const long long row = (long long)limb * L.n;
    const u64 *const *tb = L.table ? L.table + 4 * b : nullptr;
    const ulonglong2 *pa0 = reinterpret_cast<const ulonglong2 *>((tb ? tb[0] : L.a0 + b * L.a0_stride) + row);
    const ulonglong2 *pa1 = reinterpret_cast<const ulonglong2 *>((tb ? tb[1] : L.a1 + b * L.a1_stride) + row);
    const ulonglong2 *pb0 = reinterpret_cast<const ulonglong2 *>((tb ? tb[2] : L.b0 + b * L.b0_stride) + row);
    const ulonglong2 *pb1 = reinterpret_cast<const ulonglong2 *>((tb ? tb[3] : L.b1 + b * L.b1_stride) + row);
    ulonglong2 *pc0 = reinterpret_cast<ulonglong2 *>(L.c0 + b * L.c_stride + row);
    ulonglong2 *pc1 = reinterpret_cast<ulonglong2 *>(L.c1 + b * L.c1_stride + row);
    ulonglong2 *pc2 = reinterpret_cast<ulonglong2 *>(L.c2 + b * L.c2_stride + row);
    const int pairs = L.n >> 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) {
        // (the squaring case -- ct0 == ct1, ckks/evaluator.go:1083, bfv/evaluator.go:334 -- reads its one operand once: block-uniform)
        const ulonglong2 a0 = ld_stream(pa0 + e), a1 = ld_stream(pa1 + e);
        const ulonglong2 b0 = pb0 == pa0 ? a0 : ld_stream(pb0 + e), b1 = pb1 == pa1 ? a1 : ld_stream(pb1 + e);
        ulonglong2 c0, c1, c2;
        {
            const u64 m0 = mform(a0.x, q, lp.bred_hi, lp.bred_lo), m1 = mform(a1.x, q, lp.bred_hi, lp.bred_lo);
            c0.x = mred(m0, b0.x, q, lp.qinv);
            c1.x = cred(mred(m0, b1.x, q, lp.qinv) + mred(m1, b0.x, q, lp.qinv), q);
            c2.x = mred(m1, b1.x, q, lp.qinv);
        }
        {
            const u64 m0 = mform(a0.y, q, lp.bred_hi, lp.bred_lo), m1 = mform(a1.y, q, lp.bred_hi, lp.bred_lo);
            c0.y = mred(m0, b0.y, q, lp.qinv);
            c1.y = cred(mred(m0, b1.y, q, lp.qinv) + mred(m1, b0.y, q, lp.qinv), q);
            c2.y = mred(m1, b1.y, q, lp.qinv);
        }
        st_stream(pc0 + e, c0);
        st_stream(pc1 + e, c1);
        st_stream(pc2 + e, c2);
    }
}

// decryptor.Decrypt, ckks/decryptor.go:61-77 (HornerLaunch): the copy, degree x (MulCoeffsMontgomeryLvl, AddLvl), the ReduceLvl cadence
__global__ __launch_bounds__(256) void horner_kernel(HornerLaunch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const LimbParams lp = L.lp[limb];
    const u64 q = lp.q;
    const long long row = (long long)limb * L.n;
    const ulonglong2 *ps = reinterpret_cast<const ulonglong2 *>(L.sk + b * L.sk_stride + row);
    ulonglong2 *po = reinterpret_cast<ulonglong2 *>(L.out + b * L.out_stride + row);
    const int pairs = L.n >> 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) {
        const ulonglong2 s = L.degree > 0 ? ps[e] : make_ulonglong2(0, 0);        // (the key is shared by the batch: through the caches)
        ulonglong2 acc = ld_stream(reinterpret_cast<const ulonglong2 *>(L.ct[L.degree] + b * L.ct_stride[L.degree] + row) + e);    // :61 CopyLvl
        for (int i = L.degree; i > 0; --i) {
            const ulonglong2 c = ld_stream(reinterpret_cast<const ulonglong2 *>(L.ct[i - 1] + b * L.ct_stride[i - 1] + row) + e);
            acc.x = cred(mred(acc.x, s.x, q, lp.qinv) + c.x, q);                 // :67-68
            acc.y = cred(mred(acc.y, s.y, q, lp.qinv) + c.y, q);
            if ((i & 7) == 7) {                                                  // :70-72
                acc.x = bred_add(acc.x, q, lp.bred_hi);
                acc.y = bred_add(acc.y, q, lp.bred_hi);
            }
        }
        if ((L.degree & 7) != 7) {                                               // :75-77
            acc.x = bred_add(acc.x, q, lp.bred_hi);
            acc.y = bred_add(acc.y, q, lp.bred_hi);
        }
        st_stream(po + e, acc);
    }
}

hipError_t launch_horner(const HornerLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    if (L.degree < 0 || L.degree > kHornerMaxDegree) return hipErrorInvalidValue;
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(horner_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void bswap_kernel(const u64 *in, u64 *out, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256)
        out[i] = __builtin_bswap64(in[i]);
}

hipError_t launch_bswap(const u64 *in, u64 *out, size_t words, hipStream_t stream) {
    if (words == 0) return hipSuccess;
    size_t blocks = (words + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    (void)hipGetLastError();
    hipLaunchKernelGGL(bswap_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, words);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void scalar_pair_kernel(ScalarPairLaunch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const LimbParams lp = L.lp[limb];
    const u64 q = lp.q, s1 = L.sub.v[limb], s2 = L.mul.v[limb];
    const long long row = (long long)limb * L.n;
    const ulonglong2 *pi = reinterpret_cast<const ulonglong2 *>(L.in + b * L.in_stride + row);
    ulonglong2 *po = reinterpret_cast<ulonglong2 *>(L.out + b * L.out_stride + row);
    const int pairs = L.n >> 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) {
        const ulonglong2 x = ld_stream(pi + e);
        st_stream(po + e, make_ulonglong2(mred(cred(x.x + (q - s1), q), s2, q, lp.qinv), mred(cred(x.y + (q - s1), q), s2, q, lp.qinv)));
    }
}

hipError_t launch_scalar_pair(const ScalarPairLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(scalar_pair_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

// GaloisLaunch::gen carries the shift (mod 2N)
__global__ __launch_bounds__(256) void monomial_kernel(GaloisLaunch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const u64 q = L.lp[limb].q;
    const u64 *pi = L.in + b * L.in_stride + (long long)limb * L.n;
    u64 *po = L.out + b * L.out_stride + (long long)limb * L.n;
    const int n = L.n;
    const bool flip = L.gen >= (u64)n;           // X^N = -1: the whole polynomial changes sign first (:700-707)
    const int shift = (int)(L.gen % (u64)n);
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        if (L.gen == 0) {
            po[j] = pi[j];                         // :669-678
            continue;
        }
        const int src = j < shift ? n - shift + j : j - shift;
        u64 v = pi[src];
        if (flip) v = q - v;                       // tmpx
        po[j] = j < shift ? q - v : v;             // :712-723
    }
}

hipError_t launch_monomial(const GaloisLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    int gx = (L.n + 255) / 256;
    if (gx > 64) gx = 64;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(monomial_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

// ring_scaling.go:275-300.  `a` accumulates without reduction and wraps modulo 2^64 exactly like the reference's uint64.
__global__ __launch_bounds__(256) void simple_scale_kernel(ScaleLaunch L) {
    const long long b = blockIdx.y;
    const u64 *pi = L.in + b * L.in_stride;
    u64 *po = L.out + b * L.out_stride;
    const int n = L.n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        u64 a = 0;
        F128 f{0.0, 0.0};
        for (int j = 0; j < L.limbs_in; ++j) {
            const u64 x = ld_stream(pi + (long long)j * n + i);
            const u64 w = ld_const(L.wi + j);
            if (L.pow2) {
                a += (x * w) & L.add_param;                                // :205
            } else {
                u64 hi, lo;
                mul_wide64(w, x, hi, lo);                                  // :219-229
                u64 r = hi - mul_hi64(lo * L.mul_param, L.t) + L.t;
                if (r >= L.t) r -= L.t;
                a += r;
            }
            const F128 tj{L.ti[2 * j], L.ti[2 * j + 1]};
            f = f128_add(f, f128_mul(tj, f128_set_uint64(x)));            // :290
        }
        a += f128_to_uint64(f);                                            // :293
        if (L.pow2) {
            a &= L.mul_param;                                              // :209
        } else {
            a = a - mul_hi64(a, L.add_param) * L.t;                        // :233-241
            if (a >= L.t) a -= L.t;
        }
        for (int j = 0; j < L.limbs_out; ++j) st_stream(po + (long long)j * n + i, a);
    }
}

hipError_t launch_simple_scale(const ScaleLaunch &L, int batch, hipStream_t stream) {
    if (batch <= 0 || L.limbs_out <= 0) return hipSuccess;
    int gx = (L.n + 255) / 256;
    if (gx > 1024) gx = 1024;
    const dim3 grid((unsigned)gx, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(simple_scale_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

hipError_t launch_tensor(const TensorLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(tensor_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void scatter_kernel(ScatterLaunch L) {
    const long long row = (long long)blockIdx.y * L.n;
    const long long b = blockIdx.z;
    const int pairs = L.n >> 1;
    for (int k = 0; k < L.per_poly; ++k) {
        const ulonglong2 *ps = reinterpret_cast<const ulonglong2 *>(L.src[k] + b * L.stride + row);
        ulonglong2 *pd = reinterpret_cast<ulonglong2 *>(L.table[b * L.per_poly + k] + row);
        for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) st_stream(pd + e, ld_stream(ps + e));
    }
}

hipError_t launch_scatter(const ScatterLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    if (L.per_poly < 1 || L.per_poly > 4) return hipErrorInvalidValue;
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(scatter_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

// pkEncryptor.encrypt, ckks/encryptor.go:209-211: MulCoeffsMontgomery(u, pk[0]) and (u, pk[1]) in one pass over u
__global__ __launch_bounds__(256) void mul2_kernel(Mul2Launch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const LimbParams lp = L.lp[limb];
    const long long row = (long long)limb * L.n;
    const ulonglong2 *pa = reinterpret_cast<const ulonglong2 *>(L.a + b * L.a_stride + row);
    const ulonglong2 *pb0 = reinterpret_cast<const ulonglong2 *>(L.b0 + b * L.b0_stride + row);
    const ulonglong2 *pb1 = reinterpret_cast<const ulonglong2 *>(L.b1 + b * L.b1_stride + row);
    ulonglong2 *po0 = reinterpret_cast<ulonglong2 *>(L.out0 + b * L.out0_stride + row);
    ulonglong2 *po1 = reinterpret_cast<ulonglong2 *>(L.out1 + b * L.out1_stride + row);
    const int pairs = L.n >> 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) {
        const ulonglong2 a = ld_stream(pa + e);
        const ulonglong2 k0 = L.b0_stride ? ld_stream(pb0 + e) : pb0[e], k1 = L.b1_stride ? ld_stream(pb1 + e) : pb1[e];     // (a key shared by the batch: through the caches)
        st_stream(po0 + e, make_ulonglong2(mred(a.x, k0.x, lp.q, lp.qinv), mred(a.y, k0.y, lp.q, lp.qinv)));
        st_stream(po1 + e, make_ulonglong2(mred(a.x, k1.x, lp.q, lp.qinv), mred(a.y, k1.y, lp.q, lp.qinv)));
    }
}

hipError_t launch_mul2(const Mul2Launch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(mul2_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gather_kernel(GatherLaunch L) {
    const long long row = (long long)blockIdx.y * L.n;
    const long long b = blockIdx.z;
    const int pairs = L.n >> 1;
    for (int k = 0; k < L.per_poly; ++k) {
        const ulonglong2 *ps = reinterpret_cast<const ulonglong2 *>(L.table[b * L.per_poly + k] + row);
        ulonglong2 *pd = reinterpret_cast<ulonglong2 *>(L.dst[k] + b * L.stride + row);
        for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) st_stream(pd + e, ld_stream(ps + e));
    }
}

hipError_t launch_gather(const GatherLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    if (L.per_poly < 1 || L.per_poly > 4) return hipErrorInvalidValue;
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(gather_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void multicopy_kernel(MultiCopyLaunch L) {
    const int k = blockIdx.z / L.batch;
    const long long b = blockIdx.z % L.batch;
    const long long row = (long long)blockIdx.y * L.n;
    const ulonglong2 *ps = reinterpret_cast<const ulonglong2 *>(L.src[k] + b * L.src_stride[k] + row);
    ulonglong2 *pd = reinterpret_cast<ulonglong2 *>(L.dst[k] + b * L.dst_stride[k] + row);
    const int pairs = L.n >> 1;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) st_stream(pd + e, ld_stream(ps + e));
}

hipError_t launch_multicopy(const MultiCopyLaunch &L, int limbs, hipStream_t stream) {
    if (limbs <= 0 || L.batch <= 0 || L.count <= 0) return hipSuccess;
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)(L.count * L.batch)), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(multicopy_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

hipError_t launch_submul(const SubMulLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    const int pairs = L.n >> 1;
    int gx = (pairs + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();  // drop stale (non-sticky) errors of unrelated earlier calls
    hipLaunchKernelGGL(submul_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

// ---- single-limb helpers for rescale ---------------------------------------------------------
// out[j] = CRed(in[j] + add, q) on one row (ring_scaling.go:87-89,127-129), or
// out_row_i[j] = in[j] + add_i (no reduction; ring_scaling.go:101-103) for i < limbs

__global__ __launch_bounds__(256) void rowadd_kernel(RowAddLaunch L) {
    const int row = blockIdx.y;
    const long long b = blockIdx.z;
    const u64 add = L.adds.v[row];
    const u64 *pi = L.in + b * L.in_stride;
    u64 *po = L.out + b * L.out_stride + (long long)row * L.n;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < L.n; e += gridDim.x * 256) {
        u64 v = pi[e] + add;
        if (L.q) v = cred(v, L.q);
        po[e] = v;
    }
}

hipError_t launch_rowadd(const RowAddLaunch &L, int rows, int batch, hipStream_t stream) {
    if (rows <= 0 || batch <= 0) return hipSuccess;
    int gx = (L.n + 255) / 256;
    if (gx > 64) gx = 64;
    const dim3 grid((unsigned)gx, (unsigned)rows, (unsigned)batch), block(256);
    (void)hipGetLastError();  // drop stale (non-sticky) errors of unrelated earlier calls
    hipLaunchKernelGGL(rowadd_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

// ---- half-vector scalar operations: AddConst / MultByConst / MultByConstAndAdd / MultByi / DivByi of ckks.Evaluator ------------
// (ckks/evaluator.go:429-445, 588-606, 712-730, 765-779, 814-828): per limb one scalar for the coefficients below n/2 and one for the
// rest, the reference's exact element operation -- CRed(x + s), MRed(x, s), CRed(y + MRed(x, s)) -- on values that need not be canonical
__global__ __launch_bounds__(256) void half_scalar_kernel(HalfScalarLaunch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const LimbParams lp = L.lp[limb];
    const u64 slo = L.lo.v[limb], shi = L.hi.v[limb];
    const ulonglong2 *pi = reinterpret_cast<const ulonglong2 *>(L.in + b * L.in_stride + (long long)limb * L.n);
    ulonglong2 *po = reinterpret_cast<ulonglong2 *>(L.out + b * L.out_stride + (long long)limb * L.n);
    const int pairs = L.n >> 1, half = L.n >> 2;       // n / 2 coefficients = n / 4 pairs per half
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pairs; e += gridDim.x * 256) {
        const u64 sc = e < half ? slo : shi;
        const ulonglong2 x = ld_stream(pi + e);
        ulonglong2 r;
        if (L.op == 0) {
            r = make_ulonglong2(cred(x.x + sc, lp.q), cred(x.y + sc, lp.q));
        } else if (L.op == 1) {
            r = make_ulonglong2(mred(x.x, sc, lp.q, lp.qinv), mred(x.y, sc, lp.q, lp.qinv));
        } else {
            const ulonglong2 y = ld_stream(po + e);
            r = make_ulonglong2(cred(y.x + mred(x.x, sc, lp.q, lp.qinv), lp.q), cred(y.y + mred(x.y, sc, lp.q, lp.qinv), lp.q));
        }
        st_stream(po + e, r);
    }
}

hipError_t launch_half_scalar(const HalfScalarLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    int gx = ((L.n >> 1) + 255) / 256;
    if (gx > 64) gx = 64;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(half_scalar_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

// ---- hybrid key-switch inner product (ckks/evaluator.go:1511-1552) ---------------------------------
// out0 = sum_i key[i][0] (*) c2[i],  out1 = sum_i key[i][1] (*) c2[i]  over the beta digits, Montgomery products,
// canonical result.  The reference accumulates digit by digit with MulCoeffsMontgomeryAndAddNoMod and lazy
// Reduce calls (reduce&7 cadence); the sums are below 8q < 2^64 between reductions either way and the final
// Reduce makes the result canonical, so one pass over all digits yields the same values while the accumulators
// never travel through HBM.
// BETA > 0: the digit count is known at compile time, so all of a coefficient pair's digit and key loads are issued
// before the first multiply (memory-level parallelism instead of one load round trip per digit); BETA = 0: generic.
// ring.PermuteNTT's index (ring/ring_galois.go:29-50): out[j] = in[perm_index(j)], computed on the fly (two bit reversals)
__device__ __forceinline__ u32 perm_index(u32 j, u32 gen, int shift, u32 mask2) {
    const u32 t1 = 2 * (__brev(j) >> shift) + 1;
    const u32 t2 = (((gen * t1) & mask2) - 1) >> 1;
    return __brev(t2) >> shift;
}
// the digit operand of a coefficient pair: the caller's own row, the permuted digit (hoisted rotations: the Galois automorphism of the
// digits rides on the inner product's loads instead of a pass that writes permuted copies of every digit), or the plain digit
#define LR_KEYMAC_OPERAND_SETUP                                                                                                   \
    const u32 pgen = L.perm_gen;                                                                                                  \
    const int pshift = 32 - L.logn;                                                                                               \
    const u32 pmask2 = 2u * (u32)L.n - 1u;                                                                                        \
    const u64 *pc64 = reinterpret_cast<const u64 *>(pc);
#define LR_KEYMAC_OPERAND(i_)                                                                                                     \
    ((i_) == own_digit ? ld_stream(pown + e)                                                                                      \
                       : pgen ? make_ulonglong2(pc64[(long long)(i_) * L.c2_digit_stride + ix0], pc64[(long long)(i_) * L.c2_digit_stride + ix1]) \
                              : ld_stream(pc + e + (i_) * cd))

template <int BETA>
__global__ __launch_bounds__(256) void keymac_kernel(KeyMacLaunch L) {
    // x = poly of the batch (fastest): the workgroups that run together share one tile of the key, which therefore
    // stays in L2 instead of being re-read from the Infinity Cache once per poly.  tile8: x = poly * 8 + (chunk mod 8), so that
    // the workgroups of one tile have equal linear ids modulo 8 and land on ONE XCD (consecutive workgroups are dealt out to the
    // eight XCDs in turn): the tile is fetched into one L2 instead of eight -- at N = 2^16 the key (342 MB) does not fit the
    // Infinity Cache and the eight fetches were 26 % of the kernel's HBM traffic
    const int limb = blockIdx.y;
    const long long b = L.tile8 ? (long long)(blockIdx.x >> 3) : (long long)blockIdx.x;
    const int chunk = L.tile8 ? (int)(blockIdx.z * 8 + (blockIdx.x & 7)) : (int)blockIdx.z;
    const int chunks = L.tile8 ? (int)gridDim.z * 8 : (int)gridDim.z;
    const LimbParams lp = L.lp[limb];
    const long long row = (long long)limb * L.n;
    const ulonglong2 *pc = reinterpret_cast<const ulonglong2 *>(L.c2 + b * L.c2_poly_stride + row);
    const ulonglong2 *pk = reinterpret_cast<const ulonglong2 *>(L.key + (long long)(L.key_limb0 + limb) * L.n);
    ulonglong2 *po0 = reinterpret_cast<ulonglong2 *>(L.out0 + b * L.out_stride + row);
    ulonglong2 *po1 = reinterpret_cast<ulonglong2 *>(L.out1 + b * L.out1_stride + row);
    const long long cd = L.c2_digit_stride >> 1, kd = L.key_poly_stride >> 1;   // in 16-byte units
    const ulonglong2 *pown = L.alpha > 0 ? reinterpret_cast<const ulonglong2 *>(L.own + b * L.own_stride + row) : nullptr;
    const int own_digit = L.alpha > 0 ? limb / L.alpha : -1;
    const int pairs = L.n >> 1;
    const int beta = BETA > 0 ? BETA : L.beta;
    LR_KEYMAC_OPERAND_SETUP
    for (int e = chunk * 256 + threadIdx.x; e < pairs; e += chunks * 256) {
        const u32 ix0 = pgen ? perm_index(2u * (u32)e, pgen, pshift, pmask2) : 0u, ix1 = pgen ? perm_index(2u * (u32)e + 1u, pgen, pshift, pmask2) : 0u;
        u64 a0x = 0, a0y = 0, a1x = 0, a1y = 0;
        if constexpr (BETA > 0) {
            // groups of at most G digits: all loads of a group are in flight before its first multiply; beyond six digits one
            // group would need more than 100 VGPRs (4 waves per SIMD instead of 8: beta = 9 ran at 2.9 TB/s against 5.0 for beta = 6)
            constexpr int G = BETA <= 6 ? BETA : (BETA + 1) / 2;
#pragma unroll
            for (int g0 = 0; g0 < BETA; g0 += G) {
                constexpr int dummy = 0;
                (void)dummy;
                ulonglong2 c[G], k0[G], k1[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int i = g0 + u;
                    if (i < BETA) {
                        c[u] = LR_KEYMAC_OPERAND(i);
                        k0[u] = pk[e + (2 * i) * kd];
                        k1[u] = pk[e + (2 * i + 1) * kd];
                    }
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int i = g0 + u;
                    if (i < BETA) {
                        a0x += mred(k0[u].x, c[u].x, lp.q, lp.qinv);
                        a0y += mred(k0[u].y, c[u].y, lp.q, lp.qinv);
                        a1x += mred(k1[u].x, c[u].x, lp.q, lp.qinv);
                        a1y += mred(k1[u].y, c[u].y, lp.q, lp.qinv);
                        if ((i & 7) == 7) {
                            a0x = bred_add(a0x, lp.q, lp.bred_hi);
                            a0y = bred_add(a0y, lp.q, lp.bred_hi);
                            a1x = bred_add(a1x, lp.q, lp.bred_hi);
                            a1y = bred_add(a1y, lp.q, lp.bred_hi);
                        }
                    }
                }
                if (g0 + G < BETA) __builtin_amdgcn_sched_barrier(0);      // keep the next group's loads behind this group's multiplies
            }
        } else {
            for (int i = 0; i < beta; ++i) {
                const ulonglong2 c = LR_KEYMAC_OPERAND(i);
                const ulonglong2 k0 = pk[e + (2 * i) * kd], k1 = pk[e + (2 * i + 1) * kd];
                a0x += mred(k0.x, c.x, lp.q, lp.qinv);
                a0y += mred(k0.y, c.y, lp.q, lp.qinv);
                a1x += mred(k1.x, c.x, lp.q, lp.qinv);
                a1y += mred(k1.y, c.y, lp.q, lp.qinv);
                if ((i & 7) == 7) {
                    a0x = bred_add(a0x, lp.q, lp.bred_hi);
                    a0y = bred_add(a0y, lp.q, lp.bred_hi);
                    a1x = bred_add(a1x, lp.q, lp.bred_hi);
                    a1y = bred_add(a1y, lp.q, lp.bred_hi);
                }
            }
        }
        st_stream(po0 + e, make_ulonglong2(bred_add(a0x, lp.q, lp.bred_hi), bred_add(a0y, lp.q, lp.bred_hi)));
        st_stream(po1 + e, make_ulonglong2(bred_add(a1x, lp.q, lp.bred_hi), bred_add(a1y, lp.q, lp.bred_hi)));
    }
}

// The same inner product with the beta products of an output summed EXACTLY in 128 bits and reduced once: sum_i MRed(k_i, c_i)
// and MRed(sum_i k_i * c_i) are the same residue modulo q, and the canonical result is what the reference's chain of
// MulCoeffsMontgomeryAndAddNoMod + Reduce leaves.  By columns, with k = k1 2^32 + k0 and c = c1 2^32 + c0:
//   lo += k0 c0 (carry-out counted), mid += k0 c1 + k1 c0 (folded every four digits: eight products below 2^61), hi += k1 c1
// -- five multiply-adds and a carry add per product instead of the ~25 instructions of a Montgomery product: at beta = 9 the
// per-term kernel was bound by its arithmetic (1.1 ms of VALU issue in a 1.85 ms launch), not by the 7 GB it moves.
// Needs every key value below q < 2^61 and beta * q < 2^64 (KeyMacLaunch::wide, set by the host) and digit values below q, which the
// transforms that produce the digits guarantee.  The digit's OWN limbs come straight from the caller (any 64-bit word is allowed
// there, like everywhere a poly enters the library): they are reduced with BRedAdd first (own_operand) -- the middle column is a
// 64-bit accumulator without a carry-out, eight products below 2^61 fit it, a product with a 32-bit high word would not.
__device__ __forceinline__ void mac128(u64 k, u64 c, u64 &lo, u64 &mid, u64 &hi, u32 &cy) {
    const u32 k0 = (u32)k, k1 = (u32)(k >> 32), c0 = (u32)c, c1 = (u32)(c >> 32);
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(cy) : "v"(k0), "v"(c0) : "vcc");
    u64 junk;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(mid), "=s"(junk) : "v"(k0), "v"(c1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(mid), "=s"(junk) : "v"(k1), "v"(c0));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(hi), "=s"(junk) : "v"(k1), "v"(c1));
}
__device__ __forceinline__ void fold128(u64 &lo, u64 &mid, u64 &hi) {
    const u64 low_part = mid << 32;
    lo += low_part;
    hi += (mid >> 32) + (lo < low_part ? 1 : 0);
    mid = 0;
}
__device__ __forceinline__ u64 reduce128(u64 lo, u64 hi, u32 cy, const LimbParams &lp) {
    const u64 th = hi + cy;                                        // the sum is th * 2^64 + lo
    const u64 H = mul_hi64(lo * lp.qinv, lp.q);                   // Montgomery reduction of the 128-bit sum
    return bred_add(th - H + lp.q, lp.q, lp.bred_hi);
}

__device__ __forceinline__ ulonglong2 own_operand(ulonglong2 c, const LimbParams &lp) {
    return make_ulonglong2(bred_add(c.x, lp.q, lp.bred_hi), bred_add(c.y, lp.q, lp.bred_hi));
}

template <int BETA>
__device__ __forceinline__ void keymac_wide_body(const KeyMacLaunch &L, int limb);

template <int BETA>
__global__ __launch_bounds__(256) void keymac_wide_kernel(KeyMacLaunch L) {
    keymac_wide_body<BETA>(L, (int)blockIdx.y);
}

template <int BETA>
__global__ __launch_bounds__(256) void keymac_wide_pair_kernel(KeyMacPair P) {
    if ((int)blockIdx.y < P.split) keymac_wide_body<BETA>(P.a, (int)blockIdx.y);
    else keymac_wide_body<BETA>(P.b, (int)blockIdx.y - P.split);
}

template <int BETA>
__device__ __forceinline__ void keymac_wide_body(const KeyMacLaunch &L, int limb) {
    const long long b = L.tile8 ? (long long)(blockIdx.x >> 3) : (long long)blockIdx.x;
    const int chunk = L.tile8 ? (int)(blockIdx.z * 8 + (blockIdx.x & 7)) : (int)blockIdx.z;
    const int chunks = L.tile8 ? (int)gridDim.z * 8 : (int)gridDim.z;
    const LimbParams lp = L.lp[limb];
    const long long row = (long long)limb * L.n;
    const ulonglong2 *pc = reinterpret_cast<const ulonglong2 *>(L.c2 + b * L.c2_poly_stride + row);
    const ulonglong2 *pk = reinterpret_cast<const ulonglong2 *>(L.key + (long long)(L.key_limb0 + limb) * L.n);
    ulonglong2 *po0 = reinterpret_cast<ulonglong2 *>(L.out0 + b * L.out_stride + row);
    ulonglong2 *po1 = reinterpret_cast<ulonglong2 *>(L.out1 + b * L.out1_stride + row);
    const long long cd = L.c2_digit_stride >> 1, kd = L.key_poly_stride >> 1;   // in 16-byte units
    const ulonglong2 *pown = L.alpha > 0 ? reinterpret_cast<const ulonglong2 *>(L.own + b * L.own_stride + row) : nullptr;
    const int own_digit = L.alpha > 0 ? limb / L.alpha : -1;
    const int pairs = L.n >> 1;
    const int beta = BETA > 0 ? BETA : L.beta;
    constexpr int G = BETA > 0 ? (BETA <= 6 ? BETA : (BETA + 1) / 2) : 1;      // digits whose loads are in flight together
    LR_KEYMAC_OPERAND_SETUP
    for (int e = chunk * 256 + threadIdx.x; e < pairs; e += chunks * 256) {
        const u32 ix0 = pgen ? perm_index(2u * (u32)e, pgen, pshift, pmask2) : 0u, ix1 = pgen ? perm_index(2u * (u32)e + 1u, pgen, pshift, pmask2) : 0u;
        u64 lo[4] = {0, 0, 0, 0}, mid[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        u32 cy[4] = {0, 0, 0, 0};
        if constexpr (BETA > 0) {
#pragma unroll
            for (int g0 = 0; g0 < BETA; g0 += G) {
                ulonglong2 c[G], k0[G], k1[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int i = g0 + u;
                    if (i < BETA) {
                        c[u] = LR_KEYMAC_OPERAND(i);
                        k0[u] = pk[e + (2 * i) * kd];
                        k1[u] = pk[e + (2 * i + 1) * kd];
                    }
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int i = g0 + u;
                    if (i < BETA) {
                        if (i == own_digit) c[u] = own_operand(c[u], lp);      // (workgroup-uniform: the limb decides)
                        mac128(k0[u].x, c[u].x, lo[0], mid[0], hi[0], cy[0]);
                        mac128(k0[u].y, c[u].y, lo[1], mid[1], hi[1], cy[1]);
                        mac128(k1[u].x, c[u].x, lo[2], mid[2], hi[2], cy[2]);
                        mac128(k1[u].y, c[u].y, lo[3], mid[3], hi[3], cy[3]);
                        if ((i & 3) == 3 || i == BETA - 1) {
#pragma unroll
                            for (int a = 0; a < 4; ++a) fold128(lo[a], mid[a], hi[a]);
                        }
                    }
                }
                if (g0 + G < BETA) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int i = 0; i < beta; ++i) {
                ulonglong2 c = LR_KEYMAC_OPERAND(i);
                if (i == own_digit) c = own_operand(c, lp);
                const ulonglong2 k0 = pk[e + (2 * i) * kd], k1 = pk[e + (2 * i + 1) * kd];
                mac128(k0.x, c.x, lo[0], mid[0], hi[0], cy[0]);
                mac128(k0.y, c.y, lo[1], mid[1], hi[1], cy[1]);
                mac128(k1.x, c.x, lo[2], mid[2], hi[2], cy[2]);
                mac128(k1.y, c.y, lo[3], mid[3], hi[3], cy[3]);
                if ((i & 3) == 3 || i == beta - 1) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) fold128(lo[a], mid[a], hi[a]);
                }
            }
        }
        st_stream(po0 + e, make_ulonglong2(reduce128(lo[0], hi[0], cy[0], lp), reduce128(lo[1], hi[1], cy[1], lp)));
        st_stream(po1 + e, make_ulonglong2(reduce128(lo[2], hi[2], cy[2], lp), reduce128(lo[3], hi[3], cy[3], lp)));
    }
}

hipError_t launch_keymac(const KeyMacLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    if (L.perm_gen != 0 && (L.alpha > 0 || L.logn < 1 || L.logn > 31 || (1 << L.logn) != L.n)) return hipErrorInvalidValue;   // permuted digits carry their own limbs
    int gx = ((L.n >> 1) + 255) / 256;
    if (gx > 64) gx = 64;
    KeyMacLaunch K = L;
    K.tile8 = (gx % 8 == 0 && (long long)batch * 8 < (1ll << 31)) ? 1 : 0;
    const dim3 grid(K.tile8 ? (unsigned)batch * 8u : (unsigned)batch, (unsigned)limbs, K.tile8 ? (unsigned)gx / 8u : (unsigned)gx), block(256);
    (void)hipGetLastError();
    if (K.wide) {
        switch (L.beta) {
#define LR_KMW(B) \
    case B: hipLaunchKernelGGL(keymac_wide_kernel<B>, grid, block, 0, stream, K); break;
            LR_KMW(1) LR_KMW(2) LR_KMW(3) LR_KMW(4) LR_KMW(5) LR_KMW(6) LR_KMW(7) LR_KMW(8) LR_KMW(9) LR_KMW(10)
#undef LR_KMW
        default: hipLaunchKernelGGL(keymac_wide_kernel<0>, grid, block, 0, stream, K); break;
        }
        return hipGetLastError();
    }
    switch (L.beta) {
#define LR_KM(B) \
    case B: hipLaunchKernelGGL(keymac_kernel<B>, grid, block, 0, stream, K); break;
        LR_KM(1) LR_KM(2) LR_KM(3) LR_KM(4) LR_KM(5) LR_KM(6) LR_KM(7) LR_KM(8) LR_KM(9) LR_KM(10)
#undef LR_KM
    default: hipLaunchKernelGGL(keymac_kernel<0>, grid, block, 0, stream, K); break;
    }
    return hipGetLastError();
}

hipError_t launch_keymac_pair(const KeyMacLaunch &A, int limbs_a, const KeyMacLaunch &B, int limbs_b, int batch, hipStream_t stream) {
    if (limbs_a <= 0 || limbs_b <= 0 || batch <= 0) return hipErrorNotSupported;
    if (!A.wide || !B.wide || A.beta != B.beta || A.n != B.n || A.beta < 1 || A.beta > 10) return hipErrorNotSupported;
    if ((A.perm_gen != 0 && A.alpha > 0) || (B.perm_gen != 0 && B.alpha > 0)) return hipErrorInvalidValue;
    int gx = ((A.n >> 1) + 255) / 256;
    if (gx > 64) gx = 64;
    KeyMacPair P;
    P.a = A;
    P.b = B;
    P.split = limbs_a;
    P.a.tile8 = P.b.tile8 = (gx % 8 == 0 && (long long)batch * 8 < (1ll << 31)) ? 1 : 0;
    const dim3 grid(P.a.tile8 ? (unsigned)batch * 8u : (unsigned)batch, (unsigned)(limbs_a + limbs_b), P.a.tile8 ? (unsigned)gx / 8u : (unsigned)gx), block(256);
    (void)hipGetLastError();
    switch (A.beta) {
#define LR_KMP(B_) \
    case B_: hipLaunchKernelGGL(keymac_wide_pair_kernel<B_>, grid, block, 0, stream, P); break;
        LR_KMP(1) LR_KMP(2) LR_KMP(3) LR_KMP(4) LR_KMP(5) LR_KMP(6) LR_KMP(7) LR_KMP(8) LR_KMP(9) LR_KMP(10)
#undef LR_KMP
    default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

// ---- Galois automorphisms, ring/ring_galois.go ---------------------------------------------
// NTT domain (:55-101): gather with the index computed on the fly (two bit reversals);
// coefficient domain (:106-127): scatter with sign.
__global__ __launch_bounds__(256) void permute_kernel(GaloisLaunch L) {
    const int limb = blockIdx.y;
    const long long b = blockIdx.z;
    const u64 *pin = (L.in_table ? L.in_table[b] : L.in + b * L.in_stride) + (long long)limb * L.n;
    u64 *pout = L.out + b * L.out_stride + (long long)limb * L.n;
    const u32 n = (u32)L.n, mask2 = 2 * n - 1;
    const int logn = L.logn;
    for (u32 j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        if (L.ntt_domain) {
            const u32 t1 = 2 * (__brev(j) >> (32 - logn)) + 1;
            const u32 t2 = ((((u32)L.gen * t1) & mask2) - 1) >> 1;
            pout[j] = pin[__brev(t2) >> (32 - logn)];
        } else {
            const u64 raw = (u64)j * L.gen;
            const u32 idx = (u32)raw & (n - 1);
            const u64 x = pin[j], q = L.lp[limb].q;
            pout[idx] = ((raw >> logn) & 1) ? q - x : x;
        }
    }
}

// Context.Permute (ring/ring_galois.go:106-127), coefficient domain, for rows that fit the LDS (N <= 2^14: 128 KiB of the CU's 160): the
// scatter out[i * gen mod N] = +-in[i] writes 8-byte words at a stride of gen words (a rotation by one column: 40 bytes), which the
// memory system pays per touched line -- PN14QP438 RotateCols spent 454 us in it against 162 us for the row swap (gen = 2N - 1: reversed
// but contiguous).  Here a workgroup reads one row coalesced into LDS and writes it out coalesced, taking output j from LDS word
// (j * gen^-1 mod 2N) mod N, negated when that product is >= N: the same map read from the other side, same values (0 negated is q).
__global__ __launch_bounds__(1024) void permute_coeff_lds_kernel(GaloisLaunch L, u32 ginv) {
    extern __shared__ u64 lr_permute_row[];
    const int limb = blockIdx.x;
    const long long b = blockIdx.y;
    const u32 n = (u32)L.n, mask2 = 2 * n - 1;
    const ulonglong2 *pin = reinterpret_cast<const ulonglong2 *>(L.in + b * L.in_stride + (long long)limb * L.n);
    ulonglong2 *pout = reinterpret_cast<ulonglong2 *>(L.out + b * L.out_stride + (long long)limb * L.n);
    for (u32 e = threadIdx.x; e < n / 2; e += 1024) {
        const ulonglong2 v = ld_stream(pin + e);
        lr_permute_row[2 * e] = v.x;
        lr_permute_row[2 * e + 1] = v.y;
    }
    __syncthreads();
    const u64 q = L.lp[limb].q;
    for (u32 e = threadIdx.x; e < n / 2; e += 1024) {
        const u32 t0 = ((2 * e) * ginv) & mask2, t1 = ((2 * e + 1) * ginv) & mask2;
        const u64 x0 = lr_permute_row[t0 & (n - 1)], x1 = lr_permute_row[t1 & (n - 1)];
        st_stream(pout + e, make_ulonglong2(t0 >= n ? q - x0 : x0, t1 >= n ? q - x1 : x1));
    }
}

hipError_t launch_permute(const GaloisLaunch &L, int limbs, int batch, hipStream_t stream) {
    if (limbs <= 0 || batch <= 0) return hipSuccess;
    if (!L.ntt_domain && !L.in_table && (L.gen & 1) && L.logn >= 11 && L.logn <= 14 && batch <= 65535) {
        // gen^-1 modulo 2N (gen odd): Newton's iteration doubles the correct low bits
        const u32 mask2 = 2u * (u32)L.n - 1u, g = (u32)L.gen & mask2;
        u32 inv = g;
        for (int it = 0; it < 5; ++it) inv *= 2u - g * inv;
        inv &= mask2;
        const size_t lds = (size_t)L.n * sizeof(u64);
        // more than 64 KiB of dynamic LDS is an attribute of the function ON THE CURRENT DEVICE: asked for once per device of the process
        static std::atomic<int> lds_ok[64];      // 0 = not asked yet, 1 = granted, 2 = refused
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<int> &state = lds_ok[dev >= 0 && dev < 64 ? dev : 0];
        if (state.load(std::memory_order_acquire) == 0)
            state.store(hipFuncSetAttribute(reinterpret_cast<const void *>(permute_coeff_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess ? 1 : 2,
                        std::memory_order_release);
        if (state.load(std::memory_order_acquire) == 1) {
            (void)hipGetLastError();
            hipLaunchKernelGGL(permute_coeff_lds_kernel, dim3((unsigned)limbs, (unsigned)batch), dim3(1024), lds, stream, L, inv);
            return hipGetLastError();
        }
        (void)hipGetLastError();     // (the attribute was refused: the scatter form below)
    }
    int gx = (L.n + 255) / 256;
    if (gx > 64) gx = 64;
    const dim3 grid((unsigned)gx, (unsigned)limbs, (unsigned)batch), block(256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(permute_kernel, grid, block, 0, stream, L);
    return hipGetLastError();
}

}  // namespace lr
