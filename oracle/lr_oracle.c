/*
 * lr_oracle.c -- CPU restatement (plain C) of Lattigo v1.3.1's `ring` hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see lr_oracle.h).  Parity status: PINNED by the
 * reference's golden NTT vectors and big-integer identities (tests/test_oracle_*.py).
 *
 * Each function names the reference file:line it follows.  `bits.Mul64` is
 * `unsigned __int128`; all other arithmetic is uint64 wrap-around like Go's.
 * Compile with -ffp-contract=off (the float64 correction of modUpExact must not be
 * fused; Go never fuses it because it has no multiply -- SURVEY A.5).
 */
#include "lr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

static inline u64 mul_hi(u64 a, u64 b) { return (u64)(((u128)a * b) >> 64); }
static inline u64 mul_lo(u64 a, u64 b) { return a * b; }

/* ===================== ring/modular_reduction.go ========================== */

/* MRedParams, modular_reduction.go:53-64: q^(2^63-1) mod 2^64 == q^-1 mod 2^64 */
u64 oc_mred_params(u64 q) {
    u64 qinv = 1, x = q;
    for (int i = 0; i < 63; i++) {
        qinv *= x;
        x *= x;
    }
    return qinv;
}

/* BRedParams, :97-106: floor(2^128/q) as {hi, lo} */
void oc_bred_params(u64 q, u64 u[2]) {
    u128 all1 = ~(u128)0;
    u128 r = all1 / q;
    if (all1 % q == (u128)(q - 1)) r += 1; /* (2^128-1)/q -> 2^128/q */
    u[0] = (u64)(r >> 64);
    u[1] = (u64)r;
}

/* MForm, :15-22 */
u64 oc_mform(u64 a, u64 q, const u64 u[2]) {
    u64 mhi = mul_hi(a, u[1]);
    u64 r = (u64)(0 - (a * u[0] + mhi)) * q;
    if (r >= q) r -= q;
    return r;
}

/* MFormConstant, :26-30 */
u64 oc_mform_constant(u64 a, u64 q, const u64 u[2]) {
    u64 mhi = mul_hi(a, u[1]);
    return (u64)(0 - (a * u[0] + mhi)) * q;
}

/* InvMForm, :34-41 */
u64 oc_inv_mform(u64 a, u64 q, u64 qinv) {
    u64 r = mul_hi(a * qinv, q);
    r = q - r;
    if (r >= q) r -= q;
    return r;
}

/* MRed, :70-79 */
u64 oc_mred(u64 x, u64 y, u64 q, u64 qinv) {
    u128 a = (u128)x * y;
    u64 ahi = (u64)(a >> 64), alo = (u64)a;
    u64 R = alo * qinv;
    u64 H = mul_hi(R, q);
    u64 r = ahi - H + q;
    if (r >= q) r -= q;
    return r;
}

/* MRedConstant, :83-89 */
u64 oc_mred_constant(u64 x, u64 y, u64 q, u64 qinv) {
    u128 a = (u128)x * y;
    u64 ahi = (u64)(a >> 64), alo = (u64)a;
    u64 R = alo * qinv;
    u64 H = mul_hi(R, q);
    return ahi - H + q;
}

/* BRedAdd, :112-119 */
u64 oc_bred_add(u64 x, u64 q, const u64 u[2]) {
    u64 s0 = mul_hi(x, u[0]);
    u64 r = x - s0 * q;
    if (r >= q) r -= q;
    return r;
}

/* BRedAddConstant, :123-126 */
u64 oc_bred_add_constant(u64 x, u64 q, const u64 u[2]) {
    u64 s0 = mul_hi(x, u[0]);
    return x - s0 * q;
}

/* shared body of BRed/BRedConstant, :133-168 / :172-207 (same carry chain) */
static inline u64 bred_core(u64 x, u64 y, u64 q, const u64 u[2]) {
    u128 a = (u128)x * y;
    u64 ahi = (u64)(a >> 64), alo = (u64)a;
    u64 lhi = mul_hi(alo, u[1]);
    u128 m = (u128)alo * u[0];
    u64 mhi = (u64)(m >> 64), mlo = (u64)m;
    u64 s0 = mlo + lhi;
    u64 carry = s0 < mlo;
    u64 s1 = mhi + carry;
    m = (u128)ahi * u[1];
    mhi = (u64)(m >> 64); mlo = (u64)m;
    u64 t = mlo + s0;
    carry = t < mlo;
    lhi = mhi + carry;
    s0 = ahi * u[0] + s1 + lhi;
    return alo - s0 * q;
}

/* BRed, :133-168 */
u64 oc_bred(u64 x, u64 y, u64 q, const u64 u[2]) {
    u64 r = bred_core(x, y, q, u);
    if (r >= q) r -= q;
    return r;
}

/* BRedConstant, :172-207 */
u64 oc_bred_constant(u64 x, u64 y, u64 q, const u64 u[2]) {
    return bred_core(x, y, q, u);
}

/* CRed, :211-216 */
u64 oc_cred(u64 a, u64 q) { return a >= q ? a - q : a; }

/* ============================ ring/utils.go ================================ */

/* PowerOf2, utils.go:8-17 */
u64 oc_power_of_2(u64 x, u64 n, u64 q, u64 qinv) {
    /* Go: x>>(64-n) with n==0 gives shift count 64 -> 0 in Go (not UB) */
    u64 ahi = (n == 0) ? 0 : (n > 64 ? 0 : x >> (64 - n));
    u64 alo = (n >= 64) ? 0 : x << n;
    u64 R = alo * qinv;
    u64 H = mul_hi(R, q);
    u64 r = ahi - H + q;
    if (r >= q) r -= q;
    return r;
}

/* ModExp, utils.go:25-36 (Barrett square-and-multiply, LSB first) */
u64 oc_mod_exp(u64 x, u64 e, u64 p) {
    u64 u[2];
    oc_bred_params(p, u);
    u64 result = 1;
    for (u64 i = e; i > 0; i >>= 1) {
        if (i & 1) result = oc_bred(result, x, p, u);
        x = oc_bred(x, x, p, u);
    }
    return result;
}

/* gcd, utils.go:53-61 (returns 0 when either argument is 0) */
static u64 gcd_q(u64 a, u64 b) {
    if (a == 0 || b == 0) return 0;
    while (b != 0) { u64 t = a % b; a = b; b = t; }
    return a;
}

/* smallPrimes, utils.go:290-391: the table is exactly the first 2000 primes
 * (2 ... 17389); regenerated here with a sieve instead of being copied. */
#define N_SMALL_PRIMES 2000
static u64 small_primes[N_SMALL_PRIMES];
static int small_primes_ready = 0;
static void init_small_primes(void) {
    if (small_primes_ready) return;
    enum { LIM = 17400 };
    static unsigned char comp[LIM + 1];
    int n = 0;
    for (int a = 2; a <= LIM && n < N_SMALL_PRIMES; a++) {
        if (!comp[a]) {
            small_primes[n++] = (u64)a;
            for (int b = a * a; b <= LIM; b += a) comp[b] = 1;
        }
    }
    small_primes_ready = 1;
}

/* deterministic stand-in for RandUniform(num-1, mask) (utils.go:104-110): the
 * bases are random in Go; they only influence the verdict on composites. */
static u64 sm64_state = 0x9E3779B97F4A7C15ull;
static u64 sm64_next(void) {
    u64 z = (sm64_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int bit_len64(u64 x) { int n = 0; while (x) { n++; x >>= 1; } return n; }

/* IsPrime, utils.go:75-127 (50 Miller-Rabin rounds) */
int oc_is_prime(u64 num) {
    init_small_primes();
    if (num < 2) return 0;
    for (int i = 0; i < N_SMALL_PRIMES; i++) if (num == small_primes[i]) return 1;
    for (int i = 0; i < N_SMALL_PRIMES; i++) if (num % small_primes[i] == 0) return 0;
    u64 s = num - 1; int k = 0;
    while ((s & 1) == 0) { s >>= 1; k++; }
    u64 u[2];
    oc_bred_params(num, u);
    int bl = bit_len64(num);
    u64 mask = bl >= 64 ? ~(u64)0 : (((u64)1 << bl) - 1);
    for (int trial = 0; trial < 50; trial++) {
        u64 b;
        do { do { b = sm64_next() & mask; } while (b >= num - 1); } while (b < 2);
        u64 x = oc_mod_exp(b, s, num);
        if (x != 1) {
            int i = 0;
            while (x != num - 1) {
                if (i == k - 1) return 0;
                i++;
                x = oc_bred(x, x, num, u);
            }
        }
    }
    return 1;
}

/* GenerateNTTPrimes, utils.go:133-175 */
int oc_generate_ntt_primes(u64 logQ, u64 logN, u64 levels, u64 *out) {
    if (logQ > 60) return -1;
    u64 Qpow2 = (u64)1 << logQ, twoN = (u64)2 << logN;
    u64 x = Qpow2 + 1, y = Qpow2 + 1;
    u64 n = 0;
    for (;;) {
        if (oc_is_prime(x)) { out[n++] = x; if (n == levels) return (int)n; }
        x += twoN;
        if (twoN > y) {               /* :161 dead for real sizes, kept literally */
            y -= twoN;
            if (oc_is_prime(y)) { out[n++] = y; if (n == levels) return (int)n; }
        }
    }
}

/* polynomialPollardsRho, utils.go:212-218 */
static u64 poly_rho(u64 x1, u64 x2, u64 c) {
    u64 z = oc_mod_exp(x1, 2, x2);
    z += c;
    z %= x2;
    return z;
}

/* factorizationPollardsRho, utils.go:222-247 (non-standard swap kept) */
static u64 factor_rho(u64 m) {
    u64 x, y, d = 0;
    for (u64 c = 1; c < 10; c++) {
        x = 2; y = 2; d = 1;
        while (d != 0) {
            x = poly_rho(x, m, c);
            y = poly_rho(poly_rho(y, m, c), m, c);
            if (y > x) { u64 t = x; x = y; y = t; }
            d = gcd_q(x - y, m);
            if (d > 1) return d;
        }
    }
    return d;
}

/* getFactors, utils.go:251-288 (may return composite "factors"; kept) */
int oc_get_factors(u64 n, u64 *out, int cap) {
    init_small_primes();
    int nf = 0;
    u64 m = n;
    for (int i = 0; i < N_SMALL_PRIMES; i++) {
        u64 sp = small_primes[i];
        int add = 0;
        while (m % sp == 0) { m /= sp; add = 1; }
        if (add && nf < cap) out[nf++] = sp;
    }
    if (m == 1) return nf;
    for (;;) {
        u64 f = factor_rho(m);
        if (f == 0) { if (nf < cap) out[nf++] = m; break; }
        m /= f;
        if (nf > 0 && f == out[nf - 1]) continue;
        if (nf < cap) out[nf++] = f;
    }
    return nf;
}

/* primitiveRoot, utils.go:182-205 (first candidate is g = 3) */
u64 oc_primitive_root(u64 q) {
    u64 factors[128];
    int nf = oc_get_factors(q - 1, factors, 128);
    u64 g = 2;
    int not_found = 1;
    while (not_found) {
        g++;
        for (int i = 0; i < nf; i++) {
            u64 tmp = (q - 1) / factors[i];
            if (oc_mod_exp(g, tmp, q) == 1) { not_found = 1; break; }
            not_found = 0;
        }
    }
    return g;
}

/* utils.BitReverse64, utils/utils.go:58-60 */
u64 oc_bit_reverse64(u64 index, u64 bitlen) {
    u64 r = 0;
    for (int i = 0; i < 64; i++) { r = (r << 1) | ((index >> i) & 1); }
    return bitlen == 0 ? 0 : r >> (64 - bitlen);
}

/* ======================== ring/ring_context.go ============================= */

/* SetParameters (:68-126) + GenNTTParams (:129-209) */
int oc_context_new(u64 N, const u64 *moduli, int L, oc_context **out) {
    *out = NULL;
    if (N == 0 || (N & (N - 1)) != 0) return 2;          /* :71-73 */
    oc_context *c = (oc_context *)calloc(1, sizeof(*c));
    c->N = N; c->L = L;
    c->q = (u64 *)malloc(sizeof(u64) * L);
    c->mask = (u64 *)malloc(sizeof(u64) * L);
    c->bred = (u64 *)malloc(sizeof(u64) * 2 * L);
    c->mred = (u64 *)calloc(L, sizeof(u64));
    c->rescale = (u64 *)calloc((size_t)L * L, sizeof(u64));
    c->psi_mont = (u64 *)malloc(sizeof(u64) * L);
    c->psi_inv_mont = (u64 *)malloc(sizeof(u64) * L);
    c->n_inv = (u64 *)malloc(sizeof(u64) * L);
    c->ntt_psi = (u64 *)malloc(sizeof(u64) * L * N);
    c->ntt_psi_inv = (u64 *)malloc(sizeof(u64) * L * N);
    for (int i = 0; i < L; i++) {
        u64 qi = moduli[i];
        c->q[i] = qi;
        int bl = bit_len64(qi);
        c->mask[i] = bl >= 64 ? ~(u64)0 : (((u64)1 << bl) - 1);   /* :84 */
        oc_bred_params(qi, &c->bred[2 * i]);                       /* :100 */
        if ((qi & (qi - 1)) != 0 && qi != 0) c->mred[i] = oc_mred_params(qi); /* :104-106 */
    }
    /* :141-146 */
    for (int i = 0; i < L; i++) {
        u64 qi = c->q[i];
        if (!oc_is_prime(qi) || (qi & ((N << 1) - 1)) != 1) { oc_context_free(c); return 1; }
    }
    /* rescaleParams :148-158 */
    for (int j = L - 1; j > 0; j--)
        for (int i = 0; i < j; i++)
            c->rescale[(size_t)(j - 1) * L + i] =
                oc_mform(oc_mod_exp(c->q[j], c->q[i] - 2, c->q[i]), c->q[i], &c->bred[2 * i]);

    u64 logN = (u64)bit_len64(N) - 1;
    for (int i = 0; i < L; i++) {
        u64 qi = c->q[i];
        const u64 *u = &c->bred[2 * i];
        c->n_inv[i] = oc_mform(oc_mod_exp(N, qi - 2, qi), qi, u);          /* :171 */
        u64 g = oc_primitive_root(qi);                                      /* :178 */
        u64 twoN = N << 1;
        u64 power = (qi - 1) / twoN;
        u64 power_inv = (qi - 1) - power;
        u64 psi = oc_mform(oc_mod_exp(g, power, qi), qi, u);                /* :185 */
        u64 psi_inv = oc_mform(oc_mod_exp(g, power_inv, qi), qi, u);        /* :186 */
        c->psi_mont[i] = psi; c->psi_inv_mont[i] = psi_inv;
        u64 *t = c->ntt_psi + (size_t)i * N, *ti = c->ntt_psi_inv + (size_t)i * N;
        t[0] = oc_mform(1, qi, u); ti[0] = oc_mform(1, qi, u);              /* :192-193 */
        for (u64 j = 1; j < N; j++) {                                       /* :195-203 */
            u64 prev = oc_bit_reverse64(j - 1, logN), next = oc_bit_reverse64(j, logN);
            t[next] = oc_mred(t[prev], psi, qi, c->mred[i]);
            ti[next] = oc_mred(ti[prev], psi_inv, qi, c->mred[i]);
        }
    }
    *out = c;
    return 0;
}

void oc_context_free(oc_context *c) {
    if (!c) return;
    free(c->q); free(c->mask); free(c->bred); free(c->mred); free(c->rescale);
    free(c->psi_mont); free(c->psi_inv_mont); free(c->n_inv); free(c->ntt_psi); free(c->ntt_psi_inv);
    free(c);
}

/* ============================== ring/ntt.go ================================ */

/* Butterfly, ntt.go:32-40 */
static inline void butterfly(u64 U, u64 V, u64 psi, u64 q, u64 qinv, u64 *X, u64 *Y) {
    if (U > 2 * q) U -= 2 * q;
    V = oc_mred_constant(V, psi, q, qinv);
    *X = U + V;
    *Y = U + 2 * q - V;
}

/* InvButterfly, ntt.go:43-50 */
static inline void inv_butterfly(u64 U, u64 V, u64 psi, u64 q, u64 qinv, u64 *X, u64 *Y) {
    u64 x = U + V;
    if (x > 2 * q) x -= 2 * q;
    *Y = oc_mred_constant(U + 2 * q - V, psi, q, qinv);
    *X = x;
}

/* NTT, ntt.go:53-86 */
void oc_ntt_limb(const u64 *in, u64 *out, u64 N, const u64 *psi, u64 q, u64 qinv, const u64 bred[2]) {
    u64 t = N >> 1;
    u64 F = psi[1];
    for (u64 j = 0; j < t; j++) {
        u64 X, Y;
        butterfly(in[j], in[j + t], F, q, qinv, &X, &Y);
        out[j] = X; out[j + t] = Y;
    }
    for (u64 m = 2; m < N; m <<= 1) {
        t >>= 1;
        for (u64 i = 0; i < m; i++) {
            u64 j1 = (i * t) << 1, j2 = j1 + t - 1;
            F = psi[m + i];
            for (u64 j = j1; j <= j2; j++) {
                u64 X, Y;
                butterfly(out[j], out[j + t], F, q, qinv, &X, &Y);
                out[j] = X; out[j + t] = Y;
            }
        }
    }
    for (u64 i = 0; i < N; i++) out[i] = oc_bred_add(out[i], q, bred);
}

/* InvNTT, ntt.go:89-139 */
void oc_intt_limb(const u64 *in, u64 *out, u64 N, const u64 *psi_inv, u64 n_inv, u64 q, u64 qinv) {
    u64 t = 1, j1 = 0, h = N >> 1;
    for (u64 i = 0; i < h; i++) {
        u64 F = psi_inv[h + i];
        u64 X, Y;
        inv_butterfly(in[j1], in[j1 + t], F, q, qinv, &X, &Y);
        out[j1] = X; out[j1 + t] = Y;
        j1 += t << 1;
    }
    t <<= 1;
    for (u64 m = N >> 1; m > 1; m >>= 1) {
        j1 = 0; h = m >> 1;
        for (u64 i = 0; i < h; i++) {
            u64 j2 = j1 + t - 1;
            u64 F = psi_inv[h + i];
            for (u64 j = j1; j <= j2; j++) {
                u64 X, Y;
                inv_butterfly(out[j], out[j + t], F, q, qinv, &X, &Y);
                out[j] = X; out[j + t] = Y;
            }
            j1 += t << 1;
        }
        t <<= 1;
    }
    for (u64 j = 0; j < N; j++) out[j] = oc_mred(out[j], n_inv, q, qinv);
}

/* Context.NTTLvl, ntt.go:11-15 (serial over limbs) */
void oc_ntt_lvl(const oc_context *c, int level, const u64 *in, u64 *out) {
    for (int x = 0; x <= level; x++)
        oc_ntt_limb(in + (size_t)x * c->N, out + (size_t)x * c->N, c->N, c->ntt_psi + (size_t)x * c->N,
                    c->q[x], c->mred[x], &c->bred[2 * x]);
}

/* Context.InvNTTLvl, ntt.go:25-29 */
void oc_intt_lvl(const oc_context *c, int level, const u64 *in, u64 *out) {
    for (int x = 0; x <= level; x++)
        oc_intt_limb(in + (size_t)x * c->N, out + (size_t)x * c->N, c->N, c->ntt_psi_inv + (size_t)x * c->N,
                     c->n_inv[x], c->q[x], c->mred[x]);
}

/* ============================== ring/ring.go =============================== */

void oc_ewise(const oc_context *c, int op, int level, const u64 *a, const u64 *b, u64 *out, const u64 *sc) {
    const u64 N = c->N;
    for (int i = 0; i <= level; i++) {
        const u64 q = c->q[i], qinv = c->mred[i];
        const u64 *u = &c->bred[2 * i];
        const u64 *p1 = a ? a + (size_t)i * N : NULL;
        const u64 *p2 = b ? b + (size_t)i * N : NULL;
        u64 *p3 = out + (size_t)i * N;
        u64 s = 0;
        switch (op) {
        case OC_MUL_SCALAR:        /* ring.go:516,530 */
            s = oc_mform(oc_bred_add(sc[0], q, u), q, u); break;
        case OC_MUL_SCALAR_LIMBS:  /* ring.go:547,563 (scalar already reduced mod qi by the caller's big.Int Mod) */
            s = oc_mform(oc_bred_add(sc[i], q, u), q, u); break;
        case OC_ADD_SCALAR_LIMBS: case OC_SUB_SCALAR_LIMBS:
            s = sc[i]; break;
        case OC_MUL_BY_POW2:
            s = sc[0]; break;
        default: break;
        }
        if (op == OC_MUL_BY_POW2) {
            /* MulByPow2(Lvl), ring.go:629-656: MForm(p1)->p2 first, then p2 = PowerOf2(p1[j]) reads p1 */
            for (u64 j = 0; j < N; j++) p3[j] = oc_mform(p1[j], q, u);
            for (u64 j = 0; j < N; j++) p3[j] = oc_power_of_2(p1[j], s, q, qinv);
            continue;
        }
        for (u64 j = 0; j < N; j++) {
            switch (op) {
            case OC_ADD:        p3[j] = oc_cred(p1[j] + p2[j], q); break;                       /* :10-29 */
            case OC_ADD_NOMOD:  p3[j] = p1[j] + p2[j]; break;                                    /* :32-51 */
            case OC_SUB:        p3[j] = oc_cred((p1[j] + q) - p2[j], q); break;                  /* :54-74 */
            case OC_SUB_NOMOD:  p3[j] = (p1[j] + q) - p2[j]; break;                              /* :77-97 */
            case OC_NEG:        p3[j] = q - p1[j]; break;                                        /* :100-119 */
            case OC_REDUCE:     p3[j] = oc_bred_add(p1[j], q, u); break;                         /* :122-143 */
            case OC_MUL_COEFFS: p3[j] = oc_bred(p1[j], p2[j], q, u); break;                      /* :187 */
            case OC_MUL_COEFFS_AND_ADD: p3[j] = oc_cred(p3[j] + oc_bred(p1[j], p2[j], q, u), q); break; /* :198 */
            case OC_MUL_COEFFS_AND_ADD_NOMOD: p3[j] += oc_bred(p1[j], p2[j], q, u); break;       /* :209 */
            case OC_MUL_COEFFS_CONSTANT: p3[j] = oc_bred_constant(p1[j], p2[j], q, u); break;    /* :335 */
            case OC_MUL_MONT:   p3[j] = oc_mred(p1[j], p2[j], q, qinv); break;                   /* :221-244 */
            case OC_MUL_MONT_AND_ADD: p3[j] = oc_cred(p3[j] + oc_mred(p1[j], p2[j], q, qinv), q); break; /* :247-270 */
            case OC_MUL_MONT_AND_ADD_NOMOD: p3[j] += oc_mred(p1[j], p2[j], q, qinv); break;      /* :273-294 */
            case OC_MUL_MONT_CONSTANT_AND_ADD_NOMOD: p3[j] += oc_mred_constant(p1[j], p2[j], q, qinv); break; /* :297 */
            case OC_MUL_MONT_AND_SUB: p3[j] = oc_cred(p3[j] + (q - oc_mred(p1[j], p2[j], q, qinv)), q); break; /* :311 */
            case OC_MUL_MONT_AND_SUB_NOMOD: p3[j] = p3[j] + (q - oc_mred(p1[j], p2[j], q, qinv)); break;       /* :323 */
            case OC_MUL_MONT_CONSTANT: p3[j] = oc_mred_constant(p1[j], p2[j], q, qinv); break;   /* :347 */
            case OC_MFORM:      p3[j] = oc_mform(p1[j], q, u); break;                            /* :583-607 */
            case OC_INV_MFORM:  p3[j] = oc_inv_mform(p1[j], q, qinv); break;                     /* :610-618 */
            case OC_MUL_SCALAR: case OC_MUL_SCALAR_LIMBS:
                                p3[j] = oc_mred(p1[j], s, q, qinv); break;                       /* :513-571 */
            case OC_ADD_SCALAR_LIMBS: p3[j] = oc_cred(p1[j] + s, q); break;                      /* :477-486 */
            case OC_SUB_SCALAR_LIMBS: p3[j] = oc_cred(p1[j] + (q - s), q); break;                /* :500-509 */
            case OC_COPY:       p3[j] = p1[j]; break;                                            /* ring_object.go:85-109 */
            default: break;
            }
        }
    }
}

/* ===================== ring/ring_basis_extension.go ======================== */

static u64 mulmod_u128(u64 a, u64 b, u64 m) { return (u64)(((u128)a * b) % m); }

/* product of v[k], k != skip, modulo m  (stands in for big.Int Quo/Mod at :104-121) */
static u64 prod_mod_skip(const u64 *v, int n, int skip, u64 m) {
    u64 r = 1 % m;
    for (int k = 0; k < n; k++) if (k != skip) r = mulmod_u128(r, v[k] % m, m);
    return r;
}

/* modular inverse for prime modulus m (big.Int ModInverse at :106 is unique in [0,m)) */
static u64 inv_mod_prime(u64 a, u64 m) {
    u64 r = 1, e = m - 2; a %= m;
    while (e) { if (e & 1) r = mulmod_u128(r, a, m); a = mulmod_u128(a, a, m); e >>= 1; }
    return r;
}

/* basisextenderparameters, ring_basis_extension.go:76-142 */
oc_modup_params *oc_modup_params_new(const u64 *Q, int nQ, const u64 *P, int nP) {
    oc_modup_params *p = (oc_modup_params *)calloc(1, sizeof(*p));
    p->nQ = nQ; p->nP = nP;
    p->Q = (u64 *)malloc(sizeof(u64) * nQ); memcpy(p->Q, Q, sizeof(u64) * nQ);
    p->P = (u64 *)malloc(sizeof(u64) * nP); memcpy(p->P, P, sizeof(u64) * nP);
    p->bredQ = (u64 *)malloc(sizeof(u64) * 2 * nQ); p->mredQ = (u64 *)malloc(sizeof(u64) * nQ);
    p->bredP = (u64 *)malloc(sizeof(u64) * 2 * nP); p->mredP = (u64 *)malloc(sizeof(u64) * nP);
    for (int i = 0; i < nQ; i++) { oc_bred_params(Q[i], &p->bredQ[2 * i]); p->mredQ[i] = oc_mred_params(Q[i]); }
    for (int j = 0; j < nP; j++) { oc_bred_params(P[j], &p->bredP[2 * j]); p->mredP[j] = oc_mred_params(P[j]); }
    p->qib_mont = (u64 *)malloc(sizeof(u64) * nQ);
    p->qispj_mont = (u64 *)malloc(sizeof(u64) * nQ * nP);
    p->qpj_inv = (u64 *)malloc(sizeof(u64) * nP * (nQ + 1));
    for (int i = 0; i < nQ; i++) {
        u64 qi = Q[i];
        u64 qistar_mod_qi = prod_mod_skip(Q, nQ, i, qi);                   /* QiStar mod qi */
        u64 qibarre = inv_mod_prime(qistar_mod_qi, qi);                    /* :106-107 */
        p->qib_mont[i] = oc_mform(qibarre, qi, &p->bredQ[2 * i]);          /* :110 */
        for (int j = 0; j < nP; j++)                                       /* :114-117 */
            p->qispj_mont[(size_t)i * nP + j] = oc_mform(prod_mod_skip(Q, nQ, i, P[j]), P[j], &p->bredP[2 * j]);
    }
    for (int j = 0; j < nP; j++) {                                         /* :124-139 */
        u64 pj = P[j];
        u64 v = pj - prod_mod_skip(Q, nQ, -1, pj);
        u64 *row = p->qpj_inv + (size_t)j * (nQ + 1);
        row[0] = 0;
        for (int i = 1; i < nQ + 1; i++) row[i] = oc_cred(row[i - 1] + v, pj);
    }
    return p;
}

void oc_modup_params_free(oc_modup_params *p) {
    if (!p) return;
    free(p->Q); free(p->P); free(p->bredQ); free(p->mredQ); free(p->bredP); free(p->mredP);
    free(p->qib_mont); free(p->qispj_mont); free(p->qpj_inv); free(p);
}

/* modUpExact, ring_basis_extension.go:352-393 */
void oc_modup_exact(const oc_modup_params *p, const u64 *in, int n_in, u64 *out, int n_out, u64 N) {
    u64 y[64];
    for (u64 x = 0; x < N; x++) {
        double vi = 0;
        for (int i = 0; i < n_in; i++) {
            y[i] = oc_mred(in[(size_t)i * N + x], p->qib_mont[i], p->Q[i], p->mredQ[i]);
            vi += (double)y[i] / (double)p->Q[i];
        }
        u64 v = (u64)vi;
        for (int j = 0; j < n_out; j++) {
            u64 xpj = 0;
            for (int i = 0; i < n_in; i++) {
                xpj += oc_mred(y[i], p->qispj_mont[(size_t)i * p->nP + j], p->P[j], p->mredP[j]);
                if ((i & 7) == 6) xpj = oc_bred_add(xpj, p->P[j], &p->bredP[2 * j]);
            }
            out[(size_t)j * N + x] = oc_bred_add(xpj + p->qpj_inv[(size_t)j * (p->nQ + 1) + v], p->P[j], &p->bredP[2 * j]);
        }
    }
}

/* genModDownParams, :39-53: params[i] = MForm((prod other)^-1 mod m_i) */
static u64 *gen_moddown(const oc_context *cP, const oc_context *cQ) {
    u64 *r = (u64 *)malloc(sizeof(u64) * cP->L);
    for (int i = 0; i < cP->L; i++) {
        u64 qi = cP->q[i];
        u64 v = prod_mod_skip(cQ->q, cQ->L, -1, qi);
        v = oc_mod_exp(v, qi - 2, qi);
        r[i] = oc_mform(v, qi, &cP->bred[2 * i]);
    }
    return r;
}

/* NewFastBasisExtender, :57-74 */
oc_bext *oc_bext_new(const oc_context *cQ, const oc_context *cP) {
    oc_bext *b = (oc_bext *)calloc(1, sizeof(*b));
    b->cQ = cQ; b->cP = cP;
    b->qp = oc_modup_params_new(cQ->q, cQ->L, cP->q, cP->L);
    b->pq = oc_modup_params_new(cP->q, cP->L, cQ->q, cQ->L);
    b->moddown_pq = gen_moddown(cQ, cP);     /* :66 genModDownParams(contextQ, contextP) */
    b->moddown_qp = gen_moddown(cP, cQ);     /* :67 */
    b->poolQ = (u64 *)calloc((size_t)cQ->L * cQ->N, sizeof(u64));
    b->poolP = (u64 *)calloc((size_t)cP->L * cP->N, sizeof(u64));
    return b;
}

void oc_bext_free(oc_bext *b) {
    if (!b) return;
    oc_modup_params_free(b->qp); oc_modup_params_free(b->pq);
    free(b->moddown_pq); free(b->moddown_qp); free(b->poolQ); free(b->poolP); free(b);
}

/* ModUpSplitQP, :147-149 */
void oc_modup_split_qp(oc_bext *b, int level, const u64 *p1, u64 *p2) {
    oc_modup_exact(b->qp, p1, level + 1, p2, b->qp->nP, b->cQ->N);
}
/* ModUpSplitPQ, :154-156 */
void oc_modup_split_pq(oc_bext *b, int level, const u64 *p1, u64 *p2) {
    oc_modup_exact(b->pq, p1, level + 1, p2, b->pq->nP, b->cQ->N);
}

static void moddown_tail(const oc_context *c, const u64 *params, int level, const u64 *p1, const u64 *pool,
                         u64 *p2, int ntt, u64 *pool_mut) {
    const u64 N = c->N;
    for (int i = 0; i <= level; i++) {
        u64 qi = c->q[i], qinv = c->mred[i];
        const u64 *p1t = p1 + (size_t)i * N;
        u64 *p2t = p2 + (size_t)i * N;
        const u64 *p3t = pool + (size_t)i * N;
        if (ntt) {
            u64 *p3m = pool_mut + (size_t)i * N;
            oc_ntt_limb(p3m, p3m, N, c->ntt_psi + (size_t)i * N, qi, qinv, &c->bred[2 * i]);
        }
        for (u64 j = 0; j < N; j++) p2t[j] = oc_mred(p1t[j] + (qi - p3t[j]), params[i], qi, qinv);
    }
}

/* ModDownNTTPQ, :163-201 */
void oc_moddown_ntt_pq(oc_bext *b, int level, u64 *p1, u64 *p2) {
    const oc_context *cQ = b->cQ, *cP = b->cP;
    const u64 N = cQ->N;
    u64 *p1P = p1 + (size_t)cQ->L * N;
    for (int j = 0; j < cP->L; j++)
        oc_intt_limb(p1P + (size_t)j * N, p1P + (size_t)j * N, N, cP->ntt_psi_inv + (size_t)j * N, cP->n_inv[j], cP->q[j], cP->mred[j]);
    oc_modup_exact(b->pq, p1P, cP->L, b->poolQ, level + 1, N);
    moddown_tail(cQ, b->moddown_pq, level, p1, b->poolQ, p2, 1, b->poolQ);
}

/* ModDownSplitedNTTPQ, :207-242 */
void oc_moddown_split_ntt_pq(oc_bext *b, int level, const u64 *p1Q, u64 *p1P, u64 *p2) {
    const oc_context *cQ = b->cQ, *cP = b->cP;
    oc_intt_lvl(cP, cP->L - 1, p1P, p1P);
    oc_modup_exact(b->pq, p1P, cP->L, b->poolQ, level + 1, cQ->N);
    moddown_tail(cQ, b->moddown_pq, level, p1Q, b->poolQ, p2, 1, b->poolQ);
}

/* ModDownPQ, :248-275 */
void oc_moddown_pq(oc_bext *b, int level, const u64 *p1, u64 *p2) {
    const oc_context *cQ = b->cQ;
    oc_modup_exact(b->pq, p1 + (size_t)(level + 1) * cQ->N, b->qp->nP, b->poolQ, level + 1, cQ->N);
    moddown_tail(cQ, b->moddown_pq, level, p1, b->poolQ, p2, 0, NULL);
}

/* ModDownSplitedPQ, :281-308 */
void oc_moddown_split_pq(oc_bext *b, int level, const u64 *p1Q, const u64 *p1P, u64 *p2) {
    const oc_context *cQ = b->cQ;
    oc_modup_exact(b->pq, p1P, b->cP->L, b->poolQ, level + 1, cQ->N);
    moddown_tail(cQ, b->moddown_pq, level, p1Q, b->poolQ, p2, 0, NULL);
}

/* ModDownSplitedQP, :314-350 */
void oc_moddown_split_qp(oc_bext *b, int levelQ, int levelP, const u64 *p1Q, const u64 *p1P, u64 *p2) {
    const oc_context *cP = b->cP;
    oc_modup_split_qp(b, levelQ, p1Q, b->poolP);
    moddown_tail(cP, b->moddown_qp, levelP, p1P, b->poolP, p2, 0, NULL);
}

/* NewDecomposer, :415-472 */
oc_decomposer *oc_decomposer_new(const u64 *Q, int nQ, const u64 *P, int nP) {
    oc_decomposer *d = (oc_decomposer *)calloc(1, sizeof(*d));
    d->nQ = nQ; d->nP = nP; d->alpha = nP;
    d->beta = (nQ + nP - 1) / nP;                       /* ceil(len(Q)/alpha), :433 */
    d->xalpha = (int *)malloc(sizeof(int) * d->beta);
    for (int i = 0; i < d->beta; i++) d->xalpha[i] = d->alpha;
    if (nQ % d->alpha != 0) d->xalpha[d->beta - 1] = nQ % d->alpha;
    d->modup = (oc_modup_params ***)calloc(d->beta, sizeof(*d->modup));
    u64 *Pi = (u64 *)malloc(sizeof(u64) * (nQ + nP));
    memcpy(Pi, Q, sizeof(u64) * nQ); memcpy(Pi + nQ, P, sizeof(u64) * nP);
    for (int i = 0; i < d->beta; i++) {
        int cnt = d->xalpha[i] - 1;
        d->modup[i] = (oc_modup_params **)calloc(cnt > 0 ? cnt : 1, sizeof(oc_modup_params *));
        for (int j = 0; j < cnt; j++)
            d->modup[i][j] = oc_modup_params_new(Q + (size_t)i * d->alpha, j + 2, Pi, nQ + nP);
    }
    free(Pi);
    return d;
}

void oc_decomposer_free(oc_decomposer *d) {
    if (!d) return;
    for (int i = 0; i < d->beta; i++) {
        for (int j = 0; j < d->xalpha[i] - 1; j++) oc_modup_params_free(d->modup[i][j]);
        free(d->modup[i]);
    }
    free(d->modup); free(d->xalpha); free(d);
}

static inline u64 ext_one(const oc_modup_params *p, const u64 *y, int ny, int u, u64 v) {
    u64 xpj = 0;
    for (int i = 0; i < ny; i++) {
        xpj += oc_mred(y[i], p->qispj_mont[(size_t)i * p->nP + u], p->P[u], p->mredP[u]);
        if ((i & 7) == 6) xpj = oc_bred_add(xpj, p->P[u], &p->bredP[2 * u]);
    }
    return oc_bred_add(xpj + p->qpj_inv[(size_t)u * (p->nQ + 1) + v], p->P[u], &p->bredP[2 * u]);
}

/* shared body of Decompose (:476-597, split=0) and DecomposeAndSplit (:601-713, split=1) */
static void decompose_impl(const oc_decomposer *d, int level, int crt, const u64 *p0, u64 *p1Q, u64 *p1P, u64 N, int split) {
    int alphai = d->xalpha[crt];
    int st = crt * d->alpha, ed = st + alphai;
    if ((ed > level + 1 && (level + 1) % d->nP == 1) || alphai == 1) {
        for (u64 x = 0; x < N; x++) {
            u64 val = p0[(size_t)st * N + x];
            if (split) {
                for (int j = 0; j < level + 1; j++) p1Q[(size_t)j * N + x] = val;
                for (int j = 0; j < d->nP; j++) p1P[(size_t)j * N + x] = val;
            } else {
                for (int j = 0; j < level + d->nP + 1; j++) p1Q[(size_t)j * N + x] = val;
            }
        }
        return;
    }
    int index;
    if (level >= alphai + crt * d->alpha) index = d->xalpha[crt] - 2;
    else index = (level - 1) % d->alpha;
    const oc_modup_params *p = d->modup[crt][index];
    int ny = index + 2;
    u64 y[64];
    for (u64 x = 0; x < N; x++) {
        double vi = 0;
        for (int i = 0; i < ny; i++) {
            u64 c = p0[(size_t)(i + st) * N + x];
            p1Q[(size_t)(i + st) * N + x] = c;
            y[i] = oc_mred(c, p->qib_mont[i], p->Q[i], p->mredQ[i]);
            vi += (double)y[i] / (double)p->Q[i];
        }
        u64 v = (u64)vi;
        for (int j = 0; j < st; j++) p1Q[(size_t)j * N + x] = ext_one(p, y, ny, j, v);
        for (int j = d->alpha * crt; j < level + 1; j++) p1Q[(size_t)j * N + x] = ext_one(p, y, ny, j, v);
        if (split) {
            for (int j = 0, u = d->nQ; j < d->nP; j++, u++) p1P[(size_t)j * N + x] = ext_one(p, y, ny, u, v);
        } else {
            for (int u = d->nQ, j = level + 1; j < level + 1 + d->nP; u++, j++) p1Q[(size_t)j * N + x] = ext_one(p, y, ny, u, v);
        }
    }
}

void oc_decompose(const oc_decomposer *d, int level, int crt, const u64 *p0, u64 *p1, u64 N) {
    decompose_impl(d, level, crt, p0, p1, NULL, N, 0);
}
void oc_decompose_and_split(const oc_decomposer *d, int level, int crt, const u64 *p0, u64 *p1Q, u64 *p1P, u64 N) {
    decompose_impl(d, level, crt, p0, p1Q, p1P, N, 1);
}

/* ========================= ring/ring_scaling.go ============================ */

/* DivFloorByLastModulusNTT, :9-34 */
void oc_div_floor_by_last_modulus_ntt(const oc_context *c, u64 *p0, int nlimbs) {
    const u64 N = c->N;
    int level = nlimbs - 1;
    u64 *tmp = (u64 *)malloc(sizeof(u64) * N);
    u64 *last = p0 + (size_t)level * N;
    oc_intt_limb(last, last, N, c->ntt_psi_inv + (size_t)level * N, c->n_inv[level], c->q[level], c->mred[level]);
    for (int i = 0; i < level; i++) {
        oc_ntt_limb(last, tmp, N, c->ntt_psi + (size_t)i * N, c->q[i], c->mred[i], &c->bred[2 * i]);
        u64 *p = p0 + (size_t)i * N;
        u64 qi = c->q[i], rp = c->rescale[(size_t)(level - 1) * c->L + i];
        for (u64 j = 0; j < N; j++) p[j] = oc_mred(p[j] + (qi - tmp[j]), rp, qi, c->mred[i]);
    }
    free(tmp);
}

/* DivFloorByLastModulus, :37-55 */
void oc_div_floor_by_last_modulus(const oc_context *c, u64 *p0, int nlimbs) {
    const u64 N = c->N;
    int level = nlimbs - 1;
    const u64 *last = p0 + (size_t)level * N;
    for (int i = 0; i < level; i++) {
        u64 *p = p0 + (size_t)i * N;
        u64 qi = c->q[i], rp = c->rescale[(size_t)(level - 1) * c->L + i];
        for (u64 j = 0; j < N; j++)
            p[j] = oc_mred(p[j] + (qi - oc_bred_add(last[j], qi, &c->bred[2 * i])), rp, qi, c->mred[i]);
    }
}

/* DivRoundByLastModulusNTT, :72-114 */
void oc_div_round_by_last_modulus_ntt(const oc_context *c, u64 *p0, int nlimbs) {
    const u64 N = c->N;
    int level = nlimbs - 1;
    u64 *tmp = (u64 *)malloc(sizeof(u64) * N);
    u64 *last = p0 + (size_t)level * N;
    oc_intt_limb(last, last, N, c->ntt_psi_inv + (size_t)level * N, c->n_inv[level], c->q[level], c->mred[level]);
    u64 pj = c->q[level];
    u64 phalf = (pj - 1) >> 1;
    for (u64 i = 0; i < N; i++) last[i] = oc_cred(last[i] + phalf, pj);
    for (int i = 0; i < level; i++) {
        u64 *p = p0 + (size_t)i * N;
        u64 qi = c->q[i], rp = c->rescale[(size_t)(level - 1) * c->L + i];
        u64 phalf_neg = qi - oc_bred_add(phalf, qi, &c->bred[2 * i]);
        for (u64 j = 0; j < N; j++) tmp[j] = last[j] + phalf_neg;
        oc_ntt_limb(tmp, tmp, N, c->ntt_psi + (size_t)i * N, qi, c->mred[i], &c->bred[2 * i]);
        for (u64 j = 0; j < N; j++) p[j] = oc_mred(p[j] + (qi - tmp[j]), rp, qi, c->mred[i]);
    }
    free(tmp);
}

/* DivRoundByLastModulus, :117-150 */
void oc_div_round_by_last_modulus(const oc_context *c, u64 *p0, int nlimbs) {
    const u64 N = c->N;
    int level = nlimbs - 1;
    u64 *last = p0 + (size_t)level * N;
    u64 pj = c->q[level];
    u64 phalf = (pj - 1) >> 1;
    for (u64 i = 0; i < N; i++) last[i] = oc_cred(last[i] + phalf, pj);
    for (int i = 0; i < level; i++) {
        u64 *p = p0 + (size_t)i * N;
        u64 qi = c->q[i], rp = c->rescale[(size_t)(level - 1) * c->L + i];
        u64 phalf_neg = qi - oc_bred_add(phalf, qi, &c->bred[2 * i]);
        for (u64 j = 0; j < N; j++)
            p[j] = oc_mred(p[j] + (qi - oc_bred_add(last[j] + phalf_neg, qi, &c->bred[2 * i])), rp, qi, c->mred[i]);
    }
}

/* DivFloorByLastModulusMany (:65-69) / ...ManyNTT (:58-62) */
void oc_div_floor_by_last_modulus_many(const oc_context *c, u64 *p0, int nlimbs, int nb, int ntt) {
    if (ntt) oc_intt_lvl(c, nlimbs - 1, p0, p0);
    for (int k = 0; k < nb; k++) oc_div_floor_by_last_modulus(c, p0, nlimbs - k);
    if (ntt) oc_ntt_lvl(c, nlimbs - nb - 1, p0, p0);
}

/* DivRoundByLastModulusMany (:160-164) / ...ManyNTT (:153-157) */
void oc_div_round_by_last_modulus_many(const oc_context *c, u64 *p0, int nlimbs, int nb, int ntt) {
    if (ntt) oc_intt_lvl(c, nlimbs - 1, p0, p0);
    for (int k = 0; k < nb; k++) oc_div_round_by_last_modulus(c, p0, nlimbs - k);
    if (ntt) oc_ntt_lvl(c, nlimbs - nb - 1, p0, p0);
}

/* ==================== ckks/evaluator.go caller sequences ==================== */

oc_ckks_plan *oc_ckks_plan_new(const oc_context *cQ, const oc_context *cP) {
    oc_ckks_plan *p = (oc_ckks_plan *)calloc(1, sizeof(*p));
    p->cQ = cQ; p->cP = cP;
    p->bext = oc_bext_new(cQ, cP);                              /* ckks/evaluator.go:95 */
    p->dec = oc_decomposer_new(cQ->q, cQ->L, cP->q, cP->L);     /* :96 */
    p->alpha = cP->L;
    return p;
}
void oc_ckks_plan_free(oc_ckks_plan *p) {
    if (!p) return;
    oc_bext_free(p->bext); oc_decomposer_free(p->dec); free(p);
}

/* decomposeAndSplitNTT, ckks/evaluator.go:1561-1591 */
static void decompose_and_split_ntt(oc_ckks_plan *p, int level, int beta, const u64 *c2ntt, const u64 *c2inv,
                                    u64 *c2QiQ, u64 *c2QiP) {
    const oc_context *cQ = p->cQ, *cP = p->cP;
    const u64 N = cQ->N;
    oc_decompose_and_split(p->dec, level, beta, c2inv, c2QiQ, c2QiP, N);
    int st = beta * p->alpha, ed = st + p->dec->xalpha[beta];
    for (int x = 0; x <= level; x++) {
        if (st <= x && x < ed) memcpy(c2QiQ + (size_t)x * N, c2ntt + (size_t)x * N, sizeof(u64) * N);
        else oc_ntt_limb(c2QiQ + (size_t)x * N, c2QiQ + (size_t)x * N, N, cQ->ntt_psi + (size_t)x * N, cQ->q[x],
                         cQ->mred[x], &cQ->bred[2 * x]);
    }
    oc_ntt_lvl(cP, cP->L - 1, c2QiP, c2QiP);
}

/* switchKeysInPlace, ckks/evaluator.go:1475-1558 */
void oc_ckks_switch_keys(oc_ckks_plan *p, int level, const u64 *cx, const u64 *evk, u64 *p0, u64 *p1) {
    const oc_context *cQ = p->cQ, *cP = p->cP;
    const u64 N = cQ->N;
    const int nQ = cQ->L, nP = cP->L, nQP = nQ + nP;
    size_t szQ = (size_t)nQ * N, szP = (size_t)nP * N;
    u64 *c2QiQ = (u64 *)calloc(szQ, 8), *c2QiP = (u64 *)calloc(szP, 8);
    u64 *pool2P = (u64 *)calloc(szP, 8), *pool3P = (u64 *)calloc(szP, 8);
    u64 *c2 = (u64 *)calloc(szQ, 8);
    memset(p0, 0, sizeof(u64) * (size_t)(level + 1) * N);        /* :1481-1483 poolQ[i].Zero() */
    memset(p1, 0, sizeof(u64) * (size_t)(level + 1) * N);
    oc_intt_lvl(cQ, level, cx, c2);                               /* :1503 */
    int reduce = 0;
    int alpha = p->alpha;
    int beta = (level + 1 + alpha - 1) / alpha;                   /* :1508 */
    for (int i = 0; i < beta; i++) {
        decompose_and_split_ntt(p, level, i, cx, c2, c2QiQ, c2QiP);
        const u64 *k0 = evk + ((size_t)i * 2 + 0) * nQP * N;
        const u64 *k1 = evk + ((size_t)i * 2 + 1) * nQP * N;
        oc_ewise(cQ, OC_MUL_MONT_AND_ADD_NOMOD, level, k0, c2QiQ, p0, NULL);   /* :1515 */
        oc_ewise(cQ, OC_MUL_MONT_AND_ADD_NOMOD, level, k1, c2QiQ, p1, NULL);   /* :1516 */
        for (int j = 0, ki = nQ; j < nP; j++, ki++) {                            /* :1519-1534 */
            u64 pj = cP->q[j], qinv = cP->mred[j];
            const u64 *key0 = k0 + (size_t)ki * N, *key1 = k1 + (size_t)ki * N;
            const u64 *ct = c2QiP + (size_t)j * N;
            u64 *p2 = pool2P + (size_t)j * N, *p3 = pool3P + (size_t)j * N;
            for (u64 y = 0; y < N; y++) {
                p2[y] += oc_mred(key0[y], ct[y], pj, qinv);
                p3[y] += oc_mred(key1[y], ct[y], pj, qinv);
            }
        }
        if ((reduce & 7) == 1) {                                                 /* :1536-1541 */
            oc_ewise(cQ, OC_REDUCE, level, p0, NULL, p0, NULL);
            oc_ewise(cQ, OC_REDUCE, level, p1, NULL, p1, NULL);
            oc_ewise(cP, OC_REDUCE, nP - 1, pool2P, NULL, pool2P, NULL);
            oc_ewise(cP, OC_REDUCE, nP - 1, pool3P, NULL, pool3P, NULL);
        }
        reduce++;
    }
    if (((reduce - 1) & 7) != 1) {                                               /* :1547-1552 */
        oc_ewise(cQ, OC_REDUCE, level, p0, NULL, p0, NULL);
        oc_ewise(cQ, OC_REDUCE, level, p1, NULL, p1, NULL);
        oc_ewise(cP, OC_REDUCE, nP - 1, pool2P, NULL, pool2P, NULL);
        oc_ewise(cP, OC_REDUCE, nP - 1, pool3P, NULL, pool3P, NULL);
    }
    oc_moddown_split_ntt_pq(p->bext, level, p0, pool2P, p0);                     /* :1556 */
    oc_moddown_split_ntt_pq(p->bext, level, p1, pool3P, p1);                     /* :1557 */
    free(c2QiQ); free(c2QiP); free(pool2P); free(pool3P); free(c2);
}

/* The element loops of ckks.Evaluator's constant-by-ciphertext methods (test infrastructure like the rest of this file):
 * AddConst ckks/evaluator.go:429-445 (op 0: CRed(x + s)), MultByConst :712-730 and MultByi / DivByi :765-779, :814-828 (op 1:
 * MRed(x, s)), MultByConstAndAdd :588-606 (op 2: CRed(out + MRed(x, s))); s = lo[i] for the coefficients below N/2, hi[i] for the rest.
 * in, out: [level+1][N]. */
void oc_half_scalar_op(const oc_context *c, int op, int level, const u64 *in, const u64 *lo, const u64 *hi, u64 *out) {
    const u64 N = c->N;
    for (int i = 0; i <= level; i++) {
        const u64 qi = c->q[i], qinv = c->mred[i];
        const u64 *p0tmp = in + (size_t)i * N;
        u64 *p1tmp = out + (size_t)i * N;
        for (u64 j = 0; j < N; j++) {
            const u64 sc = j < (N >> 1) ? lo[i] : hi[i];
            if (op == 0) p1tmp[j] = oc_cred(p0tmp[j] + sc, qi);
            else if (op == 1) p1tmp[j] = oc_mred(p0tmp[j], sc, qi, qinv);
            else p1tmp[j] = oc_cred(p1tmp[j] + oc_mred(p0tmp[j], sc, qi, qinv), qi);
        }
    }
}

/* bfv.evaluator.switchKeys, bfv/evaluator.go:736-812 (test infrastructure like the rest of this file).  cx: [nQ][N] coefficient domain;
 * evk: [beta][2][nQ+nP][N]; p0, p1: [nQ][N] coefficient domain.  The joined context Q||P of the reference (contextKeys) is the pair
 * (cQ, cP) here: the tables of a modulus depend on the modulus and N only. */
void oc_bfv_switch_keys(oc_ckks_plan *p, const u64 *cx, const u64 *evk, u64 *p0out, u64 *p1out) {
    const oc_context *cQ = p->cQ, *cP = p->cP;
    const u64 N = cQ->N;
    const int nQ = cQ->L, nP = cP->L, nQP = nQ + nP;
    const int level = nQ - 1;                                                  /* :743 */
    const size_t szQP = (size_t)nQP * N;
    u64 *c2Qi = (u64 *)calloc(szQP, 8), *c2 = (u64 *)calloc((size_t)nQ * N, 8);
    u64 *p0 = (u64 *)calloc(szQP, 8), *p1 = (u64 *)calloc(szQP, 8);          /* :745-747 keyswitchpool[i].Zero() */
    u64 *c2QiNtt = (u64 *)malloc(N * 8);                                      /* :758 */
    oc_ntt_lvl(cQ, level, cx, c2);                                            /* :753 */
    int reduce = 0;
    const int alpha = p->alpha;
    const int beta = (nQ + alpha - 1) / alpha;                                 /* params.beta */
    for (int i = 0; i < beta; i++) {
        const int p0idxst = i * alpha;
        int p0idxed = p0idxst + p->dec->xalpha[i];
        oc_decompose(p->dec, level, i, cx, c2Qi, N);                          /* :767 */
        for (int x = 0; x < nQP; x++) {
            const oc_context *c = x < nQ ? cQ : cP;
            const int xi = x < nQ ? x : x - nQ;
            const u64 qi = c->q[xi], qinv = c->mred[xi];
            if (p0idxst <= x && x < p0idxed) {
                memcpy(c2QiNtt, c2 + (size_t)x * N, N * 8);                   /* :775-779 */
            } else {
                oc_ntt_limb(c2Qi + (size_t)x * N, c2QiNtt, N, c->ntt_psi + (size_t)xi * N, qi, qinv, c->bred + 2 * xi);   /* :781 */
            }
            const u64 *key0 = evk + (((size_t)i * 2 + 0) * nQP + x) * N, *key1 = evk + (((size_t)i * 2 + 1) * nQP + x) * N;
            u64 *p2tmp = p0 + (size_t)x * N, *p3tmp = p1 + (size_t)x * N;
            for (u64 y = 0; y < N; y++) {                                     /* :789-792 */
                p2tmp[y] += oc_mred(key0[y], c2QiNtt[y], qi, qinv);
                p3tmp[y] += oc_mred(key1[y], c2QiNtt[y], qi, qinv);
            }
        }
        if ((reduce & 7) == 7) {                                              /* :795-798 */
            oc_ewise(cQ, OC_REDUCE, level, p0, NULL, p0, NULL);
            oc_ewise(cQ, OC_REDUCE, level, p1, NULL, p1, NULL);
            oc_ewise(cP, OC_REDUCE, nP - 1, p0 + (size_t)nQ * N, NULL, p0 + (size_t)nQ * N, NULL);
            oc_ewise(cP, OC_REDUCE, nP - 1, p1 + (size_t)nQ * N, NULL, p1 + (size_t)nQ * N, NULL);
        }
        reduce++;
    }
    if (((reduce - 1) & 7) != 7) {                                            /* :803-806 */
        oc_ewise(cQ, OC_REDUCE, level, p0, NULL, p0, NULL);
        oc_ewise(cQ, OC_REDUCE, level, p1, NULL, p1, NULL);
        oc_ewise(cP, OC_REDUCE, nP - 1, p0 + (size_t)nQ * N, NULL, p0 + (size_t)nQ * N, NULL);
        oc_ewise(cP, OC_REDUCE, nP - 1, p1 + (size_t)nQ * N, NULL, p1 + (size_t)nQ * N, NULL);
    }
    oc_intt_lvl(cQ, level, p0, p0);                                           /* :808-809: contextKeys.InvNTT, limb by limb */
    oc_intt_lvl(cQ, level, p1, p1);
    oc_intt_lvl(cP, nP - 1, p0 + (size_t)nQ * N, p0 + (size_t)nQ * N);
    oc_intt_lvl(cP, nP - 1, p1 + (size_t)nQ * N, p1 + (size_t)nQ * N);
    oc_moddown_pq(p->bext, level, p0, p0out);                                 /* :811 */
    oc_moddown_pq(p->bext, level, p1, p1out);                                 /* :812 */
    free(c2Qi); free(c2); free(p0); free(p1); free(c2QiNtt);
}

/* bfv.evaluator.relinearize for a degree-2 ciphertext, bfv/evaluator.go:480-501: ct = [3][nQ][N], out = [2][nQ][N] */
void oc_bfv_relinearize(oc_ckks_plan *p, const u64 *ct, const u64 *evk, u64 *out) {
    const oc_context *cQ = p->cQ;
    const size_t sz = (size_t)cQ->L * cQ->N;
    u64 *p0 = (u64 *)malloc(sz * 8), *p1 = (u64 *)malloc(sz * 8);
    oc_bfv_switch_keys(p, ct + 2 * sz, evk, p0, p1);                          /* :493 (deg = 2, evakey[0]) */
    oc_ewise(cQ, OC_ADD, cQ->L - 1, ct, p0, out, NULL);                       /* :494 */
    oc_ewise(cQ, OC_ADD, cQ->L - 1, ct + sz, p1, out + sz, NULL);             /* :495 */
    free(p0); free(p1);
}

/* bfv.evaluator.permute, bfv/evaluator.go:711-735 (the body of RotateRows :670-681 and of RotateColumns with the key of that rotation
 * :590-592): ct = [2][nQ][N] coefficient domain, evk as for oc_bfv_switch_keys, out = [2][nQ][N].  oc_permute is declared below. */
void oc_permute(const oc_context *c, const u64 *in, u64 gen, u64 *out);
void oc_bfv_permute(oc_ckks_plan *p, const u64 *ct, u64 gen, const u64 *evk, u64 *out) {
    const oc_context *cQ = p->cQ;
    const size_t sz = (size_t)cQ->L * cQ->N;
    u64 *el0 = (u64 *)malloc(sz * 8), *el1 = (u64 *)malloc(sz * 8), *p0 = (u64 *)malloc(sz * 8), *p1 = (u64 *)malloc(sz * 8);
    oc_permute(cQ, ct, gen, el0);                                             /* :723 */
    oc_permute(cQ, ct + sz, gen, el1);                                        /* :724 */
    oc_bfv_switch_keys(p, el1, evk, p0, p1);                                  /* :729 */
    oc_ewise(cQ, OC_ADD, cQ->L - 1, el0, p0, out, NULL);                      /* :731 */
    memcpy(out + sz, p1, sz * 8);                                             /* :732 Copy */
    free(el0); free(el1); free(p0); free(p1);
}

/* MulRelin, ckks/evaluator.go:1016-1133 (ct x ct, regular case, with evaluation key) */
void oc_ckks_mulrelin(oc_ckks_plan *p, int level, const u64 *ct0, const u64 *ct1, const u64 *evk, u64 *out) {
    const oc_context *cQ = p->cQ;
    const u64 N = cQ->N;
    size_t sz = (size_t)(level + 1) * N;
    u64 *c00 = (u64 *)malloc(sz * 8), *c01 = (u64 *)malloc(sz * 8);
    u64 *c0 = (u64 *)malloc(sz * 8), *c1 = (u64 *)malloc(sz * 8), *c2 = (u64 *)malloc(sz * 8);
    u64 *q1 = (u64 *)malloc(sz * 8), *q2 = (u64 *)malloc(sz * 8);
    oc_ewise(cQ, OC_MFORM, level, ct0, NULL, c00, NULL);                      /* :1080 */
    oc_ewise(cQ, OC_MFORM, level, ct0 + sz, NULL, c01, NULL);                 /* :1081 */
    oc_ewise(cQ, OC_MUL_MONT, level, c00, ct1, c0, NULL);                     /* :1092 */
    oc_ewise(cQ, OC_MUL_MONT, level, c00, ct1 + sz, c1, NULL);                /* :1093 */
    oc_ewise(cQ, OC_MUL_MONT_AND_ADD, level, c01, ct1, c1, NULL);             /* :1094 */
    oc_ewise(cQ, OC_MUL_MONT, level, c01, ct1 + sz, c2, NULL);                /* :1095 */
    oc_ckks_switch_keys(p, level, c2, evk, q1, q2);                           /* :1101 */
    oc_ewise(cQ, OC_ADD, level, c0, q1, out, NULL);                           /* :1103 */
    oc_ewise(cQ, OC_ADD, level, c1, q2, out + sz, NULL);                      /* :1104 */
    free(c00); free(c01); free(c0); free(c1); free(c2); free(q1); free(q2);
}

/* MulRelin with evakey == nil, ckks/evaluator.go:1038-1111: the degree-2 result (c0, c1, c2).  squaring != 0 takes the
 * el0 == el1 branch (:1083-1088: c1 = 2 * c0 * c1 through AddLvl), otherwise the regular one (:1090-1096). */
void oc_ckks_mul_norelin(oc_ckks_plan *p, int level, const u64 *ct0, const u64 *ct1, int squaring, u64 *out) {
    const oc_context *cQ = p->cQ;
    const u64 N = cQ->N;
    size_t sz = (size_t)(level + 1) * N;
    u64 *c00 = (u64 *)malloc(sz * 8), *c01 = (u64 *)malloc(sz * 8);
    u64 *c0 = out, *c1 = out + sz, *c2 = out + 2 * sz;
    oc_ewise(cQ, OC_MFORM, level, ct0, NULL, c00, NULL);                      /* :1080 */
    oc_ewise(cQ, OC_MFORM, level, ct0 + sz, NULL, c01, NULL);                 /* :1081 */
    if (squaring) {
        oc_ewise(cQ, OC_MUL_MONT, level, c00, ct1, c0, NULL);                 /* :1085 */
        oc_ewise(cQ, OC_MUL_MONT, level, c00, ct1 + sz, c1, NULL);            /* :1086 */
        oc_ewise(cQ, OC_ADD, level, c1, c1, c1, NULL);                        /* :1087 */
        oc_ewise(cQ, OC_MUL_MONT, level, c01, ct1 + sz, c2, NULL);            /* :1088 */
    } else {
        oc_ewise(cQ, OC_MUL_MONT, level, c00, ct1, c0, NULL);                 /* :1092 */
        oc_ewise(cQ, OC_MUL_MONT, level, c00, ct1 + sz, c1, NULL);            /* :1093 */
        oc_ewise(cQ, OC_MUL_MONT_AND_ADD, level, c01, ct1, c1, NULL);         /* :1094 */
        oc_ewise(cQ, OC_MUL_MONT, level, c01, ct1 + sz, c2, NULL);            /* :1095 */
    }
    free(c00); free(c01);
}

/* MulRelin, plaintext x ciphertext branch, ckks/evaluator.go:1113-1131: pt = the degree-0 operand's value[0] */
void oc_ckks_mul_plain(oc_ckks_plan *p, int level, const u64 *pt, const u64 *ct, u64 *out) {
    const oc_context *cQ = p->cQ;
    size_t sz = (size_t)(level + 1) * cQ->N;
    u64 *c00 = (u64 *)calloc(sz, 8);                                          /* :1127 c00.Zero() */
    oc_ewise(cQ, OC_MFORM, level, pt, NULL, c00, NULL);                       /* :1129 */
    oc_ewise(cQ, OC_MUL_MONT, level, c00, ct, out, NULL);                     /* :1130 */
    oc_ewise(cQ, OC_MUL_MONT, level, c00, ct + sz, out + sz, NULL);           /* :1131 */
    free(c00);
}

/* pkEncryptor.encrypt, the branch through the special primes, after the sampling (ckks/encryptor.go:205-234).
 * cQP: the context over Q||P.  u = SampleTernaryMontgomeryNTT over QP (:206), pk = (pk0, pk1) over QP, e0 / e1 = the
 * residues SampleAndAdd adds (ring/gaussianSampler.go:254-274: CRed(x + e) per coefficient, e in [0, q]), pt over
 * Q[0..level] in the NTT domain.  ct = [2][level+1][N].  The reference transforms with Context.NTT (:229-230), which walks every
 * modulus of contextQ: for level < |Q|-1 Go panics there; limbs 0..level are transformed here. */
void oc_ckks_encrypt_pk(oc_ckks_plan *p, const oc_context *cQP, int level, const u64 *u, const u64 *pk0, const u64 *pk1,
                        const u64 *e0, const u64 *e1, const u64 *pt, u64 *ct) {
    const oc_context *cQ = p->cQ;
    const u64 N = cQ->N;
    const int lQP = cQP->L - 1;
    size_t sz = (size_t)(level + 1) * N, szQP = (size_t)cQP->L * N;
    u64 *p0 = (u64 *)malloc(szQP * 8), *p1 = (u64 *)malloc(szQP * 8);
    oc_ewise(cQP, OC_MUL_MONT, lQP, u, pk0, p0, NULL);                        /* :209 */
    oc_ewise(cQP, OC_MUL_MONT, lQP, u, pk1, p1, NULL);                        /* :211 */
    oc_intt_lvl(cQP, lQP, p0, p0);                                            /* :214 */
    oc_intt_lvl(cQP, lQP, p1, p1);                                            /* :215 */
    oc_ewise(cQP, OC_ADD, lQP, p0, e0, p0, NULL);                             /* :218 SampleAndAdd */
    oc_ewise(cQP, OC_ADD, lQP, p1, e1, p1, NULL);                             /* :220 */
    oc_moddown_pq(p->bext, level, p0, ct);                                    /* :223 */
    oc_moddown_pq(p->bext, level, p1, ct + sz);                               /* :226 */
    oc_ntt_lvl(cQ, level, ct, ct);                                            /* :229 */
    oc_ntt_lvl(cQ, level, ct + sz, ct + sz);                                  /* :230 */
    oc_ewise(cQ, OC_ADD, level, ct, pt, ct, NULL);                            /* :234 */
    free(p0); free(p1);
}

/* decryptor.Decrypt, ckks/decryptor.go:53-78: Horner evaluation at the secret key (NTT + Montgomery form).
 * ct = [degree+1][level+1][N] */
void oc_ckks_decrypt(oc_ckks_plan *p, int level, const u64 *ct, int degree, const u64 *sk, u64 *pt) {
    const oc_context *cQ = p->cQ;
    size_t sz = (size_t)(level + 1) * cQ->N;
    oc_ewise(cQ, OC_COPY, level, ct + (size_t)degree * sz, NULL, pt, NULL);   /* :61 */
    for (int i = degree; i > 0; i--) {                                         /* :65 */
        oc_ewise(cQ, OC_MUL_MONT, level, pt, sk, pt, NULL);                   /* :67 */
        oc_ewise(cQ, OC_ADD, level, pt, ct + (size_t)(i - 1) * sz, pt, NULL); /* :68 */
        if ((i & 7) == 7) oc_ewise(cQ, OC_REDUCE, level, pt, NULL, pt, NULL); /* :70-72 */
    }
    if ((degree & 7) != 7) oc_ewise(cQ, OC_REDUCE, level, pt, NULL, pt, NULL); /* :75-77 */
}

/* ==================== bfv/evaluator.go caller sequence ==================== */

/* tensorAndRescale, bfv/evaluator.go:278-464 (both operands of degree 1, ct0 != ct1) */
void oc_bfv_mul(oc_bext *b, u64 t, const u64 *phalf_q, const u64 *phalf_qm, const u64 *ct0, const u64 *ct1, u64 *out) {
    const oc_context *cQ = b->cQ, *cM = b->cP;
    const u64 N = cQ->N;
    const int nQ = cQ->L, nM = cM->L;
    const int lQ = nQ - 1, lM = nM - 1;
    size_t sQ = (size_t)nQ * N, sM = (size_t)nM * N;
    u64 *c0Q1 = (u64 *)malloc(2 * sQ * 8), *c0Q2 = (u64 *)malloc(2 * sM * 8);
    u64 *c1Q1 = (u64 *)malloc(2 * sQ * 8), *c1Q2 = (u64 *)malloc(2 * sM * 8);
    u64 *c2Q1 = (u64 *)malloc(3 * sQ * 8), *c2Q2 = (u64 *)malloc(3 * sM * 8);
    u64 *c00Q = (u64 *)malloc(sQ * 8), *c00M = (u64 *)malloc(sM * 8), *c01Q = (u64 *)malloc(sQ * 8), *c01M = (u64 *)malloc(sM * 8);
    for (int i = 0; i < 2; i++) {                                            /* :298-313 */
        oc_modup_split_qp(b, lQ, ct0 + i * sQ, c0Q2 + i * sM);
        oc_ntt_lvl(cQ, lQ, ct0 + i * sQ, c0Q1 + i * sQ);
        oc_ntt_lvl(cM, lM, c0Q2 + i * sM, c0Q2 + i * sM);
        oc_modup_split_qp(b, lQ, ct1 + i * sQ, c1Q2 + i * sM);
        oc_ntt_lvl(cQ, lQ, ct1 + i * sQ, c1Q1 + i * sQ);
        oc_ntt_lvl(cM, lM, c1Q2 + i * sM, c1Q2 + i * sM);
    }
    oc_ewise(cQ, OC_MFORM, lQ, c0Q1, NULL, c00Q, NULL);                       /* :327-331 */
    oc_ewise(cM, OC_MFORM, lM, c0Q2, NULL, c00M, NULL);
    oc_ewise(cQ, OC_MFORM, lQ, c0Q1 + sQ, NULL, c01Q, NULL);
    oc_ewise(cM, OC_MFORM, lM, c0Q2 + sM, NULL, c01M, NULL);
    oc_ewise(cQ, OC_MUL_MONT, lQ, c00Q, c1Q1, c2Q1, NULL);                    /* :354-367 */
    oc_ewise(cM, OC_MUL_MONT, lM, c00M, c1Q2, c2Q2, NULL);
    oc_ewise(cQ, OC_MUL_MONT, lQ, c00Q, c1Q1 + sQ, c2Q1 + sQ, NULL);
    oc_ewise(cM, OC_MUL_MONT, lM, c00M, c1Q2 + sM, c2Q2 + sM, NULL);
    oc_ewise(cQ, OC_MUL_MONT_AND_ADD_NOMOD, lQ, c01Q, c1Q1, c2Q1 + sQ, NULL);
    oc_ewise(cM, OC_MUL_MONT_AND_ADD_NOMOD, lM, c01M, c1Q2, c2Q2 + sM, NULL);
    oc_ewise(cQ, OC_MUL_MONT, lQ, c01Q, c1Q1 + sQ, c2Q1 + 2 * sQ, NULL);
    oc_ewise(cM, OC_MUL_MONT, lM, c01M, c1Q2 + sM, c2Q2 + 2 * sM, NULL);
    for (int i = 0; i < 3; i++) {                                            /* :423-463 */
        u64 *q1 = c2Q1 + i * sQ, *q2 = c2Q2 + i * sM, *o = out + i * sQ;
        oc_intt_lvl(cQ, lQ, q1, q1);
        oc_intt_lvl(cM, lM, q2, q2);
        oc_moddown_split_qp(b, lQ, lM, q1, q2, q2);
        oc_ewise(cM, OC_ADD_SCALAR_LIMBS, lM, q2, NULL, q2, phalf_qm);
        oc_modup_split_pq(b, lM, q2, o);
        oc_ewise(cQ, OC_SUB_SCALAR_LIMBS, lQ, o, NULL, o, phalf_q);
        oc_ewise(cQ, OC_MUL_SCALAR, lQ, o, NULL, o, &t);
    }
    free(c0Q1); free(c0Q2); free(c1Q1); free(c1Q2); free(c2Q1); free(c2Q2); free(c00Q); free(c00M); free(c01Q); free(c01M);
}

/* tensorAndRescale with ct0 == ct1, bfv/evaluator.go:278-464: the operand is lifted and transformed once (:298-305, :306 skips the
 * second loop) and the tensor is the squaring case of :334-349 (c1 = 2 c0[0] c0[1] by AddNoMod). */
void oc_bfv_square(oc_bext *b, u64 t, const u64 *phalf_q, const u64 *phalf_qm, const u64 *ct0, u64 *out) {
    const oc_context *cQ = b->cQ, *cM = b->cP;
    const u64 N = cQ->N;
    const int nQ = cQ->L, nM = cM->L;
    const int lQ = nQ - 1, lM = nM - 1;
    size_t sQ = (size_t)nQ * N, sM = (size_t)nM * N;
    u64 *c0Q1 = (u64 *)malloc(2 * sQ * 8), *c0Q2 = (u64 *)malloc(2 * sM * 8);
    u64 *c2Q1 = (u64 *)malloc(3 * sQ * 8), *c2Q2 = (u64 *)malloc(3 * sM * 8);
    u64 *c00Q = (u64 *)malloc(sQ * 8), *c00M = (u64 *)malloc(sM * 8), *c01Q = (u64 *)malloc(sQ * 8), *c01M = (u64 *)malloc(sM * 8);
    for (int i = 0; i < 2; i++) {                                            /* :298-304 */
        oc_modup_split_qp(b, lQ, ct0 + i * sQ, c0Q2 + i * sM);
        oc_ntt_lvl(cQ, lQ, ct0 + i * sQ, c0Q1 + i * sQ);
        oc_ntt_lvl(cM, lM, c0Q2 + i * sM, c0Q2 + i * sM);
    }
    oc_ewise(cQ, OC_MFORM, lQ, c0Q1, NULL, c00Q, NULL);                       /* :327-331 */
    oc_ewise(cM, OC_MFORM, lM, c0Q2, NULL, c00M, NULL);
    oc_ewise(cQ, OC_MFORM, lQ, c0Q1 + sQ, NULL, c01Q, NULL);
    oc_ewise(cM, OC_MFORM, lM, c0Q2 + sM, NULL, c01M, NULL);
    oc_ewise(cQ, OC_MUL_MONT, lQ, c00Q, c0Q1, c2Q1, NULL);                    /* :337-338 */
    oc_ewise(cM, OC_MUL_MONT, lM, c00M, c0Q2, c2Q2, NULL);
    oc_ewise(cQ, OC_MUL_MONT, lQ, c00Q, c0Q1 + sQ, c2Q1 + sQ, NULL);          /* :341-342 */
    oc_ewise(cM, OC_MUL_MONT, lM, c00M, c0Q2 + sM, c2Q2 + sM, NULL);
    oc_ewise(cQ, OC_ADD_NOMOD, lQ, c2Q1 + sQ, c2Q1 + sQ, c2Q1 + sQ, NULL);    /* :344-345 */
    oc_ewise(cM, OC_ADD_NOMOD, lM, c2Q2 + sM, c2Q2 + sM, c2Q2 + sM, NULL);
    oc_ewise(cQ, OC_MUL_MONT, lQ, c01Q, c0Q1 + sQ, c2Q1 + 2 * sQ, NULL);      /* :348-349 */
    oc_ewise(cM, OC_MUL_MONT, lM, c01M, c0Q2 + sM, c2Q2 + 2 * sM, NULL);
    for (int i = 0; i < 3; i++) {                                            /* :423-463 */
        u64 *q1 = c2Q1 + i * sQ, *q2 = c2Q2 + i * sM, *o = out + i * sQ;
        oc_intt_lvl(cQ, lQ, q1, q1);
        oc_intt_lvl(cM, lM, q2, q2);
        oc_moddown_split_qp(b, lQ, lM, q1, q2, q2);
        oc_ewise(cM, OC_ADD_SCALAR_LIMBS, lM, q2, NULL, q2, phalf_qm);
        oc_modup_split_pq(b, lM, q2, o);
        oc_ewise(cQ, OC_SUB_SCALAR_LIMBS, lQ, o, NULL, o, phalf_q);
        oc_ewise(cQ, OC_MUL_SCALAR, lQ, o, NULL, o, &t);
    }
    free(c0Q1); free(c0Q2); free(c2Q1); free(c2Q2); free(c00Q); free(c00M); free(c01Q); free(c01M);
}

/* ========================= ring/ring_galois.go ============================ */
static int log2_u64(u64 n) { int l = 0; while (((u64)1 << l) < n) l++; return l; }

/* PermuteNTTIndex, ring_galois.go:29-50 */
void oc_permute_ntt_index(u64 gen, u64 power, u64 N, u64 *index) {
    u64 gen_pow = oc_mod_exp(gen, power, 2 * N);
    u64 logN = (u64)log2_u64(N), mask = (N << 1) - 1;
    for (u64 i = 0; i < N; i++) {
        u64 tmp1 = 2 * oc_bit_reverse64(i, logN) + 1;
        u64 tmp2 = ((gen_pow * tmp1 & mask) - 1) >> 1;
        index[i] = oc_bit_reverse64(tmp2, logN);
    }
}

/* PermuteNTT, ring_galois.go:55-84 (not in place) */
void oc_permute_ntt(const u64 *in, u64 gen, u64 *out, int limbs, u64 N) {
    u64 logN = (u64)log2_u64(N), mask = (N << 1) - 1;
    for (u64 j = 0; j < N; j++) {
        u64 tmp1 = 2 * oc_bit_reverse64(j, logN) + 1;
        u64 tmp2 = ((gen * tmp1 & mask) - 1) >> 1;
        u64 idx = oc_bit_reverse64(tmp2, logN);
        for (int i = 0; i < limbs; i++) out[(size_t)i * N + j] = in[(size_t)i * N + idx];
    }
}

/* PermuteNTTWithIndex, ring_galois.go:89-101 */
void oc_permute_ntt_with_index(const u64 *in, const u64 *index, u64 *out, int limbs, u64 N) {
    for (u64 j = 0; j < N; j++)
        for (int i = 0; i < limbs; i++) out[(size_t)i * N + j] = in[(size_t)i * N + index[j]];
}

/* Context.Permute, ring_galois.go:106-127 (coefficient domain; 0 maps to q when the sign flips, as in Go) */
void oc_permute(const oc_context *c, const u64 *in, u64 gen, u64 *out) {
    const u64 N = c->N, mask = N - 1;
    u64 logN = (u64)log2_u64(N);
    for (u64 i = 0; i < N; i++) {
        u64 raw = i * gen, index = raw & mask, tmp = (raw >> logN) & 1;
        for (int j = 0; j < c->L; j++) {
            u64 qi = c->q[j], x = in[(size_t)j * N + i];
            out[(size_t)j * N + index] = x * (tmp ^ 1) | (qi - x) * tmp;
        }
    }
}


/* ---- ckks rotations -------------------------------------------------------------------- */
/* evaluator.permuteNTT, ckks/evaluator.go:1448-1468 (ct0 != ctOut branch) */
void oc_ckks_permute_ntt(oc_ckks_plan *p, int level, const u64 *ct, u64 gen, const u64 *evk, u64 *out) {
    const oc_context *cQ = p->cQ;
    const u64 N = cQ->N;
    const size_t s = (size_t)(level + 1) * N;
    u64 *el0 = (u64 *)calloc(s, 8), *el1 = (u64 *)calloc(s, 8), *p0 = (u64 *)calloc(s, 8), *p1 = (u64 *)calloc(s, 8);
    oc_permute_ntt(ct, gen, el0, level + 1, N);                                    /* :1458 */
    oc_permute_ntt(ct + s, gen, el1, level + 1, N);                                /* :1459 */
    oc_ckks_switch_keys(p, level, el1, evk, p0, p1);                               /* :1464 */
    oc_ewise(cQ, OC_ADD, level, el0, p0, out, NULL);                               /* :1466 */
    memcpy(out + s, p1, s * 8);                                                    /* :1467 CopyLvl */
    free(el0); free(el1); free(p0); free(p1);
}

/* RotateHoisted (:1252-1289) with switchKeyHoisted (:1292-1391) per rotation */
void oc_ckks_rotate_hoisted(oc_ckks_plan *p, int level, const u64 *ct, int n_rot, const u64 *gens,
                            const u64 *const *evks, u64 *out) {
    const oc_context *cQ = p->cQ, *cP = p->cP;
    const u64 N = cQ->N;
    const int nQ = cQ->L, nP = cP->L, nQP = nQ + nP;
    const size_t s = (size_t)(level + 1) * N, szQ = (size_t)nQ * N, szP = (size_t)nP * N;
    const int alpha = p->alpha;
    const int beta = (level + 1 + alpha - 1) / alpha;                              /* :1263 */
    const u64 *c2ntt = ct + s;
    u64 *c2inv = (u64 *)calloc(szQ, 8);
    oc_intt_lvl(cQ, level, c2ntt, c2inv);                                          /* :1260 */
    u64 *decQ = (u64 *)calloc(szQ * beta, 8), *decP = (u64 *)calloc(szP * beta, 8);
    for (int i = 0; i < beta; i++)                                                 /* :1268-1272 */
        decompose_and_split_ntt(p, level, i, c2ntt, c2inv, decQ + szQ * i, decP + szP * i);
    u64 *permQ = (u64 *)calloc(szQ, 8), *permP = (u64 *)calloc(szP, 8);
    u64 *pool2Q = (u64 *)calloc(szQ, 8), *pool3Q = (u64 *)calloc(szQ, 8);
    u64 *pool2P = (u64 *)calloc(szP, 8), *pool3P = (u64 *)calloc(szP, 8);
    for (int r = 0; r < n_rot; r++) {
        const u64 gen = gens[r];
        const u64 *evk = evks[r];
        u64 *o0 = out + (size_t)r * 2 * s, *o1 = o0 + s;
        oc_permute_ntt(ct, gen, o0, level + 1, N);                                 /* :1314-1315 */
        memset(pool2Q, 0, szQ * 8); memset(pool3Q, 0, szQ * 8);                    /* :1320-1326 */
        memset(pool2P, 0, szP * 8); memset(pool3P, 0, szP * 8);
        int reduce = 0;
        for (int i = 0; i < beta; i++) {
            oc_permute_ntt(decQ + szQ * i, gen, permQ, level + 1, N);              /* :1346 */
            oc_permute_ntt(decP + szP * i, gen, permP, nP, N);                     /* :1347 */
            const u64 *k0 = evk + ((size_t)i * 2 + 0) * nQP * N;
            const u64 *k1 = evk + ((size_t)i * 2 + 1) * nQP * N;
            oc_ewise(cQ, OC_MUL_MONT_AND_ADD_NOMOD, level, k0, permQ, pool2Q, NULL);   /* :1349 */
            oc_ewise(cQ, OC_MUL_MONT_AND_ADD_NOMOD, level, k1, permQ, pool3Q, NULL);   /* :1350 */
            for (int j = 0, ki = nQ; j < nP; j++, ki++) {                            /* :1353-1367 */
                u64 pj = cP->q[j], qinv = cP->mred[j];
                const u64 *key0 = k0 + (size_t)ki * N, *key1 = k1 + (size_t)ki * N;
                const u64 *c2 = permP + (size_t)j * N;
                u64 *p2 = pool2P + (size_t)j * N, *p3 = pool3P + (size_t)j * N;
                for (u64 y = 0; y < N; y++) {
                    p2[y] += oc_mred(key0[y], c2[y], pj, qinv);
                    p3[y] += oc_mred(key1[y], c2[y], pj, qinv);
                }
            }
            if ((reduce & 7) == 1) {                                                 /* :1369-1374 */
                oc_ewise(cQ, OC_REDUCE, level, pool2Q, NULL, pool2Q, NULL);
                oc_ewise(cQ, OC_REDUCE, level, pool3Q, NULL, pool3Q, NULL);
                oc_ewise(cP, OC_REDUCE, nP - 1, pool2P, NULL, pool2P, NULL);
                oc_ewise(cP, OC_REDUCE, nP - 1, pool3P, NULL, pool3P, NULL);
            }
            reduce++;
        }
        if (((reduce - 1) & 7) != 1) {                                               /* :1379-1384 */
            oc_ewise(cQ, OC_REDUCE, level, pool2Q, NULL, pool2Q, NULL);
            oc_ewise(cQ, OC_REDUCE, level, pool3Q, NULL, pool3Q, NULL);
            oc_ewise(cP, OC_REDUCE, nP - 1, pool2P, NULL, pool2P, NULL);
            oc_ewise(cP, OC_REDUCE, nP - 1, pool3P, NULL, pool3P, NULL);
        }
        oc_moddown_split_ntt_pq(p->bext, level, pool2Q, pool2P, pool2Q);             /* :1388 */
        oc_moddown_split_ntt_pq(p->bext, level, pool3Q, pool3P, pool3Q);             /* :1389 */
        oc_ewise(cQ, OC_ADD, level, o0, pool2Q, o0, NULL);                           /* :1391 */
        memcpy(o1, pool3Q, s * 8);                                                   /* :1392 */
    }
    free(c2inv); free(decQ); free(decP); free(permQ); free(permP);
    free(pool2Q); free(pool3Q); free(pool2P); free(pool3P);
}


/* Context.Shift, ring/ring.go:575-580: p2.Coeffs[i] = append(p1.Coeffs[i][n&mask:], p1.Coeffs[i][:n&mask]...), mask = (1 << N) - 1
 * (Go: a shift count >= 64 gives 0, so the mask is all ones for N >= 64).  Returns -1 where the slice expression would panic. */
int oc_shift(const oc_context *c, const u64 *p1, u64 n, u64 *p2) {
    const u64 N = c->N;
    const u64 mask = N >= 64 ? ~(u64)0 : (((u64)1 << N) - 1);
    const u64 m = n & mask;
    if (m > N) return -1;
    u64 *tmp = (u64 *)malloc(sizeof(u64) * N);
    for (int i = 0; i < c->L; i++) {
        const u64 *row = p1 + (size_t)i * N;
        for (u64 j = 0; j < N - m; j++) tmp[j] = row[m + j];
        for (u64 j = 0; j < m; j++) tmp[N - m + j] = row[j];
        memcpy(p2 + (size_t)i * N, tmp, sizeof(u64) * N);
    }
    free(tmp);
    return 0;
}

/* modexpMontgomery, ring/utils.go:39-50 */
static u64 modexp_montgomery(u64 x, u64 e, u64 q, u64 qinv, const u64 u[2]) {
    u64 result = oc_mform(1, q, u);
    for (u64 i = e; i > 0; i >>= 1) {
        if (i & 1) result = oc_mred(result, x, q, qinv);
        x = oc_mred(x, x, q, qinv);
    }
    return result;
}

/* Context.Rotate, ring/ring.go:775-800.  Writes into p1 (p1tmp, p2tmp := p1.Coeffs[i], p1.Coeffs[i], :791); coefficient 0 untouched. */
void oc_rotate(const oc_context *c, u64 *p1, u64 n) {
    const u64 N = c->N;
    n &= N >= 64 ? ~(u64)0 : (((u64)1 << N) - 1);                                   /* :779 */
    for (int i = 0; i < c->L; i++) {
        const u64 qi = c->q[i], qinv = c->mred[i];
        const u64 *u = &c->bred[2 * i];
        u64 root = oc_mred(c->psi_mont[i], c->psi_mont[i], qi, qinv);                 /* :785 */
        root = modexp_montgomery(root, n, qi, qinv, u);                               /* :787 */
        u64 gal = oc_mform(1, qi, u);                                                 /* :789 */
        u64 *row = p1 + (size_t)i * N;
        for (u64 j = 1; j < N; j++) {
            gal = oc_mred(gal, root, qi, qinv);                                       /* :795 */
            row[j] = oc_mred(row[j], gal, qi, qinv);                                  /* :797 */
        }
    }
}

/* Context.MultByMonomial, ring/ring.go:663-727 (through the temporary tmpx, so p1 may equal p2) */
void oc_mult_by_monomial(const oc_context *c, const u64 *p1, u64 monomial_deg, u64 *p2) {
    const u64 N = c->N;
    u64 shift = monomial_deg % (N << 1);                              /* :667 */
    if (shift == 0) {
        for (int i = 0; i < c->L; i++)
            for (u64 j = 0; j < N; j++) p2[(size_t)i * N + j] = p1[(size_t)i * N + j];   /* :669-678 */
        return;
    }
    u64 *tmpx = (u64 *)malloc(sizeof(u64) * (size_t)c->L * N);
    for (int i = 0; i < c->L; i++) {
        const u64 qi = c->q[i];
        for (u64 j = 0; j < N; j++)
            tmpx[(size_t)i * N + j] = shift < N ? p1[(size_t)i * N + j] : qi - p1[(size_t)i * N + j];   /* :684-707 */
    }
    shift %= N;                                                       /* :710 */
    for (int i = 0; i < c->L; i++) {
        const u64 qi = c->q[i];
        for (u64 j = 0; j < shift; j++) p2[(size_t)i * N + j] = qi - tmpx[(size_t)i * N + N - shift + j];   /* :712-717 */
        for (u64 j = shift; j < N; j++) p2[(size_t)i * N + j] = tmpx[(size_t)i * N + j - shift];            /* :719-726 */
    }
    free(tmpx);
}


/* ------------------------------------------------------------------------------------------------
 * Float128 (ring/float128.go): double-double arithmetic, operation for operation.  Compiled with -ffp-contract=off.
 * ---------------------------------------------------------------------------------------------- */
void oc_f128_set_uint53(u64 i, double r[2]) { r[0] = (double)i; r[1] = 0.0; }                     /* :32 */
void oc_f128_set_uint64(u64 i, double r[2]) { r[0] = (double)(i >> 12); r[1] = (double)(i & 0xfff) / 4096.0; }   /* :43 */
u64 oc_f128_to_uint53(const double f[2]) { return (u64)f[0]; }
u64 oc_f128_to_uint64(const double f[2]) {
    /* uint64(f[0]*4096) + uint64(math.Round((f[0]*4096 - float64(that)) + f[1]*4096)); a negative rounded value wraps like
     * Go's amd64 conversion (through int64) */
    const double s = f[0] * 4096.0;
    const u64 t = (u64)s;
    const double r = round((s - (double)t) + f[1] * 4096.0);
    return t + (u64)(int64_t)r;
}
static void f_two_sum(double a, double b, double *s, double *err) {
    *s = a + b;
    const double bb = *s - a;
    *err = (a - (*s - bb)) + (b - bb);
}
static void f_quick_two_sum(double a, double b, double *s, double *err) {
    *s = a + b;
    *err = b - (*s - a);
}
static void f_two_diff(double a, double b, double *s, double *err) {
    *s = a - b;
    const double bb = *s - a;
    *err = (a - (*s - bb)) - (b + bb);
}
static void f_split(double a, double *hi, double *lo) {
    const double temp = 134217729.0 * a;
    *hi = temp - (temp - a);
    *lo = a - *hi;
}
static void f_two_prod(double a, double b, double *p, double *err) {
    double ah, al, bh, bl;
    *p = a * b;
    f_split(a, &ah, &al);
    f_split(b, &bh, &bl);
    *err = ((ah * bh - *p) + ah * bl + al * bh) + al * bl;
}
void oc_f128_add(const double a[2], const double b[2], double f[2]) {
    double s1, s2, t1, t2;
    f_two_sum(a[0], b[0], &s1, &s2);
    f_two_sum(a[1], b[1], &t1, &t2);
    s2 += t1;
    f_quick_two_sum(s1, s2, &s1, &s2);
    s2 += t2;
    f_quick_two_sum(s1, s2, &f[0], &f[1]);
}
void oc_f128_mul(const double a[2], const double b[2], double f[2]) {
    double p1, p2;
    f_two_prod(a[0], b[0], &p1, &p2);
    p2 += a[0] * b[1] + a[1] * b[0];
    f_quick_two_sum(p1, p2, &f[0], &f[1]);
}
void oc_f128_div(const double a[2], const double b[2], double f[2]) {
    double p1, p2, p3, p4, v1, v2;
    const double q1 = a[0] / b[0];
    f_two_prod(q1, b[0], &p1, &p2);
    p2 += q1 * b[1];
    const double t0 = p1 + p2;
    const double t1 = p2 - (t0 - p1);
    f_two_diff(a[0], t0, &p3, &p4);
    f_two_diff(a[1], t1, &v1, &v2);
    p4 += v1;
    f_quick_two_sum(p3, p4, &p3, &p4);
    p4 += v2;
    const double r = (p3 + p4) / b[0];
    const double hi = q1 + r;
    f[1] = r - (hi - q1);
    f[0] = hi;
}

/* NewSimpleScaler, ring/ring_scaling.go:186-271.  wi[L], ti[L][2]; params[0] = reducealgoAddParam, params[1] =
 * reducealgoMulParam.  QiStar mod qi is the product of the other moduli; its inverse by Fermat (prime moduli). */
void oc_simple_scaler_new(const oc_context *c, u64 t, u64 *wi, double *ti, u64 params[2]) {
    const int pow2 = (t & (t - 1)) == 0 && t != 0;                     /* :201 */
    u64 ut[2] = {0, 0};
    if (pow2) {
        params[0] = params[1] = t - 1;                                 /* :203-204 */
    } else {
        oc_bred_params(t, ut);
        params[0] = ut[0];                                             /* :216 */
        params[1] = oc_mred_params(t);                                 /* :214 */
    }
    for (int i = 0; i < c->L; i++) {
        const u64 qi = c->q[i];
        u64 star = 1;
        for (int k = 0; k < c->L; k++)
            if (k != i) star = (u64)(((unsigned __int128)star * (c->q[k] % qi)) % qi);
        u64 bar = oc_mod_exp(star, qi - 2, qi);                        /* :252-253 */
        double a[2], b[2], tmp[2];
        oc_f128_set_uint53(t, a);
        oc_f128_set_uint64(qi, b);
        oc_f128_div(a, b, tmp);                                        /* :255 */
        oc_f128_set_uint64(bar, b);
        oc_f128_mul(tmp, b, a);                                        /* :257 */
        wi[i] = oc_f128_to_uint53(a);                                  /* :260 */
        if ((t & (t - 1)) != 0 && t != 0) wi[i] = oc_mform(wi[i], t, ut);   /* :263-265 */
        bar = (u64)(((unsigned __int128)bar * t) % qi);                /* :267-268 */
        oc_f128_set_uint64(bar, a);
        oc_f128_set_uint64(qi, b);
        oc_f128_div(a, b, &ti[2 * i]);                                 /* :270 */
    }
}

/* SimpleScaler.Scale, ring/ring_scaling.go:275-300: p1 [L][N] -> p2 [L2][N] */
void oc_simple_scale(const oc_context *c, u64 t, const u64 *wi, const double *ti, const u64 params[2], const u64 *p1,
                     u64 *p2, int L2) {
    const int pow2 = (t & (t - 1)) == 0 && t != 0;
    for (u64 i = 0; i < c->N; i++) {
        u64 a = 0;
        double b[2] = {0.0, 0.0};
        for (int j = 0; j < c->L; j++) {
            const u64 x = p1[(size_t)j * c->N + i];
            if (pow2) {
                a += (wi[j] * x) & params[0];                          /* :205 */
            } else {
                const unsigned __int128 m = (unsigned __int128)wi[j] * x;   /* :219-229 */
                const u64 R = (u64)m * params[1];
                const u64 H = (u64)(((unsigned __int128)R * t) >> 64);
                u64 r = (u64)(m >> 64) - H + t;
                if (r >= t) r -= t;
                a += r;
            }
            double fx[2], pr[2], nb[2];
            oc_f128_set_uint64(x, fx);
            oc_f128_mul(&ti[2 * j], fx, pr);
            oc_f128_add(b, pr, nb);                                    /* :290 */
            b[0] = nb[0];
            b[1] = nb[1];
        }
        a += oc_f128_to_uint64(b);                                     /* :293 */
        if (pow2) {
            a &= params[1];                                            /* :209 */
        } else {
            const u64 s0 = (u64)(((unsigned __int128)a * params[0]) >> 64);   /* :233-241 */
            a = a - s0 * t;
            if (a >= t) a -= t;
        }
        for (int j = 0; j < L2; j++) p2[(size_t)j * c->N + i] = a;      /* :296-298 */
    }
}
