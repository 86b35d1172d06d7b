// lr_precompute.cpp -- see lr_precompute.hpp
#include "lr_precompute.hpp"

#include <algorithm>

namespace lr {

BarrettConst barrett_const(u64 q) {
    // floor(2^128 / q): divide 2^128 - 1 and fix up the one case where the remainder wraps
    const u128 ones = ~(u128)0;
    u128 quo = ones / q;
    if (ones % q == (u128)(q - 1)) quo += 1;
    return BarrettConst{(u64)(quo >> 64), (u64)quo};
}

u64 montgomery_const(u64 q) {
    // q^(2^63 - 1) mod 2^64, by the same square-and-multiply ladder as MRedParams
    u64 inv = 1, sq = q;
    for (int i = 0; i < 63; ++i) {
        inv *= sq;
        sq *= sq;
    }
    return inv;
}

u64 mod_exp(u64 x, u64 e, u64 p) {
    const BarrettConst b = barrett_const(p);
    u64 acc = 1;
    while (e > 0) {
        if (e & 1) acc = bred(acc, x, p, b.hi, b.lo);
        x = bred(x, x, p, b.hi, b.lo);
        e >>= 1;
    }
    return acc;
}

u64 bit_reverse(u64 index, unsigned bit_len) {
    if (bit_len == 0) return 0;
    u64 r = 0;
    for (unsigned i = 0; i < bit_len; ++i) r |= ((index >> i) & 1) << (bit_len - 1 - i);
    return r;
}

namespace {

// the reference's smallPrimes table is the first 2000 primes (2..17389), ring/utils.go:290-391
const std::vector<u64> &small_primes() {
    static const std::vector<u64> table = [] {
        const int limit = 17400;
        std::vector<bool> composite(limit + 1, false);
        std::vector<u64> pr;
        for (int a = 2; a <= limit && pr.size() < 2000; ++a) {
            if (composite[a]) continue;
            pr.push_back((u64)a);
            for (int b = a * a; b <= limit; b += a) composite[b] = true;
        }
        return pr;
    }();
    return table;
}

u64 gcd_zero_aware(u64 a, u64 b) {  // ring/utils.go:53: 0 if either argument is 0
    if (a == 0 || b == 0) return 0;
    while (b != 0) {
        u64 t = a % b;
        a = b;
        b = t;
    }
    return a;
}

u64 rho_step(u64 x, u64 m, u64 c) {  // polynomialPollardsRho ring/utils.go:212
    u64 z = mod_exp(x, 2, m);
    z += c;
    return z % m;
}

u64 rho_factor(u64 m) {  // factorizationPollardsRho ring/utils.go:222 (with its y>x swap)
    u64 d = 0;
    for (u64 c = 1; c < 10; ++c) {
        u64 x = 2, y = 2;
        d = 1;
        while (d != 0) {
            x = rho_step(x, m, c);
            y = rho_step(rho_step(y, m, c), m, c);
            if (y > x) std::swap(x, y);
            d = gcd_zero_aware(x - y, m);
            if (d > 1) return d;
        }
    }
    return d;
}

}  // namespace

bool is_prime(u64 n) {
    if (n < 2) return false;
    for (u64 p : small_primes())
        if (n == p) return true;
    for (u64 p : small_primes())
        if (n % p == 0) return false;
    // Miller-Rabin.  The reference draws 50 random bases; for 64-bit n the first twelve
    // primes are a proven deterministic witness set, so the verdict is the same.
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        ++s;
    }
    const BarrettConst b = barrett_const(n);
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = mod_exp(a % n, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int i = 1; i < s; ++i) {
            x = bred(x, x, n, b.hi, b.lo);
            if (x == n - 1) {
                witness = false;
                break;
            }
        }
        if (witness) return false;
    }
    return true;
}

std::vector<u64> factor_list(u64 n) {
    std::vector<u64> out;
    u64 m = n;
    for (u64 p : small_primes()) {
        bool divides = false;
        while (m % p == 0) {
            m /= p;
            divides = true;
        }
        if (divides) out.push_back(p);
    }
    if (m == 1) return out;
    for (;;) {
        u64 f = rho_factor(m);
        if (f == 0) {
            out.push_back(m);
            break;
        }
        m /= f;
        if (!out.empty() && f == out.back()) continue;
        out.push_back(f);
    }
    return out;
}

u64 primitive_root(u64 q) {
    const std::vector<u64> factors = factor_list(q - 1);
    u64 g = 2;
    bool searching = true;
    while (searching) {
        ++g;
        for (u64 f : factors) {
            if (mod_exp(g, (q - 1) / f, q) == 1) {
                searching = true;
                break;
            }
            searching = false;
        }
    }
    return g;
}

static unsigned bit_length(u64 x) {
    unsigned n = 0;
    while (x) {
        ++n;
        x >>= 1;
    }
    return n;
}

int build_context(u64 N, const u64 *moduli, int L, HostContext &c) {
    if (N == 0 || (N & (N - 1)) != 0) return 2;
    c = HostContext();
    c.N = N;
    c.logN = bit_length(N) - 1;
    c.q.assign(moduli, moduli + L);
    c.mask.resize(L);
    c.mred.assign(L, 0);
    c.bred.resize(L);
    for (int i = 0; i < L; ++i) {
        const u64 qi = c.q[i];
        const unsigned bl = bit_length(qi);
        c.mask[i] = bl >= 64 ? ~(u64)0 : (((u64)1 << bl) - 1);
        c.bred[i] = barrett_const(qi);
        if (qi != 0 && (qi & (qi - 1)) != 0) c.mred[i] = montgomery_const(qi);
    }
    for (int i = 0; i < L; ++i)
        if (!is_prime(c.q[i]) || (c.q[i] & ((N << 1) - 1)) != 1) return 1;

    c.rescale.assign((size_t)L * L, 0);
    for (int j = L - 1; j > 0; --j)
        for (int i = 0; i < j; ++i)
            c.rescale[(size_t)(j - 1) * L + i] =
                mform(mod_exp(c.q[j], c.q[i] - 2, c.q[i]), c.q[i], c.bred[i].hi, c.bred[i].lo);

    c.psi_mont.resize(L);
    c.psi_inv_mont.resize(L);
    c.n_inv.resize(L);
    c.ntt_psi.assign((size_t)L * N, 0);
    c.ntt_psi_inv.assign((size_t)L * N, 0);
    for (int i = 0; i < L; ++i) {
        const u64 qi = c.q[i];
        const BarrettConst b = c.bred[i];
        c.n_inv[i] = mform(mod_exp(N, qi - 2, qi), qi, b.hi, b.lo);
        const u64 g = primitive_root(qi);
        const u64 power = (qi - 1) / (N << 1);
        const u64 power_inv = (qi - 1) - power;
        const u64 psi = mform(mod_exp(g, power, qi), qi, b.hi, b.lo);
        const u64 psi_inv = mform(mod_exp(g, power_inv, qi), qi, b.hi, b.lo);
        c.psi_mont[i] = psi;
        c.psi_inv_mont[i] = psi_inv;
        u64 *fwd = &c.ntt_psi[(size_t)i * N];
        u64 *inv = &c.ntt_psi_inv[(size_t)i * N];
        u64 run_f = mform(1, qi, b.hi, b.lo), run_i = run_f;
        fwd[0] = run_f;
        inv[0] = run_i;
        for (u64 j = 1; j < N; ++j) {
            run_f = mred(run_f, psi, qi, c.mred[i]);
            run_i = mred(run_i, psi_inv, qi, c.mred[i]);
            const u64 slot = bit_reverse(j, c.logN);
            fwd[slot] = run_f;
            inv[slot] = run_i;
        }
    }
    return 0;
}

namespace {
u64 mulmod(u64 a, u64 b, u64 m) { return (u64)(((u128)a * b) % m); }
// product of v[k] over k != skip, modulo m (big.Int Quo/Mod in the reference)
u64 product_mod(const std::vector<u64> &v, int skip, u64 m) {
    u64 r = 1 % m;
    for (int k = 0; k < (int)v.size(); ++k)
        if (k != skip) r = mulmod(r, v[k] % m, m);
    return r;
}
u64 inverse_mod_prime(u64 a, u64 m) {
    u64 r = 1, e = m - 2;
    a %= m;
    while (e) {
        if (e & 1) r = mulmod(r, a, m);
        a = mulmod(a, a, m);
        e >>= 1;
    }
    return r;
}
}  // namespace

HostModup build_modup(const std::vector<u64> &Q, const std::vector<u64> &P) {
    HostModup m;
    m.Q = Q;
    m.P = P;
    const int nQ = (int)Q.size(), nP = (int)P.size();
    m.bredQ.resize(nQ);
    m.mredQ.resize(nQ);
    m.bredP.resize(nP);
    m.mredP.resize(nP);
    for (int i = 0; i < nQ; ++i) {
        m.bredQ[i] = barrett_const(Q[i]);
        m.mredQ[i] = montgomery_const(Q[i]);
    }
    for (int j = 0; j < nP; ++j) {
        m.bredP[j] = barrett_const(P[j]);
        m.mredP[j] = montgomery_const(P[j]);
    }
    m.qib_mont.resize(nQ);
    m.qispj_mont.resize((size_t)nQ * nP);
    m.qpj_inv.resize((size_t)nP * (nQ + 1));
    for (int i = 0; i < nQ; ++i) {
        const u64 qi = Q[i];
        const u64 star_inv = inverse_mod_prime(product_mod(Q, i, qi), qi);
        m.qib_mont[i] = mform(star_inv, qi, m.bredQ[i].hi, m.bredQ[i].lo);
        for (int j = 0; j < nP; ++j)
            m.qispj_mont[(size_t)i * nP + j] = mform(product_mod(Q, i, P[j]), P[j], m.bredP[j].hi, m.bredP[j].lo);
    }
    for (int j = 0; j < nP; ++j) {
        const u64 pj = P[j];
        const u64 step = pj - product_mod(Q, -1, pj);
        u64 *row = &m.qpj_inv[(size_t)j * (nQ + 1)];
        row[0] = 0;
        for (int i = 1; i <= nQ; ++i) row[i] = cred(row[i - 1] + step, pj);
    }
    return m;
}

std::vector<u64> build_moddown(const HostContext &over, const HostContext &divisor) {
    std::vector<u64> r(over.L());
    for (int i = 0; i < over.L(); ++i) {
        const u64 m = over.q[i];
        u64 v = product_mod(divisor.q, -1, m);
        v = mod_exp(v, m - 2, m);
        r[i] = mform(v, m, over.bred[i].hi, over.bred[i].lo);
    }
    return r;
}

bool build_simple_scaler(u64 t, const std::vector<u64> &moduli, HostSimpleScaler &out) {
    if (t == 0) return false;
    out.t = t;
    out.pow2 = (t & (t - 1)) == 0;                                   // ring_scaling.go:201
    BarrettConst bt{0, 0};
    if (out.pow2) {
        out.add_param = out.mul_param = t - 1;
    } else {
        bt = barrett_const(t);
        out.add_param = bt.hi;                                       // :216
        out.mul_param = montgomery_const(t);                         // :217
    }
    const int L = (int)moduli.size();
    out.wi.assign(L, 0);
    out.ti.assign(L, F128{0.0, 0.0});
    for (int i = 0; i < L; ++i) {
        const u64 qi = moduli[i];
        u64 bar = inverse_mod_prime(product_mod(moduli, i, qi), qi);  // QiBarre, :250-253
        F128 tmp = f128_div(f128_set_uint53(t), f128_set_uint64(qi));  // :255
        tmp = f128_mul(tmp, f128_set_uint64(bar));                     // :257
        u64 w = f128_to_uint53(tmp);                                   // :260
        if (!out.pow2) w = mform(w, t, bt.hi, bt.lo);                  // :263-265
        out.wi[i] = w;
        bar = mulmod(bar, t % qi, qi);                                 // :267-268
        out.ti[i] = f128_div(f128_set_uint64(bar), f128_set_uint64(qi));   // :270
    }
    return true;
}

}  // namespace lr
