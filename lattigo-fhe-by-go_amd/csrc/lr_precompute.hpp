// lr_precompute.hpp -- host-side constants of ring.Context / FastBasisExtender / Decomposer.
//
// Everything here runs once per handle on the host.  The values (and, for the primitive
// root, the search order and the factor list it is tested against) must equal the
// reference's, because psi fixes every NTT output: ring/ring_context.go:68-209,
// ring/utils.go:25-288, ring/ring_basis_extension.go:39-142,415-472.
#pragma once
#include <cstdint>
#include <vector>

#include "lr_arith.hpp"
#include "lr_float128.hpp"

namespace lr {

struct BarrettConst { u64 hi, lo; };                 // floor(2^128/q), BRedParams modular_reduction.go:97

BarrettConst barrett_const(u64 q);
u64 montgomery_const(u64 q);                          // MRedParams modular_reduction.go:53
u64 mod_exp(u64 x, u64 e, u64 p);                     // ModExp ring/utils.go:25
bool is_prime(u64 n);                                 // IsPrime ring/utils.go:75 (deterministic bases)
std::vector<u64> factor_list(u64 n);                  // getFactors ring/utils.go:251 (quirks kept)
u64 primitive_root(u64 q);                            // primitiveRoot ring/utils.go:182
u64 bit_reverse(u64 index, unsigned bit_len);         // utils.BitReverse64 utils/utils.go:58

// ring.Context after GenNTTParams
struct HostContext {
    u64 N = 0;
    unsigned logN = 0;
    std::vector<u64> q, mask, mred, psi_mont, psi_inv_mont, n_inv;
    std::vector<BarrettConst> bred;
    std::vector<u64> rescale;       // [L][L], rescale[(j-1)*L + i] for i < j
    std::vector<u64> ntt_psi;       // [L][N] Montgomery form, bit-reversed order
    std::vector<u64> ntt_psi_inv;   // [L][N]
    int L() const { return (int)q.size(); }
};

// returns 0 ok, 1 not NTT friendly, 2 invalid degree (status codes of lattigo_ring.h)
int build_context(u64 N, const u64 *moduli, int L, HostContext &out);

// modupParams, ring_basis_extension.go:19-37 / :76-142
struct HostModup {
    std::vector<u64> Q, P;
    std::vector<u64> qib_mont;      // [nQ]
    std::vector<u64> qispj_mont;    // [nQ][nP]
    std::vector<u64> qpj_inv;       // [nP][nQ+1]
    std::vector<BarrettConst> bredQ, bredP;
    std::vector<u64> mredQ, mredP;
};
HostModup build_modup(const std::vector<u64> &Q, const std::vector<u64> &P);

// genModDownParams(contextP, contextQ), ring_basis_extension.go:39: for each modulus m of `over`,
// MForm((prod of `divisor` moduli)^-1 mod m)
std::vector<u64> build_moddown(const HostContext &over, const HostContext &divisor);

// NewSimpleScaler(t, context), ring/ring_scaling.go:186-268
struct HostSimpleScaler {
    u64 t = 0;
    bool pow2 = false;              // t is a power of two: products are masked, otherwise Montgomery-reduced modulo t
    u64 add_param = 0;              // t-1 (mask) or BRedParams(t)[0]
    u64 mul_param = 0;              // t-1 (mask) or MRedParams(t)
    std::vector<u64> wi;            // floor(QiBarre * t / qi), in Montgomery form modulo t unless pow2
    std::vector<F128> ti;           // ((QiBarre * t) mod qi) / qi as a double-double
};
// returns false when t == 0 (the reference divides by zero in BRedParams)
bool build_simple_scaler(u64 t, const std::vector<u64> &moduli, HostSimpleScaler &out);

}  // namespace lr
