// Host-side table generation (lr_precompute.cpp: Montgomery/Barrett constants, primitive roots, psi tables, basis-extension
// tables) under AddressSanitizer + UndefinedBehaviorSanitizer, CPU build only (GPU sanitizers are not available on the pool).
// Built and run by tests/test_host_logic.py::test_precompute_under_sanitizers.
#include <cstdio>
#include <vector>

#include "lr_precompute.hpp"

int main() {
    using namespace lr;
    const std::vector<std::vector<u64>> sets = {
        {576460752303439873ull, 576460752303702017ull},                     // ring/ntt_test.go
        {1152921504606584833ull, 1152921504598720513ull, 1152921504597016577ull},
        {1125899908022273ull, 1099512938497ull, 1099514314753ull},          // CKKS-size moduli
    };
    unsigned long long sum = 0;
    for (u64 logn = 1; logn <= 12; logn += 3)
        for (const auto &m : sets) {
            HostContext h;
            const int rc = build_context(1ull << logn, m.data(), (int)m.size(), h);
            if (rc != 0) {
                std::printf("build_context failed: %d\n", rc);
                return 1;
            }
            for (u64 v : h.ntt_psi) sum += v;
            for (u64 v : h.ntt_psi_inv) sum ^= v;
        }
    {
        const std::vector<u64> Q = sets[1], P = sets[2];
        const HostModup up = build_modup(Q, P);
        for (u64 v : up.qispj_mont) sum += v;
        for (u64 v : up.qpj_inv) sum ^= v;
    }
    std::printf("ok %llx\n", sum);
    return 0;
}
