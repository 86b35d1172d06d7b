// Prints the host-side SimpleScaler tables of lr_precompute.cpp (NewSimpleScaler, ring/ring_scaling.go:186) for the moduli and
// t on the command line: "<t> <q0> <q1> ..." -> one line per modulus "wi ti_hi_bits ti_lo_bits" (hex), then "add mul".
// Built and run by tests/test_oracle_scaling.py::test_host_tables_match_oracle (no GPU involved).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lr_precompute.hpp"

int main(int argc, char **argv) {
    using namespace lr;
    if (argc < 3) return 2;
    const u64 t = std::strtoull(argv[1], nullptr, 10);
    std::vector<u64> q;
    for (int i = 2; i < argc; ++i) q.push_back(std::strtoull(argv[i], nullptr, 10));
    HostSimpleScaler s;
    if (!build_simple_scaler(t, q, s)) {
        std::printf("t=0\n");
        return 1;
    }
    for (size_t i = 0; i < q.size(); ++i) {
        u64 hi, lo;
        std::memcpy(&hi, &s.ti[i].hi, 8);
        std::memcpy(&lo, &s.ti[i].lo, 8);
        std::printf("%llx %llx %llx\n", (unsigned long long)s.wi[i], (unsigned long long)hi, (unsigned long long)lo);
    }
    std::printf("%llx %llx\n", (unsigned long long)s.add_param, (unsigned long long)s.mul_param);
    return 0;
}
