// C++ twin of the reference's Test_NTT (ring/ntt_test.go:101-142) on the C++ host mirror: for each golden
// file pair, build the context from the file, run Context.NTT, compare EVERY coefficient with the expected
// file, then InvNTT and compare with the input.  Usage: test_ntt_golden <golden dir>.  Exit code 0 = all equal.
#include <cstdio>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>

#include "lattigo_ring.hpp"

struct Golden {
    uint64_t N;
    std::vector<uint64_t> moduli;
    std::vector<std::vector<uint64_t>> coeffs;
};

static Golden load(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    Golden g;
    std::string line;
    std::getline(f, line);
    g.N = std::stoull(line);
    std::getline(f, line);
    std::istringstream ms(line);
    for (uint64_t v; ms >> v;) g.moduli.push_back(v);
    for (size_t i = 0; i < g.moduli.size(); ++i) {
        std::getline(f, line);
        std::istringstream cs(line);
        std::vector<uint64_t> limb;
        for (uint64_t v; cs >> v;) limb.push_back(v);
        if (limb.size() != g.N) throw std::runtime_error("bad limb length in " + path);
        g.coeffs.push_back(limb);
    }
    return g;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const char *sizes[] = {"___8", "__16", "__32", "__64", "_128", "_256", "_512"};
    int failures = 0;
    try {
        for (const char *s : sizes) {
            Golden in = load(dir + "/test_pol_60_" + s + "_2");
            Golden want = load(dir + "/test_pol_NTT_60_" + s + "_2");
            ring::Context context(in.N, in.moduli);
            std::unique_ptr<ring::Poly> Polx(context.NewPoly());
            Polx->SetCoefficients(in.coeffs);
            context.NTT(Polx.get(), Polx.get());
            if (Polx->GetCoefficients() != want.coeffs) {
                std::printf("error : NTT coeffs N=%llu\n", (unsigned long long)in.N);
                ++failures;
            }
            context.InvNTT(Polx.get(), Polx.get());
            if (Polx->GetCoefficients() != in.coeffs) {
                std::printf("error : InvNTT coeffs N=%llu\n", (unsigned long long)in.N);
                ++failures;
            }
        }
        // misuse: fewer limbs than the context has moduli (Go: index out of range panic)
        ring::Context c2(1 << 12, {1152921504050839553ull, 1152921504053723137ull});
        std::unique_ptr<ring::Poly> small(c2.NewPolyLvl(0)), full(c2.NewPoly());
        bool threw = false;
        try {
            c2.NTT(small.get(), full.get());
        } catch (const ring::Error &e) {
            threw = e.code == LR_ERR_SHAPE;
        }
        if (!threw) {
            std::printf("error : short poly accepted\n");
            ++failures;
        }
    } catch (const std::exception &e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
    std::printf("%s\n", failures ? "FAIL" : "PASS: 7 golden NTT/InvNTT pairs, every coefficient");
    return failures ? 1 : 0;
}
