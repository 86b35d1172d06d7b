// A C++ client of the C ABI (include/lattigo_ring.h; HIP only to create the per-thread streams of the direct mode; no Python): T host threads, each the stand-in of one goroutine with its
// own evaluator (examples/dbfv/psi/psi.go:215-233), call MulRelin on one ciphertext pair in a loop
//   direct : every thread on its own contexts + plan + HIP stream
//   batcher: every thread through lr_ckks_batcher_mulrelin
// and every result is compared with the one lr_ckks_mulrelin gave for the same operands (whose parity with the oracle is the GPU suite's).
// What Python's thread switch costs is absent here: this is what a Go or C++ host sees.
//   build: tools/dbg/batcher_bench.py --build (g++ against the in-tree liblattigo_ring_hip.so and libamdhip64)
//   run  : tools/build/batcher_bench moduli.txt T iters lanes max_batch      (moduli.txt: "logN nq np" then the nq + np moduli; tools/dbg/batcher_bench.py writes it)
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "lattigo_ring.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        int rc_ = (x);                                                                          \
        if (rc_ != LR_OK) {                                                                     \
            std::fprintf(stderr, "%s: status %d: %s\n", #x, rc_, lr_last_error_string());       \
            std::exit(1);                                                                       \
        }                                                                                       \
    } while (0)

struct Caller {
    lr_context *cq = nullptr, *cp = nullptr;
    lr_ckks_plan *plan = nullptr;
    lr_poly *key = nullptr, *a0 = nullptr, *a1 = nullptr, *b0 = nullptr, *b1 = nullptr, *o0 = nullptr, *o1 = nullptr;
};

int main(int argc, char **argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: batcher_bench moduli.txt threads iters lanes max_batch\n");
        return 2;
    }
    std::FILE *f = std::fopen(argv[1], "r");
    if (!f) return 2;
    int logn = 0, nq = 0, np = 0;
    if (std::fscanf(f, "%d %d %d", &logn, &nq, &np) != 3) return 2;
    std::vector<uint64_t> Q(nq), P(np);
    for (auto &q : Q) if (std::fscanf(f, "%llu", (unsigned long long *)&q) != 1) return 2;
    for (auto &p : P) if (std::fscanf(f, "%llu", (unsigned long long *)&p) != 1) return 2;
    std::fclose(f);
    const int T = std::atoi(argv[2]), iters = std::atoi(argv[3]), lanes = std::atoi(argv[4]), max_batch = std::atoi(argv[5]);
    const uint64_t N = 1ull << logn;
    const int level = nq - 1, beta = (nq + np - 1) / np;
    // operands: uniform residues; the key image: 2 * beta polys over Q || P
    std::mt19937_64 rng(7);
    auto fill = [&](std::vector<uint64_t> &v, const std::vector<uint64_t> &mods, int polys) {
        v.resize((size_t)polys * mods.size() * N);
        for (int b = 0; b < polys; ++b)
            for (size_t i = 0; i < mods.size(); ++i)
                for (uint64_t j = 0; j < N; ++j) v[((size_t)b * mods.size() + i) * N + j] = rng() % mods[i];
    };
    std::vector<uint64_t> QP(Q);
    QP.insert(QP.end(), P.begin(), P.end());
    std::vector<uint64_t> hkey, hop[4];
    fill(hkey, QP, 2 * beta);
    for (auto &h : hop) fill(h, Q, 1);

    std::vector<Caller> cs(T);
    for (auto &c : cs) {
        CK(lr_context_create(N, Q.data(), nq, 0, &c.cq));
        CK(lr_context_create(N, P.data(), np, 0, &c.cp));
        hipStream_t st = nullptr;
        // STREAM_PRIORITY=<p>: created the way torch creates its pool streams (hipStreamCreateWithPriority)
        const char *prio = std::getenv("STREAM_PRIORITY");
        if ((prio ? hipStreamCreateWithPriority(&st, hipStreamNonBlocking, std::atoi(prio)) : hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return 1;
        CK(lr_context_set_stream(c.cq, st));
        CK(lr_context_set_stream(c.cp, st));
        CK(lr_ckks_plan_create(c.cq, c.cp, 1, &c.plan));
        CK(lr_poly_alloc(c.cq, nq + np, 2 * beta, &c.key));
        CK(lr_poly_upload_dense(c.key, hkey.data(), hkey.size()));
        lr_poly **ops[4] = {&c.a0, &c.a1, &c.b0, &c.b1};
        for (int k = 0; k < 4; ++k) {
            CK(lr_poly_alloc(c.cq, nq, 1, ops[k]));
            CK(lr_poly_upload_dense(*ops[k], hop[k].data(), hop[k].size()));
        }
        CK(lr_poly_alloc(c.cq, nq, 1, &c.o0));
        CK(lr_poly_alloc(c.cq, nq, 1, &c.o1));
    }
    // OP=rotate: RotateColumns by 3 (Galois element 5^3) instead of MulRelin, the "key" standing in for the rotation key
    const bool rotate = std::getenv("OP") && std::string(std::getenv("OP")) == "rotate";
    const uint64_t gal = 125 % (2 * N);
    // the answer, from the plain entry point
    std::vector<uint64_t> want0((size_t)nq * N), want1((size_t)nq * N), got((size_t)nq * N);
    if (rotate) CK(lr_ckks_rotate(cs[0].plan, level, cs[0].a0, cs[0].a1, gal, cs[0].key, cs[0].o0, cs[0].o1));
    else
    CK(lr_ckks_mulrelin(cs[0].plan, level, cs[0].a0, cs[0].a1, cs[0].b0, cs[0].b1, cs[0].key, cs[0].o0, cs[0].o1));
    CK(lr_poly_download_dense(cs[0].o0, want0.data(), want0.size()));
    CK(lr_poly_download_dense(cs[0].o1, want1.data(), want1.size()));

    // the batcher: its own lanes, ONE key image for every caller
    std::vector<lr_context *> lq(lanes), lp(lanes);
    std::vector<lr_ckks_plan *> lpl(lanes);
    for (int i = 0; i < lanes; ++i) {
        CK(lr_context_create(N, Q.data(), nq, 0, &lq[i]));
        CK(lr_context_create(N, P.data(), np, 0, &lp[i]));
        CK(lr_ckks_plan_create(lq[i], lp[i], max_batch, &lpl[i]));
    }
    lr_ckks_batcher *bat = nullptr;
    CK(lr_ckks_batcher_create(lpl.data(), lanes, &bat));
    lr_poly *bkey = cs[0].key;

    for (int how = 0; how < 2; ++how) {
        for (auto &c : cs) {
            CK(lr_poly_zero(c.o0));
            CK(lr_poly_zero(c.o1));
            CK(lr_context_sync(c.cq));
        }
        if (how == 1 && rotate) CK(lr_ckks_batcher_rotate(bat, level, cs[0].a0, cs[0].a1, gal, bkey, cs[0].o0, cs[0].o1));
        else if (how == 1) CK(lr_ckks_batcher_mulrelin(bat, level, cs[0].a0, cs[0].a1, cs[0].b0, cs[0].b1, bkey, cs[0].o0, cs[0].o1));   // warm-up: pools
        std::atomic<int> ready{0};
        std::atomic<bool> go{false};
        std::vector<std::thread> ths;
        for (int t = 0; t < T; ++t)
            ths.emplace_back([&, t]() {
                Caller &c = cs[t];
                ready.fetch_add(1);
                while (!go.load()) std::this_thread::yield();
                for (int i = 0; i < iters; ++i) {
                    if (rotate && how == 0)
                        CK(lr_ckks_rotate(c.plan, level, c.a0, c.a1, gal, c.key, c.o0, c.o1));
                    else if (rotate)
                        CK(lr_ckks_batcher_rotate(bat, level, c.a0, c.a1, gal, bkey, c.o0, c.o1));
                    else if (how == 0)
                        CK(lr_ckks_mulrelin(c.plan, level, c.a0, c.a1, c.b0, c.b1, c.key, c.o0, c.o1));
                    else
                        CK(lr_ckks_batcher_mulrelin(bat, level, c.a0, c.a1, c.b0, c.b1, bkey, c.o0, c.o1));
                }
                CK(lr_context_sync(c.cq));
            });
        while (ready.load() < T) std::this_thread::yield();
        const auto t0 = std::chrono::steady_clock::now();
        go.store(true);
        for (auto &th : ths) th.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        bool ok = true;
        for (auto &c : cs) {
            CK(lr_poly_download_dense(c.o0, got.data(), got.size()));
            ok = ok && std::memcmp(got.data(), want0.data(), got.size() * 8) == 0;
            CK(lr_poly_download_dense(c.o1, got.data(), got.size()));
            ok = ok && std::memcmp(got.data(), want1.data(), got.size() * 8) == 0;
        }
        uint64_t nb = 0, npd = 0;
        int largest = 0;
        CK(lr_ckks_batcher_stats(bat, &nb, &npd, &largest));
        std::printf("{\"how\": \"%s\", \"threads\": %d, \"calls_per_thread\": %d, \"calls_per_s\": %.1f, \"same_bits_as_the_plain_entry_point\": %s",
                    how == 0 ? "direct" : "batcher", T, iters, (double)T * iters / dt, ok ? "true" : "false");
        if (how == 1) std::printf(", \"lanes\": %d, \"launches\": %llu, \"mean_batch\": %.2f, \"largest_batch\": %d", lanes, (unsigned long long)nb, (double)npd / (double)nb, largest);
        std::printf("}\n");
        std::fflush(stdout);
        if (!ok) return 1;
    }
    lr_ckks_batcher_destroy(bat);
    return 0;
}
