// BASELINE.json config 5 from ONE process, through the C ABI only (include/lattigo_ring.h; no Python, no torch, no collective library, no
// direct HIP call): a batch of independent CKKS MulRelin at DefaultParams[PN16QP1761], sharded over the GPUs of the node by contiguous
// blocks -- one host thread per device, each with its own contexts, plan, key image and operands, the stand-in of one goroutine with its own
// evaluator (examples/dbfv/psi/psi.go:215-233) -- and the results gathered to device 0 by lr_poly_copy_peer per chunk, so that a chunk
// crosses xGMI while the next one is computed (device 0 computes straight into its block of the root buffers); lr_context_wait_peer_copies
// on the root at the end of the step.
//
//   build: tools/dbg/multi_gpu_bench.py --build     (g++ against the in-tree liblattigo_ring_hip.so)
//   run  : tools/build/multi_gpu_bench [--gpus G] [--units U] [--chunk C] [--steps K] [--warmup W] [--set PN16QP1761|PN15QP880|PN14QP438] [--logn n]
//
// G = min(requested, lr_device_count()).  Prints one JSON line with the keys of bench.py's `config5` object.  After the timed steps the
// root buffers are poisoned, one more step runs, and the placement of every block is checked against the producers' own outputs.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "lattigo_ring.h"

typedef unsigned __int128 u128;

#define CK(x)                                                                                   \
    do {                                                                                        \
        int rc_ = (x);                                                                          \
        if (rc_ != LR_OK) {                                                                     \
            std::fprintf(stderr, "%s: status %d: %s\n", #x, rc_, lr_last_error_string());       \
            std::exit(1);                                                                       \
        }                                                                                       \
    } while (0)

// ring.GenerateNTTPrimes (ring/utils.go:133-175): the first `count` primes 2^bits + 1 + k * 2N, ascending (deterministic Miller-Rabin)
static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((u128)a * b % m); }
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t m) {
    uint64_t r = 1;
    for (a %= m; e; e >>= 1, a = mulmod(a, a, m))
        if (e & 1) r = mulmod(r, a, m);
    return r;
}
static bool is_prime(uint64_t n) {
    if (n < 2) return false;
    for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (n % p == 0) return n == p;
    }
    uint64_t d = n - 1;
    int s = 0;
    while ((d & 1) == 0) d >>= 1, ++s;
    for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < s && comp; ++i) {
            x = mulmod(x, x, n);
            comp = x != n - 1;
        }
        if (comp) return false;
    }
    return true;
}
static std::vector<uint64_t> ntt_primes(int bits, int logn, int count) {
    std::vector<uint64_t> out;
    const uint64_t step = 2ull << logn;
    for (uint64_t x = (1ull << bits) + 1; (int)out.size() < count; x += step)
        if (is_prime(x)) out.push_back(x);
    return out;
}

struct Barrier {            // (std::barrier is C++20)
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned long gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long g = gen;
        if (++waiting == n) {
            waiting = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};

struct Device {
    lr_context *cq = nullptr, *cp = nullptr;
    lr_ckks_plan *plan = nullptr;
    lr_poly *key = nullptr, *in[4] = {nullptr, nullptr, nullptr, nullptr}, *out[2] = {nullptr, nullptr};
    std::vector<lr_poly *> vin[4], vout[2];     // per chunk views
};

static uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    int want_gpus = 8, units = 128, chunk = 32, steps = 3, warmup = 1, logn_override = 0;
    std::string set = "PN16QP1761";
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "--gpus") want_gpus = std::atoi(argv[i + 1]);
        else if (k == "--units") units = std::atoi(argv[i + 1]);
        else if (k == "--chunk") chunk = std::atoi(argv[i + 1]);
        else if (k == "--steps") steps = std::atoi(argv[i + 1]);
        else if (k == "--warmup") warmup = std::atoi(argv[i + 1]);
        else if (k == "--set") set = argv[i + 1];
        else if (k == "--logn") logn_override = std::atoi(argv[i + 1]);
        else {
            std::fprintf(stderr, "unknown option %s\n", k.c_str());
            return 2;
        }
    }
    // ckks/params.go:36-87 (LogQi, LogPi); GenModuli hands the primes of one size out in the order Q then P (ckks/utils.go:150-193)
    int logn, nq, np, q0bits, qbits, pbits;
    if (set == "PN16QP1761") logn = 16, nq = 34, np = 4, q0bits = 55, qbits = 45, pbits = 55;
    else if (set == "PN15QP880") logn = 15, nq = 18, np = 3, q0bits = 50, qbits = 40, pbits = 50;
    else if (set == "PN14QP438") logn = 14, nq = 10, np = 2, q0bits = 45, qbits = 34, pbits = 43;
    else {
        std::fprintf(stderr, "unknown parameter set %s\n", set.c_str());
        return 2;
    }
    const int gen_logn = logn;
    if (logn_override) logn = logn_override;           // (tests: the set's limb structure on a smaller ring; the primes stay NTT-friendly for it)
    std::vector<uint64_t> Q, P;
    {
        if (q0bits == pbits) {
            auto big = ntt_primes(q0bits, gen_logn, 1 + np);
            Q.push_back(big[0]);
            P.assign(big.begin() + 1, big.end());
        } else {
            Q.push_back(ntt_primes(q0bits, gen_logn, 1)[0]);
            P = ntt_primes(pbits, gen_logn, np);
        }
        auto small = ntt_primes(qbits, gen_logn, nq - 1);
        Q.insert(Q.end(), small.begin(), small.end());
    }
    const uint64_t N = 1ull << logn;
    const int level = nq - 1, beta = (nq + np - 1) / np;
    int have = 0;
    CK(lr_device_count(&have));
    const int G = std::max(1, std::min(want_gpus, have));
    if (have < 1) {
        std::fprintf(stderr, "no HIP device: the hot path has no CPU fallback\n");
        return 1;
    }
    if (chunk > units) chunk = units;
    const int total = units * G;
    const size_t poly_words = (size_t)nq * N;

    std::vector<Device> dev(G);
    lr_poly *root[2] = {nullptr, nullptr};
    Barrier bar(G);
    std::vector<double> t_full(G, 0.0), t_comp(G, 0.0);
    std::atomic<int> placement_bad{0};
    std::atomic<long long> checked{0};

    auto worker = [&](int g) {
        Device &d = dev[g];
        CK(lr_context_create(N, Q.data(), nq, g, &d.cq));
        CK(lr_context_create(N, P.data(), np, g, &d.cp));
        CK(lr_ckks_plan_create(d.cq, d.cp, chunk, &d.plan));
        CK(lr_poly_alloc(d.cq, nq + np, 2 * beta, &d.key));
        uint64_t seed = 9;                                 // the key is replicated: the same on every device
        {
            std::vector<uint64_t> h((size_t)2 * beta * (nq + np) * N);
            for (int b = 0; b < 2 * beta; ++b)
                for (int i = 0; i < nq + np; ++i) {
                    const uint64_t m = i < nq ? Q[i] : P[i - nq];
                    uint64_t *row = h.data() + ((size_t)b * (nq + np) + i) * N;
                    for (uint64_t j = 0; j < N; ++j) row[j] = splitmix(seed) % m;
                }
            CK(lr_poly_upload_dense(d.key, h.data(), h.size()));
        }
        // operands: four distinct units per device (seeded by device and unit), tiled over the block ON THE DEVICE: unit u takes pattern u % 4
        const int distinct = std::min(units, 4);
        for (int k = 0; k < 4; ++k) CK(lr_poly_alloc(d.cq, nq, units, &d.in[k]));
        // device 0 computes straight into its block of the root buffers (the first `units` slots): its share of the gather costs nothing
        if (g == 0) {
            for (int k = 0; k < 2; ++k) CK(lr_poly_alloc(d.cq, nq, total, &root[k]));
            for (int k = 0; k < 2; ++k) {
                void *base = nullptr;
                CK(lr_poly_info(root[k], nullptr, nullptr, nullptr, &base));
                CK(lr_poly_wrap(d.cq, base, nq, units, &d.out[k]));
            }
        } else {
            for (int k = 0; k < 2; ++k) CK(lr_poly_alloc(d.cq, nq, units, &d.out[k]));
        }
        {
            std::vector<uint64_t> h(poly_words);
            lr_poly *one = nullptr;
            CK(lr_poly_alloc(d.cq, nq, 1, &one));
            for (int k = 0; k < 4; ++k) {
                void *base = nullptr;
                CK(lr_poly_info(d.in[k], nullptr, nullptr, nullptr, &base));
                for (int p = 0; p < distinct; ++p) {
                    uint64_t s2 = 0xC5ull * 1000 + (uint64_t)g * 64 + (uint64_t)p * 4 + (uint64_t)k;
                    for (int i = 0; i < nq; ++i)
                        for (uint64_t j = 0; j < N; ++j) h[(size_t)i * N + j] = splitmix(s2) % Q[i];
                    CK(lr_poly_upload_dense(one, h.data(), h.size()));
                    const int cnt = (units - p + distinct - 1) / distinct;
                    lr_poly *view = nullptr;
                    CK(lr_poly_wrap_strided(d.cq, (uint64_t *)base + (size_t)p * poly_words, nq, cnt, (long long)distinct * (long long)poly_words, &view));
                    CK(lr_ewise(d.cq, LR_COPY, level, one, nullptr, view, nullptr));       // batch-1 operand: broadcast
                    CK(lr_context_sync(d.cq));
                    CK(lr_poly_free(view));
                }
            }
            CK(lr_poly_free(one));
        }
        for (int u0 = 0; u0 < units; u0 += chunk) {
            const int nb = std::min(chunk, units - u0);
            for (int k = 0; k < 4; ++k) {
                void *base = nullptr;
                lr_poly *v = nullptr;
                CK(lr_poly_info(d.in[k], nullptr, nullptr, nullptr, &base));
                CK(lr_poly_wrap(d.cq, (uint64_t *)base + (size_t)u0 * poly_words, nq, nb, &v));
                d.vin[k].push_back(v);
            }
            for (int k = 0; k < 2; ++k) {
                void *base = nullptr;
                lr_poly *v = nullptr;
                CK(lr_poly_info(d.out[k], nullptr, nullptr, nullptr, &base));
                CK(lr_poly_wrap(d.cq, (uint64_t *)base + (size_t)u0 * poly_words, nq, nb, &v));
                d.vout[k].push_back(v);
            }
        }
        CK(lr_context_sync(d.cq));
        bar.wait();
        lr_context *rootctx = dev[0].cq;

        auto step = [&](bool gather) {
            int c = 0;
            for (int u0 = 0; u0 < units; u0 += chunk, ++c) {
                const int nb = std::min(chunk, units - u0);
                CK(lr_ckks_mulrelin(d.plan, level, d.vin[0][c], d.vin[1][c], d.vin[2][c], d.vin[3][c], d.key, d.vout[0][c], d.vout[1][c]));
                if (gather && g != 0)
                    for (int k = 0; k < 2; ++k) CK(lr_poly_copy_peer(rootctx, root[k], g * units + u0, d.cq, d.out[k], u0, nb));
            }
        };
        auto finish = [&](bool gather) {
            CK(lr_context_sync(d.cq));               // this device's kernels (the copies out of it are behind them on the copy streams)
            bar.wait();
            if (gather && g == 0) {
                CK(lr_context_wait_peer_copies(rootctx));
                CK(lr_context_sync(rootctx));        // every block has landed
            }
            bar.wait();
        };
        for (int mode = 0; mode < 2; ++mode) {       // 0: products + gather, 1: products only
            const bool gather = mode == 0;
            for (int w = 0; w < warmup; ++w) step(gather);
            finish(gather);
            const auto t0 = std::chrono::steady_clock::now();
            for (int s = 0; s < steps; ++s) step(gather);
            finish(gather);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            (gather ? t_full : t_comp)[g] = dt;
        }
        // placement: poison the root, one more step, compare sampled units of every block with their producer's own output
        if (g == 0) {
            std::vector<uint64_t> ff(poly_words, ~0ull);
            lr_poly *one = nullptr;
            CK(lr_poly_alloc(d.cq, nq, 1, &one));
            CK(lr_poly_upload_dense(one, ff.data(), ff.size()));
            for (int k = 0; k < 2; ++k) CK(lr_ewise(d.cq, LR_COPY, level, one, nullptr, root[k], nullptr));     // broadcast over all `total` slots
            CK(lr_context_sync(d.cq));
            CK(lr_poly_free(one));
        }
        bar.wait();
        step(true);
        finish(true);
        {
            // one unit at a time (lr_poly_download: the per-limb form a Go caller uses): this device's own result against its slot on the root
            std::vector<uint64_t> mine(poly_words), there(poly_words), first(poly_words);
            std::vector<uint64_t *> pm(nq), pt(nq), pf(nq);
            for (int i = 0; i < nq; ++i) pm[i] = mine.data() + (size_t)i * N, pt[i] = there.data() + (size_t)i * N, pf[i] = first.data() + (size_t)i * N;
            for (int k = 0; k < 2; ++k) {
                CK(lr_poly_download(d.out[k], 0, pf.data(), nq));
                for (int u : {0, 1, units / 2, units - 1}) {
                    if (u < 0 || u >= units) continue;
                    CK(lr_poly_download(d.out[k], u, pm.data(), nq));
                    CK(lr_poly_download(root[k], g * units + u, pt.data(), nq));
                    const bool same = std::memcmp(mine.data(), there.data(), poly_words * 8) == 0;
                    const bool nontrivial = mine[0] != ~0ull || mine[1] != ~0ull;
                    if (!same || !nontrivial) placement_bad.fetch_add(1);
                    if (u == 1 && std::memcmp(mine.data(), first.data(), poly_words * 8) == 0) placement_bad.fetch_add(1);   // distinct units differ
                    checked.fetch_add(1);
                }
            }
        }
        bar.wait();
    };
    std::vector<std::thread> ths;
    for (int g = 0; g < G; ++g) ths.emplace_back(worker, g);
    for (auto &t : ths) t.join();
    double full = 0, comp = 0;
    for (int g = 0; g < G; ++g) full = std::max(full, t_full[g]), comp = std::max(comp, t_comp[g]);
    const double ct_mib = 2.0 * nq * N * 8 / (1 << 20);
    // algorithmic bytes of one product: two input ciphertexts, the key, the output ciphertext
    const double alg = 8.0 * N * (4.0 * nq + beta * 2.0 * (nq + np) + 2.0 * nq);
    std::printf("{\"value\": %.3f, \"unit\": \"MulRelin/s\", \"params\": \"%s (N=2^%d, %d Q + %d P limbs, beta=%d), level %d\", \"units_total\": %d, "
                "\"units_per_gpu\": %d, \"chunk\": %d, \"n_gpus\": %d, \"devices_visible\": %d, \"ms_per_step_compute_and_gather\": %.4f, "
                "\"ms_per_step_compute_only\": %.4f, \"compute_only_value\": %.3f, \"host\": \"one process, one host thread per device, C ABI only "
                "(tools/multi_gpu_bench.cpp)\", \"gather\": \"lr_poly_copy_peer of %d x %.1f MiB to device 0 in chunks of %d products per device, one copy "
                "stream per peer, overlapped with the next chunk's kernels; lr_context_wait_peer_copies on the root\", \"gather_bytes_to_root\": %.0f, "
                "\"roofline\": {\"bound\": \"hbm\", \"achieved\": %.2f, \"peak\": 8000.0, \"unit\": \"GB/s\", \"frac\": %.5f, \"per\": \"GPU, products only\", "
                "\"algorithmic_bytes_per_product\": %.0f}, \"placement_ok\": %s, \"placement_units_checked\": %lld, \"steps\": %d, \"warmup\": %d}\n",
                total * steps / full, set.c_str(), logn, nq, np, beta, level, total, units, chunk, G, have, full / steps * 1e3, comp / steps * 1e3,
                total * steps / comp, total, ct_mib, chunk, (double)(total - units) * 2 * nq * N * 8, alg * units * steps / comp / 1e9,
                alg * units * steps / comp / 1e9 / 8000.0, alg, placement_bad.load() == 0 ? "true" : "false", checked.load(), steps, warmup);
    return placement_bad.load() == 0 ? 0 : 3;
}
