// Host concurrency of the C ABI under the CPU sanitizers: the REAL host code -- lr_abi_*.cpp, lr_host.hpp, lr_precompute.cpp --
// compiled with g++ against the host-only HIP stand-in and the recording launch stubs of tests/cpp/hipstub/, driven by 64 threads that mix
//   * MulRelin and rotation requests through the batcher (two lanes), each caller checking that the tag of ITS operands came back,
//   * requests the batcher must refuse (level out of range, key image too small, batch above max_batch),
//   * an injected device failure in the middle of a batch (every caller of that batch gets the error, the lanes stay usable),
//   * direct pipelines on per-thread plans over contexts SHARED by the threads (the scratch pool's leases, the standalone-plan counter),
//   * rescales and in-place monomial multiplications on a shared context (ScratchLease from many threads),
//   * handle churn (poly alloc / free, plan create / destroy) and peer copies between the two "devices".
// Built twice by tests/test_host_sanitizers.py: -fsanitize=thread and -fsanitize=address,undefined.  Exit code 0 = every check held;
// a sanitizer report aborts the run.  Nothing here computes: parity is the GPU suite's business.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "lattigo_ring.h"

extern "C" {
void hipstub_fail_memcpy_async_after(long calls, long bytes);
long hipstub_live_allocations(void);
}

static std::atomic<int> g_fail{0};
#define CHECK(cond)                                                                          \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            std::fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #cond, __LINE__, lr_last_error_string()); \
            g_fail.fetch_add(1);                                                             \
        }                                                                                    \
    } while (0)
#define OK(x) CHECK((x) == LR_OK)

// DefaultParams[PN15QP880]'s first primes: congruent to 1 modulo 2^16, so NTT-friendly for every degree used here
static const uint64_t Qm[4] = {1125899908022273ull, 1099512938497ull, 1099514314753ull, 1099515691009ull};
static const uint64_t Pm[2] = {1125899908612097ull, 1125899909398529ull};
static const int NQ = 4, NP = 2, LEVEL = 3, BETA = 2;
// DefaultParams[PN16QP1761]'s first primes: congruent to 1 modulo 2^17 (the N = 2^16 section)
static const uint64_t Qm16[4] = {36028797019488257ull, 35184372744193ull, 35184373006337ull, 35184376545281ull};
static const uint64_t Pm16[2] = {36028797023420417ull, 36028797024206849ull};

struct Ct {
    lr_poly *c[2];
};
static lr_poly *poly(lr_context *ctx, int limbs, int batch, uint64_t tag) {
    lr_poly *p = nullptr;
    OK(lr_poly_alloc(ctx, limbs, batch, &p));
    uint64_t N = 0;
    OK(lr_context_info(ctx, &N, nullptr, nullptr));
    std::vector<uint64_t> h((size_t)batch * limbs * N, 0);
    for (int b = 0; b < batch; ++b) h[(size_t)b * limbs * N] = tag + (uint64_t)b;
    OK(lr_poly_upload_dense(p, h.data(), h.size()));
    return p;
}
static uint64_t tag_of(const lr_poly *p, int b, int limbs, uint64_t N) {
    std::vector<uint64_t> h((size_t)(b + 1) * limbs * N);
    int batch = 0;
    OK(lr_poly_info(p, nullptr, nullptr, &batch, nullptr));
    h.resize((size_t)batch * limbs * N);
    OK(lr_poly_download_dense(p, h.data(), h.size()));
    return h[(size_t)b * limbs * N];
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? std::atoi(argv[1]) : 64, ITERS = argc > 2 ? std::atoi(argv[2]) : 12;
    const uint64_t N = 1 << 12;
    // ---- a lone plan at N = 2^16 forks its independent launches to an auxiliary stream (PlanFork): one ciphertext, device 1
    {
        lr_context *q = nullptr, *p = nullptr;
        OK(lr_context_create(1 << 16, Qm16, NQ, 1, &q));
        OK(lr_context_create(1 << 16, Pm16, NP, 1, &p));
        lr_ckks_plan *pl = nullptr;
        OK(lr_ckks_plan_create(q, p, 1, &pl));
        lr_poly *key = poly(q, NQ + NP, 2 * BETA, 5), *a0 = poly(q, NQ, 1, 100), *a1 = poly(q, NQ, 1, 200), *o0 = poly(q, NQ, 1, 0), *o1 = poly(q, NQ, 1, 0);
        for (int i = 0; i < 3; ++i) OK(lr_ckks_mulrelin(pl, LEVEL, a0, a1, a0, a1, key, o0, o1));
        uint64_t forks = 0;
        OK(lr_ckks_plan_stats(pl, &forks, nullptr));
        CHECK(forks > 0);
        CHECK(tag_of(o0, 0, NQ, 1 << 16) == 100 && tag_of(o1, 0, NQ, 1 << 16) == 200);
        for (lr_poly *x : {key, a0, a1, o0, o1}) OK(lr_poly_free(x));
        OK(lr_ckks_plan_destroy(pl));
        OK(lr_context_destroy(q));
        OK(lr_context_destroy(p));
    }
    // ---- the batcher: two lanes over their own contexts
    const int MAXB = 8, LANES = 2;
    lr_context *lq[LANES], *lp[LANES];
    lr_ckks_plan *lplan[LANES];
    for (int i = 0; i < LANES; ++i) {
        OK(lr_context_create(N, Qm, NQ, 0, &lq[i]));
        OK(lr_context_create(N, Pm, NP, 0, &lp[i]));
        OK(lr_ckks_plan_create(lq[i], lp[i], MAXB, &lplan[i]));
    }
    lr_ckks_batcher *bat = nullptr;
    OK(lr_ckks_batcher_create(lplan, LANES, &bat));
    CHECK(lr_ckks_plan_destroy(lplan[0]) == LR_ERR_ARG);          // a lane cannot go while its batcher lives
    CHECK(lr_context_destroy(lq[1]) == LR_ERR_ARG);
    lr_poly *key = poly(lq[0], NQ + NP, 2 * BETA, 7), *small_key = poly(lq[0], NQ + NP, 1, 7);
    // ---- the BFV batcher (Mul + Relinearize, the reference's own pooled workload): two lanes, each a Mul plan and a key-switch plan over one contextQ
    lr_context *bq[LANES], *bp[LANES], *bm[LANES];
    lr_bfv_plan *bmul[LANES];
    lr_ckks_plan *bks[LANES];
    for (int i = 0; i < LANES; ++i) {
        OK(lr_context_create(N, Qm, NQ, 0, &bq[i]));
        OK(lr_context_create(N, Pm, NP, 0, &bp[i]));
        OK(lr_context_create(N, Qm16, NQ, 0, &bm[i]));              // (any other NTT-friendly basis of the same length stands in for QMul)
        OK(lr_bfv_plan_create(bq[i], bm[i], 65537, MAXB, &bmul[i]));
        OK(lr_ckks_plan_create(bq[i], bp[i], MAXB, &bks[i]));
    }
    lr_bfv_batcher *bbat = nullptr;
    OK(lr_bfv_batcher_create(bmul, bks, LANES, &bbat));
    CHECK(lr_bfv_plan_destroy(bmul[1]) == LR_ERR_ARG && lr_ckks_plan_destroy(bks[0]) == LR_ERR_ARG && lr_context_destroy(bm[0]) == LR_ERR_ARG);
    lr_poly *bkey = poly(bq[0], NQ + NP, 2 * BETA, 11);
    // ---- contexts shared by all threads (immutable after creation; their scratch is leased per call), on both "devices"
    lr_context *sq = nullptr, *sp = nullptr, *rq = nullptr;
    OK(lr_context_create(N, Qm, NQ, 0, &sq));
    OK(lr_context_create(N, Pm, NP, 0, &sp));
    OK(lr_context_create(N, Qm, NQ, 1, &rq));                    // the "root" of the peer copies lives on device 1
    lr_poly *root = nullptr;
    OK(lr_poly_alloc(rq, NQ, T, &root));
    lr_poly *skey = poly(sq, NQ + NP, 2 * BETA, 9);
    std::atomic<int> injected_seen{0}, refused{0}, served{0};

    auto worker = [&](int t) {
        std::mt19937 rng(1234 + t);
        lr_context *mq = nullptr;                                    // the caller's own context, like every goroutine's evaluator
        OK(lr_context_create(N, Qm, NQ, 0, &mq));
        lr_ckks_plan *mine = nullptr;
        OK(lr_ckks_plan_create(sq, sp, 2, &mine));                   // own plan over the SHARED contexts
        const uint64_t base = 1000ull * (uint64_t)(t + 1);
        lr_poly *a0 = poly(mq, NQ, 1, base + 1), *a1 = poly(mq, NQ, 1, base + 2), *b0 = poly(mq, NQ, 1, base + 3), *b1 = poly(mq, NQ, 1, base + 4);
        lr_poly *o0 = poly(mq, NQ, 1, 0), *o1 = poly(mq, NQ, 1, 0);
        lr_poly *s0 = poly(sq, NQ, 2, base + 5), *s1 = poly(sq, NQ, 2, base + 7), *so0 = poly(sq, NQ, 2, 0), *so1 = poly(sq, NQ, 2, 0);
        for (int it = 0; it < ITERS; ++it) {
            switch (rng() % 9) {
            case 0:
            case 1: {   // MulRelin through the batcher: whatever batch the request lands in, the tags of THIS caller come back
                const int rc = lr_ckks_batcher_mulrelin(bat, LEVEL, a0, a1, b0, b1, key, o0, o1);
                if (rc == LR_OK) {
                    CHECK(tag_of(o0, 0, NQ, N) == base + 1 && tag_of(o1, 0, NQ, N) == base + 2);
                    served.fetch_add(1);
                } else {
                    CHECK(rc == LR_ERR_HIP && std::strstr(lr_last_error_string(), "injected") != nullptr);      // the injected failure, by name
                    injected_seen.fetch_add(1);
                }
                break;
            }
            case 2: {   // rotation through the batcher
                const int rc = lr_ckks_batcher_rotate(bat, LEVEL, a0, a1, 5 + 2 * (uint64_t)(rng() % 2), key, o0, o1);
                if (rc == LR_OK) {
                    CHECK(tag_of(o0, 0, NQ, N) == base + 1 && tag_of(o1, 0, NQ, N) == 0);
                    served.fetch_add(1);
                } else {
                    CHECK(rc == LR_ERR_HIP);
                    injected_seen.fetch_add(1);
                }
                break;
            }
            case 3: {   // requests the batcher refuses before they reach a lane
                CHECK(lr_ckks_batcher_mulrelin(bat, NQ, a0, a1, b0, b1, key, o0, o1) == LR_ERR_SHAPE);          // level out of range
                CHECK(lr_ckks_batcher_mulrelin(bat, LEVEL, a0, a1, b0, b1, small_key, o0, o1) == LR_ERR_SHAPE);  // key image with too few digits
                CHECK(lr_ckks_batcher_mulrelin(bat, LEVEL, a0, nullptr, b0, b1, key, o0, o1) == LR_ERR_ARG);
                refused.fetch_add(3);
                break;
            }
            case 4: {   // a direct pipeline on this thread's plan over the shared contexts
                OK(lr_ckks_mulrelin(mine, LEVEL, s0, s1, s0, s1, skey, so0, so1));
                CHECK(tag_of(so0, 1, NQ, N) == base + 6 && tag_of(so1, 1, NQ, N) == base + 8);
                OK(lr_ckks_rotate(mine, LEVEL, s0, s1, 5, skey, so0, so1));
                break;
            }
            case 5: {   // scratch leases on the shared context: rounding rescale (builds / reads the per-level table under its mutex), in-place monomial
                lr_poly *r = poly(sq, NQ, 2, base);
                OK(lr_div_round_by_last_modulus_ntt(sq, r));
                OK(lr_poly_set_limbs(r, NQ));
                OK(lr_div_floor_by_last_modulus(sq, r));
                OK(lr_poly_set_limbs(r, NQ));
                OK(lr_mult_by_monomial(sq, r, 3, r));
                // wire image out and back in: the big-endian staging buffer is a lease of the shared context's pool
                std::vector<uint8_t> image((size_t)NQ * N * 8 + 2);
                size_t written = 0;
                OK(lr_poly_marshal(r, 1, image.data(), image.size(), &written));
                CHECK(written == image.size() && image[1] == NQ);
                OK(lr_poly_unmarshal(r, 0, image.data(), written));
                CHECK(lr_poly_unmarshal(r, 0, image.data(), written - 8) != LR_OK);
                OK(lr_poly_free(r));
                break;
            }
            case 6: {   // handle churn
                lr_ckks_plan *tmp = nullptr;
                OK(lr_ckks_plan_create(sq, sp, 1, &tmp));
                lr_bext *bx = nullptr;
                OK(lr_bext_create(sq, sp, &bx));
                lr_poly *pp = poly(sp, NP, 2, 1), *pq = poly(sq, NQ, 2, 1);
                OK(lr_modup_split_qp(bx, LEVEL, pq, pp));
                OK(lr_moddown_split_ntt_pq(bx, LEVEL, pq, pp, pq));
                OK(lr_poly_free(pp));
                OK(lr_poly_free(pq));
                OK(lr_bext_destroy(bx));
                OK(lr_ckks_plan_destroy(tmp));
                break;
            }
            case 7: {   // BFV Mul then Relinearize through the BFV batcher (the stubs' Mul moves no tag: what is checked is that the calls come back,
                        // in order, with the staging and the pointer tables of the right size under the sanitizers)
                lr_poly *d2 = poly(mq, NQ, 1, 0);
                OK(lr_bfv_batcher_mul(bbat, a0, a1, b0, b1, o0, o1, d2));
                OK(lr_bfv_batcher_relinearize(bbat, o0, o1, d2, bkey, a1, b1));
                CHECK(lr_bfv_batcher_mul(bbat, a0, a1, b0, b1, o0, o0, d2) == LR_ERR_ARG);
                OK(lr_poly_free(d2));
                OK(lr_poly_free(a1));                                   // (a1 / b1 were overwritten: fresh tags for the other cases)
                OK(lr_poly_free(b1));
                a1 = poly(mq, NQ, 1, base + 2);
                b1 = poly(mq, NQ, 1, base + 4);
                break;
            }
            default: {  // a finished result goes to its slot on the other "device"
                OK(lr_poly_copy_peer(rq, root, t, mq, a0, 0, 1));
                break;
            }
            }
            if (t == 0 && it == ITERS / 2) hipstub_fail_memcpy_async_after(2, 6 * MAXB * 8);   // one failure: the third upload of a lane's pointer table from now
        }
        OK(lr_poly_copy_peer(rq, root, t, mq, a0, 0, 1));
        OK(lr_context_sync(mq));
        for (lr_poly *x : {a0, a1, b0, b1, o0, o1, s0, s1, so0, so1}) OK(lr_poly_free(x));
        OK(lr_ckks_plan_destroy(mine));
        OK(lr_context_destroy(mq));
    };
    std::vector<std::thread> ths;
    for (int t = 0; t < T; ++t) ths.emplace_back(worker, t);
    for (auto &th : ths) th.join();
    hipstub_fail_memcpy_async_after(-1, 0);
    OK(lr_context_wait_peer_copies(rq));
    OK(lr_context_sync(rq));
    for (int t = 0; t < T; ++t) CHECK(tag_of(root, t, NQ, N) == 1000ull * (uint64_t)(t + 1) + 1);
    uint64_t batches = 0, products = 0;
    int largest = 0;
    OK(lr_ckks_batcher_stats(bat, &batches, &products, &largest));
    CHECK(largest >= 1 && largest <= MAXB && batches >= 1 && products >= (uint64_t)served.load());
    // the lanes survived the injected failure: one more product
    {
        lr_poly *a = poly(sq, NQ, 1, 42), *o = poly(sq, NQ, 1, 0), *o2 = poly(sq, NQ, 1, 0);
        OK(lr_ckks_batcher_mulrelin(bat, LEVEL, a, a, a, a, key, o, o2));
        CHECK(tag_of(o, 0, NQ, N) == 42);
        for (lr_poly *x : {a, o, o2}) OK(lr_poly_free(x));
    }
    {
        uint64_t bb = 0, bpn = 0;
        int bl = 0;
        OK(lr_bfv_batcher_stats(bbat, &bb, &bpn, &bl));
        CHECK(bl <= MAXB && bpn >= bb);
    }
    lr_bfv_batcher_destroy(bbat);
    OK(lr_poly_free(bkey));
    for (int i = 0; i < LANES; ++i) {
        OK(lr_bfv_plan_destroy(bmul[i]));
        OK(lr_ckks_plan_destroy(bks[i]));
        for (lr_context *c : {bq[i], bp[i], bm[i]}) OK(lr_context_destroy(c));
    }
    lr_ckks_batcher_destroy(bat);
    for (lr_poly *x : {key, small_key, skey, root}) OK(lr_poly_free(x));
    for (int i = 0; i < LANES; ++i) {
        OK(lr_ckks_plan_destroy(lplan[i]));
        OK(lr_context_destroy(lq[i]));
        OK(lr_context_destroy(lp[i]));
    }
    for (lr_context *c : {sq, sp, rq}) OK(lr_context_destroy(c));
    CHECK(hipstub_live_allocations() == 0);                          // every device buffer and pinned table went with its handle
    std::printf("threads %d iterations %d: served %d, refused %d, callers of the failed batch %d, batches %llu (largest %d), failures %d\n", T, ITERS,
                served.load(), refused.load(), injected_seen.load(), (unsigned long long)batches, largest, g_fail.load());
    return g_fail.load() == 0 ? 0 : 1;
}
