// lr_abi_batcher.cpp -- C ABI: lr_ckks_batcher, the merger of concurrent one-ciphertext calls into batched launches.
#include "lr_host.hpp"

// ------------------------------------------------------------------------------------------
// Batcher: the reference's concurrency model is one evaluator per goroutine, one ciphertext per call
// (examples/dbfv/psi/psi.go:215-233).  On this device a product of one ciphertext fills a fraction of the chip and the streams of
// many host threads do not add up (profiles/r03: 16 threads x batch 1 = 9.1k products/s against 21k/s for one batched call).  The
// batcher turns concurrent calls back into batched launches: a call queues its request; whichever caller finds a free lane takes
// every queued request with the same (level, key) up to max_batch, runs them as ONE MulRelin whose first kernel reads the operands
// through a pointer table and whose results are scattered to the callers' polys by one copy kernel, waits for the lane's stream and
// wakes the callers.  No thread of its own, no timer: while a lane runs, arrivals queue up and form the next batch.
// ------------------------------------------------------------------------------------------
struct lr_ckks_batcher_request {
    int level = 0, polys = 0;
    int kind = 0;              // 0: MulRelin (a0, a1) x (b0, b1); 1: rotation / conjugation of (a0, a1) by `gen` with the key `evk`
    u64 gen = 0;
    const lr_poly *a0 = nullptr, *a1 = nullptr, *b0 = nullptr, *b1 = nullptr, *evk = nullptr;
    lr_poly *o0 = nullptr, *o1 = nullptr;
    bool done = false;
    int status = LR_OK;
    std::string error;
    // requests that may share a launch: same operation, level and key image
    bool same_batch(const lr_ckks_batcher_request &o) const { return kind == o.kind && gen == o.gen && level == o.level && evk == o.evk; }
};
struct lr_ckks_batcher_lane {
    lr_ckks_plan *plan = nullptr;
    bool busy = false;
    u64 **h_table = nullptr;   // pinned: [4 * max_batch] operand pointers, then [2 * max_batch] result pointers
    u64 **d_table = nullptr;
    Pool o0, o1;               // staged results
    hipStream_t stream = nullptr;   // created here, set on the lane's two contexts for the batcher's lifetime
    std::vector<lr_ckks_batcher_request *> take;    // the batch being run; reserved at creation, so that forming a batch allocates nothing
    int device() const { return plan ? plan->device : 0; }
};
struct lr_ckks_batcher : BatchQueue<lr_ckks_batcher_request, lr_ckks_batcher_lane> {
    typedef lr_ckks_batcher_request Request;
    typedef lr_ckks_batcher_lane Lane;
};

namespace lr_host {

int batcher_run(lr_ckks_batcher *B, lr_ckks_batcher::Lane &lane, const std::vector<lr_ckks_batcher::Request *> &reqs) {
    lr_ckks_plan *pl = lane.plan;
    lr_context *cQ = pl->cQ;
    const int level = reqs[0]->level, L1 = level + 1, n = (int)cQ->h.N;
    const long long s = (long long)L1 * n;
    LR_HIP(hipSetDevice(cQ->device));
    int batch = 0;
    const int mb = B->max_batch;
    for (const auto *r : reqs)
        for (int i = 0; i < r->polys; ++i, ++batch) {
            if (r->kind == 0) {
                lane.h_table[4 * batch + 0] = r->a0->d + i * r->a0->stride();
                lane.h_table[4 * batch + 1] = r->a1->d + i * r->a1->stride();
                lane.h_table[4 * batch + 2] = r->b0->d + i * r->b0->stride();
                lane.h_table[4 * batch + 3] = r->b1->d + i * r->b1->stride();
            }
            lane.h_table[4 * mb + 2 * batch + 0] = r->o0->d + i * r->o0->stride();
            lane.h_table[4 * mb + 2 * batch + 1] = r->o1->d + i * r->o1->stride();
        }
    LR_TRY(lane.o0.ensure(cQ, (size_t)batch * s));
    LR_TRY(lane.o1.ensure(cQ, (size_t)batch * s));
    if (reqs[0]->kind == 1) {
        // rotations: table rows [0, batch) = the first components, [batch, 2 batch) = the second ones (lr_ckks_rotate, batched)
        int k = 0;
        for (const auto *r : reqs)
            for (int i = 0; i < r->polys; ++i, ++k) {
                lane.h_table[k] = r->a0->d + i * r->a0->stride();
                lane.h_table[batch + k] = r->a1->d + i * r->a1->stride();
            }
        LR_HIP(hipMemcpyAsync(lane.d_table, lane.h_table, (size_t)6 * mb * sizeof(u64 *), hipMemcpyHostToDevice, cQ->stream));
        LR_TRY(same_stream(pl->cQ, pl->cP));
        for (Pool *p : {&pl->c0, &pl->c2x, &pl->q1, &pl->q2}) LR_TRY(p->ensure(cQ, (size_t)batch * s));
        const u64 *const *tab = (const u64 *const *)lane.d_table;
        LR_TRY(run_permute_ntt(cQ, L1, batch, nullptr, 0, pl->c0.d, s, reqs[0]->gen, tab));             // ckks/evaluator.go:1458
        LR_TRY(run_permute_ntt(cQ, L1, batch, nullptr, 0, pl->c2x.d, s, reqs[0]->gen, tab + batch));    // :1459
        KeySwitchEpilogue fin{lane.o0.d, lane.o1.d, s, pl->c0.d, nullptr, s};
        LR_TRY(switch_keys_core(pl, level, batch, pl->c2x.d, s, reqs[0]->evk, pl->q1.d, s, pl->q2.d, s, &fin));   // :1464-1467
    } else {
    LR_HIP(hipMemcpyAsync(lane.d_table, lane.h_table, (size_t)6 * mb * sizeof(u64 *), hipMemcpyHostToDevice, cQ->stream));
    TensorLaunch T{};
    T.table = (const u64 *const *)lane.d_table;
    LR_TRY(mulrelin_core(pl, level, batch, T, reqs[0]->evk, lane.o0.d, lane.o1.d, s));
    }
    ScatterLaunch S{{lane.o0.d, lane.o1.d}, s, lane.d_table + 4 * mb, 2, n};
    LR_HIP(launch_scatter(S, L1, batch, cQ->stream));
    LR_HIP(hipStreamSynchronize(cQ->stream));
    return LR_OK;
}

}  // namespace lr_host

extern "C" int lr_ckks_batcher_create(lr_ckks_plan *const *plans, int n_lanes, lr_ckks_batcher **out) {
    return guarded([&]() -> int {
    if (!plans || !out || n_lanes < 1) return fail(LR_ERR_ARG, "plans / out null or no lanes");
    struct Undo {   // a creation that fails half-way takes the lanes built so far down again (streams, tables)
        void operator()(lr_ckks_batcher *b) const { lr_ckks_batcher_destroy(b); }
    };
    std::unique_ptr<lr_ckks_batcher, Undo> B(new lr_ckks_batcher);
    B->max_batch = plans[0] ? plans[0]->max_batch : 0;
    for (int i = 0; i < n_lanes; ++i) {
        lr_ckks_plan *pl = plans[i];
        if (!pl) return fail(LR_ERR_ARG, "null plan");
        if (pl->max_batch != B->max_batch || pl->cQ->h.N != plans[0]->cQ->h.N || pl->cQ->h.q != plans[0]->cQ->h.q ||
            pl->cP->h.q != plans[0]->cP->h.q || pl->device != plans[0]->device)
            return fail(LR_ERR_SHAPE, "the lanes' plans differ in ring, device or max_batch");
        for (int j = 0; j < i; ++j)
            if (plans[j] == pl || plans[j]->cQ == pl->cQ || plans[j]->cP == pl->cP)
                return fail(LR_ERR_ARG, "every lane needs its own plan over its own pair of contexts");
        if (pl->lane_of) return fail(LR_ERR_ARG, "a plan can be the lane of one batcher only");
    }
    LR_HIP(hipSetDevice(plans[0]->device));
    B->lanes.resize(n_lanes);
    for (int i = 0; i < n_lanes; ++i) {
        auto &ln = B->lanes[i];
        ln.plan = plans[i];
        if (!ln.plan->lane_of) standalone_plans(ln.plan->device).fetch_sub(1);
        ln.plan->lane_of = B.get();
        ln.plan->cQ->lane_of = ln.plan->cP->lane_of = B.get();
        ln.take.reserve((size_t)std::max(1, B->max_batch));
        LR_HIP(create_stream(&ln.stream, (i + 1) % 3));   // lane 0: greatest priority, lane 1: least, lane 2: default, ...
        LR_TRY(lr_context_set_stream(ln.plan->cQ, ln.stream));
        LR_TRY(lr_context_set_stream(ln.plan->cP, ln.stream));
        LR_HIP(hipHostMalloc((void **)&ln.h_table, (size_t)6 * B->max_batch * sizeof(u64 *)));
        LR_HIP(hipMalloc((void **)&ln.d_table, (size_t)6 * B->max_batch * sizeof(u64 *)));
    }
    *out = B.release();
    return LR_OK;
    });
}

extern "C" void lr_ckks_batcher_destroy(lr_ckks_batcher *B) {
    if (!B) return;
    for (auto &ln : B->lanes) {
        if (ln.plan && ln.plan->lane_of == B) {
            ln.plan->lane_of = nullptr;
            standalone_plans(ln.plan->device).fetch_add(1);
        }
        if (ln.plan) {
            for (lr_context *c : {ln.plan->cQ, ln.plan->cP})
                if (c->lane_of == B) c->lane_of = nullptr;
        }
        if (ln.stream) {   // back to the library's stream (ordered behind the lane's work), then the lane stream can go
            for (lr_context *c : {ln.plan->cQ, ln.plan->cP}) {
                if (lr_context_set_stream(c, nullptr) != LR_OK && c->stream == ln.stream) {
                    // the ordered hand-over failed: drain the device and put the library's stream back by hand, so that the context
                    // does not keep a handle to the stream destroyed below
                    (void)hipDeviceSynchronize();
                    (void)hipGetLastError();
                    c->stream = shared_stream(c->device);
                }
            }
            (void)hipStreamSynchronize(ln.stream);
            (void)hipStreamDestroy(ln.stream);
        }
        if (ln.h_table) (void)hipHostFree(ln.h_table);
        if (ln.d_table) (void)hipFree(ln.d_table);
    }
    delete B;
}

extern "C" int lr_ckks_batcher_stats(lr_ckks_batcher *B, uint64_t *batches, uint64_t *products, int *largest) {
    return guarded([&]() -> int {
    if (!B) return fail(LR_ERR_ARG, "null batcher");
    std::lock_guard<std::mutex> g(B->m);
    if (batches) *batches = B->batches;
    if (products) *products = B->products;
    if (largest) *largest = B->largest;
    return LR_OK;
    });
}

namespace lr_host {

// kind 0: MulRelin of (a0, a1) x (b0, b1); kind 1: rotation / conjugation of (a0, a1) by the Galois element `gen` (b0 = b1 = null)
int batcher_submit(lr_ckks_batcher *B, int kind, int level, const lr_poly *a0, const lr_poly *a1, const lr_poly *b0, const lr_poly *b1,
                   u64 gen, const lr_poly *evk, lr_poly *o0, lr_poly *o1) {
    if (!B || !a0 || !a1 || (kind == 0 && (!b0 || !b1)) || !evk || !o0 || !o1) return fail(LR_ERR_ARG, "null argument");
    lr_ckks_plan *pl0 = B->lanes[0].plan;
    if (level < 0 || level + 1 > pl0->cQ->h.L()) return fail(LR_ERR_SHAPE, "level out of range");
    const int polys = a0->batch;
    if (polys < 1 || polys > B->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the batcher's max_batch");
    if (kind == 0) {
        b0 = b0 ? b0 : a0;
        b1 = b1 ? b1 : a1;
    } else {
        b0 = a0;      // (checked twice below, never read)
        b1 = a1;
    }
    for (const lr_poly *p : {a0, a1, b0, b1, (const lr_poly *)o0, (const lr_poly *)o1}) {
        LR_TRY(check_ct(pl0, level, p, polys));
        if (p->device != pl0->device) return fail(LR_ERR_ARG, "poly lives on another device than the batcher");
    }
    const int beta = (level + 1 + pl0->cP->h.L() - 1) / pl0->cP->h.L();
    if (evk->N != pl0->cQ->h.N || evk->limbs < pl0->cQ->h.L() + pl0->cP->h.L() || evk->batch < 2 * beta)
        return fail(LR_ERR_SHAPE, "evaluation key image: limbs or digits");
    // the operands were produced on the streams of the callers' own contexts: finished before another stream reads them
    LR_HIP(hipSetDevice(pl0->device));
    {
        hipStream_t seen[6];
        int ns = 0;
        for (const lr_poly *p : {a0, a1, b0, b1, (const lr_poly *)o0, (const lr_poly *)o1}) {
            if (!p->ctx) continue;
            hipStream_t st = p->ctx->stream;
            bool dup = false;
            for (int i = 0; i < ns; ++i) dup = dup || seen[i] == st;
            if (dup) continue;
            seen[ns++] = st;
            LR_HIP(hipStreamSynchronize(st));
        }
    }
    lr_ckks_batcher::Request req;
    req.kind = kind;
    req.gen = gen;
    req.level = level; req.polys = polys;
    req.a0 = a0; req.a1 = a1; req.b0 = b0; req.b1 = b1; req.evk = evk; req.o0 = o0; req.o1 = o1;
    return B->submit(req, [B](lr_ckks_batcher::Lane &lane, const std::vector<lr_ckks_batcher::Request *> &take) { return batcher_run(B, lane, take); });
}

}  // namespace lr_host

extern "C" int lr_ckks_batcher_mulrelin(lr_ckks_batcher *B, int level, const lr_poly *a0, const lr_poly *a1, const lr_poly *b0,
                                        const lr_poly *b1, const lr_poly *evk, lr_poly *o0, lr_poly *o1) {
    return guarded([&]() -> int {
    if (!b0 || !b1) return fail(LR_ERR_ARG, "null argument");
    return batcher_submit(B, 0, level, a0, a1, b0, b1, 0, evk, o0, o1);
    });
}

extern "C" int lr_ckks_batcher_rotate(lr_ckks_batcher *B, int level, const lr_poly *c0, const lr_poly *c1, uint64_t gen, const lr_poly *rotkey,
                                      lr_poly *o0, lr_poly *o1) {
    return guarded([&]() -> int {
    if (!B) return fail(LR_ERR_ARG, "null argument");
    const u64 two_n = B->lanes[0].plan->cQ->h.N << 1;
    return batcher_submit(B, 1, level, c0, c1, nullptr, nullptr, gen & (two_n - 1), rotkey, o0, o1);
    });
}
