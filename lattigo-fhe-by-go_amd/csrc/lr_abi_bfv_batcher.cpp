// lr_abi_bfv_batcher.cpp -- C ABI: lr_bfv_batcher.  The workload the reference itself hands to a pool of goroutines is BFV: every task of
// examples/dbfv/psi/psi.go:215-233 runs evaluator.Mul and evaluator.Relinearize on one ciphertext pair.  One such call fills a fraction of
// the device (PN14QP438, one pair: Mul 118 us, Relinearize 115 us; a batch of 256: 13 us and 6 us per pair), so concurrent calls are merged
// into batched ones exactly like the CKKS batcher's (BatchQueue, lr_host.hpp): the callers' operand polys are gathered into the lane's
// staging polys by one kernel through a pointer table, the lane's plans run lr_bfv_mul / lr_bfv_relinearize on the staged batch, and one
// kernel scatters the results to the callers' polys.  Same bits as the direct calls (the pipelines are the same functions).
#include "lr_host.hpp"

struct lr_bfv_batcher_request {
    int kind = 0, polys = 0;          // kind 0: Mul (a0, a1) x (b0, b1) -> (o0, o1, o2); 1: Relinearize of (a0, a1, b0 = c2) with evk -> (o0, o1)
    const lr_poly *a0 = nullptr, *a1 = nullptr, *b0 = nullptr, *b1 = nullptr, *evk = nullptr;
    lr_poly *o0 = nullptr, *o1 = nullptr, *o2 = nullptr;
    bool done = false;
    int status = LR_OK;
    std::string error;
    bool same_batch(const lr_bfv_batcher_request &o) const { return kind == o.kind && evk == o.evk; }
};
struct lr_bfv_batcher_lane {
    lr_bfv_plan *mul = nullptr;
    lr_ckks_plan *ks = nullptr;       // the key-switch half of bfv.NewEvaluator over (contextQ, contextP); may be null (no Relinearize)
    bool busy = false;
    u64 **h_table = nullptr, **d_table = nullptr;     // [4 * max_batch] operand pointers, then [3 * max_batch] result pointers
    lr_poly *in[4] = {nullptr, nullptr, nullptr, nullptr}, *out[3] = {nullptr, nullptr, nullptr};      // staging, max_batch polys each
    hipStream_t stream = nullptr;
    std::vector<lr_bfv_batcher_request *> take;
    int device() const { return mul ? mul->device : 0; }
};
struct lr_bfv_batcher : BatchQueue<lr_bfv_batcher_request, lr_bfv_batcher_lane> {
    typedef lr_bfv_batcher_request Request;
    typedef lr_bfv_batcher_lane Lane;
};

namespace lr_host {

// the first `batch` polys of a staging poly as a poly of its own (a host object; the memory stays the lane's)
struct StagedView {
    lr_poly v;
    StagedView(const lr_poly *full, int batch) : v(*full) {
        v.owned = false;
        v.batch = batch;
    }
};

int bfv_batcher_run(lr_bfv_batcher *B, lr_bfv_batcher::Lane &lane, const std::vector<lr_bfv_batcher::Request *> &reqs) {
    lr_context *cQ = lane.mul->cQ;
    const int nQ = cQ->h.L(), n = (int)cQ->h.N, mb = B->max_batch;
    const int kind = reqs[0]->kind;
    const int n_in = kind == 0 ? 4 : 3, n_out = kind == 0 ? 3 : 2;
    LR_HIP(hipSetDevice(cQ->device));
    int batch = 0;
    for (const auto *r : reqs)
        for (int i = 0; i < r->polys; ++i, ++batch) {
            const lr_poly *ins[4] = {r->a0, r->a1, r->b0, r->b1};
            lr_poly *outs[3] = {r->o0, r->o1, r->o2};
            for (int k = 0; k < n_in; ++k) lane.h_table[n_in * batch + k] = ins[k]->d + i * ins[k]->stride();
            for (int k = 0; k < n_out; ++k) lane.h_table[4 * mb + n_out * batch + k] = outs[k]->d + i * outs[k]->stride();
        }
    LR_HIP(hipMemcpyAsync(lane.d_table, lane.h_table, (size_t)7 * mb * sizeof(u64 *), hipMemcpyHostToDevice, cQ->stream));
    GatherLaunch G;
    for (int k = 0; k < 4; ++k) G.dst[k] = lane.in[k]->d;
    G.stride = lane.in[0]->stride();
    G.table = (const u64 *const *)lane.d_table;
    G.per_poly = n_in;
    G.n = n;
    LR_HIP(launch_gather(G, nQ, batch, cQ->stream));
    StagedView i0(lane.in[0], batch), i1(lane.in[1], batch), i2(lane.in[2], batch), i3(lane.in[3], batch);
    StagedView o0(lane.out[0], batch), o1(lane.out[1], batch), o2(lane.out[2], batch);
    if (kind == 0) LR_TRY(lr_bfv_mul(lane.mul, &i0.v, &i1.v, &i2.v, &i3.v, &o0.v, &o1.v, &o2.v));                       // bfv/evaluator.go:467
    else LR_TRY(lr_bfv_relinearize(lane.ks, &i0.v, &i1.v, &i2.v, reqs[0]->evk, &o0.v, &o1.v));                             // :512
    ScatterLaunch S{{lane.out[0]->d, lane.out[1]->d, lane.out[2]->d, nullptr}, lane.out[0]->stride(), lane.d_table + 4 * mb, n_out, n};
    LR_HIP(launch_scatter(S, nQ, batch, cQ->stream));
    LR_HIP(hipStreamSynchronize(cQ->stream));
    return LR_OK;
}

int bfv_batcher_submit(lr_bfv_batcher *B, int kind, const lr_poly *a0, const lr_poly *a1, const lr_poly *b0, const lr_poly *b1, const lr_poly *evk,
                       lr_poly *o0, lr_poly *o1, lr_poly *o2) {
    if (!B || !a0 || !a1 || !b0 || (kind == 0 && (!b1 || !o2)) || (kind == 1 && !evk) || !o0 || !o1) return fail(LR_ERR_ARG, "null argument");
    lr_bfv_plan *pl0 = B->lanes[0].mul;
    lr_context *cQ = pl0->cQ;
    if (kind == 1 && !B->lanes[0].ks) return fail(LR_ERR_ARG, "this batcher was created without key-switch plans: no Relinearize");
    const int polys = a0->batch, nQ = cQ->h.L();
    if (polys < 1 || polys > B->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the batcher's max_batch");
    const lr_poly *all[7] = {a0, a1, b0, kind == 0 ? b1 : a0, o0, o1, kind == 0 ? o2 : o0};
    for (const lr_poly *p : all) {
        if (p->N != cQ->h.N || p->limbs < nQ || p->batch != polys) return fail(LR_ERR_SHAPE, "BFV batcher: polys must hold |Q| limbs and share the batch");
        if (p->device != pl0->device) return fail(LR_ERR_ARG, "poly lives on another device than the batcher");
    }
    if (o0->d == o1->d || (kind == 0 && (o2->d == o0->d || o2->d == o1->d))) return fail(LR_ERR_ARG, "the result polys must be distinct");
    if (kind == 1) {
        lr_ckks_plan *ks = B->lanes[0].ks;
        const int nP = ks->cP->h.L(), beta = (nQ + nP - 1) / nP;
        if (evk->N != cQ->h.N || evk->limbs < nQ + nP || evk->batch < 2 * beta) return fail(LR_ERR_SHAPE, "evaluation key image: limbs or digits");
    }
    // the operands were produced on the streams of the callers' own contexts: finished before a lane's stream reads them
    LR_HIP(hipSetDevice(pl0->device));
    {
        hipStream_t seen[7];
        int ns = 0;
        for (const lr_poly *p : all) {
            if (!p->ctx) continue;
            hipStream_t st = p->ctx->stream;
            bool dup = false;
            for (int i = 0; i < ns; ++i) dup = dup || seen[i] == st;
            if (dup) continue;
            seen[ns++] = st;
            LR_HIP(hipStreamSynchronize(st));
        }
    }
    lr_bfv_batcher::Request req;
    req.kind = kind;
    req.polys = polys;
    req.a0 = a0; req.a1 = a1; req.b0 = b0; req.b1 = b1; req.evk = evk; req.o0 = o0; req.o1 = o1; req.o2 = o2;
    return B->submit(req, [B](lr_bfv_batcher::Lane &lane, const std::vector<lr_bfv_batcher::Request *> &take) { return bfv_batcher_run(B, lane, take); });
}

}  // namespace lr_host

extern "C" int lr_bfv_batcher_create(lr_bfv_plan *const *mul_plans, lr_ckks_plan *const *ks_plans, int n_lanes, lr_bfv_batcher **out) {
    return guarded([&]() -> int {
    if (!mul_plans || !out || n_lanes < 1) return fail(LR_ERR_ARG, "plans / out null or no lanes");
    struct Undo {   // a creation that fails half-way takes the lanes built so far down again (streams, tables, staging)
        void operator()(lr_bfv_batcher *b) const { lr_bfv_batcher_destroy(b); }
    };
    std::unique_ptr<lr_bfv_batcher, Undo> B(new lr_bfv_batcher);
    B->max_batch = mul_plans[0] ? mul_plans[0]->max_batch : 0;
    for (int i = 0; i < n_lanes; ++i) {
        lr_bfv_plan *pl = mul_plans[i];
        if (!pl) return fail(LR_ERR_ARG, "null plan");
        lr_ckks_plan *ks = ks_plans ? ks_plans[i] : nullptr;
        if (ks_plans && !ks) return fail(LR_ERR_ARG, "key-switch plans: all lanes or none");
        if (pl->max_batch != B->max_batch || pl->cQ->h.N != mul_plans[0]->cQ->h.N || pl->cQ->h.q != mul_plans[0]->cQ->h.q ||
            pl->cM->h.q != mul_plans[0]->cM->h.q || pl->device != mul_plans[0]->device || pl->t != mul_plans[0]->t)
            return fail(LR_ERR_SHAPE, "the lanes' plans differ in ring, device, t or max_batch");
        if (ks && (ks->cQ != pl->cQ || ks->max_batch < B->max_batch || ks->cP->h.q != ks_plans[0]->cP->h.q))
            return fail(LR_ERR_SHAPE, "a lane's key-switch plan must be built over the lane's contextQ, with the same max_batch and special primes");
        for (int j = 0; j < i; ++j)
            if (mul_plans[j] == pl || mul_plans[j]->cQ == pl->cQ || mul_plans[j]->cM == pl->cM || (ks && (ks_plans[j] == ks || ks_plans[j]->cP == ks->cP)))
                return fail(LR_ERR_ARG, "every lane needs its own plans over its own contexts");
        if (pl->lane_of || (ks && ks->lane_of)) return fail(LR_ERR_ARG, "a plan can be the lane of one batcher only");
    }
    LR_HIP(hipSetDevice(mul_plans[0]->device));
    B->lanes.resize(n_lanes);
    for (int i = 0; i < n_lanes; ++i) {
        auto &ln = B->lanes[i];
        ln.mul = mul_plans[i];
        ln.ks = ks_plans ? ks_plans[i] : nullptr;
        ln.mul->lane_of = B.get();
        if (ln.ks) {
            standalone_plans(ln.ks->device).fetch_sub(1);
            ln.ks->lane_of = B.get();
        }
        ln.take.reserve((size_t)std::max(1, B->max_batch));
        LR_HIP(create_stream(&ln.stream, (i + 1) % 3));   // lanes in different priority classes: different hardware queues (see the CKKS batcher)
        for (lr_context *c : {ln.mul->cQ, ln.mul->cM, ln.ks ? ln.ks->cP : (lr_context *)nullptr}) {
            if (!c) continue;
            if (c->lane_of) return fail(LR_ERR_ARG, "a context can serve one batcher lane only");
            c->lane_of = B.get();
            LR_TRY(lr_context_set_stream(c, ln.stream));
        }
        LR_HIP(hipHostMalloc((void **)&ln.h_table, (size_t)7 * B->max_batch * sizeof(u64 *)));
        LR_HIP(hipMalloc((void **)&ln.d_table, (size_t)7 * B->max_batch * sizeof(u64 *)));
        const int nQ = ln.mul->cQ->h.L();
        for (int k = 0; k < 4; ++k) LR_TRY(lr_poly_alloc(ln.mul->cQ, nQ, B->max_batch, &ln.in[k]));
        for (int k = 0; k < 3; ++k) LR_TRY(lr_poly_alloc(ln.mul->cQ, nQ, B->max_batch, &ln.out[k]));
    }
    *out = B.release();
    return LR_OK;
    });
}

extern "C" void lr_bfv_batcher_destroy(lr_bfv_batcher *B) {
    if (!B) return;
    for (auto &ln : B->lanes) {
        if (!ln.mul) continue;
        std::vector<lr_context *> ctxs = {ln.mul->cQ, ln.mul->cM};
        if (ln.ks) ctxs.push_back(ln.ks->cP);
        if (ln.mul->lane_of == B) ln.mul->lane_of = nullptr;
        if (ln.ks && ln.ks->lane_of == B) {
            ln.ks->lane_of = nullptr;
            standalone_plans(ln.ks->device).fetch_add(1);
        }
        for (lr_context *c : ctxs)
            if (c->lane_of == B) c->lane_of = nullptr;
        if (ln.stream) {   // back to the library's stream (ordered behind the lane's work), then the lane stream can go
            for (lr_context *c : ctxs) {
                if (c->stream != ln.stream) continue;
                if (lr_context_set_stream(c, nullptr) != LR_OK && c->stream == ln.stream) {
                    (void)hipDeviceSynchronize();
                    (void)hipGetLastError();
                    c->stream = shared_stream(c->device);
                }
            }
            (void)hipStreamSynchronize(ln.stream);
            (void)hipStreamDestroy(ln.stream);
        }
        for (lr_poly *p : ln.in) (void)lr_poly_free(p);
        for (lr_poly *p : ln.out) (void)lr_poly_free(p);
        if (ln.h_table) (void)hipHostFree(ln.h_table);
        if (ln.d_table) (void)hipFree(ln.d_table);
    }
    delete B;
}

extern "C" int lr_bfv_batcher_stats(lr_bfv_batcher *B, uint64_t *batches, uint64_t *products, int *largest) {
    return guarded([&]() -> int {
    if (!B) return fail(LR_ERR_ARG, "null batcher");
    std::lock_guard<std::mutex> g(B->m);
    if (batches) *batches = B->batches;
    if (products) *products = B->products;
    if (largest) *largest = B->largest;
    return LR_OK;
    });
}

extern "C" int lr_bfv_batcher_mul(lr_bfv_batcher *B, const lr_poly *ct0_c0, const lr_poly *ct0_c1, const lr_poly *ct1_c0, const lr_poly *ct1_c1,
                                  lr_poly *out_c0, lr_poly *out_c1, lr_poly *out_c2) {
    return guarded([&]() -> int { return bfv_batcher_submit(B, 0, ct0_c0, ct0_c1, ct1_c0, ct1_c1, nullptr, out_c0, out_c1, out_c2); });
}

extern "C" int lr_bfv_batcher_relinearize(lr_bfv_batcher *B, const lr_poly *c0, const lr_poly *c1, const lr_poly *c2, const lr_poly *evk,
                                          lr_poly *out0, lr_poly *out1) {
    return guarded([&]() -> int { return bfv_batcher_submit(B, 1, c0, c1, c2, nullptr, evk, out0, out1, nullptr); });
}
