// lr_host.hpp -- what the translation units of the C ABI (lr_abi_*.cpp) share: the error / exception boundary, the handle structs, the
// scratch pools and the raw-pointer forms of the ring operations that the pipelines compose.  Host code only; the kernels and their launch
// structs are in lr_device.hpp.  Everything here lives in namespace lr_host (or is a handle struct named by include/lattigo_ring.h).
#pragma once
#include "lattigo_ring.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "lr_device.hpp"
#include "lr_precompute.hpp"

using namespace lr;

namespace lr_host {

extern thread_local std::string g_error;      // lr_abi_core.cpp

inline int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

#define LR_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(LR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define LR_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != LR_OK) return rc_; \
    } while (0)

// Every extern "C" body runs inside guarded(): the host side uses std::vector / std::map / std::string / new, and no
// exception may cross the C boundary into a Go / Python / C caller (include/lattigo_ring.h, conventions).
template <class F>
int guarded(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try { g_error = "out of host memory"; } catch (...) {}
        return LR_ERR_NOMEM;
    } catch (const std::exception &e) {
        try { g_error = std::string("internal error: ") + e.what(); } catch (...) {}
        return LR_ERR_INTERNAL;
    } catch (...) {
        try { g_error = "internal error (unknown exception)"; } catch (...) {}
        return LR_ERR_INTERNAL;
    }
}

template <class T>
int to_device(T **dst, const T *src, size_t count) {
    *dst = nullptr;
    if (count == 0) return LR_OK;
    LR_HIP(hipMalloc((void **)dst, count * sizeof(T)));
    LR_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return LR_OK;
}

// One stream per device, shared by every handle on that device and never destroyed: operations of
// related handles (contextQ / contextP / basis extender / plan) are ordered by construction, and a
// poly can be released safely whatever order a garbage-collected host language frees handles in.
// The library's own streams.  The runtime binds a stream to one of a few hardware queues per priority class when it is created (four
// per class unless GPU_MAX_HW_QUEUES says otherwise), in an order that depends on every stream the process has created before; two
// streams on one hardware queue run one after the other.  Streams that must overlap -- a batcher's lanes, a plan's auxiliary stream
// beside the caller's -- therefore get different priority CLASSES, the one way to be sure of different queues (profiles/r03/
// hw_queues.txt: the same two lanes gave 10.8 k or 13.6 k products/s depending on what earlier legs of the process had created).
// cls: 0 = the default class (hipStreamCreateWithFlags), 1 = the device's greatest priority, 2 = its least.
hipError_t create_stream(hipStream_t *s, int cls = 0);
hipStream_t shared_stream(int device);

// Device scratch of a context, leased per call: a context may be shared by threads that each own their plans / extenders
// (the reference's goroutine-per-evaluator model, examples/dbfv/psi/psi.go:221; ring.Context allocates its temporaries per
// call).  Two calls in flight therefore never share a buffer; a buffer returns to the free list when its call has
// enqueued its last kernel, and the next lease's work follows on the same stream.
struct ScratchPool {
    struct Buf { u64 *d; size_t words; };
    std::mutex mu;
    std::vector<Buf> free_list;
    const hipStream_t *owner_stream = nullptr;   // the owning context's stream (read at acquire: is a capture in progress?)
    bool pinned = false;                         // a lease was handed out during a stream capture: its address may be baked into a graph
    int acquire(size_t words, Buf *out) {
        out->d = nullptr;
        out->words = 0;
        if (words == 0) return LR_OK;
        bool capturing = false;
        if (owner_stream && *owner_stream) {
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(*owner_stream, &st) == hipSuccess) capturing = st != hipStreamCaptureStatusNone;
            else (void)hipGetLastError();
        }
        std::vector<Buf> drop;
        {
            std::lock_guard<std::mutex> lock(mu);
            if (capturing) pinned = true;
            size_t best = free_list.size();
            for (size_t i = 0; i < free_list.size(); ++i)
                if (free_list[i].words >= words && (best == free_list.size() || free_list[i].words < free_list[best].words)) best = i;
            if (best != free_list.size()) {
                *out = free_list[best];
                free_list.erase(free_list.begin() + (long)best);
                return LR_OK;
            }
            // nothing fits: the pool grows by one buffer.  Once a lease has been handed out during a stream capture the idle buffers stay
            // for the context's life: their addresses may be baked into a HIP graph (the pair-flag memset node and the temporaries of a
            // captured pipeline call).  Otherwise the idle buffers -- every one of them too small for this request -- go now, so that a
            // context that sees growing sizes does not keep every earlier buffer (hipFree waits for the work that still uses them).
            if (!pinned) drop.swap(free_list);
        }
        for (Buf &b : drop) (void)hipFree(b.d);
        LR_HIP(hipMalloc((void **)&out->d, words * sizeof(u64)));
        out->words = words;
        return LR_OK;
    }
    void release(Buf b) {
        if (!b.d) return;
        std::lock_guard<std::mutex> lock(mu);
        free_list.push_back(b);
    }
    ~ScratchPool() {
        for (Buf &b : free_list) (void)hipFree(b.d);
    }
};

struct ScratchLease {
    ScratchPool *pool = nullptr;
    ScratchPool::Buf buf{nullptr, 0};
    int take(ScratchPool *p, size_t words) {
        pool = p;
        return p->acquire(words, &buf);
    }
    u64 *d() const { return buf.d; }
    ~ScratchLease() {
        if (pool) pool->release(buf);
    }
};

}  // namespace lr_host
using namespace lr_host;

// ------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------
struct lr_context {
    int device = 0;
    hipStream_t stream = nullptr;  // the device's shared stream unless lr_context_set_stream replaced it
    HostContext h;
    LimbParams *d_lp = nullptr;
    Twiddle *d_fwd = nullptr;
    Twiddle *d_inv = nullptr;
    Twiddle *d_fwd_fin = nullptr;  // lane-transposed tables of the last four stages, [L][15][N/16]
    Twiddle *d_inv_fin = nullptr;
    u64 *d_rescale = nullptr;   // [L][L]
    Options opt;                // environment switches, read once at creation
    ScratchPool scratch;        // rescale / staging temporaries, leased per call (thread-safe)
    char last_ntt_kernel[32] = "";   // name of the kernel the last NTT launch of this context dispatched (diagnostics, bench.py); a context may be
    mutable std::mutex diag_mu;      // shared by threads, so the name is written and read under diag_mu (found by ThreadSanitizer, round 4)
    u32 *d_stamps = nullptr;    // timeline builds (Options::timeline): [workgroup][wave 16][stamp 16] of the last stamped launch
    size_t stamp_words = 0, stamp_used = 0;
    // DivRoundByLastModulusNTT: per level, -(pHalfNegQi[i] * NTT_i(1 + X + ... + X^(N-1))) * rescaleParams[i] for i < level,
    // [level][N], built on first use (rescale_round_table)
    struct RoundTable { u64 *plus; EpiLimb *epi; u64 *zeros; };   // zeros: [level][N], the `plus` operand of the flooring division's epilogue
    std::map<int, RoundTable> rescale_round;    // per level; epi = rescaleParams as (c, c / q) doubles for the NTT epilogue
    std::mutex rescale_mu;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int ntt_mode = 0;           // lazy-correction cadence allowed by the largest modulus (lr_ntt.hip)
    bool use_asm = true;        // hand-scheduled assembly NTT where it applies (LR_NO_ASM=1 disables)
    int asm_fwd = -1, asm_inv = -1;   // variant of the assembly kernels all moduli allow, -1 = none
    // variant 3 = dual kernels: FP64 body for the limbs below 2^46, integer body (mode 2) for the others
    Twiddle *d_fwd_fp = nullptr, *d_inv_fp = nullptr, *d_fwd_fin_fp = nullptr, *d_inv_fin_fp = nullptr;
    FpLimb *d_fp_lp = nullptr;
    const void *lane_of = nullptr;   // the batcher whose lane runs on this context (its stream is the lane's): not destroyed while that lives
};

struct lr_poly {
    lr_context *ctx = nullptr;     // creator; only dereferenced while the caller holds it (operations: its stream at call time), never on free
    int device = 0;
    u64 N = 0;
    u64 *d = nullptr;
    bool owned = false;
    int limbs = 0;        // logical limb count (rescale shrinks it)
    int alloc_limbs = 0;  // limbs the storage holds per poly, fixed at allocation
    int batch = 0;
    long long stride_words = 0;   // u64 elements between consecutive batch polys (alloc_limbs * N unless wrapped with a stride)
    long long stride() const { return stride_words; }
};

namespace lr_host {

struct DevModup {
    HostModup h;
    u64 *Q = nullptr, *mredQ = nullptr, *qib = nullptr, *P = nullptr, *mredP = nullptr, *bredP_hi = nullptr,
        *qispj = nullptr, *qpj_inv = nullptr;
    ulonglong2 *qispj_shoup = nullptr;
    double *Qrcp = nullptr;
    u64 *invtop0 = nullptr, *invtop1 = nullptr;     // set_inverse_top
    int lazy_terms = 0, exact_terms = 0, word_barrett = 0, wide_ok = 0, fast_div_ok = 1;
    // The table's input limbs are the limbs limb0 .. of ring `hc`: the multipliers that take the lazy halves of an inverse sub-block
    // transform straight to the y_i (ExtTables::invtop0 / invtop1)
    int set_inverse_top(const HostContext &hc, int limb0) {
        const size_t nQ = h.Q.size();
        if (limb0 < 0 || (size_t)limb0 + nQ > hc.q.size() || hc.N < 2) return LR_OK;
        std::vector<u64> k0(nQ), k1(nQ);
        for (size_t i = 0; i < nQ; ++i) {
            const u64 q = h.Q[i], qinv = h.mredQ[i];
            if (hc.q[limb0 + i] != q) return LR_OK;      // (not this ring's limbs: the table stays without the multipliers)
            const u64 n_inv = inv_mform(hc.n_inv[limb0 + i], q, qinv);
            const u64 psi_inv1 = inv_mform(hc.ntt_psi_inv[(size_t)(limb0 + i) * hc.N + 1], q, qinv);
            const u64 w1n = (u64)(((u128)psi_inv1 * n_inv) % q);
            k0[i] = (u64)(((u128)h.qib_mont[i] * n_inv) % q);
            k1[i] = (u64)(((u128)h.qib_mont[i] * w1n) % q);
        }
        LR_TRY(to_device(&invtop0, k0.data(), k0.size()));
        LR_TRY(to_device(&invtop1, k1.data(), k1.size()));
        return LR_OK;
    }
    int init(const std::vector<u64> &Qv, const std::vector<u64> &Pv, bool ext_narrow, bool ieee_div = false) {
        h = build_modup(Qv, Pv);
        std::vector<u64> bh(h.P.size());
        for (size_t j = 0; j < bh.size(); ++j) bh[j] = h.bredP[j].hi;
        LR_TRY(to_device(&Q, h.Q.data(), h.Q.size()));
        LR_TRY(to_device(&mredQ, h.mredQ.data(), h.mredQ.size()));
        LR_TRY(to_device(&qib, h.qib_mont.data(), h.qib_mont.size()));
        LR_TRY(to_device(&P, h.P.data(), h.P.size()));
        LR_TRY(to_device(&mredP, h.mredP.data(), h.mredP.size()));
        LR_TRY(to_device(&bredP_hi, bh.data(), bh.size()));
        LR_TRY(to_device(&qispj, h.qispj_mont.data(), h.qispj_mont.size()));
        LR_TRY(to_device(&qpj_inv, h.qpj_inv.data(), h.qpj_inv.size()));
        const size_t nQ = h.Q.size(), nP = h.P.size();
        std::vector<ulonglong2> sh(nQ * nP);
        u64 pmax = 0;
        for (size_t j = 0; j < nP; ++j) pmax = h.P[j] > pmax ? h.P[j] : pmax;
        for (size_t i = 0; i < nQ; ++i)
            for (size_t j = 0; j < nP; ++j) {
                const u64 plain = inv_mform(h.qispj_mont[i * nP + j], h.P[j], h.mredP[j]);
                sh[i * nP + j] = make_ulonglong2(plain, shoup_companion(plain, h.P[j]));
            }
        LR_TRY(to_device(&qispj_shoup, sh.data(), sh.size()));
        {
            // correctly rounded reciprocals of the moduli as the reference's float64(q_i) sees them (div_by_const, lr_bext.hip)
            std::vector<double> rc(nQ);
            for (size_t i = 0; i < nQ; ++i) {
                const double b = (double)h.Q[i];
                rc[i] = 1.0 / b;
                // Markstein's final-rounding step (div_by_const) is proven for divisors whose significand is not all ones; the
                // quotients stay far from the overflow and subnormal ranges for every 2 <= b < 2^64.  A modulus that fails the check
                // (none of the reference's parameter sets does) keeps the plain IEEE division of the reference-shaped kernel.
                u64 bits;
                std::memcpy(&bits, &b, sizeof bits);
                const u64 frac = bits & (((u64)1 << 52) - 1);
                if (h.Q[i] < 2 || frac == (((u64)1 << 52) - 1)) fast_div_ok = 0;
            }
            if (ieee_div) fast_div_ok = 0;                             // Options::ext_ieee_div: the reference-shaped kernel
            LR_TRY(to_device(&Qrcp, rc.data(), rc.size()));
        }
        const u128 room = ((u128)1 << 64) - pmax;
        lazy_terms = (int)std::min<u128>(room / ((u128)5 * pmax), 1 << 20);   // 4p per term + p per unit of the correction v <= terms
        exact_terms = (int)std::min<u128>(room / ((u128)2 * pmax), 1 << 20);
        word_barrett = 1;
        {
            u64 qmax = 0;
            for (size_t i = 0; i < nQ; ++i) qmax = h.Q[i] > qmax ? h.Q[i] : qmax;
            // a group's sum of y_i * c_ij (y_i < q_i) over n terms is below n * qmax * p_j, which must stay below p_j * 2^64
            wide_ok = ext_narrow ? 0 : (int)std::min<u128>((((u128)1 << 64) - 1) / qmax, 1 << 20);
        }
        for (size_t j = 0; j < nP; ++j) word_barrett &= (h.P[j] >> 32) != 0 && h.P[j] != ((u64)1 << 32) ? 1 : 0;
        return LR_OK;
    }
    ExtTables tables() const {
        ExtTables t;
        t.nQ = (int)h.Q.size();
        t.nP = (int)h.P.size();
        t.Q = Q; t.mredQ = mredQ; t.qib_mont = qib; t.P = P; t.mredP = mredP; t.bredP_hi = bredP_hi;
        t.qispj_mont = qispj; t.qpj_inv = qpj_inv;
        t.qispj_shoup = qispj_shoup;
        t.Qrcp = Qrcp;
        t.fast_div_ok = fast_div_ok;
        t.lazy_terms = lazy_terms;
        t.exact_terms = exact_terms;
        t.word_barrett = word_barrett;
        t.wide_ok = wide_ok;
        t.invtop0 = invtop0;
        t.invtop1 = invtop1;
        return t;
    }
    ~DevModup() {
        for (u64 *p : {Q, mredQ, qib, P, mredP, bredP_hi, qispj, qpj_inv, invtop0, invtop1})
            if (p) (void)hipFree(p);
        if (qispj_shoup) (void)hipFree(qispj_shoup);
        if (Qrcp) (void)hipFree(Qrcp);
    }
};

// a grow-on-demand device buffer
struct Pool {
    u64 *d = nullptr;
    size_t words = 0;
    int ensure(lr_context *c, size_t need) {
        if (need <= words) return LR_OK;
        LR_HIP(hipStreamSynchronize(c->stream));
        if (d) LR_HIP(hipFree(d));
        d = nullptr;
        words = 0;
        LR_HIP(hipMalloc((void **)&d, need * sizeof(u64)));
        words = need;
        return LR_OK;
    }
    ~Pool() {
        if (d) (void)hipFree(d);
    }
};

}  // namespace lr_host

struct lr_bext {
    int device = 0;
    lr_context *cQ = nullptr, *cP = nullptr;
    DevModup qp, pq;
    std::vector<u64> moddown_pq, moddown_qp;  // host copies (Montgomery form)
    u64 *d_moddown_pq = nullptr, *d_moddown_qp = nullptr;
    // the ModDown P->Q constants once more for the epilogue of the FP64 forward kernels: plain value c = MRed(moddown_pq[i], 1)
    // and RN(c / q_i) as doubles (zero for the limbs of 2^46 and more, which stay on the separate subtract-multiply)
    EpiLimb *d_moddown_pq_epi = nullptr;
    Pool poolQ, poolP;
    Pool zerosQ;              // one poly of zeros over Q: the `plus` operand of the forward kernels' epilogue in the NTT-domain ModDown
    ~lr_bext() {
        if (d_moddown_pq_epi) (void)hipFree(d_moddown_pq_epi);
        if (d_moddown_pq) (void)hipFree(d_moddown_pq);
        if (d_moddown_qp) (void)hipFree(d_moddown_qp);
    }
};

struct lr_decomposer {
    int device = 0;
    lr_context *cQ = nullptr, *cP = nullptr;
    int nQ = 0, nP = 0, alpha = 0, beta = 0;
    std::vector<int> xalpha;
    std::vector<std::vector<std::unique_ptr<DevModup>>> modup;  // [beta][xalpha-1]
};

struct lr_simple_scaler {
    int device = 0;
    lr_context *ctx = nullptr;
    HostSimpleScaler h;
    u64 *d_wi = nullptr;
    double *d_ti = nullptr;
    ~lr_simple_scaler() {
        if (d_wi) (void)hipFree(d_wi);
        if (d_ti) (void)hipFree(d_ti);
    }
};

struct lr_ckks_plan {
    int device = 0;
    lr_context *cQ = nullptr, *cP = nullptr;
    lr_bext *bext = nullptr;
    lr_decomposer *dec = nullptr;
    int max_batch = 0;
    Options opt;           // environment switches, read once at plan creation
    Pool c2QiQ, c2QiP, poolPP, c2, c0, c1, c2x, q1, q2, permQ, permP;
    Pool encQ, encP;       // pk-encryption temporaries over Q||P (lr_ckks_encrypt_pk)
    Pool bfvP;             // bfv relinearize: keyswitchpool[2], [3] (two polys over Q)
    Pool zerosQ;           // one poly of zeros over Q: the `plus` operand of the NTT epilogue where a caller has none
    Pool stageQ, stageP;   // N = 2^16: the extensions land here and the transforms go out of place (fused top stage, see ks_decompose)
    // small batches: independent launches of one pipeline side by side (PlanFork); stream and events are created at the first fork
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fork_failed = false;
    unsigned long long forks = 0, grouped_ext = 0;     // diagnostics (lr_ckks_plan_stats)
    const void *lane_of = nullptr;   // the batcher this plan is a lane of (lanes never fork: their batcher keeps the device busy)
};

// what bfv.NewEvaluator builds for Mul (bfv/evaluator.go:89-112): lr_abi_bfv.cpp
struct lr_bfv_plan {
    int device = 0;
    lr_context *cQ = nullptr, *cM = nullptr;
    lr_bext *bext = nullptr;
    u64 t = 0;
    LimbScalars phalf_q, phalf_m;     // pHalf = (prod QMul) >> 1 reduced modulo each prime
    LimbScalars t_mont;               // MForm(t mod q_i), bfv/evaluator.go:462
    u64 *d_phalf_q = nullptr, *d_phalf_m = nullptr, *d_t_mont = nullptr;   // the same as device arrays (extension epilogues)
    int max_batch = 0;
    bool no_ext_epilogue = false;     // Options::bfv_no_ext_epilogue: separate subtract-multiply / scalar passes after the extensions
    bool no_gather = false;           // Options::bfv_no_gather: every operand / product in launches of its own at every batch size
    long long gather_below = 1536;    // Options::bfv_gather_below: workgroups of the four operands' joint transform up to which they are gathered (PN14QP438:
                                      // gathered 346 / 565 / 1015 / 1912 us per batch of 16 / 32 / 64 / 128, per operand 490 / 618 / 1081 / 1805)
    Pool liftQ, liftM;                // the four operand polys over Q and over QMul, slots a0, a1, b0, b1 of [batch][limbs][N] each
    Pool prodQ, prodM;                // the three products, slots c0, c1, c2
    Pool stageIn, stageOut;           // small batches: the operands gathered into one batch of 4 B, the results before they are scattered
    const void *lane_of = nullptr;    // the batcher this plan is a lane of (not destroyed while that lives)
    ~lr_bfv_plan() {
        for (u64 *p : {d_phalf_q, d_phalf_m, d_t_mont})
            if (p) (void)hipFree(p);
    }
};


// The queue behind the batchers (lr_abi_batcher.cpp, lr_abi_bfv_batcher.cpp): concurrent one-ciphertext calls of many host threads -- the
// reference's one evaluator per goroutine (examples/dbfv/psi/psi.go:215-233) -- merged into batched launches.  A call queues its request;
// whichever caller finds a free lane takes every queued request that may share a launch with the head of the queue (Request::same_batch),
// in arrival order, up to max_batch polys, runs them as ONE pipeline call (`run`, which ends with a synchronisation of the lane's
// stream), and wakes the others.  No thread of its own, no timer: while a lane runs, arrivals queue up and form the next batch.
//   Request: int polys; bool done; int status; std::string error; bool same_batch(const Request &) const
//   Lane   : bool busy; hipStream_t stream; std::vector<Request *> take (reserved at creation); int device() const
template <class Request, class Lane>
struct BatchQueue {
    std::vector<Lane> lanes;
    int max_batch = 0;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Request *> queue;
    unsigned long long batches = 0, products = 0;
    int largest = 0;

    template <class Run>
    int submit(Request &req, Run run) {
        std::unique_lock<std::mutex> lk(m);
        queue.push_back(&req);
        for (;;) {
            if (req.done) break;
            int free_lane = -1;
            for (size_t i = 0; i < lanes.size() && free_lane < 0; ++i)
                if (!lanes[i].busy) free_lane = (int)i;
            if (free_lane < 0 || queue.empty()) {
                cv.wait(lk);
                continue;
            }
            // lead: everything queued that may share the head's launch, in arrival order, up to max_batch polys
            Lane &lane = lanes[free_lane];
            std::vector<Request *> &take = lane.take;
            take.clear();
            int total = 0;
            const Request *head = queue.front();
            for (auto it = queue.begin(); it != queue.end();) {
                Request *r = *it;
                if (r->same_batch(*head) && total + r->polys <= max_batch) {
                    take.push_back(r);      // (capacity reserved at creation: a request taken off the queue is always completed, nothing throws in between)
                    total += r->polys;
                    it = queue.erase(it);
                } else {
                    ++it;
                }
            }
            lane.busy = true;
            lk.unlock();
            int rc = guarded([&]() -> int { return run(lane, take); });
            std::string msg;
            if (rc != LR_OK) {
                try { msg = g_error; } catch (...) {}
                // a batch that failed half-way may have kernels queued that read the callers' operands through the lane's pointer table, and
                // the copy out of the pinned table may be pending: nothing of it may outlive this point -- the callers are about to be woken,
                // and the next batch on this lane rewrites the table and the plan's pools (only the success path ends in a synchronisation)
                (void)hipSetDevice(lane.device());
                (void)hipStreamSynchronize(lane.stream);
                (void)hipGetLastError();
            }
            lk.lock();
            lane.busy = false;
            batches += 1;
            products += (unsigned long long)total;
            largest = std::max(largest, total);
            for (auto *r : take) {
                r->status = rc;
                if (rc != LR_OK) {
                    try { r->error = msg; } catch (...) {}
                }
                r->done = true;
            }
            cv.notify_all();
        }
        lk.unlock();
        if (req.status != LR_OK) return fail(req.status, req.error);
        return LR_OK;
    }
};

namespace lr_host {

struct Rows {  // a strided view of rows inside a batch buffer
    u64 *base;
    long long stride;  // between batch polys
    int limb0, step;
};

// hole/group: digit groups of NttLaunch (the polys of group g skip the items [g*hole, (g+1)*hole))
// epilogue of the forward kernels (NttLaunch::epi_*); every limb of the launch must take it (ntt_epilogue_limb)
struct NttEpilogue {
    const u64 *x;
    long long x_stride;
    const u64 *plus;
    long long plus_stride;
    const EpiLimb *consts;
};

// extensions recorded instead of launched (ks_decompose: the digits of one key switch go out as one grouped launch)
struct ExtPending {
    ExtLaunch L;
    int n_in;
};


// switchKeysInPlace, ckks/evaluator.go:1475-1558, on raw buffers: cx/p0/p1 have `q_stride` between batch polys
// `fin` (optional): the ModDown results go to fin->out0/out1 with fin->plus0/plus1 added (CRed), i.e. the two
// Context.Add calls that follow the key switch in MulRelin (:1103-1104) ride on the last ModDown pass
struct KeySwitchEpilogue {
    u64 *out0, *out1;
    long long out_stride;
    const u64 *plus0, *plus1;
    long long plus_stride;
};

// Fork: launches of the calling thread that go to a plan's auxiliary stream instead of the context's (PlanFork, lr_abi_ckks.cpp): two independent
// transforms of a small batch run side by side instead of one after the other.  Only forward transforms are forked (they lease no scratch).
extern thread_local hipStream_t g_fork_stream;   // lr_abi_ring.cpp
inline hipStream_t stream_of(const lr_context *c) { return g_fork_stream ? g_fork_stream : c->stream; }

// lr_abi_core.cpp
int options_from_public(const lr_options *pub, Options *out);
void options_to_public(const Options &o, lr_options *p);
// lr_abi_ring.cpp: transforms and the coefficient-wise family on raw rows
bool ntt_epilogue_ok(const lr_context *c);
bool ntt_epilogue_limb(const lr_context *c, int l);
EpiLimb make_epi_limb(const lr_context *c, int l, u64 cc);
bool ntt_split15(const lr_context *c, long long workgroups);
int run_ntt(lr_context *c, bool inverse, Rows in, Rows out, int mod0, int mod_step, int count, int batch, int hole = 0, int group = 0, const NttEpilogue *epi = nullptr, bool pretop = false, bool lazy = false);
int check_pair(const lr_context *c, int level, const lr_poly *in, const lr_poly *out);
Rows rows_of(const lr_poly *p, int limb0 = 0, int step = 1, bool broadcast_ok = false, int target_batch = 0);
int run_ewise(lr_context *c, int op, int limbs, int batch, const u64 *a, long long a_stride, const u64 *b, long long b_stride, u64 *out, long long out_stride, const LimbScalars *sc, int lp_offset = 0);
int rescale_ntt_domain(lr_context *c, lr_poly *p0, bool round);
int rescale_round_table(lr_context *c, int level, const u64 **out, const EpiLimb **epi_out, const u64 **zeros_out = nullptr);
int check_rescale(lr_context *c, lr_poly *p0);
// lr_abi_bext.cpp: basis extension and decomposition on raw rows
ExtSegment segment(u64 *out, long long stride, int limb0, int col0, int count);
int flush_ext(lr_context *c, std::vector<ExtPending> &pending, int batch, unsigned long long *grouped_launches = nullptr);
int run_ext(lr_context *c, const DevModup &m, int n_in, Rows in, int batch, ExtSegment s0, ExtSegment s1, const ExtSegment *s2 = nullptr, std::vector<ExtPending> *collect = nullptr, bool inv_top = false);
int run_submul(lr_context *c, int limbs, int batch, const u64 *a, long long a_stride, const u64 *b, long long b_stride, long long b_row_stride, u64 *out, long long out_stride, const u64 *d_consts, bool reduce_b, const LimbScalars *addend, const u64 *plus = nullptr, long long plus_stride = 0, const LimbScalars *post = nullptr, int limb0 = 0);
int same_degree(const lr_context *a, const lr_context *b);
int same_stream(const lr_context *a, const lr_context *b);
bool digit_is_extended(const lr_decomposer *d, int level, int crt);
int moddown_pq_core(lr_bext *b, int level, const u64 *p1Q, long long p1Q_stride, Rows pP, int batch, lr_poly *p2, bool ntt, const u64 *x2 = nullptr,
                    long long x2_stride = 0);
bool moddown_epilogue_available(const lr_bext *b);
int decompose_core(lr_decomposer *d, int level, int crt, Rows in, int batch, u64 *outQ, long long outQ_stride, u64 *outP, long long outP_stride, bool split, bool top = false, bool skip_own = false, std::vector<ExtPending> *collect = nullptr, bool inv_top = false);
// lr_abi_ckks.cpp: the key switch and the pipelines over it
std::atomic<int> &standalone_plans(int device);
int run_permute_ntt(lr_context *c, int limbs, int batch, const u64 *in, long long in_stride, u64 *out, long long out_stride, u64 gen, const u64 *const *in_table = nullptr);
int ks_decompose(lr_ckks_plan *pl, int level, int batch, const u64 *cx, long long cx_stride, bool copy_own, bool coeff_input = false);
int ks_accumulate(lr_ckks_plan *pl, int level, int batch, const u64 *digQ, const u64 *digP, const u64 *own, long long own_stride, const lr_poly *evk, u64 *p0, long long p0_stride, u64 *p1, long long p1_stride, const KeySwitchEpilogue *fin, bool coeff_out = false, u64 perm_gen = 0);
int switch_keys_core(lr_ckks_plan *pl, int level, int batch, const u64 *cx, long long cx_stride, const lr_poly *evk, u64 *p0, long long p0_stride, u64 *p1, long long p1_stride, const KeySwitchEpilogue *fin = nullptr);
int check_ct(const lr_ckks_plan *pl, int level, const lr_poly *p, int batch);
int mulrelin_core(lr_ckks_plan *pl, int level, int batch, TensorLaunch T, const lr_poly *evk, u64 *o0, u64 *o1, long long o_stride);

}  // namespace lr_host
using namespace lr_host;
