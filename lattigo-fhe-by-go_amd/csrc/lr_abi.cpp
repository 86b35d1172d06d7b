// lr_abi.cpp -- the C ABI of include/lattigo_ring.h: handles, shape checks, and the composition
// of the HIP kernels into the reference's ring.Context / FastBasisExtender / Decomposer methods
// and the ckks.Evaluator call sequences.  No CPU fallback exists: every arithmetic entry point
// launches gfx950 kernels and fails with LR_ERR_HIP when no device is available.
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

namespace {

thread_local std::string g_error = "";

int fail(int code, const std::string &msg) {
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
hipError_t create_stream(hipStream_t *s, int cls = 0) {
    if (cls == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) {
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    }
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, cls == 1 ? greatest : least);
}

hipStream_t shared_stream(int device) {
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    std::lock_guard<std::mutex> lock(mu);
    auto it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess || create_stream(&s) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    streams[device] = s;
    return s;
}

// Device scratch of a context, leased per call: a context may be shared by threads that each own their plans / extenders
// (the reference's goroutine-per-evaluator model, examples/dbfv/psi/psi.go:221; ring.Context allocates its temporaries per
// call).  Two calls in flight therefore never share a buffer; a buffer returns to the free list when its call has
// enqueued its last kernel, and the next lease's work follows on the same stream.
struct ScratchPool {
    struct Buf { u64 *d; size_t words; };
    std::mutex mu;
    std::vector<Buf> free_list;
    int acquire(size_t words, Buf *out) {
        out->d = nullptr;
        out->words = 0;
        if (words == 0) return LR_OK;
        {
            std::lock_guard<std::mutex> lock(mu);
            size_t best = free_list.size();
            for (size_t i = 0; i < free_list.size(); ++i)
                if (free_list[i].words >= words && (best == free_list.size() || free_list[i].words < free_list[best].words)) best = i;
            if (best != free_list.size()) {
                *out = free_list[best];
                free_list.erase(free_list.begin() + (long)best);
                return LR_OK;
            }
            // nothing fits: the pool grows by one buffer.  Idle buffers are NOT freed while the context lives: their addresses may be
            // baked into a captured HIP graph (the pair-flag memset node and the temporaries of a captured pipeline call), and work
            // enqueued on a stream the caller installed later may still use them; everything goes in the destructor.
        }
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

}  // namespace

// The test-only override: the ONE place of the library that reads LR_* environment variables.  A flag
// variable that is set switches its alternative ON (it never switches a caller's choice off); a value variable replaces the field.
void lr::Options::apply_env() {
    auto flag = [](const char *name, bool &field) {
        if (std::getenv(name) != nullptr) field = true;
    };
    auto num = [](const char *name, int &field) {
        if (const char *v = std::getenv(name)) field = std::atoi(v);
    };
    flag("LR_NO_ASM", no_asm);
    flag("LR_NO_FP", no_fp);
    flag("LR_NO_EPILOGUE", no_epilogue);
    flag("LR_NO_INT_EPILOGUE", no_int_epilogue);
    flag("LR_RESCALE_UNFUSED", rescale_unfused);
    flag("LR_NO_STAGING", no_staging);
    flag("LR_EXT_NARROW", ext_narrow);
    flag("LR_ASM_14_1024", asm14_1024);
    flag("LR_NO_EXTTOP", no_exttop);
    flag("LR_NO_EXT_GROUP", no_ext_group);
    flag("LR_NO_FORK", no_fork);
    if (const char *sp = std::getenv("LR_NTT_SPLIT15")) split15 = std::atoi(sp) != 0 ? 1 : 0;
    flag("LR_RESCALE_UNPAIRED", rescale_unpaired);
    flag("LR_NO_PAIR", no_pair);
    flag("LR_NO_EXT_CHUNKS", no_ext_chunks);
    flag("LR_NO_INVTOP", no_invtop);
    flag("LR_ASM_14_NO_WIDE_SMALL", no_wide14_small);
    flag("LR_NO_INVFUSE", no_invfuse);
    flag("LR_KEYMAC_NARROW", keymac_narrow);
    flag("LR_NTT_TIMELINE", timeline);
    num("LR_NTT_MODE", ntt_mode);
    num("LR_NTT_STAGGER", stagger);
    num("LR_NTT_PERSIST", persist);
    num("LR_ASM_VARIANT", asm_variant);
    flag("LR_EXT_IEEE_DIV", ext_ieee_div);
    flag("LR_NTT_NO_GRID_PADDING", no_grid_padding);
    flag("LR_BFV_NO_EXT_EPILOGUE", bfv_no_ext_epilogue);
    flag("LR_BFV_NO_GATHER", bfv_no_gather);
    if (const char *gb = std::getenv("LR_BFV_GATHER_BELOW")) bfv_gather_below = std::atoll(gb);
    num("LR_NTT_SPLIT15_BELOW", split15_max_workgroups);
    num("LR_FORK_BELOW", fork_below_workgroups);
}

namespace {

// public struct -> internal image.  The caller's struct may be shorter than this library's (an older header): only the first
// struct_size bytes are read, the rest keeps the defaults.  A threshold of 0 means "the built-in default".
int options_from_public(const lr_options *pub, Options *out) {
    Options o;
    if (pub) {
        if (pub->struct_size < 2 * sizeof(uint32_t)) return fail(LR_ERR_ARG, "lr_options: struct_size is not set (use lr_options_init)");
        if (pub->version != LR_OPTIONS_VERSION) return fail(LR_ERR_ARG, "lr_options: unknown version");
        lr_options p;
        (void)lr_options_init(&p);
        std::memcpy(&p, pub, std::min<size_t>(pub->struct_size, sizeof p));
        o.no_asm = p.no_asm != 0;
        o.no_fp = p.no_fp != 0;
        o.ntt_mode = p.ntt_mode;
        o.asm_variant = p.asm_variant;
        o.asm14_1024 = p.asm14_1024 != 0;
        o.no_wide14_small = p.no_wide14_small != 0;
        if (p.wide14_max_items > 0) o.wide14_max_items = p.wide14_max_items;
        o.split15 = p.ntt_split15 < 0 ? -1 : (p.ntt_split15 != 0 ? 1 : 0);
        if (p.split15_max_workgroups > 0) o.split15_max_workgroups = p.split15_max_workgroups;
        o.no_invfuse = p.no_invfuse != 0;
        o.no_grid_padding = p.no_grid_padding != 0;
        o.stagger = p.ntt_stagger;
        o.persist = p.ntt_persist;
        o.timeline = p.ntt_timeline != 0;
        o.no_epilogue = p.no_epilogue != 0;
        o.no_int_epilogue = p.no_int_epilogue != 0;
        o.rescale_unfused = p.rescale_unfused != 0;
        o.rescale_unpaired = p.rescale_unpaired != 0;
        if (p.pair_max_workgroups > 0) o.pair_max_workgroups = p.pair_max_workgroups;
        o.ext_narrow = p.ext_narrow != 0;
        o.ext_ieee_div = p.ext_ieee_div != 0;
        o.no_ext_chunks = p.no_ext_chunks != 0;
        o.no_staging = p.no_staging != 0;
        o.no_exttop = p.no_exttop != 0;
        o.no_invtop = p.no_invtop != 0;
        o.no_ext_group = p.no_ext_group != 0;
        o.keymac_narrow = p.keymac_narrow != 0;
        o.no_pair = p.no_pair != 0;
        o.no_fork = p.no_fork != 0;
        if (p.fork_below_workgroups > 0) o.fork_below_workgroups = p.fork_below_workgroups;
        o.bfv_no_ext_epilogue = p.bfv_no_ext_epilogue != 0;
        o.bfv_no_gather = p.bfv_no_gather != 0;
        if (p.bfv_gather_below > 0) o.bfv_gather_below = p.bfv_gather_below;
    }
    o.apply_env();
#ifndef LR_BUILD_DIAG
    if (o.timeline || o.persist > 0)
        return fail(LR_ERR_UNSUPPORTED, "ntt_timeline / ntt_persist need the diagnostics build of the library (LR_BUILD_DIAG=1 csrc/build.sh): "
                                        "the clock-stamping and persistent code objects are not part of the default build");
#endif
    *out = o;
    return LR_OK;
}

void options_to_public(const Options &o, lr_options *p) {
    (void)lr_options_init(p);
    p->no_asm = o.no_asm; p->no_fp = o.no_fp; p->ntt_mode = o.ntt_mode; p->asm_variant = o.asm_variant; p->asm14_1024 = o.asm14_1024;
    p->no_wide14_small = o.no_wide14_small; p->wide14_max_items = o.wide14_max_items; p->ntt_split15 = o.split15;
    p->split15_max_workgroups = o.split15_max_workgroups; p->no_invfuse = o.no_invfuse; p->no_grid_padding = o.no_grid_padding;
    p->ntt_stagger = o.stagger; p->ntt_persist = o.persist; p->ntt_timeline = o.timeline; p->no_epilogue = o.no_epilogue;
    p->no_int_epilogue = o.no_int_epilogue; p->rescale_unfused = o.rescale_unfused; p->rescale_unpaired = o.rescale_unpaired;
    p->pair_max_workgroups = o.pair_max_workgroups; p->ext_narrow = o.ext_narrow; p->ext_ieee_div = o.ext_ieee_div;
    p->no_ext_chunks = o.no_ext_chunks; p->no_staging = o.no_staging; p->no_exttop = o.no_exttop; p->no_invtop = o.no_invtop;
    p->no_ext_group = o.no_ext_group; p->keymac_narrow = o.keymac_narrow; p->no_pair = o.no_pair; p->no_fork = o.no_fork;
    p->fork_below_workgroups = o.fork_below_workgroups; p->bfv_no_ext_epilogue = o.bfv_no_ext_epilogue; p->bfv_no_gather = o.bfv_no_gather;
    p->bfv_gather_below = o.bfv_gather_below;
}

}  // namespace

extern "C" int lr_options_init(lr_options *opt) {
    if (!opt) return LR_ERR_ARG;
    std::memset(opt, 0, sizeof *opt);
    opt->struct_size = (uint32_t)sizeof *opt;
    opt->version = LR_OPTIONS_VERSION;
    opt->ntt_mode = opt->asm_variant = opt->ntt_split15 = opt->ntt_stagger = opt->ntt_persist = -1;
    return LR_OK;
}

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
    char last_ntt_kernel[32] = "";   // name of the kernel the last NTT launch of this context dispatched (diagnostics, bench.py)
    u32 *d_stamps = nullptr;    // timeline builds (Options::timeline): [workgroup][wave 16][stamp 16] of the last stamped launch
    size_t stamp_words = 0, stamp_used = 0;
    // DivRoundByLastModulusNTT: per level, -(pHalfNegQi[i] * NTT_i(1 + X + ... + X^(N-1))) * rescaleParams[i] for i < level,
    // [level][N], built on first use (rescale_round_table)
    struct RoundTable { u64 *plus; EpiLimb *epi; };
    std::map<int, RoundTable> rescale_round;    // per level; epi = rescaleParams as (c, c / q) doubles for the NTT epilogue
    std::mutex rescale_mu;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int ntt_mode = 0;           // lazy-correction cadence allowed by the largest modulus (lr_ntt.hip)
    bool use_asm = true;        // hand-scheduled assembly NTT where it applies (LR_NO_ASM=1 disables)
    int asm_fwd = -1, asm_inv = -1;   // variant of the assembly kernels all moduli allow, -1 = none
    // variant 3 = dual kernels: FP64 body for the limbs below 2^46, integer body (mode 2) for the others
    Twiddle *d_fwd_fp = nullptr, *d_inv_fp = nullptr, *d_fwd_fin_fp = nullptr, *d_inv_fin_fp = nullptr;
    FpLimb *d_fp_lp = nullptr;
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

namespace {

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

}  // namespace

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

// ------------------------------------------------------------------------------------------
// misc
// ------------------------------------------------------------------------------------------
extern "C" const char *lr_last_error_string(void) { return g_error.c_str(); }

#ifdef LR_BUILD_DIAG
extern "C" const char *lr_build_info(void) { return "lattigo_ring 0.2 gfx950 hip diag"; }   // + the clock-stamping and persistent code objects
#else
extern "C" const char *lr_build_info(void) { return "lattigo_ring 0.2 gfx950 hip"; }
#endif

extern "C" int lr_device_count(int *count) {
    return guarded([&]() -> int {
    if (!count) return fail(LR_ERR_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(LR_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return LR_OK;
    });
}

// ------------------------------------------------------------------------------------------
// Context
// ------------------------------------------------------------------------------------------
extern "C" int lr_context_create(uint64_t N, const uint64_t *moduli, int n_moduli, int device, lr_context **out) {
    return lr_context_create_ex(N, moduli, n_moduli, device, nullptr, out);
}

extern "C" int lr_context_get_options(const lr_context *c, lr_options *out) {
    return guarded([&]() -> int {
    if (!c || !out) return fail(LR_ERR_ARG, "null argument");
    options_to_public(c->opt, out);
    return LR_OK;
    });
}

extern "C" int lr_context_create_ex(uint64_t N, const uint64_t *moduli, int n_moduli, int device, const lr_options *options, lr_context **out) {
    return guarded([&]() -> int {
    if (!out) return fail(LR_ERR_ARG, "out is null");
    *out = nullptr;
    Options parsed;
    LR_TRY(options_from_public(options, &parsed));
    if (!moduli || n_moduli <= 0 || n_moduli > kMaxLimbs) return fail(LR_ERR_ARG, "bad modulus list (1..64 moduli)");
    std::unique_ptr<lr_context> c(new (std::nothrow) lr_context());
    if (!c) return fail(LR_ERR_ARG, "out of host memory");
    const int rc = build_context(N, moduli, n_moduli, c->h);
    if (rc == 2) return fail(LR_ERR_INVALID_DEGREE, "invalid ring degree (must be a power of 2)");
    if (rc == 1) return fail(LR_ERR_NOT_NTT_FRIENDLY, "warning : provided modulus does not allow NTT");
    for (u64 q : c->h.q)
        if (q >> 61) return fail(LR_ERR_UNSUPPORTED, "modulus must be below 2^61 (the reference's lazy NTT has the same limit)");
    c->device = device;
    c->opt = parsed;
    {
        u64 qmax = 0, qmin = ~(u64)0;
        for (u64 q : c->h.q) {
            qmax = q > qmax ? q : qmax;
            qmin = q < qmin ? q : qmin;
        }
        if (qmax < (1ull << 57)) c->ntt_mode = 2;
        else if (qmin >= (1ull << 57)) c->ntt_mode = qmax <= (1ull << 60) ? 1 : 0;
        else c->ntt_mode = 3;                         // mixed sizes: generic path
        if (c->opt.ntt_mode >= 0) {
            const int f = c->opt.ntt_mode;            // testing aid: 0 and 3 are always valid where 1 is
            if ((f == 0 && c->ntt_mode == 1) || f == 3) c->ntt_mode = f;
        }
        if (qmin >= (1ull << 32)) c->ntt_mode |= 256;
        c->use_asm = !c->opt.no_asm;
        if (qmin > (1ull << 33)) {                    // 32-bit Barrett constant of the assembly kernels
            c->asm_fwd = qmax < (1ull << 57) ? 2 : qmax <= (1ull << 60) ? 1 : 0;
            c->asm_inv = qmax <= (1ull << 60) ? 1 : 0;
            if (c->opt.asm_variant >= 0) {   // testing aid: a more conservative variant
                const int f = c->opt.asm_variant;
                if (f == 0 || (f == 1 && c->asm_fwd >= 1)) c->asm_fwd = f;
                if (f == 0) c->asm_inv = 0;
            }
        }
        // the FP64 body takes any modulus below 2^46; the integer body next to it needs the others in (2^33, 2^57)
        if (qmin < kFpLimit && qmax < (1ull << 57) && !c->opt.no_fp && c->opt.asm_variant < 0) {
            bool ok = true;
            for (u64 q : c->h.q) ok = ok && (q < kFpLimit || q > (1ull << 33));
            if (ok) c->asm_fwd = c->asm_inv = 3;
        }
    }
    LR_HIP(hipSetDevice(device));
    c->stream = shared_stream(device);
    if (!c->stream) return fail(LR_ERR_HIP, "could not create the device stream");
    LR_HIP(hipEventCreate(&c->ev0));
    LR_HIP(hipEventCreate(&c->ev1));

    const int L = n_moduli;
    std::vector<LimbParams> lp(L);
    std::vector<Twiddle> fwd((size_t)L * N), inv((size_t)L * N);
    for (int i = 0; i < L; ++i) {
        const u64 q = c->h.q[i], qinv = c->h.mred[i];
        LimbParams &p = lp[i];
        p.q = q;
        p.qinv = qinv;
        p.bred_hi = c->h.bred[i].hi;
        p.bred_lo = c->h.bred[i].lo;
        p.n_inv_mont = c->h.n_inv[i];
        p.n_inv = inv_mform(c->h.n_inv[i], q, qinv);
        p.n_inv_shoup = shoup_companion(p.n_inv, q);
        {
            const u64 qh = (q >> 32) + 1;
            unsigned g = 0;
            while ((qh >> (g + 1)) != 0) ++g;  // bitlen(qh) - 1
            const u64 m = ((u64)1 << (32 + g)) / qh;
            p.red_m = m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)m;
            p.red_g = g;
        }
        for (u64 j = 0; j < N; ++j) {
            const u64 wf = inv_mform(c->h.ntt_psi[(size_t)i * N + j], q, qinv);
            const u64 wi = inv_mform(c->h.ntt_psi_inv[(size_t)i * N + j], q, qinv);
            fwd[(size_t)i * N + j] = make_ulonglong2(wf, shoup_companion(wf, q));
            inv[(size_t)i * N + j] = make_ulonglong2(wi, shoup_companion(wi, q));
        }
        if (N >= 2) {
            // heap index 0 is unused by the transform.  Forward: q - psi[1], the twiddle that turns the X-form butterfly
            // into the Y output of the N = 2^16 top stage (assembly sub-block 1).  Inverse: psi_inv[1] * N^-1, the
            // twiddle of the last inverse stage fused with the scaling.
            const u64 nw1 = q - fwd[(size_t)i * N + 1].x;
            fwd[(size_t)i * N] = make_ulonglong2(nw1, shoup_companion(nw1, q));
            const u64 w1n = (u64)(((u128)inv[(size_t)i * N + 1].x * p.n_inv) % q);
            inv[(size_t)i * N] = make_ulonglong2(w1n, shoup_companion(w1n, q));
        }
    }
    LR_TRY(to_device(&c->d_lp, lp.data(), lp.size()));
    const bool fp_tables = c->asm_fwd == 3 || c->asm_inv == 3;
    // FP64 body: the same table with every (w, floor(w 2^64 / q)) replaced by the doubles (w, RN(w / q)); zero for the other limbs
    auto to_fp = [&](std::vector<Twiddle> &t, size_t per_limb) {
        for (int i = 0; i < L; ++i) {
            const u64 q = c->h.q[i];
            for (size_t j = 0; j < per_limb; ++j) {
                Twiddle &e = t[(size_t)i * per_limb + j];
                if (q < kFpLimit) {
                    const double w = (double)e.x, wq = w / (double)q;
                    std::memcpy(&e.x, &w, 8);
                    std::memcpy(&e.y, &wq, 8);
                } else {
                    e = make_ulonglong2(0, 0);
                }
            }
        }
    };
    if (N >= 4096) {
        const size_t blocks = N >> 4;
        std::vector<Twiddle> ffin((size_t)L * 15 * blocks), ifin((size_t)L * 15 * blocks);
        for (int i = 0; i < L; ++i)
            for (int cc = 0; cc < 4; ++cc)
                for (int j = 0; j < (1 << cc); ++j)
                    for (size_t bk = 0; bk < blocks; ++bk) {
                        const size_t src = (size_t)i * N + (((blocks + bk) << cc) + j);
                        const size_t dst = ((size_t)i * 15 + ((1u << cc) - 1 + j)) * blocks + bk;
                        ffin[dst] = fwd[src];
                        ifin[dst] = inv[src];
                    }
        LR_TRY(to_device(&c->d_fwd_fin, ffin.data(), ffin.size()));
        LR_TRY(to_device(&c->d_inv_fin, ifin.data(), ifin.size()));
        if (fp_tables) {
            to_fp(ffin, 15 * blocks);
            to_fp(ifin, 15 * blocks);
            LR_TRY(to_device(&c->d_fwd_fin_fp, ffin.data(), ffin.size()));
            LR_TRY(to_device(&c->d_inv_fin_fp, ifin.data(), ifin.size()));
        }
    }
    LR_TRY(to_device(&c->d_fwd, fwd.data(), fwd.size()));
    LR_TRY(to_device(&c->d_inv, inv.data(), inv.size()));
    if (fp_tables) {
        std::vector<FpLimb> fl(L);
        for (int i = 0; i < L; ++i) {
            const u64 q = c->h.q[i];
            fl[i] = q < kFpLimit ? FpLimb{(double)q, 1.0 / (double)q, (double)lp[i].n_inv, (double)lp[i].n_inv / (double)q} : FpLimb{0.0, 0.0, 0.0, 0.0};
        }
        LR_TRY(to_device(&c->d_fp_lp, fl.data(), fl.size()));
        to_fp(fwd, N);
        to_fp(inv, N);
        LR_TRY(to_device(&c->d_fwd_fp, fwd.data(), fwd.size()));
        LR_TRY(to_device(&c->d_inv_fp, inv.data(), inv.size()));
    }
    LR_TRY(to_device(&c->d_rescale, c->h.rescale.data(), c->h.rescale.size()));
    *out = c.release();
    return LR_OK;
    });
}

extern "C" int lr_context_ntt_variants(const lr_context *c, int *forward, int *inverse) {
    return guarded([&]() -> int {
    if (!c || !forward || !inverse) return fail(LR_ERR_ARG, "null argument");
    *forward = c->use_asm ? c->asm_fwd : -1;
    *inverse = c->use_asm ? c->asm_inv : -1;
    return LR_OK;
    });
}

extern "C" int lr_context_destroy(lr_context *c) {
    return guarded([&]() -> int {
    if (!c) return LR_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();   // whatever stream the handle last ran on (its own, the shared one, a caller's)
    for (void *p : {(void *)c->d_lp, (void *)c->d_fwd, (void *)c->d_inv, (void *)c->d_fwd_fin, (void *)c->d_inv_fin, (void *)c->d_rescale,
                    (void *)c->d_fwd_fp, (void *)c->d_inv_fp, (void *)c->d_fwd_fin_fp, (void *)c->d_inv_fin_fp, (void *)c->d_fp_lp})
        if (p) (void)hipFree(p);
    if (c->d_stamps) (void)hipFree(c->d_stamps);
    for (auto &kv : c->rescale_round) {
        if (kv.second.plus) (void)hipFree(kv.second.plus);
        if (kv.second.epi) (void)hipFree(kv.second.epi);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
    return LR_OK;
    });
}

extern "C" int lr_context_set_stream(lr_context *c, void *hip_stream) {
    return guarded([&]() -> int {
    if (!c) return fail(LR_ERR_ARG, "null context");
    LR_HIP(hipSetDevice(c->device));
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : shared_stream(c->device);
    if (next != c->stream) {
        // work already enqueued through this context (and the scratch it leased, which later calls reuse) is ordered before
        // whatever follows on the new stream: an event on the old stream that the new one waits for -- no host synchronisation
        hipEvent_t ev = nullptr;
        LR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e1 = hipEventRecord(ev, c->stream);
        hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(next, ev, 0) : e1;
        (void)hipEventDestroy(ev);
        if (e2 != hipSuccess) return fail(LR_ERR_HIP, std::string("set_stream: ") + hipGetErrorString(e2));
        c->stream = next;
    }
    return LR_OK;
    });
}

extern "C" int lr_context_sync(lr_context *c) {
    return guarded([&]() -> int {
    if (!c) return fail(LR_ERR_ARG, "null context");
    LR_HIP(hipSetDevice(c->device));
    LR_HIP(hipStreamSynchronize(c->stream));
    return LR_OK;
    });
}

extern "C" int lr_context_info(const lr_context *c, uint64_t *N, int *n_moduli, int *device) {
    return guarded([&]() -> int {
    if (!c) return fail(LR_ERR_ARG, "null context");
    if (N) *N = c->h.N;
    if (n_moduli) *n_moduli = c->h.L();
    if (device) *device = c->device;
    return LR_OK;
    });
}

extern "C" int lr_context_get_table(const lr_context *c, int which, uint64_t *dst, size_t dst_count) {
    return guarded([&]() -> int {
    if (!c || !dst) return fail(LR_ERR_ARG, "null argument");
    const HostContext &h = c->h;
    const size_t L = (size_t)h.L();
    std::vector<u64> tmp;
    const std::vector<u64> *src = nullptr;
    switch (which) {
    case LR_TAB_MODULUS: src = &h.q; break;
    case LR_TAB_MRED: src = &h.mred; break;
    case LR_TAB_PSI_MONT: src = &h.psi_mont; break;
    case LR_TAB_PSI_INV_MONT: src = &h.psi_inv_mont; break;
    case LR_TAB_NTT_PSI: src = &h.ntt_psi; break;
    case LR_TAB_NTT_PSI_INV: src = &h.ntt_psi_inv; break;
    case LR_TAB_NTT_N_INV: src = &h.n_inv; break;
    case LR_TAB_RESCALE: src = &h.rescale; break;
    case LR_TAB_MASK: src = &h.mask; break;
    case LR_TAB_BRED:
        tmp.resize(2 * L);
        for (size_t i = 0; i < L; ++i) {
            tmp[2 * i] = h.bred[i].hi;
            tmp[2 * i + 1] = h.bred[i].lo;
        }
        src = &tmp;
        break;
    default: return fail(LR_ERR_ARG, "unknown table id");
    }
    if (dst_count != src->size()) return fail(LR_ERR_SHAPE, "table size mismatch");
    std::memcpy(dst, src->data(), src->size() * sizeof(u64));
    return LR_OK;
    });
}

// ------------------------------------------------------------------------------------------
// Poly
// ------------------------------------------------------------------------------------------
extern "C" int lr_poly_alloc(lr_context *c, int limbs, int batch, lr_poly **out) {
    return guarded([&]() -> int {
    if (!c || !out) return fail(LR_ERR_ARG, "null argument");
    *out = nullptr;
    if (limbs <= 0 || limbs > kMaxLimbs || batch <= 0) return fail(LR_ERR_SHAPE, "limbs must be 1..64 and batch >= 1");
    LR_HIP(hipSetDevice(c->device));
    std::unique_ptr<lr_poly> p(new lr_poly());
    p->ctx = c;
    p->device = c->device;
    p->N = c->h.N;
    p->limbs = p->alloc_limbs = limbs;
    p->stride_words = (long long)limbs * (long long)c->h.N;
    p->batch = batch;
    p->owned = true;
    const size_t bytes = (size_t)batch * limbs * c->h.N * sizeof(u64);
    LR_HIP(hipMalloc((void **)&p->d, bytes));
    LR_HIP(hipMemsetAsync(p->d, 0, bytes, c->stream));
    *out = p.release();
    return LR_OK;
    });
}

static int poly_wrap(lr_context *c, void *device_ptr, int limbs, int batch, long long stride_words, lr_poly **out) {
    if (!c || !out || !device_ptr) return fail(LR_ERR_ARG, "null argument");
    *out = nullptr;
    if (limbs <= 0 || limbs > kMaxLimbs || batch <= 0) return fail(LR_ERR_SHAPE, "limbs must be 1..64 and batch >= 1");
    if (((uintptr_t)device_ptr & 15) != 0) return fail(LR_ERR_ARG, "device pointer must be 16-byte aligned");
    if (stride_words < (long long)limbs * (long long)c->h.N || (stride_words & 1) != 0)
        return fail(LR_ERR_SHAPE, "poly stride must be an even number of words and at least limbs * N");
    lr_poly *p = new lr_poly();
    p->ctx = c;
    p->device = c->device;
    p->N = c->h.N;
    p->d = (u64 *)device_ptr;
    p->limbs = p->alloc_limbs = limbs;
    p->stride_words = stride_words;
    p->batch = batch;
    p->owned = false;
    *out = p;
    return LR_OK;
}

extern "C" int lr_poly_wrap(lr_context *c, void *device_ptr, int limbs, int batch, lr_poly **out) {
    return guarded([&]() -> int {
    if (!c) return fail(LR_ERR_ARG, "null argument");
    return poly_wrap(c, device_ptr, limbs, batch, (long long)limbs * (long long)c->h.N, out);
    });
}

extern "C" int lr_poly_wrap_strided(lr_context *c, void *device_ptr, int limbs, int batch, long long poly_stride_words, lr_poly **out) {
    return guarded([&]() -> int {
    return poly_wrap(c, device_ptr, limbs, batch, poly_stride_words, out);
    });
}

extern "C" int lr_poly_free(lr_poly *p) {
    return guarded([&]() -> int {
    if (!p) return LR_OK;
    if (p->owned && p->d) {
        (void)hipSetDevice(p->device);
        (void)hipDeviceSynchronize();   // the handle's work may be on its contexts' caller-supplied stream
        (void)hipFree(p->d);
        (void)hipGetLastError();
    }
    delete p;
    return LR_OK;
    });
}

extern "C" int lr_poly_info(const lr_poly *p, uint64_t *N, int *limbs, int *batch, void **device_ptr) {
    return guarded([&]() -> int {
    if (!p) return fail(LR_ERR_ARG, "null poly");
    if (N) *N = p->N;
    if (limbs) *limbs = p->limbs;
    if (batch) *batch = p->batch;
    if (device_ptr) *device_ptr = p->d;
    return LR_OK;
    });
}

extern "C" int lr_poly_set_limbs(lr_poly *p, int limbs) {
    return guarded([&]() -> int {
    if (!p) return fail(LR_ERR_ARG, "null poly");
    if (limbs < 0 || limbs > p->alloc_limbs) return fail(LR_ERR_SHAPE, "limb count exceeds the allocation");
    p->limbs = limbs;
    return LR_OK;
    });
}

extern "C" int lr_poly_zero(lr_poly *p) {
    return guarded([&]() -> int {
    if (!p) return fail(LR_ERR_ARG, "null poly");
    LR_HIP(hipSetDevice(p->device));
    LR_HIP(hipMemsetAsync(p->d, 0, (size_t)p->batch * p->stride() * sizeof(u64), p->ctx->stream));
    return LR_OK;
    });
}

extern "C" int lr_poly_upload(lr_poly *p, int batch_index, const uint64_t *const *limb_ptrs, int limbs) {
    return guarded([&]() -> int {
    if (!p || !limb_ptrs) return fail(LR_ERR_ARG, "null argument");
    if (batch_index < 0 || batch_index >= p->batch || limbs < 0 || limbs > p->limbs)
        return fail(LR_ERR_SHAPE, "upload: batch index or limb count out of range");
    LR_HIP(hipSetDevice(p->device));
    const size_t row = p->N * sizeof(u64);
    for (int i = 0; i < limbs; ++i) {
        if (!limb_ptrs[i]) return fail(LR_ERR_ARG, "null limb pointer");
        LR_HIP(hipMemcpyAsync(p->d + batch_index * p->stride() + (long long)i * p->N, limb_ptrs[i], row,
                              hipMemcpyHostToDevice, p->ctx->stream));
    }
    LR_HIP(hipStreamSynchronize(p->ctx->stream));
    return LR_OK;
    });
}

// One limb at a time: the form a cgo caller built for the reference's go 1.13 needs -- a Go pointer may be passed to C for the
// duration of a call, but it may not be stored in C memory (an array of limb pointers), and runtime.Pinner is go 1.21.  The copy
// has completed when the call returns.
extern "C" int lr_poly_upload_limb(lr_poly *p, int batch_index, int limb, const uint64_t *src) {
    return guarded([&]() -> int {
    if (!p || !src) return fail(LR_ERR_ARG, "null argument");
    if (batch_index < 0 || batch_index >= p->batch || limb < 0 || limb >= p->limbs)
        return fail(LR_ERR_SHAPE, "upload: batch index or limb out of range");
    LR_HIP(hipSetDevice(p->device));
    LR_HIP(hipMemcpyAsync(p->d + batch_index * p->stride() + (long long)limb * p->N, src, p->N * sizeof(u64), hipMemcpyHostToDevice,
                          p->ctx->stream));
    LR_HIP(hipStreamSynchronize(p->ctx->stream));
    return LR_OK;
    });
}

extern "C" int lr_poly_download_limb(const lr_poly *p, int batch_index, int limb, uint64_t *dst) {
    return guarded([&]() -> int {
    if (!p || !dst) return fail(LR_ERR_ARG, "null argument");
    if (batch_index < 0 || batch_index >= p->batch || limb < 0 || limb >= p->limbs)
        return fail(LR_ERR_SHAPE, "download: batch index or limb out of range");
    LR_HIP(hipSetDevice(p->device));
    LR_HIP(hipMemcpyAsync(dst, p->d + batch_index * p->stride() + (long long)limb * p->N, p->N * sizeof(u64), hipMemcpyDeviceToHost,
                          p->ctx->stream));
    LR_HIP(hipStreamSynchronize(p->ctx->stream));
    return LR_OK;
    });
}

extern "C" int lr_poly_download(const lr_poly *p, int batch_index, uint64_t *const *limb_ptrs, int limbs) {
    return guarded([&]() -> int {
    if (!p || !limb_ptrs) return fail(LR_ERR_ARG, "null argument");
    if (batch_index < 0 || batch_index >= p->batch || limbs < 0 || limbs > p->limbs)
        return fail(LR_ERR_SHAPE, "download: batch index or limb count out of range");
    LR_HIP(hipSetDevice(p->device));
    const size_t row = p->N * sizeof(u64);
    for (int i = 0; i < limbs; ++i) {
        if (!limb_ptrs[i]) return fail(LR_ERR_ARG, "null limb pointer");
        LR_HIP(hipMemcpyAsync(limb_ptrs[i], p->d + batch_index * p->stride() + (long long)i * p->N, row,
                              hipMemcpyDeviceToHost, p->ctx->stream));
    }
    LR_HIP(hipStreamSynchronize(p->ctx->stream));
    return LR_OK;
    });
}

static int dense_copy(const lr_poly *p, u64 *host, const u64 *host_src, size_t count) {
    const size_t N = p->N;
    if (count != (size_t)p->batch * p->limbs * N) return fail(LR_ERR_SHAPE, "dense copy: element count != batch*limbs*N");
    LR_HIP(hipSetDevice(p->device));
    LR_HIP(hipStreamSynchronize(p->ctx->stream));
    // logical limbs per poly; the device stride is larger after a rescale re-sliced the poly
    const size_t chunk = (size_t)p->limbs * N;
    const bool dense = p->stride() == (long long)chunk;
    const int pieces = dense ? 1 : p->batch;
    const size_t piece = dense ? count : chunk;
    for (int b = 0; b < pieces; ++b) {
        u64 *dev = p->d + (long long)b * p->stride();
        if (host_src)
            LR_HIP(hipMemcpy(dev, host_src + (size_t)b * chunk, piece * sizeof(u64), hipMemcpyHostToDevice));
        else
            LR_HIP(hipMemcpy(host + (size_t)b * chunk, dev, piece * sizeof(u64), hipMemcpyDeviceToHost));
    }
    return LR_OK;
}

// Poly.MarshalBinary / UnmarshalBinary image (ring/ring_object.go:159-176,222-229,252-270): byte 0 = log2 N, byte 1 = number
// of moduli, then limb-major big-endian words.  The payload goes host <-> device as it is; the byte swap runs on
// the device.
extern "C" int lr_poly_unmarshal(lr_poly *p, int batch_index, const uint8_t *data, size_t len) {
    return guarded([&]() -> int {
    if (!p || !data) return fail(LR_ERR_ARG, "null argument");
    if (batch_index < 0 || batch_index >= p->batch) return fail(LR_ERR_SHAPE, "batch index out of range");
    if (len < 2) return fail(LR_ERR_ARG, "error : invalid polynomial encoding");
    const unsigned logn = data[0];
    const int limbs = data[1];
    if (logn > 63 || ((u64)1 << logn) != p->N) return fail(LR_ERR_SHAPE, "encoded degree differs from the poly's");
    if (limbs > p->limbs) return fail(LR_ERR_SHAPE, "encoding has more moduli than the poly");
    const size_t words = (size_t)limbs * p->N;
    if (len - 2 != words * 8) return fail(LR_ERR_ARG, "error : invalid polynomial encoding");   // :262-264
    LR_HIP(hipSetDevice(p->device));
    u64 *stage = nullptr;
    LR_HIP(hipMalloc((void **)&stage, words * 8 + 8));
    hipError_t e = hipMemcpyAsync(stage, data + 2, words * 8, hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess) e = launch_bswap(stage, p->d + (long long)batch_index * p->stride(), words, p->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);
    (void)hipFree(stage);
    LR_HIP(e);
    return LR_OK;
    });
}

extern "C" int lr_poly_marshal(const lr_poly *p, int batch_index, uint8_t *data, size_t capacity, size_t *written) {
    return guarded([&]() -> int {
    if (!p || !data) return fail(LR_ERR_ARG, "null argument");
    if (batch_index < 0 || batch_index >= p->batch) return fail(LR_ERR_SHAPE, "batch index out of range");
    if (p->limbs > 255) return fail(LR_ERR_UNSUPPORTED, "the encoding holds the number of moduli in one byte");
    const size_t words = (size_t)p->limbs * p->N;
    if (capacity < words * 8 + 2) return fail(LR_ERR_ARG, "Data array is too small to write ring.Poly");   // :164-167
    unsigned logn = 0;
    while (((u64)1 << logn) < p->N) ++logn;
    data[0] = (uint8_t)logn;                                                                              // :168
    data[1] = (uint8_t)p->limbs;                                                                          // :169
    LR_HIP(hipSetDevice(p->device));
    u64 *stage = nullptr;
    LR_HIP(hipMalloc((void **)&stage, words * 8 + 8));
    hipError_t e = launch_bswap(p->d + (long long)batch_index * p->stride(), stage, words, p->ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(data + 2, stage, words * 8, hipMemcpyDeviceToHost, p->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);
    (void)hipFree(stage);
    LR_HIP(e);
    if (written) *written = words * 8 + 2;
    return LR_OK;
    });
}

extern "C" int lr_poly_upload_dense(lr_poly *p, const uint64_t *host, size_t count) {
    return guarded([&]() -> int {
    if (!p || !host) return fail(LR_ERR_ARG, "null argument");
    return dense_copy(p, nullptr, host, count);
    });
}

extern "C" int lr_poly_download_dense(const lr_poly *p, uint64_t *host, size_t count) {
    return guarded([&]() -> int {
    if (!p || !host) return fail(LR_ERR_ARG, "null argument");
    return dense_copy(p, host, nullptr, count);
    });
}

// ------------------------------------------------------------------------------------------
// NTT
// ------------------------------------------------------------------------------------------
namespace {

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

// N = 2^16: the input rows either carry the top stage already (`pretop`) or are disjoint from the output rows
// The forward kernels with the subtract-multiply-add epilogue: the dual kernels "m4" (FP64 body below 2^46, integer body -- mode 2 -- for
// the other limbs) where the context runs the dual kernels, the integer kernels "m5" where it runs mode 1 (q <= 2^60: the reference's
// 60-bit rings).  Contexts on the other integer variants (q up to 2^61, or every modulus in [2^46, 2^57)) keep the separate pass.
bool ntt_epilogue_ok(const lr_context *c) {
    const unsigned logn = c->h.logN;
    if (!c->use_asm || logn < 12 || logn > 16 || !ntt_asm_available((int)logn) || c->opt.no_epilogue) return false;
    return c->asm_fwd == 3 || (c->asm_fwd == 1 && !c->opt.no_int_epilogue);
}
// does limb l of the context take the epilogue (otherwise: plain transform + submul_kernel)?
bool ntt_epilogue_limb(const lr_context *c, int l) {
    if (!ntt_epilogue_ok(c)) return false;
    return c->asm_fwd == 1 || c->h.q[l] < kFpLimit || !c->opt.no_int_epilogue;
}
// the epilogue constant cc (plain domain, below q) of limb l in the form that limb's kernel body reads
EpiLimb make_epi_limb(const lr_context *c, int l, u64 cc) {
    const u64 q = c->h.q[l];
    if (c->asm_fwd == 3 && q < kFpLimit) return EpiLimb{(double)cc, (double)cc / (double)q};
    const u64 pair[2] = {cc, shoup_companion(cc, q)};
    EpiLimb e;
    static_assert(sizeof(e) == sizeof(pair), "EpiLimb is 16 bytes");
    std::memcpy(&e, pair, sizeof e);
    return e;
}

// N = 2^15 transforms as two 2^14 sub-blocks (run_ntt_launch): for launches of at most Options::split15_max_workgroups (128) workgroups -- split, they still fit
// one round on the 256 CUs.
bool ntt_split15(const lr_context *c, long long workgroups) {
    if (c->h.logN != 15 || !c->use_asm || c->opt.timeline || !ntt_asm_available(15)) return false;
    const int variant_f = c->asm_fwd, variant_i = c->asm_inv;
    if (variant_f < 0 || variant_i < 0) return false;
    if (c->opt.split15 >= 0) return c->opt.split15 == 1;
    if (c->opt.persist > 0) return false;          // (LR_NTT_PERSIST asks for the persistent one-workgroup kernels: diagnostics)
    return workgroups <= c->opt.split15_max_workgroups;
}

// Fork: launches of the calling thread that go to a plan's auxiliary stream instead of the context's (PlanFork, below): two independent
// transforms of a small batch run side by side instead of one after the other.  Only forward transforms are forked (they lease no scratch).
thread_local hipStream_t g_fork_stream = nullptr;
inline hipStream_t stream_of(const lr_context *c) { return g_fork_stream ? g_fork_stream : c->stream; }

int run_ntt_launch(lr_context *c, bool inverse, Rows in, Rows out, int mod0, int mod_step, int count, int batch, int hole,
                   int group, const NttEpilogue *epi, bool pretop, bool lazy);

// Polys per workgroup of the persistent forward 2^15 kernels (0 = the one-poly kernels).  LR_NTT_PERSIST overrides; the default keeps
// at least four rounds of workgroups on the 256 CUs (the dispatcher balances limbs of different cost -- FP64 and integer bodies in one
// dual launch -- by rounds) and at most kPersistMax polys per workgroup.
constexpr int kPersistDefault = 0;
int ntt_persist(const lr_context *c, const NttLaunch &a, unsigned logn, bool inverse) {
    if (logn != 15 || inverse) return 0;
    const int polys = a.hole > 0 ? a.group : a.batch;
    int p = c->opt.persist >= 0 ? c->opt.persist : kPersistDefault;
    if (p > polys) p = polys;
    return p >= 2 ? p : 0;
}

// The assembly kernels of the integer variants put the polynomial on grid.y (limit 65535): longer plain launches are cut into
// chunks along the batch on the same kernel (no silent change of code path).  Grouped launches (key-switch digits) beyond
// the limit are refused: 65536 ciphertexts in one key switch exceed the device memory by orders of magnitude.
// pretop (N = 2^16, forward, assembly kernels): the producer of the input rows has already applied the stage over index bit 15
// (ext_sum_kernel<.., true>); the launch goes straight to the plain sub-block kernels, which read their own half only.
// lazy (inverse, N = 2^15 / 2^16 on the assembly sub-block kernels): the rows are left as the two halves of every limb before the last
// Gentleman-Sande stage and the scaling -- for a consumer that applies them itself (the top-stage basis extension, ExtLaunch::inv_top)
int run_ntt(lr_context *c, bool inverse, Rows in, Rows out, int mod0, int mod_step, int count, int batch, int hole = 0,
            int group = 0, const NttEpilogue *epi = nullptr, bool pretop = false, bool lazy = false) {
    if (count <= 0 || batch <= 0) return LR_OK;
    if (hole > 0 && (group <= 0 || batch % group != 0)) return fail(LR_ERR_ARG, "digit groups must divide the batch");
    // N = 2^16: the streaming top-stage kernel carries poly * limbs on grid.y
    const int kChunk = c->h.logN == 16 || (c->h.logN == 15 && c->opt.split15 == 1) ? std::max(1, 65535 / count) : 65535;
    if (hole > 0) {
        if (group > kChunk || batch / group > 65535) return fail(LR_ERR_UNSUPPORTED, "grouped NTT launch: more than 65535 polys per digit group");
        return run_ntt_launch(c, inverse, in, out, mod0, mod_step, count, batch, hole, group, epi, pretop, lazy);
    }
    for (int b0 = 0; b0 < batch; b0 += kChunk) {
        const int nb = std::min(kChunk, batch - b0);
        Rows ci = in, co = out;
        ci.base = in.base + (long long)b0 * in.stride;
        co.base = out.base + (long long)b0 * out.stride;
        NttEpilogue e2;
        if (epi) {
            e2 = *epi;
            e2.x = epi->x + (long long)b0 * epi->x_stride;
            e2.plus = epi->plus + (long long)b0 * epi->plus_stride;
        }
        LR_TRY(run_ntt_launch(c, inverse, ci, co, mod0, mod_step, count, nb, 0, 0, epi ? &e2 : nullptr, pretop, lazy));
    }
    return LR_OK;
}

int run_ntt_launch(lr_context *c, bool inverse, Rows in, Rows out, int mod0, int mod_step, int count, int batch, int hole,
                   int group, const NttEpilogue *epi, bool pretop, bool lazy) {
    const unsigned logn = c->h.logN;
    if (logn < 1 || logn > 16)
        return fail(LR_ERR_UNSUPPORTED, "NTT kernels cover 2 <= N <= 2^16");
    NttLaunch a;
    a.in = in.base;
    a.out = out.base;
    a.in_poly_stride = in.stride;
    a.out_poly_stride = out.stride;
    a.in_limb0 = in.limb0;
    a.in_limb_step = in.step;
    a.out_limb0 = out.limb0;
    a.out_limb_step = out.step;
    a.mod0 = mod0;
    a.mod_step = mod_step;
    a.n_items = count;
    a.sub_log = 0;
    a.hole = hole;
    a.group = group;
    a.fuse_top = 0;
    a.batch = batch;
    a.lp = c->d_lp;
    a.tw = inverse ? c->d_inv : c->d_fwd;
    a.tw_fin = inverse ? c->d_inv_fin : c->d_fwd_fin;
    const int variant = inverse ? c->asm_inv : c->asm_fwd;
    a.fp_tw_delta = a.fp_fin_delta = 0;
    a.fp_lp = nullptr;
    a.epi_x = a.epi_plus = nullptr;
    a.epi_x_stride = a.epi_plus_stride = 0;
    a.epi_consts = nullptr;
    a.stagger_gx = a.stagger_unit = 0;
    if (variant == 3) {
        a.fp_tw_delta = (const char *)(inverse ? c->d_inv_fp : c->d_fwd_fp) - (const char *)a.tw;
        a.fp_fin_delta = (const char *)(inverse ? c->d_inv_fin_fp : c->d_fwd_fin_fp) - (const char *)a.tw_fin;
        a.fp_lp = c->d_fp_lp;
    }
    char *kn = c->last_ntt_kernel;
    // N = 2^14: 512 threads per transform put two workgroups on a CU (best throughput); a launch that does not fill the chip anyway takes
    // the 1024-thread plan, whose one workgroup is done sooner (PN14QP438, one ciphertext: MulRelin 115 -> 102 us, BFV Mul 136 -> 125 us)
    const bool wide14 = c->opt.asm14_1024 || (logn == 14 && !c->opt.no_wide14_small && (long long)count * batch <= c->opt.wide14_max_items);
    // N = 2^15, a launch too small to fill the chip with one workgroup per transform (a one-workgroup 2^15 transform takes ~42 us whatever
    // surrounds it): two 2^14 sub-blocks per limb on the "h" kernels, twice the workgroups at about half the latency.  The stage over
    // index bit 14 is the streaming ntt_top_kernel's (forward: before, unless the caller's basis extension has applied it -- pretop;
    // inverse: after, with the scaling).  A caller that passes pretop has decided for the split itself (ntt_split15).
    if (lazy && !(inverse && !epi && (logn == 15 || logn == 16) && variant >= 0 && c->use_asm && ntt_asm_available((int)logn)))
        return fail(LR_ERR_ARG, "lazy inverse outputs: assembly sub-block kernels of N = 2^15 / 2^16 only");
    if (logn == 15 && (pretop || lazy || (ntt_split15(c, (long long)count * batch) && !(epi && !pretop)))) {
        if (variant < 0 || !c->use_asm || !ntt_asm_available(15)) return fail(LR_ERR_ARG, "pre-applied top stage: assembly kernels only");
        if (epi) {
            if (inverse || hole > 0 || !ntt_epilogue_ok(c)) return fail(LR_ERR_ARG, "NTT epilogue: not available for this launch");
            a.epi_x = epi->x;
            a.epi_x_stride = epi->x_stride;
            a.epi_plus = epi->plus;
            a.epi_plus_stride = epi->plus_stride;
            a.epi_consts = epi->consts;
            LR_HIP(launch_ntt_asm16(a, 0, 'h', c->asm_fwd == 3 ? 4 : 5, stream_of(c), kn, c->opt.stagger, 15));
            return LR_OK;
        }
        if (!inverse) {
            NttLaunch sub = a;
            if (!pretop) {
                LR_HIP(launch_ntt_top(a, 0, stream_of(c), 15));
                sub.in = a.out;                  // continue in place on the output rows
                sub.in_poly_stride = a.out_poly_stride;
                sub.in_limb0 = a.out_limb0;
                sub.in_limb_step = a.out_limb_step;
            }
            LR_HIP(launch_ntt_asm16(sub, 0, 'h', variant, stream_of(c), kn, c->opt.stagger, 15));
            return LR_OK;
        }
        LR_HIP(launch_ntt_asm16(a, 1, 'h', variant, stream_of(c), kn, c->opt.stagger, 15));
        if (lazy) return LR_OK;
        NttLaunch top = a;
        top.in = a.out;
        top.in_poly_stride = a.out_poly_stride;
        top.in_limb0 = a.out_limb0;
        top.in_limb_step = a.out_limb_step;
        LR_HIP(launch_ntt_top(top, 1, stream_of(c), 15));
        return LR_OK;
    }
    if (epi) {
        if (inverse || hole > 0 || !ntt_epilogue_ok(c) || (logn == 16 && !pretop && !ntt_rows_disjoint(a, 16)))
            return fail(LR_ERR_ARG, "NTT epilogue: not available for this launch");
        a.epi_x = epi->x;
        a.epi_x_stride = epi->x_stride;
        a.epi_plus = epi->plus;
        a.epi_plus_stride = epi->plus_stride;
        a.epi_consts = epi->consts;
        if (logn == 16)
            LR_HIP(launch_ntt_asm16(a, 0, pretop ? 'p' : 's', c->asm_fwd == 3 ? 4 : 5, stream_of(c), kn, c->opt.stagger));
        else
            LR_HIP(launch_ntt_asm(a, (int)logn, 0, c->asm_fwd == 3 ? 4 : 5, stream_of(c), wide14, kn, false, c->opt.stagger, 0, !c->opt.no_grid_padding));
        return LR_OK;
    }
    if (logn == 16 && variant >= 0 && c->use_asm && ntt_asm_available(16)) {
        // two 2^15 sub-blocks per limb on the assembly kernels + the streaming stage over bit 15
        if (!inverse) {
            if (pretop) {
                LR_HIP(launch_ntt_asm16(a, 0, 'p', variant, stream_of(c), kn, c->opt.stagger));
                return LR_OK;
            }
            if (ntt_rows_disjoint(a, 16)) {
                LR_HIP(launch_ntt_asm16(a, 0, 's', variant, stream_of(c), kn, c->opt.stagger));     // top stage fused into the loads
                return LR_OK;
            }
            LR_HIP(launch_ntt_top(a, 0, stream_of(c)));
            NttLaunch sub = a;
            sub.in = a.out;                      // continue in place on the output rows
            sub.in_poly_stride = a.out_poly_stride;
            sub.in_limb0 = a.out_limb0;
            sub.in_limb_step = a.out_limb_step;
            LR_HIP(launch_ntt_asm16(sub, 0, 'p', variant, stream_of(c), kn, c->opt.stagger));
            return LR_OK;
        }
        if (lazy) {
            LR_HIP(launch_ntt_asm16(a, 1, 's', variant, stream_of(c), kn, c->opt.stagger));
            return LR_OK;
        }
        if (!c->opt.no_invfuse && hole == 0) {
            // the last stage inside the sub-block kernels: the wave that finishes second of a limb's two sub-blocks combines both
            // halves (gen_intt.py: fused_last); one u32 flag per wave pair, zeroed here, addressed through NttLaunch::epi_x
            ScratchLease flags;
            const size_t flag_bytes = (size_t)batch * (size_t)count * 16 * sizeof(u32);
            LR_TRY(flags.take(&c->scratch, (flag_bytes + 7) / 8));
            LR_HIP(hipMemsetAsync(flags.d(), 0, flag_bytes, stream_of(c)));
            a.epi_x = flags.d();
            LR_HIP(launch_ntt_asm16(a, 1, 'f', variant, stream_of(c), kn, c->opt.stagger));
            return LR_OK;
        }
        LR_HIP(launch_ntt_asm16(a, 1, 's', variant, stream_of(c), kn, c->opt.stagger));
        NttLaunch top = a;
        top.in = a.out;
        top.in_poly_stride = a.out_poly_stride;
        top.in_limb0 = a.out_limb0;
        top.in_limb_step = a.out_limb_step;
        LR_HIP(launch_ntt_top(top, 1, stream_of(c)));
        return LR_OK;
    }
    if (pretop) return fail(LR_ERR_ARG, "pre-applied top stage: only for forward N = 2^16 launches on the assembly kernels");
    if (logn != 16 && variant >= 0 && c->use_asm && ntt_asm_available((int)logn)) {
        if (c->opt.timeline && logn == 15 && (variant == 1 || variant == 3) && hole == 0) {
            // diagnostics: the stamped build of the same kernel; stamps land in the context's buffer (lr_context_timeline)
            const size_t words = (size_t)batch * (size_t)count * 16 * 16;
            if (words > c->stamp_words) {
                LR_HIP(hipStreamSynchronize(stream_of(c)));
                if (c->d_stamps) LR_HIP(hipFree(c->d_stamps));
                c->d_stamps = nullptr;
                c->stamp_words = 0;
                LR_HIP(hipMalloc((void **)&c->d_stamps, words * sizeof(u32)));
                c->stamp_words = words;
            }
            c->stamp_used = words;
            a.epi_x = reinterpret_cast<const u64 *>(c->d_stamps);
            LR_HIP(launch_ntt_asm(a, (int)logn, inverse, variant, stream_of(c), false, kn, true, c->opt.stagger, ntt_persist(c, a, logn, inverse), !c->opt.no_grid_padding));
            return LR_OK;
        }
        LR_HIP(launch_ntt_asm(a, (int)logn, inverse, variant, stream_of(c), wide14, kn, false, c->opt.stagger, ntt_persist(c, a, logn, inverse), !c->opt.no_grid_padding));
        return LR_OK;
    }
    std::snprintf(c->last_ntt_kernel, sizeof c->last_ntt_kernel, "ntt_%s_kernel<%u>", inverse ? "inv" : "fwd", logn);
    LR_HIP(launch_ntt(a, (int)logn, inverse, c->ntt_mode, stream_of(c)));
    return LR_OK;
}

int check_pair(const lr_context *c, int level, const lr_poly *in, const lr_poly *out) {
    if (!c || !in || !out) return fail(LR_ERR_ARG, "null argument");
    if (in->N != c->h.N || out->N != c->h.N) return fail(LR_ERR_SHAPE, "ring degree mismatch");
    if (level < 0 || level + 1 > c->h.L()) return fail(LR_ERR_SHAPE, "level exceeds the context's modulus count");
    if (level + 1 > in->limbs || level + 1 > out->limbs) return fail(LR_ERR_SHAPE, "poly has fewer limbs than level+1");
    if (in->batch != out->batch && in->batch != 1) return fail(LR_ERR_SHAPE, "batch mismatch");
    return LR_OK;
}

Rows rows_of(const lr_poly *p, int limb0 = 0, int step = 1, bool broadcast_ok = false, int target_batch = 0) {
    Rows r;
    r.base = p->d;
    r.stride = (broadcast_ok && p->batch == 1 && target_batch > 1) ? 0 : p->stride();
    r.limb0 = limb0;
    r.step = step;
    return r;
}

}  // namespace

extern "C" int lr_ntt(lr_context *c, int level, const lr_poly *in, lr_poly *out) {
    return guarded([&]() -> int {
    LR_TRY(check_pair(c, level, in, out));
    if (in->batch != out->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(c->device));
    return run_ntt(c, false, rows_of(in), rows_of(out), 0, 1, level + 1, out->batch);
    });
}

extern "C" int lr_intt(lr_context *c, int level, const lr_poly *in, lr_poly *out) {
    return guarded([&]() -> int {
    LR_TRY(check_pair(c, level, in, out));
    if (in->batch != out->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(c->device));
    return run_ntt(c, true, rows_of(in), rows_of(out), 0, 1, level + 1, out->batch);
    });
}

static int ntt_limb(lr_context *c, bool inverse, int mod_index, const lr_poly *in, int in_limb, lr_poly *out, int out_limb) {
    if (!c || !in || !out) return fail(LR_ERR_ARG, "null argument");
    if (mod_index < 0 || mod_index >= c->h.L()) return fail(LR_ERR_SHAPE, "modulus index out of range");
    if (in_limb < 0 || in_limb >= in->limbs || out_limb < 0 || out_limb >= out->limbs)
        return fail(LR_ERR_SHAPE, "limb index out of range");
    if (in->batch != out->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(c->device));
    return run_ntt(c, inverse, rows_of(in, in_limb, 0), rows_of(out, out_limb, 0), mod_index, 0, 1, out->batch);
}

extern "C" int lr_ntt_limb(lr_context *c, int mod_index, const lr_poly *in, int in_limb, lr_poly *out, int out_limb) {
    return guarded([&]() -> int {
    return ntt_limb(c, false, mod_index, in, in_limb, out, out_limb);
    });
}
extern "C" int lr_intt_limb(lr_context *c, int mod_index, const lr_poly *in, int in_limb, lr_poly *out, int out_limb) {
    return guarded([&]() -> int {
    return ntt_limb(c, true, mod_index, in, in_limb, out, out_limb);
    });
}

static int ntt_host(lr_context *c, bool inverse, int level, const uint64_t *const *in_limbs, uint64_t *const *out_limbs) {
    if (!c || !in_limbs || !out_limbs) return fail(LR_ERR_ARG, "null argument");
    if (level < 0 || level + 1 > c->h.L()) return fail(LR_ERR_SHAPE, "level exceeds the context's modulus count");
    lr_poly *tmp = nullptr;
    LR_TRY(lr_poly_alloc(c, level + 1, 1, &tmp));
    int rc = lr_poly_upload(tmp, 0, in_limbs, level + 1);
    if (rc == LR_OK) rc = inverse ? lr_intt(c, level, tmp, tmp) : lr_ntt(c, level, tmp, tmp);
    if (rc == LR_OK) rc = lr_poly_download(tmp, 0, out_limbs, level + 1);
    lr_poly_free(tmp);
    return rc;
}

// the package-level ring.NTT / ring.InvNTT (ring/ntt.go:53,89): one limb under modulus `mod_index` of the context, host slices in
// and out (upload, kernel, download); may be in place
extern "C" int lr_ntt_host_limb(lr_context *c, int mod_index, int inverse, const uint64_t *in, uint64_t *out) {
    return guarded([&]() -> int {
    if (!c || !in || !out) return fail(LR_ERR_ARG, "null argument");
    if (mod_index < 0 || mod_index >= c->h.L()) return fail(LR_ERR_SHAPE, "modulus index out of range");
    lr_poly *tmp = nullptr;
    LR_TRY(lr_poly_alloc(c, 1, 1, &tmp));
    int rc = lr_poly_upload_limb(tmp, 0, 0, in);
    if (rc == LR_OK) rc = ntt_limb(c, inverse != 0, mod_index, tmp, 0, tmp, 0);
    if (rc == LR_OK) rc = lr_poly_download_limb(tmp, 0, 0, out);
    lr_poly_free(tmp);
    return rc;
    });
}

extern "C" int lr_ntt_host(lr_context *c, int level, const uint64_t *const *in_limbs, uint64_t *const *out_limbs) {
    return guarded([&]() -> int {
    return ntt_host(c, false, level, in_limbs, out_limbs);
    });
}
extern "C" int lr_intt_host(lr_context *c, int level, const uint64_t *const *in_limbs, uint64_t *const *out_limbs) {
    return guarded([&]() -> int {
    return ntt_host(c, true, level, in_limbs, out_limbs);
    });
}

// ------------------------------------------------------------------------------------------
// coefficient-wise
// ------------------------------------------------------------------------------------------
namespace {

bool op_reads_b(int op) {
    return op == LR_ADD || op == LR_ADD_NOMOD || op == LR_SUB || op == LR_SUB_NOMOD ||
           (op >= LR_MUL_COEFFS && op <= LR_MUL_MONT_CONSTANT);
}

// raw form used by the pipelines: pointers are already offset to limb 0 of the operands
int run_ewise(lr_context *c, int op, int limbs, int batch, const u64 *a, long long a_stride, const u64 *b,
              long long b_stride, u64 *out, long long out_stride, const LimbScalars *sc, int lp_offset = 0) {
    EwiseLaunch L;
    L.a = a;
    L.b = b;
    L.out = out;
    L.a_stride = a_stride;
    L.b_stride = b_stride;
    L.out_stride = out_stride;
    L.n = (int)c->h.N;
    L.lp = c->d_lp + lp_offset;
    L.has_scalars = sc ? 1 : 0;
    if (sc) L.scalars = *sc;
    LR_HIP(launch_ewise(op, L, limbs, batch, c->stream));
    return LR_OK;
}

}  // namespace

extern "C" int lr_ewise(lr_context *c, int op, int level, const lr_poly *a, const lr_poly *b, lr_poly *out,
                        const uint64_t *scalars) {
    return guarded([&]() -> int {
    if (!c || !a || !out) return fail(LR_ERR_ARG, "null argument");
    if (op < 0 || op >= LR_EWISE_OP_COUNT) return fail(LR_ERR_ARG, "unknown coefficient-wise op");
    LR_TRY(check_pair(c, level, a, out));
    const bool needs_b = op_reads_b(op);
    if (needs_b) {
        if (!b) return fail(LR_ERR_ARG, "this op needs a second operand");
        LR_TRY(check_pair(c, level, b, out));
    }
    if (c->h.N < 2) return fail(LR_ERR_UNSUPPORTED, "N must be at least 2");
    LR_HIP(hipSetDevice(c->device));
    LimbScalars sc;
    const LimbScalars *scp = nullptr;
    const int limbs = level + 1;
    if (op == LR_MUL_SCALAR || op == LR_MUL_SCALAR_LIMBS || op == LR_ADD_SCALAR_LIMBS || op == LR_SUB_SCALAR_LIMBS ||
        op == LR_MUL_BY_POW2) {
        if (!scalars) return fail(LR_ERR_ARG, "this op needs scalars");
        for (int i = 0; i < limbs; ++i) {
            const u64 q = c->h.q[i];
            const BarrettConst bc = c->h.bred[i];
            switch (op) {
            case LR_MUL_SCALAR: sc.v[i] = mform(bred_add(scalars[0], q, bc.hi), q, bc.hi, bc.lo); break;      // ring.go:516
            case LR_MUL_SCALAR_LIMBS: sc.v[i] = mform(bred_add(scalars[i], q, bc.hi), q, bc.hi, bc.lo); break; // ring.go:547
            case LR_MUL_BY_POW2: sc.v[i] = scalars[0]; break;
            default: sc.v[i] = scalars[i]; break;
            }
        }
        scp = &sc;
    }
    const int batch = out->batch;
    const long long as = (a->batch == 1 && batch > 1) ? 0 : a->stride();
    const long long bs = (b && b->batch == 1 && batch > 1) ? 0 : (b ? b->stride() : 0);
    if (op == LR_MUL_BY_POW2 && a->d == out->d) {
        // MulByPow2 in place: the reference first overwrites p2 with MForm(p1), ring/ring.go:630
        LR_TRY(run_ewise(c, LR_MFORM, limbs, batch, a->d, as, nullptr, 0, out->d, out->stride(), nullptr));
    }
    return run_ewise(c, op, limbs, batch, a->d, as, needs_b ? b->d : nullptr, bs, out->d, out->stride(), scp);
    });
}

// ------------------------------------------------------------------------------------------
// half-vector scalar operations (the constant-by-ciphertext methods of ckks.Evaluator)
// ------------------------------------------------------------------------------------------
extern "C" int lr_half_scalar_op(lr_context *c, int op, int level, const lr_poly *in, const uint64_t *lo, const uint64_t *hi, lr_poly *out) {
    return guarded([&]() -> int {
    LR_TRY(check_pair(c, level, in, out));
    if (in->batch != out->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    if (!lo || !hi) return fail(LR_ERR_ARG, "null scalar array");
    if (op < 0 || op > 2) return fail(LR_ERR_ARG, "half-vector scalar op: 0 = add, 1 = multiply, 2 = multiply and add");
    if (c->h.N < 4) return fail(LR_ERR_UNSUPPORTED, "half-vector scalar op: ring degree below 4");
    LR_HIP(hipSetDevice(c->device));
    HalfScalarLaunch L;
    L.in = in->d;
    L.out = out->d;
    L.in_stride = in->stride();
    L.out_stride = out->stride();
    L.n = (int)c->h.N;
    L.op = op;
    L.lp = c->d_lp;
    std::memset(&L.lo, 0, sizeof(L.lo));
    std::memset(&L.hi, 0, sizeof(L.hi));
    for (int i = 0; i <= level; ++i) {
        L.lo.v[i] = lo[i];
        L.hi.v[i] = hi[i];
    }
    LR_HIP(launch_half_scalar(L, level + 1, out->batch, c->stream));
    return LR_OK;
    });
}

// ------------------------------------------------------------------------------------------
// Galois automorphisms (ring/ring_galois.go)
// ------------------------------------------------------------------------------------------
static int permute_common(lr_context *c, int level, const lr_poly *in, u64 gen, lr_poly *out, bool ntt_domain) {
    LR_TRY(check_pair(c, level, in, out));
    if (in->batch != out->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    if (in->d == out->d) return fail(LR_ERR_ARG, "Permute is not in place (ring/ring_galois.go:54)");
    if (c->h.N < 2 || c->h.logN > 31) return fail(LR_ERR_UNSUPPORTED, "ring degree");
    LR_HIP(hipSetDevice(c->device));
    GaloisLaunch L;
    L.in = in->d;
    L.out = out->d;
    L.in_stride = in->stride();
    L.out_stride = out->stride();
    L.n = (int)c->h.N;
    L.logn = (int)c->h.logN;
    L.ntt_domain = ntt_domain ? 1 : 0;
    // only gen mod 2N matters in either domain (indices are taken mod 2N resp. mod N with the sign from bit logN)
    L.gen = gen & ((c->h.N << 1) - 1);
    L.lp = c->d_lp;
    LR_HIP(launch_permute(L, level + 1, out->batch, c->stream));
    return LR_OK;
}

extern "C" int lr_permute_ntt(lr_context *c, int level, const lr_poly *in, uint64_t gen, lr_poly *out) {
    return guarded([&]() -> int {
    if (!c || !in || !out) return fail(LR_ERR_ARG, "null argument");
    return permute_common(c, level, in, gen, out, true);
    });
}

extern "C" int lr_permute(lr_context *c, const lr_poly *in, uint64_t gen, lr_poly *out) {
    return guarded([&]() -> int {
    if (!c || !in || !out) return fail(LR_ERR_ARG, "null argument");
    return permute_common(c, c->h.L() - 1, in, gen, out, false);
    });
}

extern "C" int lr_mult_by_monomial(lr_context *c, const lr_poly *in, uint64_t monomial_deg, lr_poly *out) {
    return guarded([&]() -> int {
    if (!c || !in || !out) return fail(LR_ERR_ARG, "null argument");
    const int level = c->h.L() - 1;
    LR_TRY(check_pair(c, level, in, out));
    if (in->batch != out->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(c->device));
    GaloisLaunch L;
    L.in = in->d;
    L.out = out->d;
    L.in_stride = in->stride();
    L.out_stride = out->stride();
    // in place: through a temporary, as the reference does for every call (tmpx, ring/ring.go:682-693)
    ScratchLease tmp;
    const bool alias = in->d == out->d;
    if (alias) {
        LR_TRY(tmp.take(&c->scratch, (size_t)out->batch * (size_t)in->stride()));
        LR_HIP(hipMemcpyAsync(tmp.d(), in->d, (size_t)out->batch * (size_t)in->stride() * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        L.in = tmp.d();
    }
    L.n = (int)c->h.N;
    L.logn = (int)c->h.logN;
    L.ntt_domain = 0;
    L.gen = monomial_deg % (c->h.N << 1);      // ring/ring.go:667
    L.lp = c->d_lp;
    LR_HIP(launch_monomial(L, level + 1, out->batch, c->stream));
    return LR_OK;
    });
}

// Context.Shift (ring/ring.go:575-580): p2 = p1 rotated left by n coefficient positions, every limb.  The reference masks n with
// (1 << N) - 1, which in Go is all ones for N >= 64 and 2^N - 1 below, and slices p1.Coeffs[i][n:]: n > N panics (here: LR_ERR_ARG).
extern "C" int lr_shift(lr_context *c, const lr_poly *in, uint64_t n, lr_poly *out) {
    return guarded([&]() -> int {
    if (!c || !in || !out) return fail(LR_ERR_ARG, "null argument");
    const int level = c->h.L() - 1;
    LR_TRY(check_pair(c, level, in, out));
    if (in->batch != out->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    const u64 N = c->h.N;
    const u64 m = N >= 64 ? n : (n & (((u64)1 << N) - 1));
    if (m > N) return fail(LR_ERR_ARG, "Shift: n exceeds the ring degree (the reference's slice expression panics)");
    LR_HIP(hipSetDevice(c->device));
    const u64 *src = in->d;
    ScratchLease tmp;
    if (in->d == out->d) {
        LR_TRY(tmp.take(&c->scratch, (size_t)out->batch * (size_t)in->stride()));
        LR_HIP(hipMemcpyAsync(tmp.d(), in->d, (size_t)out->batch * (size_t)in->stride() * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        src = tmp.d();
    }
    const size_t pitch = (size_t)N * sizeof(u64), rows = (size_t)(level + 1);
    for (int b = 0; b < out->batch; ++b) {
        const u64 *s = src + (long long)b * in->stride();
        u64 *d = out->d + (long long)b * out->stride();
        if (m < N) LR_HIP(hipMemcpy2DAsync(d, pitch, s + m, pitch, (size_t)(N - m) * sizeof(u64), rows, hipMemcpyDeviceToDevice, c->stream));
        if (m > 0) LR_HIP(hipMemcpy2DAsync(d + (N - m), pitch, s, pitch, (size_t)m * sizeof(u64), rows, hipMemcpyDeviceToDevice, c->stream));
    }
    return LR_OK;
    });
}

// Context.Rotate (ring/ring.go:775-800): coefficient j of every limb is multiplied by omega^(n j), omega = psi^2, for j = 1 .. N-1;
// coefficient 0 is left as it is.  The reference writes the result into p1 whatever p2 is (`p1tmp, p2tmp := p1.Coeffs[i], p1.Coeffs[i]`,
// :791), so this entry point takes one poly.  n is masked like Shift's.  The factors gal_j = MForm(omega^(n j)) are canonical residues and
// MRed(x, gal_j) is the canonical x * omega^(n j): the table is built on the host per call (the reference's only caller is its test
// suite, ring_test.go:435) and applied by the Montgomery product kernel.
extern "C" int lr_rotate(lr_context *c, lr_poly *p1, uint64_t n) {
    return guarded([&]() -> int {
    if (!c || !p1) return fail(LR_ERR_ARG, "null argument");
    const int level = c->h.L() - 1;
    LR_TRY(check_pair(c, level, p1, p1));
    const u64 N = c->h.N;
    if (N < 2) return fail(LR_ERR_UNSUPPORTED, "N must be at least 2");
    const u64 m = N >= 64 ? n : (n & (((u64)1 << N) - 1));
    LR_HIP(hipSetDevice(c->device));
    const int L = level + 1;
    std::vector<u64> gal((size_t)L * N);
    for (int i = 0; i < L; ++i) {
        const u64 q = c->h.q[i], qinv = c->h.mred[i];
        const BarrettConst bc = c->h.bred[i];
        const u64 omega = mred(c->h.psi_mont[i], c->h.psi_mont[i], q, qinv);              // psi^2 in Montgomery form (:785)
        // root = omega^m in Montgomery form (:787): square and multiply on Montgomery residues
        u64 root = mform(1, q, bc.hi, bc.lo), base = omega;
        for (u64 e = m; e > 0; e >>= 1) {
            if (e & 1) root = mred(root, base, q, qinv);
            base = mred(base, base, q, qinv);
        }
        u64 g = mform(1, q, bc.hi, bc.lo);                                               // :789
        gal[(size_t)i * N] = g;
        for (u64 j = 1; j < N; ++j) {
            g = mred(g, root, q, qinv);                                                  // :795
            gal[(size_t)i * N + j] = g;
        }
    }
    ScratchLease table, heads;
    const size_t rows = (size_t)p1->batch * (size_t)L;
    LR_TRY(table.take(&c->scratch, gal.size()));
    LR_TRY(heads.take(&c->scratch, rows));
    // the multiply below is not ordered against a host buffer that dies with this call: finish the upload first
    LR_HIP(hipMemcpyAsync(table.d(), gal.data(), gal.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    LR_HIP(hipStreamSynchronize(c->stream));
    const size_t pitch = (size_t)N * sizeof(u64);
    // coefficient 0 of every row is not touched by the reference (the loop starts at j = 1): keep it aside, put it back afterwards
    for (int b = 0; b < p1->batch; ++b)
        LR_HIP(hipMemcpy2DAsync(heads.d() + (size_t)b * L, sizeof(u64), p1->d + (long long)b * p1->stride(), pitch, sizeof(u64), (size_t)L,
                                hipMemcpyDeviceToDevice, c->stream));
    LR_TRY(run_ewise(c, LR_MUL_MONT, L, p1->batch, p1->d, p1->stride(), table.d(), 0, p1->d, p1->stride(), nullptr));
    for (int b = 0; b < p1->batch; ++b)
        LR_HIP(hipMemcpy2DAsync(p1->d + (long long)b * p1->stride(), pitch, heads.d() + (size_t)b * L, sizeof(u64), sizeof(u64), (size_t)L,
                                hipMemcpyDeviceToDevice, c->stream));
    return LR_OK;
    });
}

extern "C" int lr_permute_ntt_index(uint64_t gen, uint64_t power, uint64_t N, uint64_t *index) {
    return guarded([&]() -> int {
    if (!index) return fail(LR_ERR_ARG, "null argument");
    if (N == 0 || (N & (N - 1)) != 0) return fail(LR_ERR_INVALID_DEGREE, "invalid ring degree (must be a power of 2)");
    const u64 gen_pow = mod_exp(gen, power, 2 * N);
    unsigned logn = 0;
    while ((1ull << logn) < N) ++logn;
    const u64 mask = (N << 1) - 1;
    for (u64 i = 0; i < N; ++i) {
        const u64 t1 = 2 * bit_reverse(i, logn) + 1;
        const u64 t2 = ((gen_pow * t1 & mask) - 1) >> 1;
        index[i] = bit_reverse(t2, logn);
    }
    return LR_OK;
    });
}

// ------------------------------------------------------------------------------------------
// basis extension
// ------------------------------------------------------------------------------------------
namespace {

ExtSegment segment(u64 *out, long long stride, int limb0, int col0, int count);

// extensions recorded instead of launched (ks_decompose: the digits of one key switch go out as one grouped launch)
struct ExtPending {
    ExtLaunch L;
    int n_in;
};

// One extension launch, cut into column ranges where that fills the chip better.
int launch_ext_chunked(lr_context *c, const ExtLaunch &L, int n_in, int batch) {
    // A small batch: every thread of the extension walks all target columns of its coefficients, and a launch of n / 2 threads per
    // poly is 32 - 128 workgroups.  The columns are independent: the launch is cut into records over disjoint column ranges that go
    // out as one grouped launch (grid z = range), e.g. ModDown's extension of one PN15QP880 ciphertext 64 -> 256 workgroups.
    if (!c->opt.no_ext_chunks && n_in <= 8) {
        int total = 0;
        for (int k = 0; k < kExtSegments; ++k) total += L.seg[k].count;
        const bool top = L.seg[0].top_tw != nullptr;
        const long long blocks = ((long long)(L.n / (top ? 4 : 2)) + 255) / 256 * batch;
        int chunks = blocks > 0 && blocks < 128 ? (int)std::min<long long>((256 + blocks - 1) / blocks, kExtGroupMax) : 1;
        if (chunks > total) chunks = total;
        if (chunks > 1) {
            ExtLaunch Ls[kExtGroupMax];
            int seg = 0, off = 0;               // next column: segment `seg`, offset `off` inside it
            for (int ch = 0; ch < chunks; ++ch) {
                int want = total / chunks + (ch < total % chunks ? 1 : 0);
                ExtLaunch &R = Ls[ch];
                R = L;
                for (int k = 0; k < kExtSegments; ++k) R.seg[k].count = 0;
                int filled = 0;
                while (want > 0 && seg < kExtSegments) {
                    const int avail = L.seg[seg].count - off;
                    if (avail <= 0) {
                        ++seg;
                        off = 0;
                        continue;
                    }
                    const int take = std::min(avail, want);
                    ExtSegment piece = L.seg[seg];
                    piece.limb0 += off;
                    piece.col0 += off;
                    piece.top_mod0 += off;
                    piece.count = take;
                    R.seg[filled++] = piece;
                    off += take;
                    want -= take;
                }
            }
            const hipError_t e = launch_ext_group(Ls, chunks, n_in, batch, c->stream);
            if (e == hipSuccess) return LR_OK;
            if (e != hipErrorNotSupported) return fail(LR_ERR_HIP, std::string("launch_ext_group: ") + hipGetErrorString(e));
        }
    }
    LR_HIP(launch_ext(L, n_in, batch, c->stream));
    return LR_OK;
}

int flush_ext(lr_context *c, std::vector<ExtPending> &pending, int batch, unsigned long long *grouped_launches = nullptr) {
    size_t i = 0;
    while (i < pending.size()) {
        size_t j = i + 1;
        while (j < pending.size() && pending[j].n_in == pending[i].n_in && j - i < (size_t)kExtGroupMax) ++j;
        bool grouped = false;
        if (j - i > 1) {
            ExtLaunch Ls[kExtGroupMax];
            for (size_t k = i; k < j; ++k) Ls[k - i] = pending[k].L;
            const hipError_t e = launch_ext_group(Ls, (int)(j - i), pending[i].n_in, batch, c->stream);
            if (e == hipSuccess) {
                grouped = true;
                if (grouped_launches) *grouped_launches += 1;
            }
            else if (e != hipErrorNotSupported) return fail(LR_ERR_HIP, std::string("launch_ext_group: ") + hipGetErrorString(e));
        }
        if (!grouped)
            for (size_t k = i; k < j; ++k) LR_TRY(launch_ext_chunked(c, pending[k].L, pending[k].n_in, batch));
        i = j;
    }
    pending.clear();
    return LR_OK;
}

int run_ext(lr_context *c, const DevModup &m, int n_in, Rows in, int batch, ExtSegment s0, ExtSegment s1,
            const ExtSegment *s2 = nullptr, std::vector<ExtPending> *collect = nullptr, bool inv_top = false) {
    if (n_in < 1 || n_in > 40 || n_in > (int)m.h.Q.size()) return fail(LR_ERR_UNSUPPORTED, "basis extension from 1..40 limbs");
    ExtLaunch L;
    L.t = m.tables();
    L.in = in.base;
    L.in_stride = in.stride;
    L.in_limb0 = in.limb0;
    L.n = (int)c->h.N;
    L.seg[0] = s0;
    L.seg[1] = s1;
    L.seg[2] = s2 ? *s2 : segment(nullptr, 0, 0, 0, 0);
    L.inv_top = 0;
    if (inv_top) {
        if (!L.seg[0].top_tw || !L.t.invtop0 || !L.t.invtop1) return fail(LR_ERR_INTERNAL, "lazy inverse input without the top-stage extension");
        L.inv_top = 1;
    }
    if (collect) {
        collect->push_back(ExtPending{L, n_in});
        return LR_OK;
    }
    return launch_ext_chunked(c, L, n_in, batch);
}

ExtSegment segment(u64 *out, long long stride, int limb0, int col0, int count) {
    ExtSegment s;
    s.out = out;
    s.stride = stride;
    s.limb0 = limb0;
    s.col0 = col0;
    s.count = count;
    s.top_tw = nullptr;
    s.top_mod0 = 0;
    s.epi_mode = 0;
    s.epi_x = nullptr;
    s.epi_x_stride = 0;
    s.epi_c = s.epi_s = nullptr;
    return s;
}

int run_submul(lr_context *c, int limbs, int batch, const u64 *a, long long a_stride, const u64 *b, long long b_stride,
               long long b_row_stride, u64 *out, long long out_stride, const u64 *d_consts, bool reduce_b,
               const LimbScalars *addend, const u64 *plus = nullptr, long long plus_stride = 0, const LimbScalars *post = nullptr,
               int limb0 = 0) {
    // limb0 > 0: the launch covers the limbs limb0 .. limb0 + limbs - 1; the row pointers (a, b, out, plus) and d_consts are
    // passed already advanced to that limb, the modulus table is advanced here (addend / post are not supported then)
    SubMulLaunch L;
    L.plus = plus;
    L.plus_stride = plus_stride;
    L.has_post = post ? 1 : 0;
    if (post) L.post = *post;
    else std::memset(&L.post, 0, sizeof(L.post));
    L.a = a;
    L.b = b;
    L.out = out;
    L.a_stride = a_stride;
    L.b_stride = b_stride;
    L.out_stride = out_stride;
    L.b_row_stride = b_row_stride;
    L.n = (int)c->h.N;
    L.lp = c->d_lp + limb0;
    L.consts = d_consts;
    L.reduce_b = reduce_b ? 1 : 0;
    if (addend) L.addend = *addend;
    else std::memset(&L.addend, 0, sizeof(L.addend));
    LR_HIP(launch_submul(L, limbs, batch, c->stream));
    return LR_OK;
}

int same_degree(const lr_context *a, const lr_context *b) {
    if (a->h.N != b->h.N) return fail(LR_ERR_SHAPE, "contexts have different ring degrees");
    if (a->device != b->device) return fail(LR_ERR_ARG, "contexts live on different devices");
    return LR_OK;
}

// The pipelines interleave launches of contextQ and contextP; both must be on ONE stream or the kernels race.
// (lr_context_set_stream changes one context: call it on both, or on neither.)
int same_stream(const lr_context *a, const lr_context *b) {
    if (a->stream != b->stream)
        return fail(LR_ERR_ARG, "the contexts of this handle run on different streams: call lr_context_set_stream on both");
    return LR_OK;
}

}  // namespace

extern "C" int lr_bext_create(lr_context *cQ, lr_context *cP, lr_bext **out) {
    return guarded([&]() -> int {
    if (!cQ || !cP || !out) return fail(LR_ERR_ARG, "null argument");
    *out = nullptr;
    LR_TRY(same_degree(cQ, cP));
    LR_HIP(hipSetDevice(cQ->device));
    std::unique_ptr<lr_bext> b(new lr_bext());
    b->cQ = cQ;
    b->cP = cP;
    b->device = cQ->device;
    Options o = cQ->opt;         // the extender takes its options from its first context (+ the test-only override, as at every creation)
    o.apply_env();
    LR_TRY(b->qp.init(cQ->h.q, cP->h.q, o.ext_narrow, o.ext_ieee_div));
    LR_TRY(b->pq.init(cP->h.q, cQ->h.q, o.ext_narrow, o.ext_ieee_div));
    LR_TRY(b->pq.set_inverse_top(cP->h, 0));
    b->moddown_pq = build_moddown(cQ->h, cP->h);  // genModDownParams(contextQ, contextP), ring_basis_extension.go:66
    b->moddown_qp = build_moddown(cP->h, cQ->h);  // :67
    LR_TRY(to_device(&b->d_moddown_pq, b->moddown_pq.data(), b->moddown_pq.size()));
    LR_TRY(to_device(&b->d_moddown_qp, b->moddown_qp.data(), b->moddown_qp.size()));
    {
        std::vector<EpiLimb> ec(cQ->h.L());
        for (int i = 0; i < cQ->h.L(); ++i) {
            const u64 q = cQ->h.q[i], cc = inv_mform(b->moddown_pq[i], q, cQ->h.mred[i]);
            ec[i] = make_epi_limb(cQ, i, cc);
        }
        LR_TRY(to_device(&b->d_moddown_pq_epi, ec.data(), ec.size()));
    }
    *out = b.release();
    return LR_OK;
    });
}

extern "C" int lr_bext_destroy(lr_bext *b) {
    return guarded([&]() -> int {
    if (!b) return LR_OK;
    (void)hipSetDevice(b->device);
    (void)hipDeviceSynchronize();   // the handle's work may be on its contexts' caller-supplied stream
    delete b;
    return LR_OK;
    });
}

extern "C" int lr_bext_get_table(const lr_bext *b, int which, uint64_t *dst, size_t dst_count) {
    return guarded([&]() -> int {
    if (!b || !dst) return fail(LR_ERR_ARG, "null argument");
    const std::vector<u64> &src = which == 0 ? b->moddown_pq : b->moddown_qp;
    if (which < 0 || which > 1) return fail(LR_ERR_ARG, "unknown table id");
    if (dst_count != src.size()) return fail(LR_ERR_SHAPE, "table size mismatch");
    std::memcpy(dst, src.data(), src.size() * sizeof(u64));
    return LR_OK;
    });
}

extern "C" int lr_modup_split_qp(lr_bext *b, int level, const lr_poly *p1, lr_poly *p2) {
    return guarded([&]() -> int {
    if (!b || !p1 || !p2) return fail(LR_ERR_ARG, "null argument");
    const int nP = b->cP->h.L();
    if (level < 0 || level + 1 > b->cQ->h.L() || level + 1 > p1->limbs || nP > p2->limbs)
        return fail(LR_ERR_SHAPE, "ModUpSplitQP: limb counts");
    if (p1->batch != p2->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(b->cQ->device));
    return run_ext(b->cQ, b->qp, level + 1, rows_of(p1), p2->batch, segment(p2->d, p2->stride(), 0, 0, nP),
                   segment(nullptr, 0, 0, 0, 0));
    });
}

extern "C" int lr_modup_split_pq(lr_bext *b, int level, const lr_poly *p1, lr_poly *p2) {
    return guarded([&]() -> int {
    if (!b || !p1 || !p2) return fail(LR_ERR_ARG, "null argument");
    const int nQ = b->cQ->h.L();
    if (level < 0 || level + 1 > b->cP->h.L() || level + 1 > p1->limbs || nQ > p2->limbs)
        return fail(LR_ERR_SHAPE, "ModUpSplitPQ: limb counts");
    if (p1->batch != p2->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(b->cQ->device));
    return run_ext(b->cQ, b->pq, level + 1, rows_of(p1), p2->batch, segment(p2->d, p2->stride(), 0, 0, nQ),
                   segment(nullptr, 0, 0, 0, 0));
    });
}

namespace {

// shared tail of the four ModDown...PQ variants: P part (coefficient domain, rows p_limb0.. of pP)
// -> poolQ[0..level] by modUpExact, optional NTT, then p2 = MRed(p1Q + (q - pool), P^-1)
int moddown_pq_core(lr_bext *b, int level, const u64 *p1Q, long long p1Q_stride, Rows pP, int batch, lr_poly *p2, bool ntt) {
    lr_context *cQ = b->cQ;
    const int nP = b->cP->h.L();
    const long long pool_stride = (long long)cQ->h.L() * (long long)cQ->h.N;
    if (!ntt && !cQ->opt.no_epilogue && ext_epilogue_supported(b->pq.tables(), nP, (int)cQ->h.N)) {
        // coefficient domain: the subtract-multiply rides in the extension's stores (ExtSegment::epi_mode 1)
        ExtSegment sd = segment(p2->d, p2->stride(), 0, 0, level + 1);
        sd.epi_mode = 1;
        sd.epi_x = p1Q;
        sd.epi_x_stride = p1Q_stride;
        sd.epi_c = b->d_moddown_pq;
        return run_ext(cQ, b->pq, nP, pP, batch, sd, segment(nullptr, 0, 0, 0, 0));
    }
    LR_TRY(b->poolQ.ensure(cQ, (size_t)batch * pool_stride));
    LR_TRY(run_ext(cQ, b->pq, nP, pP, batch, segment(b->poolQ.d, pool_stride, 0, 0, level + 1), segment(nullptr, 0, 0, 0, 0)));
    if (ntt) {
        Rows pr{b->poolQ.d, pool_stride, 0, 1};
        LR_TRY(run_ntt(cQ, false, pr, pr, 0, 1, level + 1, batch));
    }
    return run_submul(cQ, level + 1, batch, p1Q, p1Q_stride, b->poolQ.d, pool_stride, (long long)cQ->h.N, p2->d,
                      p2->stride(), b->d_moddown_pq, false, nullptr);
}

}  // namespace

extern "C" int lr_moddown_ntt_pq(lr_bext *b, int level, lr_poly *p1, lr_poly *p2) {
    return guarded([&]() -> int {
    if (!b || !p1 || !p2) return fail(LR_ERR_ARG, "null argument");
    const int nQ = b->cQ->h.L(), nP = b->cP->h.L();
    if (level < 0 || level + 1 > nQ || p1->limbs < nQ + nP || p2->limbs < level + 1)
        return fail(LR_ERR_SHAPE, "ModDownNTTPQ: limb counts");
    if (p1->batch != p2->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_TRY(same_stream(b->cQ, b->cP));
    LR_HIP(hipSetDevice(b->cQ->device));
    Rows pP = rows_of(p1, nQ, 1);
    LR_TRY(run_ntt(b->cP, true, pP, pP, 0, 1, nP, p1->batch));  // ring_basis_extension.go:172-174
    return moddown_pq_core(b, level, p1->d, p1->stride(), pP, p1->batch, p2, true);
    });
}

extern "C" int lr_moddown_split_ntt_pq(lr_bext *b, int level, const lr_poly *p1Q, lr_poly *p1P, lr_poly *p2) {
    return guarded([&]() -> int {
    if (!b || !p1Q || !p1P || !p2) return fail(LR_ERR_ARG, "null argument");
    const int nQ = b->cQ->h.L(), nP = b->cP->h.L();
    if (level < 0 || level + 1 > nQ || p1Q->limbs < level + 1 || p1P->limbs < nP || p2->limbs < level + 1)
        return fail(LR_ERR_SHAPE, "ModDownSplitedNTTPQ: limb counts");
    if (p1Q->batch != p2->batch || p1P->batch != p2->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_TRY(same_stream(b->cQ, b->cP));
    LR_HIP(hipSetDevice(b->cQ->device));
    Rows pP = rows_of(p1P);
    LR_TRY(run_ntt(b->cP, true, pP, pP, 0, 1, nP, p2->batch));  // :215
    return moddown_pq_core(b, level, p1Q->d, p1Q->stride(), pP, p2->batch, p2, true);
    });
}

extern "C" int lr_moddown_pq(lr_bext *b, int level, const lr_poly *p1, lr_poly *p2) {
    return guarded([&]() -> int {
    if (!b || !p1 || !p2) return fail(LR_ERR_ARG, "null argument");
    const int nQ = b->cQ->h.L(), nP = b->cP->h.L();
    if (level < 0 || level + 1 > nQ || p1->limbs < level + 1 + nP || p2->limbs < level + 1)
        return fail(LR_ERR_SHAPE, "ModDownPQ: limb counts");
    if (p1->batch != p2->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_TRY(same_stream(b->cQ, b->cP));
    LR_HIP(hipSetDevice(b->cQ->device));
    return moddown_pq_core(b, level, p1->d, p1->stride(), rows_of(p1, level + 1, 1), p1->batch, p2, false);
    });
}

extern "C" int lr_moddown_split_pq(lr_bext *b, int level, const lr_poly *p1Q, const lr_poly *p1P, lr_poly *p2) {
    return guarded([&]() -> int {
    if (!b || !p1Q || !p1P || !p2) return fail(LR_ERR_ARG, "null argument");
    const int nQ = b->cQ->h.L(), nP = b->cP->h.L();
    if (level < 0 || level + 1 > nQ || p1Q->limbs < level + 1 || p1P->limbs < nP || p2->limbs < level + 1)
        return fail(LR_ERR_SHAPE, "ModDownSplitedPQ: limb counts");
    if (p1Q->batch != p2->batch || p1P->batch != p2->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_TRY(same_stream(b->cQ, b->cP));
    LR_HIP(hipSetDevice(b->cQ->device));
    return moddown_pq_core(b, level, p1Q->d, p1Q->stride(), rows_of(p1P), p2->batch, p2, false);
    });
}

extern "C" int lr_moddown_split_qp(lr_bext *b, int levelQ, int levelP, const lr_poly *p1Q, const lr_poly *p1P, lr_poly *p2) {
    return guarded([&]() -> int {
    if (!b || !p1Q || !p1P || !p2) return fail(LR_ERR_ARG, "null argument");
    lr_context *cP = b->cP;
    const int nQ = b->cQ->h.L(), nP = cP->h.L();
    if (levelQ < 0 || levelQ + 1 > nQ || levelP < 0 || levelP + 1 > nP || p1Q->limbs < levelQ + 1 ||
        p1P->limbs < levelP + 1 || p2->limbs < levelP + 1)
        return fail(LR_ERR_SHAPE, "ModDownSplitedQP: limb counts");
    if (p1Q->batch != p2->batch || p1P->batch != p2->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_TRY(same_stream(b->cQ, b->cP));
    LR_HIP(hipSetDevice(cP->device));
    const int batch = p2->batch;
    const long long pool_stride = (long long)nP * (long long)cP->h.N;
    if (!cP->opt.no_epilogue && ext_epilogue_supported(b->qp.tables(), levelQ + 1, (int)cP->h.N)) {
        ExtSegment sd = segment(p2->d, p2->stride(), 0, 0, levelP + 1);
        sd.epi_mode = 1;
        sd.epi_x = p1P->d;
        sd.epi_x_stride = p1P->stride();
        sd.epi_c = b->d_moddown_qp;
        return run_ext(b->cQ, b->qp, levelQ + 1, rows_of(p1Q), batch, sd, segment(nullptr, 0, 0, 0, 0));
    }
    LR_TRY(b->poolP.ensure(cP, (size_t)batch * pool_stride));
    // ModUpSplitQP(levelQ, p1Q, polypool), :332
    LR_TRY(run_ext(b->cQ, b->qp, levelQ + 1, rows_of(p1Q), batch, segment(b->poolP.d, pool_stride, 0, 0, nP),
                   segment(nullptr, 0, 0, 0, 0)));
    return run_submul(cP, levelP + 1, batch, p1P->d, p1P->stride(), b->poolP.d, pool_stride, (long long)cP->h.N, p2->d,
                      p2->stride(), b->d_moddown_qp, false, nullptr);
    });
}

// ------------------------------------------------------------------------------------------
// Decomposer
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// SimpleScaler (ring/ring_scaling.go:166-300)
// ------------------------------------------------------------------------------------------
extern "C" int lr_simple_scaler_create(lr_context *c, uint64_t t, lr_simple_scaler **out) {
    return guarded([&]() -> int {
    if (!out) return fail(LR_ERR_ARG, "out is null");
    *out = nullptr;
    if (!c) return fail(LR_ERR_ARG, "null context");
    std::unique_ptr<lr_simple_scaler> s(new (std::nothrow) lr_simple_scaler);
    if (!s) return fail(LR_ERR_ARG, "out of host memory");
    if (!build_simple_scaler(t, c->h.q, s->h)) return fail(LR_ERR_ARG, "t must be non-zero (BRedParams divides by it, ring/modular_reduction.go:97)");
    s->device = c->device;
    s->ctx = c;
    LR_HIP(hipSetDevice(c->device));
    std::vector<double> ti(2 * s->h.ti.size());
    for (size_t i = 0; i < s->h.ti.size(); ++i) {
        ti[2 * i] = s->h.ti[i].hi;
        ti[2 * i + 1] = s->h.ti[i].lo;
    }
    LR_TRY(to_device(&s->d_wi, s->h.wi.data(), s->h.wi.size()));
    LR_TRY(to_device(&s->d_ti, ti.data(), ti.size()));
    *out = s.release();
    return LR_OK;
    });
}

extern "C" int lr_simple_scaler_destroy(lr_simple_scaler *s) {
    return guarded([&]() -> int {
    if (!s) return LR_OK;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();   // the handle's work may be on its contexts' caller-supplied stream
    delete s;
    return LR_OK;
    });
}

extern "C" int lr_simple_scaler_tables(const lr_simple_scaler *s, uint64_t *wi, double *ti, int count) {
    return guarded([&]() -> int {
    if (!s || !wi || !ti) return fail(LR_ERR_ARG, "null argument");
    if (count != (int)s->h.wi.size()) return fail(LR_ERR_SHAPE, "table size mismatch");
    for (int i = 0; i < count; ++i) {
        wi[i] = s->h.wi[i];
        ti[2 * i] = s->h.ti[i].hi;
        ti[2 * i + 1] = s->h.ti[i].lo;
    }
    return LR_OK;
    });
}

extern "C" int lr_simple_scale(lr_simple_scaler *s, const lr_poly *p1, lr_poly *p2) {
    return guarded([&]() -> int {
    if (!s || !p1 || !p2) return fail(LR_ERR_ARG, "null argument");
    lr_context *c = s->ctx;
    if (p1->N != c->h.N || p2->N != c->h.N) return fail(LR_ERR_SHAPE, "ring degree mismatch");
    if (p1->limbs < c->h.L()) return fail(LR_ERR_SHAPE, "p1 must hold every modulus of the scaler's context (index out of range in the reference)");
    if (p1->device != c->device || p2->device != c->device) return fail(LR_ERR_ARG, "poly lives on another device");
    if (p1->batch != p2->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(c->device));
    ScaleLaunch L;
    L.in = p1->d;
    L.out = p2->d;
    L.in_stride = p1->stride();
    L.out_stride = p2->stride();
    L.wi = s->d_wi;
    L.ti = s->d_ti;
    L.t = s->h.t;
    L.add_param = s->h.add_param;
    L.mul_param = s->h.mul_param;
    L.pow2 = s->h.pow2 ? 1 : 0;
    L.limbs_in = c->h.L();
    L.limbs_out = p2->limbs;
    L.n = (int)c->h.N;
    LR_HIP(launch_simple_scale(L, p1->batch, c->stream));
    return LR_OK;
    });
}

extern "C" int lr_decomposer_create(lr_context *cQ, lr_context *cP, lr_decomposer **out) {
    return guarded([&]() -> int {
    if (!cQ || !cP || !out) return fail(LR_ERR_ARG, "null argument");
    *out = nullptr;
    LR_TRY(same_degree(cQ, cP));
    LR_HIP(hipSetDevice(cQ->device));
    std::unique_ptr<lr_decomposer> d(new lr_decomposer());
    d->cQ = cQ;
    d->cP = cP;
    d->device = cQ->device;
    const std::vector<u64> &Q = cQ->h.q, &P = cP->h.q;
    d->nQ = (int)Q.size();
    d->nP = (int)P.size();
    d->alpha = d->nP;
    d->beta = (d->nQ + d->alpha - 1) / d->alpha;  // ceil(len(Q)/alpha), ring_basis_extension.go:433
    d->xalpha.assign(d->beta, d->alpha);
    if (d->nQ % d->alpha != 0) d->xalpha[d->beta - 1] = d->nQ % d->alpha;
    std::vector<u64> QP(Q);
    QP.insert(QP.end(), P.begin(), P.end());
    d->modup.resize(d->beta);
    Options o = cQ->opt;
    o.apply_env();
    const bool narrow = o.ext_narrow;
    for (int i = 0; i < d->beta; ++i) {
        for (int j = 0; j + 1 < d->xalpha[i]; ++j) {
            std::vector<u64> Qi(Q.begin() + (size_t)i * d->alpha, Q.begin() + (size_t)i * d->alpha + j + 2);
            std::unique_ptr<DevModup> m(new DevModup());
            LR_TRY(m->init(Qi, QP, narrow, o.ext_ieee_div));
            LR_TRY(m->set_inverse_top(cQ->h, i * d->alpha));
            d->modup[i].push_back(std::move(m));
        }
    }
    *out = d.release();
    return LR_OK;
    });
}

extern "C" int lr_decomposer_destroy(lr_decomposer *d) {
    return guarded([&]() -> int {
    if (!d) return LR_OK;
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();   // the handle's work may be on its contexts' caller-supplied stream
    delete d;
    return LR_OK;
    });
}

namespace {

// Decompose (split == false, outP ignored) / DecomposeAndSplit.  in: rows of p0 (coefficient domain).
// does digit `crt` at `level` go through the extension kernel (false: the trivial-copy branch, :490-497 / :613-623)?
bool digit_is_extended(const lr_decomposer *d, int level, int crt) {
    const int alphai = d->xalpha[crt];
    const int ed = crt * d->alpha + alphai;
    return !((ed > level + 1 && (level + 1) % d->nP == 1) || alphai == 1);
}

// top: write the first forward stage over index bit logN - 1 instead of the plain extension (N = 2^16 key switch; split form only,
// extended digits only -- the caller checks digit_is_extended and ext_top_supported)
// skip_own: do not write the rows the digit owns (the key switch reads them from the NTT-domain input, or copies them in)
int decompose_core(lr_decomposer *d, int level, int crt, Rows in, int batch, u64 *outQ, long long outQ_stride, u64 *outP,
                   long long outP_stride, bool split, bool top = false, bool skip_own = false, std::vector<ExtPending> *collect = nullptr,
                   bool inv_top = false) {
    lr_context *c = d->cQ;
    if (crt < 0 || crt >= d->beta) return fail(LR_ERR_SHAPE, "crtDecompLevel out of range");
    if (level < 0 || level + 1 > d->nQ) return fail(LR_ERR_SHAPE, "level out of range");
    const int alphai = d->xalpha[crt];
    const int st = crt * d->alpha, ed = st + alphai;
    if (st > level) return fail(LR_ERR_SHAPE, "digit lies above the level");
    const int n = (int)c->h.N;
    if ((ed > level + 1 && (level + 1) % d->nP == 1) || alphai == 1) {
        if (top) return fail(LR_ERR_ARG, "top-stage extension requested for a digit that takes the copy branch");
        // no reconstruction needed: every target limb receives limb p0idxst, :490-497 / :613-623
        RowAddLaunch L;
        L.in = in.base + (long long)(in.limb0 + st) * n;
        L.in_stride = in.stride;
        L.n = n;
        L.q = 0;
        std::memset(&L.adds, 0, sizeof(L.adds));
        L.out = outQ;
        L.out_stride = outQ_stride;
        LR_HIP(launch_rowadd(L, split ? level + 1 : level + 1 + d->nP, batch, c->stream));
        if (split) {
            L.out = outP;
            L.out_stride = outP_stride;
            LR_HIP(launch_rowadd(L, d->nP, batch, c->stream));
        }
        return LR_OK;
    }
    int index;
    if (level >= alphai + crt * d->alpha) index = alphai - 2;
    else index = (level - 1) % d->alpha;
    const DevModup &m = *d->modup[crt][index];
    Rows digit = in;
    digit.limb0 = in.limb0 + st;
    // rows 0..level take table columns 0..level (the own-digit rows are rewritten by the
    // "index greater" loop of the reference, :571 / :687, so the copy at :553 / :669 is dead);
    // the special primes take columns nQ.., written to the P poly or to rows level+1.. of p1.
    ExtSegment sq = segment(outQ, outQ_stride, 0, 0, level + 1);
    ExtSegment sp = split ? segment(outP, outP_stride, 0, d->nQ, d->nP) : segment(outQ, outQ_stride, level + 1, d->nQ, d->nP);
    if (top) {
        if (!split) return fail(LR_ERR_ARG, "top-stage extension: split form only");
        sq.top_tw = d->cQ->d_fwd;      // rows 0..level of the Q part are the context's limbs 0..level
        sp.top_tw = d->cP->d_fwd;
    }
    if (skip_own && split) {
        // rows [st, own_end) are the digit's own: two Q segments around them
        const int own_end = ed > level + 1 ? level + 1 : ed;
        ExtSegment lo = segment(outQ, outQ_stride, 0, 0, st);
        ExtSegment hi = segment(outQ, outQ_stride, own_end, own_end, level + 1 - own_end);
        lo.top_tw = hi.top_tw = sq.top_tw;
        hi.top_mod0 = own_end;
        return run_ext(c, m, index + 2, digit, batch, lo, hi, &sp, collect, inv_top);
    }
    return run_ext(c, m, index + 2, digit, batch, sq, sp, nullptr, collect, inv_top);
}

}  // namespace

extern "C" int lr_decompose(lr_decomposer *d, int level, int crt, const lr_poly *p0, lr_poly *p1) {
    return guarded([&]() -> int {
    if (!d || !p0 || !p1) return fail(LR_ERR_ARG, "null argument");
    if (p0->limbs < level + 1 || p1->limbs < level + 1 + d->nP) return fail(LR_ERR_SHAPE, "Decompose: limb counts");
    if (p0->batch != p1->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(d->cQ->device));
    return decompose_core(d, level, crt, rows_of(p0), p1->batch, p1->d, p1->stride(), nullptr, 0, false);
    });
}

extern "C" int lr_decompose_and_split(lr_decomposer *d, int level, int crt, const lr_poly *p0, lr_poly *p1Q, lr_poly *p1P) {
    return guarded([&]() -> int {
    if (!d || !p0 || !p1Q || !p1P) return fail(LR_ERR_ARG, "null argument");
    if (p0->limbs < level + 1 || p1Q->limbs < level + 1 || p1P->limbs < d->nP)
        return fail(LR_ERR_SHAPE, "DecomposeAndSplit: limb counts");
    if (p0->batch != p1Q->batch || p0->batch != p1P->batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    LR_HIP(hipSetDevice(d->cQ->device));
    return decompose_core(d, level, crt, rows_of(p0), p1Q->batch, p1Q->d, p1Q->stride(), p1P->d, p1P->stride(), true);
    });
}

// ------------------------------------------------------------------------------------------
// RNS rescale (ring/ring_scaling.go:9-164)
// ------------------------------------------------------------------------------------------
namespace {

int check_rescale(lr_context *c, lr_poly *p0) {
    if (!c || !p0) return fail(LR_ERR_ARG, "null argument");
    if (p0->N != c->h.N) return fail(LR_ERR_SHAPE, "ring degree mismatch");
    if (p0->limbs < 2) return fail(LR_ERR_SHAPE, "cannot divide by the last modulus of a 1-limb polynomial");
    if (p0->limbs > c->h.L()) return fail(LR_ERR_SHAPE, "poly has more limbs than the context has moduli");
    return LR_OK;
}

// round == true adds the pHalf centring of :83-89 / :125-129
int rescale_coeff_domain(lr_context *c, lr_poly *p0, bool round) {
    const int level = p0->limbs - 1, n = (int)c->h.N, batch = p0->batch;
    u64 *last = p0->d + (long long)level * n;
    LimbScalars add;
    std::memset(&add, 0, sizeof(add));
    if (round) {
        const u64 pj = c->h.q[level], phalf = (pj - 1) >> 1;
        RowAddLaunch L;
        L.in = last;
        L.out = last;
        L.in_stride = L.out_stride = p0->stride();
        L.n = n;
        L.q = pj;
        std::memset(&L.adds, 0, sizeof(L.adds));
        L.adds.v[0] = phalf;
        LR_HIP(launch_rowadd(L, 1, batch, c->stream));
        for (int i = 0; i < level; ++i) add.v[i] = c->h.q[i] - bred_add(phalf, c->h.q[i], c->h.bred[i].hi);  // pHalfNegQi
    }
    LR_TRY(run_submul(c, level, batch, p0->d, p0->stride(), last, p0->stride(), 0, p0->d, p0->stride(),
                      c->d_rescale + (size_t)(level - 1) * c->h.L(), true, &add));
    p0->limbs = level;
    return LR_OK;
}

// The rounding variant transforms (t + pHalfNegQi[i]) under modulus i, t = the centred last limb (:101-105).  The transform is
// linear and the addend is the same in every coefficient: NTT_i(t + a_i * ones) = NTT_i(t) + a_i * NTT_i(ones), so the
// polynomial is transformed as it is (one source row for all limbs, like the floor variant) and the constant vector joins
// the subtract-multiply as its `plus` operand, already multiplied by -rescaleParams[i]: the same canonical residue without the
// pass that writes `level` shifted copies of the row.  The table depends on the level only and is built once.
int rescale_round_table(lr_context *c, int level, const u64 **out, const EpiLimb **epi_out) {
    std::lock_guard<std::mutex> lock(c->rescale_mu);
    auto it = c->rescale_round.find(level);
    if (it != c->rescale_round.end()) {
        *out = it->second.plus;
        *epi_out = it->second.epi;
        return LR_OK;
    }
    // built into locals; the cache only ever holds complete tables (a failure below leaves no entry behind)
    const int n = (int)c->h.N;
    const long long words = (long long)level * n;
    struct Guard {
        u64 *table = nullptr;
        EpiLimb *epi = nullptr;
        ~Guard() {
            if (table) (void)hipFree(table);
            if (epi) (void)hipFree(epi);
        }
    } g;
    ScratchLease tmpbuf;
    LR_TRY(tmpbuf.take(&c->scratch, (size_t)words));
    LR_HIP(hipMalloc((void **)&g.table, (size_t)words * sizeof(u64)));
    LR_HIP(hipMemsetAsync(g.table, 0, (size_t)words * sizeof(u64), c->stream));
    const u64 pj = c->h.q[level], phalf = (pj - 1) >> 1;
    RowAddLaunch M;
    M.in = g.table;                     // a row of zeros
    M.in_stride = 0;
    M.out = tmpbuf.d();
    M.out_stride = words;
    M.n = n;
    M.q = 0;
    std::memset(&M.adds, 0, sizeof(M.adds));
    for (int i = 0; i < level; ++i) M.adds.v[i] = c->h.q[i] - bred_add(phalf, c->h.q[i], c->h.bred[i].hi);   // pHalfNegQi
    LR_HIP(launch_rowadd(M, level, 1, c->stream));
    Rows tmp{tmpbuf.d(), words, 0, 1};
    LR_TRY(run_ntt(c, false, tmp, tmp, 0, 1, level, 1));
    // table = MRed(0 + (q - NTT(a_i * ones)), rescaleParams[i])
    LR_TRY(run_submul(c, level, 1, g.table, words, tmpbuf.d(), words, (long long)n, g.table, words,
                      c->d_rescale + (size_t)(level - 1) * c->h.L(), false, nullptr));
    {
        std::vector<EpiLimb> ec(c->h.L());
        for (int i = 0; i < level; ++i) {
            const u64 q = c->h.q[i], cc = inv_mform(c->h.rescale[(size_t)(level - 1) * c->h.L() + i], q, c->h.mred[i]);
            ec[i] = make_epi_limb(c, i, cc);
        }
        LR_TRY(to_device(&g.epi, ec.data(), ec.size()));
    }
    c->rescale_round[level] = lr_context::RoundTable{g.table, g.epi};
    *out = g.table;
    *epi_out = g.epi;
    g.table = nullptr;
    g.epi = nullptr;
    return LR_OK;
}

int rescale_ntt_domain(lr_context *c, lr_poly *p0, bool round) {
    const int level = p0->limbs - 1, n = (int)c->h.N, batch = p0->batch;
    const long long tmp_stride = (long long)level * n;
    const u64 *plus = nullptr;
    const EpiLimb *ec = nullptr;
    if (round && !c->opt.rescale_unfused) LR_TRY(rescale_round_table(c, level, &plus, &ec));
    ScratchLease scratch;
    LR_TRY(scratch.take(&c->scratch, (size_t)batch * tmp_stride));
    Rows last{p0->d, p0->stride(), level, 0};
    // N = 2^15, a small launch whose every target limb takes the epilogue: the last limb's inverse sub-blocks stay lazy and ONE streaming
    // kernel does what lies between them and the targets' forward sub-blocks (last inverse stage + scaling, + pHalf, forward top stage)
    bool fuse_mid = round && plus && ntt_epilogue_ok(c) && c->h.logN == 15 && !c->opt.no_invtop && c->asm_inv >= 0 &&
                    ntt_split15(c, (long long)level * batch);
    for (int l = 0; l < level && fuse_mid; ++l) fuse_mid = ntt_epilogue_limb(c, l);
    LR_TRY(run_ntt(c, true, last, last, level, 0, 1, batch, 0, 0, nullptr, false, fuse_mid));  // :15 / :80
    Rows tmp{scratch.d(), tmp_stride, 0, 1};
    if (round && !fuse_mid) {
        const u64 pj = c->h.q[level], phalf = (pj - 1) >> 1;
        RowAddLaunch L;
        L.in = p0->d + (long long)level * n;
        L.out = p0->d + (long long)level * n;
        L.in_stride = L.out_stride = p0->stride();
        L.n = n;
        L.q = pj;
        std::memset(&L.adds, 0, sizeof(L.adds));
        L.adds.v[0] = phalf;
        LR_HIP(launch_rowadd(L, 1, batch, c->stream));            // :87-89
    }
    if (round && plus && ntt_epilogue_ok(c)) {
        // (x - NTT_i(t)) * rescaleParams[i] + plus inside the forward transform's copy-out for the runs of limbs below 2^46
        const long long n64 = (long long)n;
        int l0 = 0;
        while (l0 < level) {
            const bool fpc = ntt_epilogue_limb(c, l0);
            int l1 = l0 + 1;
            while (l1 < level && ntt_epilogue_limb(c, l1) == fpc) ++l1;
            if (fpc && ntt_split15(c, (long long)(l1 - l0) * batch)) {
                // N = 2^15, a small launch: the transforms with the epilogue on two workgroups each (2^14 sub-blocks).  Every target
                // limb has its own top-stage twiddle, so the stage over bit 14 goes to the scratch rows first (the streaming kernel,
                // the last limb's row broadcast to one row per target limb); the sub-blocks read those and write p0's rows.
                NttLaunch t;
                std::memset(&t, 0, sizeof t);
                t.in = p0->d;
                t.in_poly_stride = p0->stride();
                t.in_limb0 = level;
                t.in_limb_step = 0;
                t.out = scratch.d();
                t.out_poly_stride = tmp_stride;
                t.out_limb0 = l0;
                t.out_limb_step = 1;
                t.mod0 = l0;
                t.mod_step = 1;
                t.n_items = l1 - l0;
                t.batch = batch;
                t.lp = c->d_lp;
                t.tw = c->d_fwd;
                if (fuse_mid) LR_HIP(launch_rescale_mid(t, c->d_inv, level, (c->h.q[level] - 1) >> 1, 15, stream_of(c)));
                else LR_HIP(launch_ntt_top(t, 0, stream_of(c), 15));
                const NttEpilogue ep{p0->d, p0->stride(), plus, 0, ec};
                Rows src{scratch.d(), tmp_stride, l0, 1}, dst{p0->d, p0->stride(), l0, 1};
                LR_TRY(run_ntt(c, false, src, dst, l0, 1, l1 - l0, batch, 0, 0, &ep, true));
            } else if (fpc) {
                const NttEpilogue ep{p0->d, p0->stride(), plus, 0, ec};
                Rows dst{p0->d, p0->stride(), l0, 1};
                LR_TRY(run_ntt(c, false, last, dst, l0, 1, l1 - l0, batch, 0, 0, &ep));
            } else {
                Rows dst{scratch.d(), tmp_stride, l0, 1};
                LR_TRY(run_ntt(c, false, last, dst, l0, 1, l1 - l0, batch));
                LR_TRY(run_submul(c, l1 - l0, batch, p0->d + l0 * n64, p0->stride(), scratch.d() + l0 * n64, tmp_stride, n64,
                                  p0->d + l0 * n64, p0->stride(), c->d_rescale + (size_t)(level - 1) * c->h.L() + l0, false, nullptr,
                                  plus + l0 * n64, 0, nullptr, l0));
            }
            l0 = l1;
        }
        p0->limbs = level;
        return LR_OK;
    }
    if (round && plus) {
        LR_TRY(run_ntt(c, false, last, tmp, 0, 1, level, batch));  // NTT_i(t); the shift by pHalfNegQi[i] rides in `plus`
    } else if (round) {
        const u64 pj = c->h.q[level], phalf = (pj - 1) >> 1;
        RowAddLaunch M;
        M.in = p0->d + (long long)level * n;
        M.in_stride = p0->stride();
        M.out = scratch.d();
        M.out_stride = tmp_stride;
        M.n = n;
        M.q = 0;
        std::memset(&M.adds, 0, sizeof(M.adds));
        for (int i = 0; i < level; ++i) M.adds.v[i] = c->h.q[i] - bred_add(phalf, c->h.q[i], c->h.bred[i].hi);
        LR_HIP(launch_rowadd(M, level, batch, c->stream));        // :101-103
        LR_TRY(run_ntt(c, false, tmp, tmp, 0, 1, level, batch));  // :105
    } else {
        LR_TRY(run_ntt(c, false, last, tmp, 0, 1, level, batch));  // :19: NTT of the last limb under modulus i
    }
    LR_TRY(run_submul(c, level, batch, p0->d, p0->stride(), scratch.d(), tmp_stride, (long long)n, p0->d, p0->stride(),
                      c->d_rescale + (size_t)(level - 1) * c->h.L(), false, nullptr, plus, 0));
    p0->limbs = level;
    return LR_OK;
}

}  // namespace

extern "C" int lr_div_floor_by_last_modulus_ntt(lr_context *c, lr_poly *p0) {
    return guarded([&]() -> int {
    LR_TRY(check_rescale(c, p0));
    LR_HIP(hipSetDevice(c->device));
    return rescale_ntt_domain(c, p0, false);
    });
}
extern "C" int lr_div_floor_by_last_modulus(lr_context *c, lr_poly *p0) {
    return guarded([&]() -> int {
    LR_TRY(check_rescale(c, p0));
    LR_HIP(hipSetDevice(c->device));
    return rescale_coeff_domain(c, p0, false);
    });
}
extern "C" int lr_div_round_by_last_modulus_ntt(lr_context *c, lr_poly *p0) {
    return guarded([&]() -> int {
    LR_TRY(check_rescale(c, p0));
    LR_HIP(hipSetDevice(c->device));
    return rescale_ntt_domain(c, p0, true);
    });
}
extern "C" int lr_div_round_by_last_modulus(lr_context *c, lr_poly *p0) {
    return guarded([&]() -> int {
    LR_TRY(check_rescale(c, p0));
    LR_HIP(hipSetDevice(c->device));
    return rescale_coeff_domain(c, p0, true);
    });
}

static int rescale_many(lr_context *c, lr_poly *p0, int nb, int ntt_domain, bool round) {
    LR_TRY(check_rescale(c, p0));
    if (nb < 0 || nb >= p0->limbs) return fail(LR_ERR_SHAPE, "nbRescales must be below the limb count");
    LR_HIP(hipSetDevice(c->device));
    Rows r = rows_of(p0);
    if (ntt_domain) LR_TRY(run_ntt(c, true, r, r, 0, 1, p0->limbs, p0->batch));   // :59 / :154
    for (int k = 0; k < nb; ++k) LR_TRY(rescale_coeff_domain(c, p0, round));
    if (ntt_domain) LR_TRY(run_ntt(c, false, r, r, 0, 1, p0->limbs, p0->batch));  // :61 / :156
    return LR_OK;
}
extern "C" int lr_div_floor_by_last_modulus_many(lr_context *c, lr_poly *p0, int nb, int ntt_domain) {
    return guarded([&]() -> int {
    return rescale_many(c, p0, nb, ntt_domain, false);
    });
}
extern "C" int lr_div_round_by_last_modulus_many(lr_context *c, lr_poly *p0, int nb, int ntt_domain) {
    return guarded([&]() -> int {
    return rescale_many(c, p0, nb, ntt_domain, true);
    });
}

// ------------------------------------------------------------------------------------------
// ckks.Evaluator call sequences
// ------------------------------------------------------------------------------------------
namespace {
// live plans per device that are not lanes of a batcher: one = a lone evaluator, whose small launches may run side by side (PlanFork)
std::atomic<int> &standalone_plans(int device) {
    static std::atomic<int> counts[64];
    return counts[device >= 0 && device < 64 ? device : 0];
}
}  // namespace

extern "C" int lr_ckks_plan_create(lr_context *cQ, lr_context *cP, int max_batch, lr_ckks_plan **out) {
    return lr_ckks_plan_create_ex(cQ, cP, max_batch, nullptr, out);
}

extern "C" int lr_ckks_plan_create_ex(lr_context *cQ, lr_context *cP, int max_batch, const lr_options *options, lr_ckks_plan **out) {
    return guarded([&]() -> int {
    if (!cQ || !cP || !out) return fail(LR_ERR_ARG, "null argument");
    *out = nullptr;
    Options parsed;
    LR_TRY(options_from_public(options, &parsed));
    if (max_batch < 1) return fail(LR_ERR_ARG, "max_batch must be >= 1");
    LR_TRY(same_degree(cQ, cP));
    std::unique_ptr<lr_ckks_plan> p(new lr_ckks_plan());
    p->cQ = cQ;
    p->cP = cP;
    p->device = cQ->device;
    p->max_batch = max_batch;
    p->opt = parsed;
    LR_TRY(lr_bext_create(cQ, cP, &p->bext));
    int rc = lr_decomposer_create(cQ, cP, &p->dec);
    if (rc != LR_OK) {
        lr_bext_destroy(p->bext);
        return rc;
    }
    standalone_plans(p->device).fetch_add(1);
    *out = p.release();
    return LR_OK;
    });
}

extern "C" int lr_ckks_plan_stats(const lr_ckks_plan *p, uint64_t *forks, uint64_t *grouped_extensions) {
    return guarded([&]() -> int {
    if (!p) return fail(LR_ERR_ARG, "null plan");
    if (forks) *forks = p->forks;
    if (grouped_extensions) *grouped_extensions = p->grouped_ext;
    return LR_OK;
    });
}

extern "C" int lr_ckks_plan_destroy(lr_ckks_plan *p) {
    return guarded([&]() -> int {
    if (!p) return LR_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();   // the handle's work may be on its contexts' caller-supplied stream
    lr_bext_destroy(p->bext);
    lr_decomposer_destroy(p->dec);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->aux) (void)hipStreamDestroy(p->aux);
    if (!p->lane_of) standalone_plans(p->device).fetch_sub(1);
    delete p;
    return LR_OK;
    });
}

namespace {

// Two independent launches of one pipeline side by side: between the constructor and join() the calling thread's forward transforms go
// to the plan's auxiliary stream, which starts behind everything enqueued on the contexts' stream so far; join() makes the contexts'
// stream wait for them.  Worth it only on an otherwise idle device and while the forked launch is far from filling it
// (Options::fork_below_workgroups, 256).  "Otherwise idle" is a structural test, not a momentary one: the plan is the only one alive on its device that is not a
// batcher's lane -- the lone evaluator, for whom latency is what there is.
// Tried and dropped (profiles/r03/fork_policies.txt): forking whenever the launch is small (sixteen threads with a plan each lose a
// quarter of their rate), counting the calls being enqueued at the moment (the count is below the threads most of the time), auxiliary
// streams shared between plans (unrelated pipelines queue behind each other's fork events), an auxiliary stream created with every
// plan (twice the streams on the runtime's four hardware queues: slower without a single fork), lanes of a batcher that fork while
// they are the only lane running (13.4 k products/s against 12.5 k from a C++ host at sixteen callers, 10.5 k against 13.6 k from
// Python threads: the extra streams share hardware queues with the lanes' own, see GPU_MAX_HW_QUEUES in DESIGN 9).
// Capturable: the auxiliary stream joins the capture at the fork and leaves it at the join (it is created by the first fork, i.e. in
// the warm-up call the capture contract asks for).
struct PlanFork {
    lr_ckks_plan *pl;
    bool on = false;
    int rc = LR_OK;
    PlanFork(lr_ckks_plan *p, int workgroups) : pl(p) {
        if (pl->opt.no_fork || pl->fork_failed || g_fork_stream) return;
        // ... and only where one workgroup of the forked launch runs long enough to pay for the two stream hand-overs (~ 19 us): the
        // 2^15 sub-blocks of N = 2^16 (42 us).  Since small 2^15 launches run on 2^14 sub-blocks (20 us, like the 2^14 kernels) a fork
        // there costs more than it hides: PN15QP880 batch 1 4.64 k products/s forked, 5.06 k in order; PN14QP438 6.42 k / 7.30 k;
        // PN16QP1761 1.69 k / 1.64 k (profiles/r03/fork_policies.txt).
        if (pl->cQ->h.logN != 16) return;
        if (pl->lane_of || standalone_plans(pl->device).load(std::memory_order_relaxed) != 1 || workgroups >= pl->opt.fork_below_workgroups) return;
        if (!pl->aux) {
            if (create_stream(&pl->aux, 1) != hipSuccess ||
                hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                pl->fork_failed = true;   // the pipelines stay in order on one stream
                return;
            }
        }
        hipError_t e = hipEventRecord(pl->ev_fork, pl->cQ->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(pl->aux, pl->ev_fork, 0);
        if (e != hipSuccess) {
            rc = fail(LR_ERR_HIP, std::string("fork: ") + hipGetErrorString(e));
            return;
        }
        on = true;
        pl->forks += 1;
        g_fork_stream = pl->aux;
    }
    // the launches that follow go to the contexts' stream again (and run beside the forked ones until join())
    void back() {
        if (on) g_fork_stream = nullptr;
    }
    int join() {
        if (!on) return LR_OK;
        on = false;
        g_fork_stream = nullptr;
        hipError_t e = hipEventRecord(pl->ev_join, pl->aux);
        if (e == hipSuccess) e = hipStreamWaitEvent(pl->cQ->stream, pl->ev_join, 0);
        if (e != hipSuccess) return fail(LR_ERR_HIP, std::string("join: ") + hipGetErrorString(e));
        return LR_OK;
    }
    ~PlanFork() { (void)join(); }   // error paths: the contexts' stream still waits for whatever was forked
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

int run_permute_ntt(lr_context *c, int limbs, int batch, const u64 *in, long long in_stride, u64 *out, long long out_stride,
                    u64 gen, const u64 *const *in_table = nullptr) {
    GaloisLaunch L;
    L.in_table = in_table;
    L.in = in;
    L.out = out;
    L.in_stride = in_stride;
    L.out_stride = out_stride;
    L.n = (int)c->h.N;
    L.logn = (int)c->h.logN;
    L.ntt_domain = 1;
    L.gen = gen & ((c->h.N << 1) - 1);
    L.lp = c->d_lp;
    LR_HIP(launch_permute(L, limbs, batch, c->stream));
    return LR_OK;
}

// Digit decomposition of switchKeysInPlace / RotateHoisted (ckks/evaluator.go:1503-1510, 1258-1272, 1561-1591):
// pl->c2QiQ = [beta][batch][|Q|][N], pl->c2QiP = [beta][batch][|P|][N], both in the NTT domain.  The limbs a digit
// owns are the NTT-domain input itself; they are copied into the digit only when `copy_own` (the hoisted path
// permutes whole digits), otherwise the inner product reads them in place.
// coeff_input (bfv.switchKeys, bfv/evaluator.go:736-770): cx is in the coefficient domain -- the digits are decomposed from cx itself
// and the digits' own limbs are NTT(cx) (:753, kept in pl->c2); otherwise (ckks) cx is in the NTT domain, the digits come from
// InvNTT(cx) and the own limbs are cx.
int ks_decompose(lr_ckks_plan *pl, int level, int batch, const u64 *cx, long long cx_stride, bool copy_own, bool coeff_input = false) {
    lr_context *cQ = pl->cQ, *cP = pl->cP;
    lr_decomposer *dec = pl->dec;
    const int nQ = cQ->h.L(), nP = cP->h.L(), n = (int)cQ->h.N;
    const int alpha = dec->alpha;
    const int beta = (level + 1 + alpha - 1) / alpha;  // :1508
    const long long sQ = (long long)nQ * n, sP = (long long)nP * n;
    const long long dQ = (long long)batch * sQ, dP = (long long)batch * sP;
    LR_TRY(pl->c2QiQ.ensure(cQ, (size_t)beta * dQ));
    LR_TRY(pl->c2.ensure(cQ, (size_t)batch * sQ));
    LR_TRY(pl->c2QiP.ensure(cQ, (size_t)beta * dP));
    // N = 2^16: a forward transform whose input and output rows are disjoint computes its top stage while loading (one
    // launch); in place it needs a separate streaming pass first.  The extensions therefore land in staging buffers of the
    // same shape and the transforms write the pools the consumers read.
    const bool asm16 = cQ->h.logN == 16 && cQ->use_asm && cQ->asm_fwd >= 0 && cP->use_asm && cP->asm_fwd >= 0;
    // N = 2^15 and a key switch whose largest transform launch is small: the same arrangement on the 2^14 sub-block kernels (the
    // extension applies the stage over bit 14, run_ntt_launch takes pretop as the decision for the split)
    const bool asm15 = cQ->h.logN == 15 && ntt_split15(cQ, (long long)std::max(1, level + 1 - alpha) * beta * batch) &&
                       ntt_split15(cP, (long long)nP * beta * batch);
    // ... or, better, the extension itself applies the stage over index bit 15 (each of its threads holds the coefficients j and
    // j + N/2) and the plain sub-block kernels transform in place, reading their own half only.  Possible when every digit of
    // this level goes through the sum-form extension kernel (no trivial-copy digit).
    bool exttop = (asm16 || asm15) && !pl->opt.no_exttop;
    for (int i = 0; i < beta && exttop; ++i) {
        if (!digit_is_extended(dec, level, i)) {
            exttop = false;
            break;
        }
        const int alphai = dec->xalpha[i];
        const int index = level >= alphai + i * dec->alpha ? alphai - 2 : (level - 1) % dec->alpha;
        exttop = ext_top_supported(dec->modup[i][index]->tables(), index + 2, n);
    }
    const bool staged = asm16 && !exttop && !pl->opt.no_staging;
    if (staged) {
        LR_TRY(pl->stageQ.ensure(cQ, (size_t)beta * dQ));
        LR_TRY(pl->stageP.ensure(cQ, (size_t)beta * dP));
    }
    u64 *const srcQ = staged ? pl->stageQ.d : pl->c2QiQ.d, *const srcP = staged ? pl->stageP.d : pl->c2QiP.d;
    Rows cxr{const_cast<u64 *>(cx), cx_stride, 0, 1};
    Rows c2r{pl->c2.d, sQ, 0, 1};
    // the digits' extensions apply the top stage of the transforms that follow them (exttop): then they also take the last stage and
    // the scaling of the inverse transform in front of them (its sub-blocks leave the rows lazy; nothing else reads c2 on this path)
    bool invtop = exttop && !coeff_input && !pl->opt.no_invtop && cQ->asm_inv >= 0;
    for (int i = 0; i < beta && invtop; ++i) {
        const int alphai = dec->xalpha[i];
        const int index = level >= alphai + i * dec->alpha ? alphai - 2 : (level - 1) % dec->alpha;
        invtop = dec->modup[i][index]->invtop0 != nullptr && index + 2 <= 8;
    }
    LR_TRY(run_ntt(cQ, !coeff_input, cxr, c2r, 0, 1, level + 1, batch, 0, 0, nullptr, false, invtop));  // ckks :1503 (InvNTT) / bfv :753 (NTT)
    if (coeff_input) {
        // the decomposition reads the caller's coefficient-domain rows; the transformed copy serves the digits' own limbs
        if (copy_own) return fail(LR_ERR_UNSUPPORTED, "coefficient-domain key switch: own limbs are read in place");
        c2r = cxr;
    }
    int full = 0;   // leading digits that own exactly alpha limbs at this level: their transforms share one launch
    std::vector<ExtPending> pending;   // the digits' extensions: independent, same shape -> one grouped launch (copy-branch digits launch at once)
    pending.reserve((size_t)beta);
    for (int i = 0; i < beta; ++i) {
        u64 *dq = pl->c2QiQ.d + (long long)i * dQ;
        // decomposeAndSplitNTT, :1561-1591
        LR_TRY(decompose_core(dec, level, i, c2r, batch, srcQ + (long long)i * dQ, sQ, srcP + (long long)i * dP, sP, true, exttop, true,
                              pl->opt.no_ext_group ? nullptr : &pending, invtop));
        const int d0 = i * alpha;
        int d1 = d0 + dec->xalpha[i];
        if (d1 > level + 1) d1 = level + 1;
        if (copy_own)   // :1579-1584
            LR_TRY(run_ewise(cQ, LR_COPY, d1 - d0, batch, cx + (long long)d0 * n, cx_stride, nullptr, 0, dq + (long long)d0 * n,
                             sQ, nullptr, d0));
        if (d1 - d0 == alpha && full == i) ++full;
    }
    LR_TRY(flush_ext(cQ, pending, batch, &pl->grouped_ext));
    // the digits' P rows beside their Q rows (another kernel variant, so another launch: at a small batch each fills a fraction of the chip)
    PlanFork forkP(pl, nP * beta * batch);
    LR_TRY(forkP.rc);
    auto partial_digits = [&]() -> int {   // the digits that own fewer than alpha limbs at this level: their own launches
        for (int i = full; i < beta; ++i) {
            u64 *dq = pl->c2QiQ.d + (long long)i * dQ, *sq = srcQ + (long long)i * dQ;
            const int d0 = i * alpha;
            int d1 = d0 + dec->xalpha[i];
            if (d1 > level + 1) d1 = level + 1;
            Rows lo{dq, sQ, 0, 1}, lo_in{sq, sQ, 0, 1};
            LR_TRY(run_ntt(cQ, false, lo_in, lo, 0, 1, d0, batch, 0, 0, nullptr, exttop));                  // limbs below the digit
            Rows hi{dq, sQ, d1, 1}, hi_in{sq, sQ, d1, 1};
            LR_TRY(run_ntt(cQ, false, hi_in, hi, d1, 1, level + 1 - d1, batch, 0, 0, nullptr, exttop));     // limbs above the digit
        }
        return LR_OK;
    };
    if (forkP.on) {
        // beside the full digits' grouped launch: the P rows and the partial digits' Q rows (PN16QP1761, one ciphertext: 70 + 34 us
        // next to 99 us)
        Rows pr{pl->c2QiP.d, sP, 0, 1}, pr_in{srcP, sP, 0, 1};                   // :1590, every digit's P rows
        LR_TRY(run_ntt(cP, false, pr_in, pr, 0, 1, nP, beta * batch, 0, 0, nullptr, exttop));
        LR_TRY(partial_digits());
        forkP.back();
    }
    const bool p_rows_done = forkP.on;
    if (full > 0 && level + 1 - alpha > 0) {
        // limbs outside each digit's own block, all full digits at once (grid z = digit)
        Rows in{srcQ, sQ, 0, 1}, all{pl->c2QiQ.d, sQ, 0, 1};
        LR_TRY(run_ntt(cQ, false, in, all, 0, 1, level + 1 - alpha, full * batch, alpha, batch, nullptr, exttop));
    }
    for (int i = p_rows_done ? beta : full; i < beta; ++i) {
        u64 *dq = pl->c2QiQ.d + (long long)i * dQ, *sq = srcQ + (long long)i * dQ;
        const int d0 = i * alpha;
        int d1 = d0 + dec->xalpha[i];
        if (d1 > level + 1) d1 = level + 1;
        Rows lo{dq, sQ, 0, 1}, lo_in{sq, sQ, 0, 1};
        LR_TRY(run_ntt(cQ, false, lo_in, lo, 0, 1, d0, batch, 0, 0, nullptr, exttop));                  // limbs below the digit
        Rows hi{dq, sQ, d1, 1}, hi_in{sq, sQ, d1, 1};
        LR_TRY(run_ntt(cQ, false, hi_in, hi, d1, 1, level + 1 - d1, batch, 0, 0, nullptr, exttop));     // limbs above the digit
    }
    if (!p_rows_done) {
        Rows pr{pl->c2QiP.d, sP, 0, 1}, pr_in{srcP, sP, 0, 1};                   // :1590, every digit's P rows
        LR_TRY(run_ntt(cP, false, pr_in, pr, 0, 1, nP, beta * batch, 0, 0, nullptr, exttop));
    }
    return forkP.join();
}

// exact 128-bit sums in the key inner product: beta products below q^2 each must stay below q * 2^64
bool keymac_wide_ok(const lr_ckks_plan *pl, const lr_context *c, int beta) {
    if (pl->opt.keymac_narrow) return false;
    u64 qmax = 0;
    for (u64 q : c->h.q) qmax = q > qmax ? q : qmax;
    return (u128)qmax * (u128)beta < ((u128)1 << 64);
}

// Inner product of the digits with a switching key and the two ModDownSplitedNTTPQ (:1511-1557 / :1339-1387).
// digQ/digP: [beta][batch][|Q| resp. |P|][N]; own/own_stride: where the digits' own limbs live when they were not
// copied (nullptr: inside digQ).
int ks_accumulate(lr_ckks_plan *pl, int level, int batch, const u64 *digQ, const u64 *digP, const u64 *own, long long own_stride,
                  const lr_poly *evk, u64 *p0, long long p0_stride, u64 *p1, long long p1_stride, const KeySwitchEpilogue *fin,
                  bool coeff_out = false) {
    lr_context *cQ = pl->cQ, *cP = pl->cP;
    const int nQ = cQ->h.L(), nP = cP->h.L(), n = (int)cQ->h.N;
    const int alpha = pl->dec->alpha;
    const int beta = (level + 1 + alpha - 1) / alpha;
    if (evk->batch < 2 * beta || evk->limbs < nQ + nP) return fail(LR_ERR_SHAPE, "evaluation key: need batch >= 2*beta and |Q|+|P| limbs");
    const long long sQ = (long long)nQ * n, sP = (long long)nP * n;
    const long long dQ = (long long)batch * sQ, dP = (long long)batch * sP;
    LR_TRY(pl->poolPP.ensure(cQ, (size_t)2 * batch * sP));   // P parts of both accumulators, [2][batch][|P|][N]
    u64 *const pool2P = pl->poolPP.d, *const pool3P = pl->poolPP.d + (long long)batch * sP;
    // sum over the digits of evakey[i][0/1] (*) c2_i, canonical, Q part then P part
    {
        KeyMacLaunch K;
        K.tile8 = 0;
        K.wide = 0;
        K.key = evk->d;
        K.key_poly_stride = evk->stride();
        K.n = n;
        K.beta = beta;
        K.c2 = digQ;
        K.c2_digit_stride = dQ;
        K.c2_poly_stride = sQ;
        K.key_limb0 = 0;
        K.out0 = p0;
        K.out1 = p1;
        K.out_stride = p0_stride;
        K.out1_stride = p1_stride;
        K.lp = cQ->d_lp;
        K.wide = keymac_wide_ok(pl, cQ, beta) ? 1 : 0;
        K.own = own;
        K.own_stride = own_stride;
        K.alpha = own ? alpha : 0;
        const KeyMacLaunch KQ = K;
        K.c2 = digP;
        K.c2_digit_stride = dP;
        K.c2_poly_stride = sP;
        K.key_limb0 = nQ;
        K.out0 = pool2P;
        K.out1 = pool3P;
        K.out_stride = sP;
        K.out1_stride = sP;
        K.lp = cP->d_lp;
        K.wide = keymac_wide_ok(pl, cP, beta) ? 1 : 0;
        K.own = nullptr;
        K.own_stride = 0;
        K.alpha = 0;
        // a small batch: the Q part and the P part as one launch (they share nothing and each is a few hundred workgroups)
        hipError_t pe = hipErrorNotSupported;
        if (!pl->opt.no_pair && (long long)batch * (level + 1) <= pl->opt.pair_max_workgroups) pe = launch_keymac_pair(KQ, level + 1, K, nP, batch, cQ->stream);
        if (pe == hipErrorNotSupported) {
            LR_HIP(launch_keymac(KQ, level + 1, batch, cQ->stream));
            LR_HIP(launch_keymac(K, nP, batch, cQ->stream));
        } else if (pe != hipSuccess) {
            return fail(LR_ERR_HIP, std::string("launch_keymac_pair: ") + hipGetErrorString(pe));
        }
    }
    lr_bext *bx = pl->bext;
    if (coeff_out) {
        // bfv.switchKeys' tail (bfv/evaluator.go:806-811): InvNTT over Q||P, then ModDownPQ in the coefficient domain (in place:
        // the extension reads x where it stores, ExtSegment::epi_mode 1)
        if (fin) return fail(LR_ERR_ARG, "coefficient-domain key switch: no epilogue");
        Rows q0r{p0, p0_stride, 0, 1}, q1r{p1, p1_stride, 0, 1}, pr{pool2P, sP, 0, 1};
        // the two accumulators as ONE batch where base + p * stride reaches both: laid out back to back (the relinearisation's pool), or
        // one poly each at any distance (see lr_ckks_rescale)
        auto words = [](const u64 *a, const u64 *b) { return (long long)(((intptr_t)b - (intptr_t)a) / (intptr_t)sizeof(u64)); };
        const bool back_to_back = p0_stride == p1_stride && p1 == p0 + (long long)batch * p0_stride;
        const bool one_each = batch == 1 && p0 != p1;
        const bool pair = !pl->opt.no_pair && (back_to_back || one_each);
        const long long pair_stride = back_to_back ? p0_stride : words(p0, p1);
        if (pair) {
            Rows qr{p0, pair_stride, 0, 1};
            LR_TRY(run_ntt(cQ, true, qr, qr, 0, 1, level + 1, 2 * batch));
        } else {
            LR_TRY(run_ntt(cQ, true, q0r, q0r, 0, 1, level + 1, batch));
            LR_TRY(run_ntt(cQ, true, q1r, q1r, 0, 1, level + 1, batch));
        }
        LR_TRY(run_ntt(cP, true, pr, pr, 0, 1, nP, 2 * batch));
        const bool fused = !cQ->opt.no_epilogue && ext_epilogue_supported(bx->pq.tables(), nP, n);
        if (pair && fused) {
            ExtSegment sd = segment(p0, pair_stride, 0, 0, level + 1);
            sd.epi_mode = 1;
            sd.epi_x = p0;
            sd.epi_x_stride = pair_stride;
            sd.epi_c = bx->d_moddown_pq;
            return run_ext(cQ, bx->pq, nP, pr, 2 * batch, sd, segment(nullptr, 0, 0, 0, 0));      // (pool2P / pool3P lie back to back)
        }
        for (int k = 0; k < 2; ++k) {
            u64 *pq = k == 0 ? p0 : p1;
            const long long pqs = k == 0 ? p0_stride : p1_stride;
            Rows pk{k == 0 ? pool2P : pool3P, sP, 0, 1};
            if (fused) {
                ExtSegment sd = segment(pq, pqs, 0, 0, level + 1);
                sd.epi_mode = 1;
                sd.epi_x = pq;
                sd.epi_x_stride = pqs;
                sd.epi_c = bx->d_moddown_pq;
                LR_TRY(run_ext(cQ, bx->pq, nP, pk, batch, sd, segment(nullptr, 0, 0, 0, 0)));
            } else {
                LR_TRY(bx->poolQ.ensure(cQ, (size_t)batch * sQ));
                LR_TRY(run_ext(cQ, bx->pq, nP, pk, batch, segment(bx->poolQ.d, sQ, 0, 0, level + 1), segment(nullptr, 0, 0, 0, 0)));
                LR_TRY(run_submul(cQ, level + 1, batch, pq, pqs, bx->poolQ.d, sQ, (long long)n, pq, pqs, bx->d_moddown_pq, false, nullptr));
            }
        }
        return LR_OK;
    }
    // ModDownSplitedNTTPQ x2; the two calls share every launch up to the final subtract-multiply
    {
        Rows pr{pool2P, sP, 0, 1};
        LR_TRY(bx->poolQ.ensure(cQ, (size_t)2 * batch * sQ));
        u64 *ext_out = bx->poolQ.d;
        const bool asm16 = cQ->h.logN == 16 && cQ->use_asm && cQ->asm_fwd >= 0;
        const bool asm15 = cQ->h.logN == 15 && ntt_split15(cQ, (long long)(level + 1) * batch);     // (one launch per component)
        const bool exttop = (asm16 || asm15) && !pl->opt.no_exttop && ext_top_supported(bx->pq.tables(), nP, n);
        if (asm16 && !exttop && !pl->opt.no_staging) {
            LR_TRY(pl->stageQ.ensure(cQ, (size_t)2 * batch * sQ));     // (the digits' staging area is free again)
            ext_out = pl->stageQ.d;
        }
        ExtSegment mseg = segment(ext_out, sQ, 0, 0, level + 1);
        if (exttop) mseg.top_tw = cQ->d_fwd;                           // the ModDown transform's top stage inside the extension
        // ... and the last stage of the inverse transform in front of it (see ks_decompose)
        const bool invtop = exttop && !pl->opt.no_invtop && cP->asm_inv >= 0 && bx->pq.invtop0 != nullptr && nP <= 8;
        LR_TRY(run_ntt(cP, true, pr, pr, 0, 1, nP, 2 * batch, 0, 0, nullptr, false, invtop));
        LR_TRY(run_ext(cQ, bx->pq, nP, pr, 2 * batch, mseg, segment(nullptr, 0, 0, 0, 0), nullptr, nullptr, invtop));
        Rows qr{bx->poolQ.d, sQ, 0, 1}, qr_in{ext_out, sQ, 0, 1};
        if (ntt_epilogue_ok(cQ)) {
            // the subtract-multiply and the addition of MulRelin / the rotations inside the forward transform's copy-out, for
            // every run of limbs below 2^46 (FP64 body); the other limbs keep the separate pass
            const long long n64 = (long long)n;
            // without `fin` (plain SwitchKeysInPlace) the results replace p0 / p1 and nothing is added
            u64 *const outs[2] = {fin ? fin->out0 : p0, fin ? fin->out1 : p1};
            const long long out_strides[2] = {fin ? fin->out_stride : p0_stride, fin ? fin->out_stride : p1_stride};
            const u64 *const pluses[2] = {fin ? fin->plus0 : nullptr, fin ? fin->plus1 : nullptr};
            const long long plus_stride = fin ? fin->plus_stride : 0;
            const bool need_zeros = !pluses[0] || !pluses[1];
            if (need_zeros && pl->zerosQ.words < (size_t)sQ) {
                LR_TRY(pl->zerosQ.ensure(cQ, (size_t)sQ));
                LR_HIP(hipMemsetAsync(pl->zerosQ.d, 0, (size_t)sQ * sizeof(u64), cQ->stream));
            }
            int l0 = 0;
            while (l0 <= level) {
                const bool fpc = ntt_epilogue_limb(cQ, l0);
                int l1 = l0 + 1;
                while (l1 <= level && ntt_epilogue_limb(cQ, l1) == fpc) ++l1;
                if (fpc && batch == 1 && !pl->opt.no_pair && outs[0] != outs[1]) {
                    // one ciphertext: the two components as a batch of two whose strides are the distances between their operands
                    // (ext_out holds them back to back; x, plus and the outputs are separate allocations) -- one launch instead of two
                    auto words = [](const u64 *a, const u64 *b) { return (long long)(((intptr_t)b - (intptr_t)a) / (intptr_t)sizeof(u64)); };
                    Rows src{ext_out, sQ, l0, 1};
                    Rows dst{outs[0], words(outs[0], outs[1]), l0, 1};
                    // (a component without an addend -- the rotations' second one -- adds the row of zeros: one more distance)
                    const u64 *plus_a = pluses[0] ? pluses[0] : pl->zerosQ.d, *plus_b = pluses[1] ? pluses[1] : pl->zerosQ.d;
                    const NttEpilogue ep{p0, words(p0, p1), plus_a, words(plus_a, plus_b), bx->d_moddown_pq_epi};
                    LR_TRY(run_ntt(cQ, false, src, dst, l0, 1, l1 - l0, 2, 0, 0, &ep, exttop));
                } else if (fpc) {
                    // the two components are independent launches: side by side while one alone leaves most of the chip idle
                    PlanFork fork1(pl, (l1 - l0) * batch);
                    LR_TRY(fork1.rc);
                    for (int k = 1; k >= 0; --k) {
                        Rows src{ext_out + (long long)k * batch * sQ, sQ, l0, 1};
                        Rows dst{outs[k], out_strides[k], l0, 1};
                        const u64 *plus = pluses[k];
                        const NttEpilogue ep{k == 0 ? p0 : p1, k == 0 ? p0_stride : p1_stride, plus ? plus : pl->zerosQ.d,
                                             plus ? plus_stride : 0, bx->d_moddown_pq_epi};
                        LR_TRY(run_ntt(cQ, false, src, dst, l0, 1, l1 - l0, batch, 0, 0, &ep, exttop));
                        fork1.back();
                    }
                    LR_TRY(fork1.join());
                } else {
                    Rows src{ext_out, sQ, l0, 1}, dst{bx->poolQ.d, sQ, l0, 1};
                    LR_TRY(run_ntt(cQ, false, src, dst, l0, 1, l1 - l0, 2 * batch, 0, 0, nullptr, exttop));
                    for (int k = 0; k < 2; ++k) {
                        const u64 *pq = (k == 0 ? p0 : p1) + l0 * n64;
                        const u64 *ext = bx->poolQ.d + (long long)k * batch * sQ + l0 * n64;
                        const u64 *plus = pluses[k];
                        LR_TRY(run_submul(cQ, l1 - l0, batch, pq, k == 0 ? p0_stride : p1_stride, ext, sQ, n64,
                                          outs[k] + l0 * n64, out_strides[k], bx->d_moddown_pq + l0, false,
                                          nullptr, plus ? plus + l0 * n64 : nullptr, plus_stride, nullptr, l0));
                    }
                }
                l0 = l1;
            }
            return LR_OK;
        }
        LR_TRY(run_ntt(cQ, false, qr_in, qr, 0, 1, level + 1, 2 * batch, 0, 0, nullptr, exttop));
    }
    for (int k = 0; k < 2; ++k) {
        u64 *pq = k == 0 ? p0 : p1;
        const long long pqs = k == 0 ? p0_stride : p1_stride;
        const u64 *ext = bx->poolQ.d + (long long)k * batch * sQ;
        if (fin)
            LR_TRY(run_submul(cQ, level + 1, batch, pq, pqs, ext, sQ, (long long)n, k == 0 ? fin->out0 : fin->out1,
                              fin->out_stride, bx->d_moddown_pq, false, nullptr, k == 0 ? fin->plus0 : fin->plus1, fin->plus_stride));
        else
            LR_TRY(run_submul(cQ, level + 1, batch, pq, pqs, ext, sQ, (long long)n, pq, pqs, bx->d_moddown_pq, false, nullptr));
    }
    return LR_OK;
}

// switchKeysInPlace, ckks/evaluator.go:1475-1558, on raw buffers: cx/p0/p1 have `q_stride` between batch polys
int switch_keys_core(lr_ckks_plan *pl, int level, int batch, const u64 *cx, long long cx_stride, const lr_poly *evk, u64 *p0,
                     long long p0_stride, u64 *p1, long long p1_stride, const KeySwitchEpilogue *fin = nullptr) {
    LR_TRY(ks_decompose(pl, level, batch, cx, cx_stride, false));
    return ks_accumulate(pl, level, batch, pl->c2QiQ.d, pl->c2QiP.d, cx, cx_stride, evk, p0, p0_stride, p1, p1_stride, fin);
}

int check_ct(const lr_ckks_plan *pl, int level, const lr_poly *p, int batch) {
    if (!p) return fail(LR_ERR_ARG, "null poly");
    if (p->N != pl->cQ->h.N) return fail(LR_ERR_SHAPE, "ring degree mismatch");
    if (p->limbs < level + 1) return fail(LR_ERR_SHAPE, "poly has fewer limbs than level+1");
    if (p->batch != batch) return fail(LR_ERR_SHAPE, "batch mismatch");
    return LR_OK;
}

}  // namespace

extern "C" int lr_ckks_switch_keys(lr_ckks_plan *pl, int level, const lr_poly *cx, const lr_poly *evk, lr_poly *p0, lr_poly *p1) {
    return guarded([&]() -> int {
    if (!pl || !cx || !evk || !p0 || !p1) return fail(LR_ERR_ARG, "null argument");
    if (level < 0 || level + 1 > pl->cQ->h.L()) return fail(LR_ERR_SHAPE, "level out of range");
    const int batch = cx->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    LR_TRY(check_ct(pl, level, cx, batch));
    LR_TRY(check_ct(pl, level, p0, batch));
    LR_TRY(check_ct(pl, level, p1, batch));
    LR_TRY(same_stream(pl->cQ, pl->cP));
    LR_HIP(hipSetDevice(pl->cQ->device));
    return switch_keys_core(pl, level, batch, cx->d, cx->stride(), evk, p0->d, p0->stride(), p1->d, p1->stride());
    });
}

// permuteNTT (ckks/evaluator.go:1448-1468): RotateColumns with a specific rotation key / Conjugate.
// gen = the Galois element (ring.PermuteNTTIndex's `gen^power`); the two trailing Context calls (:1466-1467)
// ride on the last ModDown pass.
// bfv.evaluator.switchKeys (bfv/evaluator.go:736-812): cx in the coefficient domain over all of Q, evk over Q||P in the NTT +
// Montgomery domain like the reference's SwitchingKey; p0 / p1 <- the two key-switched polys over Q, coefficient domain.  Same
// machinery as the CKKS key switch (one plan over contextQ / contextP serves both), with the transforms the other way round.
static int bfv_switch_keys_core(lr_ckks_plan *pl, int batch, const u64 *cx, long long cx_stride, const lr_poly *evk, u64 *p0,
                                long long p0_stride, u64 *p1, long long p1_stride) {
    const int level = pl->cQ->h.L() - 1;
    LR_TRY(ks_decompose(pl, level, batch, cx, cx_stride, false, true));
    const long long sQ = (long long)pl->cQ->h.L() * (long long)pl->cQ->h.N;
    return ks_accumulate(pl, level, batch, pl->c2QiQ.d, pl->c2QiP.d, pl->c2.d, sQ, evk, p0, p0_stride, p1, p1_stride, nullptr, true);
}

extern "C" int lr_bfv_switch_keys(lr_ckks_plan *pl, const lr_poly *cx, const lr_poly *evk, lr_poly *p0, lr_poly *p1) {
    return guarded([&]() -> int {
    if (!pl || !cx || !evk || !p0 || !p1) return fail(LR_ERR_ARG, "null argument");
    const int level = pl->cQ->h.L() - 1;
    if (cx == p0 || cx == p1 || p0 == p1) return fail(LR_ERR_ARG, "bfv switch keys: cx, p0 and p1 must be distinct polys");
    LR_TRY(check_ct(pl, level, cx, cx->batch));
    LR_TRY(check_ct(pl, level, p0, cx->batch));
    LR_TRY(check_ct(pl, level, p1, cx->batch));
    if (cx->batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    LR_TRY(same_stream(pl->cQ, pl->cP));
    LR_HIP(hipSetDevice(pl->device));
    return bfv_switch_keys_core(pl, cx->batch, cx->d, cx->stride(), evk, p0->d, p0->stride(), p1->d, p1->stride());
    });
}

// bfv.evaluator.Relinearize on a degree-2 ciphertext (bfv/evaluator.go:480-501, 512-524): out = (c0 + p0, c1 + p1) with
// (p0, p1) = switchKeys(c2, evakey[0]); all polys over Q in the coefficient domain.  out0 / out1 may be c0 / c1.
extern "C" int lr_bfv_relinearize(lr_ckks_plan *pl, const lr_poly *c0, const lr_poly *c1, const lr_poly *c2, const lr_poly *evk,
                                  lr_poly *out0, lr_poly *out1) {
    return guarded([&]() -> int {
    if (!pl || !c0 || !c1 || !c2 || !evk || !out0 || !out1) return fail(LR_ERR_ARG, "null argument");
    lr_context *cQ = pl->cQ;
    const int level = cQ->h.L() - 1, batch = c2->batch;
    for (const lr_poly *p : {c0, c1, c2, (const lr_poly *)out0, (const lr_poly *)out1}) LR_TRY(check_ct(pl, level, p, batch));
    if (out0 == out1 || c2 == out0 || c2 == out1) return fail(LR_ERR_ARG, "bfv relinearize: out0, out1 and c2 must be distinct polys");
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    LR_TRY(same_stream(pl->cQ, pl->cP));
    LR_HIP(hipSetDevice(pl->device));
    const long long sQ = (long long)cQ->h.L() * (long long)cQ->h.N;
    LR_TRY(pl->bfvP.ensure(cQ, (size_t)2 * batch * sQ));        // keyswitchpool[2], [3] (:489-490)
    u64 *p0 = pl->bfvP.d, *p1 = pl->bfvP.d + (long long)batch * sQ;
    LR_TRY(bfv_switch_keys_core(pl, batch, c2->d, c2->stride(), evk, p0, sQ, p1, sQ));
    if (batch == 1 && !pl->opt.no_pair && c0->d != c1->d && out0->d != out1->d && out0->d != c1->d && out1->d != c0->d) {
        // one ciphertext: the two additions as one launch over two "polys" at the distances between the components
        auto words = [](const u64 *a, const u64 *b) { return (long long)(((intptr_t)b - (intptr_t)a) / (intptr_t)sizeof(u64)); };
        return run_ewise(cQ, LR_ADD, level + 1, 2, c0->d, words(c0->d, c1->d), p0, sQ, out0->d, words(out0->d, out1->d), nullptr);   // :494-495
    }
    LR_TRY(run_ewise(cQ, LR_ADD, level + 1, batch, c0->d, c0->stride(), p0, sQ, out0->d, out0->stride(), nullptr));   // :494
    return run_ewise(cQ, LR_ADD, level + 1, batch, c1->d, c1->stride(), p1, sQ, out1->d, out1->stride(), nullptr);    // :495
    });
}

// bfv.evaluator.permute (bfv/evaluator.go:711-735), the body of RotateRows (:670-681) and of RotateColumns with the key of that
// rotation (:590-592, and each step of rotateColumnsPow2 :636-662): Context.Permute of both components (coefficient domain, :723-724),
// switchKeys of the second (:729), Add and Copy (:731-732).  The key switch accumulates straight into the outputs (the reference's
// keyswitchpool[2], [3] and its Copy are the same values); out may be the input (the reference's polypool branch, :717-721).
extern "C" int lr_bfv_rotate(lr_ckks_plan *pl, const lr_poly *c0, const lr_poly *c1, uint64_t gen, const lr_poly *rotkey, lr_poly *o0,
                             lr_poly *o1) {
    return guarded([&]() -> int {
    if (!pl || !c0 || !c1 || !rotkey || !o0 || !o1) return fail(LR_ERR_ARG, "null argument");
    lr_context *cQ = pl->cQ;
    const int level = cQ->h.L() - 1, batch = c0->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    for (const lr_poly *p : {c0, c1, (const lr_poly *)o0, (const lr_poly *)o1}) LR_TRY(check_ct(pl, level, p, batch));
    if (o0->d == o1->d) return fail(LR_ERR_ARG, "bfv rotate: the two output polys must be distinct");
    if (cQ->h.N < 2 || cQ->h.logN > 31) return fail(LR_ERR_UNSUPPORTED, "ring degree");
    LR_TRY(same_stream(pl->cQ, pl->cP));
    LR_HIP(hipSetDevice(pl->device));
    const int n = (int)cQ->h.N, L1 = level + 1;
    const long long s = (long long)L1 * n;
    for (Pool *p : {&pl->c0, &pl->c2x}) LR_TRY(p->ensure(cQ, (size_t)batch * s));
    GaloisLaunch G;
    G.n = n;
    G.logn = (int)cQ->h.logN;
    G.ntt_domain = 0;
    G.gen = gen & ((cQ->h.N << 1) - 1);
    G.lp = cQ->d_lp;
    if (batch == 1 && !pl->opt.no_pair && c0->d != c1->d) {
        // one ciphertext: both components in one launch, the strides are the distances between them (see lr_ckks_rotate)
        auto words = [](const u64 *a, const u64 *b) { return (long long)(((intptr_t)b - (intptr_t)a) / (intptr_t)sizeof(u64)); };
        G.in = c0->d; G.in_stride = words(c0->d, c1->d); G.out = pl->c0.d; G.out_stride = words(pl->c0.d, pl->c2x.d);
        LR_HIP(launch_permute(G, L1, 2, cQ->stream));                                          // :723-724
    } else {
        G.in = c0->d; G.in_stride = c0->stride(); G.out = pl->c0.d; G.out_stride = s;
        LR_HIP(launch_permute(G, L1, batch, cQ->stream));                                      // :723
        G.in = c1->d; G.in_stride = c1->stride(); G.out = pl->c2x.d;
        LR_HIP(launch_permute(G, L1, batch, cQ->stream));                                      // :724
    }
    LR_TRY(bfv_switch_keys_core(pl, batch, pl->c2x.d, s, rotkey, o0->d, o0->stride(), o1->d, o1->stride()));   // :729 (p1 lands in out1: :732)
    return run_ewise(cQ, LR_ADD, L1, batch, pl->c0.d, s, o0->d, o0->stride(), o0->d, o0->stride(), nullptr);   // :731
    });
}

extern "C" int lr_ckks_rotate(lr_ckks_plan *pl, int level, const lr_poly *c0, const lr_poly *c1, uint64_t gen, const lr_poly *rotkey,
                              lr_poly *o0, lr_poly *o1) {
    return guarded([&]() -> int {
    if (!pl || !c0 || !c1 || !rotkey || !o0 || !o1) return fail(LR_ERR_ARG, "null argument");
    if (level < 0 || level + 1 > pl->cQ->h.L()) return fail(LR_ERR_SHAPE, "level out of range");
    const int batch = c0->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    for (const lr_poly *p : {c0, c1, (const lr_poly *)o0, (const lr_poly *)o1}) LR_TRY(check_ct(pl, level, p, batch));
    if (o0->stride() != o1->stride()) return fail(LR_ERR_SHAPE, "output polys must share their stride");
    lr_context *cQ = pl->cQ;
    LR_TRY(same_stream(pl->cQ, pl->cP));
    LR_HIP(hipSetDevice(cQ->device));
    const int n = (int)cQ->h.N, L1 = level + 1;
    const long long s = (long long)L1 * n;
    for (Pool *p : {&pl->c0, &pl->c2x, &pl->q1, &pl->q2}) LR_TRY(p->ensure(cQ, (size_t)batch * s));
    if (batch == 1 && !pl->opt.no_pair && c0->d != c1->d) {
        // one ciphertext: both components in one launch, the strides are the distances between them (see ks_accumulate)
        auto words = [](const u64 *a, const u64 *b) { return (long long)(((intptr_t)b - (intptr_t)a) / (intptr_t)sizeof(u64)); };
        LR_TRY(run_permute_ntt(cQ, L1, 2, c0->d, words(c0->d, c1->d), pl->c0.d, words(pl->c0.d, pl->c2x.d), gen));    // :1458-1459
    } else {
        LR_TRY(run_permute_ntt(cQ, L1, batch, c0->d, c0->stride(), pl->c0.d, s, gen));    // :1458
        LR_TRY(run_permute_ntt(cQ, L1, batch, c1->d, c1->stride(), pl->c2x.d, s, gen));   // :1459
    }
    KeySwitchEpilogue fin{o0->d, o1->d, o0->stride(), pl->c0.d, nullptr, s};
    return switch_keys_core(pl, level, batch, pl->c2x.d, s, rotkey, pl->q1.d, s, pl->q2.d, s, &fin);   // :1464-1467
    });
}

// RotateHoisted + switchKeyHoisted (ckks/evaluator.go:1252-1391): n_rot rotations of one ciphertext share the
// digit decomposition; per rotation the digits are permuted, multiplied into the rotation key and brought down.
extern "C" int lr_ckks_rotate_hoisted(lr_ckks_plan *pl, int level, const lr_poly *c0, const lr_poly *c1, int n_rot,
                                      const uint64_t *gens, const lr_poly *const *rotkeys, lr_poly *const *outs0,
                                      lr_poly *const *outs1) {
    return guarded([&]() -> int {
    if (!pl || !c0 || !c1 || !gens || !rotkeys || !outs0 || !outs1) return fail(LR_ERR_ARG, "null argument");
    if (n_rot < 0) return fail(LR_ERR_ARG, "negative rotation count");
    if (level < 0 || level + 1 > pl->cQ->h.L()) return fail(LR_ERR_SHAPE, "level out of range");
    const int batch = c0->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    LR_TRY(check_ct(pl, level, c0, batch));
    LR_TRY(check_ct(pl, level, c1, batch));
    lr_context *cQ = pl->cQ, *cP = pl->cP;
    LR_TRY(same_stream(pl->cQ, pl->cP));
    LR_HIP(hipSetDevice(cQ->device));
    const int nQ = cQ->h.L(), nP = cP->h.L(), n = (int)cQ->h.N, L1 = level + 1;
    const int alpha = pl->dec->alpha;
    const int beta = (L1 + alpha - 1) / alpha;
    const long long s = (long long)L1 * n, sQ = (long long)nQ * n, sP = (long long)nP * n;
    for (int r = 0; r < n_rot; ++r) {
        if (!rotkeys[r] || !outs0[r] || !outs1[r]) return fail(LR_ERR_ARG, "null argument");
        LR_TRY(check_ct(pl, level, outs0[r], batch));
        LR_TRY(check_ct(pl, level, outs1[r], batch));
        if (outs0[r]->stride() != outs1[r]->stride()) return fail(LR_ERR_SHAPE, "output polys must share their stride");
        if (outs0[r]->d == c0->d || outs1[r]->d == c0->d || outs0[r]->d == c1->d || outs1[r]->d == c1->d)
            return fail(LR_ERR_ARG, "hoisted rotations are not in place");
    }
    LR_TRY(ks_decompose(pl, level, batch, c1->d, c1->stride(), true));                         // :1258-1272
    for (Pool *p : {&pl->c0, &pl->q1, &pl->q2}) LR_TRY(p->ensure(cQ, (size_t)batch * s));
    LR_TRY(pl->permQ.ensure(cQ, (size_t)beta * batch * sQ));
    LR_TRY(pl->permP.ensure(cQ, (size_t)beta * batch * sP));
    for (int r = 0; r < n_rot; ++r) {
        LR_TRY(run_permute_ntt(cQ, L1, batch, c0->d, c0->stride(), pl->c0.d, s, gens[r]));     // :1314-1318
        LR_TRY(run_permute_ntt(cQ, L1, beta * batch, pl->c2QiQ.d, sQ, pl->permQ.d, sQ, gens[r]));   // :1346, all digits
        LR_TRY(run_permute_ntt(cP, nP, beta * batch, pl->c2QiP.d, sP, pl->permP.d, sP, gens[r]));   // :1347
        KeySwitchEpilogue fin{outs0[r]->d, outs1[r]->d, outs0[r]->stride(), pl->c0.d, nullptr, s};   // :1389-1390
        LR_TRY(ks_accumulate(pl, level, batch, pl->permQ.d, pl->permP.d, nullptr, 0, rotkeys[r], pl->q1.d, s, pl->q2.d, s, &fin));
    }
    return LR_OK;
    });
}

namespace {

// ckks/evaluator.go:1080-1104 after the argument checks: T holds the four operands (strided or through a pointer table)
int mulrelin_core(lr_ckks_plan *pl, int level, int batch, TensorLaunch T, const lr_poly *evk, u64 *o0, u64 *o1, long long o_stride) {
    lr_context *cQ = pl->cQ;
    LR_TRY(same_stream(pl->cQ, pl->cP));
    LR_HIP(hipSetDevice(cQ->device));
    const int n = (int)cQ->h.N, L1 = level + 1;
    const long long s = (long long)L1 * n;
    for (Pool *p : {&pl->c0, &pl->c1, &pl->c2x, &pl->q1, &pl->q2}) LR_TRY(p->ensure(cQ, (size_t)batch * s));
    // :1080-1095: MForm x2, MulCoeffsMontgomery x3, MulCoeffsMontgomeryAndAdd, one pass
    T.c0 = pl->c0.d; T.c1 = pl->c1.d; T.c2 = pl->c2x.d;
    T.c_stride = T.c1_stride = T.c2_stride = s;
    T.n = n;
    T.lp = cQ->d_lp;
    LR_HIP(launch_tensor(T, L1, batch, cQ->stream));
    // :1101 key switch of the degree-2 part, :1103-1104 the two additions fused into its last pass
    KeySwitchEpilogue fin{o0, o1, o_stride, pl->c0.d, pl->c1.d, s};
    LR_TRY(switch_keys_core(pl, level, batch, pl->c2x.d, s, evk, pl->q1.d, s, pl->q2.d, s, &fin));
    return LR_OK;
}

}  // namespace

extern "C" int lr_ckks_mulrelin(lr_ckks_plan *pl, int level, const lr_poly *a0, const lr_poly *a1, const lr_poly *b0,
                                const lr_poly *b1, const lr_poly *evk, lr_poly *o0, lr_poly *o1) {
    return guarded([&]() -> int {
    if (!pl || !a0 || !a1 || !b0 || !b1 || !evk || !o0 || !o1) return fail(LR_ERR_ARG, "null argument");
    if (level < 0 || level + 1 > pl->cQ->h.L()) return fail(LR_ERR_SHAPE, "level out of range");
    const int batch = a0->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    for (const lr_poly *p : {a0, a1, b0, b1, (const lr_poly *)o0, (const lr_poly *)o1}) LR_TRY(check_ct(pl, level, p, batch));
    if (o0->stride() != o1->stride()) return fail(LR_ERR_SHAPE, "output polys must share their stride");
    TensorLaunch T;
    T.a0 = a0->d; T.a1 = a1->d; T.b0 = b0->d; T.b1 = b1->d;
    T.a0_stride = a0->stride(); T.a1_stride = a1->stride(); T.b0_stride = b0->stride(); T.b1_stride = b1->stride();
    return mulrelin_core(pl, level, batch, T, evk, o0->d, o1->d, o0->stride());
    });
}

// ------------------------------------------------------------------------------------------
// Batcher: the reference's concurrency model is one evaluator per goroutine, one ciphertext per call
// (examples/dbfv/psi/psi.go:215-233).  On this device a product of one ciphertext fills a fraction of the chip and the streams of
// many host threads do not add up (profiles/r03: 16 threads x batch 1 = 9.1k products/s against 21k/s for one batched call).  The
// batcher turns concurrent calls back into batched launches: a call queues its request; whichever caller finds a free lane takes
// every queued request with the same (level, key) up to max_batch, runs them as ONE MulRelin whose first kernel reads the operands
// through a pointer table and whose results are scattered to the callers' polys by one copy kernel, waits for the lane's stream and
// wakes the callers.  No thread of its own, no timer: while a lane runs, arrivals queue up and form the next batch.
// ------------------------------------------------------------------------------------------
struct lr_ckks_batcher {
    struct Request {
        int level = 0, polys = 0;
        int kind = 0;              // 0: MulRelin (a0, a1) x (b0, b1); 1: rotation / conjugation of (a0, a1) by `gen` with the key `evk`
        u64 gen = 0;
        const lr_poly *a0 = nullptr, *a1 = nullptr, *b0 = nullptr, *b1 = nullptr, *evk = nullptr;
        lr_poly *o0 = nullptr, *o1 = nullptr;
        bool done = false;
        int status = LR_OK;
        std::string error;
    };
    struct Lane {
        lr_ckks_plan *plan = nullptr;
        bool busy = false;
        u64 **h_table = nullptr;   // pinned: [4 * max_batch] operand pointers, then [2 * max_batch] result pointers
        u64 **d_table = nullptr;
        Pool o0, o1;               // staged results
        hipStream_t stream = nullptr;   // created here, set on the lane's two contexts for the batcher's lifetime
        std::vector<Request *> take;    // the batch being run; reserved at creation, so that forming a batch allocates nothing (a request
                                        // taken off the queue is always completed: nothing can throw between the two)
    };
    std::vector<Lane> lanes;
    int max_batch = 0;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Request *> queue;
    unsigned long long batches = 0, products = 0;
    int largest = 0;
};

namespace {

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

}  // namespace

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
        if (ln.stream) {   // back to the library's stream (ordered behind the lane's work), then the lane stream can go
            (void)lr_context_set_stream(ln.plan->cQ, nullptr);
            (void)lr_context_set_stream(ln.plan->cP, nullptr);
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

namespace {

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
    std::unique_lock<std::mutex> lk(B->m);
    B->queue.push_back(&req);
    for (;;) {
        if (req.done) break;
        int free_lane = -1;
        for (size_t i = 0; i < B->lanes.size() && free_lane < 0; ++i)
            if (!B->lanes[i].busy) free_lane = (int)i;
        if (free_lane < 0 || B->queue.empty()) {
            B->cv.wait(lk);
            continue;
        }
        // lead: everything queued that shares the head's (level, key), in arrival order, up to max_batch polys
        auto &lane = B->lanes[free_lane];
        std::vector<lr_ckks_batcher::Request *> &take = lane.take;
        take.clear();
        int total = 0;
        const lr_ckks_batcher::Request *head = B->queue.front();
        for (auto it = B->queue.begin(); it != B->queue.end();) {
            lr_ckks_batcher::Request *r = *it;
            if (r->kind == head->kind && r->gen == head->gen && r->level == head->level && r->evk == head->evk && total + r->polys <= B->max_batch) {
                take.push_back(r);
                total += r->polys;
                it = B->queue.erase(it);
            } else {
                ++it;
            }
        }
        lane.busy = true;
        lk.unlock();
        int rc = guarded([&]() -> int { return batcher_run(B, lane, take); });
        std::string msg;
        if (rc != LR_OK) {
            try { msg = g_error; } catch (...) {}
        }
        lk.lock();
        lane.busy = false;
        B->batches += 1;
        B->products += (unsigned long long)total;
        B->largest = std::max(B->largest, total);
        for (auto *r : take) {
            r->status = rc;
            if (rc != LR_OK) {
                try { r->error = msg; } catch (...) {}
            }
            r->done = true;
        }
        B->cv.notify_all();
    }
    lk.unlock();
    if (req.status != LR_OK) return fail(req.status, req.error);
    return LR_OK;
}

}  // namespace

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

// MulRelin with evakey == nil (ckks/evaluator.go:1038-1111): the degree-2 tensor, no key switch.  The squaring branch
// (:1083-1088, c1 = 2 c0 c1 by AddLvl) and the regular one (:1090-1096, MulCoeffsMontgomeryAndAddLvl) produce the same canonical
// residues when ct0 == ct1, so one kernel serves both.  Outputs may alias the inputs (the reference goes through its pools then).
extern "C" int lr_ckks_mul_norelin(lr_ckks_plan *pl, int level, const lr_poly *a0, const lr_poly *a1, const lr_poly *b0,
                                   const lr_poly *b1, lr_poly *o0, lr_poly *o1, lr_poly *o2) {
    return guarded([&]() -> int {
    if (!pl || !a0 || !a1 || !b0 || !b1 || !o0 || !o1 || !o2) return fail(LR_ERR_ARG, "null argument");
    if (level < 0 || level + 1 > pl->cQ->h.L()) return fail(LR_ERR_SHAPE, "level out of range");
    const int batch = a0->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    for (const lr_poly *p : {a0, a1, b0, b1, (const lr_poly *)o0, (const lr_poly *)o1, (const lr_poly *)o2}) LR_TRY(check_ct(pl, level, p, batch));
    lr_context *cQ = pl->cQ;
    LR_HIP(hipSetDevice(cQ->device));
    TensorLaunch T;
    T.a0 = a0->d; T.a1 = a1->d; T.b0 = b0->d; T.b1 = b1->d;
    T.a0_stride = a0->stride(); T.a1_stride = a1->stride(); T.b0_stride = b0->stride(); T.b1_stride = b1->stride();
    T.c0 = o0->d; T.c1 = o1->d; T.c2 = o2->d;
    T.c_stride = o0->stride(); T.c1_stride = o1->stride(); T.c2_stride = o2->stride();
    T.n = (int)cQ->h.N;
    T.lp = cQ->d_lp;
    LR_HIP(launch_tensor(T, level + 1, batch, cQ->stream));
    return LR_OK;
    });
}

// MulRelin, plaintext x ciphertext (ckks/evaluator.go:1113-1131): out_k = MRed(MForm(pt), ct_k), k = 0, 1
extern "C" int lr_ckks_mul_plain(lr_ckks_plan *pl, int level, const lr_poly *pt, const lr_poly *c0, const lr_poly *c1,
                                 lr_poly *o0, lr_poly *o1) {
    return guarded([&]() -> int {
    if (!pl || !pt || !c0 || !c1 || !o0 || !o1) return fail(LR_ERR_ARG, "null argument");
    if (level < 0 || level + 1 > pl->cQ->h.L()) return fail(LR_ERR_SHAPE, "level out of range");
    const int batch = c0->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    for (const lr_poly *p : {c0, c1, (const lr_poly *)o0, (const lr_poly *)o1}) LR_TRY(check_ct(pl, level, p, batch));
    if (pt->N != pl->cQ->h.N || pt->limbs < level + 1 || (pt->batch != batch && pt->batch != 1)) return fail(LR_ERR_SHAPE, "plaintext: limbs or batch");
    lr_context *cQ = pl->cQ;
    LR_HIP(hipSetDevice(cQ->device));
    const int n = (int)cQ->h.N, L1 = level + 1;
    const long long s = (long long)L1 * n;
    LR_TRY(pl->c0.ensure(cQ, (size_t)pt->batch * s));
    LR_TRY(run_ewise(cQ, LR_MFORM, L1, pt->batch, pt->d, pt->stride(), nullptr, 0, pl->c0.d, s, nullptr));            // :1129
    const long long ms = pt->batch == 1 && batch > 1 ? 0 : s;
    LR_TRY(run_ewise(cQ, LR_MUL_MONT, L1, batch, pl->c0.d, ms, c0->d, c0->stride(), o0->d, o0->stride(), nullptr));   // :1130
    return run_ewise(cQ, LR_MUL_MONT, L1, batch, pl->c0.d, ms, c1->d, c1->stride(), o1->d, o1->stride(), nullptr);    // :1131
    });
}

// pkEncryptor.encrypt, the branch through the special primes, after the sampling (ckks/encryptor.go:205-234).
// u, pk0, pk1, e0, e1 hold |Q|+|P| limbs (the layout of contextQP); pk0 / pk1 may have batch 1.
extern "C" int lr_ckks_encrypt_pk(lr_ckks_plan *pl, int level, const lr_poly *u, const lr_poly *pk0, const lr_poly *pk1,
                                  const lr_poly *e0, const lr_poly *e1, const lr_poly *pt, lr_poly *o0, lr_poly *o1) {
    return guarded([&]() -> int {
    if (!pl || !u || !pk0 || !pk1 || !e0 || !e1 || !pt || !o0 || !o1) return fail(LR_ERR_ARG, "null argument");
    lr_context *cQ = pl->cQ, *cP = pl->cP;
    const int nQ = cQ->h.L(), nP = cP->h.L(), n = (int)cQ->h.N;
    if (level < 0 || level + 1 > nQ) return fail(LR_ERR_SHAPE, "level out of range");
    const int batch = u->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    for (const lr_poly *p : {u, pk0, pk1, e0, e1}) {
        if (p->N != cQ->h.N || p->limbs < nQ + nP) return fail(LR_ERR_SHAPE, "encrypt: u, pk and e hold |Q|+|P| limbs");
        if (p->batch != batch && !((p == pk0 || p == pk1) && p->batch == 1)) return fail(LR_ERR_SHAPE, "batch mismatch");
    }
    LR_TRY(check_ct(pl, level, o0, batch));
    LR_TRY(check_ct(pl, level, o1, batch));
    if (pt->N != cQ->h.N || pt->limbs < level + 1 || (pt->batch != batch && pt->batch != 1)) return fail(LR_ERR_SHAPE, "plaintext: limbs or batch");
    LR_TRY(same_stream(cQ, cP));
    LR_HIP(hipSetDevice(cQ->device));
    const long long sQP = (long long)(nQ + nP) * n, offP = (long long)nQ * n;
    LR_TRY(pl->encQ.ensure(cQ, (size_t)2 * batch * sQP));
    u64 *const pool[2] = {pl->encQ.d, pl->encQ.d + (long long)batch * sQP};
    const lr_poly *pk[2] = {pk0, pk1}, *e[2] = {e0, e1};
    lr_poly *outs[2] = {o0, o1};
    for (int k = 0; k < 2; ++k) {
        const long long ks = pk[k]->batch == 1 && batch > 1 ? 0 : pk[k]->stride();
        // :209-211 contextQP.MulCoeffsMontgomery(u, pk[k], pool[k]): the Q rows under contextQ's moduli, the P rows under contextP's
        LR_TRY(run_ewise(cQ, LR_MUL_MONT, nQ, batch, u->d, u->stride(), pk[k]->d, ks, pool[k], sQP, nullptr));
        LR_TRY(run_ewise(cP, LR_MUL_MONT, nP, batch, u->d + offP, u->stride(), pk[k]->d + offP, ks, pool[k] + offP, sQP, nullptr));
    }
    {   // :214-215 contextQP.InvNTT, both polys in one launch per basis
        Rows q{pool[0], sQP, 0, 1}, p{pool[0], sQP, nQ, 1};
        LR_TRY(run_ntt(cQ, true, q, q, 0, 1, nQ, 2 * batch));
        LR_TRY(run_ntt(cP, true, p, p, 0, 1, nP, 2 * batch));
    }
    for (int k = 0; k < 2; ++k) {
        // :218-220 SampleAndAdd: CRed(x + e) per coefficient (ring/gaussianSampler.go:268)
        LR_TRY(run_ewise(cQ, LR_ADD, nQ, batch, pool[k], sQP, e[k]->d, e[k]->stride(), pool[k], sQP, nullptr));
        LR_TRY(run_ewise(cP, LR_ADD, nP, batch, pool[k] + offP, sQP, e[k]->d + offP, e[k]->stride(), pool[k] + offP, sQP, nullptr));
        // :223-226 ModDownPQ(level, pool[k], ct[k]): the P part is read at rows level+1.. (ring_basis_extension.go:255)
        Rows pP{pool[k], sQP, level + 1, 1};
        LR_TRY(moddown_pq_core(pl->bext, level, pool[k], sQP, pP, batch, outs[k], false));
        Rows r = rows_of(outs[k]);
        LR_TRY(run_ntt(cQ, false, r, r, 0, 1, level + 1, batch));                                                     // :229-230
    }
    const long long ps = pt->batch == 1 && batch > 1 ? 0 : pt->stride();
    return run_ewise(cQ, LR_ADD, level + 1, batch, o0->d, o0->stride(), pt->d, ps, o0->d, o0->stride(), nullptr);     // :234
    });
}

// decryptor.Decrypt (ckks/decryptor.go:53-78): Horner evaluation of the ciphertext at the secret key
extern "C" int lr_ckks_decrypt(lr_ckks_plan *pl, int level, const lr_poly *const *ct, int degree, const lr_poly *sk, lr_poly *pt) {
    return guarded([&]() -> int {
    if (!pl || !ct || !sk || !pt) return fail(LR_ERR_ARG, "null argument");
    if (degree < 0) return fail(LR_ERR_ARG, "negative degree");
    lr_context *cQ = pl->cQ;
    if (level < 0 || level + 1 > cQ->h.L()) return fail(LR_ERR_SHAPE, "level out of range");
    const int batch = pt->batch, L1 = level + 1;
    for (int i = 0; i <= degree; ++i) {
        if (!ct[i]) return fail(LR_ERR_ARG, "null argument");
        LR_TRY(check_ct(pl, level, ct[i], batch));
    }
    LR_TRY(check_ct(pl, level, pt, batch));
    if (sk->N != cQ->h.N || sk->limbs < L1 || (sk->batch != batch && sk->batch != 1)) return fail(LR_ERR_SHAPE, "secret key: limbs or batch");
    LR_HIP(hipSetDevice(cQ->device));
    const long long ss = sk->batch == 1 && batch > 1 ? 0 : sk->stride();
    LR_TRY(run_ewise(cQ, LR_COPY, L1, batch, ct[degree]->d, ct[degree]->stride(), nullptr, 0, pt->d, pt->stride(), nullptr));   // :61
    for (int i = degree; i > 0; --i) {
        LR_TRY(run_ewise(cQ, LR_MUL_MONT, L1, batch, pt->d, pt->stride(), sk->d, ss, pt->d, pt->stride(), nullptr));            // :67
        LR_TRY(run_ewise(cQ, LR_ADD, L1, batch, pt->d, pt->stride(), ct[i - 1]->d, ct[i - 1]->stride(), pt->d, pt->stride(), nullptr));   // :68
        if ((i & 7) == 7) LR_TRY(run_ewise(cQ, LR_REDUCE, L1, batch, pt->d, pt->stride(), nullptr, 0, pt->d, pt->stride(), nullptr));     // :70
    }
    if ((degree & 7) != 7) LR_TRY(run_ewise(cQ, LR_REDUCE, L1, batch, pt->d, pt->stride(), nullptr, 0, pt->d, pt->stride(), nullptr));    // :75
    return LR_OK;
    });
}

// diagnostics: the basis extension's division by a table constant (lr_bext.hip: div_by_const) against the IEEE division of
// ring/ring_basis_extension.go:372 on `samples` pseudo-random and adversarial operand pairs; *mismatches must come back 0
extern "C" int lr_selftest_division(lr_context *c, uint64_t samples, uint64_t seed, uint64_t *mismatches) {
    return guarded([&]() -> int {
        if (!c || !mismatches) return fail(LR_ERR_ARG, "null argument");
        LR_HIP(hipSetDevice(c->device));
        unsigned long long *d = nullptr;
        LR_HIP(hipMalloc((void **)&d, sizeof(unsigned long long)));
        const int per_thread = 4096;
        const int blocks = (int)std::min<uint64_t>(std::max<uint64_t>(1, samples / (256ull * per_thread)), 1u << 20);
        hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream);
        if (e == hipSuccess) e = launch_div_selftest(seed, blocks, per_thread, d, c->stream);
        unsigned long long h = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d);
        LR_HIP(e);
        *mismatches = h;
        return LR_OK;
    });
}

extern "C" int lr_context_timeline(lr_context *c, uint32_t *dst, size_t capacity, size_t *count) {
    return guarded([&]() -> int {
    if (!c || !count) return fail(LR_ERR_ARG, "null argument");
    *count = c->stamp_used;
    if (!dst) return LR_OK;                                   // size query
    if (capacity < c->stamp_used) return fail(LR_ERR_SHAPE, "timeline: destination too small");
    LR_HIP(hipSetDevice(c->device));
    LR_HIP(hipStreamSynchronize(c->stream));
    if (c->stamp_used) LR_HIP(hipMemcpy(dst, c->d_stamps, c->stamp_used * sizeof(u32), hipMemcpyDeviceToHost));
    return LR_OK;
    });
}

extern "C" int lr_context_last_ntt_kernel(const lr_context *c, char *buf, size_t capacity) {
    return guarded([&]() -> int {
    if (!c || !buf || capacity == 0) return fail(LR_ERR_ARG, "null argument");
    std::snprintf(buf, capacity, "%s", c->last_ntt_kernel);
    return LR_OK;
    });
}

extern "C" int lr_ckks_rescale(lr_ckks_plan *pl, lr_poly *c0, lr_poly *c1) {
    return guarded([&]() -> int {
    if (!pl || !c0 || !c1) return fail(LR_ERR_ARG, "null argument");
    lr_context *c = pl->cQ;
    LR_TRY(check_rescale(c, c0));
    LR_TRY(check_rescale(c, c1));
    LR_HIP(hipSetDevice(c->device));
    // ckks/evaluator.go:958-960 divides the two components one after the other.  They are independent, and at a small batch every
    // launch of one component leaves most of the chip idle: where the two polys can be addressed as ONE batch -- base + p * stride
    // reaches both, i.e. always for one poly each (stride = the distance between them) and for batches laid out back to back --
    // every launch carries both (PN15QP880, one ciphertext: 121 -> 66 us).
    lr_poly *lo = c0->d <= c1->d ? c0 : c1, *hi = lo == c0 ? c1 : c0;
    const long long gap = hi->d - lo->d;
    const bool same_shape = c0->limbs == c1->limbs && c0->batch == c1->batch && c0->N == c1->N && c0->d != c1->d;
    const bool one_each = same_shape && c0->batch == 1 && gap >= (long long)lo->limbs * (long long)lo->N;
    const bool back_to_back = same_shape && c0->stride() == c1->stride() && gap == (long long)lo->batch * lo->stride();
    if (!c->opt.rescale_unpaired && (one_each || back_to_back) && (long long)c0->batch * 2 * c0->limbs <= c->opt.pair_max_workgroups) {
        lr_poly both = *lo;
        both.owned = false;
        both.batch = 2 * lo->batch;
        if (one_each) both.stride_words = gap;
        LR_TRY(rescale_ntt_domain(c, &both, true));
        c0->limbs = c1->limbs = both.limbs;
        return LR_OK;
    }
    LR_TRY(rescale_ntt_domain(c, c0, true));
    return rescale_ntt_domain(c, c1, true);
    });
}

// ------------------------------------------------------------------------------------------
// bfv.Evaluator.Mul (tensorAndRescale, bfv/evaluator.go:278-464)
// ------------------------------------------------------------------------------------------
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
    ~lr_bfv_plan() {
        for (u64 *p : {d_phalf_q, d_phalf_m, d_t_mont})
            if (p) (void)hipFree(p);
    }
};

namespace {
// (prod of moduli) >> 1, then reduced modulo every prime of `targets` (little-endian multi-precision)
void half_product_residues(const std::vector<u64> &moduli, const std::vector<u64> &targets, LimbScalars &out) {
    std::vector<u64> big(1, 1);
    for (u64 m : moduli) {
        u64 carry = 0;
        for (size_t i = 0; i < big.size(); ++i) {
            const u128 p = (u128)big[i] * m + carry;
            big[i] = (u64)p;
            carry = (u64)(p >> 64);
        }
        if (carry) big.push_back(carry);
    }
    for (size_t i = 0; i < big.size(); ++i) big[i] = (big[i] >> 1) | (i + 1 < big.size() ? (big[i + 1] << 63) : 0);
    std::memset(&out, 0, sizeof(out));
    for (size_t k = 0; k < targets.size(); ++k) {
        u64 r = 0;
        for (size_t i = big.size(); i-- > 0;) r = (u64)((((u128)r << 64) | big[i]) % targets[k]);
        out.v[k] = r;
    }
}
}  // namespace

extern "C" int lr_bfv_plan_create(lr_context *cQ, lr_context *cM, uint64_t t, int max_batch, lr_bfv_plan **out) {
    return lr_bfv_plan_create_ex(cQ, cM, t, max_batch, nullptr, out);
}

extern "C" int lr_bfv_plan_create_ex(lr_context *cQ, lr_context *cM, uint64_t t, int max_batch, const lr_options *options, lr_bfv_plan **out) {
    return guarded([&]() -> int {
    if (!cQ || !cM || !out) return fail(LR_ERR_ARG, "null argument");
    *out = nullptr;
    Options parsed;
    LR_TRY(options_from_public(options, &parsed));
    if (max_batch < 1) return fail(LR_ERR_ARG, "max_batch must be >= 1");
    LR_TRY(same_degree(cQ, cM));
    std::unique_ptr<lr_bfv_plan> p(new lr_bfv_plan());
    p->cQ = cQ;
    p->cM = cM;
    p->device = cQ->device;
    p->t = t;
    p->max_batch = max_batch;
    half_product_residues(cM->h.q, cQ->h.q, p->phalf_q);
    half_product_residues(cM->h.q, cM->h.q, p->phalf_m);
    std::memset(&p->t_mont, 0, sizeof(p->t_mont));
    for (int i = 0; i < cQ->h.L(); ++i)
        p->t_mont.v[i] = mform(bred_add(t, cQ->h.q[i], cQ->h.bred[i].hi), cQ->h.q[i], cQ->h.bred[i].hi, cQ->h.bred[i].lo);
    LR_HIP(hipSetDevice(cQ->device));
    LR_TRY(to_device(&p->d_phalf_q, p->phalf_q.v, (size_t)cQ->h.L()));
    LR_TRY(to_device(&p->d_phalf_m, p->phalf_m.v, (size_t)cM->h.L()));
    LR_TRY(to_device(&p->d_t_mont, p->t_mont.v, (size_t)cQ->h.L()));
    p->no_ext_epilogue = parsed.bfv_no_ext_epilogue;
    p->no_gather = parsed.bfv_no_gather;
    p->gather_below = parsed.bfv_gather_below;
    LR_TRY(lr_bext_create(cQ, cM, &p->bext));
    *out = p.release();
    return LR_OK;
    });
}

extern "C" int lr_bfv_plan_destroy(lr_bfv_plan *p) {
    return guarded([&]() -> int {
    if (!p) return LR_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();   // the handle's work may be on its contexts' caller-supplied stream
    lr_bext_destroy(p->bext);
    delete p;
    return LR_OK;
    });
}

extern "C" int lr_bfv_mul(lr_bfv_plan *pl, const lr_poly *a0, const lr_poly *a1, const lr_poly *b0, const lr_poly *b1,
                          lr_poly *o0, lr_poly *o1, lr_poly *o2) {
    return guarded([&]() -> int {
    if (!pl || !a0 || !a1 || !b0 || !b1 || !o0 || !o1 || !o2) return fail(LR_ERR_ARG, "null argument");
    lr_context *cQ = pl->cQ, *cM = pl->cM;
    const int nQ = cQ->h.L(), nM = cM->h.L(), n = (int)cQ->h.N;
    const int batch = a0->batch;
    if (batch > pl->max_batch) return fail(LR_ERR_SHAPE, "batch exceeds the plan's max_batch");
    for (const lr_poly *p : {a0, a1, b0, b1, (const lr_poly *)o0, (const lr_poly *)o1, (const lr_poly *)o2}) {
        if (p->N != cQ->h.N || p->limbs < nQ || p->batch != batch) return fail(LR_ERR_SHAPE, "BFV Mul: operands must hold |Q| limbs and share the batch");
    }
    LR_TRY(same_stream(cQ, cM));
    LR_HIP(hipSetDevice(cQ->device));
    const long long sQ = (long long)nQ * n, sM = (long long)nM * n;
    const lr_poly *A[2] = {a0, a1}, *B[2] = {b0, b1};
    lr_poly *O[3] = {o0, o1, o2};
    LR_TRY(pl->liftQ.ensure(cQ, (size_t)4 * batch * sQ));
    LR_TRY(pl->liftM.ensure(cQ, (size_t)4 * batch * sM));
    LR_TRY(pl->prodQ.ensure(cQ, (size_t)3 * batch * sQ));
    LR_TRY(pl->prodM.ensure(cQ, (size_t)3 * batch * sM));
    const long long slotQ = (long long)batch * sQ, slotM = (long long)batch * sM;
    // slots: a0, a1, b0, b1
    u64 *const aQ[2] = {pl->liftQ.d, pl->liftQ.d + slotQ}, *const bQ[2] = {pl->liftQ.d + 2 * slotQ, pl->liftQ.d + 3 * slotQ};
    u64 *const aM[2] = {pl->liftM.d, pl->liftM.d + slotM}, *const bM[2] = {pl->liftM.d + 2 * slotM, pl->liftM.d + 3 * slotM};
    u64 *const cQ3[3] = {pl->prodQ.d, pl->prodQ.d + slotQ, pl->prodQ.d + 2 * slotQ};
    u64 *const cM3[3] = {pl->prodM.d, pl->prodM.d + slotM, pl->prodM.d + 2 * slotM};
    lr_bext *bx = pl->bext;
    // A small batch: the four operand polys (unrelated addresses) are gathered into one batch of 4 B and every step of :298-313 runs
    // once on it; the three products go down as one batch of 3 B and are scattered to the callers' polys at the end.  One ciphertext
    // pair at PN14QP438: 26 launches of 3 - 6 workgroups in a row, 424 us; 11 launches, 138 us (profiles/r03/bfv_small_batch.txt).
    // The copies (two passes over 7 polys) buy nothing once a launch of one operand fills the chip.
    const bool gathered = !pl->no_gather && (long long)4 * batch * std::max(nQ, nM) * (n >= (1 << 15) ? 2 : 1) <= pl->gather_below;
    if (gathered) {
        LR_TRY(pl->stageIn.ensure(cQ, (size_t)4 * batch * sQ));
        LR_TRY(pl->stageOut.ensure(cQ, (size_t)3 * batch * sQ));
        MultiCopyLaunch G;
        const lr_poly *srcs[4] = {a0, a1, b0, b1};
        for (int k = 0; k < 4; ++k) {
            G.src[k] = srcs[k]->d;
            G.src_stride[k] = srcs[k]->stride();
            G.dst[k] = pl->stageIn.d + k * slotQ;
            G.dst_stride[k] = sQ;
        }
        G.count = 4;
        G.batch = batch;
        G.n = n;
        LR_HIP(launch_multicopy(G, nQ, cQ->stream));
        Rows in4{pl->stageIn.d, sQ, 0, 1};
        LR_TRY(run_ext(cQ, bx->qp, nQ, in4, 4 * batch, segment(pl->liftM.d, sM, 0, 0, nM), segment(nullptr, 0, 0, 0, 0)));
        LR_TRY(run_ntt(cQ, false, in4, Rows{pl->liftQ.d, sQ, 0, 1}, 0, 1, nQ, 4 * batch));
        LR_TRY(run_ntt(cM, false, Rows{pl->liftM.d, sM, 0, 1}, Rows{pl->liftM.d, sM, 0, 1}, 0, 1, nM, 4 * batch));
    } else {
        // :298-313  basis extension Q -> QMul, then NTT in both bases
        for (int i = 0; i < 2; ++i) {
            for (int side = 0; side < 2; ++side) {
                const lr_poly *src = side == 0 ? A[i] : B[i];
                u64 *dQ = side == 0 ? aQ[i] : bQ[i];
                u64 *dM = side == 0 ? aM[i] : bM[i];
                LR_TRY(run_ext(cQ, bx->qp, nQ, rows_of(src), batch, segment(dM, sM, 0, 0, nM), segment(nullptr, 0, 0, 0, 0)));
                LR_TRY(run_ntt(cQ, false, rows_of(src), Rows{dQ, sQ, 0, 1}, 0, 1, nQ, batch));
                LR_TRY(run_ntt(cM, false, Rows{dM, sM, 0, 1}, Rows{dM, sM, 0, 1}, 0, 1, nM, batch));
            }
        }
    }
    // :327-367 MForm x2 and the four products per base, one pass each (the middle component comes out reduced where
    // the reference leaves it in [0,2q): the InvNTT that follows is canonical either way)
    for (int base = 0; base < 2; ++base) {
        lr_context *cx = base == 0 ? cQ : cM;
        const long long sx = base == 0 ? sQ : sM;
        TensorLaunch T;
        T.a0 = base == 0 ? aQ[0] : aM[0];
        T.a1 = base == 0 ? aQ[1] : aM[1];
        T.b0 = base == 0 ? bQ[0] : bM[0];
        T.b1 = base == 0 ? bQ[1] : bM[1];
        T.a0_stride = T.a1_stride = T.b0_stride = T.b1_stride = sx;
        T.c0 = base == 0 ? cQ3[0] : cM3[0];
        T.c1 = base == 0 ? cQ3[1] : cM3[1];
        T.c2 = base == 0 ? cQ3[2] : cM3[2];
        T.c_stride = T.c1_stride = T.c2_stride = sx;
        T.n = n;
        T.lp = cx->d_lp;
        LR_HIP(launch_tensor(T, base == 0 ? nQ : nM, batch, cx->stream));
    }
    // :423-463 back to coefficients, divide by Q (result over QMul), centre, back to Q, times t
    const LimbScalars &tsc = pl->t_mont;
    const long long poolM_stride = sM;
    // the element-wise tails ride in the extensions' stores where the extension kernel in use has the epilogue (ExtSegment::epi_mode)
    const bool fuse_down = !pl->no_ext_epilogue && ext_epilogue_supported(bx->qp.tables(), nQ, n);
    const bool fuse_up = !pl->no_ext_epilogue && ext_epilogue_supported(bx->pq.tables(), nM, n);
    // the three products one after the other, or (gathered) as one batch of 3 B whose results are scattered afterwards
    const int rounds = gathered ? 1 : 3, nb = gathered ? 3 * batch : batch;
    if (!fuse_down) LR_TRY(bx->poolP.ensure(cM, (size_t)nb * poolM_stride));
    for (int i = 0; i < rounds; ++i) {
        u64 *const outp = gathered ? pl->stageOut.d : O[i]->d;
        const long long outs = gathered ? sQ : O[i]->stride();
        Rows q1{cQ3[i], sQ, 0, 1}, q2{cM3[i], sM, 0, 1};
        LR_TRY(run_ntt(cQ, true, q1, q1, 0, 1, nQ, nb));
        LR_TRY(run_ntt(cM, true, q2, q2, 0, 1, nM, nb));
        // ModDownSplitedQP(levelQ, levelQMul, c2Q1, c2Q2, c2Q2), ring_basis_extension.go:314, with the AddScalarBigint(pHalf) of :457
        if (fuse_down) {
            ExtSegment sd = segment(cM3[i], sM, 0, 0, nM);
            sd.epi_mode = 1;
            sd.epi_x = cM3[i];                     // read and written at the same position by the same thread
            sd.epi_x_stride = sM;
            sd.epi_c = bx->d_moddown_qp;
            sd.epi_s = pl->d_phalf_m;
            LR_TRY(run_ext(cQ, bx->qp, nQ, q1, nb, sd, segment(nullptr, 0, 0, 0, 0)));
        } else {
            LR_TRY(run_ext(cQ, bx->qp, nQ, q1, nb, segment(bx->poolP.d, poolM_stride, 0, 0, nM), segment(nullptr, 0, 0, 0, 0)));
            LR_TRY(run_submul(cM, nM, nb, cM3[i], sM, bx->poolP.d, poolM_stride, (long long)n, cM3[i], sM, bx->d_moddown_qp, false, nullptr,
                              nullptr, 0, &pl->phalf_m));
        }
        // :458 ModUpSplitPQ, :459 SubScalarBigint(pHalf), :462 MulScalar(t)
        if (fuse_up) {
            ExtSegment su = segment(outp, outs, 0, 0, nQ);
            su.epi_mode = 2;
            su.epi_c = pl->d_t_mont;
            su.epi_s = pl->d_phalf_q;
            LR_TRY(run_ext(cQ, bx->pq, nM, q2, nb, su, segment(nullptr, 0, 0, 0, 0)));
        } else {
            LR_TRY(run_ext(cQ, bx->pq, nM, q2, nb, segment(outp, outs, 0, 0, nQ), segment(nullptr, 0, 0, 0, 0)));
            ScalarPairLaunch S;
            S.in = outp;
            S.out = outp;
            S.in_stride = S.out_stride = outs;
            S.n = n;
            S.lp = cQ->d_lp;
            S.sub = pl->phalf_q;
            S.mul = tsc;
            LR_HIP(launch_scalar_pair(S, nQ, nb, cQ->stream));
        }
    }
    if (gathered) {
        MultiCopyLaunch S;
        for (int k = 0; k < 3; ++k) {
            S.src[k] = pl->stageOut.d + k * slotQ;
            S.src_stride[k] = sQ;
            S.dst[k] = O[k]->d;
            S.dst_stride[k] = O[k]->stride();
        }
        S.src[3] = nullptr; S.dst[3] = nullptr; S.src_stride[3] = S.dst_stride[3] = 0;
        S.count = 3;
        S.batch = batch;
        S.n = n;
        LR_HIP(launch_multicopy(S, nQ, cQ->stream));
    }
    return LR_OK;
    });
}

// ------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------
extern "C" int lr_timer_start(lr_context *c) {
    return guarded([&]() -> int {
    if (!c) return fail(LR_ERR_ARG, "null context");
    LR_HIP(hipSetDevice(c->device));
    LR_HIP(hipEventRecord(c->ev0, c->stream));
    return LR_OK;
    });
}

extern "C" int lr_timer_stop(lr_context *c, float *elapsed_ms) {
    return guarded([&]() -> int {
    if (!c || !elapsed_ms) return fail(LR_ERR_ARG, "null argument");
    LR_HIP(hipSetDevice(c->device));
    LR_HIP(hipEventRecord(c->ev1, c->stream));
    LR_HIP(hipEventSynchronize(c->ev1));
    LR_HIP(hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    return LR_OK;
    });
}
