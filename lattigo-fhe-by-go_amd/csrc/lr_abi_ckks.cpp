// lr_abi_ckks.cpp -- C ABI: lr_ckks_plan and the ckks.Evaluator / bfv key-switch call sequences (switchKeysInPlace, MulRelin, rotations,
// hoisted rotations, Relinearize, pk-encrypt, decrypt, Rescale).
#include "lr_host.hpp"

// ------------------------------------------------------------------------------------------
// ckks.Evaluator call sequences
// ------------------------------------------------------------------------------------------
namespace lr_host {
// live plans per device that are not lanes of a batcher: one = a lone evaluator, whose small launches may run side by side (PlanFork)
std::atomic<int> &standalone_plans(int device) {
    static std::atomic<int> counts[64];
    return counts[device >= 0 && device < 64 ? device : 0];
}
}  // namespace lr_host

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
    if (p->lane_of) return fail(LR_ERR_ARG, "this plan is a lane of a live batcher: destroy the batcher first (it holds the plan and its contexts)");
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();   // the handle's work may be on its contexts' caller-supplied stream
    lr_bext_destroy(p->bext);
    lr_decomposer_destroy(p->dec);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->aux) (void)hipStreamDestroy(p->aux);
    standalone_plans(p->device).fetch_sub(1);
    delete p;
    return LR_OK;
    });
}

namespace lr_host {

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


int run_permute_ntt(lr_context *c, int limbs, int batch, const u64 *in, long long in_stride, u64 *out, long long out_stride,
                    u64 gen, const u64 *const *in_table) {
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
int ks_decompose(lr_ckks_plan *pl, int level, int batch, const u64 *cx, long long cx_stride, bool copy_own, bool coeff_input) {
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
                  bool coeff_out, u64 perm_gen) {
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
        K.perm_gen = (unsigned)(perm_gen & ((cQ->h.N << 1) - 1));      // hoisted rotations: the digits through the Galois permutation
        K.logn = (int)cQ->h.logN;
        if (K.perm_gen != 0 && own) return fail(LR_ERR_ARG, "permuted digits carry their own limbs");
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
                     long long p0_stride, u64 *p1, long long p1_stride, const KeySwitchEpilogue *fin) {
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

}  // namespace lr_host

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
    if (pl->opt.no_epilogue) {
        LR_TRY(pl->permQ.ensure(cQ, (size_t)beta * batch * sQ));
        LR_TRY(pl->permP.ensure(cQ, (size_t)beta * batch * sP));
    }
    for (int r = 0; r < n_rot; ++r) {
        LR_TRY(run_permute_ntt(cQ, L1, batch, c0->d, c0->stride(), pl->c0.d, s, gens[r]));     // :1314-1318
        KeySwitchEpilogue fin{outs0[r]->d, outs1[r]->d, outs0[r]->stride(), pl->c0.d, nullptr, s};   // :1389-1390
        if (pl->opt.no_epilogue) {
            // the reference's shape: permuted copies of every digit (:1346-1347), then the inner product over them
            LR_TRY(run_permute_ntt(cQ, L1, beta * batch, pl->c2QiQ.d, sQ, pl->permQ.d, sQ, gens[r]));
            LR_TRY(run_permute_ntt(cP, nP, beta * batch, pl->c2QiP.d, sP, pl->permP.d, sP, gens[r]));
            LR_TRY(ks_accumulate(pl, level, batch, pl->permQ.d, pl->permP.d, nullptr, 0, rotkeys[r], pl->q1.d, s, pl->q2.d, s, &fin));
        } else {
            // the permutation of the digits rides on the inner product's loads (KeyMacLaunch::perm_gen): same values, 2 x beta x (|Q| + |P|)
            // rows per rotation less to write and read back
            LR_TRY(ks_accumulate(pl, level, batch, pl->c2QiQ.d, pl->c2QiP.d, nullptr, 0, rotkeys[r], pl->q1.d, s, pl->q2.d, s, &fin, false, gens[r]));
        }
    }
    return LR_OK;
    });
}

namespace lr_host {

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

}  // namespace lr_host

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
    const bool fused = !pl->opt.no_epilogue;     // Options::no_epilogue: the reference's call-by-call shape (one launch per Context call)
    if (fused) {
        // :209-211 both products with the public key in one pass over u (Q rows under contextQ's moduli, P rows under contextP's)
        Mul2Launch M;
        M.a = u->d; M.a_stride = u->stride();
        M.b0 = pk0->d; M.b0_stride = pk0->batch == 1 && batch > 1 ? 0 : pk0->stride();
        M.b1 = pk1->d; M.b1_stride = pk1->batch == 1 && batch > 1 ? 0 : pk1->stride();
        M.out0 = pool[0]; M.out1 = pool[1];
        M.out0_stride = M.out1_stride = sQP;
        M.n = n;
        M.lp = cQ->d_lp;
        LR_HIP(launch_mul2(M, nQ, batch, cQ->stream));
        M.a += offP; M.b0 += offP; M.b1 += offP; M.out0 += offP; M.out1 += offP;
        M.lp = cP->d_lp;
        LR_HIP(launch_mul2(M, nP, batch, cQ->stream));
    } else {
        for (int k = 0; k < 2; ++k) {
            const long long ks = pk[k]->batch == 1 && batch > 1 ? 0 : pk[k]->stride();
            // :209-211 contextQP.MulCoeffsMontgomery(u, pk[k], pool[k]): the Q rows under contextQ's moduli, the P rows under contextP's
            LR_TRY(run_ewise(cQ, LR_MUL_MONT, nQ, batch, u->d, u->stride(), pk[k]->d, ks, pool[k], sQP, nullptr));
            LR_TRY(run_ewise(cP, LR_MUL_MONT, nP, batch, u->d + offP, u->stride(), pk[k]->d + offP, ks, pool[k] + offP, sQP, nullptr));
        }
    }
    {   // :214-215 contextQP.InvNTT, both polys in one launch per basis
        Rows q{pool[0], sQP, 0, 1}, p{pool[0], sQP, nQ, 1};
        LR_TRY(run_ntt(cQ, true, q, q, 0, 1, nQ, 2 * batch));
        LR_TRY(run_ntt(cP, true, p, p, 0, 1, nP, 2 * batch));
    }
    // the Q rows' share of SampleAndAdd rides in the ModDown's extension epilogue (x = CRed(pool + e) where the extension reads x) when the
    // call is at the top level (below it the reference's ModDownPQ reads rows level+1.. of Q as its "P part": those rows need their e first)
    const bool add_in_ext = fused && level == nQ - 1 && moddown_epilogue_available(pl->bext);
    for (int k = 0; k < 2; ++k) {
        // :218-220 SampleAndAdd: CRed(x + e) per coefficient (ring/gaussianSampler.go:268)
        if (!add_in_ext) LR_TRY(run_ewise(cQ, LR_ADD, nQ, batch, pool[k], sQP, e[k]->d, e[k]->stride(), pool[k], sQP, nullptr));
        LR_TRY(run_ewise(cP, LR_ADD, nP, batch, pool[k] + offP, sQP, e[k]->d + offP, e[k]->stride(), pool[k] + offP, sQP, nullptr));
        // :223-226 ModDownPQ(level, pool[k], ct[k]): the P part is read at rows level+1.. (ring_basis_extension.go:255)
        Rows pP{pool[k], sQP, level + 1, 1};
        LR_TRY(moddown_pq_core(pl->bext, level, pool[k], sQP, pP, batch, outs[k], false, add_in_ext ? e[k]->d : nullptr, e[k]->stride()));
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
    bool aliased = false;      // the fused pass reads every ct[i] where it writes pt: fine for ct[degree] == pt only if nothing else is pt
    for (int i = 0; i < degree; ++i) aliased = aliased || ct[i]->d == pt->d;
    if (degree <= kHornerMaxDegree && !aliased && !pl->opt.no_epilogue) {
        // one pass: every component and the key read once, the plaintext written once (the element operations and the reduction
        // cadence are the reference's, HornerLaunch); Options::no_epilogue keeps the call-by-call form below
        HornerLaunch H;
        std::memset(&H, 0, sizeof H);
        for (int i = 0; i <= degree; ++i) {
            H.ct[i] = ct[i]->d;
            H.ct_stride[i] = ct[i]->stride();
        }
        H.sk = sk->d;
        H.sk_stride = ss;
        H.out = pt->d;
        H.out_stride = pt->stride();
        H.degree = degree;
        H.n = (int)cQ->h.N;
        H.lp = cQ->d_lp;
        LR_HIP(launch_horner(H, L1, batch, cQ->stream));
        return LR_OK;
    }
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
