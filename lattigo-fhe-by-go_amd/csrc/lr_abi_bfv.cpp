// lr_abi_bfv.cpp -- C ABI: lr_bfv_plan and bfv.Evaluator.Mul (tensorAndRescale).
#include "lr_host.hpp"

// ------------------------------------------------------------------------------------------
// bfv.Evaluator.Mul (tensorAndRescale, bfv/evaluator.go:278-464)
// ------------------------------------------------------------------------------------------
namespace lr_host {
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
}  // namespace lr_host

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
    if (p->lane_of) return fail(LR_ERR_ARG, "this plan is a lane of a live batcher: destroy the batcher first");
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
    // ct0 == ct1 (:306, :334 "squaring case"): the second operand is not lifted and transformed again.  Its tensor (c0 = c0[0]^2,
    // c1 = 2 c0[0] c0[1] by AddNoMod, c2 = c0[1]^2) and the regular one give the same canonical polys after the InvNTT -- MRed(MForm(x), y)
    // and MRed(MForm(y), x) are the same residue below q --, so the tensor kernel simply reads the first operand's slots twice.
    const bool square = a0 == b0 && a1 == b1;
    const int sides = square ? 1 : 2;
    // slots: a0, a1, b0, b1
    u64 *const aQ[2] = {pl->liftQ.d, pl->liftQ.d + slotQ};
    u64 *const aM[2] = {pl->liftM.d, pl->liftM.d + slotM};
    u64 *const bQ[2] = {square ? aQ[0] : pl->liftQ.d + 2 * slotQ, square ? aQ[1] : pl->liftQ.d + 3 * slotQ};
    u64 *const bM[2] = {square ? aM[0] : pl->liftM.d + 2 * slotM, square ? aM[1] : pl->liftM.d + 3 * slotM};
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
        const int polys = 2 * sides;
        for (int k = 0; k < 4; ++k) {
            G.src[k] = k < polys ? srcs[k]->d : nullptr;
            G.src_stride[k] = k < polys ? srcs[k]->stride() : 0;
            G.dst[k] = k < polys ? pl->stageIn.d + k * slotQ : nullptr;
            G.dst_stride[k] = k < polys ? sQ : 0;
        }
        G.count = polys;
        G.batch = batch;
        G.n = n;
        LR_HIP(launch_multicopy(G, nQ, cQ->stream));
        Rows in4{pl->stageIn.d, sQ, 0, 1};
        LR_TRY(run_ext(cQ, bx->qp, nQ, in4, polys * batch, segment(pl->liftM.d, sM, 0, 0, nM), segment(nullptr, 0, 0, 0, 0)));
        LR_TRY(run_ntt(cQ, false, in4, Rows{pl->liftQ.d, sQ, 0, 1}, 0, 1, nQ, polys * batch));
        LR_TRY(run_ntt(cM, false, Rows{pl->liftM.d, sM, 0, 1}, Rows{pl->liftM.d, sM, 0, 1}, 0, 1, nM, polys * batch));
    } else {
        // :298-313  basis extension Q -> QMul, then NTT in both bases
        for (int i = 0; i < 2; ++i) {
            for (int side = 0; side < sides; ++side) {
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
