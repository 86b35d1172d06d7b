// lr_abi_bext.cpp -- C ABI: ring.FastBasisExtender (ModUp / ModDown) and ring.Decomposer.
#include "lr_host.hpp"

// ------------------------------------------------------------------------------------------
// basis extension
// ------------------------------------------------------------------------------------------
namespace lr_host {



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

int flush_ext(lr_context *c, std::vector<ExtPending> &pending, int batch, unsigned long long *grouped_launches) {
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
            const ExtSegment *s2, std::vector<ExtPending> *collect, bool inv_top) {
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
    s.epi_x2 = nullptr;
    s.epi_x2_stride = 0;
    s.epi_c = s.epi_s = nullptr;
    return s;
}

int run_submul(lr_context *c, int limbs, int batch, const u64 *a, long long a_stride, const u64 *b, long long b_stride,
               long long b_row_stride, u64 *out, long long out_stride, const u64 *d_consts, bool reduce_b,
               const LimbScalars *addend, const u64 *plus, long long plus_stride, const LimbScalars *post,
               int limb0) {
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

}  // namespace lr_host

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

namespace lr_host {

// shared tail of the four ModDown...PQ variants: P part (coefficient domain, rows p_limb0.. of pP)
// -> poolQ[0..level] by modUpExact, optional NTT, then p2 = MRed(p1Q + (q - pool), P^-1)
// x2 (coefficient-domain form only, optional): the Q part is CRed(p1Q + x2) -- the residues SampleAndAdd adds in front of the ModDown of
// pkEncryptor.encrypt, folded into the extension's epilogue; false in *x2_taken when the kernel in use has no epilogue (the caller adds first)
int moddown_pq_core(lr_bext *b, int level, const u64 *p1Q, long long p1Q_stride, Rows pP, int batch, lr_poly *p2, bool ntt, const u64 *x2,
                    long long x2_stride) {
    lr_context *cQ = b->cQ;
    const int nP = b->cP->h.L();
    const long long pool_stride = (long long)cQ->h.L() * (long long)cQ->h.N;
    if (!ntt && !cQ->opt.no_epilogue && ext_epilogue_supported(b->pq.tables(), nP, (int)cQ->h.N)) {
        // coefficient domain: the subtract-multiply rides in the extension's stores (ExtSegment::epi_mode 1)
        ExtSegment sd = segment(p2->d, p2->stride(), 0, 0, level + 1);
        sd.epi_mode = 1;
        sd.epi_x = p1Q;
        sd.epi_x_stride = p1Q_stride;
        sd.epi_x2 = x2;
        sd.epi_x2_stride = x2_stride;
        sd.epi_c = b->d_moddown_pq;
        return run_ext(cQ, b->pq, nP, pP, batch, sd, segment(nullptr, 0, 0, 0, 0));
    }
    if (x2) return fail(LR_ERR_INTERNAL, "moddown_pq_core: the addend form needs the extension epilogue (moddown_epilogue_available)");
    LR_TRY(b->poolQ.ensure(cQ, (size_t)batch * pool_stride));
    LR_TRY(run_ext(cQ, b->pq, nP, pP, batch, segment(b->poolQ.d, pool_stride, 0, 0, level + 1), segment(nullptr, 0, 0, 0, 0)));
    if (ntt && ntt_epilogue_ok(cQ) && !cQ->opt.no_epilogue) {
        // NTT domain: the subtract-multiply rides in the forward transform's copy-out (the kernels the key switch uses for its ModDown,
        // ks_accumulate) for every run of limbs that takes the epilogue; the other limbs keep the separate pass.  Nothing is added.
        if (b->zerosQ.words < (size_t)pool_stride) {
            LR_TRY(b->zerosQ.ensure(cQ, (size_t)pool_stride));
            LR_HIP(hipMemsetAsync(b->zerosQ.d, 0, (size_t)pool_stride * sizeof(u64), cQ->stream));
        }
        const long long n64 = (long long)cQ->h.N;
        int l0 = 0;
        while (l0 <= level) {
            const bool fpc = ntt_epilogue_limb(cQ, l0);
            int l1 = l0 + 1;
            while (l1 <= level && ntt_epilogue_limb(cQ, l1) == fpc) ++l1;
            Rows src{b->poolQ.d, pool_stride, l0, 1};
            if (fpc) {
                const NttEpilogue ep{p1Q, p1Q_stride, b->zerosQ.d, 0, b->d_moddown_pq_epi};
                Rows dst{p2->d, p2->stride(), l0, 1};
                LR_TRY(run_ntt(cQ, false, src, dst, l0, 1, l1 - l0, batch, 0, 0, &ep));
            } else {
                LR_TRY(run_ntt(cQ, false, src, src, l0, 1, l1 - l0, batch));
                LR_TRY(run_submul(cQ, l1 - l0, batch, p1Q + l0 * n64, p1Q_stride, b->poolQ.d + l0 * n64, pool_stride, n64, p2->d + l0 * n64,
                                  p2->stride(), b->d_moddown_pq + l0, false, nullptr, nullptr, 0, nullptr, l0));
            }
            l0 = l1;
        }
        return LR_OK;
    }
    if (ntt) {
        Rows pr{b->poolQ.d, pool_stride, 0, 1};
        LR_TRY(run_ntt(cQ, false, pr, pr, 0, 1, level + 1, batch));
    }
    return run_submul(cQ, level + 1, batch, p1Q, p1Q_stride, b->poolQ.d, pool_stride, (long long)cQ->h.N, p2->d,
                      p2->stride(), b->d_moddown_pq, false, nullptr);
}

bool moddown_epilogue_available(const lr_bext *b) {
    return !b->cQ->opt.no_epilogue && ext_epilogue_supported(b->pq.tables(), b->cP->h.L(), (int)b->cQ->h.N);
}

}  // namespace lr_host

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

namespace lr_host {

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
                   long long outP_stride, bool split, bool top, bool skip_own, std::vector<ExtPending> *collect,
                   bool inv_top) {
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

}  // namespace lr_host

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
