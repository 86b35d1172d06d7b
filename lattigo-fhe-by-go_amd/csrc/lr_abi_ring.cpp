// lr_abi_ring.cpp -- C ABI: the ring.Context methods -- NTT dispatch (assembly code objects / C++ kernels), the coefficient-wise family,
// Galois automorphisms, SimpleScaler, the RNS rescale -- and the context diagnostics.
#include "lr_host.hpp"

// ------------------------------------------------------------------------------------------
// NTT
// ------------------------------------------------------------------------------------------
namespace lr_host {



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
int run_ntt(lr_context *c, bool inverse, Rows in, Rows out, int mod0, int mod_step, int count, int batch, int hole,
            int group, const NttEpilogue *epi, bool pretop, bool lazy) {
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
    // the launcher writes the kernel's name into a local buffer; it reaches the context under its diagnostics mutex on every way out
    struct KernelNote {
        lr_context *c;
        char buf[32];
        ~KernelNote() {
            if (!buf[0]) return;
            std::lock_guard<std::mutex> lock(c->diag_mu);
            std::memcpy(c->last_ntt_kernel, buf, sizeof buf);
        }
    } note{c, ""};
    char *kn = note.buf;
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
    std::snprintf(note.buf, sizeof note.buf, "ntt_%s_kernel<%u>", inverse ? "inv" : "fwd", logn);
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

Rows rows_of(const lr_poly *p, int limb0, int step, bool broadcast_ok, int target_batch) {
    Rows r;
    r.base = p->d;
    r.stride = (broadcast_ok && p->batch == 1 && target_batch > 1) ? 0 : p->stride();
    r.limb0 = limb0;
    r.step = step;
    return r;
}

}  // namespace lr_host

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
namespace lr_host {

bool op_reads_b(int op) {
    return op == LR_ADD || op == LR_ADD_NOMOD || op == LR_SUB || op == LR_SUB_NOMOD ||
           (op >= LR_MUL_COEFFS && op <= LR_MUL_MONT_CONSTANT);
}

// raw form used by the pipelines: pointers are already offset to limb 0 of the operands
int run_ewise(lr_context *c, int op, int limbs, int batch, const u64 *a, long long a_stride, const u64 *b,
              long long b_stride, u64 *out, long long out_stride, const LimbScalars *sc, int lp_offset) {
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

}  // namespace lr_host

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


// ------------------------------------------------------------------------------------------
// RNS rescale (ring/ring_scaling.go:9-164)
// ------------------------------------------------------------------------------------------
namespace lr_host {

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
int rescale_round_table(lr_context *c, int level, const u64 **out, const EpiLimb **epi_out, const u64 **zeros_out) {
    std::lock_guard<std::mutex> lock(c->rescale_mu);
    auto it = c->rescale_round.find(level);
    if (it != c->rescale_round.end()) {
        *out = it->second.plus;
        *epi_out = it->second.epi;
        if (zeros_out) *zeros_out = it->second.zeros;
        return LR_OK;
    }
    // built into locals; the cache only ever holds complete tables (a failure below leaves no entry behind)
    const int n = (int)c->h.N;
    const long long words = (long long)level * n;
    struct Guard {
        u64 *table = nullptr, *zeros = nullptr;
        EpiLimb *epi = nullptr;
        ~Guard() {
            if (table) (void)hipFree(table);
            if (zeros) (void)hipFree(zeros);
            if (epi) (void)hipFree(epi);
        }
    } g;
    ScratchLease tmpbuf;
    LR_TRY(tmpbuf.take(&c->scratch, (size_t)words));
    LR_HIP(hipMalloc((void **)&g.table, (size_t)words * sizeof(u64)));
    LR_HIP(hipMemsetAsync(g.table, 0, (size_t)words * sizeof(u64), c->stream));
    // the flooring division (DivFloorByLastModulusNTT) takes the same epilogue with nothing to add: rows of zeros in the same layout
    LR_HIP(hipMalloc((void **)&g.zeros, (size_t)words * sizeof(u64)));
    LR_HIP(hipMemsetAsync(g.zeros, 0, (size_t)words * sizeof(u64), c->stream));
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
    c->rescale_round[level] = lr_context::RoundTable{g.table, g.epi, g.zeros};
    *out = g.table;
    *epi_out = g.epi;
    if (zeros_out) *zeros_out = g.zeros;
    g.table = nullptr;
    g.zeros = nullptr;
    g.epi = nullptr;
    return LR_OK;
}

int rescale_ntt_domain(lr_context *c, lr_poly *p0, bool round) {
    const int level = p0->limbs - 1, n = (int)c->h.N, batch = p0->batch;
    const long long tmp_stride = (long long)level * n;
    const u64 *plus = nullptr, *zeros = nullptr;
    const EpiLimb *ec = nullptr;
    if (!c->opt.rescale_unfused && (round || ntt_epilogue_ok(c))) LR_TRY(rescale_round_table(c, level, &plus, &ec, &zeros));
    if (!round) plus = nullptr;       // (the rounding's addend; the flooring division adds the rows of zeros in the epilogue, nothing elsewhere)
    const u64 *const epi_plus = round ? plus : zeros;
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
    if (epi_plus && ntt_epilogue_ok(c)) {
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
                const NttEpilogue ep{p0->d, p0->stride(), epi_plus, 0, ec};
                Rows src{scratch.d(), tmp_stride, l0, 1}, dst{p0->d, p0->stride(), l0, 1};
                LR_TRY(run_ntt(c, false, src, dst, l0, 1, l1 - l0, batch, 0, 0, &ep, true));
            } else if (fpc) {
                const NttEpilogue ep{p0->d, p0->stride(), epi_plus, 0, ec};
                Rows dst{p0->d, p0->stride(), l0, 1};
                LR_TRY(run_ntt(c, false, last, dst, l0, 1, l1 - l0, batch, 0, 0, &ep));
            } else {
                Rows dst{scratch.d(), tmp_stride, l0, 1};
                LR_TRY(run_ntt(c, false, last, dst, l0, 1, l1 - l0, batch));
                LR_TRY(run_submul(c, l1 - l0, batch, p0->d + l0 * n64, p0->stride(), scratch.d() + l0 * n64, tmp_stride, n64,
                                  p0->d + l0 * n64, p0->stride(), c->d_rescale + (size_t)(level - 1) * c->h.L() + l0, false, nullptr,
                                  plus ? plus + l0 * n64 : nullptr, 0, nullptr, l0));
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

}  // namespace lr_host

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
    std::lock_guard<std::mutex> lock(c->diag_mu);
    std::snprintf(buf, capacity, "%s", c->last_ntt_kernel);
    return LR_OK;
    });
}
