// lr_abi_core.cpp -- C ABI (include/lattigo_ring.h): errors, options, contexts, polys and their host <-> device movement, timers.
// No CPU fallback exists anywhere in the library: every arithmetic entry point launches gfx950 kernels and fails with LR_ERR_HIP
// when no device is available.
#include "lr_host.hpp"

thread_local std::string lr_host::g_error = "";

namespace lr_host {
hipError_t create_stream(hipStream_t *s, int cls) {
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
    static std::map<int, hipStream_t> &streams = *new std::map<int, hipStream_t>();   // never destroyed, like the streams it holds: handles may be
                                                                                    // released by a garbage-collected host after main() returns
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
}  // namespace lr_host

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

namespace lr_host {

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

}  // namespace lr_host

extern "C" int lr_options_init(lr_options *opt) {
    if (!opt) return LR_ERR_ARG;
    std::memset(opt, 0, sizeof *opt);
    opt->struct_size = (uint32_t)sizeof *opt;
    opt->version = LR_OPTIONS_VERSION;
    opt->ntt_mode = opt->asm_variant = opt->ntt_split15 = opt->ntt_stagger = opt->ntt_persist = -1;
    return LR_OK;
}

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
    c->scratch.owner_stream = &c->stream;
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
    if (c->lane_of) return fail(LR_ERR_ARG, "this context belongs to a lane of a live batcher: destroy the batcher first");
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();   // whatever stream the handle last ran on (its own, the shared one, a caller's)
    for (void *p : {(void *)c->d_lp, (void *)c->d_fwd, (void *)c->d_inv, (void *)c->d_fwd_fin, (void *)c->d_inv_fin, (void *)c->d_rescale,
                    (void *)c->d_fwd_fp, (void *)c->d_inv_fp, (void *)c->d_fwd_fin_fp, (void *)c->d_inv_fin_fp, (void *)c->d_fp_lp})
        if (p) (void)hipFree(p);
    if (c->d_stamps) (void)hipFree(c->d_stamps);
    for (auto &kv : c->rescale_round) {
        if (kv.second.plus) (void)hipFree(kv.second.plus);
        if (kv.second.zeros) (void)hipFree(kv.second.zeros);
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
        if (e1 != hipSuccess) {
            // the old stream is gone (a caller-owned stream destroyed before the switch: whatever it carried has completed or was
            // dropped with it): no event can order against it -- drain the device instead and install the new stream all the same
            (void)hipGetLastError();
            LR_HIP(hipDeviceSynchronize());
        } else if (e2 != hipSuccess) {
            return fail(LR_ERR_HIP, std::string("set_stream: ") + hipGetErrorString(e2));
        }
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
    // the big-endian image lands in a buffer of the context's scratch pool (no allocation, and no hipFree with its device-wide wait, per
    // call); the lease goes back after the synchronisation below
    ScratchLease stage;
    LR_TRY(stage.take(&p->ctx->scratch, words + 1));
    LR_HIP(hipMemcpyAsync(stage.d(), data + 2, words * 8, hipMemcpyHostToDevice, p->ctx->stream));
    hipError_t e = launch_bswap(stage.d(), p->d + (long long)batch_index * p->stride(), words, p->ctx->stream);
    const hipError_t es = hipStreamSynchronize(p->ctx->stream);
    LR_HIP(e);
    LR_HIP(es);
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
    ScratchLease stage;
    LR_TRY(stage.take(&p->ctx->scratch, words + 1));
    hipError_t e = launch_bswap(p->d + (long long)batch_index * p->stride(), stage.d(), words, p->ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(data + 2, stage.d(), words * 8, hipMemcpyDeviceToHost, p->ctx->stream);
    const hipError_t es = hipStreamSynchronize(p->ctx->stream);     // also on the error path: the lease must not return while work is queued
    LR_HIP(e);
    LR_HIP(es);
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
