/*
 * lattigo_ring.h -- C ABI of the MI355X-native `ring` hot path.
 *
 * This is the drop-in boundary for Lattigo v1.3.1's `ring` package
 * (github.com/ldsec/lattigo/ring).  The reference has no FFI of its own: bfv/ckks
 * call methods on *ring.Context, *ring.FastBasisExtender and *ring.Decomposer
 * directly.  A Go shim package with the same exported identifiers forwards each
 * method to the entry point below that names it (see INTEGRATION.md for the cgo
 * stub).  Every entry point cites the reference symbol it replaces
 * (path:line relative to the Lattigo tree).
 *
 * Conventions
 *  - plain C, opaque handles, pointers and sizes only; no exceptions or aborts cross
 *    the boundary.  Every function returns an lr_status (0 = ok).
 *  - a handle is bound to one HIP device and one HIP stream.  Calls are asynchronous
 *    on that stream unless the name ends in _host or the doc says "synchronises".
 *    Distinct handles may be used from distinct threads; one handle is single-threaded,
 *    like the reference's evaluators (examples/dbfv/psi/psi.go:221).
 *  - lr_poly is a device-resident batch of polynomials laid out (poly, limb, coeff)-major:
 *        coeff(b, i, j) = base[(b * limbs + i) * N + j]          uint64
 *    i.e. the dense image of `batch` Go values `Poly.Coeffs [][]uint64`
 *    (ring/ring_object.go:11-13).  Every operation applies to all polys of the batch;
 *    an operand with batch == 1 is broadcast (shared keys / constants).
 *  - "level" has the reference's meaning: limbs 0..level are touched, the rest ignored
 *    (ring/ntt.go:11, ring/ring.go:20).  Passing more limbs than a poly owns is
 *    LR_ERR_SHAPE (Go would panic with an index error).
 *  - in == out aliasing is legal wherever the reference allows it (everywhere; the Galois permutations are "not in place" in the
 *    reference as well, ring/ring_galois.go:54, and return LR_ERR_ARG).
 *  - results are bit-identical to the reference on the same inputs.
 *  - threads: an lr_context is immutable after creation and may be shared by threads, each with its own polys / extender /
 *    decomposer / plan (the reference's goroutine-per-evaluator model); its temporaries are leased per call.  Every other handle
 *    is single-threaded, like the reference's FastBasisExtender and evaluators (their scratch polys, ring_basis_extension.go:16-17).
 *  - configuration is an lr_options struct handed to the *_create_ex entry points (below); the plain *_create forms use the defaults.
 *    The LR_* environment variables of INTEGRATION.md section 7 are a TEST-ONLY override of the same fields, read in one place
 *    (lr::Options::apply_env) when a handle is created; a deployment sets none of them.
 */
#ifndef LATTIGO_RING_H
#define LATTIGO_RING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum lr_status {
    LR_OK = 0,
    LR_ERR_INVALID_DEGREE = 1, /* N not a power of two: ring/ring_context.go:71-73 (panic in Go)            */
    LR_ERR_NOT_NTT_FRIENDLY = 2, /* "provided modulus does not allow NTT": ring/ring_context.go:141-146     */
    LR_ERR_SHAPE = 3,          /* limb/batch/N mismatch (Go: index-out-of-range panic)                       */
    LR_ERR_ARG = 4,            /* null handle, bad enum, unsupported parameter                               */
    LR_ERR_HIP = 5,            /* HIP runtime failure; see lr_last_error_string()                            */
    LR_ERR_UNSUPPORTED = 6,    /* size outside what the kernels are built for                                */
    LR_ERR_NOMEM = 7,          /* host allocation failed (std::bad_alloc caught at the boundary)                */
    LR_ERR_INTERNAL = 8        /* any other C++ exception caught at the boundary; see lr_last_error_string()    */
} lr_status;

typedef struct lr_context lr_context;       /* ring.Context           ring/ring_context.go:18-51            */
typedef struct lr_poly lr_poly;             /* batch of ring.Poly     ring/ring_object.go:11-13             */
typedef struct lr_bext lr_bext;             /* ring.FastBasisExtender ring/ring_basis_extension.go:9-18     */
typedef struct lr_decomposer lr_decomposer; /* ring.Decomposer        ring/ring_basis_extension.go:398-407  */
typedef struct lr_simple_scaler lr_simple_scaler; /* ring.SimpleScaler  ring/ring_scaling.go:168-181           */
typedef struct lr_ckks_plan lr_ckks_plan;   /* scratch pools + tables of ckks.evaluator, ckks/evaluator.go:63-97 */

const char *lr_last_error_string(void);     /* thread-local, never NULL */
int lr_device_count(int *count);
/* version / build info: "lattigo_ring <ver> gfx950 hip" */
const char *lr_build_info(void);

/* ------------------------------------------------------------------ options ---------- */
/* Every switch selects between code paths that give THE SAME BITS (each has a parity test that runs both); the defaults are the measured
 * best on MI355X.  A caller fills the struct with lr_options_init and changes what it wants; `struct_size` lets a library newer than its
 * caller tell which fields the caller knows (fields beyond struct_size take their defaults), `version` must be LR_OPTIONS_VERSION.
 * Flags: 0 = the default path, non-zero = the alternative the field names.  Thresholds: 0 = the built-in default (in parentheses;
 * measured on PN15QP880 / PN16QP1761 / PN14QP438, DESIGN.md "decision table").  INTEGRATION.md section 7 maps every LR_* test variable
 * to its field. */
#define LR_OPTIONS_VERSION 1
typedef struct lr_options {
    uint32_t struct_size;           /* sizeof(lr_options) as the caller compiled it                                                   */
    uint32_t version;               /* LR_OPTIONS_VERSION                                                                             */
    /* --- transforms (lr_context) */
    int32_t no_asm;                 /* C++ NTT kernels only (lr_ntt.hip) instead of the gfx950 assembly code objects                  */
    int32_t no_fp;                  /* integer butterflies for every modulus (no FP64 body for the limbs below 2^46)                  */
    int32_t ntt_mode;               /* -1 = by modulus size; 0 / 3: a more conservative lazy-correction mode of the C++ kernels       */
    int32_t asm_variant;            /* -1 = by modulus size; 0 / 1: a more conservative integer variant of the assembly kernels       */
    int32_t asm14_1024;             /* N = 2^14: the 1024-thread kernels at every launch size                                         */
    int32_t no_wide14_small;        /* N = 2^14: the 512-thread kernels for small launches too                                        */
    int32_t wide14_max_items;       /* N = 2^14: launches of at most this many transforms use the 1024-thread kernels (256)           */
    int32_t ntt_split15;            /* N = 2^15 as two 2^14 sub-blocks: -1 = launches of at most split15_max_workgroups, 0 never, 1 always */
    int32_t split15_max_workgroups; /* (128)                                                                                          */
    int32_t no_invfuse;             /* N = 2^16 inverse: lazy sub-blocks + a separate last-stage pass instead of the pair-flag kernels */
    int32_t no_grid_padding;        /* assembly launches with the limb count on grid x as it is (not padded to a multiple of eight)   */
    int32_t ntt_stagger;            /* -1 / 0 = off; kilo-clocks per step of a start-up stagger of the first round of workgroups      */
    int32_t ntt_persist;            /* diagnostics builds only (LR_BUILD_DIAG): polys per workgroup of the persistent forward kernels */
    int32_t ntt_timeline;           /* diagnostics builds only: 2^15 launches run the clock-stamping builds (lr_context_timeline)     */
    /* --- epilogues and the rescale (lr_context) */
    int32_t no_epilogue;            /* separate subtract-multiply passes instead of the forward kernels' / extensions' epilogues      */
    int32_t no_int_epilogue;        /* the epilogue on the FP64 bodies only                                                           */
    int32_t rescale_unfused;        /* rounding rescale with explicit shifted copies                                                  */
    int32_t rescale_unpaired;       /* lr_ckks_rescale: the two components one after the other at every batch size                    */
    int32_t pair_max_workgroups;    /* two components of one ciphertext as ONE launch while it has at most this many workgroups (256) */
    /* --- basis extension (lr_bext, lr_decomposer: taken from their first context) */
    int32_t ext_narrow;             /* one Montgomery product per term instead of the 128-bit column sums                             */
    int32_t ext_ieee_div;           /* IEEE division in the float correction instead of the reciprocal + two residual steps           */
    int32_t no_ext_chunks;          /* a small batch's extension as one launch over all target columns                                */
    /* --- key switch and the pipelines built on it (lr_ckks_plan) */
    int32_t no_staging;             /* N = 2^16: in-place forward transforms after the digits' extensions                             */
    int32_t no_exttop;              /* the top transform stage as its own pass instead of inside the extension                        */
    int32_t no_invtop;              /* the inverse transform's last stage as its own pass instead of inside the extension             */
    int32_t no_ext_group;           /* one extension launch per digit instead of one grouped launch                                   */
    int32_t keymac_narrow;          /* one Montgomery product per term in the key inner product                                       */
    int32_t no_pair;                /* a single ciphertext's two components as two launches                                           */
    int32_t no_fork;                /* never run two independent launches of a lone plan side by side on an auxiliary stream          */
    int32_t fork_below_workgroups;  /* fork only while the forked launch has fewer workgroups than this (256)                         */
    /* --- bfv Mul (lr_bfv_plan) */
    int32_t bfv_no_ext_epilogue;    /* SubScalarBigint / MulScalar as separate passes                                                 */
    int32_t bfv_no_gather;          /* never gather the four operand polys of a small batch into one buffer                           */
    int64_t bfv_gather_below;       /* gather while the joint transform has fewer workgroups than this (1536)                         */
} lr_options;
/* fills *opt with the defaults (struct_size = sizeof(lr_options) of THIS library, version = LR_OPTIONS_VERSION) */
int lr_options_init(lr_options *opt);
/* the options a handle ended up with, after the test-only environment override (diagnostics; struct_size / version of the library) */
int lr_context_get_options(const lr_context *ctx, lr_options *out);

/* ------------------------------------------------------------------ Context ---------- */
/* ring.NewContextWithParams = SetParameters + GenNTTParams (ring/ring_context.go:60,68,129).
 * Computes every constant and psi table on the host exactly as the reference does
 * (incl. primitiveRoot's search order, ring/utils.go:182) and uploads them to `device`. */
int lr_context_create(uint64_t N, const uint64_t *moduli, int n_moduli, int device, lr_context **out);
/* the same with explicit options (NULL = defaults).  LR_ERR_ARG for a version this library does not know. */
int lr_context_create_ex(uint64_t N, const uint64_t *moduli, int n_moduli, int device, const lr_options *opt, lr_context **out);
int lr_context_destroy(lr_context *ctx);
/* diagnostics: the assembly NTT variant the context's moduli select (forward, inverse): 0..2 = integer lazy-correction modes,
 * 3 = dual kernels (FP64 butterflies for the limbs below 2^46, integer body for the others), -1 = C++ kernels only */
int lr_context_ntt_variants(const lr_context *ctx, int *forward, int *inverse);
/* use an externally owned hipStream_t; NULL = the library's own stream of that device (a non-blocking stream every context of the
 * device shares by default).  NULL does NOT mean HIP's legacy default stream: handle 0 cannot be expressed here, and the library's
 * stream does not synchronise with the legacy stream -- a caller whose other work (e.g. a framework's collectives) is ordered on
 * its "current" stream must create an explicit stream, make it current and pass its handle (bench.py's config-5 leg).
 * A stream belongs to a CONTEXT; polys, extenders, decomposers and plans run on the stream their contexts have at call time.
 * Handles built over two contexts (lr_bext, lr_ckks_plan, lr_bfv_plan) interleave launches of both: set the same stream on both
 * contexts, or the pipeline entry points return LR_ERR_ARG.  Switching is ordered on the device: the new stream waits for an
 * event recorded on the old one (work already enqueued, and the scratch later calls reuse, stay in order); change streams between
 * calls, not while another thread is inside a call on this context.  THE OLD STREAM MUST STILL BE ALIVE when the switch is made: a
 * caller that owns the stream calls lr_context_set_stream(ctx, NULL) (or installs its next stream) BEFORE destroying it -- recording
 * an event on a destroyed hipStream_t is a use-after-free inside the HIP runtime (it crashes, it does not return an error; measured in
 * round 4), so the library cannot detect it.  If the runtime does report an error for the old stream, the library drains the device
 * instead and installs the new stream all the same.
 * HIP graphs: a pipeline call captured after one warm-up call bakes the addresses of the context's pooled scratch into the graph;
 * the pool never frees a buffer while the context lives, so replays stay valid until lr_context_destroy. */
int lr_context_set_stream(lr_context *ctx, void *hip_stream);
/* diagnostics: name of the kernel the last NTT / InvNTT launch of this context dispatched, e.g. "lr_ntt_fwd15_m1" (assembly
 * code object) or "ntt_fwd_kernel<15>" (C++ kernel); bench.py reports it next to the roofline figures */
int lr_context_last_ntt_kernel(const lr_context *ctx, char *buf, size_t capacity);
/* diagnostics: a context created with LR_NTT_TIMELINE=1 runs its forward N = 2^15 launches of the 60-bit integer kernel on a
 * build of the same kernel that stamps the shader clock (low word of s_memtime) at 13 phase boundaries in every wave.  Copies
 * the stamps of the last such launch to dst: [workgroup = poly * limbs + limb][wave 0..15][16] uint32 (tools/timeline.py names
 * the phases).  dst == NULL: only *count (words needed).  Synchronises.  The transform's results are unchanged. */
int lr_context_timeline(lr_context *ctx, uint32_t *dst, size_t capacity, size_t *count);
/* Diagnostics: the basis extension divides float64(y_i) by float64(q_i) (ring/ring_basis_extension.go:372) with a reciprocal from the
 * host and two residual corrections instead of the generic IEEE expansion; this runs both on `samples` pseudo-random and adversarial
 * operand pairs on the device and counts the quotients that differ in any bit (must be 0). */
int lr_selftest_division(lr_context *ctx, uint64_t samples, uint64_t seed, uint64_t *mismatches);
int lr_context_sync(lr_context *ctx);       /* hipStreamSynchronize on the context's stream */
int lr_context_info(const lr_context *ctx, uint64_t *N, int *n_moduli, int *device);

/* Read-back of the precomputed constants, for parity tests; mirrors the getters
 * GetBredParams/GetMredParams/GetPsi/GetPsiInv/GetNttPsi/GetNttPsiInv/GetNttNInv
 * (ring/ring_context.go:253-285).  dst is host memory of the stated element count. */
typedef enum lr_table {
    LR_TAB_MODULUS = 0,    /* [L]                                  */
    LR_TAB_BRED = 1,       /* [L][2] = {hi, lo} of floor(2^128/q)  */
    LR_TAB_MRED = 2,       /* [L]    q^-1 mod 2^64                 */
    LR_TAB_PSI_MONT = 3,   /* [L]                                  */
    LR_TAB_PSI_INV_MONT = 4, /* [L]                                */
    LR_TAB_NTT_PSI = 5,    /* [L][N] bit-reversed, Montgomery form */
    LR_TAB_NTT_PSI_INV = 6,/* [L][N]                               */
    LR_TAB_NTT_N_INV = 7,  /* [L]                                  */
    LR_TAB_RESCALE = 8,    /* [L][L] row j-1, col i (i<j): rescaleParams[j-1][i], ring_context.go:148-158 */
    LR_TAB_MASK = 9        /* [L]                                  */
} lr_table;
int lr_context_get_table(const lr_context *ctx, int which, uint64_t *dst, size_t dst_count);

/* ------------------------------------------------------------------ Poly ------------- */
/* Context.NewPoly / NewPolyLvl (ring/ring_context.go:288,300), for `batch` polys at once; zero-filled. */
int lr_poly_alloc(lr_context *ctx, int limbs, int batch, lr_poly **out);
/* wrap caller-owned device memory (e.g. a torch uint64/int64 tensor) without copying */
int lr_poly_wrap(lr_context *ctx, void *device_ptr, int limbs, int batch, lr_poly **out);
/* the same with `poly_stride_words` uint64 elements between consecutive polys (>= limbs * N): a component of an array of
 * ciphertexts laid out [ciphertext][component][limb][N] is one lr_poly with the stride of a whole ciphertext */
int lr_poly_wrap_strided(lr_context *ctx, void *device_ptr, int limbs, int batch, long long poly_stride_words, lr_poly **out);
int lr_poly_free(lr_poly *p);
int lr_poly_info(const lr_poly *p, uint64_t *N, int *limbs, int *batch, void **device_ptr);
/* Go boundary: gather from / scatter to the per-limb slices of one Poly ([][]uint64 cannot be
 * passed through cgo as a whole; the shim passes `limbs` pinned *uint64).  Synchronises. */
int lr_poly_upload(lr_poly *p, int batch_index, const uint64_t *const *limb_ptrs, int limbs);
int lr_poly_download(const lr_poly *p, int batch_index, uint64_t *const *limb_ptrs, int limbs);
/* the same one limb at a time (`limb` < the poly's limb count): what a cgo caller under the reference's go 1.13 uses -- a Go slice's
 * pointer may cross for the duration of a call, an array of such pointers in C memory may not (go/ring/poly.go).  Synchronises. */
int lr_poly_upload_limb(lr_poly *p, int batch_index, int limb, const uint64_t *src);
int lr_poly_download_limb(const lr_poly *p, int batch_index, int limb, uint64_t *dst);
/* Poly.MarshalBinary / UnmarshalBinary image of ONE poly (ring/ring_object.go:159-176,222-229,252-270): byte 0 = log2 N,
 * byte 1 = number of moduli, then limb-major big-endian uint64 (WriteCoeffsTo :146, DecodeCoeffs :197).  The bytes
 * cross PCIe as they are and are swapped on the device, so serialized ciphertexts and keys go disk -> HBM without
 * a host pass.  An encoding with fewer moduli than the poly fills its first rows.  Synchronises. */
int lr_poly_unmarshal(lr_poly *p, int batch_index, const uint8_t *data, size_t len);
int lr_poly_marshal(const lr_poly *p, int batch_index, uint8_t *data, size_t capacity, size_t *written);
/* dense host image [batch][limbs][N].  Synchronises. */
int lr_poly_upload_dense(lr_poly *p, const uint64_t *host, size_t count);
int lr_poly_download_dense(const lr_poly *p, uint64_t *host, size_t count);
int lr_poly_zero(lr_poly *p);               /* Poly.Zero, ring/ring_object.go:60 */
/* Rescale re-slices its argument (`p0.Coeffs = p0.Coeffs[:level]`, ring/ring_scaling.go:33);
 * the device buffer keeps its stride, only the logical limb count changes. */
int lr_poly_set_limbs(lr_poly *p, int limbs);

/* ------------------------------------------------------------------ NTT -------------- */
/* Context.NTTLvl / Context.NTT (ring/ntt.go:11,4): forward negacyclic NTT of limbs 0..level of
 * every poly in the batch; natural order in, bit-reversed order out, canonical output [0,q).
 * Accepts any uint64 input the reference accepts (values >= q, see ring/ring_scaling.go:19,102). */
int lr_ntt(lr_context *ctx, int level, const lr_poly *in, lr_poly *out);
/* Context.InvNTTLvl / Context.InvNTT (ring/ntt.go:25,18). */
int lr_intt(lr_context *ctx, int level, const lr_poly *in, lr_poly *out);
/* package-level ring.NTT / ring.InvNTT on ONE limb under modulus index `mod_index`
 * (ring/ntt.go:53,89): in/out limb index select rows of the polys.  Used by callers that
 * transform a foreign limb under another limb's modulus (ring/ring_scaling.go:19,105). */
int lr_ntt_limb(lr_context *ctx, int mod_index, const lr_poly *in, int in_limb, lr_poly *out, int out_limb);
int lr_intt_limb(lr_context *ctx, int mod_index, const lr_poly *in, int in_limb, lr_poly *out, int out_limb);
/* literal drop-in for a Go `Context.NTT(p1, p2)` call on host slices: upload, transform,
 * download.  Synchronises. */
int lr_ntt_host(lr_context *ctx, int level, const uint64_t *const *in_limbs, uint64_t *const *out_limbs);
int lr_intt_host(lr_context *ctx, int level, const uint64_t *const *in_limbs, uint64_t *const *out_limbs);
/* package-level ring.NTT / ring.InvNTT (ring/ntt.go:53,89) on one limb under modulus `mod_index`, host slices, may be in place */
int lr_ntt_host_limb(lr_context *ctx, int mod_index, int inverse, const uint64_t *in, uint64_t *out);

/* ------------------------------------------------------------------ coefficient-wise -- */
/* One entry point for the whole ring/ring.go family; `op` selects the method. */
typedef enum lr_ewise_op {
    LR_ADD = 0,                 /* Add/AddLvl                       ring/ring.go:10,20   */
    LR_ADD_NOMOD = 1,           /* AddNoMod/AddNoModLvl             :32,42               */
    LR_SUB = 2,                 /* Sub/SubLvl                       :54,64               */
    LR_SUB_NOMOD = 3,           /* SubNoMod/SubNoModLvl             :77,87               */
    LR_NEG = 4,                 /* Neg/NegLvl                       :100,110             */
    LR_REDUCE = 5,              /* Reduce/ReduceLvl                 :122,133             */
    LR_MUL_COEFFS = 6,          /* MulCoeffs (Barrett)              :187                 */
    LR_MUL_COEFFS_AND_ADD = 7,  /* MulCoeffsAndAdd                  :198                 */
    LR_MUL_COEFFS_AND_ADD_NOMOD = 8, /* MulCoeffsAndAddNoMod        :209                 */
    LR_MUL_COEFFS_CONSTANT = 9, /* MulCoeffsConstant                :335                 */
    LR_MUL_MONT = 10,           /* MulCoeffsMontgomery(Lvl)         :221,233             */
    LR_MUL_MONT_AND_ADD = 11,   /* MulCoeffsMontgomeryAndAdd(Lvl)   :247,259             */
    LR_MUL_MONT_AND_ADD_NOMOD = 12, /* ...AndAddNoMod(Lvl)          :273,285             */
    LR_MUL_MONT_CONSTANT_AND_ADD_NOMOD = 13, /* ...ConstantAndAddNoModLvl :297           */
    LR_MUL_MONT_AND_SUB = 14,   /* MulCoeffsMontgomeryAndSub        :311                 */
    LR_MUL_MONT_AND_SUB_NOMOD = 15, /* ...AndSubNoMod               :323                 */
    LR_MUL_MONT_CONSTANT = 16,  /* MulCoeffsMontgomeryConstant      :347                 */
    LR_MFORM = 17,              /* MForm/MFormLvl                   :583,595             */
    LR_INV_MFORM = 18,          /* InvMForm                         :610                 */
    LR_MUL_SCALAR = 19,         /* MulScalar/MulScalarLvl           :513,526  scalars[0]            */
    LR_MUL_SCALAR_LIMBS = 20,   /* MulScalarBigint(Lvl)             :541,557  scalars[i] = s mod qi */
    LR_ADD_SCALAR_LIMBS = 21,   /* AddScalarBigint                  :477      (writes its 1st arg)  */
    LR_SUB_SCALAR_LIMBS = 22,   /* SubScalarBigint                  :500      (writes its 1st arg)  */
    LR_COPY = 23,               /* Copy/CopyLvl                     ring/ring_object.go:85,98       */
    LR_MUL_BY_POW2 = 24,        /* MulByPow2/MulByPow2Lvl           ring/ring.go:629,645 scalars[0] */
    LR_EWISE_OP_COUNT = 25
} lr_ewise_op;
/* out <- op(a, b) (accumulating ops read out as well).  b may be NULL for 2-operand ops.
 * scalars: host pointer, 1 value (MUL_SCALAR, MUL_BY_POW2) or level+1 values (*_LIMBS). */
int lr_ewise(lr_context *ctx, int op, int level, const lr_poly *a, const lr_poly *b, lr_poly *out,
             const uint64_t *scalars);

/* ------------------------------------------------------------------ Galois automorphisms */
/* The constant-by-ciphertext methods of ckks.Evaluator index Coeffs directly with one scalar for the coefficients below N/2 and one
 * for the rest (AddConst ckks/evaluator.go:429-445, MultByConstAndAdd :588-606, MultByConst :712-730, MultByi :765-779, DivByi
 * :814-828): out[i][j] = OP(in[i][j], j < N/2 ? lo[i] : hi[i]) for the limbs 0..level, with the reference's exact element operation,
 * op 0: CRed(x + s)   op 1: MRed(x, s)   op 2: CRed(out + MRed(x, s)).  lo / hi: level+1 host words each (the scalars as the
 * reference computes them: scaleUpExact + MForm, or nttPsi[i][1] and its negation).  in may be out. */
int lr_half_scalar_op(lr_context *ctx, int op, int level, const lr_poly *in, const uint64_t *lo, const uint64_t *hi, lr_poly *out);
/* ring.PermuteNTT (ring/ring_galois.go:55) on limbs 0..level: out[i][j] = in[i][index(j)], index(j) =
 * bitrev(((gen * (2*bitrev(j)+1) mod 2N) - 1) / 2), computed on the fly.  PermuteNTTWithIndex (:89) with the
 * table of PermuteNTTIndex(gen, power, N) (:29) is the same call with gen^power mod 2N.  Not in place
 * ("Careful, not inplace!"): in == out is LR_ERR_ARG. */
int lr_permute_ntt(lr_context *ctx, int level, const lr_poly *in, uint64_t gen, lr_poly *out);
/* PermuteNTTIndex (:29): host table of N entries */
int lr_permute_ntt_index(uint64_t gen, uint64_t power, uint64_t N, uint64_t *index);
/* Context.Permute (:106), coefficient domain, all limbs of the context: out[j][i*gen mod N] = +-in[j][i]
 * (a zero coefficient whose sign flips becomes q, as in the reference).  Not in place. */
int lr_permute(lr_context *ctx, const lr_poly *in, uint64_t gen, lr_poly *out);
/* Context.MultByMonomial (ring/ring.go:663): out = in * X^monomial_deg in Z_q[X]/(X^N+1), coefficient domain, all limbs.
 * Negated coefficients are q - x without reduction, as in the reference (a zero coefficient becomes q).  in == out is allowed
 * (staged through a temporary, as the reference's tmpx, :682). */
int lr_mult_by_monomial(lr_context *ctx, const lr_poly *in, uint64_t monomial_deg, lr_poly *out);

/* Context.Shift (ring/ring.go:575): out = in rotated left by n coefficient positions, every limb (n masked with (1 << N) - 1 as Go
 * evaluates it: all ones for N >= 64); n > N is the reference's slice panic: LR_ERR_ARG.  in == out is allowed. */
int lr_shift(lr_context *ctx, const lr_poly *in, uint64_t n, lr_poly *out);
/* Context.Rotate (ring/ring.go:775): coefficient j of every limb times omega^(n j), omega = psi^2, j = 1 .. N-1, canonical; coefficient 0
 * untouched.  The reference writes into p1 whatever its p2 argument is (:791), hence one poly here. */
int lr_rotate(lr_context *ctx, lr_poly *p1, uint64_t n);

/* ------------------------------------------------------------------ SimpleScaler --- */
/* NewSimpleScaler(t, context) (ring/ring_scaling.go:186): per modulus qi the integer part wi and the double-double
 * fractional part ti of ((Q/qi)^-1 mod qi) * t / qi, computed with the operation sequence of ring/float128.go.
 * t a power of two selects the masked reduction (:201-211), otherwise Montgomery / Barrett modulo t (:214-243).
 * t == 0 is LR_ERR_ARG (the reference divides by zero). */
int lr_simple_scaler_create(lr_context *ctx, uint64_t t, lr_simple_scaler **out);
int lr_simple_scaler_destroy(lr_simple_scaler *s);
/* host copies of wi[count] and ti[count][2] (hi, lo), count = number of moduli */
int lr_simple_scaler_tables(const lr_simple_scaler *s, uint64_t *wi, double *ti, int count);
/* SimpleScaler.Scale (:275): p2[j][i] = round(t/Q * p1[.][i]) mod t for every limb j of p2 (p2 may belong to another
 * context of the same degree, e.g. bfv's contextT; bfv/encoder.go:142).  p1 holds all moduli of the scaler's context,
 * coefficient domain.  p1 == p2 is allowed. */
int lr_simple_scale(lr_simple_scaler *s, const lr_poly *p1, lr_poly *p2);

/* ------------------------------------------------------------------ basis extension --- */
/* NewFastBasisExtender(contextQ, contextP), ring/ring_basis_extension.go:57 */
int lr_bext_create(lr_context *ctxQ, lr_context *ctxP, lr_bext **out);
int lr_bext_destroy(lr_bext *b);
/* ModUpSplitQP (:147): p1 over Q[0..level] -> p2 over all of P.  Coefficient domain. */
int lr_modup_split_qp(lr_bext *b, int level, const lr_poly *p1, lr_poly *p2);
/* ModUpSplitPQ (:154): p1 over P[0..level] -> p2 over all of Q. */
int lr_modup_split_pq(lr_bext *b, int level, const lr_poly *p1, lr_poly *p2);
/* ModDownNTTPQ (:163): p1 over Q||P (|Q|+|P| limbs, NTT domain; its P limbs are left in the
 * coefficient domain, as in Go) -> p2 over Q[0..level]. */
int lr_moddown_ntt_pq(lr_bext *b, int level, lr_poly *p1, lr_poly *p2);
/* ModDownSplitedNTTPQ (:207): (p1Q, p1P) NTT domain -> p2; p1P is left in the coefficient domain. */
int lr_moddown_split_ntt_pq(lr_bext *b, int level, const lr_poly *p1Q, lr_poly *p1P, lr_poly *p2);
/* ModDownPQ (:248): p1 over Q[0..level]||P, coefficient domain. */
int lr_moddown_pq(lr_bext *b, int level, const lr_poly *p1, lr_poly *p2);
/* ModDownSplitedPQ (:281) */
int lr_moddown_split_pq(lr_bext *b, int level, const lr_poly *p1Q, const lr_poly *p1P, lr_poly *p2);
/* ModDownSplitedQP (:314): divide by Q, result over P[0..levelP]. */
int lr_moddown_split_qp(lr_bext *b, int levelQ, int levelP, const lr_poly *p1Q, const lr_poly *p1P, lr_poly *p2);
/* tables for parity tests: 0 = modDownParamsPQ [|Q|], 1 = modDownParamsQP [|P|] (:39-53) */
int lr_bext_get_table(const lr_bext *b, int which, uint64_t *dst, size_t dst_count);

/* NewDecomposer(Q, P), :415 */
int lr_decomposer_create(lr_context *ctxQ, lr_context *ctxP, lr_decomposer **out);
int lr_decomposer_destroy(lr_decomposer *d);
/* Decompose (:476): p0 over Q (coefficient domain) -> p1 over Q[0..level]||P. */
int lr_decompose(lr_decomposer *d, int level, int crt_decomp_level, const lr_poly *p0, lr_poly *p1);
/* DecomposeAndSplit (:601): -> p1Q over Q[0..level], p1P over P. */
int lr_decompose_and_split(lr_decomposer *d, int level, int crt_decomp_level, const lr_poly *p0,
                           lr_poly *p1Q, lr_poly *p1P);

/* ------------------------------------------------------------------ RNS rescale ------- */
/* In place on p0 (limbs = level+1); on return p0 owns one limb less (see lr_poly_set_limbs).
 * DivFloorByLastModulusNTT (ring/ring_scaling.go:9), DivFloorByLastModulus (:37),
 * DivRoundByLastModulusNTT (:72), DivRoundByLastModulus (:117). */
int lr_div_floor_by_last_modulus_ntt(lr_context *ctx, lr_poly *p0);
int lr_div_floor_by_last_modulus(lr_context *ctx, lr_poly *p0);
int lr_div_round_by_last_modulus_ntt(lr_context *ctx, lr_poly *p0);
int lr_div_round_by_last_modulus(lr_context *ctx, lr_poly *p0);
/* ...Many / ...ManyNTT (:58,65,153,160) */
int lr_div_floor_by_last_modulus_many(lr_context *ctx, lr_poly *p0, int nb_rescales, int ntt_domain);
int lr_div_round_by_last_modulus_many(lr_context *ctx, lr_poly *p0, int nb_rescales, int ntt_domain);

/* ------------------------------------------------------------------ caller sequences -- */
/* The ring-level call sequence of ckks.Evaluator, kept on the device for a whole batch.
 * lr_ckks_plan owns what ckks.NewEvaluator builds: FastBasisExtender, Decomposer and the
 * scratch pools (ckks/evaluator.go:81-112). */
int lr_ckks_plan_create(lr_context *ctxQ, lr_context *ctxP, int max_batch, lr_ckks_plan **out);
/* the same with explicit options; NULL = the options of ctxQ (what lr_ckks_plan_create does) */
int lr_ckks_plan_create_ex(lr_context *ctxQ, lr_context *ctxP, int max_batch, const lr_options *opt, lr_ckks_plan **out);
/* Diagnostics of the small-batch paths of the key switch (no reference counterpart).  forks: how often two independent launches of a
 * pipeline (the digits' P rows beside their Q rows; ModDown's two components) went out side by side on the plan's auxiliary stream
 * instead of in order -- done at N = 2^16 (where one workgroup of such a launch runs long enough to pay for the hand-over) while the
 * forked launch is far from filling the chip AND the plan is the only one alive on its device that is not a batcher's lane (a lone
 * evaluator); LR_NO_FORK=1 at plan creation switches it off.  grouped_extensions:
 * launches that carried the basis extensions of all digits of a key switch at once (LR_NO_EXT_GROUP=1: one launch per digit).
 * Results are the same bits either way.  Either pointer may be NULL. */
int lr_ckks_plan_stats(const lr_ckks_plan *plan, uint64_t *forks, uint64_t *grouped_extensions);
int lr_ckks_plan_destroy(lr_ckks_plan *plan);
/* switchKeysInPlace (ckks/evaluator.go:1475): cx over Q[0..level], NTT domain;
 * evk = SwitchingKey.evakey as one poly, batch = beta*2, limbs = |Q|+|P|
 * (ckks/keygen.go:68-70: evakey[i][0], evakey[i][1] in NTT + Montgomery form);
 * p0, p1 over Q[0..level] receive the two key-switched components (any strides; neither may alias cx). */
int lr_ckks_switch_keys(lr_ckks_plan *plan, int level, const lr_poly *cx, const lr_poly *evk,
                        lr_poly *p0, lr_poly *p1);
/* bfv.evaluator.switchKeys (bfv/evaluator.go:736-812) on the same plan (what bfv.NewEvaluator builds for it -- decomposer,
 * baseconverterQ1P, key-switch pools, bfv/evaluator.go:100-112 -- is what the CKKS plan holds): cx over all of Q in the
 * COEFFICIENT domain, evk = SwitchingKey.evakey as for lr_ckks_switch_keys (bfv/keygen.go: NTT + Montgomery form over Q||P);
 * p0, p1 <- the two key-switched polys over Q, coefficient domain.  cx, p0 and p1 must be distinct polys. */
int lr_bfv_switch_keys(lr_ckks_plan *plan, const lr_poly *cx, const lr_poly *evk, lr_poly *p0, lr_poly *p1);
/* bfv.evaluator.Relinearize of a degree-2 ciphertext (bfv/evaluator.go:480-501, 512-524): (out0, out1) = (c0 + p0, c1 + p1) with
 * (p0, p1) = switchKeys(c2, evakey.evakey[0]); every poly over Q, coefficient domain; out0 / out1 may be c0 / c1. */
int lr_bfv_relinearize(lr_ckks_plan *plan, const lr_poly *c0, const lr_poly *c1, const lr_poly *c2, const lr_poly *evk,
                       lr_poly *out0, lr_poly *out1);
/* bfv.evaluator.permute (bfv/evaluator.go:711-735) = RotateRows (:670) with gen = galElRotRow, RotateColumns (:579) with the key of
 * that rotation and gen = galElRotColLeft[k], and one step of rotateColumnsPow2 (:636): Context.Permute of both components by the
 * Galois element `gen` (coefficient domain), switchKeys of the second with `rotkey`, Add + Copy.  Every poly over Q, coefficient
 * domain; (out_c0, out_c1) may be (c0, c1); out_c0 != out_c1. */
int lr_bfv_rotate(lr_ckks_plan *plan, const lr_poly *c0, const lr_poly *c1, uint64_t gen, const lr_poly *rotkey,
                  lr_poly *out_c0, lr_poly *out_c1);
/* MulRelin (ckks/evaluator.go:1016), ciphertext x ciphertext, with evaluation key.
 * ct0_c0/ct0_c1 etc. are the degree-0/1 components (Ciphertext.Value()[0], [1]). */
int lr_ckks_mulrelin(lr_ckks_plan *plan, int level, const lr_poly *ct0_c0, const lr_poly *ct0_c1,
                     const lr_poly *ct1_c0, const lr_poly *ct1_c1, const lr_poly *evk,
                     lr_poly *out_c0, lr_poly *out_c1);
/* Batcher for the reference's concurrency model -- one evaluator per goroutine, one ciphertext per call
 * (examples/dbfv/psi/psi.go:215-233, examples/ckks: each worker holds its own ckks.Evaluator).  Concurrent calls of
 * lr_ckks_batcher_mulrelin from any number of host threads are merged into batched launches: a call queues its request, and whichever
 * caller finds a free lane runs everything queued with the same (level, evk) -- up to max_batch polys, in arrival order -- as ONE
 * MulRelin (the operands are read in place through a pointer table, the results are copied out to the callers' polys by one kernel),
 * waits for it and wakes the others.  The call returns when its own result is complete on the device (it may be used on any
 * stream afterwards); operands are synchronised with the streams of the contexts they were created on before they are queued.
 * plans[i]: one plan per lane, each over its OWN pair of contexts, same moduli, device and max_batch; the batcher creates one
 * stream per lane and sets it on the lane's contexts (lr_context_set_stream; destroy puts the library's stream back).  The lanes'
 * streams are created in different priority classes (lane 0 the device's greatest priority, lane 1 its least, lane 2 the default,
 * ...): the runtime binds a stream to a hardware queue of its class, and two lanes on one queue would run one after the other.  Two lanes
 * let the next batch's launches overlap the running one.  The plans and contexts stay owned by the caller and must outlive the
 * batcher; while it exists they are not used for anything else.  evk must be the SAME key image handle in the calls that are to
 * share a batch (the relinearisation key is shared by the evaluators of one party, ckks/evaluator.go:1016).
 * Results are the same bits as lr_ckks_mulrelin's.  An error of a batch is returned by every call that was part of it. */
typedef struct lr_ckks_batcher lr_ckks_batcher;
int lr_ckks_batcher_create(lr_ckks_plan *const *plans, int n_lanes, lr_ckks_batcher **out);
void lr_ckks_batcher_destroy(lr_ckks_batcher *batcher);
int lr_ckks_batcher_mulrelin(lr_ckks_batcher *batcher, int level, const lr_poly *ct0_c0, const lr_poly *ct0_c1,
                             const lr_poly *ct1_c0, const lr_poly *ct1_c1, const lr_poly *evk, lr_poly *out_c0, lr_poly *out_c1);
/* The same for evaluator.permuteNTT (RotateColumns with the key of that rotation, Conjugate; ckks/evaluator.go:1448): calls with the
 * same (level, Galois element, key image) in flight together run as one batched lr_ckks_rotate.  Same bits as lr_ckks_rotate; the
 * outputs may be the inputs.  MulRelin and rotation requests never share a launch; they take the lanes in arrival order. */
int lr_ckks_batcher_rotate(lr_ckks_batcher *batcher, int level, const lr_poly *ct_c0, const lr_poly *ct_c1, uint64_t galois_element,
                           const lr_poly *rotkey, lr_poly *out_c0, lr_poly *out_c1);
/* launches so far, polys they carried, the largest batch (any of the pointers may be NULL) */
int lr_ckks_batcher_stats(lr_ckks_batcher *batcher, uint64_t *batches, uint64_t *products, int *largest);
/* MulRelin with evakey == nil (ckks/evaluator.go:1038-1111): the degree-2 result (out_c0, out_c1, out_c2), no key switch.
 * Outputs may alias the inputs (ctOut == ct0 / ct1: the reference goes through its pools and copies, :1105-1111); ct0 == ct1 is
 * the squaring branch (:1083-1088), whose result equals the regular branch's on the same operands. */
int lr_ckks_mul_norelin(lr_ckks_plan *plan, int level, const lr_poly *ct0_c0, const lr_poly *ct0_c1,
                        const lr_poly *ct1_c0, const lr_poly *ct1_c1, lr_poly *out_c0, lr_poly *out_c1, lr_poly *out_c2);
/* MulRelin, plaintext x ciphertext branch (ckks/evaluator.go:1113-1131): out_ck = MRed(MForm(pt), ct_ck).  pt: the plaintext's
 * value (NTT domain), batch 1 (broadcast) or the ciphertexts' batch. */
int lr_ckks_mul_plain(lr_ckks_plan *plan, int level, const lr_poly *pt, const lr_poly *ct_c0, const lr_poly *ct_c1,
                      lr_poly *out_c0, lr_poly *out_c1);
/* pkEncryptor.encrypt, the branch through the special primes, after the sampling (ckks/encryptor.go:205-234):
 * ct = ModDownPQ(InvNTT(u * pk_k) + e_k) -> NTT, + pt on component 0.  u (SampleTernaryMontgomeryNTT, :206), pk0 / pk1 and
 * e0 / e1 (the residues gaussianSampler.SampleAndAdd adds, coefficient domain, values in [0, q]) hold |Q|+|P| limbs in contextQP's
 * order; pk may have batch 1.  pt over Q[0..level], NTT domain.  Sampling stays on the host (out of scope).
 * The reference's Context.NTT at :229 walks every modulus of contextQ (a Go index panic for level < |Q|-1); here limbs 0..level. */
int lr_ckks_encrypt_pk(lr_ckks_plan *plan, int level, const lr_poly *u, const lr_poly *pk0, const lr_poly *pk1,
                       const lr_poly *e0, const lr_poly *e1, const lr_poly *pt, lr_poly *out_c0, lr_poly *out_c1);
/* decryptor.Decrypt (ckks/decryptor.go:53-78): Horner evaluation of ct[0..degree] at the secret key (NTT + Montgomery form,
 * batch 1 or the ciphertexts' batch) with the reference's lazy-reduction cadence; pt_out over Q[0..level]. */
int lr_ckks_decrypt(lr_ckks_plan *plan, int level, const lr_poly *const *ct, int degree, const lr_poly *sk, lr_poly *pt_out);
/* Rescale, one level (ckks/evaluator.go:933-968 inner loop): DivRoundByLastModulusNTT on both components. */
int lr_ckks_rescale(lr_ckks_plan *plan, lr_poly *c0, lr_poly *c1);
/* permuteNTT (ckks/evaluator.go:1448-1468), the body of RotateColumns with a specific rotation key (:1222) and of
 * Conjugate (:1446): both components are permuted by the Galois element `gen` (what ring.PermuteNTTIndex turns
 * into RotationKeys.permuteNTTLeftIndex[k] / permuteNTTConjugateIndex), the second is key-switched with `rotkey`
 * (a SwitchingKey image as for lr_ckks_switch_keys).  out may alias the input. */
int lr_ckks_rotate(lr_ckks_plan *plan, int level, const lr_poly *c0, const lr_poly *c1, uint64_t gen,
                   const lr_poly *rotkey, lr_poly *out_c0, lr_poly *out_c1);
/* RotateHoisted + switchKeyHoisted (ckks/evaluator.go:1252-1391): n_rot rotations of one ciphertext share the
 * digit decomposition of its second component.  gens[r], rotkeys[r] -> (outs_c0[r], outs_c1[r]); not in place. */
int lr_ckks_rotate_hoisted(lr_ckks_plan *plan, int level, const lr_poly *c0, const lr_poly *c1, int n_rot,
                           const uint64_t *gens, const lr_poly *const *rotkeys, lr_poly *const *outs_c0,
                           lr_poly *const *outs_c1);

/* bfv.Evaluator.Mul = tensorAndRescale (bfv/evaluator.go:467,278) for two degree-1 ciphertexts, device-resident
 * for a whole batch.  lr_bfv_plan owns what bfv.NewEvaluator builds for it: baseconverterQ1Q2 =
 * NewFastBasisExtender(contextQ, contextQMul), pHalf = (prod QMul) >> 1 and the scratch pools (bfv/evaluator.go:89-112). */
typedef struct lr_bfv_plan lr_bfv_plan;
int lr_bfv_plan_create(lr_context *ctxQ, lr_context *ctxQMul, uint64_t t, int max_batch, lr_bfv_plan **out);
/* the same with explicit options; NULL = the options of ctxQ */
int lr_bfv_plan_create_ex(lr_context *ctxQ, lr_context *ctxQMul, uint64_t t, int max_batch, const lr_options *opt, lr_bfv_plan **out);
int lr_bfv_plan_destroy(lr_bfv_plan *plan);
/* operands and results over Q in the coefficient domain, as BFV ciphertexts are; out has degree 2.  ct0 == ct1 (the same two handles: the
 * reference's squaring case, bfv/evaluator.go:306,334-349) lifts and transforms the operand once; outputs may be operands. */
int lr_bfv_mul(lr_bfv_plan *plan, const lr_poly *ct0_c0, const lr_poly *ct0_c1, const lr_poly *ct1_c0,
               const lr_poly *ct1_c1, lr_poly *out_c0, lr_poly *out_c1, lr_poly *out_c2);

/* The batcher for the workload the reference itself pools: every task of examples/dbfv/psi/psi.go:215-233 runs evaluator.Mul and
 * evaluator.Relinearize on one BFV ciphertext pair.  Concurrent calls from any number of host threads are merged into batched launches
 * as by lr_ckks_batcher: a call queues its request, whichever caller finds a free lane runs everything queued of the same kind (and, for
 * Relinearize, the same key image handle) -- up to max_batch polys, in arrival order -- and wakes the others; the call returns when its own
 * result is complete on the device.  The callers' polys are read and written in place through a pointer table (one gather and one
 * scatter kernel around the staged pipeline).  mul_plans[i] / ks_plans[i]: one lr_bfv_plan over (contextQ_i, contextQMul_i) and one
 * lr_ckks_plan over the SAME contextQ_i and (contextQ_i, contextP_i) per lane -- the two halves of what bfv.NewEvaluator builds
 * (bfv/evaluator.go:89-112) -- each lane over its own contexts, same moduli, device, t and max_batch; ks_plans may be NULL (no
 * Relinearize).  The batcher sets one stream per lane on the lane's contexts; plans and contexts stay owned by the caller, must outlive
 * the batcher and are used for nothing else while it exists (destroying one first is LR_ERR_ARG).  Same bits as lr_bfv_mul /
 * lr_bfv_relinearize; an error of a batch is returned by every call that was part of it. */
typedef struct lr_bfv_batcher lr_bfv_batcher;
int lr_bfv_batcher_create(lr_bfv_plan *const *mul_plans, lr_ckks_plan *const *ks_plans, int n_lanes, lr_bfv_batcher **out);
void lr_bfv_batcher_destroy(lr_bfv_batcher *batcher);
/* evaluator.Mul (bfv/evaluator.go:467) of two degree-1 ciphertexts: coefficient domain, degree-2 result */
int lr_bfv_batcher_mul(lr_bfv_batcher *batcher, const lr_poly *ct0_c0, const lr_poly *ct0_c1, const lr_poly *ct1_c0, const lr_poly *ct1_c1,
                       lr_poly *out_c0, lr_poly *out_c1, lr_poly *out_c2);
/* evaluator.Relinearize (bfv/evaluator.go:512) of a degree-2 ciphertext with the key image evk (one handle for all callers) */
int lr_bfv_batcher_relinearize(lr_bfv_batcher *batcher, const lr_poly *c0, const lr_poly *c1, const lr_poly *c2, const lr_poly *evk,
                               lr_poly *out0, lr_poly *out1);
int lr_bfv_batcher_stats(lr_bfv_batcher *batcher, uint64_t *batches, uint64_t *products, int *largest);

/* ------------------------------------------------------------------ multi-device ------ */
/* a batch of independent ciphertexts shards across the GPUs of a node by contiguous blocks (replicated contexts, tables
 * and keys, created per device with lr_context_create(..., device, ...)); nothing crosses devices but finished results.  The reference's
 * parallel model is goroutines in ONE process, one evaluator each (examples/dbfv/psi/psi.go:215-233): here one host thread per device,
 * every handle bound to its device, and these three entry points for the exchange -- no second process, no collective library.
 *
 * lr_poly_copy_peer: polys [src_index, src_index + count) of src -> slots [dst_index, ...) of dst (same N and limb count; any two devices
 * of the process, or the same one).  Asynchronous and ordered on the devices: the copy runs on a copy stream the library keeps per
 * (destination device, source device) pair -- xGMI is point-to-point, so the copies from different peers into one root use different links
 * at once -- behind an event recorded NOW on src_ctx's stream (it waits for everything enqueued through src_ctx before this call, e.g. the
 * kernels that produce the chunk), and overlaps whatever src_ctx is given next.  Nothing is enqueued on dst_ctx's stream: the consumer
 * calls lr_context_wait_peer_copies(dst_ctx) when it wants to read -- dst_ctx's stream then waits (on the device) for every copy into its
 * device enqueued so far.  Callable from any thread; the source must not be overwritten before the copy has run (the caller's ordering:
 * later work of src_ctx on those polys must follow an lr_context_wait_peer_copies / lr_context_sync of the consumer), and the destination
 * slots must not be in use by work still queued on dst_ctx when the call is made (the copy is not ordered behind dst_ctx's stream, so that a
 * root which computes its own share into other slots of the same poly keeps overlapping with the incoming copies).
 *
 * lr_gather_blocks: the whole gather in one call: block r = the first counts[r] polys of srcs[r], placed in dst one behind the other in
 * block order (= global unit order under contiguous-block sharding), then lr_context_wait_peer_copies(dst_ctx).  A producer that works in
 * chunks calls lr_poly_copy_peer per chunk instead and overlaps the copies with the next chunk's kernels (tools/multi_gpu_bench.cpp). */
int lr_poly_copy_peer(lr_context *dst_ctx, lr_poly *dst, int dst_index, lr_context *src_ctx, const lr_poly *src, int src_index, int count);
int lr_context_wait_peer_copies(lr_context *ctx);
int lr_gather_blocks(lr_context *dst_ctx, lr_poly *dst, lr_context *const *src_ctxs, const lr_poly *const *srcs, const int *counts, int n_blocks);

/* ------------------------------------------------------------------ measurement ------- */
/* HIP events on the context's stream (bench.py's roofline leg). */
int lr_timer_start(lr_context *ctx);
int lr_timer_stop(lr_context *ctx, float *elapsed_ms);   /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif /* LATTIGO_RING_H */
