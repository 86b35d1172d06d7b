// lr_device.hpp -- device-side parameter blocks and launch plumbing shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include "lr_arith.hpp"
#include "lr_float128.hpp"

namespace lr {

// per-modulus constants, one entry per limb of a context (device array)
struct LimbParams {
    u64 q;
    u64 qinv;        // q^-1 mod 2^64              (mredParams)
    u64 bred_hi;     // floor(2^128/q) >> 64       (bredParams[0])
    u64 bred_lo;     //                            (bredParams[1])
    u64 n_inv_mont;  // Go's nttNInv (Montgomery form of N^-1)
    u64 n_inv;       // N^-1 mod q, plain
    u64 n_inv_shoup; // floor(n_inv * 2^64 / q)
    // quotient estimate for values < 2^64 when q >= 2^57 (lr_ntt.hip, est_quotient): with
    // qh = (q >> 32) + 1, red_m = min(floor(2^(32+red_g) / qh), 2^32-1), red_g = bitlen(qh) - 1
    u32 red_m;
    u32 red_g;
};

constexpr int kMaxLimbs = 64;

// The library's configuration, one copy per handle, fixed when the handle is created: the internal image of the public lr_options
// (include/lattigo_ring.h; from_public / to_public in lr_abi_core.cpp).  A deployment configures through lr_options and the *_create_ex
// entry points; the LR_* environment variables named below are a TEST-ONLY override of the same fields, read in exactly one place
// (apply_env) at handle creation -- a caller's environment cannot change the code path of a live handle in the middle of a run.
struct Options {
    bool no_asm = false;           // LR_NO_ASM: C++ NTT kernels only
    bool no_fp = false;            // LR_NO_FP: integer bodies for every modulus
    bool no_epilogue = false;      // LR_NO_EPILOGUE: separate subtract-multiply instead of the forward kernels' epilogue
    bool no_int_epilogue = false;  // LR_NO_INT_EPILOGUE: the epilogue on the FP64 bodies only (limbs of 2^46 and more keep the separate pass)
    bool rescale_unfused = false;  // LR_RESCALE_UNFUSED: the rounding rescale with explicit shifted copies
    bool no_staging = false;       // LR_NO_STAGING: N = 2^16 key switch with in-place forward transforms
    bool ext_narrow = false;       // LR_EXT_NARROW: per-term basis extension instead of the 128-bit column sums
    bool asm14_1024 = false;       // LR_ASM_14_1024: the 1024-thread plan at N = 2^14
    bool timeline = false;         // LR_NTT_TIMELINE: plain 2^15 launches (forward / inverse, integer variant 1 and dual variant 3) run the stamped diagnostics builds
    bool keymac_narrow = false;    // LR_KEYMAC_NARROW: one Montgomery product per term in the key inner product instead of the 128-bit sums
    bool no_invfuse = false;       // LR_NO_INVFUSE: N = 2^16 inverse transforms as lazy sub-blocks + the separate last-stage pass (ntt_top_kernel) instead of the pair-flag kernels
    bool no_wide14_small = false;  // LR_ASM_14_NO_WIDE_SMALL: the 512-thread plan at N = 2^14 for small launches too
    bool no_invtop = false;        // LR_NO_INVTOP: the inverse transforms in front of a top-stage extension finish with their own last-stage pass
    bool no_ext_chunks = false;    // LR_NO_EXT_CHUNKS: a basis extension of a small batch as one launch over all target columns instead of column ranges on grid z
    bool no_pair = false;          // LR_NO_PAIR: ModDown's two components of a single ciphertext as two launches instead of one with distance strides
    bool rescale_unpaired = false; // LR_RESCALE_UNPAIRED: lr_ckks_rescale divides the two components one after the other at every batch size
    int split15 = -1;              // LR_NTT_SPLIT15: N = 2^15 transforms as two 2^14 sub-blocks: 0 never, 1 always, unset = launches of at most kSplit15Below workgroups
    bool no_fork = false;          // LR_NO_FORK: the key switch's independent launches in order on one stream at every batch size
    bool no_ext_group = false;     // LR_NO_EXT_GROUP: one extension launch per key-switch digit instead of one grouped launch
    bool no_exttop = false;        // LR_NO_EXTTOP: N = 2^16 key switch with staged extensions and fused-top transforms instead of the top stage inside the extension
    int persist = -1;              // LR_NTT_PERSIST: polys per workgroup of the persistent forward 2^15 kernels (0 = one-poly workgroups, -1 = default)
    int stagger = -1;              // LR_NTT_STAGGER: start-up stagger of the assembly NTT kernels in kilo-clocks per step (0 = off)
    int ntt_mode = -1;             // LR_NTT_MODE
    int asm_variant = -1;          // LR_ASM_VARIANT
    bool ext_ieee_div = false;     // LR_EXT_IEEE_DIV: IEEE division in the extension's float correction (the reference-shaped kernel)
    bool no_grid_padding = false;  // LR_NTT_NO_GRID_PADDING: assembly launches with the limb count on grid x unpadded
    bool bfv_no_ext_epilogue = false;  // LR_BFV_NO_EXT_EPILOGUE
    bool bfv_no_gather = false;    // LR_BFV_NO_GATHER
    // launch-shape thresholds (measured defaults: DESIGN.md decision table)
    int split15_max_workgroups = 128;   // LR_NTT_SPLIT15_BELOW: N = 2^15 launches of at most this many workgroups run as 2^14 sub-blocks
    int wide14_max_items = 256;         // N = 2^14 launches of at most this many transforms use the 1024-thread kernels
    int pair_max_workgroups = 256;      // two components of one ciphertext / the Q and P parts of one inner product as one launch up to here
    int fork_below_workgroups = 256;    // LR_FORK_BELOW: a lone plan forks a launch below this many workgroups
    long long bfv_gather_below = 1536;  // LR_BFV_GATHER_BELOW
    void apply_env();              // the test-only override: the ONE place that reads LR_* variables
    static Options from_env() {    // defaults + the override: what the plain *_create entry points use
        Options o;
        o.apply_env();
        return o;
    }
};
// small per-limb host values travelling in the kernel-argument segment (no host->device copy)
struct LimbScalars { u64 v[kMaxLimbs]; };

// a twiddle factor in the kernels' internal form: x = psi power (plain domain), y = floor(x*2^64/q)
typedef ulonglong2 Twiddle;

constexpr u64 kFpLimit = 1ull << 46;
// per-limb constants of the FP64 butterflies (moduli below 2^46); q = 0.0: the limb stays on the integer body
struct FpLimb {
    double q, q_inv;            // q and RN(1/q)
    double n_inv, n_inv_q;      // N^-1 mod q and RN(n_inv / q): the scaling of the inverse transform
};

// constant of the forward kernels' epilogue, 16 bytes per limb: (c, RN(c / q)) as doubles for a limb the FP64 body takes, the Shoup pair
// (c, floor(c 2^64 / q)) as two u64 in the same bytes for a limb on an integer body (make_epi_limb, lr_abi_ring.cpp)
struct EpiLimb { double c, c_over_q; };

// Addressing of one NTT launch.  Work item (b, i): batch element b, i-th limb of the launch.
struct NttLaunch {
    const u64 *in;
    u64 *out;
    long long in_poly_stride;   // u64 elements between consecutive batch polys
    long long out_poly_stride;
    int in_limb0, in_limb_step;   // input row  = in_limb0  + i * in_limb_step
    int out_limb0, out_limb_step; // output row = out_limb0 + i * out_limb_step
    int mod0, mod_step;           // modulus    = mod0      + i * mod_step
    int n_items;                  // limbs per poly in this launch
    int batch;
    const LimbParams *lp;       // [L]
    const Twiddle *tw;          // [L][N] forward or inverse table
    const Twiddle *tw_fin;      // [L][15][N/16] lane-transposed copy of the last four stages (N >= 2^12), or null
    int sub_log;                // log2(sub-blocks per limb): 0, or 1 when N = 2^16 runs as two 2^15 sub-transforms
    // digit groups (key switching): batch = groups * group polys; the polys of group g skip the items
    // [g*hole, (g+1)*hole): item = i + (i >= g*hole ? hole : 0), i < n_items.  hole = 0: plain launch.
    int hole;
    int group;
    int fuse_top;               // sub_log = 1, forward, out of place: the sub-transforms compute the stage over bit 15 while
                                // loading (each reads both halves of the limb), no separate streaming pass.
                                // Persistent kernels (lr_ntt_fwd15p_*, sub_log = 0): polys per workgroup; grid y (x for the dual
                                // kernels) counts chunks of that many polys inside a group, and `group` is set for plain launches too
    // dual assembly kernels ("m3"): a limb whose fp_lp entry is set runs on the FP64 body, with twiddles (w, RN(w/q)) as
    // doubles in tables laid out exactly like tw / tw_fin, found at tw + fp_tw_delta / tw_fin + fp_fin_delta (bytes)
    long long fp_tw_delta, fp_fin_delta;
    const FpLimb *fp_lp;        // [L], or null for every other kernel
    // epilogue kernels ("m4", forward, FP64 limbs only): out = (x - NTT(in)) * c + plus (mod q), x and plus laid out like the
    // output rows (row = out_limb0 + i * out_limb_step) with their own strides between polys
    const u64 *epi_x;
    long long epi_x_stride;
    const u64 *epi_plus;
    long long epi_plus_stride;
    const EpiLimb *epi_consts;  // [L]: per limb (c, RN(c / q)) as doubles or the Shoup pair of c (EpiLimb), indexed like lp
    // assembly kernels: start-up stagger of the first round of workgroups (gen_ntt.py: stagger), set by the launcher
    int stagger_gx;             // the grid's x extent (linear workgroup id = x + stagger_gx * y)
    int stagger_unit;           // kilo-clocks per step of the 16-step start offset; 0 = all workgroups start at once
};
static_assert(sizeof(NttLaunch) == 176, "NttLaunch layout is shared with asmgen/gen_ntt.py (fields are read at fixed offsets)");

// ---- coefficient-wise launches (lr_ewise.hip) ----
struct EwiseLaunch {
    const u64 *a;
    const u64 *b;
    u64 *out;
    long long a_stride, b_stride, out_stride;  // u64 elements between batch polys (0 = broadcast)
    int n;                                      // ring degree
    const LimbParams *lp;
    int has_scalars;
    LimbScalars scalars;                        // per limb, pre-processed per op on the host
};

// out = MRed(a + (q - b), consts[limb])
struct SubMulLaunch {
    const u64 *a;
    const u64 *b;
    u64 *out;
    long long a_stride, b_stride, out_stride;
    long long b_row_stride;  // n, or 0 when every limb subtracts the same row (the last limb, ring_scaling.go:41)
    int n;
    const LimbParams *lp;
    const u64 *consts;  // device [limbs], Montgomery form
    int reduce_b;       // 1: b is first reduced with BRedAdd (coefficient-domain rescale, ring_scaling.go:50,146)
    LimbScalars addend; // b + addend[limb] before the reduction (pHalfNegQi); zeros when unused
    const u64 *plus;    // optional: out = CRed(plus + result) (the Context.Add that follows a ModDown in ckks MulRelin)
    long long plus_stride;
    int has_post;       // 1: out = CRed(result + post[limb]) (the AddScalarBigint that follows ModDownSplitedQP in bfv Mul)
    LimbScalars post;
};

// out = MRed(CRed(x + (q - sub[limb])), mul[limb]): SubScalarBigint then MulScalar (bfv/evaluator.go:459,462) in one pass
struct ScalarPairLaunch {
    const u64 *in;
    u64 *out;
    long long in_stride, out_stride;
    int n;
    const LimbParams *lp;
    LimbScalars sub, mul;
};
hipError_t launch_scalar_pair(const ScalarPairLaunch &L, int limbs, int batch, hipStream_t stream);

// degree-2 tensor product of two degree-1 ciphertexts (ckks/evaluator.go:1080-1095) in one pass
struct TensorLaunch {
    const u64 *a0, *a1, *b0, *b1;
    long long a0_stride, a1_stride, b0_stride, b1_stride;
    u64 *c0, *c1, *c2;        // may alias the inputs (every thread reads its four operands before it writes)
    long long c_stride;       // of c0
    int n;
    const LimbParams *lp;
    long long c1_stride, c2_stride;
    const u64 *const *table = nullptr;   // batcher form: the operands of batch poly b are table[4b .. 4b+3] (a0, a1, b0, b1), a*/b* and their strides unused
};
hipError_t launch_tensor(const TensorLaunch &L, int limbs, int batch, hipStream_t stream);

// decryptor.Decrypt (ckks/decryptor.go:53-78) in one pass: Horner evaluation of ct[0..degree] at the secret key with the reference's
// element operations and reduction cadence -- acc = ct[degree]; for i = degree..1: acc = CRed(MRed(acc, sk) + ct[i-1]), BRedAdd when
// i & 7 == 7; a final BRedAdd unless degree & 7 == 7 -- every operand read once, the result written once
constexpr int kHornerMaxDegree = 8;
struct HornerLaunch {
    const u64 *ct[kHornerMaxDegree + 1];
    long long ct_stride[kHornerMaxDegree + 1];
    const u64 *sk;
    long long sk_stride;            // 0: one key for the whole batch
    u64 *out;
    long long out_stride;
    int degree, n;
    const LimbParams *lp;
};
hipError_t launch_horner(const HornerLaunch &L, int limbs, int batch, hipStream_t stream);

// out0 = MRed(a, b0), out1 = MRed(a, b1): the two products of pkEncryptor.encrypt with the public key (ckks/encryptor.go:209-211) reading u once
struct Mul2Launch {
    const u64 *a, *b0, *b1;
    u64 *out0, *out1;
    long long a_stride, b0_stride, b1_stride, out0_stride, out1_stride;     // between batch polys (0 = broadcast)
    int n;
    const LimbParams *lp;
};
hipError_t launch_mul2(const Mul2Launch &L, int limbs, int batch, hipStream_t stream);

// batcher form of a result copy: dst[b] = table[b * per_poly + k] for the per_poly staged polys src[k] (rows [limbs][n], batch stride `stride`)
struct ScatterLaunch {
    const u64 *src[4];          // per_poly <= 4 of them are used
    long long stride;
    u64 *const *table;
    int per_poly, n;
};
hipError_t launch_scatter(const ScatterLaunch &L, int limbs, int batch, hipStream_t stream);
// the other way round (the BFV batcher's operands): dst[k] + b * stride <- table[b * per_poly + k], k < per_poly <= 4
struct GatherLaunch {
    u64 *dst[4];
    long long stride;
    const u64 *const *table;
    int per_poly, n;
};
hipError_t launch_gather(const GatherLaunch &L, int limbs, int batch, hipStream_t stream);

// up to four polys with unrelated addresses copied to / from the slots of one contiguous buffer (BFV Mul at a small batch: the four
// operand polys become one batch of 4 B, the three results leave one batch of 3 B): poly k = z / batch, batch element z % batch
struct MultiCopyLaunch {
    const u64 *src[4];
    long long src_stride[4];
    u64 *dst[4];
    long long dst_stride[4];
    int count, batch, n;
};
hipError_t launch_multicopy(const MultiCopyLaunch &L, int limbs, hipStream_t stream);

// out_row[r] = in + adds[r] (optionally CRed)
struct RowAddLaunch {
    const u64 *in;   // one row per batch poly
    u64 *out;
    long long in_stride, out_stride;
    int n;
    u64 q;           // 0: no reduction
    LimbScalars adds; // per output row
};

// half-vector scalar operations (the constant-by-ciphertext methods of ckks.Evaluator, ckks/evaluator.go:373-830): coefficients
// j < n/2 of limb i take the scalar lo[i], the others hi[i] (in the NTT domain the two halves are the slots' real and
// imaginary... conjugate positions: x^(n/2) acts as +i on one half and -i on the other)
struct HalfScalarLaunch {
    const u64 *in;
    u64 *out;
    long long in_stride, out_stride;
    int n;
    int op;                 // 0: out = CRed(in + s)   1: out = MRed(in, s)   2: out = CRed(out + MRed(in, s))
    const LimbParams *lp;
    LimbScalars lo, hi;
};

// key-switch inner product over all digits (lr_ewise.hip)
struct KeyMacLaunch {
    const u64 *c2;                 // [beta][batch][limbs][N], NTT domain
    long long c2_digit_stride;     // u64 elements between digits
    long long c2_poly_stride;      // between batch polys
    const u64 *key;                // SwitchingKey image [2*beta][key limbs][N] (NTT + Montgomery form)
    long long key_poly_stride;     // between key polys
    int key_limb0;                 // first key limb of this segment (0 for Q, |Q| for P)
    u64 *out0, *out1;
    long long out_stride;          // of out0
    int n, beta;
    const LimbParams *lp;
    // limbs [i*alpha, (i+1)*alpha) of digit i are the NTT-domain input itself (ckks/evaluator.go:1579-1584):
    // read from `own` (poly stride own_stride) instead of a copy inside c2; alpha = 0 disables
    const u64 *own;
    long long own_stride;
    int alpha;
    long long out1_stride;         // of out1 (the two outputs of lr_ckks_switch_keys may have different allocations)
    int tile8;                     // set by the launcher: grid x = poly * 8 + (chunk mod 8) (one XCD per key tile)
    int wide;                      // 1: exact 128-bit sums + one Montgomery reduction per output (needs beta * max q < 2^64, q < 2^61)
    // hoisted rotations (ckks/evaluator.go:1346-1347): the digits are read THROUGH ring.PermuteNTT's index for the Galois element perm_gen
    // (reduced modulo 2N; 0 = as they are) instead of from permuted copies; needs own == nullptr (the digits carry their own limbs)
    unsigned perm_gen;
    int logn;
};
hipError_t launch_keymac(const KeyMacLaunch &L, int limbs, int batch, hipStream_t stream);
// the Q part and the P part of one inner product as ONE launch (grid y = limbs_a + limbs_b; both wide, same beta): two launches of a
// small batch run one after the other although they share nothing; hipErrorNotSupported: launch them separately
struct KeyMacPair {
    KeyMacLaunch a, b;
    int split;                     // grid y < split: a with limb y; otherwise b with limb y - split
};
hipError_t launch_keymac_pair(const KeyMacLaunch &A, int limbs_a, const KeyMacLaunch &B, int limbs_b, int batch, hipStream_t stream);

// Poly.MarshalBinary payload (ring/ring_object.go:146-156,197-207): big-endian words <-> device rows
hipError_t launch_bswap(const u64 *in, u64 *out, size_t words, hipStream_t stream);

// ring/ring_galois.go
struct GaloisLaunch {
    const u64 *in;
    u64 *out;
    long long in_stride, out_stride;
    int n, logn;
    int ntt_domain;    // 1: PermuteNTT gather, 0: Context.Permute scatter with sign
    u64 gen;           // reduced modulo 2N
    const LimbParams *lp;
    const u64 *const *in_table = nullptr;   // batcher form: batch poly b is read from in_table[b] (in / in_stride unused)
};
hipError_t launch_permute(const GaloisLaunch &L, int limbs, int batch, hipStream_t stream);
// Context.MultByMonomial (ring/ring.go:663): out = in * X^shift in Z_q[X]/(X^N+1), shift already reduced modulo 2N;
// like the reference, negated coefficients are q - x without reduction (0 becomes q).  Not in place.
hipError_t launch_monomial(const GaloisLaunch &L, int limbs, int batch, hipStream_t stream);

// SimpleScaler.Scale (ring/ring_scaling.go:275-300): one thread per coefficient reconstructs round(t/Q * x) mod t from all
// limbs of the input (integer parts in Z_t, fractional parts in double-double) and writes it to every limb of the output
struct ScaleLaunch {
    const u64 *in;
    u64 *out;
    long long in_stride, out_stride;   // between batch polys, in words
    const u64 *wi;                     // [limbs_in]
    const double *ti;                  // [limbs_in][2] (hi, lo)
    u64 t, add_param, mul_param;
    int pow2, limbs_in, limbs_out, n;
};
hipError_t launch_simple_scale(const ScaleLaunch &L, int batch, hipStream_t stream);

// ---- basis extension (lr_bext.hip) ----
struct ExtTables {        // device pointers; modupParams of ring_basis_extension.go:19-37
    int nQ, nP;
    const u64 *Q;         // [nQ]
    const u64 *mredQ;     // [nQ]
    const u64 *qib_mont;  // [nQ]
    const u64 *P;         // [nP]
    const u64 *mredP;     // [nP]
    const u64 *bredP_hi;  // [nP]
    const u64 *qispj_mont;// [nQ][nP]
    const u64 *qpj_inv;   // [nP][nQ+1]
    // the same (Q/q_i) mod p_j out of Montgomery form, with the Shoup companion floor(c * 2^64 / p_j): the
    // products y_i * c accumulate lazily and are reduced once, to the same canonical residue
    const ulonglong2 *qispj_shoup;  // [nQ][nP] {c, companion}
    const double *Qrcp;   // [nQ] RN(1 / float64(q_i)): div_by_const
    int fast_div_ok;      // every float64(q_i) satisfies div_by_const's precondition (significand not all ones, q_i >= 2): checked on the
                          // host at creation; 0 sends every extension of this table through the reference-shaped kernel and its IEEE division
    int lazy_terms;       // how many [0,4p) terms, each with one p of the correction v * qpjInv[1], fit in 64 bits
    int exact_terms;      // the same for [0,2p) terms
    int word_barrett;     // every p_j > 2^32: floor(2^64 / p_j) fits one word (ext_sum_kernel's final reduction)
    int wide_ok;          // how many input terms keep n * max q_i below 2^64 (ext_wide_kernel: one Montgomery reduction per group)
    // inputs that are the LAZY outputs of the inverse sub-block kernels (the two halves U, V of a limb before its last Gentleman-Sande
    // stage and the scaling): qib_mont[i] * N^-1 and qib_mont[i] * psi_inv[1] * N^-1 mod q_i, so that MRed(U + V, .) and
    // MRed(U + bound - V, .) are the y_i of the finished coefficients j and j + N/2 -- the last inverse stage costs the extension one
    // addition per coefficient instead of a pass over the rows (ExtLaunch::inv_top, top-stage variant of the sum-form kernel only);
    // null where the table was not given a context (set_inverse_top)
    const u64 *invtop0, *invtop1;
};

constexpr int kExtSegments = 3;   // key-switch digits: rows below the digit, rows above it, the special primes

struct ExtSegment {       // rows [limb0, limb0+count) of `out` receive table columns [col0, col0+count)
    u64 *out;
    long long stride;     // u64 elements between batch polys
    int limb0, col0, count;
    // top-stage variant (N = 2^16 key switch, ext_sum_kernel<.., true>): forward twiddle table of the context that owns the segment's
    // output moduli ([L][N] entries {w, floor(w 2^64 / q)}) and the modulus index of the segment's first row; nullptr = plain extension
    const Twiddle *top_tw;
    int top_mod0;
    // epilogue on the canonical extension value e of a row (target modulus p, table column col), plain (non-top) kernels only:
    //   1: out = CRed(MRed(x + (p - e), c[col]) + s[col])   x = the row of `epi_x` at the output's position: the subtract-multiply of a
    //      ModDown (ring_basis_extension.go:237-239) with the scalar addition that follows it in bfv/evaluator.go:457
    //   2: out = MRed(CRed(e + (p - s[col])), c[col])          bfv/evaluator.go:459,462 (SubScalarBigint, MulScalar)
    int epi_mode;               // 0 = none
    const u64 *epi_x;
    long long epi_x_stride;
    // mode 1 only, optional: x = CRed(x + x2) first -- the Gaussian residues SampleAndAdd adds to the Q rows in front of the ModDown of
    // pkEncryptor.encrypt (ckks/encryptor.go:218-226), read at the output's position like x
    const u64 *epi_x2;
    long long epi_x2_stride;
    const u64 *epi_c, *epi_s;   // device arrays over the table columns (epi_s == nullptr: zeros)
};

struct ExtLaunch {
    ExtTables t;
    const u64 *in;
    long long in_stride;
    int in_limb0;
    int n;
    ExtSegment seg[kExtSegments];   // unused segments have count == 0
    int inv_top;                    // 1: the input rows are lazy inverse sub-block outputs (ExtTables::invtop0 / invtop1)
};

constexpr int kExtGroupMax = 9;    // extensions per grouped launch (the digits of one key switch: beta = 9 at PN16QP1761)
struct ExtGroupLaunch {
    ExtLaunch L[kExtGroupMax];
};

#if defined(__HIPCC__)
// Tables that no kernel writes, read at wave-uniform addresses: through the constant address space, so that the
// compiler keeps using scalar loads after the kernel has started storing (it cannot prove that the stores do not
// alias a plain global pointer and falls back to one vector load round trip per use).
typedef const unsigned long __attribute__((address_space(4))) lr_const_u64;
__device__ __forceinline__ unsigned long ld_const(const unsigned long *p) { return *(lr_const_u64 *)(unsigned long)p; }
__device__ __forceinline__ ulonglong2 ld_const(const ulonglong2 *p) {
    lr_const_u64 *q = (lr_const_u64 *)(unsigned long)p;
    return make_ulonglong2(q[0], q[1]);
}

// streaming access to poly data (each element is read or written once per launch): the `nt` cache policy keeps
// it from displacing the twiddle / key tables in L2 and the Infinity Cache
typedef unsigned long long lr_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 ld_stream(const ulonglong2 *p) {
    const lr_u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const lr_u64x2 *>(p));
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ unsigned long ld_stream(const unsigned long *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(unsigned long *p, unsigned long v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(ulonglong2 *p, ulonglong2 v) {
    lr_u64x2 t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<lr_u64x2 *>(p));
}
#endif

// host launchers (defined next to their kernels)
// mode: lazy-correction cadence of the forward butterflies (lr_ntt.hip): 0 = q < 2^61, 1 = q <= 2^60, 2 = q < 2^57
hipError_t launch_ntt(const NttLaunch &a, int logn, bool inverse, int mode, hipStream_t stream);
// hand-scheduled assembly forward NTT (lr_asm.cpp); N = 2^14 / 2^15, lazy mode 1 only
bool ntt_asm_available(int logn);
// kernel_name (optional, >= 32 bytes): receives the name of the code object that was launched
// stagger: Options::stagger (kilo-clocks per step; 0 = off, -1 = the launcher's default for the kernel)
hipError_t launch_ntt_asm(const NttLaunch &a, int logn, int inverse, int variant, hipStream_t stream, bool wide14 = false,
                          char *kernel_name = nullptr, bool timeline = false, int stagger = -1, int persist = 0, bool pad_grid = true);
hipError_t launch_ntt_asm16(const NttLaunch &a, int inverse, char kind, int variant, hipStream_t stream, char *kernel_name = nullptr,
                            int stagger = -1, int full_logn = 16);
// N = 2^16 helpers (lr_ntt.hip): the streaming stage over bit 15, and whether no input row of a launch is an output row
hipError_t launch_ntt_top(const NttLaunch &a, int inverse, hipStream_t stream, int logn = 16);
// the rescale's three streaming passes between the lazy inverse sub-blocks of the last limb and the forward sub-blocks of the targets (lr_ntt.hip)
hipError_t launch_rescale_mid(const NttLaunch &a, const Twiddle *tw_inv, int last_mod, u64 phalf, int logn, hipStream_t stream);
bool ntt_rows_disjoint(const NttLaunch &a, int logn);
hipError_t launch_ewise(int op, const EwiseLaunch &L, int limbs, int batch, hipStream_t stream);
hipError_t launch_submul(const SubMulLaunch &L, int limbs, int batch, hipStream_t stream);
hipError_t launch_rowadd(const RowAddLaunch &L, int rows, int batch, hipStream_t stream);
hipError_t launch_half_scalar(const HalfScalarLaunch &L, int limbs, int batch, hipStream_t stream);
hipError_t launch_ext(const ExtLaunch &L, int n_in, int batch, hipStream_t stream);
// `count` extensions of the same shape (n_in input limbs each) as one launch; hipErrorNotSupported when the shape has no grouped form
hipError_t launch_ext_group(const ExtLaunch *Ls, int count, int n_in, int batch, hipStream_t stream);
bool ext_top_supported(const ExtTables &t, int n_in, int n);
bool ext_epilogue_supported(const ExtTables &t, int n_in, int n);
hipError_t launch_div_selftest(u64 seed, int blocks, int per_thread, unsigned long long *d_mismatches, hipStream_t stream);

}  // namespace lr
