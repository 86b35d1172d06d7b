// Recording stand-ins for the kernel launchers of lr_device.hpp (lr_ntt.hip, lr_ewise.hip, lr_bext.hip, lr_asm.cpp), for the CPU-sanitizer
// build of the host side (see hip/hip_runtime.h in this directory).  TEST INFRASTRUCTURE: no arithmetic of the hot path lives here.
//
// What a stub does instead of launching a kernel:
//   * counts the launch;
//   * touches the FIRST and the LAST word of every row the real kernel would read or write, at the address the launch struct names --
//     "device" memory is malloc'ed at its exact size, so a wrong stride, limb offset, batch count or pool size in the host code is an
//     AddressSanitizer report, with the stack of the entry point that built the launch;
//   * carries ONE tag word per poly (word 0 of the first row) along the dataflow of the pipelines the batcher runs -- operand -> tensor /
//     permutation -> the epilogue's `plus` operand -> staged result -> scatter -> the caller's poly -- so that a test can tell that every
//     caller got the result of ITS OWN operands back, whatever batch it was merged into.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <initializer_list>

#include "lattigo_ring.h"
#include "lr_device.hpp"

namespace lr {

std::atomic<unsigned long long> g_stub_launches{0};
namespace {
thread_local volatile u64 t_sink;          // (per thread: the reads are the point, not the sink)
inline void rd(const u64 *p) { t_sink = *p; }
inline void rows_r(const u64 *base, long long poly_stride, long long row0, long long row_step, int rows, int batch, long long n) {
    if (!base) return;
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < rows; ++i) {
            const u64 *r = base + b * poly_stride + (row0 + i * row_step) * n;
            rd(r);
            rd(r + n - 1);
        }
}
inline void rows_w(u64 *base, long long poly_stride, long long row0, long long row_step, int rows, int batch, long long n) {
    if (!base) return;
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < rows; ++i) {
            u64 *r = base + b * poly_stride + (row0 + i * row_step) * n;
            r[n - 1] = r[n - 1];            // (a write at the row's end: extent check; the value is kept)
            if (n > 1) r[1] = r[1];
        }
}
inline int item_of(const NttLaunch &a, int b, int i) {
    if (a.hole <= 0) return i;
    const int g = b / a.group;
    return i + (i >= g * a.hole ? a.hole : 0);
}
hipError_t ntt_like(const NttLaunch &a, long long n, bool epilogue) {
    g_stub_launches.fetch_add(1);
    if (a.n_items <= 0 || a.batch <= 0) return hipSuccess;
    for (int b = 0; b < a.batch; ++b)
        for (int i = 0; i < a.n_items; ++i) {
            const int it = item_of(a, b, i);
            const u64 *in = a.in + b * a.in_poly_stride + ((long long)a.in_limb0 + (long long)it * a.in_limb_step) * n;
            u64 *out = a.out + b * a.out_poly_stride + ((long long)a.out_limb0 + (long long)it * a.out_limb_step) * n;
            const u64 tag = in[0];
            rd(in + n - 1);
            out[n - 1] = out[n - 1];
            if (epilogue) {
                const long long row = (long long)a.out_limb0 + (long long)it * a.out_limb_step;
                rd(a.epi_x + b * a.epi_x_stride + row * n);
                rd(a.epi_x + b * a.epi_x_stride + row * n + n - 1);
                const u64 *plus = a.epi_plus + b * a.epi_plus_stride + row * n;
                rd(plus + n - 1);
                out[0] = plus[0];                       // the tag rides on the `plus` operand (MulRelin's c0 / c1, the rotations' permuted c0)
            } else {
                out[0] = tag;
            }
        }
    return hipSuccess;
}
}  // namespace

bool ntt_asm_available(int logn) { return logn >= 12 && logn <= 16; }
hipError_t launch_ntt(const NttLaunch &a, int logn, bool, int, hipStream_t) { return ntt_like(a, 1ll << logn, false); }
hipError_t launch_ntt_asm(const NttLaunch &a, int logn, int inverse, int variant, hipStream_t, bool, char *kernel_name, bool timeline, int, int, bool) {
    if (logn < 12 || logn > 15) return hipErrorNotSupported;
    if (kernel_name) std::snprintf(kernel_name, 32, "stub_%s%d_m%d", inverse ? "inv" : "fwd", logn, variant);
    return ntt_like(a, 1ll << logn, !timeline && (variant == 4 || variant == 5));
}
hipError_t launch_ntt_asm16(const NttLaunch &a, int inverse, char kind, int variant, hipStream_t, char *kernel_name, int, int full_logn) {
    if (kernel_name) std::snprintf(kernel_name, 32, "stub_%s%d%c_m%d", inverse ? "inv" : "fwd", full_logn, kind, variant);
    return ntt_like(a, 1ll << full_logn, !inverse && (variant == 4 || variant == 5));
}
hipError_t launch_ntt_top(const NttLaunch &a, int, hipStream_t, int logn) { return ntt_like(a, 1ll << logn, false); }
hipError_t launch_rescale_mid(const NttLaunch &a, const Twiddle *, int, u64, int logn, hipStream_t) { return ntt_like(a, 1ll << logn, false); }
bool ntt_rows_disjoint(const NttLaunch &a, int logn) {          // (the predicate of lr_ntt.hip, host arithmetic only)
    const long long n_full = 1ll << logn;
    const long long last = (long long)(a.n_items - 1 + (a.hole > 0 ? a.hole : 0));
    const u64 *in_lo = a.in + (long long)a.in_limb0 * n_full;
    const u64 *in_hi = a.in + (long long)(a.batch - 1) * a.in_poly_stride + ((long long)a.in_limb0 + last * a.in_limb_step + 1) * n_full;
    const u64 *out_lo = a.out + (long long)a.out_limb0 * n_full;
    const u64 *out_hi = a.out + (long long)(a.batch - 1) * a.out_poly_stride + ((long long)a.out_limb0 + last * a.out_limb_step + 1) * n_full;
    if (a.in_poly_stride < 0 || a.out_poly_stride < 0 || a.in_limb_step < 0 || a.out_limb_step < 0) return false;
    return in_hi <= out_lo || out_hi <= in_lo;
}

hipError_t launch_ewise(int op, const EwiseLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    rows_r(L.a, L.a_stride, 0, 1, limbs, batch, L.n);
    if (L.b) rows_r(L.b, L.b_stride, 0, 1, limbs, batch, L.n);
    rows_w(L.out, L.out_stride, 0, 1, limbs, batch, L.n);
    for (int b = 0; b < batch; ++b) {
        u64 &o = L.out[b * L.out_stride];
        const u64 x = L.a[b * L.a_stride], y = L.b ? L.b[b * L.b_stride] : 0;
        o = op == LR_ADD ? x + y : op == LR_COPY ? x : o;
    }
    return hipSuccess;
}
hipError_t launch_submul(const SubMulLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    rows_r(L.a, L.a_stride, 0, 1, limbs, batch, L.n);
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < limbs; ++i) {
            rd(L.b + b * L.b_stride + (long long)i * L.b_row_stride);
            rd(L.b + b * L.b_stride + (long long)i * L.b_row_stride + L.n - 1);
        }
    if (L.plus) rows_r(L.plus, L.plus_stride, 0, 1, limbs, batch, L.n);
    rows_w(L.out, L.out_stride, 0, 1, limbs, batch, L.n);
    for (int b = 0; b < batch; ++b) L.out[b * L.out_stride] = L.plus ? L.plus[b * L.plus_stride] : 0;
    return hipSuccess;
}
hipError_t launch_rowadd(const RowAddLaunch &L, int rows, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    rows_r(L.in, L.in_stride, 0, 0, 1, batch, L.n);
    rows_w(L.out, L.out_stride, 0, 1, rows, batch, L.n);
    return hipSuccess;
}
hipError_t launch_half_scalar(const HalfScalarLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    rows_r(L.in, L.in_stride, 0, 1, limbs, batch, L.n);
    rows_w(L.out, L.out_stride, 0, 1, limbs, batch, L.n);
    return hipSuccess;
}
hipError_t launch_scalar_pair(const ScalarPairLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    rows_r(L.in, L.in_stride, 0, 1, limbs, batch, L.n);
    rows_w(L.out, L.out_stride, 0, 1, limbs, batch, L.n);
    return hipSuccess;
}
hipError_t launch_tensor(const TensorLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    for (int b = 0; b < batch; ++b) {
        const u64 *const *tb = L.table ? L.table + 4 * b : nullptr;
        const u64 *a0 = tb ? tb[0] : L.a0 + b * L.a0_stride, *a1 = tb ? tb[1] : L.a1 + b * L.a1_stride;
        const u64 *b0 = tb ? tb[2] : L.b0 + b * L.b0_stride, *b1 = tb ? tb[3] : L.b1 + b * L.b1_stride;
        for (const u64 *p : {a0, a1, b0, b1}) rows_r(p, 0, 0, 1, limbs, 1, L.n);
        const u64 t0 = a0[0], t1 = a1[0], t2 = b0[0] ^ b1[0];
        rows_w(L.c0 + b * L.c_stride, 0, 0, 1, limbs, 1, L.n);
        rows_w(L.c1 + b * L.c1_stride, 0, 0, 1, limbs, 1, L.n);
        rows_w(L.c2 + b * L.c2_stride, 0, 0, 1, limbs, 1, L.n);
        L.c0[b * L.c_stride] = t0;
        L.c1[b * L.c1_stride] = t1;
        L.c2[b * L.c2_stride] = t2;
    }
    return hipSuccess;
}
hipError_t launch_horner(const HornerLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    for (int i = 0; i <= L.degree; ++i) rows_r(L.ct[i], L.ct_stride[i], 0, 1, limbs, batch, L.n);
    rows_r(L.sk, L.sk_stride, 0, 1, limbs, batch, L.n);
    rows_w(L.out, L.out_stride, 0, 1, limbs, batch, L.n);
    return hipSuccess;
}
hipError_t launch_scatter(const ScatterLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < L.per_poly; ++k) {
            const u64 *s = L.src[k] + b * L.stride;
            u64 *d = L.table[b * L.per_poly + k];
            rows_r(s, 0, 0, 1, limbs, 1, L.n);
            rows_w(d, 0, 0, 1, limbs, 1, L.n);
            d[0] = s[0];
        }
    return hipSuccess;
}
hipError_t launch_mul2(const Mul2Launch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    rows_r(L.a, L.a_stride, 0, 1, limbs, batch, L.n);
    rows_r(L.b0, L.b0_stride, 0, 1, limbs, batch, L.n);
    rows_r(L.b1, L.b1_stride, 0, 1, limbs, batch, L.n);
    rows_w(L.out0, L.out0_stride, 0, 1, limbs, batch, L.n);
    rows_w(L.out1, L.out1_stride, 0, 1, limbs, batch, L.n);
    return hipSuccess;
}
hipError_t launch_gather(const GatherLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < L.per_poly; ++k) {
            const u64 *s = L.table[b * L.per_poly + k];
            u64 *d = L.dst[k] + b * L.stride;
            rows_r(s, 0, 0, 1, limbs, 1, L.n);
            rows_w(d, 0, 0, 1, limbs, 1, L.n);
            d[0] = s[0];
        }
    return hipSuccess;
}
hipError_t launch_multicopy(const MultiCopyLaunch &L, int limbs, hipStream_t) {
    g_stub_launches.fetch_add(1);
    for (int k = 0; k < L.count; ++k) {
        rows_r(L.src[k], L.src_stride[k], 0, 1, limbs, L.batch, L.n);
        rows_w(L.dst[k], L.dst_stride[k], 0, 1, limbs, L.batch, L.n);
        for (int b = 0; b < L.batch; ++b) L.dst[k][b * L.dst_stride[k]] = L.src[k][b * L.src_stride[k]];
    }
    return hipSuccess;
}
static void keymac_touch(const KeyMacLaunch &L, int limbs, int batch) {
    for (int d = 0; d < L.beta; ++d) {
        for (int i = 0; i < limbs; ++i) {
            const bool own = L.own && L.alpha > 0 && i >= d * L.alpha && i < (d + 1) * L.alpha;
            for (int b = 0; b < batch; ++b) {
                const u64 *r = own ? L.own + b * L.own_stride + (long long)i * L.n : L.c2 + d * L.c2_digit_stride + b * L.c2_poly_stride + (long long)i * L.n;
                rd(r);
                rd(r + L.n - 1);
            }
            for (int k = 0; k < 2; ++k) {
                const u64 *kr = L.key + (long long)(2 * d + k) * L.key_poly_stride + (long long)(L.key_limb0 + i) * L.n;
                rd(kr);
                rd(kr + L.n - 1);
            }
        }
    }
    rows_w(L.out0, L.out_stride, 0, 1, limbs, batch, L.n);
    rows_w(L.out1, L.out1_stride, 0, 1, limbs, batch, L.n);
}
hipError_t launch_keymac(const KeyMacLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    keymac_touch(L, limbs, batch);
    return hipSuccess;
}
hipError_t launch_keymac_pair(const KeyMacLaunch &A, int limbs_a, const KeyMacLaunch &B, int limbs_b, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    keymac_touch(A, limbs_a, batch);
    keymac_touch(B, limbs_b, batch);
    return hipSuccess;
}
hipError_t launch_bswap(const u64 *in, u64 *out, size_t words, hipStream_t) {
    g_stub_launches.fetch_add(1);
    if (words) {
        rd(in);
        rd(in + words - 1);
        out[0] = out[0];
        out[words - 1] = out[words - 1];
    }
    return hipSuccess;
}
hipError_t launch_permute(const GaloisLaunch &L, int limbs, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    for (int b = 0; b < batch; ++b) {
        const u64 *in = L.in_table ? L.in_table[b] : L.in + b * L.in_stride;
        rows_r(in, 0, 0, 1, limbs, 1, L.n);
        rows_w(L.out + b * L.out_stride, 0, 0, 1, limbs, 1, L.n);
        L.out[b * L.out_stride] = in[0];
    }
    return hipSuccess;
}
hipError_t launch_monomial(const GaloisLaunch &L, int limbs, int batch, hipStream_t s) { return launch_permute(L, limbs, batch, s); }
hipError_t launch_simple_scale(const ScaleLaunch &L, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    rows_r(L.in, L.in_stride, 0, 1, L.limbs_in, batch, L.n);
    rows_w(L.out, L.out_stride, 0, 1, L.limbs_out, batch, L.n);
    return hipSuccess;
}
static void ext_touch(const ExtLaunch &L, int n_in, int batch) {
    rows_r(L.in, L.in_stride, L.in_limb0, 1, n_in, batch, L.n);
    for (int s = 0; s < kExtSegments; ++s) {
        const ExtSegment &g = L.seg[s];
        if (g.count <= 0) continue;
        if (g.epi_mode && g.epi_x) rows_r(g.epi_x, g.epi_x_stride, g.limb0, 1, g.count, batch, L.n);
        if (g.epi_mode && g.epi_x2) rows_r(g.epi_x2, g.epi_x2_stride, g.limb0, 1, g.count, batch, L.n);
        rows_w(g.out, g.stride, g.limb0, 1, g.count, batch, L.n);
    }
}
hipError_t launch_ext(const ExtLaunch &L, int n_in, int batch, hipStream_t) {
    g_stub_launches.fetch_add(1);
    ext_touch(L, n_in, batch);
    return hipSuccess;
}
hipError_t launch_ext_group(const ExtLaunch *Ls, int count, int n_in, int batch, hipStream_t) {
    if (count > kExtGroupMax) return hipErrorNotSupported;
    g_stub_launches.fetch_add(1);
    for (int i = 0; i < count; ++i) ext_touch(Ls[i], n_in, batch);
    return hipSuccess;
}
bool ext_top_supported(const ExtTables &t, int n_in, int n) {
    return (n & 3) == 0 && n_in >= 1 && n_in <= 8 && t.exact_terms >= 4 && t.fast_div_ok && t.word_barrett && t.lazy_terms >= (n_in < 2 ? 2 : n_in);
}
bool ext_epilogue_supported(const ExtTables &t, int n_in, int n) {
    if ((n & 1) != 0 || t.exact_terms < 4 || !t.fast_div_ok) return false;
    if (t.lazy_terms >= (n_in < 2 ? 2 : n_in) && t.word_barrett) return true;
    if (t.lazy_terms >= n_in) return false;
    return t.wide_ok >= n_in || (t.wide_ok >= 16 && n_in > 16) || (t.wide_ok >= 8 && n_in > 8);
}
hipError_t launch_div_selftest(u64, int, int, unsigned long long *d_mismatches, hipStream_t) {
    g_stub_launches.fetch_add(1);
    *d_mismatches = 0;
    return hipSuccess;
}

}  // namespace lr
