/* A plain C99 caller of include/lattigo_ring.h, shaped like the cgo binding of INTEGRATION.md: Go's Poly.Coeffs is a [][]uint64, so the
 * boundary takes one pointer per limb (lr_poly_upload / lr_poly_download, lr_ntt_host).  Built and run by tests/test_c_abi_caller.py.
 *
 *   ring_abi_caller info                  no device needed: build info, an argument error and its text
 *   ring_abi_caller ntt <file>            file: "N L" / L moduli / L*N input coefficients / L*N expected Context.NTT output (decimal)
 *                                         device path: create, alloc, upload per limb, NTT in place, download, compare; InvNTT back;
 *                                         then the same transform through the host-pointer form lr_ntt_host
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lattigo_ring.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != LR_OK) {                                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lr_last_error_string());                    \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static int info(void) {
    lr_context *ctx = NULL;
    const uint64_t not_a_prime_field[1] = {12};
    int rc;
    printf("build: %s\n", lr_build_info());
    rc = lr_context_create(3, not_a_prime_field, 1, 0, &ctx);      /* N = 3 is not a power of two: ring_context.go:71-73 panics */
    if (rc == LR_OK || ctx != NULL) {
        fprintf(stderr, "an invalid degree was accepted\n");
        return 1;
    }
    printf("invalid degree -> status %d: %s\n", rc, lr_last_error_string());
    if (strlen(lr_last_error_string()) == 0) return 1;
    return 0;
}

static int ntt(const char *path) {
    FILE *f = fopen(path, "r");
    unsigned long long N, L, v;
    uint64_t *moduli, *in, *want, *got;
    const uint64_t **src;
    uint64_t **dst;
    lr_context *ctx = NULL;
    lr_poly *p = NULL;
    size_t i, n;
    if (!f || fscanf(f, "%llu %llu", &N, &L) != 2) return 2;
    n = (size_t)(N * L);
    moduli = malloc(sizeof(uint64_t) * L);
    in = malloc(sizeof(uint64_t) * n);
    want = malloc(sizeof(uint64_t) * n);
    got = malloc(sizeof(uint64_t) * n);
    src = malloc(sizeof(*src) * L);
    dst = malloc(sizeof(*dst) * L);
    for (i = 0; i < L; ++i) { if (fscanf(f, "%llu", &v) != 1) return 2; moduli[i] = v; }
    for (i = 0; i < n; ++i) { if (fscanf(f, "%llu", &v) != 1) return 2; in[i] = v; }
    for (i = 0; i < n; ++i) { if (fscanf(f, "%llu", &v) != 1) return 2; want[i] = v; }
    fclose(f);
    for (i = 0; i < L; ++i) { src[i] = in + i * N; dst[i] = got + i * N; }

    CHECK(lr_context_create(N, moduli, (int)L, 0, &ctx));
    CHECK(lr_poly_alloc(ctx, (int)L, 1, &p));
    CHECK(lr_poly_upload(p, 0, src, (int)L));
    CHECK(lr_ntt(ctx, (int)L - 1, p, p));                           /* context.NTT(p, p), ring/ntt.go:4 */
    CHECK(lr_poly_download(p, 0, dst, (int)L));
    if (memcmp(got, want, sizeof(uint64_t) * n) != 0) { fprintf(stderr, "NTT differs from the expected vector\n"); return 1; }
    CHECK(lr_intt(ctx, (int)L - 1, p, p));
    CHECK(lr_poly_download(p, 0, dst, (int)L));
    for (i = 0; i < n; ++i)
        if (got[i] != in[i] % moduli[i / N]) { fprintf(stderr, "InvNTT(NTT(x)) != x mod q at %zu\n", i); return 1; }
    memset(got, 0, sizeof(uint64_t) * n);
    CHECK(lr_ntt_host(ctx, (int)L - 1, src, dst));                  /* upload -> kernel -> download in one call */
    if (memcmp(got, want, sizeof(uint64_t) * n) != 0) { fprintf(stderr, "lr_ntt_host differs from the expected vector\n"); return 1; }
    CHECK(lr_poly_free(p));
    CHECK(lr_context_destroy(ctx));
    printf("ntt ok: N=%llu L=%llu\n", N, L);
    free(moduli); free(in); free(want); free(got); free(src); free(dst);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "info") == 0) return info();
    if (argc >= 3 && strcmp(argv[1], "ntt") == 0) return ntt(argv[2]);
    fprintf(stderr, "usage: %s info | ntt <file>\n", argv[0]);
    return 2;
}
