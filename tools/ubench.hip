// ubench.hip -- gfx950 VALU micro-benchmarks that decide the NTT butterfly design (hard part 1):
// integer multiply issue rates and the cost of whole modular-multiplication / butterfly variants, all in registers.
// Build: hipcc --offload-arch=gfx950 -O3 -I../lattigo-fhe-by-go_amd/csrc -o ubench ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "lr_arith.hpp"
using namespace lr;

#define CHAINS 8
#define ITERS 2048

template <int OP>
__global__ __launch_bounds__(256) void k_op(u64 *out, u64 a0, u64 b0, u64 q, u64 qinv, u64 w, u64 ws) {
    u64 x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = a0 + threadIdx.x * 977 + c * 131 + blockIdx.x;
    const u32 b32 = (u32)b0 | 1u;
    const u64 q4 = q << 2;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            u64 v = x[c];
            if constexpr (OP == 0) { u32 lo = (u32)v * b32; v = ((u64)(u32)(v >> 32) << 32) | lo; }             // v_mul_lo_u32
            if constexpr (OP == 1) { u32 lo = __umulhi((u32)v, b32) + (u32)it; v = (v & 0xFFFFFFFF00000000ull) | lo; } // v_mul_hi_u32
            if constexpr (OP == 2) { v = (u64)(u32)v * b32 + v; }                                               // v_mad_u64_u32
            if constexpr (OP == 3) { v = v + b0 + (u64)it; }                                                    // 64-bit add
            if constexpr (OP == 4) { v = v * b0 + 1; }                                                          // 64x64 low mul
            if constexpr (OP == 5) { v = __umul64hi(v, b0) + v; }                                               // 64x64 high mul
            if constexpr (OP == 6) { v = mred_constant(v, w, q, qinv); }                                        // Go MRedConstant
            if constexpr (OP == 7) { v = mul_shoup_lazy(v, w, ws, q); }                                         // truncated Shoup
            if constexpr (OP == 8) { v = mul_shoup_exact(v, w, ws, q); }                                        // exact Shoup
            if constexpr (OP == 9) { u32 lo = __umul24((u32)v, b32) + (u32)it; v = (v & 0xFFFFFFFF00000000ull) | lo; } // v_mul_u32_u24
            if constexpr (OP == 10) { double d = __longlong_as_double((long long)v); d = fma(d, 1.0000001, 0.5); v = (u64)__double_as_longlong(d); } // v_fma_f64
            if constexpr (OP == 11) { float f = __uint_as_float((u32)v); f = fmaf(f, 1.0001f, 0.5f); v = (v & 0xFFFFFFFF00000000ull) | __float_as_uint(f); } // v_fma_f32
            x[c] = v;
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// full butterflies on CHAINS independent (U,V) pairs
template <int KIND>
__global__ __launch_bounds__(256) void k_bfly(u64 *out, u64 a0, u64 q, u64 qinv, u64 w, u64 ws) {
    u64 U[CHAINS], V[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { U[c] = (a0 + threadIdx.x * 977 + c) % q; V[c] = (a0 * 3 + threadIdx.x * 13 + c * 7) % q; }
    const u64 q2 = q << 1, q4 = q << 2;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if constexpr (KIND == 0) {  // reference butterfly, ring/ntt.go:32-40
                u64 u = U[c] > q2 ? U[c] - q2 : U[c];
                u64 v = mred_constant(V[c], w, q, qinv);
                U[c] = u + v; V[c] = u + q2 - v;
            } else if constexpr (KIND == 1) {  // ours: truncated Shoup, [0,8q)
                u64 u = U[c] >= q4 ? U[c] - q4 : U[c];
                u64 v = mul_shoup_lazy(V[c], w, ws, q);
                U[c] = u + v; V[c] = u + q4 - v;
            } else {  // exact Shoup, [0,4q)
                u64 u = U[c] >= q2 ? U[c] - q2 : U[c];
                u64 v = mul_shoup_exact(V[c], w, ws, q);
                U[c] = u + v; V[c] = u + q2 - v;
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc ^= U[c] ^ V[c];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const u64 q = 1152921504050839553ull, qinv = 0;  // values only need to be data-dependent
    u64 qi = 1, xx = q; for (int i = 0; i < 63; ++i) { qi *= xx; xx *= xx; }
    const u64 w = 123456789123456789ull % q, ws = (u64)((((unsigned __int128)w) << 64) / q);
    const int blocks = 256 * 8;
    u64 *out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    const char *names[] = {"v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "add_u64", "mul_lo_u64", "mul_hi_u64",
                           "MRedConstant(Go)", "mul_shoup_lazy(9 mul)", "mul_shoup_exact", "v_mul_u32_u24", "v_fma_f64", "v_fma_f32"};
    const double total = (double)blocks * 256 * CHAINS * ITERS;
    printf("{\"ubench\": [\n");
#define RUN(OP) { double ms = time_ms([&] { hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), 0, 0, out, 0x123456789abcdefull, 0x9e3779b97f4a7c15ull, q, qi, w, ws); }); \
      printf("  {\"op\": \"%s\", \"ms\": %.3f, \"Gops_per_s\": %.1f},\n", names[OP], ms, total / ms / 1e6); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    const char *bn[] = {"butterfly_reference(MRedConstant)", "butterfly_shoup_lazy_8q", "butterfly_shoup_exact_4q"};
#define RUNB(K) { double ms = time_ms([&] { hipLaunchKernelGGL(k_bfly<K>, dim3(blocks), dim3(256), 0, 0, out, 0x123456789abcdefull, q, qi, w, ws); }); \
      double bps = total / ms / 1e6; \
      printf("  {\"op\": \"%s\", \"ms\": %.3f, \"Gbutterflies_per_s\": %.1f, \"limb_ntt_2p15_per_s_ceiling\": %.0f, \"equiv_GBs\": %.0f},\n", bn[K], ms, bps, bps * 1e9 / 245760.0, bps * 1e9 / 245760.0 * 524288.0 / 1e9); }
    RUNB(0) RUNB(1) RUNB(2)
    printf("  {}\n]}\n");
    (void)qinv;
    return 0;
}
