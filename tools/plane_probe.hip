// plane_probe.hip -- VERDICT r03 item 6: would storing the key switch's internal rows (limbs below 2^46) as two planes -- u32 low words + u16
// high halves, four coefficients per lane so that the accesses stay dwordx4 + dwordx2 -- pay?  A streaming probe: the same number of
// coefficients through (a) 8-byte words and (b) the 6-byte plane form, as a read-modify-write pass, a read-only pass (the consumer: a
// transform's loads) and a write-only pass (the producer: an extension's stores).  Prints milliseconds and GB/s of real bytes per variant;
// tools/dbg/plane_probe.py samples package power beside it.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/build/plane_probe tools/plane_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// mode 0: read + write, 1: read only (xor-reduce into one word per thread), 2: write only
template <int MODE>
__global__ __launch_bounds__(256) void words_kernel(const u64 *in, u64 *out, u64 *sink, size_t quads, u64 k) {
    // the product's access shape (lr_ewise.hip, the transforms' column loads): 16 bytes per lane, a wave's accesses contiguous
    u64 acc = 0;
    const size_t pairs = quads * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (size_t)gridDim.x * 256) {
        u64x2 a;
        if (MODE != 2) a = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in) + i);
        else a = u64x2{i, k};
        a.x += k; a.y += k;                              // one 64-bit operation per coefficient stands for the consumer's first use
        if (MODE == 1) acc ^= a.x ^ a.y;
        else __builtin_nontemporal_store(a, reinterpret_cast<u64x2 *>(out) + i);
    }
    if (MODE == 1) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void planes_kernel(const u32 *in_lo, const u32 *in_hi, u32 *out_lo, u32 *out_hi, u64 *sink, size_t quads, u64 k) {
    u64 acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)gridDim.x * 256) {
        u32x4 lo;
        u32x2 hi;
        if (MODE != 2) {
            lo = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in_lo) + i);       // four low words
            hi = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(in_hi) + i);       // four 16-bit high halves
        } else {
            lo = u32x4{(u32)i, (u32)k, (u32)i, (u32)k};
            hi = u32x2{(u32)k, (u32)i};
        }
        u64 v0 = (u64)lo.x | ((u64)(hi.x & 0xFFFFu) << 32), v1 = (u64)lo.y | ((u64)(hi.x >> 16) << 32);
        u64 v2 = (u64)lo.z | ((u64)(hi.y & 0xFFFFu) << 32), v3 = (u64)lo.w | ((u64)(hi.y >> 16) << 32);
        v0 += k; v1 += k; v2 += k; v3 += k;
        if (MODE == 1) acc ^= v0 ^ v1 ^ v2 ^ v3;
        else {
            const u32x4 slo{(u32)v0, (u32)v1, (u32)v2, (u32)v3};
            const u32x2 shi{(u32)((v0 >> 32) & 0xFFFFu) | ((u32)(v1 >> 32) << 16), (u32)((v2 >> 32) & 0xFFFFu) | ((u32)(v3 >> 32) << 16)};
            __builtin_nontemporal_store(slo, reinterpret_cast<u32x4 *>(out_lo) + i);
            __builtin_nontemporal_store(shi, reinterpret_cast<u32x2 *>(out_hi) + i);
        }
    }
    if (MODE == 1) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const size_t coeffs = (size_t)1 << 28;                 // 2 GiB as words, 1.5 GiB as planes: far above the 256 MiB Infinity Cache
    const size_t quads = coeffs / 4;
    const int reps = argc > 1 ? std::atoi(argv[1]) : 10;
    const double seconds = argc > 2 ? std::atof(argv[2]) : 0.0;    // > 0: loop every variant for that long (the power sampler's window)
    u64 *a, *b, *sink;
    u32 *lo, *hi, *lo2, *hi2;
    CK(hipMalloc(&a, coeffs * 8)); CK(hipMalloc(&b, coeffs * 8));
    CK(hipMalloc(&lo, coeffs * 4)); CK(hipMalloc(&hi, coeffs * 2)); CK(hipMalloc(&lo2, coeffs * 4)); CK(hipMalloc(&hi2, coeffs * 2));
    const int blocks = 256 * 16;
    CK(hipMalloc(&sink, (size_t)blocks * 256 * 8));
    CK(hipMemset(a, 1, coeffs * 8)); CK(hipMemset(lo, 1, coeffs * 4)); CK(hipMemset(hi, 1, coeffs * 2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char *name; double bytes; int kind, mode; };
    const V vs[] = {{"words  read+write (16 B / coefficient)", 16.0 * coeffs, 0, 0}, {"planes read+write (12 B / coefficient)", 12.0 * coeffs, 1, 0},
                    {"words  read only  ( 8 B / coefficient)", 8.0 * coeffs, 0, 1}, {"planes read only  ( 6 B / coefficient)", 6.0 * coeffs, 1, 1},
                    {"words  write only ( 8 B / coefficient)", 8.0 * coeffs, 0, 2}, {"planes write only ( 6 B / coefficient)", 6.0 * coeffs, 1, 2}};
    auto launch = [&](const V &v) {
        const u64 k = 0x1234567;
        if (v.kind == 0) {
            if (v.mode == 0) hipLaunchKernelGGL(words_kernel<0>, dim3(blocks), dim3(256), 0, 0, a, b, sink, quads, k);
            if (v.mode == 1) hipLaunchKernelGGL(words_kernel<1>, dim3(blocks), dim3(256), 0, 0, a, b, sink, quads, k);
            if (v.mode == 2) hipLaunchKernelGGL(words_kernel<2>, dim3(blocks), dim3(256), 0, 0, a, b, sink, quads, k);
        } else {
            if (v.mode == 0) hipLaunchKernelGGL(planes_kernel<0>, dim3(blocks), dim3(256), 0, 0, lo, hi, lo2, hi2, sink, quads, k);
            if (v.mode == 1) hipLaunchKernelGGL(planes_kernel<1>, dim3(blocks), dim3(256), 0, 0, lo, hi, lo2, hi2, sink, quads, k);
            if (v.mode == 2) hipLaunchKernelGGL(planes_kernel<2>, dim3(blocks), dim3(256), 0, 0, lo, hi, lo2, hi2, sink, quads, k);
        }
    };
    for (const V &v : vs) {
        for (int w = 0; w < 3; ++w) launch(v);
        CK(hipDeviceSynchronize());
        int n = reps;
        if (seconds > 0) {
            std::printf("BEGIN %s\n", v.name);
            std::fflush(stdout);
        }
        float total = 0;
        int done = 0;
        do {
            CK(hipEventRecord(e0));
            for (int r = 0; r < n; ++r) launch(v);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
            done += n;
        } while (seconds > 0 && total < seconds * 1e3);
        const double ms = total / done;
        std::printf("%s: %.4f ms per pass over 2^28 coefficients, %.0f GB/s of real bytes, %.2f coefficients per ns\n", v.name, ms, v.bytes / ms / 1e6, coeffs / ms / 1e6);
        if (seconds > 0) std::printf("END %s\n", v.name);
        std::fflush(stdout);
    }
    return 0;
}
