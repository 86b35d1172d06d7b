// lr_asm.cpp -- loader/launcher of the hand-scheduled gfx950 assembly kernels (asmgen/gen_ntt.py).
//
// The code objects are generated and assembled at build time (build.sh) and embedded in this
// library (lr_asm_blob.cpp).  They implement exactly the forward and inverse NTT of lr_ntt.hip for
// N = 2^14 and 2^15 and every modulus in (2^33, 2^61), in three lazy-correction variants; everything
// else stays on the C++ kernels.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "lr_device.hpp"

struct lr_asm_blob {
    const char *name;
    const unsigned char *data;
    unsigned long size;
};
extern "C" const lr_asm_blob lr_asm_blobs[];
extern "C" const int lr_asm_blob_count;

namespace lr {

namespace {
struct AsmKernels {
    std::map<std::string, hipFunction_t> fn;
    std::vector<hipModule_t> mods;
    bool ok = false;
};

AsmKernels *kernels_for_current_device() {
    static std::mutex mu;
    static std::map<int, AsmKernels> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find(dev);
    if (it != table.end()) return it->second.ok ? &it->second : nullptr;
    AsmKernels &k = table[dev];
    k.ok = true;
    for (int i = 0; i < lr_asm_blob_count && k.ok; ++i) {
        hipModule_t mod = nullptr;
        hipFunction_t fn = nullptr;
        k.ok = hipModuleLoadData(&mod, lr_asm_blobs[i].data) == hipSuccess &&
               hipModuleGetFunction(&fn, mod, lr_asm_blobs[i].name) == hipSuccess;
        if (k.ok) {
            k.mods.push_back(mod);
            k.fn[lr_asm_blobs[i].name] = fn;
        }
    }
    if (!k.ok) (void)hipGetLastError();
    return k.ok ? &k : nullptr;
}
}  // namespace

// Start-up stagger of the first round of workgroups (gen_ntt.py: stagger): kilo-clocks per step of the 16-step offset.
// request: Options::stagger (LR_NTT_STAGGER; -1 = default).  Default OFF: measured on R15 x 256 polys with 0 / 2 / 4 / 5 / 6 / 8 / 12
// kilo-clocks per step (tools/dbg/stagger_sweep.py, profiles/r02/stagger_sweep.txt), every setting was 0-5 % slower than none
// on the integer and on the FP64 bodies alike -- the CUs do not run in a bandwidth convoy that a phase offset could break.
static int stagger_unit(int request, bool one_wg_per_cu, bool dual, unsigned long long workgroups) {
    (void)one_wg_per_cu;
    (void)dual;
    (void)workgroups;
    return request > 0 ? request : 0;
}

bool ntt_asm_available(int logn) { return logn >= 12 && logn <= 16 && kernels_for_current_device() != nullptr; }

// variant = lazy-correction mode of asmgen/gen_ntt.py (forward 0, 1, 2) / gen_intt.py (inverse 0, 1)
// persist > 0 (forward 2^15, variants 0..3, no epilogue): the persistent kernels lr_ntt_fwd15p_*, `persist` polys per workgroup in a loop
// with the next poly's column loads prefetched (gen_ntt.py: persist)
hipError_t launch_ntt_asm(const NttLaunch &a, int logn, int inverse, int variant, hipStream_t stream, bool wide14, char *kernel_name, bool timeline,
                          int stagger, int persist, bool pad_grid) {
    AsmKernels *k = kernels_for_current_device();
    if (!k || logn < 12 || logn > 15) return hipErrorNotSupported;
    if (persist > 0 && (logn != 15 || inverse || variant > 3)) persist = 0;     // (no persistent form of the epilogue kernels m4 / m5)
    // N = 2^12 (256 threads, four columns per thread, 36 KiB LDS image) and N = 2^13, 2^14 (512 threads, two columns,
    // 72 KiB) run several workgroups per CU: one
    // covers the other's load and store phases.  wide14 (Options::asm14_1024) selects the 1024-thread kernels (testing aid).
    const bool x = logn <= 13 || (logn == 14 && !wide14);
    char name[32];
    std::snprintf(name, sizeof name, "lr_ntt_%s%d%s_m%d%s", inverse ? "inv" : "fwd", logn, x ? "x" : persist > 0 ? "p" : "", variant, timeline ? "t" : "");
    auto it = k->fn.find(name);
    if (it == k->fn.end()) return hipErrorNotSupported;
    if (kernel_name) std::snprintf(kernel_name, 32, "%s", name);
    if (a.n_items <= 0 || a.batch <= 0) return hipSuccess;
    const bool swapped = variant == 3 || variant == 4;        // dual kernels: x = polynomial (gen_ntt.py: swap_grid)
    if (!swapped && a.hole == 0 && a.batch > 65535) return hipErrorInvalidValue;   // run_ntt chunks such launches
    NttLaunch args = a;
    size_t size = sizeof(NttLaunch);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    (void)hipGetLastError();
    // grid: limb of the launch, polynomial, z = digit group (NttLaunch::hole / group; plain launches are one group).
    // Consecutive workgroups go to the eight XCDs in turn.  Integer variants: x = limb, so XCD j only sees the limbs j mod 8
    // and keeps their twiddles in its L2.  Dual kernels (variant 3): x = polynomial, every XCD sees every modulus -- a launch
    // that mixes the cheaper FP64 limbs with integer ones otherwise waits for the XCD holding the latter (gen_ntt.py: swap_grid)
    unsigned gy = (unsigned)a.batch, gz = 1;
    if (a.hole > 0) {
        if (a.group <= 0 || a.batch % a.group != 0) return hipErrorInvalidValue;
        gy = (unsigned)a.group;
        gz = (unsigned)(a.batch / a.group);
    }
    const unsigned threads = !x ? 1024 : (logn == 12 || (logn == 13 && !inverse)) ? 256 : 512;
    if (persist > 0) {
        // chunks of `persist` consecutive polys of a group per workgroup; the kernel reads the group size for plain launches too
        args.group = (int)gy;
        args.fuse_top = persist;
        gy = (gy + (unsigned)persist - 1) / (unsigned)persist;
    }
    // x = limb: padded to a multiple of eight when the launch has more limbs than one XCD's L2 holds tables for (32 N bytes per limb
    // against 4 MiB: four limbs at 2^15, eight at 2^14) and their count is not such a multiple -- otherwise workgroup (x, y) lands on
    // XCD (x + n_items * y) mod 8 and every XCD sees every limb in turn.  The fifteen limbs of a rounding rescale at 2^15 put 15 MiB of
    // twiddles and 3.8 MiB of epilogue rows through each 4 MiB L2: 4.16 GB per launch on the fabric against 2.59 GB padded
    // (profiles/r03/rescale_grid_padding.txt; the time moves by 1 % only -- the misses were Infinity Cache hits).  The padding
    // workgroups leave at once (gen_ntt.py: L_limb_ok).  Seven limbs at 2^14 fit the L2 as they are: padded, that launch was 5 % slower.
    unsigned gx_items = (unsigned)a.n_items;
    if (!swapped && !timeline && pad_grid && (long long)a.n_items * (32ll << logn) > (4ll << 20) && (a.n_items & 7) != 0)
        gx_items = ((unsigned)a.n_items + 7u) & ~7u;
    const unsigned gx = swapped ? gy : gx_items, gyy = swapped ? (unsigned)a.n_items : gy;
    args.stagger_gx = (int)gx;
    args.stagger_unit = stagger_unit(stagger, threads == 1024, swapped, (unsigned long long)gx * gyy * gz);
    return hipModuleLaunchKernel(it->second, gx, gyy, gz, threads, 1, 1, 0, stream, nullptr, extra);
}

// N = 2^16 runs as two 2^15 sub-blocks per limb (grid x = 2 * n_items).  kind: 's' = forward with the stage over
// bit 15 fused into the loads (out of place only) / inverse sub-blocks (lazy outputs, ntt_top_kernel follows),
// 'p' = forward sub-blocks after a separate ntt_top_kernel pass (in place allowed).
// full_logn = 15, kind 'h': the same for N = 2^15 as two 2^14 sub-blocks (plain forward / lazy inverse sub-blocks only), which launches
// too small to fill the chip with one workgroup per transform use (run_ntt_launch).
hipError_t launch_ntt_asm16(const NttLaunch &a, int inverse, char kind, int variant, hipStream_t stream, char *kernel_name, int stagger, int full_logn) {
    AsmKernels *k = kernels_for_current_device();
    if (!k) return hipErrorNotSupported;
    char name[32];
    std::snprintf(name, sizeof name, "lr_ntt_%s%d%c_m%d", inverse ? "inv" : "fwd", full_logn, kind, variant);
    auto it = k->fn.find(name);
    if (it == k->fn.end()) return hipErrorNotSupported;
    if (kernel_name) std::snprintf(kernel_name, 32, "%s", name);
    if (a.n_items <= 0 || a.batch <= 0) return hipSuccess;
    NttLaunch args = a;
    args.sub_log = 1;
    size_t size = sizeof(NttLaunch);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    (void)hipGetLastError();
    unsigned gy = (unsigned)a.batch, gz = 1;
    if (a.hole > 0) {
        if (a.group <= 0 || a.batch % a.group != 0) return hipErrorInvalidValue;
        gy = (unsigned)a.group;
        gz = (unsigned)(a.batch / a.group);
    }
    if (gz > 65535u) return hipErrorInvalidValue;
    args.stagger_gx = (int)gy;
    args.stagger_unit = stagger_unit(stagger, true, variant >= 3, (unsigned long long)gy * 2u * (unsigned)a.n_items * gz);
    return hipModuleLaunchKernel(it->second, gy, 2u * (unsigned)a.n_items, gz, 1024, 1, 1, 0, stream, nullptr, extra);
}

}  // namespace lr
