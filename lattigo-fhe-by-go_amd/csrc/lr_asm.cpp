// lr_asm.cpp -- loader/launcher of the hand-scheduled gfx950 assembly kernels (asmgen/gen_ntt.py).
//
// The code objects are generated and assembled at build time (build.sh) and embedded in this
// library (lr_asm_blob.cpp).  They implement exactly the forward and inverse NTT of lr_ntt.hip in
// lazy mode 1 (every modulus in [2^57, 2^60]) for N = 2^14 and 2^15; everything else stays on the
// C++ kernels.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include "lr_device.hpp"

extern "C" {
extern const unsigned char lr_hsaco_fwd14[];
extern const unsigned long lr_hsaco_fwd14_size;
extern const unsigned char lr_hsaco_fwd15[];
extern const unsigned long lr_hsaco_fwd15_size;
extern const unsigned char lr_hsaco_inv14[];
extern const unsigned long lr_hsaco_inv14_size;
extern const unsigned char lr_hsaco_inv15[];
extern const unsigned long lr_hsaco_inv15_size;
}

namespace lr {

namespace {
struct AsmKernels {
    hipModule_t mod[4] = {nullptr, nullptr, nullptr, nullptr};
    hipFunction_t fn[4] = {nullptr, nullptr, nullptr, nullptr};   // fwd14, fwd15, inv14, inv15
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
    AsmKernels k;
    const void *image[4] = {lr_hsaco_fwd14, lr_hsaco_fwd15, lr_hsaco_inv14, lr_hsaco_inv15};
    const char *name[4] = {"lr_ntt_fwd14_asm", "lr_ntt_fwd15_asm", "lr_ntt_inv14_asm", "lr_ntt_inv15_asm"};
    k.ok = true;
    for (int i = 0; i < 4 && k.ok; ++i)
        k.ok = hipModuleLoadData(&k.mod[i], image[i]) == hipSuccess &&
               hipModuleGetFunction(&k.fn[i], k.mod[i], name[i]) == hipSuccess;
    if (!k.ok) (void)hipGetLastError();
    table[dev] = k;
    return k.ok ? &table[dev] : nullptr;
}
}  // namespace

bool ntt_asm_available(int logn) { return (logn == 14 || logn == 15) && kernels_for_current_device() != nullptr; }

hipError_t launch_ntt_asm(const NttLaunch &a, int logn, int inverse, hipStream_t stream) {
    AsmKernels *k = kernels_for_current_device();
    if (!k || (logn != 14 && logn != 15)) return hipErrorNotSupported;
    if (a.n_items <= 0 || a.batch <= 0) return hipSuccess;
    if (a.batch > 65535) return hipErrorInvalidValue;
    NttLaunch args = a;
    size_t size = sizeof(NttLaunch);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    (void)hipGetLastError();
    // grid: x = limb of the launch, y = polynomial; blocks b and b+8 share an XCD, so the limbs an XCD
    // sees (and whose twiddles live in its L2) are x mod 8 when n_items is a multiple of 8
    return hipModuleLaunchKernel(k->fn[(inverse ? 2 : 0) + (logn - 14)], (unsigned)a.n_items, (unsigned)a.batch, 1, 1024, 1, 1, 0, stream,
                                 nullptr, extra);
}

}  // namespace lr
