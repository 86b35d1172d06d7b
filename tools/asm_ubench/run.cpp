// Loads the probe code objects produced by gen.py and reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    std::ifstream list(dir + "/variants.txt");
    std::string name; long long count;
    void *out; CK(hipMalloc(&out, 1 << 20));
    void *big; CK(hipMalloc(&big, (size_t)5 << 30)); CK(hipMemset(big, 1, (size_t)5 << 30));
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("device clock %d kHz, %d CUs\n", clk_khz, cus);
    fflush(stdout);
    const std::string only = argc > 2 ? argv[2] : "";
    // energy mode (third argument "energy"): every variant runs for about three seconds back to back while rocm-smi is read once in the
    // middle: package power and shader clock next to the instruction rate -> energy per wave-instruction above the idle floor
    const bool energy = argc > 3 && std::string(argv[3]) == "energy";
    while (list >> name >> count) {
        if (!only.empty() && only.find("," + name + ",") == std::string::npos) continue;
        printf("running %s\n", name.c_str());
        fflush(stdout);
        std::ifstream f(dir + "/" + name + ".hsaco", std::ios::binary);
        std::vector<char> img((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        hipModule_t mod; hipFunction_t fn;
        CK(hipModuleLoadData(&mod, img.data()));
        CK(hipModuleGetFunction(&fn, mod, ("probe_" + name).c_str()));
        // the code objects declare the 176-byte kernel-argument block of the NTT kernels (NttLaunch); pass exactly that size
        struct { unsigned long long a, b; void *out; void *in; char pad[144]; } args = {0x0123456789ABCDEFull, 0x0FEDCBA987654321ull, out, big, {0}};
        static_assert(sizeof(args) == 176, "kernarg block");
        size_t size = sizeof(args);
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int rounds = 4;
        if (energy) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                CK(hipModuleLaunchKernel(fn, cus * rounds, 1, 1, 1024, 1, 1, 0, 0, nullptr, extra));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const int reps = (int)(3000.0f / ms) + 1;
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) CK(hipModuleLaunchKernel(fn, cus * rounds, 1, 1, 1024, 1, 1, 0, 0, nullptr, extra));
            CK(hipEventRecord(e1, 0));
            struct timespec ts = {1, 500000000};
            nanosleep(&ts, nullptr);
            std::string smi;
            if (FILE *p = popen("/opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null", "r")) {
                char buf[4096];
                size_t n;
                while ((n = fread(buf, 1, sizeof buf, p)) > 0) smi.append(buf, n);
                pclose(p);
            }
            CK(hipEventSynchronize(e1));
            float total; CK(hipEventElapsedTime(&total, e0, e1));
            auto field = [&](const char *key) {
                const size_t at = smi.find(key);
                if (at == std::string::npos) return std::string("?");
                const size_t q0 = smi.find('"', at + strlen(key) + 1), q1 = smi.find('"', q0 + 1);
                return smi.substr(q0 + 1, q1 - q0 - 1);
            };
            // wave-instructions per second on the whole chip: reps launches x (cus x rounds) workgroups x 16 waves x count
            const double wi = (double)reps * cus * rounds * 16 * (double)count / (total * 1e-3);
            printf("%-14s %7.0f ms  %8.3f G wave-instr/s  power %s W  sclk %s\n", name.c_str(), total, wi / 1e9,
                   field("Package Power (W)\"").c_str(), field("sclk clock speed:\"").c_str());
            fflush(stdout);
            CK(hipModuleUnload(mod));
            continue;
        }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            CK(hipModuleLaunchKernel(fn, cus * rounds, 1, 1, 1024, 1, 1, 0, 0, nullptr, extra));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2 && name.rfind("mem_", 0) == 0) {
                // every workgroup reads 64 items of 256 KiB
                const double bytes = (double)cus * rounds * 64 * 262144;
                printf("%-14s %8.3f ms  %7.1f GB/s aggregate, %6.2f us per 256 KiB item per CU\n", name.c_str(), ms, bytes / (ms * 1e-3) / 1e9,
                       ms * 1e3 / (rounds * 64));
            } else if (rep == 2) {
                // per SIMD: rounds workgroups x 4 waves x count instructions
                const double instr = (double)rounds * 4 * count;
                printf("%-14s %8.3f ms  %6.2f cycles/instr at %d MHz (nominal)\n", name.c_str(), ms, ms * 1e-3 * clk_khz * 1e3 / instr, clk_khz / 1000);
            }
        }
        fflush(stdout);
        CK(hipModuleUnload(mod));
    }
    return 0;
}
