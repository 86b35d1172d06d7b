// Implementation of the host-only HIP stand-in (hip/hip_runtime.h in this directory): test infrastructure for the CPU sanitizers.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

struct hipstub_stream { std::atomic<long> ops{0}; std::mutex m; bool alive = true; };
struct hipstub_event { std::atomic<long> stamp{0}; };

namespace {
std::atomic<long> g_allocs{0}, g_streams{0}, g_events{0}, g_fail_after{-1}, g_fail_bytes{0}, g_clock{0};
thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;
hipError_t set(hipError_t e) {
    if (e != hipSuccess) t_last = e;
    return e;
}
void touch(hipStream_t s) {
    if (!s) return;
    // a stream is used by one host thread at a time in this library (a context's stream; a lane's stream under the lane's `busy` flag):
    // the lock makes a violation of that a TSan-visible ordering instead of silent, and the yield widens the windows between calls
    std::lock_guard<std::mutex> lk(s->m);
    s->ops.fetch_add(1, std::memory_order_relaxed);
    std::this_thread::yield();
}
}  // namespace

extern "C" {
hipError_t hipSetDevice(int device) { return device >= 0 && device < 2 ? (t_device = device, hipSuccess) : set(hipErrorInvalidValue); }
hipError_t hipGetDevice(int *device) { *device = t_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int *count) { *count = 2; return hipSuccess; }       // two "devices": the peer paths have something to cross
hipError_t hipDeviceSynchronize(void) { std::this_thread::yield(); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipGetLastError(void) { hipError_t e = t_last; t_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorUnknown ? "injected failure (hipstub)" : "hipstub error"; }
hipError_t hipMalloc(void **p, size_t bytes) {
    *p = std::malloc(bytes ? bytes : 1);
    if (!*p) return set(hipErrorOutOfMemory);
    g_allocs.fetch_add(1);
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (p) { std::free(p); g_allocs.fetch_sub(1); }
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { if (bytes) std::memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    // injected failure: the call-th host-to-device copy of exactly g_fail_bytes bytes from now (the batcher's pointer table has a size of its own)
    if (g_fail_after.load() >= 0 && (long)bytes == g_fail_bytes.load() && g_fail_after.fetch_sub(1) == 0) return set(hipErrorUnknown);
    touch(s);
    if (bytes) std::memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, size_t bytes, hipStream_t s) { touch(s); if (bytes) std::memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s) {
    touch(s);
    for (size_t r = 0; r < height; ++r) std::memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int value, size_t bytes, hipStream_t s) { touch(s); if (bytes) std::memset(p, value, bytes); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new hipstub_stream(); g_streams.fetch_add(1); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipStreamDestroy(hipStream_t s) { if (s) { delete s; g_streams.fetch_sub(1); } return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) {
    touch(s);
    std::this_thread::sleep_for(std::chrono::microseconds(150));      // a device that takes a moment: arrivals queue up behind a busy lane and get merged
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { if (!e) return set(hipErrorInvalidHandle); touch(s); return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipstub_event(); g_events.fetch_add(1); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { if (!e) return set(hipErrorInvalidHandle); touch(s); e->stamp.store(g_clock.fetch_add(1) + 1); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->stamp.load() - a->stamp.load()); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { if (e) { delete e; g_events.fetch_sub(1); } return hipSuccess; }

void hipstub_fail_memcpy_async_after(long calls, long bytes) { g_fail_bytes.store(bytes); g_fail_after.store(calls); }
long hipstub_live_allocations(void) { return g_allocs.load(); }
long hipstub_live_streams(void) { return g_streams.load(); }
long hipstub_live_events(void) { return g_events.load(); }
}
