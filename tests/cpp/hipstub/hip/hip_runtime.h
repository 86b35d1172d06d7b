// A HOST-ONLY stand-in for <hip/hip_runtime.h>, for ONE purpose: compiling the host side of the C ABI (lattigo-fhe-by-go_amd/csrc/lr_abi_*.cpp,
// lr_host.hpp, lr_precompute.cpp) with g++ and running its concurrency -- the batcher's queue / lanes / condition variable, the scratch
// pools, the fork bookkeeping, handle lifetimes -- under ThreadSanitizer and AddressSanitizer + UBSan on a CPU (tests/test_host_sanitizers.py;
// GPU sanitizers are not available on the pool).  TEST INFRASTRUCTURE, never part of the product and never a fallback: "device" memory is
// host memory, streams run everything at once in call order, the kernel launchers are the recording stubs of stub_launch.cpp, and no
// arithmetic of the hot path exists here.  Only the API the host code uses is declared.
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
enum {
    hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801, hipErrorUnknown = 999,
    hipErrorPeerAccessAlreadyEnabled = 704, hipErrorInvalidHandle = 400
};
struct hipstub_stream;
struct hipstub_event;
typedef hipstub_stream *hipStream_t;
typedef hipstub_event *hipEvent_t;
typedef void *hipFunction_t;
typedef void *hipModule_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };

struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { ulonglong2 v; v.x = x; v.y = y; return v; }
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

extern "C" {
hipError_t hipSetDevice(int device);
hipError_t hipGetDevice(int *device);
hipError_t hipGetDeviceCount(int *count);
hipError_t hipDeviceSynchronize(void);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipDeviceCanAccessPeer(int *can, int device, int peer);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipGetLastError(void);
const char *hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags = 0);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpyPeerAsync(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t s);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemsetAsync(void *p, int value, size_t bytes, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int priority);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipEventDestroy(hipEvent_t e);

// ---- test controls (not HIP): fault injection and bookkeeping the tests read
void hipstub_fail_memcpy_async_after(long calls, long bytes);   // the call-th hipMemcpyAsync of exactly `bytes` bytes from now fails once with hipErrorUnknown (calls < 0: never)
long hipstub_live_allocations(void);                // hipMalloc / hipHostMalloc blocks not yet freed
long hipstub_live_streams(void);
long hipstub_live_events(void);
}
