// lr_abi_peer.cpp -- C ABI: moving finished polys between the devices of ONE process (the batch shards by independent
// ciphertexts, the only exchange is the gather of results).  The reference's parallel model is goroutines in one process, one evaluator
// each (examples/dbfv/psi/psi.go:215-233); on a multi-GPU node that is one host thread per device, each with its own contexts, and the
// results brought to a root device by direct peer copies: xGMI is point-to-point (seven links into a root), so per-peer copy streams
// fill all of them at once, where a ring collective would be bound by one link and move every block through every device.
#include "lr_host.hpp"

namespace lr_host {

// one copy stream per (destination device, source device), created on the destination device and kept for the life of the process
struct PeerStreams {
    std::mutex mu;
    std::map<std::pair<int, int>, hipStream_t> streams;
    std::map<std::pair<int, int>, bool> peer_enabled;
    int get(int dst_dev, int src_dev, hipStream_t *out) {
        std::lock_guard<std::mutex> lock(mu);
        const auto key = std::make_pair(dst_dev, src_dev);
        auto it = streams.find(key);
        if (it != streams.end()) {
            *out = it->second;
            return LR_OK;
        }
        LR_HIP(hipSetDevice(dst_dev));
        if (dst_dev != src_dev && !peer_enabled[key]) {
            // direct access where the topology allows it (xGMI); without it the runtime stages the copy, which is still correct
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, dst_dev, src_dev) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(src_dev, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
            (void)hipGetLastError();
            peer_enabled[key] = true;
        }
        hipStream_t s = nullptr;
        LR_HIP(create_stream(&s));
        streams[key] = s;
        *out = s;
        return LR_OK;
    }
    // the copy streams that feed `dst_dev`
    std::vector<hipStream_t> into(int dst_dev) {
        std::lock_guard<std::mutex> lock(mu);
        std::vector<hipStream_t> v;
        for (auto &kv : streams)
            if (kv.first.first == dst_dev) v.push_back(kv.second);
        return v;
    }
};

PeerStreams &peer_streams() {
    static PeerStreams *p = new PeerStreams();      // never destroyed: streams outlive every handle, like shared_stream's
    return *p;
}

}  // namespace lr_host

extern "C" int lr_poly_copy_peer(lr_context *dst_ctx, lr_poly *dst, int dst_index, lr_context *src_ctx, const lr_poly *src, int src_index, int count) {
    return guarded([&]() -> int {
    if (!dst_ctx || !dst || !src_ctx || !src) return fail(LR_ERR_ARG, "null argument");
    if (count < 0 || dst_index < 0 || src_index < 0 || (long long)dst_index + count > dst->batch || (long long)src_index + count > src->batch)
        return fail(LR_ERR_SHAPE, "peer copy: poly range outside the source or the destination batch");
    if (dst->N != src->N || dst->limbs != src->limbs) return fail(LR_ERR_SHAPE, "peer copy: ring degree or limb count differ");
    if (dst->device != dst_ctx->device || src->device != src_ctx->device) return fail(LR_ERR_ARG, "peer copy: a poly does not live on its context's device");
    if (count == 0) return LR_OK;
    const int dd = dst->device, sd = src->device;
    hipStream_t copy = nullptr;
    LR_TRY(peer_streams().get(dd, sd, &copy));
    // the copy starts behind whatever has been enqueued through the producer's context so far: an event on its stream, no host wait
    LR_HIP(hipSetDevice(sd));
    hipEvent_t produced = nullptr;
    LR_HIP(hipEventCreateWithFlags(&produced, hipEventDisableTiming));
    hipError_t e = hipEventRecord(produced, src_ctx->stream);
    if (e == hipSuccess) {
        (void)hipSetDevice(dd);
        e = hipStreamWaitEvent(copy, produced, 0);
    }
    (void)hipEventDestroy(produced);        // (released by the runtime once the wait has been satisfied)
    if (e != hipSuccess) return fail(LR_ERR_HIP, std::string("peer copy: ordering behind the producer: ") + hipGetErrorString(e));
    LR_HIP(hipSetDevice(dd));
    const size_t poly_bytes = (size_t)dst->limbs * (size_t)dst->N * sizeof(u64);
    const bool dense = dst->stride() == (long long)dst->limbs * (long long)dst->N && src->stride() == dst->stride();
    const int pieces = dense ? 1 : count;
    const size_t bytes = dense ? poly_bytes * (size_t)count : poly_bytes;
    for (int k = 0; k < pieces; ++k) {
        u64 *d = dst->d + (long long)(dst_index + k) * dst->stride();
        const u64 *s = src->d + (long long)(src_index + k) * src->stride();
        if (dd == sd) LR_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, copy));
        else LR_HIP(hipMemcpyPeerAsync(d, dd, s, sd, bytes, copy));
    }
    return LR_OK;
    });
}

extern "C" int lr_context_wait_peer_copies(lr_context *ctx) {
    return guarded([&]() -> int {
    if (!ctx) return fail(LR_ERR_ARG, "null context");
    LR_HIP(hipSetDevice(ctx->device));
    for (hipStream_t s : peer_streams().into(ctx->device)) {
        hipEvent_t done = nullptr;
        LR_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        hipError_t e = hipEventRecord(done, s);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, done, 0);
        (void)hipEventDestroy(done);
        if (e != hipSuccess) return fail(LR_ERR_HIP, std::string("wait for peer copies: ") + hipGetErrorString(e));
    }
    return LR_OK;
    });
}

extern "C" int lr_gather_blocks(lr_context *dst_ctx, lr_poly *dst, lr_context *const *src_ctxs, const lr_poly *const *srcs, const int *counts, int n_blocks) {
    return guarded([&]() -> int {
    if (!dst_ctx || !dst || !src_ctxs || !srcs || !counts || n_blocks < 0) return fail(LR_ERR_ARG, "null argument");
    long long total = 0;
    for (int r = 0; r < n_blocks; ++r) {
        if (!src_ctxs[r] || !srcs[r] || counts[r] < 0 || counts[r] > srcs[r]->batch) return fail(LR_ERR_SHAPE, "gather: block count outside its source");
        total += counts[r];
    }
    if (total > dst->batch) return fail(LR_ERR_SHAPE, "gather: the blocks do not fit the destination batch");
    int slot = 0;
    for (int r = 0; r < n_blocks; ++r) {       // block r lands behind the blocks before it, in global unit order (sharding by contiguous blocks)
        LR_TRY(lr_poly_copy_peer(dst_ctx, dst, slot, src_ctxs[r], srcs[r], 0, counts[r]));
        slot += counts[r];
    }
    return lr_context_wait_peer_copies(dst_ctx);
    });
}
