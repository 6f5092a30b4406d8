// Multi-GPU behind the C ABI: assembling the full frame from tile-sharded contexts.
//
// The reference is single-device (one wgpu adapter, raytracer/src/lib.rs:329-335); a caller that owns the
// frame loop -- State::new / State::redraw, lib.rs:217-234 and :241-307 -- drives N GPUs by holding one
// mrt_ctx per GPU, each with mrt_set_shard(i, N).  Pixels are independent (own RNG stream keyed by the GLOBAL
// pixel index, own texel, read-only scene), so the only exchange is one gather of finished RGBA32F bands
// per frame.  Two forms:
//   * mrt_gather            one process, N contexts: every shard's bands are copied straight into their place
//                           in the root's full frame by peer-to-peer copies over xGMI (band b of shard i lands at
//                           band b*N + i; one strided copy when shard and root share a device), each shard on its
//                           own source stream, i.e. all peer links at once; nothing to reduce, so no ring;
//   * mrt_gather_rccl       one process per GPU: grouped ncclSend / ncclRecv on a caller-supplied RCCL
//                           communicator (one message per peer, straight to the root), then the same strided
//                           un-permute on the root.  RCCL is resolved at run time from the process (the
//                           caller's own librccl -- torch's, or /opt/rocm's), so this library does not link it.
// Band layout (mrt_shard_info): local row r of shard (i, N) is global row ((r/8)*N + i)*8 + r%8.

#include <dlfcn.h>

#include <cstring>

#include "mrt_ctx.h"

using mrt::fail;

namespace {

size_t band_bytes(const mrt_ctx* c) { return (size_t)mrt::kBandRows * c->args.width * 4 * sizeof(float); }

// the root's full-frame buffer: local_bands * world bands (the tail beyond `height` rows is shard padding)
int ensure_gather_buffer(mrt_ctx* R) {
    const size_t need = band_bytes(R) * R->local_bands * R->shard_world;
    if (R->d_gather && R->gather_bytes == need) return MRT_OK;
    HIP_TRY(R, hipSetDevice(R->device));
    if (R->d_gather) { MRT_TRY(mrt::wait_stream(R, R->stream, __func__)); (void)hipFree(R->d_gather); R->d_gather = nullptr; }
    HIP_TRY(R, hipMalloc((void**)&R->d_gather, need ? need : 16));
    R->gather_bytes = need;
    return MRT_OK;
}

int ensure_event(mrt_ctx* c) {
    if (c->ev_gather) return MRT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_gather, hipEventDisableTiming));
    return MRT_OK;
}

// bands of shard `rank` (packed, `src` on device src_dev) -> their interleaved places in the full frame `dst_full` (on
// device dst_dev), on `stream`: one strided copy within a device, one peer copy per band (8 rows x W, 0.5 MB at C4; a few
// dozen per frame) across devices -- the 1-D peer copy is the form every HIP runtime supports between any two GPUs
hipError_t scatter_bands(float* dst_full, int dst_dev, const float* src, int src_dev, uint32_t rank, uint32_t world,
                         uint32_t local_bands, size_t bb, hipStream_t stream, bool per_band = false) {
    if (local_bands == 0 || bb == 0) return hipSuccess;
    if (dst_dev == src_dev && !per_band)
        return hipMemcpy2DAsync((char*)dst_full + (size_t)rank * bb, (size_t)world * bb, src, bb, bb, local_bands,
                                hipMemcpyDeviceToDevice, stream);
    for (uint32_t b = 0; b < local_bands; b++) {
        hipError_t e = hipMemcpyPeerAsync((char*)dst_full + ((size_t)b * world + rank) * bb, dst_dev,
                                          (const char*)src + (size_t)b * bb, src_dev, bb, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- RCCL, resolved at run time ------------------------------------------------------------------
typedef int (*nccl_p2p_fn)(void*, size_t, int, int, void*, hipStream_t);     // ncclSend / ncclRecv (buff, count, dtype, peer, comm, stream)
typedef int (*nccl_void_fn)(void);
typedef int (*nccl_query_fn)(void*, int*);
typedef const char* (*nccl_err_fn)(int);
struct Rccl {
    nccl_p2p_fn send = nullptr, recv = nullptr;
    nccl_void_fn group_start = nullptr, group_end = nullptr;
    nccl_query_fn comm_rank = nullptr, comm_count = nullptr;
    nccl_err_fn error_string = nullptr;
    bool ok = false;
    std::string why;
};
constexpr int kNcclFloat = 7;        // ncclFloat32 in rccl.h's ncclDataType_t

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        // the library the caller created its communicator with is already in the process: prefer it
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) { x.why = std::string("librccl.so.1 not loadable: ") + (dlerror() ? dlerror() : "?"); return x; }
        x.send = (nccl_p2p_fn)dlsym(h, "ncclSend");
        x.recv = (nccl_p2p_fn)dlsym(h, "ncclRecv");
        x.group_start = (nccl_void_fn)dlsym(h, "ncclGroupStart");
        x.group_end = (nccl_void_fn)dlsym(h, "ncclGroupEnd");
        x.comm_rank = (nccl_query_fn)dlsym(h, "ncclCommUserRank");
        x.comm_count = (nccl_query_fn)dlsym(h, "ncclCommCount");
        x.error_string = (nccl_err_fn)dlsym(h, "ncclGetErrorString");
        x.ok = x.send && x.recv && x.group_start && x.group_end && x.comm_rank && x.comm_count;
        if (!x.ok) x.why = "librccl lacks ncclSend/ncclRecv/ncclGroup*/ncclComm*";
        return x;
    }();
    return r;
}

}  // namespace

extern "C" {

uint32_t mrt_shard_global_row(uint32_t local_row, uint32_t rank, uint32_t world) {
    return ((local_row / mrt::kBandRows) * world + rank) * mrt::kBandRows + local_row % mrt::kBandRows;
}

uint32_t mrt_shard_local_rows(uint32_t height, uint32_t world) {
    if (world == 0) return 0;
    return ((mrt::total_bands(height) + world - 1) / world) * mrt::kBandRows;
}

int mrt_unshard_rows(const float* gathered, uint32_t world, uint32_t width, uint32_t height, float* out) {
    if (!gathered || !out || world == 0 || width == 0) return MRT_ERR_INVALID_ARG;
    const uint32_t lrows = mrt_shard_local_rows(height, world);
    const size_t row_floats = (size_t)width * 4;
    for (uint32_t r = 0; r < world; r++)
        for (uint32_t lr = 0; lr < lrows; lr++) {
            const uint32_t g = mrt_shard_global_row(lr, r, world);
            if (g >= height) continue;                                 // shard padding
            std::memcpy(out + (size_t)g * row_floats, gathered + ((size_t)r * lrows + lr) * row_floats,
                        row_floats * sizeof(float));
        }
    return MRT_OK;
}

int mrt_gather(mrt_ctx* const* ctxs, uint32_t n, uint32_t root) {
    if (!ctxs || n == 0 || root >= n) return fail(nullptr, MRT_ERR_INVALID_ARG, "mrt_gather: bad arguments");
    for (uint32_t i = 0; i < n; i++) if (!ctxs[i]) return fail(nullptr, MRT_ERR_INVALID_ARG, "mrt_gather: ctxs[%u] is null", i);
    mrt_ctx* const R = ctxs[root];
    for (uint32_t i = 0; i < n; i++) {
        const mrt_ctx* c = ctxs[i];
        if (c->shard_world != n || c->shard_rank != i)
            return fail(R, MRT_ERR_STATE, "mrt_gather: ctxs[%u] is shard %u of %u, expected %u of %u (mrt_set_shard)", i,
                        c->shard_rank, c->shard_world, i, n);
        if (c->args.width != R->args.width || c->args.height != R->args.height)
            return fail(R, MRT_ERR_STATE, "mrt_gather: ctxs[%u] renders %ux%u, the root %ux%u", i, c->args.width, c->args.height,
                        R->args.width, R->args.height);
    }
    int st = ensure_gather_buffer(R);
    if (st != MRT_OK) return st;
    const size_t bb = band_bytes(R);
    // Write-after-read: the shards' copies below run on the shards' OWN streams and overwrite the root's full frame.  What the
    // caller has queued on the root's stream so far -- a consumer of the previous gather's mrt_gathered_device_ptr among it --
    // must have finished first: an event on the root's stream, which every other stream waits for before its copies.
    HIP_TRY(R, hipSetDevice(R->device));
    if (!R->ev_gather_root) HIP_TRY(R, hipEventCreateWithFlags(&R->ev_gather_root, hipEventDisableTiming));
    HIP_TRY(R, hipEventRecord(R->ev_gather_root, R->stream));
    for (uint32_t i = 0; i < n; i++) {
        mrt_ctx* c = ctxs[i];
        if ((st = ensure_event(c)) != MRT_OK) { if (c != R) R->err = c->err; return st; }
        HIP_TRY(R, hipSetDevice(c->device));
        if (c->stream != R->stream) HIP_TRY(R, hipStreamWaitEvent(c->stream, R->ev_gather_root, 0));
        if (c->device != R->device) {
            // direct peer writes over xGMI; "already enabled" is fine, and without peer access HIP stages the copy
            hipError_t pe = hipDeviceEnablePeerAccess(R->device, 0);
            if (pe != hipSuccess) (void)hipGetLastError();
        }
        // on the SOURCE's stream, i.e. after its finalize pass; the root's stream then waits for every shard
        const float* src = c->d_fb[c->target ^ 1];
        HIP_TRY(R, scatter_bands(R->d_gather, R->device, src, c->device, i, n, c->local_bands, bb, c->stream, R->gather_per_band));
        HIP_TRY(R, hipEventRecord(c->ev_gather, c->stream));
    }
    HIP_TRY(R, hipSetDevice(R->device));
    for (uint32_t i = 0; i < n; i++)
        if (ctxs[i] != R) HIP_TRY(R, hipStreamWaitEvent(R->stream, ctxs[i]->ev_gather, 0));
    return MRT_OK;
}

int mrt_gather_rccl(mrt_ctx* c, void* nccl_comm, uint32_t root) {
    if (!c || !nccl_comm) return MRT_ERR_INVALID_ARG;
    const Rccl& N = rccl();
    if (!N.ok) return fail(c, MRT_ERR_STATE, "mrt_gather_rccl: %s", N.why.c_str());
    const uint32_t world = c->shard_world, rank = c->shard_rank;
    if (root >= world) return fail(c, MRT_ERR_INVALID_ARG, "mrt_gather_rccl: root %u of %u", root, world);
    int crank = -1, ccount = -1;
    if (N.comm_rank(nccl_comm, &crank) != 0 || N.comm_count(nccl_comm, &ccount) != 0 || crank != (int)rank || ccount != (int)world)
        return fail(c, MRT_ERR_STATE, "mrt_gather_rccl: communicator is rank %d of %d, the context is shard %u of %u", crank, ccount,
                    rank, world);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t local_floats = mrt::local_texels(c) * 4;
    const float* src = c->d_fb[c->target ^ 1];
    auto nccl_try = [&](int r, const char* what) -> int {
        if (r == 0) return MRT_OK;
        return fail(c, MRT_ERR_HIP, "%s failed: %s", what, N.error_string ? N.error_string(r) : "rccl error");
    };
    int st;
    if (rank != root) {
        // one message straight to the root over this GPU's own xGMI link
        return nccl_try(N.send(const_cast<float*>(src), local_floats, kNcclFloat, (int)root, nccl_comm, c->stream), "ncclSend");
    }
    if ((st = ensure_gather_buffer(c)) != MRT_OK) return st;
    const size_t stage_need = local_floats * sizeof(float) * world;
    if (!c->d_gather_stage || c->gather_stage_bytes != stage_need) {
        if (c->d_gather_stage) { MRT_TRY(mrt::wait_stream(c, c->stream, __func__)); (void)hipFree(c->d_gather_stage); c->d_gather_stage = nullptr; }
        HIP_TRY(c, hipMalloc((void**)&c->d_gather_stage, stage_need ? stage_need : 16));
        c->gather_stage_bytes = stage_need;
    }
    // (no write-after-read hazard here: the receives into the staging buffer and the un-permute into d_gather are all on the
    // context's own stream, i.e. behind whatever the caller queued there to read the previous frame)
    if ((st = nccl_try(N.group_start(), "ncclGroupStart")) != MRT_OK) return st;
    for (uint32_t r = 0; r < world; r++) {
        if (r == root) continue;
        st = nccl_try(N.recv(c->d_gather_stage + (size_t)r * local_floats, local_floats, kNcclFloat, (int)r, nccl_comm, c->stream), "ncclRecv");
        if (st != MRT_OK) { (void)N.group_end(); return st; }
    }
    if ((st = nccl_try(N.group_end(), "ncclGroupEnd")) != MRT_OK) return st;
    const size_t bb = band_bytes(c);
    for (uint32_t r = 0; r < world; r++) {
        const float* from = r == root ? src : c->d_gather_stage + (size_t)r * local_floats;
        HIP_TRY(c, scatter_bands(c->d_gather, c->device, from, c->device, r, world, c->local_bands, bb, c->stream));
    }
    return MRT_OK;
}

int mrt_debug_set_gather_per_band(mrt_ctx* root, int enabled) {
    if (!root) return MRT_ERR_INVALID_ARG;
    root->gather_per_band = enabled != 0;
    return MRT_OK;
}

void* mrt_gathered_device_ptr(mrt_ctx* root) { return root ? (void*)root->d_gather : nullptr; }

int mrt_read_gathered(mrt_ctx* R, float* out, size_t cap) {
    if (!R || !out) return MRT_ERR_INVALID_ARG;
    if (!R->d_gather) return fail(R, MRT_ERR_STATE, "mrt_read_gathered: nothing gathered yet (mrt_gather / mrt_gather_rccl on the root)");
    const size_t n = (size_t)R->args.width * R->args.height * 4;
    if (cap < n) return fail(R, MRT_ERR_TOO_SMALL, "mrt_read_gathered: need %zu floats", n);
    HIP_TRY(R, hipSetDevice(R->device));
    HIP_TRY(R, hipMemcpyAsync(out, R->d_gather, n * sizeof(float), hipMemcpyDeviceToHost, R->stream));
    MRT_TRY(mrt::wait_stream(R, R->stream, __func__));
    return MRT_OK;
}

}  // extern "C"
