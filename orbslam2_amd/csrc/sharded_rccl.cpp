// Brute-force top-2 against a descriptor database sharded over GPUs, one process (or thread) per GPU:
// the per-shard kernel, one RCCL all-gather of 8 bytes per query and rank over xGMI, the merge kernel.
// Host code only.  RCCL is looked up at run time (dlopen) the first time a sharded entry point is called,
// so liborbx.so loads and every single-GPU entry point works on a machine without RCCL.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include <rccl/rccl.h>  // types and prototypes only: nothing here links against librccl

#include "orbx_internal.h"

namespace {

struct RcclApi {
    void *lib;
    decltype(&ncclGetUniqueId) get_unique_id;
    decltype(&ncclCommInitRank) comm_init_rank;
    decltype(&ncclCommDestroy) comm_destroy;
    decltype(&ncclAllGather) all_gather;
    decltype(&ncclGetErrorString) error_string;
};

std::mutex g_api_mutex;
RcclApi g_api;
bool g_api_tried = false;
thread_local char g_rccl_error[256] = "";

void set_error(const char *what, const char *detail)
{
    snprintf(g_rccl_error, sizeof(g_rccl_error), "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
}

// The library a process already holds wins (a Python process that imported torch has torch's copy mapped
// under the same soname; two RCCL copies in one process would each bring their own topology state).
const RcclApi *rccl()
{
    std::lock_guard<std::mutex> lock(g_api_mutex);
    if (g_api_tried)
        return g_api.lib ? &g_api : nullptr;
    g_api_tried = true;
    const char *override_path = getenv("ORBX_RCCL_LIB");
    void *h = nullptr;
    if (override_path && *override_path)
        h = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h)
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); i++)
        h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h)
        return nullptr;
    RcclApi a;
    a.lib = h;
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    a.all_gather = reinterpret_cast<decltype(a.all_gather)>(dlsym(h, "ncclAllGather"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
    if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather || !a.error_string)
        return nullptr;
    g_api = a;
    return &g_api;
}

}  // namespace

struct orbx_sharded_matcher {
    int device, rank, world, max_queries;
    orbx_matcher *matcher;   // per-shard kernels, merge kernel, stream
    hipStream_t own_stream;
    hipStream_t stream;      // the stream kernels and the all-gather are enqueued on
    ncclComm_t comm;
    uint8_t *d_block;        // this rank's result block: 8 bytes per query
    uint8_t *d_all;          // the blocks of all ranks side by side, in rank order
};

static_assert(ORBX_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" const char *orbx_sharded_last_error(void) { return g_rccl_error; }

extern "C" int orbx_sharded_unique_id(void *id_out)
{
    if (!id_out)
        return ORBX_E_ARG;
    const RcclApi *api = rccl();
    if (!api) {
        set_error("RCCL not found (librccl.so.1; ORBX_RCCL_LIB overrides the path)", dlerror());
        return ORBX_E_RCCL;
    }
    ncclUniqueId id;
    ncclResult_t r = api->get_unique_id(&id);
    if (r != ncclSuccess) {
        set_error("ncclGetUniqueId", api->error_string(r));
        return ORBX_E_RCCL;
    }
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ORBX_OK;
}

extern "C" void orbx_sharded_matcher_destroy(orbx_sharded_matcher *sm)
{
    if (!sm)
        return;
    (void)hipSetDevice(sm->device);
    if (sm->stream)
        (void)hipStreamSynchronize(sm->stream);
    if (sm->comm) {
        const RcclApi *api = rccl();
        if (api)
            (void)api->comm_destroy(sm->comm);
    }
    if (sm->d_block)
        (void)hipFree(sm->d_block);
    if (sm->d_all)
        (void)hipFree(sm->d_all);
    orbx_matcher_destroy(sm->matcher);
    if (sm->own_stream)
        (void)hipStreamDestroy(sm->own_stream);
    delete sm;
}

extern "C" int orbx_sharded_matcher_create(int device, int rank, int world, const void *unique_id, int max_queries,
                                           orbx_sharded_matcher **out)
{
    if (!out || !unique_id || world < 1 || rank < 0 || rank >= world || max_queries < 1)
        return ORBX_E_ARG;
    *out = nullptr;
    const RcclApi *api = rccl();
    if (!api) {
        set_error("RCCL not found (librccl.so.1; ORBX_RCCL_LIB overrides the path)", dlerror());
        return ORBX_E_RCCL;
    }
    orbx_sharded_matcher *sm = new (std::nothrow) orbx_sharded_matcher();
    if (!sm)
        return ORBX_E_CAPACITY;
    memset(sm, 0, sizeof(*sm));
    sm->device = device;
    sm->rank = rank;
    sm->world = world;
    sm->max_queries = max_queries;
    int rc = orbx_matcher_create(device, max_queries, &sm->matcher);
    if (rc != ORBX_OK) {
        delete sm;
        return rc;
    }
    hipError_t e = hipSetDevice(device);
    // RCCL work and the kernels around it share one stream, so the three steps need no events between them
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&sm->own_stream, hipStreamNonBlocking);
    sm->stream = sm->own_stream;
    if (e == hipSuccess) rc = orbx_matcher_set_stream(sm->matcher, sm->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&sm->d_block, (size_t)8 * max_queries);
    if (e == hipSuccess) e = hipMalloc((void **)&sm->d_all, (size_t)8 * max_queries * world);
    if (e != hipSuccess || rc != ORBX_OK) {
        if (e != hipSuccess)
            orbx_note_hip_error(e);
        orbx_sharded_matcher_destroy(sm);
        return e != hipSuccess ? ORBX_E_HIP : rc;
    }
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = api->comm_init_rank(&sm->comm, world, id, rank);  // collective: every rank calls it
    if (r != ncclSuccess) {
        sm->comm = nullptr;
        set_error("ncclCommInitRank", api->error_string(r));
        orbx_sharded_matcher_destroy(sm);
        return ORBX_E_RCCL;
    }
    *out = sm;
    return ORBX_OK;
}

extern "C" int orbx_sharded_matcher_set_stream(orbx_sharded_matcher *sm, void *hip_stream)
{
    if (!sm)
        return ORBX_E_ARG;
    sm->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : sm->own_stream;
    return orbx_matcher_set_stream(sm->matcher, sm->stream);
}

extern "C" int orbx_sharded_matcher_synchronize(orbx_sharded_matcher *sm)
{
    if (!sm)
        return ORBX_E_ARG;
    hipError_t e = hipSetDevice(sm->device);
    if (e == hipSuccess) e = hipStreamSynchronize(sm->stream);
    if (e != hipSuccess) {
        orbx_note_hip_error(e);
        return ORBX_E_HIP;
    }
    return ORBX_OK;
}

extern "C" int orbx_match_top2_sharded_device(orbx_sharded_matcher *sm, const uint8_t *d_queries, int nq,
                                              const uint8_t *d_shard, int64_t shard_rows, int64_t idx_base,
                                              int32_t *d_best_idx, uint16_t *d_d1, uint16_t *d_d2)
{
    if (!sm || !d_queries || nq < 1 || nq > sm->max_queries || shard_rows < 0 || (shard_rows > 0 && !d_shard) ||
        !d_best_idx || !d_d1 || !d_d2)
        return ORBX_E_ARG;
    const RcclApi *api = rccl();
    if (!api)
        return ORBX_E_RCCL;
    uint8_t *blk = sm->d_block;
    int rc = orbx_match_top2_device(sm->matcher, d_queries, nq, d_shard, shard_rows, idx_base,
                                    reinterpret_cast<int32_t *>(blk), reinterpret_cast<uint16_t *>(blk + (size_t)4 * nq),
                                    reinterpret_cast<uint16_t *>(blk + (size_t)6 * nq));
    if (rc != ORBX_OK)
        return rc;
    // 8*nq bytes from every rank, laid down in rank order = ascending row order of the shards
    ncclResult_t r = api->all_gather(blk, sm->d_all, (size_t)8 * nq, ncclUint8, sm->comm, sm->stream);
    if (r != ncclSuccess) {
        set_error("ncclAllGather", api->error_string(r));
        return ORBX_E_RCCL;
    }
    return orbx_top2_merge_shards_device(sm->matcher, sm->world, nq, sm->d_all, (int64_t)8 * nq, d_best_idx, d_d1, d_d2);
}
