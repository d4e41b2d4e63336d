// comm.hip -- tc_comm: the collectives of the multi-GPU entry points.
//
// One rank per GPU.  The data path has exactly two exchange steps: the all-reduce of the packed normal equations of an
// ICP iteration (TC_ICP_SUMS_STRIDE doubles = 256 bytes: latency bound, the per-link xGMI bandwidth is irrelevant) and
// the all-gather of the normals of a replicated cloud (n x 24 bytes, once per cloud).  Both are RCCL calls enqueued on
// the context's stream, so an iteration is kernels -> ncclAllReduce -> kernels with no host wait in between.
//
// librccl is bound at run time: a process that already carries RCCL (PyTorch-ROCm loads its own copy with
// torch.distributed) must not get a second instance, and a machine without RCCL must still be able to load the library
// for single-GPU work.  Resolution order: symbols visible in the process, librccl.so.1, librccl.so.
#include "tc_internal.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <vector>

namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL 2 enums)
typedef struct { char internal[TC_COMM_ID_BYTES]; } NcclUniqueId;
enum { kNcclSuccess = 0 };
enum { kNcclUint8 = 1, kNcclUint32 = 3, kNcclFloat64 = 8 };
enum { kNcclSum = 0 };

struct Rccl {
    bool ok = false;
    std::string why;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = RTLD_DEFAULT;
        if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) { r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
        r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather;
        if (!r.ok) r.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclAllGather";
    });
    return r;
}

std::string nccl_err(int rc) {
    Rccl &r = rccl();
    return std::string("RCCL: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error") + " (" + std::to_string(rc) + ")";
}

// host path: device -> host, callback, host -> device; blocking
tc_status host_collective(tc_comm *c, int op, void *d_buf, size_t count, size_t bytes) {
    tc_context *ctx = c->ctx;
    std::vector<char> h(bytes);
    TC_HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int rc = c->host_fn(c->host_user, op, h.data(), count);
    if (rc != 0) return tc::fail(ctx, TC_GPU, "host collective callback failed (" + std::to_string(rc) + ")");
    TC_HIP_TRY(ctx, hipMemcpyAsync(d_buf, h.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // `h` dies with this frame
    return TC_OK;
}

}  // namespace

namespace tc {

tc_status comm_allreduce_f64(tc_comm *c, double *d_buf, size_t count) {
    if (!c) return TC_OK;
    if (c->nccl) {        // (also with one rank: the call then is RCCL's in-place no-op, and a 1-GPU box exercises the real path)
        ProfScope ps(c->ctx, "comm_allreduce_f64");      // tc_profile_enable(ctx, 1): what the exchange step of an iteration costs
        const int rc = rccl().AllReduce(d_buf, d_buf, count, kNcclFloat64, kNcclSum, c->nccl, c->ctx->stream);
        return rc == kNcclSuccess ? TC_OK : fail(c->ctx, TC_GPU, nccl_err(rc));
    }
    if (c->nranks <= 1) return TC_OK;
    return host_collective(c, TC_COLL_SUM_F64, d_buf, count, count * sizeof(double));
}

tc_status comm_allreduce_u32(tc_comm *c, uint32_t *d_buf, size_t count) {
    if (!c) return TC_OK;
    if (c->nccl) {
        const int rc = rccl().AllReduce(d_buf, d_buf, count, kNcclUint32, kNcclSum, c->nccl, c->ctx->stream);
        return rc == kNcclSuccess ? TC_OK : fail(c->ctx, TC_GPU, nccl_err(rc));
    }
    if (c->nranks <= 1) return TC_OK;
    return host_collective(c, TC_COLL_SUM_U32, d_buf, count, count * sizeof(uint32_t));
}

tc_status comm_allgather(tc_comm *c, void *d_buf, size_t bytes_per_rank) {
    if (!c) return TC_OK;
    if (c->nccl) {
        // in place: the send buffer is this rank's slot of the receive buffer
        const int rc = rccl().AllGather((const char *)d_buf + (size_t)c->rank * bytes_per_rank, d_buf, bytes_per_rank, kNcclUint8, c->nccl,
                                        c->ctx->stream);
        return rc == kNcclSuccess ? TC_OK : fail(c->ctx, TC_GPU, nccl_err(rc));
    }
    if (c->nranks <= 1) return TC_OK;
    return host_collective(c, TC_COLL_ALLGATHER_U8, d_buf, bytes_per_rank, bytes_per_rank * (size_t)c->nranks);
}

tc_status comm_agree(tc_comm *c, tc_status local) {
    if (!c || c->nranks <= 1) return local;
    tc_context *ctx = c->ctx;
    // The agreement word is allocated when the communicator is created (alloc_agree_word), so that nothing fallible sits in
    // front of the all-reduce here: whatever happens to this rank on the way, it JOINS the collective -- a rank that returned
    // early would leave its peers waiting in it for ever -- and reports its own failure afterwards.
    if (!c->agree_word) return fail(ctx, TC_GPU, "communicator: no agreement word (created for one rank?)");
    uint32_t *h = (uint32_t *)((char *)ctx->pinned + 2048 + 8192 + 128);
    *h = local == TC_OK ? 0u : 1u;
    tc_status mine = local;
    if (hipMemcpyAsync(c->agree_word, h, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess && mine == TC_OK)
        mine = fail(ctx, TC_GPU, "communicator: writing the agreement word failed");
    const tc_status ar = comm_allreduce_u32(c, (uint32_t *)c->agree_word, 1);
    if (mine == TC_OK && ar != TC_OK) mine = ar;
    if (hipMemcpyAsync(h, c->agree_word, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess && mine == TC_OK)
        mine = fail(ctx, TC_GPU, "communicator: reading the agreement word failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && mine == TC_OK) mine = fail(ctx, TC_GPU, "communicator: stream synchronisation failed");
    if (mine != TC_OK) return mine;
    if (*h != 0u) return fail(ctx, TC_GPU, "sharded call: the set-up failed on " + std::to_string(*h) + " peer rank(s); no rank entered the loop");
    return TC_OK;
}

// every creator calls it: a failure surfaces at creation, on every rank, not inside the first sharded call on one of them
tc_status alloc_agree_word(tc_comm *c) {
    if (c->nranks <= 1) return TC_OK;
    if (hipSetDevice(c->ctx->device) != hipSuccess || hipMalloc(&c->agree_word, 64) != hipSuccess) {
        c->agree_word = nullptr;
        return fail(c->ctx, TC_GPU, "communicator: cannot allocate the agreement word");
    }
    return TC_OK;
}

}  // namespace tc

extern "C" {

tc_status tc_comm_unique_id(uint8_t id[TC_COMM_ID_BYTES]) try {
    if (!id) return TC_INVALID_DATA;
    Rccl &r = rccl();
    if (!r.ok) return TC_UNSUPPORTED;
    NcclUniqueId u;
    if (r.GetUniqueId(&u) != kNcclSuccess) return TC_GPU;
    std::memcpy(id, u.internal, TC_COMM_ID_BYTES);
    return TC_OK;
} TC_CATCH_STATUS(nullptr)

static tc_status comm_args(tc_context *ctx, int nranks, int rank, tc_comm **out) {
    if (!ctx || !out) return TC_INVALID_DATA;
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return tc::fail(ctx, TC_INVALID_DATA, "communicator: need 0 <= rank < nranks");
    return TC_OK;
}

tc_status tc_comm_create(tc_context *ctx, int nranks, int rank, const uint8_t id[TC_COMM_ID_BYTES], tc_comm **out) try {
    if (tc_status s = comm_args(ctx, nranks, rank, out)) return s;
    if (!id) return TC_INVALID_DATA;
    Rccl &r = rccl();
    if (!r.ok) return tc::fail(ctx, TC_UNSUPPORTED, r.why);
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    NcclUniqueId u;
    std::memcpy(u.internal, id, TC_COMM_ID_BYTES);
    void *nc = nullptr;
    const int rc = r.CommInitRank(&nc, nranks, u, rank);
    if (rc != kNcclSuccess) return tc::fail(ctx, TC_GPU, nccl_err(rc));
    tc_comm *c = new tc_comm();
    c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->nccl = nc; c->own_nccl = true;
    if (tc_status s = tc::alloc_agree_word(c)) { tc_comm_destroy(c); return s; }
    *out = c;
    return TC_OK;
} TC_CATCH_STATUS(ctx)

tc_status tc_comm_adopt(tc_context *ctx, void *nccl_comm, int nranks, int rank, tc_comm **out) try {
    if (tc_status s = comm_args(ctx, nranks, rank, out)) return s;
    if (!nccl_comm) return tc::fail(ctx, TC_INVALID_DATA, "communicator: null ncclComm_t");
    Rccl &r = rccl();
    if (!r.ok) return tc::fail(ctx, TC_UNSUPPORTED, r.why);
    tc_comm *c = new tc_comm();
    c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->nccl = nccl_comm; c->own_nccl = false;
    if (tc_status s = tc::alloc_agree_word(c)) { tc_comm_destroy(c); return s; }
    *out = c;
    return TC_OK;
} TC_CATCH_STATUS(ctx)

tc_status tc_comm_create_host(tc_context *ctx, int nranks, int rank, tc_host_collective_fn fn, void *user, tc_comm **out) try {
    if (tc_status s = comm_args(ctx, nranks, rank, out)) return s;
    if (!fn && nranks > 1) return tc::fail(ctx, TC_INVALID_DATA, "communicator: a host collective callback is required for nranks > 1");
    tc_comm *c = new tc_comm();
    c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->host_fn = fn; c->host_user = user;
    if (tc_status s = tc::alloc_agree_word(c)) { tc_comm_destroy(c); return s; }
    *out = c;
    return TC_OK;
} TC_CATCH_STATUS(ctx)

tc_status tc_comm_create_local(tc_context *ctx, tc_comm **out) { return tc_comm_create_host(ctx, 1, 0, nullptr, nullptr, out); }

int tc_comm_rank(const tc_comm *c) { return c ? c->rank : 0; }
int tc_comm_size(const tc_comm *c) { return c ? c->nranks : 1; }

void tc_comm_destroy(tc_comm *c) try {
    if (!c) return;
    if (c->nccl && c->own_nccl) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        (void)rccl().CommDestroy(c->nccl);
    }
    if (c->agree_word) (void)hipFree(c->agree_word);
    delete c;
} TC_CATCH_VOID

}  // extern "C"
