// Multi-GPU exchange of the path: one process per GPU, contiguous pair ranges per
// rank, and ONE collective -- the all-gather of the per-pair parameter rows (float64, 48-56 B per pair).
// It runs on RCCL (ncclAllGather over xGMI) on the context's own stream, straight from this library:
// librccl.so is opened on first use, so single-GPU users never load it.
//
// The reference has no distributed code at all (results.py:41-50 is a serial loop over independent
// pairs); these entry points are what a sharded driver of that loop needs and nothing more.
#include <dlfcn.h>
#include <string.h>

#include "gme_internal.h"

namespace {

// the few RCCL symbols used, with the ABI of /opt/rocm/include/rccl/rccl.h (ROCm 7.2)
struct UniqueId { char internal[128]; };
typedef int (*fn_get_id)(UniqueId*);
typedef int (*fn_init_rank)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*fn_destroy)(void* comm);
typedef int (*fn_allgather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream);
typedef int (*fn_allreduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_comm_int)(void* comm, int* out);
constexpr int kFloat64 = 8, kMax = 2;      // ncclFloat64, ncclMax

struct Rccl {
    void* so = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_errstr errstr = nullptr;
    fn_comm_int count = nullptr, user_rank = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.so) return GME_OK;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    GME_REQUIRE(so != nullptr, GME_ERR_HIP, "cannot open librccl.so: %s", dlerror());
    Rccl r;
    r.so = so;
    r.get_id = (fn_get_id)dlsym(so, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(so, "ncclCommInitRank");
    r.destroy = (fn_destroy)dlsym(so, "ncclCommDestroy");
    r.allgather = (fn_allgather)dlsym(so, "ncclAllGather");
    r.allreduce = (fn_allreduce)dlsym(so, "ncclAllReduce");
    r.errstr = (fn_errstr)dlsym(so, "ncclGetErrorString");
    r.count = (fn_comm_int)dlsym(so, "ncclCommCount");
    r.user_rank = (fn_comm_int)dlsym(so, "ncclCommUserRank");
    GME_REQUIRE(r.get_id && r.init_rank && r.destroy && r.allgather && r.allreduce && r.count && r.user_rank, GME_ERR_HIP,
                "librccl.so lacks a symbol");
    g_rccl = r;
    return GME_OK;
}

#define GME_RCCL_TRY(expr)                                                                       \
    do {                                                                                         \
        const int r__ = (expr);                                                                  \
        if (r__ != 0) {                                                                          \
            gme_set_error("%s failed: %s", #expr, g_rccl.errstr ? g_rccl.errstr(r__) : "rccl");  \
            return GME_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

}  // namespace

extern "C" int gme_comm_probe(void) { return rccl_load(); }

extern "C" int gme_comm_unique_id(char id_out[128])
{
    GME_REQUIRE(id_out != nullptr, GME_ERR_ARG, "gme_comm_unique_id: null pointer");
    int rc = rccl_load();
    if (rc) return rc;
    UniqueId id;
    GME_RCCL_TRY(g_rccl.get_id(&id));
    memcpy(id_out, id.internal, 128);
    return GME_OK;
}

extern "C" int gme_comm_init(gme_ctx* ctx, const char id[128], int rank, int world)
{
    GME_REQUIRE(ctx != nullptr && id != nullptr, GME_ERR_ARG, "gme_comm_init: null pointer");
    GME_REQUIRE(world >= 1 && rank >= 0 && rank < world, GME_ERR_ARG, "gme_comm_init: rank %d of %d", rank, world);
    std::lock_guard<std::mutex> lock(ctx->mu);
    GME_HIP_TRY(hipSetDevice(ctx->device));
    GME_REQUIRE(ctx->comm == nullptr, GME_ERR_STATE, "gme_comm_init: the context already has a communicator");
    int rc = rccl_load();
    if (rc) return rc;
    UniqueId uid;
    memcpy(uid.internal, id, 128);
    void* comm = nullptr;
    GME_RCCL_TRY(g_rccl.init_rank(&comm, world, uid, rank));
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return GME_OK;
}

extern "C" int gme_comm_destroy(gme_ctx* ctx)
{
    GME_REQUIRE(ctx != nullptr, GME_ERR_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->comm) return GME_OK;
    GME_HIP_TRY(hipSetDevice(ctx->device));
    GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
    void* comm = ctx->comm;
    ctx->comm = nullptr; ctx->comm_world = 0;
    GME_RCCL_TRY(g_rccl.destroy(comm));
    return GME_OK;
}

// rows[n_local][k] of every rank -> out[world][n_max][k] (each rank's block padded with zeros to n_max rows)
extern "C" int gme_shard_gather(gme_ctx* ctx, const double* rows, int n_local, int k, int n_max, double* out)
{
    GME_REQUIRE(ctx != nullptr && out != nullptr && (rows != nullptr || n_local == 0), GME_ERR_ARG, "gme_shard_gather: null pointer");
    GME_REQUIRE(n_local >= 0 && k >= 1 && n_max >= n_local && n_max >= 1, GME_ERR_ARG, "gme_shard_gather: %d rows of %d, padded to %d", n_local, k, n_max);
    std::lock_guard<std::mutex> lock(ctx->mu);
    GME_HIP_TRY(hipSetDevice(ctx->device));
    GME_REQUIRE(ctx->comm != nullptr, GME_ERR_STATE, "gme_shard_gather before gme_comm_init");
    const size_t block = (size_t)n_max * k, send_bytes = block * sizeof(double);
    void* base = nullptr;
    int rc = ctx_scratch(ctx, send_bytes * ((size_t)ctx->comm_world + 1) + 512, &base);
    if (rc) return rc;
    double* d_send = (double*)base;
    double* d_recv = (double*)((uint8_t*)base + ((send_bytes + 255) & ~(size_t)255));
    GME_HIP_TRY(hipMemsetAsync(d_send, 0, send_bytes, ctx->stream));
    if (n_local) GME_HIP_TRY(hipMemcpyAsync(d_send, rows, (size_t)n_local * k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    GME_RCCL_TRY(g_rccl.allgather(d_send, d_recv, block, kFloat64, ctx->comm, ctx->stream));
    GME_HIP_TRY(hipMemcpyAsync(out, d_recv, send_bytes * ctx->comm_world, hipMemcpyDeviceToHost, ctx->stream));
    GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GME_OK;
}

// element-wise maximum of v[n] over the ranks, in place; with n = 1 and a dummy value it is the barrier
extern "C" int gme_comm_allreduce_max(gme_ctx* ctx, double* v, int n)
{
    GME_REQUIRE(ctx != nullptr && v != nullptr && n >= 1 && n <= 4096, GME_ERR_ARG, "gme_comm_allreduce_max: bad arguments");
    std::lock_guard<std::mutex> lock(ctx->mu);
    GME_HIP_TRY(hipSetDevice(ctx->device));
    GME_REQUIRE(ctx->comm != nullptr, GME_ERR_STATE, "gme_comm_allreduce_max before gme_comm_init");
    void* base = nullptr;
    int rc = ctx_scratch(ctx, (size_t)n * sizeof(double), &base);
    if (rc) return rc;
    GME_HIP_TRY(hipMemcpyAsync(base, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    GME_RCCL_TRY(g_rccl.allreduce(base, base, (size_t)n, kFloat64, kMax, ctx->comm, ctx->stream));
    GME_HIP_TRY(hipMemcpyAsync(v, base, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GME_OK;
}

// what RCCL itself says about the communicator (ncclCommUserRank / ncclCommCount), not what the caller passed in
extern "C" int gme_comm_info(gme_ctx* ctx, int* rank_out, int* world_out)
{
    GME_REQUIRE(ctx != nullptr, GME_ERR_ARG, "null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    GME_REQUIRE(ctx->comm != nullptr, GME_ERR_STATE, "gme_comm_info before gme_comm_init");
    int r = -1, n = -1;
    GME_RCCL_TRY(g_rccl.user_rank(ctx->comm, &r));
    GME_RCCL_TRY(g_rccl.count(ctx->comm, &n));
    if (rank_out) *rank_out = r;
    if (world_out) *world_out = n;
    return GME_OK;
}

// ---- per-pair summary rows of the last block-matching field, and their exchange ---------------------------------
static int grow(double** ptr, size_t* have, size_t want)
{
    if (*ptr && *have >= want) return GME_OK;
    if (*ptr) hipFree(*ptr);
    *ptr = nullptr; *have = 0;
    if (hipMalloc((void**)ptr, want) != hipSuccess) { gme_set_error("out of device memory (%zu bytes of summary rows)", want); return GME_ERR_NOMEM; }
    *have = want;
    return GME_OK;
}

static int summary_rows(gme_seq* s, int n_max)
{
    gme_ctx* ctx = s->ctx;
    GME_REQUIRE(s->mv != nullptr && s->mv_pairs > 0, GME_ERR_STATE, "no motion field yet: call gme_seq_bbme first");
    GME_REQUIRE(n_max >= s->mv_pairs, GME_ERR_ARG, "%d rows do not hold the %d pairs of this sequence", n_max, s->mv_pairs);
    const size_t bytes = (size_t)n_max * 6 * sizeof(double);
    if (!s->summary || s->summary_bytes < bytes) {
        GME_HIP_TRY(hipStreamSynchronize(ctx->stream));        // an earlier gather may still read the old rows
        int rc = grow(&s->summary, &s->summary_bytes, bytes);
        if (rc) return rc;
    }
    if (n_max > s->mv_pairs)                                   // the padding other ranks receive
        GME_HIP_TRY(hipMemsetAsync(s->summary + (size_t)s->mv_pairs * 6, 0, (size_t)(n_max - s->mv_pairs) * 6 * sizeof(double), ctx->stream));
    return launch_mv_summary(ctx, s->mv, s->mv_pairs, s->mv_h * s->mv_w, s->summary);
}

extern "C" int gme_seq_mv_summary(gme_seq* s, double* rows_out)
{
    GME_REQUIRE(s != nullptr && rows_out != nullptr, GME_ERR_ARG, "gme_seq_mv_summary: null pointer");
    gme_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    GME_HIP_TRY(hipSetDevice(ctx->device));
    int rc = summary_rows(s, s->mv_pairs);
    if (rc) return rc;
    GME_HIP_TRY(hipMemcpyAsync(rows_out, s->summary, (size_t)s->mv_pairs * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GME_OK;
}

// The rows never visit the host on the way: summary kernel -> ncclAllGather of the device rows (xGMI) -> one copy of
// the gathered block into `out`.  In split-phase mode (gme_seq_set_split_phase) the call returns once that copy is
// queued and gme_seq_wait delivers it, so the next step's search runs behind the collective without a gap.
extern "C" int gme_seq_mv_summary_gather(gme_seq* s, int n_max, double* out)
{
    GME_REQUIRE(s != nullptr && out != nullptr, GME_ERR_ARG, "gme_seq_mv_summary_gather: null pointer");
    gme_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    GME_HIP_TRY(hipSetDevice(ctx->device));
    GME_REQUIRE(ctx->comm != nullptr, GME_ERR_STATE, "gme_seq_mv_summary_gather before gme_comm_init");
    int rc = summary_rows(s, n_max);
    if (rc) return rc;
    const size_t block = (size_t)n_max * 6, bytes = block * sizeof(double) * (size_t)ctx->comm_world;
    if (!s->gathered || s->gathered_bytes < bytes) {
        GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
        rc = grow(&s->gathered, &s->gathered_bytes, bytes);
        if (rc) return rc;
    }
    GME_RCCL_TRY(g_rccl.allgather(s->summary, s->gathered, block, kFloat64, ctx->comm, ctx->stream));
    GME_HIP_TRY(hipMemcpyAsync(out, s->gathered, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (s->split_phase) { GME_HIP_TRY(hipEventRecord(s->ready, ctx->stream)); return GME_OK; }
    GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return GME_OK;
}
