// C ABI of libgme_hip.so (include/gme_hip.h): contexts, sequences, host<->device
// plumbing around the kernels in bbme_*.hip, gme_kernels.hip and synth_kernels.hip.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "gme_internal.h"

static thread_local char g_err[512] = "";

void gme_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gme_last_error(void) { return g_err; }

// first chunk of a block-matching call names the plan, later chunks only add their patches
void plan_note(gme_ctx* ctx, long long patches, const char* fmt, ...)
{
    ctx->plan_patches += patches;
    if (ctx->plan[0]) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->plan, sizeof(ctx->plan), fmt, ap);
    va_end(ap);
}

extern "C" int gme_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { gme_set_error("hipGetDeviceCount failed (no HIP device?)"); return 0; }
    return n;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
extern "C" gme_ctx* gme_create(int device_id)
{
    int n = gme_device_count();
    if (device_id < 0 || device_id >= n) {
        gme_set_error("gme_create: device %d not available (%d HIP devices visible)", device_id, n);
        return nullptr;
    }
    gme_ctx* ctx = new (std::nothrow) gme_ctx();
    if (!ctx) return nullptr;
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipMalloc((void**)&ctx->status, GME_STATUS_WORDS * sizeof(int)) != hipSuccess ||
        hipMemsetAsync(ctx->status, 0, GME_STATUS_WORDS * sizeof(int), ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        gme_set_error("gme_create: HIP initialisation failed on device %d: %s", device_id,
                      hipGetErrorString(hipGetLastError()));
        if (ctx->status) hipFree(ctx->status);
        if (ctx->ev0) hipEventDestroy(ctx->ev0);
        if (ctx->ev1) hipEventDestroy(ctx->ev1);
        if (ctx->stream) hipStreamDestroy(ctx->stream);
        delete ctx;
        return nullptr;
    }
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        gme_set_error("gme_create: device %d is %s; this library carries gfx950 code only", device_id,
                      ctx->prop.gcnArchName);
        hipFree(ctx->status);
        hipEventDestroy(ctx->ev0);
        hipEventDestroy(ctx->ev1);
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

extern "C" void gme_destroy(gme_ctx* ctx)
{
    if (!ctx) return;
    { std::lock_guard<std::mutex> lock(ctx->mu); }        // let a call in flight on another thread finish
    gme_comm_destroy(ctx);
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->redo_list) hipFree(ctx->redo_list);
    if (ctx->stage) hipFree(ctx->stage);
    if (ctx->status) hipFree(ctx->status);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->copy_stream2) hipStreamDestroy(ctx->copy_stream2);
    if (ctx->back_stream) hipStreamDestroy(ctx->back_stream);
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Every entry point that touches a context holds its mutex for the whole call: the context owns ONE
// stream and ONE growable scratch buffer, and ctypes releases the GIL, so two Python threads inside
// the module-level bbme/motion/utils functions (one shared default context) would otherwise carve
// the same scratch offsets or free the buffer under each other's kernels.  Calls on one context
// serialise; use one context per thread (sequence.ShardedSequence does) for concurrency.
#define GME_ENTER(c)                                                  \
    GME_REQUIRE((c) != nullptr, GME_ERR_ARG, "null context");         \
    std::lock_guard<std::mutex> gme_lock__((c)->mu);                  \
    GME_HIP_TRY(hipSetDevice((c)->device))

// wait for the stream and report a tripped in-kernel guard
static int ctx_finish(gme_ctx* ctx)
{
    int st = 0;
    GME_HIP_TRY(hipMemcpyAsync(&st, ctx->status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (st != 0) {
        hipMemsetAsync(ctx->status, 0, sizeof(int), ctx->stream);
        gme_set_error("a search walk exceeded its iteration guard (internal error)");
        return GME_ERR_STATE;
    }
    return GME_OK;
}

// Small results and arguments of the split-phase calls (parameters, sums, squared errors: a few dozen KB) do not go
// through hipMemcpyAsync when the caller's buffer is page-locked: the copy engines serve copies in order, and a 60 KB result
// queued behind another context's 177 MB upload arrived 3.4 ms late -- every stage of a streamed estimate waited for
// whatever chunk was crossing the link (round 3 timeline).  Page-locked memory (gme_host_alloc /
// hipHostMalloc) is mapped into the device's address space, so a kernel on the context's own stream moves the words
// itself.  Ordinary memory falls back to hipMemcpyAsync.
__global__ void __launch_bounds__(256) k_copy_words(const uint32_t* src, uint32_t* dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

static bool mapped_host_view(const void* host, void** dev)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return false; }      // ordinary memory: not an error
    if (at.type != hipMemoryTypeHost || at.devicePointer == nullptr) return false;
    *dev = at.devicePointer;
    return true;
}

// dst <- src of `bytes` (a multiple of 4) on the context's stream; exactly one of the two is host memory
static int copy_small(gme_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, bool prefer_kernel)
{
    if (bytes == 0) return GME_OK;
    void* view = nullptr;
    if (prefer_kernel && bytes % 4 == 0 && bytes <= ((size_t)16 << 20) && mapped_host_view(kind == hipMemcpyDeviceToHost ? dst : src, &view)) {
        const uint32_t* s4 = (const uint32_t*)(kind == hipMemcpyDeviceToHost ? src : view);
        uint32_t* d4 = (uint32_t*)(kind == hipMemcpyDeviceToHost ? view : dst);
        const size_t n = bytes / 4;
        const unsigned grid = (unsigned)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
        hipLaunchKernelGGL(k_copy_words, dim3(grid), dim3(256), 0, ctx->stream, s4, d4, n);
        GME_HIP_TRY(hipGetLastError());
        return GME_OK;
    }
    GME_HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
    return GME_OK;
}

extern "C" int gme_sync(gme_ctx* ctx)
{
    GME_ENTER(ctx);
    return ctx_finish(ctx);
}

extern "C" int gme_last_bbme_info(gme_ctx* ctx, char* plan, int plan_len, int64_t* patches, int64_t* surviving,
                                  int64_t* redo_tiles)
{
    GME_ENTER(ctx);
    if (plan && plan_len > 0) snprintf(plan, plan_len, "%s", ctx->plan);
    if (patches) *patches = ctx->plan_patches;
    if (surviving || redo_tiles) {
        uint32_t st[8 * 16];
        GME_HIP_TRY(hipMemcpyAsync(st, ctx->status + GME_STATUS_STATS, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
        int64_t n = 0, r = 0;
        for (int x = 0; x < 8; ++x) { n += st[16 * x]; r += st[16 * x + 1]; }
        if (surviving) *surviving = n;
        if (redo_tiles) *redo_tiles = r;
    }
    return GME_OK;
}

extern "C" int gme_last_bbme_listed(gme_ctx* ctx, int64_t* listed)
{
    GME_ENTER(ctx);
    uint32_t st[8 * 16];
    GME_HIP_TRY(hipMemcpyAsync(st, ctx->status + GME_STATUS_STATS, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
    int64_t n = 0;
    for (int x = 0; x < 8; ++x) n += st[16 * x + 2];
    if (listed) *listed = n;
    return GME_OK;
}

extern "C" void* gme_stream(gme_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int gme_device_info(gme_ctx* ctx, char* name, int name_len, int* cu_count, int* clock_khz)
{
    GME_REQUIRE(ctx != nullptr, GME_ERR_ARG, "null context");
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = ctx->prop.clockRate;
    return GME_OK;
}

extern "C" int gme_device_bus_id(gme_ctx* ctx, char* out, int out_len)
{
    GME_ENTER(ctx);
    GME_REQUIRE(out != nullptr && out_len >= 16, GME_ERR_ARG, "gme_device_bus_id: buffer of at least 16 bytes");
    GME_HIP_TRY(hipDeviceGetPCIBusId(out, out_len, ctx->device));
    const hipError_t q = hipStreamQuery(ctx->stream);                  // the device still answers (work in flight is fine)
    GME_REQUIRE(q == hipSuccess || q == hipErrorNotReady, GME_ERR_HIP, "gme_device_bus_id: %s", hipGetErrorString(q));
    return GME_OK;
}

extern "C" int gme_timer_start(gme_ctx* ctx)
{
    GME_ENTER(ctx);
    GME_HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return GME_OK;
}

extern "C" int gme_timer_stop(gme_ctx* ctx, float* elapsed_ms)
{
    GME_ENTER(ctx);
    GME_HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    GME_HIP_TRY(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    GME_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return GME_OK;
}

int ctx_scratch(gme_ctx* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->scratch_bytes) {
        GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        const size_t want = (bytes + (1u << 20)) & ~((size_t)(1u << 20) - 1);
        if (hipMalloc(&ctx->scratch, want) != hipSuccess) {
            gme_set_error("out of device memory (%zu bytes of scratch)", want);
            return GME_ERR_NOMEM;
        }
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return GME_OK;
}

int ctx_redo_list(gme_ctx* ctx, size_t entries, uint32_t** out)
{
    if (entries > ctx->redo_cap) {
        GME_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->redo_list) hipFree(ctx->redo_list);
        ctx->redo_list = nullptr;
        ctx->redo_cap = 0;
        const size_t want = (entries + 4095) & ~(size_t)4095;
        if (hipMalloc((void**)&ctx->redo_list, want * sizeof(uint32_t)) != hipSuccess) {
            gme_set_error("out of device memory (%zu redo entries)", want);
            return GME_ERR_NOMEM;
        }
        ctx->redo_cap = want;
    }
    *out = ctx->redo_list;
    return GME_OK;
}

int plane_alloc(gme_ctx* ctx, Plane* p, int count, int H, int W)
{
    p->H = H; p->W = W; p->count = count;
    p->pitch = round_up(W, 64);
    p->stride = (int64_t)round_up(p->pitch * H, 256);
    p->ptr = nullptr;
    if (count == 0 || H == 0 || W == 0) return GME_OK;
    // one guard row behind the stack: dword loads near the last row never leave the allocation
    if (hipMalloc((void**)&p->ptr, p->bytes() + p->pitch) != hipSuccess) {
        gme_set_error("out of device memory (%zu bytes of frames)", p->bytes());
        return GME_ERR_NOMEM;
    }
    // on the context's stream: a null-stream memset would not be ordered against the
    // kernels this (non-blocking) stream runs next and could land on top of their output
    GME_HIP_TRY(hipMemsetAsync(p->ptr, 0, p->bytes() + p->pitch, ctx->stream));
    return GME_OK;
}

void plane_free(Plane* p)
{
    if (p->ptr) hipFree(p->ptr);
    p->ptr = nullptr;
}

// carve `n` sub-buffers out of the context scratch, 256-byte aligned
struct Carver {
    uint8_t* base = nullptr;
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; }
};

// ---------------------------------------------------------------------------
// single-pair calls on host buffers
// ---------------------------------------------------------------------------
extern "C" int gme_bbme_u8(gme_ctx* ctx, const uint8_t* prev, const uint8_t* cur, int H, int W, int stride,
                           int block_size, int search_window, int procedure, int pnorm, int32_t* mf_out)
{
    GME_ENTER(ctx);
    int rc = GME_OK;
    GME_REQUIRE(prev && cur && mf_out, GME_ERR_ARG, "gme_bbme_u8: null pointer");
    GME_REQUIRE(H > 0 && W > 0 && stride >= W, GME_ERR_ARG, "gme_bbme_u8: bad shape H=%d W=%d stride=%d", H, W, stride);
    GME_REQUIRE(block_size >= 1, GME_ERR_ARG, "gme_bbme_u8: block_size %d", block_size);
    rc = bbme_check_args(H, W, block_size, search_window, procedure, pnorm);
    if (rc) return rc;
    const int h = H / block_size, w = W / block_size;
    if (h == 0 || w == 0) return GME_OK;
    const int pitch = round_up(W, 64);
    const size_t plane = (size_t)round_up(pitch * H, 256);
    Carver c;
    const size_t o_prev = c.take(plane + pitch), o_cur = c.take(plane + pitch);
    const size_t o_mf = c.take((size_t)h * w * 2 * sizeof(int32_t));
    const int aux = bbme_aux_kind(block_size, search_window, procedure, pnorm);
    const bool want_sq = aux != 0;
    const size_t o_sq = c.take(want_sq ? plane * 4 : 0);
    void* base = nullptr;
    rc = ctx_scratch(ctx, c.off, &base);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)base;
    GME_HIP_TRY(hipMemsetAsync(b + o_prev, 0, o_mf - o_prev, ctx->stream));
    GME_HIP_TRY(hipMemcpy2DAsync(b + o_prev, pitch, prev, stride, W, H, hipMemcpyHostToDevice, ctx->stream));
    GME_HIP_TRY(hipMemcpy2DAsync(b + o_cur, pitch, cur, stride, W, H, hipMemcpyHostToDevice, ctx->stream));
    BbmeJob job;
    job.prev = b + o_prev; job.cur = b + o_cur; job.plane_stride = 0; job.pairs = 1;
    job.H = H; job.W = W; job.pitch = pitch;
    job.bs = block_size; job.sw = search_window; job.procedure = procedure; job.pnorm = pnorm;
    job.mf = (int32_t*)(b + o_mf); job.sqbox_cur = nullptr; job.sqbox_stride = 0;
    if (want_sq) {
        rc = launch_aux_table(ctx, aux, b + o_cur, 0, 1, H, W, pitch, (uint32_t*)(b + o_sq), 0);
        if (rc) return rc;
        job.sqbox_cur = (const uint32_t*)(b + o_sq);
    }
    rc = launch_bbme(ctx, job);
    if (rc) return rc;
    GME_HIP_TRY(hipMemcpyAsync(mf_out, b + o_mf, (size_t)h * w * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return ctx_finish(ctx);
}

extern "C" int gme_pyrdown_u8(gme_ctx* ctx, const uint8_t* src, int H, int W, int stride, uint8_t* dst)
{
    GME_ENTER(ctx);
    int rc = GME_OK;
    GME_REQUIRE(src && dst && H > 0 && W > 0 && stride >= W, GME_ERR_ARG, "gme_pyrdown_u8: bad arguments");
    Plane s, d;
    s.H = H; s.W = W; s.pitch = round_up(W, 64); s.stride = round_up(s.pitch * H, 256); s.count = 1;
    d.H = (H + 1) / 2; d.W = (W + 1) / 2; d.pitch = round_up(d.W, 64); d.stride = round_up(d.pitch * d.H, 256); d.count = 1;
    Carver c;
    const size_t o_s = c.take(s.stride), o_d = c.take(d.stride);
    void* base = nullptr;
    rc = ctx_scratch(ctx, c.off, &base);
    if (rc) return rc;
    s.ptr = (uint8_t*)base + o_s; d.ptr = (uint8_t*)base + o_d;
    GME_HIP_TRY(hipMemcpy2DAsync(s.ptr, s.pitch, src, stride, W, H, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_pyrdown(ctx, s, d);
    if (rc) return rc;
    GME_HIP_TRY(hipMemcpy2DAsync(dst, d.W, d.ptr, d.pitch, d.W, d.H, hipMemcpyDeviceToHost, ctx->stream));
    return ctx_finish(ctx);
}

extern "C" int gme_affine_field(gme_ctx* ctx, const double params[6], int h, int w, int16_t* mf_out)
{
    GME_ENTER(ctx);
    int rc = GME_OK;
    GME_REQUIRE(params && mf_out && h >= 0 && w >= 0, GME_ERR_ARG, "gme_affine_field: bad arguments");
    if (h == 0 || w == 0) return GME_OK;
    Carver c;
    const size_t o_p = c.take(6 * sizeof(double)), o_f = c.take((size_t)h * w * 2 * sizeof(int16_t));
    void* base = nullptr;
    rc = ctx_scratch(ctx, c.off, &base);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)base;
    GME_HIP_TRY(hipMemcpyAsync(b + o_p, params, 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = launch_affine_field(ctx, (const double*)(b + o_p), 1, h, w, (int16_t*)(b + o_f));
    if (rc) return rc;
    GME_HIP_TRY(hipMemcpyAsync(mf_out, b + o_f, (size_t)h * w * 2 * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->stream));
    return ctx_finish(ctx);
}

extern "C" int gme_compensate_u8(gme_ctx* ctx, const uint8_t* frame, int H, int W, int stride, const int32_t* mf,
                                 int h, int w, uint8_t* out)
{
    GME_ENTER(ctx);
    int rc = GME_OK;
    GME_REQUIRE(frame && mf && out && H > 0 && W > 0 && stride >= W, GME_ERR_ARG, "gme_compensate_u8: bad arguments");
    GME_REQUIRE(h > 0 && w > 0 && h <= H, GME_ERR_ARG,
                "gme_compensate_u8: field of %d x %d blocks on a %d-row frame (motion.py:303 divides H by it)", h, w, H);
    const int pitch = round_up(W, 64);
    const size_t plane = (size_t)round_up(pitch * H, 256);
    Carver c;
    const size_t o_f = c.take(plane), o_o = c.take(plane), o_m = c.take((size_t)h * w * 2 * sizeof(int32_t));
    void* base = nullptr;
    rc = ctx_scratch(ctx, c.off, &base);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)base;
    GME_HIP_TRY(hipMemcpy2DAsync(b + o_f, pitch, frame, stride, W, H, hipMemcpyHostToDevice, ctx->stream));
    GME_HIP_TRY(hipMemcpyAsync(b + o_m, mf, (size_t)h * w * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    rc = launch_compensate(ctx, b + o_f, 0, 1, H, W, pitch, (const int32_t*)(b + o_m), nullptr, h, w, b + o_o, 0, pitch,
                           nullptr, 0, nullptr);
    if (rc) return rc;
    GME_HIP_TRY(hipMemcpy2DAsync(out, W, b + o_o, pitch, W, H, hipMemcpyDeviceToHost, ctx->stream));
    return ctx_finish(ctx);
}

extern "C" int gme_sse_u8(gme_ctx* ctx, const uint8_t* a, const uint8_t* b_, int H, int W, int stride_a, int stride_b,
                          int64_t* sse_out)
{
    GME_ENTER(ctx);
    int rc = GME_OK;
    GME_REQUIRE(a && b_ && sse_out && H > 0 && W > 0 && stride_a >= W && stride_b >= W, GME_ERR_ARG,
                "gme_sse_u8: bad arguments");
    const int pitch = round_up(W, 64);
    const size_t plane = (size_t)round_up(pitch * H, 256);
    Carver c;
    const size_t o_a = c.take(plane), o_b = c.take(plane), o_s = c.take(sizeof(unsigned long long));
    void* base = nullptr;
    rc = ctx_scratch(ctx, c.off, &base);
    if (rc) return rc;
    uint8_t* b = (uint8_t*)base;
    GME_HIP_TRY(hipMemcpy2DAsync(b + o_a, pitch, a, stride_a, W, H, hipMemcpyHostToDevice, ctx->stream));
    GME_HIP_TRY(hipMemcpy2DAsync(b + o_b, pitch, b_, stride_b, W, H, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_sse(ctx, b + o_a, 0, pitch, b + o_b, 0, pitch, 1, H, W, (unsigned long long*)(b + o_s));
    if (rc) return rc;
    unsigned long long v = 0;
    GME_HIP_TRY(hipMemcpyAsync(&v, b + o_s, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    rc = ctx_finish(ctx);
    *sse_out = (int64_t)v;
    return rc;
}

// ---------------------------------------------------------------------------
// sequences
// ---------------------------------------------------------------------------
extern "C" gme_seq* gme_seq_create(gme_ctx* ctx, int n_frames, int H, int W)
{
    if (!ctx) { gme_set_error("null context"); return nullptr; }
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { gme_set_error("hipSetDevice(%d) failed", ctx->device); return nullptr; }
    if (n_frames < 1 || H < 1 || W < 1) { gme_set_error("gme_seq_create: bad shape"); return nullptr; }
    gme_seq* s = new (std::nothrow) gme_seq();
    if (!s) return nullptr;
    s->ctx = ctx; s->N = s->N_cap = n_frames; s->H = H; s->W = W;
    if (plane_alloc(ctx, &s->level[2], n_frames, H, W) != GME_OK) { delete s; return nullptr; }
    return s;
}

static void free_fit(FitLevelBuf& f)
{
    if (f.gt) hipFree(f.gt);
    if (f.model) hipFree(f.model);
    if (f.mask) hipFree(f.mask);
    if (f.diff) hipFree(f.diff);
    if (f.list) hipFree(f.list);
    if (f.thr) hipFree(f.thr);
    if (f.sums) hipFree(f.sums);
    f = FitLevelBuf();
}

extern "C" void gme_seq_destroy(gme_seq* s)
{
    if (!s) return;
    std::lock_guard<std::mutex> lock(s->ctx->mu);
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    for (int l = 0; l < 3; ++l) { plane_free(&s->level[l]); free_fit(s->fit[l]); }
    s->fit_mv.gt = nullptr;
    free_fit(s->fit_mv);
    if (s->mv_params) hipFree(s->mv_params);
    plane_free(&s->comp);
    if (s->mv) hipFree(s->mv);
    for (int l = 0; l < 3; ++l) if (s->sqbox[l]) hipFree(s->sqbox[l]);
    if (s->params0) hipFree(s->params0);
    if (s->params_in) hipFree(s->params_in);
    if (s->solve_flags) hipFree(s->solve_flags);
    if (s->sse) hipFree(s->sse);
    if (s->comp_params) hipFree(s->comp_params);
    if (s->synth_canvas) hipFree(s->synth_canvas);
    if (s->summary) hipFree(s->summary);
    if (s->gathered) hipFree(s->gathered);
    if (s->ready) hipEventDestroy(s->ready);
    if (s->upload_gate) hipEventDestroy(s->upload_gate);
    if (s->uploaded) hipEventDestroy(s->uploaded);
    delete s;
}

// Split-phase calls: with the switch on, gme_seq_gme_begin / gme_seq_gme_fit / gme_seq_compensate queue their work
// (including the copy of their result into the caller's buffer, which should be page-locked: gme_host_alloc) and
// return; gme_seq_wait blocks until the result of the LAST such call has arrived -- not until the stream is idle, so
// the searches queued behind the copy keep running, and one host thread can drive several sequences on several
// streams, solving one's 3x3 systems while the others' kernels run.
extern "C" int gme_seq_set_split_phase(gme_seq* s, int on)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    GME_ENTER(s->ctx);
    // release-to-system: the results behind this event are written by a kernel into host memory the CALLER chose; whatever
    // its coherence (hipHostMallocNonCoherent, hipHostRegister'ed, HIP_HOST_COHERENT=0), they are visible once it fires
    if (on && !s->ready) GME_HIP_TRY(hipEventCreateWithFlags(&s->ready, hipEventDisableTiming | hipEventReleaseToSystem));
    s->split_phase = on != 0;
    return GME_OK;
}

extern "C" int gme_seq_wait(gme_seq* s)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    GME_ENTER(s->ctx);
    GME_REQUIRE(s->ready != nullptr, GME_ERR_STATE, "gme_seq_wait without gme_seq_set_split_phase");
    GME_HIP_TRY(hipEventSynchronize(s->ready));
    return GME_OK;
}

// 1 when the result of the last split-phase call has arrived (gme_seq_wait would return at once), 0 when it is still
// on its way: lets one host thread serve whichever of several sequences is ready first instead of blocking on one.
extern "C" int gme_seq_poll(gme_seq* s)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    GME_ENTER(s->ctx);
    GME_REQUIRE(s->ready != nullptr, GME_ERR_STATE, "gme_seq_poll without gme_seq_set_split_phase");
    const hipError_t e = hipEventQuery(s->ready);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    gme_set_error("hipEventQuery failed: %s", hipGetErrorString(e));
    return GME_ERR_HIP;
}

// new frame data ends a staged GME run: searches gme_seq_gme_begin deferred must not see other frames
// than the stages already done, so the run is dropped (gme_seq_gme_fit then asks for a new begin)
static void gme_drop_run(gme_seq* s)
{
    s->bbme_pending[0] = s->bbme_pending[1] = s->bbme_pending[2] = false;
    s->gme_pairs = 0;
}

// The sequence acts as one of `n_frames` frames (1 <= n_frames <= the count it was created with) from now on: every
// stage call covers the pairs of frames [0, n_frames) only.  Buffers stay sized for the full count, so a caller that
// streams chunks of different lengths through one sequence (sequence.StreamEstimator) pays for the pairs it has, not for
// the capacity, and nothing is reallocated.  Ends a staged GME run like new frame data does.
extern "C" int gme_seq_set_frames(gme_seq* s, int n_frames)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    GME_ENTER(s->ctx);
    GME_REQUIRE(n_frames >= 1 && n_frames <= s->N_cap, GME_ERR_ARG, "gme_seq_set_frames: %d frames in a sequence created for %d", n_frames, s->N_cap);
    if (n_frames != s->N) {
        s->N = n_frames;
        s->pyramids_valid = false;
        s->sqbox_valid[0] = s->sqbox_valid[1] = s->sqbox_valid[2] = false;
        gme_drop_run(s);
    }
    return GME_OK;
}

// One upload lane per device for the split-phase uploads of ALL contexts: host-to-device copies issued from several
// streams at once share the link badly (two concurrent copy streams moved 22-25 GB/s where one moves 38-54, DESIGN.md
// section 5; three lanes uploading on their own streams reached 62 % of the copy-alone rate, one shared lane does the
// rest), so they queue on ONE stream in call order; events tie each copy to its sequence's own stream.
namespace {
struct Uploader {
    std::mutex mu;
    hipStream_t stream = nullptr;
    uint8_t* stage = nullptr;        // device staging of tight host frames, repacked into the pitched planes
    size_t stage_bytes = 0;
};
Uploader g_uploader[16];
}  // namespace

// the copies of one gme_seq_upload call on `stream`; `stage` / `stage_bytes` is the staging buffer that belongs to that stream
static int upload_copies(gme_seq* s, hipStream_t stream, uint8_t** stage, size_t* stage_bytes, int first, int count,
                         const uint8_t* frames, int row_stride, int64_t frame_stride)
{
    const Plane& p = s->level[2];
    gme_ctx* ctx = s->ctx;
    if (row_stride == s->W && frame_stride == (int64_t)s->W * s->H && p.pitch != s->W && count > 0) {
        // tight frames: linear copies (what the DMA engines move at link speed; 2-D copies of 720-byte rows reach
        // ~37 GB/s) through a device staging buffer, spread into the pitched planes by k_repack
        const size_t frame_bytes = (size_t)s->H * s->W;
        size_t per = ((size_t)64 << 20) / frame_bytes;
        if (per < 1) per = 1;
        if (per > (size_t)count) per = (size_t)count;
        if (per * frame_bytes > *stage_bytes) {
            GME_HIP_TRY(hipStreamSynchronize(stream));
            if (ctx->copy_stream && stage == &ctx->stage) GME_HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
            if (*stage) hipFree(*stage);
            *stage = nullptr; *stage_bytes = 0;
            if (hipMalloc((void**)stage, per * frame_bytes) != hipSuccess) { gme_set_error("out of device memory (staging)"); return GME_ERR_NOMEM; }
            *stage_bytes = per * frame_bytes;
        }
        for (int f0 = 0; f0 < count; f0 += (int)per) {
            const int n = count - f0 < (int)per ? count - f0 : (int)per;
            GME_HIP_TRY(hipMemcpyAsync(*stage, frames + (int64_t)f0 * frame_stride, (size_t)n * frame_bytes, hipMemcpyHostToDevice, stream));
            int rc2 = launch_repack(ctx, stream, *stage, n, s->H, s->W, p.at(first + f0), p.pitch, p.stride);
            if (rc2) return rc2;
        }
    } else if (p.stride == (int64_t)p.pitch * s->H && frame_stride == (int64_t)row_stride * s->H) {
        // planes and host frames are both back to back: one 2-D copy of count*H rows
        GME_HIP_TRY(hipMemcpy2DAsync(p.at(first), p.pitch, frames, row_stride, s->W, (size_t)s->H * count,
                                     hipMemcpyHostToDevice, stream));
    } else {
        for (int i = 0; i < count; ++i)
            GME_HIP_TRY(hipMemcpy2DAsync(p.at(first + i), p.pitch, frames + (int64_t)i * frame_stride, row_stride, s->W,
                                         s->H, hipMemcpyHostToDevice, stream));
    }
    return GME_OK;
}

extern "C" int gme_seq_upload(gme_seq* s, int first, int count, const uint8_t* frames, int row_stride,
                              int64_t frame_stride)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    GME_ENTER(s->ctx);
    GME_REQUIRE(frames && first >= 0 && count >= 0 && first + count <= s->N_cap && row_stride >= s->W, GME_ERR_ARG,
                "gme_seq_upload: frames [%d, %d) outside the sequence of %d", first, first + count, s->N_cap);
    gme_ctx* ctx = s->ctx;
    s->pyramids_valid = false;
    s->sqbox_valid[0] = s->sqbox_valid[1] = s->sqbox_valid[2] = false;
    gme_drop_run(s);
    if (s->split_phase && ctx->device >= 0 && ctx->device < 16) {
        // split-phase: the copies are queued and the call returns; `frames` must stay untouched until a later
        // gme_seq_wait / gme_sync on this sequence has returned.  They run on the device's shared upload stream, behind
        // whatever of this context still reads the planes (gate) and in front of whatever it queues next (s->uploaded);
        // with page-locked frames (gme_host_alloc) the host thread is free at once and every context's kernels run beside the copy.
        Uploader& u = g_uploader[ctx->device];
        std::lock_guard<std::mutex> ulock(u.mu);
        if (!u.stream) GME_HIP_TRY(hipStreamCreateWithFlags(&u.stream, hipStreamNonBlocking));
        if (!s->upload_gate) {
            GME_HIP_TRY(hipEventCreateWithFlags(&s->upload_gate, hipEventDisableTiming));
            GME_HIP_TRY(hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming));
        }
        GME_HIP_TRY(hipEventRecord(s->upload_gate, ctx->stream));
        GME_HIP_TRY(hipStreamWaitEvent(u.stream, s->upload_gate, 0));
        int rc = upload_copies(s, u.stream, &u.stage, &u.stage_bytes, first, count, frames, row_stride, frame_stride);
        if (rc) return rc;
        GME_HIP_TRY(hipEventRecord(s->uploaded, u.stream));
        GME_HIP_TRY(hipStreamWaitEvent(ctx->stream, s->uploaded, 0));
        GME_HIP_TRY(hipEventRecord(s->ready, ctx->stream));
        return GME_OK;
    }
    int rc = upload_copies(s, ctx->stream, &ctx->stage, &ctx->stage_bytes, first, count, frames, row_stride, frame_stride);
    if (rc) return rc;
    GME_HIP_TRY(hipStreamSynchronize(ctx->stream));      // the host buffer may be reused on return
    return GME_OK;
}

extern "C" int gme_seq_synth(gme_seq* s, uint64_t seed, int t0)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    GME_ENTER(s->ctx);
    int rc = GME_OK;
    if (!s->synth_canvas) {
        if (hipMalloc((void**)&s->synth_canvas, (size_t)2048 * 4096) != hipSuccess) {
            gme_set_error("out of device memory (synthetic canvas)");
            return GME_ERR_NOMEM;
        }
        s->synth_valid = false;
    }
    if (!s->synth_valid || s->synth_seed != seed) {
        rc = launch_synth_canvas(s->ctx, seed, s->synth_canvas);
        if (rc) return rc;
        s->synth_seed = seed; s->synth_valid = true;
    }
    s->pyramids_valid = false;
    s->sqbox_valid[0] = s->sqbox_valid[1] = s->sqbox_valid[2] = false;
    gme_drop_run(s);
    return launch_synth_frames(s->ctx, seed, t0, s->synth_canvas, s->level[2]);
}

extern "C" int gme_seq_invalidate(gme_seq* s)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    s->pyramids_valid = false;
    s->sqbox_valid[0] = s->sqbox_valid[1] = s->sqbox_valid[2] = false;
    gme_drop_run(s);
    return GME_OK;
}

extern "C" int gme_seq_read_frame(gme_seq* s, int level, int index, uint8_t* out)
{
    GME_REQUIRE(s != nullptr && out != nullptr, GME_ERR_ARG, "gme_seq_read_frame: null pointer");
    GME_ENTER(s->ctx);
    GME_REQUIRE(level >= 0 && level <= 2 && index >= 0 && index < s->N_cap, GME_ERR_ARG, "gme_seq_read_frame: bad index");
    GME_REQUIRE(level == 2 || s->pyramids_valid, GME_ERR_STATE, "pyramid levels exist only after gme_seq_gme_begin");
    const Plane& p = s->level[level];
    GME_HIP_TRY(hipMemcpy2DAsync(out, p.W, p.at(index), p.pitch, p.W, p.H, hipMemcpyDeviceToHost, s->ctx->stream));
    return ctx_finish(s->ctx);
}

template <typename T>
static int ensure(T** ptr, size_t* have, size_t want_bytes)
{
    if (*ptr && *have >= want_bytes) return GME_OK;
    if (*ptr) hipFree(*ptr);
    *ptr = nullptr; *have = 0;
    if (want_bytes == 0) return GME_OK;
    if (hipMalloc((void**)ptr, want_bytes) != hipSuccess) {
        gme_set_error("out of device memory (%zu bytes)", want_bytes);
        return GME_ERR_NOMEM;
    }
    *have = want_bytes;
    return GME_OK;
}

// auxiliary table (bbme_aux_kind) for every frame of one pyramid level
static int seq_sqbox(gme_seq* s, int level, int kind)
{
    const Plane& p = s->level[level];
    const size_t bytes = (size_t)p.stride * p.count * sizeof(uint32_t);
    int rc = ensure(&s->sqbox[level], &s->sqbox_bytes[level], bytes);
    if (rc) return rc;
    if (!s->sqbox_valid[level] || s->sqbox_kind[level] != kind) {
        rc = launch_aux_table(s->ctx, kind, p.ptr, p.stride, s->N, p.H, p.W, p.pitch, s->sqbox[level], p.stride);
        if (rc) return rc;
        s->sqbox_valid[level] = true;
        s->sqbox_kind[level] = kind;
    }
    return GME_OK;
}

extern "C" int gme_seq_bbme(gme_seq* s, int fd, int bs, int sw, int procedure, int pnorm)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    GME_ENTER(s->ctx);
    int rc = GME_OK;
    GME_REQUIRE(fd >= 1 && fd < s->N, GME_ERR_ARG, "frame_distance %d needs at least %d frames", fd, fd + 1);
    GME_REQUIRE(bs >= 1, GME_ERR_ARG, "block_size %d", bs);
    rc = bbme_check_args(s->H, s->W, bs, sw, procedure, pnorm);
    if (rc) return rc;
    const int pairs = s->N - fd, h = s->H / bs, w = s->W / bs;
    rc = ensure(&s->mv, &s->mv_bytes, (size_t)pairs * h * w * 2 * sizeof(int32_t));
    if (rc) return rc;
    s->mv_h = h; s->mv_w = w; s->mv_pairs = pairs;
    const Plane& p = s->level[2];
    BbmeJob job;
    job.prev = p.at(0); job.cur = p.at(fd); job.plane_stride = p.stride; job.pairs = pairs;
    job.H = s->H; job.W = s->W; job.pitch = p.pitch;
    job.bs = bs; job.sw = sw; job.procedure = procedure; job.pnorm = pnorm;
    job.mf = s->mv; job.sqbox_cur = nullptr; job.sqbox_stride = 0;
    if (const int aux = bbme_aux_kind(bs, sw, procedure, pnorm)) {
        rc = seq_sqbox(s, 2, aux);
        if (rc) return rc;
        job.sqbox_cur = s->sqbox[2] + (size_t)fd * p.stride;
        job.sqbox_stride = p.stride;
    }
    return launch_bbme(s->ctx, job);
}

// ---------------------------------------------------------------------------
// host frames -> fields with the upload overlapped (results.py:41-50 hands over host arrays)
// ---------------------------------------------------------------------------
extern "C" void* gme_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        gme_set_error("gme_host_alloc: cannot pin %zu bytes", bytes);
        return nullptr;
    }
    return p;
}

extern "C" void gme_host_free(void* p)
{
    if (p) hipHostFree(p);
}

extern "C" int gme_seq_bbme_streamed(gme_seq* s, const uint8_t* frames, int row_stride, int64_t frame_stride, int count,
                                     int fd, int bs, int sw, int procedure, int pnorm, int chunk_frames, int32_t* mf_out)
{
    GME_REQUIRE(s != nullptr && frames != nullptr && mf_out != nullptr, GME_ERR_ARG, "gme_seq_bbme_streamed: null pointer");
    gme_ctx* ctx = s->ctx;
    GME_ENTER(ctx);
    GME_REQUIRE(count >= 1 && count <= s->N && row_stride >= s->W, GME_ERR_ARG,
                "gme_seq_bbme_streamed: %d frames into a sequence of %d", count, s->N);
    GME_REQUIRE(fd >= 1 && fd < count, GME_ERR_ARG, "frame_distance %d needs at least %d frames", fd, fd + 1);
    GME_REQUIRE(bs >= 1 && chunk_frames >= 1, GME_ERR_ARG, "block_size %d, chunk of %d frames", bs, chunk_frames);
    int rc = bbme_check_args(s->H, s->W, bs, sw, procedure, pnorm);
    if (rc) return rc;
    const int pairs = count - fd, h = s->H / bs, w = s->W / bs;
    const size_t per = (size_t)h * w * 2;
    rc = ensure(&s->mv, &s->mv_bytes, (size_t)(s->N - fd) * per * sizeof(int32_t));
    if (rc) return rc;
    s->mv_h = h; s->mv_w = w; s->mv_pairs = pairs;
    const Plane& p = s->level[2];
    const int aux = bbme_aux_kind(bs, sw, procedure, pnorm);
    if (aux) {
        rc = ensure(&s->sqbox[2], &s->sqbox_bytes[2], (size_t)p.stride * p.count * sizeof(uint32_t));
        if (rc) return rc;
    }
    if (!ctx->copy_stream) {
        GME_HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        GME_HIP_TRY(hipStreamCreateWithFlags(&ctx->back_stream, hipStreamNonBlocking));
    }
    s->pyramids_valid = false;
    s->sqbox_valid[0] = s->sqbox_valid[1] = s->sqbox_valid[2] = false;
    gme_drop_run(s);
    // tight frames (the usual NumPy stack) whose rows are narrower than the plane pitch: linear copy + repack
    const bool tight = row_stride == s->W && frame_stride == (int64_t)s->W * s->H && p.pitch != s->W;
    // page-locked source (gme_host_alloc / hipHostMalloc)?  GME_UPLOAD_ZEROCOPY=1 lets the repack kernel read it across
    // the link itself instead of the copy engines (measured slower: 34 vs 40 GB/s)
    bool host_mapped = false;
    const uint8_t* dev_view = nullptr;
    if (tight && getenv("GME_UPLOAD_ZEROCOPY")) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, frames) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr) {
            host_mapped = true;
            dev_view = (const uint8_t*)at.devicePointer;
        } else {
            (void)hipGetLastError();                           // pageable memory: not an error
        }
    }
    // GME_UPLOAD_ENGINES=2 splits every chunk over two streams (two copy engines); measured slower on MI355X boxes of
    // this pool (22-25 GB/s against 38-39 GB/s for one stream), so one is the default
    const bool two_engines = getenv("GME_UPLOAD_ENGINES") && atoi(getenv("GME_UPLOAD_ENGINES")) >= 2;
    if (tight && two_engines && !ctx->copy_stream2) GME_HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream2, hipStreamNonBlocking));
    if (tight && !host_mapped) {
        const size_t want = (size_t)(chunk_frames < count ? chunk_frames : count) * s->H * s->W;
        if (want > ctx->stage_bytes) {
            GME_HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
            if (ctx->stage) hipFree(ctx->stage);
            ctx->stage = nullptr; ctx->stage_bytes = 0;
            if (hipMalloc((void**)&ctx->stage, want) != hipSuccess) { gme_set_error("out of device memory (%zu bytes of staging)", want); return GME_ERR_NOMEM; }
            ctx->stage_bytes = want;
        }
    }
    const int nchunks = (count + chunk_frames - 1) / chunk_frames;
    std::vector<hipEvent_t> up(nchunks, nullptr), done(nchunks, nullptr);
    auto cleanup = [&]() { for (auto e : up) if (e) hipEventDestroy(e); for (auto e : done) if (e) hipEventDestroy(e); };
#define STREAM_TRY(expr) do { if ((expr) != hipSuccess) { gme_set_error("%s failed: %s", #expr, hipGetErrorString(hipGetLastError())); \
                               hipStreamSynchronize(ctx->copy_stream); hipStreamSynchronize(ctx->back_stream); hipStreamSynchronize(ctx->stream); \
                               cleanup(); return GME_ERR_HIP; } } while (0)
    // earlier work of this context may still read or write the planes / the field buffer
    hipEvent_t gate = nullptr;
    STREAM_TRY(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    up.push_back(gate);                                        // destroyed with the others
    STREAM_TRY(hipEventRecord(gate, ctx->stream));
    STREAM_TRY(hipStreamWaitEvent(ctx->copy_stream, gate, 0));
    STREAM_TRY(hipStreamWaitEvent(ctx->back_stream, gate, 0));
    int p_done = 0;                                            // pairs searched so far
    int tab_done = 0;                                          // frames whose auxiliary table rows exist (always from frame 0:
                                                               // a later call with a smaller frame distance reads them as `cur`)
    int back_first = 0, back_count = 0, back_chunk = -1;       // fields of the previous chunk, still to be read back
    auto read_back = [&]() -> bool {
        if (back_count == 0) return true;
        if (hipStreamWaitEvent(ctx->back_stream, done[back_chunk], 0) != hipSuccess) return false;
        const bool ok = hipMemcpyAsync(mf_out + per * back_first, s->mv + per * back_first, per * back_count * sizeof(int32_t),
                                       hipMemcpyDeviceToHost, ctx->back_stream) == hipSuccess;
        back_count = 0;
        return ok;
    };
#define LTRY(expr) do { if ((expr) != hipSuccess) return false; } while (0)
    // upload of chunk c on the copy stream + the event behind it
    auto upload_chunk = [&](int c) -> bool {
        const int f0 = c * chunk_frames, f1 = f0 + chunk_frames < count ? f0 + chunk_frames : count;
        LTRY(hipEventCreateWithFlags(&up[c], hipEventDisableTiming));
        LTRY(hipEventCreateWithFlags(&done[c], hipEventDisableTiming));
        // upload of chunk c on the copy stream: it runs while the compute stream still searches chunk c - 1
        if (tight && host_mapped) {
            // page-locked frames are mapped into the device's address space: the repack kernel reads them across the
            // link itself (thousands of loads in flight instead of one DMA queue) and writes the pitched planes
            if (launch_repack(ctx, ctx->copy_stream, dev_view + (int64_t)f0 * frame_stride, f1 - f0, s->H, s->W, p.at(f0), p.pitch, p.stride) != GME_OK) return false;
        } else if (tight) {
            // one linear copy (link speed), then spread into the pitched planes on the same stream: the next chunk's
            // copy into the staging buffer is ordered behind this repack
            const size_t bytes = (size_t)(f1 - f0) * s->H * s->W;
            size_t head = bytes;
            if (two_engines && bytes >= (1u << 20)) {
                // second half through a second stream: two copy engines share the link (one alone moved ~40 GB/s)
                head = (bytes / 2) & ~(size_t)4095;
                if (c > 0) LTRY(hipStreamWaitEvent(ctx->copy_stream2, up[c - 1], 0));      // staging buffer free again
                LTRY(hipMemcpyAsync(ctx->stage + head, frames + (int64_t)f0 * frame_stride + head, bytes - head,
                                          hipMemcpyHostToDevice, ctx->copy_stream2));
                LTRY(hipEventRecord(done[c], ctx->copy_stream2));                          // done[c] is re-recorded behind the kernel below
            }
            LTRY(hipMemcpyAsync(ctx->stage, frames + (int64_t)f0 * frame_stride, head, hipMemcpyHostToDevice, ctx->copy_stream));
            if (head != bytes) LTRY(hipStreamWaitEvent(ctx->copy_stream, done[c], 0));
            if (launch_repack(ctx, ctx->copy_stream, ctx->stage, f1 - f0, s->H, s->W, p.at(f0), p.pitch, p.stride) != GME_OK) return false;
        } else if (p.stride == (int64_t)p.pitch * s->H && frame_stride == (int64_t)row_stride * s->H) {
            LTRY(hipMemcpy2DAsync(p.at(f0), p.pitch, frames + (int64_t)f0 * frame_stride, row_stride, s->W,
                                        (size_t)s->H * (f1 - f0), hipMemcpyHostToDevice, ctx->copy_stream));
        } else {
            for (int i = f0; i < f1; ++i)
                LTRY(hipMemcpy2DAsync(p.at(i), p.pitch, frames + (int64_t)i * frame_stride, row_stride, s->W, s->H,
                                            hipMemcpyHostToDevice, ctx->copy_stream));
        }
        LTRY(hipEventRecord(up[c], ctx->copy_stream));
        return true;
    };
#undef LTRY
    if (!upload_chunk(0)) { STREAM_TRY(hipErrorUnknown); }
    for (int c = 0; c < nchunks; ++c) {
        const int f1 = (c + 1) * chunk_frames < count ? (c + 1) * chunk_frames : count;
        // The NEXT chunk's upload is queued before this chunk's search is launched: a kernel launch behind a cross-stream
        // wait on a copy holds the calling thread until that copy is done, and with the upload queued only afterwards the
        // link idled for the host's latency between chunks (94 % of the copy-alone rate; round 3).
        if (c + 1 < nchunks && !upload_chunk(c + 1)) { STREAM_TRY(hipErrorUnknown); }
        // read-back of the previous chunk's fields only now, behind the next chunk's upload in program order: a copy
        // into pageable memory may hold the calling thread until that chunk's kernel is done, and the upload
        // queued above keeps the link busy meanwhile
        if (!read_back()) { STREAM_TRY(hipErrorUnknown); }
        STREAM_TRY(hipStreamWaitEvent(ctx->stream, up[c], 0));
        const int p1 = f1 - fd;                                // pairs [p_done, p1) have both frames on the device now
        if (p1 > p_done) {
            BbmeJob job;
            job.prev = p.at(p_done); job.cur = p.at(p_done + fd); job.plane_stride = p.stride; job.pairs = p1 - p_done;
            job.H = s->H; job.W = s->W; job.pitch = p.pitch;
            job.bs = bs; job.sw = sw; job.procedure = procedure; job.pnorm = pnorm;
            job.mf = s->mv + per * p_done; job.sqbox_cur = nullptr; job.sqbox_stride = 0;
            job.chained = p_done > 0;
            if (aux) {
                // box sums of squares of every frame uploaded since the last table launch
                const int t0 = tab_done;
                rc = launch_aux_table(ctx, aux, p.at(t0), p.stride, f1 - t0, p.H, p.W, p.pitch, s->sqbox[2] + (size_t)t0 * p.stride, p.stride);
                tab_done = f1;
                if (rc == GME_OK) { job.sqbox_cur = s->sqbox[2] + (size_t)(p_done + fd) * p.stride; job.sqbox_stride = p.stride; }
            }
            if (rc == GME_OK) rc = launch_bbme(ctx, job);
            if (rc != GME_OK) {
                hipStreamSynchronize(ctx->copy_stream); hipStreamSynchronize(ctx->back_stream); hipStreamSynchronize(ctx->stream);
                cleanup();
                return rc;
            }
            STREAM_TRY(hipEventRecord(done[c], ctx->stream));
            back_first = p_done; back_count = p1 - p_done; back_chunk = c;
            p_done = p1;
        }
    }
    if (!read_back()) { STREAM_TRY(hipErrorUnknown); }
#undef STREAM_TRY
    hipError_t e1 = hipStreamSynchronize(ctx->copy_stream), e2 = hipStreamSynchronize(ctx->back_stream);
    if (ctx->copy_stream2 && hipStreamSynchronize(ctx->copy_stream2) != hipSuccess) e1 = hipErrorUnknown;
    rc = ctx_finish(ctx);
    cleanup();
    if (aux) { s->sqbox_valid[2] = (count == s->N && tab_done == count); s->sqbox_kind[2] = aux; }
    if (e1 != hipSuccess || e2 != hipSuccess) { gme_set_error("gme_seq_bbme_streamed: copy stream failed"); return GME_ERR_HIP; }
    return rc;
}

extern "C" int gme_seq_read_mv(gme_seq* s, int first_pair, int count, int32_t* mf_out)
{
    GME_REQUIRE(s != nullptr && mf_out != nullptr, GME_ERR_ARG, "gme_seq_read_mv: null pointer");
    GME_ENTER(s->ctx);
    GME_REQUIRE(s->mv != nullptr, GME_ERR_STATE, "gme_seq_read_mv before gme_seq_bbme");
    GME_REQUIRE(first_pair >= 0 && count >= 0 && first_pair + count <= s->mv_pairs, GME_ERR_ARG,
                "gme_seq_read_mv: pairs [%d, %d) outside [0, %d)", first_pair, first_pair + count, s->mv_pairs);
    const size_t per = (size_t)s->mv_h * s->mv_w * 2;
    GME_HIP_TRY(hipMemcpyAsync(mf_out, s->mv + per * first_pair, per * count * sizeof(int32_t), hipMemcpyDeviceToHost,
                               s->ctx->stream));
    return ctx_finish(s->ctx);
}

static int alloc_fit(FitLevelBuf& f, int pairs, int h, int w, bool full)
{
    free_fit(f);
    f.h = h; f.w = w;
    const size_t n = (size_t)pairs * h * w;
    if (n == 0) return GME_OK;
    bool ok = hipMalloc((void**)&f.gt, n * 2 * sizeof(int32_t)) == hipSuccess;
    if (full) {
        ok = ok && hipMalloc((void**)&f.model, n * 2 * sizeof(int16_t)) == hipSuccess;
        ok = ok && hipMalloc((void**)&f.mask, n) == hipSuccess;
        ok = ok && hipMalloc((void**)&f.diff, n * sizeof(int32_t)) == hipSuccess;
        if ((size_t)h * w * 16 > 40 * 1024) ok = ok && hipMalloc(&f.list, n * 16) == hipSuccess;
        ok = ok && hipMalloc((void**)&f.thr, (size_t)pairs * sizeof(int32_t)) == hipSuccess;
        ok = ok && hipMalloc((void**)&f.sums, (size_t)pairs * 15 * sizeof(double)) == hipSuccess;
    }
    if (!ok) { gme_set_error("out of device memory (GME level buffers)"); return GME_ERR_NOMEM; }
    return GME_OK;
}


// BBME of one pyramid level for the pairs of the current GME run (gme_seq_gme_begin's arguments):
// level 0 = dense field (bs 2, diamond, MSE: motion.py:27-29; bbme.py:18 default norm).
static int gme_level_bbme(gme_seq* s, int l)
{
    if (!s->bbme_pending[l]) return GME_OK;
    s->bbme_pending[l] = false;
    const Plane& p = s->level[l];
    const int fd = s->gme_fd;
    BbmeJob job;
    job.prev = p.at(0); job.cur = p.at(fd); job.plane_stride = p.stride; job.pairs = s->gme_pairs;
    job.H = p.H; job.W = p.W; job.pitch = p.pitch;
    job.bs = l == 0 ? 2 : s->gme_bs;
    job.sw = l == 0 ? 2 : s->gme_sw;
    job.procedure = l == 0 ? GME_SEARCH_DIAMOND : s->gme_procedure;
    job.pnorm = GME_NORM_MSE;
    job.mf = s->fit[l].gt; job.sqbox_cur = nullptr; job.sqbox_stride = 0;
    if (s->fit[l].h == 0 || s->fit[l].w == 0) return GME_OK;
    if (const int aux = bbme_aux_kind(job.bs, job.sw, job.procedure, job.pnorm)) {      // BASELINE config 4
        const int rc = seq_sqbox(s, l, aux);
        if (rc) return rc;
        job.sqbox_cur = s->sqbox[l] + (size_t)fd * p.stride;
        job.sqbox_stride = p.stride;
    }
    return launch_bbme(s->ctx, job);
}

// launch whatever gme_seq_gme_begin deferred, up to and including `level`
static int gme_flush_bbme(gme_seq* s, int level)
{
    for (int l = 0; l <= level && l <= 2; ++l) {
        const int rc = gme_level_bbme(s, l);
        if (rc) return rc;
    }
    return GME_OK;
}

// pyramids, buffers, dense field and first parameters (motion.py:123-128,160-188) of a staged run; the context is locked
static int gme_begin_common(gme_seq* s, int fd, int bbme_bs, int procedure, int sw)
{
    gme_ctx* ctx = s->ctx;
    int rc = GME_OK;
    GME_REQUIRE(fd >= 1 && fd < s->N, GME_ERR_ARG, "frame_distance %d needs at least %d frames", fd, fd + 1);
    GME_REQUIRE(bbme_bs >= 1, GME_ERR_ARG, "block_size %d", bbme_bs);
    const int pairs = s->N - fd;
    // pyramids (utils.py:34-51): level 1 = pyrDown(level 2), level 0 = pyrDown(level 1)
    for (int l = 1; l >= 0; --l) {
        const Plane& src = s->level[l + 1];
        if (!s->level[l].ptr) {
            rc = plane_alloc(ctx, &s->level[l], s->N_cap, (src.H + 1) / 2, (src.W + 1) / 2);
            if (rc) return rc;
        }
    }
    // argument checks for the three BBME runs before anything is launched
    rc = bbme_check_args(s->level[0].H, s->level[0].W, 2, 2, GME_SEARCH_DIAMOND, GME_NORM_MSE);
    if (rc) return rc;
    for (int l = 1; l <= 2; ++l) {
        rc = bbme_check_args(s->level[l].H, s->level[l].W, bbme_bs, sw, procedure, GME_NORM_MSE);
        if (rc) return rc;
    }
    if (!s->pyramids_valid) {
        for (int l = 1; l >= 0; --l) {
            Plane src = s->level[l + 1], dst = s->level[l];       // views over the frames in use (gme_seq_set_frames)
            src.count = dst.count = s->N;
            rc = launch_pyrdown(ctx, src, dst);
            if (rc) return rc;
        }
        s->pyramids_valid = true;
        s->sqbox_valid[0] = s->sqbox_valid[1] = false;
    }
    const int cap_pairs = s->N_cap - fd;                   // buffers hold the sequence's full count: gme_seq_set_frames never reallocates
    if (s->gme_alloc_pairs < (size_t)cap_pairs || s->gme_bs != bbme_bs) {
        rc = alloc_fit(s->fit[0], cap_pairs, s->level[0].H / 2, s->level[0].W / 2, false);
        if (rc) return rc;
        for (int l = 1; l <= 2; ++l) {
            rc = alloc_fit(s->fit[l], cap_pairs, s->level[l].H / bbme_bs, s->level[l].W / bbme_bs, true);
            if (rc) return rc;
        }
        if (s->params0) hipFree(s->params0);
        if (s->params_in) hipFree(s->params_in);
        if (s->solve_flags) hipFree(s->solve_flags);
        s->params0 = nullptr; s->params_in = nullptr; s->solve_flags = nullptr;
        if (hipMalloc((void**)&s->params0, (size_t)cap_pairs * 6 * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&s->solve_flags, (size_t)(cap_pairs > 0 ? cap_pairs : 1) * sizeof(int32_t)) != hipSuccess ||
            hipMalloc((void**)&s->params_in, (size_t)cap_pairs * 6 * sizeof(double)) != hipSuccess) {
            gme_set_error("out of device memory (parameters)");
            return GME_ERR_NOMEM;
        }
        s->gme_alloc_pairs = cap_pairs;
    }
    s->gme_fd = fd; s->gme_bs = bbme_bs; s->gme_pairs = pairs;
    s->gme_procedure = procedure; s->gme_sw = sw;
    s->bbme_pending[0] = s->bbme_pending[1] = s->bbme_pending[2] = true;
    // Launch order: dense field + first parameters, hand those to the host, and only then the level-1
    // search; the level-2 search (the longest kernel) is launched by gme_seq_gme_fit(level 1) after
    // it has the level-1 sums.  The host's projection / 3x3 solves between the stages then run
    // while the GPU searches the next level instead of leaving it idle (motion.py:123-136 is a
    // strict chain only through the small fits, not through the searches).
    rc = gme_level_bbme(s, 0);
    if (rc) return rc;
    GME_REQUIRE(s->fit[0].h > 0 && s->fit[0].w > 0, GME_ERR_GEOMETRY, "frames too small for a dense field");
    return launch_first_params(ctx, s->fit[0].gt, pairs, s->fit[0].h * s->fit[0].w, s->params0);
}

extern "C" int gme_seq_gme_begin(gme_seq* s, int fd, int bbme_bs, int procedure, int sw, float* params0_out)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    gme_ctx* ctx = s->ctx;
    GME_ENTER(ctx);
    int rc = gme_begin_common(s, fd, bbme_bs, procedure, sw);
    if (rc) return rc;
    if (params0_out) {
        rc = copy_small(ctx, params0_out, s->params0, (size_t)s->gme_pairs * 6 * sizeof(float), hipMemcpyDeviceToHost, s->split_phase);
        if (rc) return rc;
        if (s->split_phase) GME_HIP_TRY(hipEventRecord(s->ready, ctx->stream));
        else rc = ctx_finish(ctx);
        if (rc) return rc;
        return gme_level_bbme(s, 1);                       // runs while the caller projects the parameters
    }
    return gme_flush_bbme(s, 2);
}

static int fit_level_launch(gme_seq* s, int level, const double* dparams, double outlier_fraction, double* sums_out);

// gme_seq_gme_begin + the projection of the first parameters (motion.py:191-207 on the float32 vector: two exact doublings)
// + gme_seq_gme_fit(level 1) without the trip to the host in between: the first parameters never leave the device on the
// way to the level-1 fit, so a staged run has three dependent host round trips instead of four.  params0_out may be NULL.
extern "C" int gme_seq_gme_begin_fit(gme_seq* s, int fd, int bbme_bs, int procedure, int sw, double outlier_fraction,
                                     float* params0_out, double* sums1_out)
{
    GME_REQUIRE(s != nullptr && sums1_out != nullptr, GME_ERR_ARG, "gme_seq_gme_begin_fit: null pointer");
    gme_ctx* ctx = s->ctx;
    GME_ENTER(ctx);
    int rc = gme_begin_common(s, fd, bbme_bs, procedure, sw);
    if (rc) return rc;
    rc = launch_project_first(ctx, s->params0, s->gme_pairs, s->params_in);
    if (rc) return rc;
    if (params0_out)
        { rc = copy_small(ctx, params0_out, s->params0, (size_t)s->gme_pairs * 6 * sizeof(float), hipMemcpyDeviceToHost, s->split_phase); if (rc) return rc; }
    rc = gme_level_bbme(s, 1);
    if (rc) return rc;
    rc = fit_level_launch(s, 1, s->params_in, outlier_fraction, sums1_out);
    if (rc) return rc;
    return gme_level_bbme(s, 2);                           // searched while the caller solves level 1
}

static int ensure_fit_mv(gme_seq* s)
{
    FitLevelBuf& f = s->fit_mv;
    if (f.h == s->mv_h && f.w == s->mv_w && s->fit_mv_pairs == s->mv_pairs && f.model) { f.gt = s->mv; return GME_OK; }
    f.gt = nullptr;                       // borrowed from s->mv, never freed here
    if (f.model) hipFree(f.model);
    if (f.mask) hipFree(f.mask);
    if (f.diff) hipFree(f.diff);
    if (f.list) hipFree(f.list);
    if (f.thr) hipFree(f.thr);
    if (f.sums) hipFree(f.sums);
    f = FitLevelBuf();
    f.h = s->mv_h; f.w = s->mv_w;
    const size_t n = (size_t)s->mv_pairs * f.h * f.w;
    if (hipMalloc((void**)&f.model, n * 2 * sizeof(int16_t)) != hipSuccess || hipMalloc((void**)&f.mask, n) != hipSuccess ||
        hipMalloc((void**)&f.diff, n * sizeof(int32_t)) != hipSuccess ||
        ((size_t)f.h * f.w * 16 > 40 * 1024 && hipMalloc(&f.list, n * 16) != hipSuccess) ||
        hipMalloc((void**)&f.thr, (size_t)s->mv_pairs * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void**)&f.sums, (size_t)s->mv_pairs * 15 * sizeof(double)) != hipSuccess) {
        gme_set_error("out of device memory (fit buffers)");
        return GME_ERR_NOMEM;
    }
    f.gt = s->mv;
    s->fit_mv_pairs = s->mv_pairs;
    return GME_OK;
}

extern "C" int gme_seq_gme_fit(gme_seq* s, int level, const double* params_in, double outlier_fraction,
                               double* sums_out)
{
    GME_REQUIRE(s != nullptr && params_in != nullptr && sums_out != nullptr, GME_ERR_ARG, "gme_seq_gme_fit: null pointer");
    gme_ctx* ctx = s->ctx;
    GME_ENTER(ctx);
    int rc = GME_OK;
    GME_REQUIRE(level == 1 || level == 2 || level == -1, GME_ERR_ARG, "gme_seq_gme_fit: level %d (1, 2 or -1)", level);
    int pairs, level_H, level_W;
    const FitLevelBuf* f;
    double* dparams;
    if (level == -1) {
        GME_REQUIRE(s->mv != nullptr && s->mv_pairs > 0, GME_ERR_STATE, "gme_seq_gme_fit(level -1) before gme_seq_bbme");
        rc = ensure_fit_mv(s);
        if (rc) return rc;
        f = &s->fit_mv; pairs = s->mv_pairs; level_H = s->H; level_W = s->W;
        size_t have = s->mv_params_bytes;
        rc = ensure(&s->mv_params, &have, (size_t)pairs * 6 * sizeof(double));
        s->mv_params_bytes = have;
        if (rc) return rc;
        dparams = s->mv_params;
    } else {
        GME_REQUIRE(s->gme_pairs > 0 && s->params_in, GME_ERR_STATE, "gme_seq_gme_fit before gme_seq_gme_begin");
        rc = gme_flush_bbme(s, level);
        if (rc) return rc;
        f = &s->fit[level]; pairs = s->gme_pairs; level_H = s->level[level].H; level_W = s->level[level].W;
        dparams = s->params_in;
    }
    if (level == -1) {
        const int n = f->h * f->w;
        GME_REQUIRE(n > 0, GME_ERR_GEOMETRY, "level %d holds no block (motion.py:243 would index an empty list)", level);
        // int(0.3 * len), motion.py:242; a negative fraction selects the unmasked fit (motion.py:33-88)
        const int drop = outlier_fraction < 0 ? -1 : (int)(outlier_fraction * (double)n);
        GME_REQUIRE(drop <= n, GME_ERR_ARG, "outlier fraction %g out of range", outlier_fraction);
        GME_HIP_TRY(hipMemcpyAsync(dparams, params_in, (size_t)pairs * 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        rc = launch_fit_level(ctx, f->gt, pairs, f->h, f->w, dparams, drop, level_H, level_W, f->model, f->mask, f->diff,
                              f->thr, f->sums, f->list);
        if (rc) return rc;
        GME_HIP_TRY(hipMemcpyAsync(sums_out, f->sums, (size_t)pairs * 15 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (s->split_phase) GME_HIP_TRY(hipEventRecord(s->ready, ctx->stream));
        else rc = ctx_finish(ctx);
        return rc;
    }
    rc = copy_small(ctx, dparams, params_in, (size_t)pairs * 6 * sizeof(double), hipMemcpyHostToDevice, s->split_phase);
    if (rc) return rc;
    rc = fit_level_launch(s, level, dparams, outlier_fraction, sums_out);
    if (rc) return rc;
    if (level == 1) return gme_level_bbme(s, 2);           // searched while the caller solves level 1
    return GME_OK;
}

// model field, mask and sums of GME level 1 or 2 from parameters already on the device; result copy + event / wait
static int fit_level_launch(gme_seq* s, int level, const double* dparams, double outlier_fraction, double* sums_out)
{
    gme_ctx* ctx = s->ctx;
    const FitLevelBuf* f = &s->fit[level];
    const int pairs = s->gme_pairs, n = f->h * f->w;
    GME_REQUIRE(n > 0, GME_ERR_GEOMETRY, "level %d holds no block (motion.py:243 would index an empty list)", level);
    // int(0.3 * len), motion.py:242; a negative fraction selects the unmasked fit (motion.py:33-88)
    const int drop = outlier_fraction < 0 ? -1 : (int)(outlier_fraction * (double)n);
    GME_REQUIRE(drop <= n, GME_ERR_ARG, "outlier fraction %g out of range", outlier_fraction);
    int rc = launch_fit_level(ctx, f->gt, pairs, f->h, f->w, dparams, drop, s->level[level].H, s->level[level].W, f->model, f->mask,
                              f->diff, f->thr, f->sums, f->list);
    if (rc) return rc;
    if (!sums_out) return GME_OK;                          // gme_seq_gme_device_solve: the sums stay on the device
    rc = copy_small(ctx, sums_out, f->sums, (size_t)pairs * 15 * sizeof(double), hipMemcpyDeviceToHost, s->split_phase);
    if (rc) return rc;
    if (s->split_phase) GME_HIP_TRY(hipEventRecord(s->ready, ctx->stream));
    else rc = ctx_finish(ctx);
    return rc;
}

extern "C" int gme_seq_gme_read_stage(gme_seq* s, int level, int pair, int32_t* gt, int16_t* model, uint8_t* mask,
                                      int64_t* threshold)
{
    GME_REQUIRE(s != nullptr, GME_ERR_ARG, "null sequence");
    gme_ctx* ctx = s->ctx;
    GME_ENTER(ctx);
    int rc = GME_OK;
    GME_REQUIRE(level >= -1 && level <= 2 && pair >= 0 && pair < (level < 0 ? s->fit_mv_pairs : s->gme_pairs), GME_ERR_ARG,
                "gme_seq_gme_read_stage: bad index");
    if (level >= 0) {
        rc = gme_flush_bbme(s, level);
        if (rc) return rc;
    }
    const FitLevelBuf& f = level < 0 ? s->fit_mv : s->fit[level];
    const size_t n = (size_t)f.h * f.w;
    int32_t thr = 0;
    if (gt && n) GME_HIP_TRY(hipMemcpyAsync(gt, f.gt + n * 2 * pair, n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (level != 0) {
        if (model && n) GME_HIP_TRY(hipMemcpyAsync(model, f.model + n * 2 * pair, n * 2 * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->stream));
        if (mask && n) GME_HIP_TRY(hipMemcpyAsync(mask, f.mask + n * pair, n, hipMemcpyDeviceToHost, ctx->stream));
        if (threshold) GME_HIP_TRY(hipMemcpyAsync(&thr, f.thr + pair, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    rc = ctx_finish(ctx);
    if (threshold) *threshold = thr;
    return rc;
}

static int ensure_comp(gme_seq* s, int fd, int pairs)
{
    gme_ctx* ctx = s->ctx;
    if (!s->comp.ptr || s->comp.count < pairs) {
        const int cap_pairs = s->N_cap - fd > pairs ? s->N_cap - fd : pairs;
        plane_free(&s->comp);
        int rc = plane_alloc(ctx, &s->comp, cap_pairs, s->H, s->W);
        if (rc) return rc;
        if (s->sse) hipFree(s->sse);
        if (s->comp_params) hipFree(s->comp_params);
        s->sse = nullptr; s->comp_params = nullptr;
        if (hipMalloc((void**)&s->sse, (size_t)cap_pairs * sizeof(unsigned long long)) != hipSuccess ||
            hipMalloc((void**)&s->comp_params, (size_t)cap_pairs * 6 * sizeof(double)) != hipSuccess) {
            gme_set_error("out of device memory");
            return GME_ERR_NOMEM;
        }
    }
    return GME_OK;
}

extern "C" int gme_seq_compensate(gme_seq* s, int fd, int bs, const double* params, int64_t* sse_out)
{
    GME_REQUIRE(s != nullptr && params != nullptr, GME_ERR_ARG, "gme_seq_compensate: null pointer");
    gme_ctx* ctx = s->ctx;
    GME_ENTER(ctx);
    int rc = GME_OK;
    GME_REQUIRE(fd >= 1 && fd < s->N, GME_ERR_ARG, "frame_distance %d needs at least %d frames", fd, fd + 1);
    const int pairs = s->N - fd;
    GME_REQUIRE(bs >= 1, GME_ERR_ARG, "block_size %d", bs);
    const int h = s->H / bs, w = s->W / bs;
    GME_REQUIRE(h > 0 && w > 0, GME_ERR_GEOMETRY, "block_size %d does not fit a %d x %d frame", bs, s->H, s->W);
    rc = ensure_comp(s, fd, pairs);
    if (rc) return rc;
    rc = copy_small(ctx, s->comp_params, params, (size_t)pairs * 6 * sizeof(double), hipMemcpyHostToDevice, s->split_phase);
    if (rc) return rc;
    const Plane& p = s->level[2];
    rc = launch_compensate(ctx, p.at(0), p.stride, pairs, s->H, s->W, p.pitch, nullptr, s->comp_params, h, w, s->comp.ptr,
                           s->comp.stride, s->comp.pitch, p.at(fd), p.stride, s->sse);
    if (rc) return rc;
    if (sse_out) {
        rc = copy_small(ctx, sse_out, s->sse, (size_t)pairs * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->split_phase);
        if (rc) return rc;
        if (s->split_phase) { GME_HIP_TRY(hipEventRecord(s->ready, ctx->stream)); return GME_OK; }
        return ctx_finish(ctx);
    }
    return GME_OK;
}

// `count` compensated frames starting at pair `first` into out[count][H][W] (tight) with ONE wait: what results.py's
// per-pair imwrite loop (results.py:59-76) needs of a finished chunk (a blocking read per pair stalled the one host
// thread that drives all lanes of sequence.StreamEstimator).
extern "C" int gme_seq_read_compensated_range(gme_seq* s, int first, int count, uint8_t* out)
{
    GME_REQUIRE(s != nullptr && out != nullptr, GME_ERR_ARG, "gme_seq_read_compensated_range: null pointer");
    GME_ENTER(s->ctx);
    GME_REQUIRE(s->comp.ptr && first >= 0 && count >= 0 && first + count <= s->comp.count, GME_ERR_STATE,
                "gme_seq_read_compensated_range: pairs %d .. %d of %d", first, first + count, s->comp.ptr ? s->comp.count : 0);
    for (int k = 0; k < count; ++k)
        GME_HIP_TRY(hipMemcpy2DAsync(out + (size_t)k * s->H * s->W, s->W, s->comp.at(first + k), s->comp.pitch, s->W, s->H,
                                     hipMemcpyDeviceToHost, s->ctx->stream));
    GME_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    return GME_OK;
}

// The device solve on its own (what gme_seq_gme_device_solve runs between its stages), for callers and tests that hold
// normal-equation sums: sums[P][15] = F | Sx | Sy -> params_out[P][6] (first components doubled if `project`,
// motion.py:191-207) and flags_out[P]: bit 1 where a displacement of the h x w model field of those parameters lies within
// 1e-9 of a rounding tie, bit 4 for a singular system.
extern "C" int gme_solve_fit_sums(gme_ctx* ctx, const double* sums, int pairs, int project, int h, int w, double* params_out,
                                  int32_t* flags_out)
{
    GME_ENTER(ctx);
    GME_REQUIRE(sums != nullptr && params_out != nullptr && flags_out != nullptr && pairs >= 0 && h >= 0 && w >= 0, GME_ERR_ARG,
                "gme_solve_fit_sums: bad arguments");
    if (pairs == 0) return GME_OK;
    void* buf = nullptr;
    const size_t b_sums = (size_t)pairs * 15 * sizeof(double), b_par = (size_t)pairs * 6 * sizeof(double), b_fl = (size_t)pairs * sizeof(int32_t);
    int rc = ctx_scratch(ctx, b_sums + b_par + b_fl, &buf);
    if (rc) return rc;
    double* d_sums = (double*)buf;
    double* d_par = (double*)((char*)buf + b_sums);
    int32_t* d_fl = (int32_t*)((char*)buf + b_sums + b_par);
    GME_HIP_TRY(hipMemcpyAsync(d_sums, sums, b_sums, hipMemcpyHostToDevice, ctx->stream));
    GME_HIP_TRY(hipMemsetAsync(d_fl, 0, b_fl, ctx->stream));
    rc = launch_solve3(ctx, d_sums, pairs, project, h, w, d_par, d_fl, 1);
    if (rc) return rc;
    GME_HIP_TRY(hipMemcpyAsync(params_out, d_par, b_par, hipMemcpyDeviceToHost, ctx->stream));
    GME_HIP_TRY(hipMemcpyAsync(flags_out, d_fl, b_fl, hipMemcpyDeviceToHost, ctx->stream));
    return ctx_finish(ctx);
}

// motion.global_motion_estimation + get_motion_field_affine + compensate_frame + the squared error of results.py:50-59,109
// for every pair in ONE call with ONE host round trip: the 3x3 solves of motion.py:262-264,280-282 run on the device
// (k_solve3, gme_kernels.hip).  Opt-in (the package uses it under GME_DEVICE_SOLVE=1): LAPACK's last bits are not
// reproduced, parameters are within rtol 1e-10 of the host path's; flags_out[p] != 0 names the pairs the caller must
// redo through the host path (a model displacement within 1e-9 of a rounding tie: bit 1 at level 2, bit 2 in the final
// field; bit 4: a singular system, numpy.linalg.LinAlgError upstream) -- for all other pairs model fields, masks,
// compensated frames and squared errors are bit-equal to the staged calls'.  Split-phase like them.
extern "C" int gme_seq_gme_device_solve(gme_seq* s, int fd, int bbme_bs, int procedure, int sw, double outlier_fraction,
                                        double* params_out, int64_t* sse_out, int32_t* flags_out)
{
    GME_REQUIRE(s != nullptr && params_out != nullptr && flags_out != nullptr, GME_ERR_ARG, "gme_seq_gme_device_solve: null pointer");
    gme_ctx* ctx = s->ctx;
    GME_ENTER(ctx);
    int rc = gme_begin_common(s, fd, bbme_bs, procedure, sw);
    if (rc) return rc;
    const int pairs = s->gme_pairs;
    const int h = s->H / bbme_bs, w = s->W / bbme_bs;
    GME_REQUIRE(h > 0 && w > 0, GME_ERR_GEOMETRY, "block_size %d does not fit a %d x %d frame", bbme_bs, s->H, s->W);
    rc = ensure_comp(s, fd, pairs);
    if (rc) return rc;
    GME_HIP_TRY(hipMemsetAsync(s->solve_flags, 0, (size_t)(pairs > 0 ? pairs : 1) * sizeof(int32_t), ctx->stream));
    rc = launch_project_first(ctx, s->params0, pairs, s->params_in);
    if (rc) return rc;
    rc = gme_level_bbme(s, 1);
    if (rc) return rc;
    rc = gme_level_bbme(s, 2);                             // independent of the parameters: queued ahead of the level-1 fit
    if (rc) return rc;
    rc = fit_level_launch(s, 1, s->params_in, outlier_fraction, nullptr);
    if (rc) return rc;
    // level-1 solution, projected (motion.py:191-207), used for the level-2 model field: flag bit 1 where that rounds near a tie
    rc = launch_solve3(ctx, s->fit[1].sums, pairs, 1, s->fit[2].h, s->fit[2].w, s->params_in, s->solve_flags, 1);
    if (rc) return rc;
    rc = fit_level_launch(s, 2, s->params_in, outlier_fraction, nullptr);
    if (rc) return rc;
    rc = launch_solve3(ctx, s->fit[2].sums, pairs, 0, h, w, s->comp_params, s->solve_flags, 2);
    if (rc) return rc;
    const Plane& p = s->level[2];
    rc = launch_compensate(ctx, p.at(0), p.stride, pairs, s->H, s->W, p.pitch, nullptr, s->comp_params, h, w, s->comp.ptr,
                           s->comp.stride, s->comp.pitch, p.at(fd), p.stride, s->sse);
    if (rc) return rc;
    rc = copy_small(ctx, params_out, s->comp_params, (size_t)pairs * 6 * sizeof(double), hipMemcpyDeviceToHost, s->split_phase);
    if (rc) return rc;
    if (sse_out) { rc = copy_small(ctx, sse_out, s->sse, (size_t)pairs * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->split_phase); if (rc) return rc; }
    rc = copy_small(ctx, flags_out, s->solve_flags, (size_t)pairs * sizeof(int32_t), hipMemcpyDeviceToHost, s->split_phase);
    if (rc) return rc;
    if (s->split_phase) { GME_HIP_TRY(hipEventRecord(s->ready, ctx->stream)); return GME_OK; }
    return ctx_finish(ctx);
}

extern "C" int gme_seq_read_compensated(gme_seq* s, int pair, uint8_t* out)
{
    GME_REQUIRE(s != nullptr && out != nullptr, GME_ERR_ARG, "gme_seq_read_compensated: null pointer");
    GME_ENTER(s->ctx);
    GME_REQUIRE(s->comp.ptr && pair >= 0 && pair < s->comp.count, GME_ERR_STATE, "gme_seq_read_compensated: no such pair");
    GME_HIP_TRY(hipMemcpy2DAsync(out, s->W, s->comp.at(pair), s->comp.pitch, s->W, s->H, hipMemcpyDeviceToHost, s->ctx->stream));
    return ctx_finish(s->ctx);
}
