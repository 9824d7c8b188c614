// Context, stream and memory management of libvkx.so.
#include "vkx_internal.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void vkx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

VKX_EXPORT const char *vkx_last_error(void) { return g_err; }
VKX_EXPORT int vkx_version(void) { return 1; }

VKX_EXPORT int vkx_device_count(int *count)
{
    VKX_REQUIRE(count != nullptr, "count is NULL");
    VKX_HIP(hipGetDeviceCount(count));
    return VKX_OK;
}

VKX_EXPORT int vkx_device_pci_bus_id(int device, char *buf, int len)
{
    VKX_REQUIRE(buf != nullptr && len >= 16, "buffer of at least 16 bytes");
    VKX_HIP(hipDeviceGetPCIBusId(buf, len, device));
    return VKX_OK;
}

VKX_EXPORT int vkx_ctx_create(int device, vkx_ctx **out)
{
    VKX_REQUIRE(out != nullptr, "out is NULL");
    int n = 0;
    VKX_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        vkx_set_error("vkx_ctx_create: device %d out of range (%d visible)", device, n);
        return VKX_ERR_INVALID;
    }
    VKX_HIP(hipSetDevice(device));
    // VKX_SYNC=block | yield: how a host thread waits for the device (hipDeviceScheduleBlockingSync / hipDeviceScheduleYield instead of the
    // runtime's spin).  A pool of workers sharing one GPU spends most of every page waiting for the device: spinning, 12 workers
    // burn 13 CPUs for 1.5 k pages/s (profiles/r6d_pool_scale_np.json: cpu_ms_per_page 9.1 against 3.7 alone); blocking gives the CPUs
    // back at the price of a wake-up per wait.  Default: the runtime's own policy.
    static const int sync_mode = [] { const char *e = getenv("VKX_SYNC"); return !e ? 0 : (e[0] == 'b' ? 1 : (e[0] == 'y' ? 2 : 0)); }();
    if (sync_mode) {
        if (hipSetDeviceFlags(sync_mode == 1 ? hipDeviceScheduleBlockingSync : hipDeviceScheduleYield) != hipSuccess) (void)hipGetLastError();
    }
    vkx_ctx *ctx = new (std::nothrow) vkx_ctx();
    if (!ctx) { vkx_set_error("out of host memory"); return VKX_ERR_NOMEM; }
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        vkx_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return VKX_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return VKX_OK;
}

static void scratch_release(vkx_scratch *s)
{
    if (s->ptr) (void)hipFree(s->ptr);
    s->ptr = nullptr;
    s->cap = 0;
}

VKX_EXPORT int vkx_ctx_destroy(vkx_ctx *ctx)
{
    if (!ctx) return VKX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &st : ctx->copy_stream)
        if (st) (void)hipStreamSynchronize(st);
    scratch_release(&ctx->owner);
    scratch_release(&ctx->paint_owner);
    scratch_release(&ctx->cells);
    scratch_release(&ctx->misc);
    scratch_release(&ctx->tables);
    for (auto &s : ctx->stage) scratch_release(&s);
    for (auto &s : ctx->chain) scratch_release(&s);
    scratch_release(&ctx->chain_cells);
    scratch_release(&ctx->chain_bins);
    scratch_release(&ctx->chain_misc);
    if (ctx->chain_done) (void)hipEventDestroy(ctx->chain_done);
    if (ctx->chain_setup_done) (void)hipEventDestroy(ctx->chain_setup_done);
    if (ctx->lattices_ready) (void)hipEventDestroy(ctx->lattices_ready);
    scratch_release(&ctx->noise_table);
    scratch_release(&ctx->np_tabs);
    scratch_release(&ctx->noise_rows);
    scratch_release(&ctx->np_work[0]);
    scratch_release(&ctx->np_work[1]);
    scratch_release(&ctx->camera_work);
    scratch_release(&ctx->mls_work);
    scratch_release(&ctx->fog_work);
    scratch_release(&ctx->glass_win);
    scratch_release(&ctx->pz_tabs);
    scratch_release(&ctx->pz_work);
    scratch_release(&ctx->pz_draws);
    for (auto &t : ctx->resize_tabs) scratch_release(&t.buf);
    for (auto &l : ctx->launches) { (void)hipEventDestroy(l.start); (void)hipEventDestroy(l.stop); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto e : ctx->order_events) (void)hipEventDestroy(e);
    for (auto &st : ctx->copy_stream)
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (ctx->desc_ring) (void)hipHostFree(ctx->desc_ring);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return VKX_OK;
}

VKX_EXPORT int vkx_ctx_sync(vkx_ctx *ctx)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    vkx_device_guard guard(ctx);
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}

vkx_device_guard::vkx_device_guard(const vkx_ctx *ctx)
{
    if (!ctx || hipGetDevice(&prev) != hipSuccess || prev == ctx->device) return;
    switched = hipSetDevice(ctx->device) == hipSuccess;
}

vkx_device_guard::~vkx_device_guard()
{
    if (switched) (void)hipSetDevice(prev);
}

VKX_EXPORT int vkx_ctx_set_stream(vkx_ctx *ctx, void *hip_stream)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (hip_stream) {
        // a borrowed stream must live on the ctx's device: its scratch and kernels do
        hipDevice_t dev = -1;
        vkx_device_guard guard(ctx);
        if (hipStreamGetDevice((hipStream_t)hip_stream, &dev) == hipSuccess && (int)dev != ctx->device) {
            vkx_set_error("vkx_ctx_set_stream: the stream belongs to device %d, the ctx to device %d", (int)dev, ctx->device);
            return VKX_ERR_INVALID;
        }
    }
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next != ctx->stream && ctx->desc_off != 0) {
        // slices of the descriptor ring may still be read by work queued on the stream being left; the ring's reuse
        // rules (wrap, reallocation) only ever wait for the CURRENT stream
        vkx_device_guard guard(ctx);
        VKX_HIP(hipStreamSynchronize(ctx->stream));
    }
    ctx->stream = next;
    return VKX_OK;
}

VKX_EXPORT void *vkx_ctx_stream(vkx_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int vkx_scratch_reserve(vkx_ctx *ctx, vkx_scratch *s, size_t bytes)
{
    if (bytes <= s->cap) return VKX_OK;
    vkx_device_guard guard(ctx);
    if (s->ptr) {
        // the old block may still be in use by work queued on the stream -- or on the context's other streams (the side
        // passes of the numpy streams and of the chain setup)
        VKX_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->own_stream && ctx->own_stream != ctx->stream) VKX_HIP(hipStreamSynchronize(ctx->own_stream));
        for (hipStream_t st : ctx->copy_stream)
            if (st && st != ctx->stream) VKX_HIP(hipStreamSynchronize(st));
        VKX_HIP(hipFree(s->ptr));
        s->ptr = nullptr;
        s->cap = 0;
    }
    size_t cap = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&s->ptr, cap);
    if (e != hipSuccess) {
        s->ptr = nullptr;
        vkx_set_error("hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? VKX_ERR_NOMEM : VKX_ERR_HIP;
    }
    s->cap = cap;
    return VKX_OK;
}

VKX_EXPORT int vkx_malloc(vkx_ctx *ctx, size_t bytes, void **dptr)
{
    VKX_REQUIRE(ctx && dptr, "NULL argument");
    vkx_device_guard guard(ctx);
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        vkx_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? VKX_ERR_NOMEM : VKX_ERR_HIP;
    }
    return VKX_OK;
}

VKX_EXPORT int vkx_free(vkx_ctx *ctx, void *dptr)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (!dptr) return VKX_OK;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    VKX_HIP(hipFree(dptr));
    return VKX_OK;
}

VKX_EXPORT int vkx_upload(vkx_ctx *ctx, void *dptr, const void *hptr, size_t bytes)
{
    VKX_REQUIRE(ctx && (bytes == 0 || (dptr && hptr)), "NULL argument");
    if (!bytes) return VKX_OK;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}

VKX_EXPORT int vkx_download(vkx_ctx *ctx, void *hptr, const void *dptr, size_t bytes)
{
    VKX_REQUIRE(ctx && (bytes == 0 || (dptr && hptr)), "NULL argument");
    if (!bytes) return VKX_OK;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}

VKX_EXPORT int vkx_memset(vkx_ctx *ctx, void *dptr, int value, size_t bytes)
{
    VKX_REQUIRE(ctx && (bytes == 0 || dptr), "NULL argument");
    if (!bytes) return VKX_OK;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return VKX_OK;
}

// ---- pinned host memory, copy streams, ordering --------------------------------------------------------------------
// Everything queued on ANY stream of the context has run: the ring's readers are not only on the compute stream (k_chain_prologue
// reads the mapped ring on the side stream at execution time; vkx_camera_states_dev / vkx_mls_states_dev copy from it on a
// stream of the caller's choice).
static int ring_quiesce(vkx_ctx *ctx)
{
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->own_stream != ctx->stream) VKX_HIP(hipStreamSynchronize(ctx->own_stream));
    for (hipStream_t s : ctx->copy_stream)
        if (s && s != ctx->stream) VKX_HIP(hipStreamSynchronize(s));
    return VKX_OK;
}

int vkx_desc_ring_take(vkx_ctx *ctx, size_t bytes, void **hptr)
{
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > ctx->desc_cap) {
        // (re)allocate: nothing queued may still read the old ring
        vkx_device_guard guard(ctx);
        int qrc = ring_quiesce(ctx);
        if (qrc) return qrc;
        if (ctx->desc_ring) VKX_HIP(hipHostFree(ctx->desc_ring));
        ctx->desc_ring = nullptr;
        const size_t cap = bytes * 4 > ((size_t)4 << 20) ? bytes * 4 : ((size_t)4 << 20);
        VKX_HIP(hipHostMalloc((void **)&ctx->desc_ring, cap, hipHostMallocDefault));
        ctx->desc_cap = cap;
        ctx->desc_off = 0;
    }
    if (ctx->desc_off + bytes > ctx->desc_cap) {
        // wrap around: the copies queued from the ring so far must have been issued to the device
        vkx_device_guard guard(ctx);
        int qrc = ring_quiesce(ctx);
        if (qrc) return qrc;
        ctx->desc_off = 0;
    }
    *hptr = ctx->desc_ring + ctx->desc_off;
    ctx->desc_off += bytes;
    return VKX_OK;
}

// The device-side address of a block of the page-locked ring (the ring is mapped: a kernel reads it in place, over the link).
// For tables a kernel reads ONCE -- a 768-byte LUT staged to LDS per workgroup, an edge table, a layer plane that every pixel touches
// once -- reading the ring in place costs the link what the copy would have cost it, and saves the copy's dispatch: with several
// worker processes on one GPU every dispatch costs ~16 us of the device's serial budget whatever its size (profiles/r6b_pool_trace_w8.json),
// and a C4 page issued 44 runtime copy kernels of 98 dispatches (profiles/r6a_page_dispatches.txt).  nullptr when the mapping fails.
const void *vkx_ring_device_ptr(const void *ring_host)
{
    void *mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, const_cast<void *>(ring_host), 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return mapped;
}

namespace {
// (one word per lane: a lane's read of page-locked host memory is a round trip over the link -- a single workgroup looping over
// 35 KB of job records took 60 us)
__global__ void k_small_copy(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, unsigned n_words)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] = src[i];
}
}  // namespace

int vkx_small_to_device(vkx_ctx *ctx, void *dev, const void *ring_host, size_t bytes)
{
    void *mapped = nullptr;
    if (bytes % 4 != 0 || bytes > ((size_t)1 << 20) || hipHostGetDevicePointer(&mapped, const_cast<void *>(ring_host), 0) != hipSuccess) {
        (void)hipGetLastError();
        VKX_HIP(hipMemcpyAsync(dev, ring_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return VKX_OK;
    }
    k_small_copy<<<vkx_blocks(bytes / 4, 256), 256, 0, ctx->stream>>>((uint32_t *)dev, (const uint32_t *)mapped, (unsigned)(bytes / 4));
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

int vkx_small_to_host(vkx_ctx *ctx, void *host, const void *dev, size_t bytes)
{
    void *mapped = nullptr;
    if (bytes % 4 != 0 || bytes > ((size_t)1 << 20) || hipHostGetDevicePointer(&mapped, host, 0) != hipSuccess) {
        (void)hipGetLastError();
        VKX_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return VKX_OK;
    }
    k_small_copy<<<vkx_blocks(bytes / 4, 256), 256, 0, ctx->stream>>>((uint32_t *)mapped, (const uint32_t *)dev, (unsigned)(bytes / 4));
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

hipStream_t vkx_stream_by_id(vkx_ctx *ctx, int id, int *rc)
{
    *rc = VKX_OK;
    if (id == VKX_STREAM_COMPUTE) return ctx->stream;
    if (id != VKX_STREAM_COPY_IN && id != VKX_STREAM_COPY_OUT) {
        vkx_set_error("stream id %d is not one of VKX_STREAM_*", id);
        *rc = VKX_ERR_INVALID;
        return nullptr;
    }
    hipStream_t &st = ctx->copy_stream[id - VKX_STREAM_COPY_IN];
    if (!st) {
        vkx_device_guard guard(ctx);
        // The side streams carry copies and the microsecond kernels that run UNDER a large kernel of the compute stream (carry
        // resolution of the numpy streams, cell setup of the chain).  VKX_SIDE_PRIORITY=1 creates them with the highest priority.
        // Measured (tools/probes/alloc_order.py): no gain for the first context of a process -- the dispatcher places the few
        // workgroups of a side kernel as CU slots free up either way (C3 step 16.54 vs 16.57 ms) -- and a SECOND context with
        // priority streams runs its large kernels 12 - 15 % slower (k_chain_fused 4.06 vs 3.53 ms per 96 pages; the process has
        // more priority streams than the device has priority queues).  Off by default.
        static const bool prio = [] { const char *e = getenv("VKX_SIDE_PRIORITY"); return e && e[0] == '1'; }();
        int least = 0, greatest = 0;
        hipError_t e = hipErrorUnknown;
        if (prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        }
        if (e != hipSuccess) {
            vkx_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            *rc = VKX_ERR_HIP;
            st = nullptr;
        }
    }
    return st;
}

VKX_EXPORT int vkx_host_alloc(vkx_ctx *ctx, size_t bytes, void **hptr)
{
    VKX_REQUIRE(ctx && hptr, "NULL argument");
    vkx_device_guard guard(ctx);
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        vkx_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? VKX_ERR_NOMEM : VKX_ERR_HIP;
    }
    return VKX_OK;
}

VKX_EXPORT int vkx_host_free(vkx_ctx *ctx, void *hptr)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (!hptr) return VKX_OK;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipHostFree(hptr));
    return VKX_OK;
}

VKX_EXPORT int vkx_upload_async(vkx_ctx *ctx, void *dptr, const void *hptr, size_t bytes)
{
    VKX_REQUIRE(ctx && (bytes == 0 || (dptr && hptr)), "NULL argument");
    if (!bytes) return VKX_OK;
    int rc;
    hipStream_t st = vkx_stream_by_id(ctx, VKX_STREAM_COPY_IN, &rc);
    if (rc) return rc;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, st));
    return VKX_OK;
}

VKX_EXPORT int vkx_download_async(vkx_ctx *ctx, void *hptr, const void *dptr, size_t bytes)
{
    VKX_REQUIRE(ctx && (bytes == 0 || (dptr && hptr)), "NULL argument");
    if (!bytes) return VKX_OK;
    int rc;
    hipStream_t st = vkx_stream_by_id(ctx, VKX_STREAM_COPY_OUT, &rc);
    if (rc) return rc;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, st));
    return VKX_OK;
}

VKX_EXPORT int vkx_memcpy_async(vkx_ctx *ctx, int stream, void *dst, const void *src, size_t bytes, int to_device)
{
    VKX_REQUIRE(ctx && (bytes == 0 || (dst && src)), "NULL argument");
    if (!bytes) return VKX_OK;
    int rc;
    hipStream_t st = vkx_stream_by_id(ctx, stream, &rc);
    if (rc) return rc;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipMemcpyAsync(dst, src, bytes,
                           to_device == 2 ? hipMemcpyDeviceToDevice : (to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost), st));
    return VKX_OK;
}

static hipEvent_t take_order_event(vkx_ctx *ctx)
{
    if (!ctx->order_events.empty()) {
        hipEvent_t e = ctx->order_events.back();
        ctx->order_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}

VKX_EXPORT int vkx_ctx_order(vkx_ctx *ctx, int later_stream, int earlier_stream)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    int rc;
    hipStream_t later = vkx_stream_by_id(ctx, later_stream, &rc);
    if (rc) return rc;
    hipStream_t earlier = vkx_stream_by_id(ctx, earlier_stream, &rc);
    if (rc) return rc;
    if (later == earlier) return VKX_OK;
    vkx_device_guard guard(ctx);
    hipEvent_t e = take_order_event(ctx);
    VKX_REQUIRE(e != nullptr, "hipEventCreate failed");
    VKX_HIP(hipEventRecord(e, earlier));
    VKX_HIP(hipStreamWaitEvent(later, e, 0));
    ctx->order_events.push_back(e);      // a recorded event may be re-recorded once the wait has been queued
    return VKX_OK;
}

// `later` continues after everything queued on `earlier` so far (internal form of vkx_ctx_order on raw streams)
int vkx_stream_order(vkx_ctx *ctx, hipStream_t later, hipStream_t earlier)
{
    if (later == earlier) return VKX_OK;
    hipEvent_t e = take_order_event(ctx);
    VKX_REQUIRE(e != nullptr, "hipEventCreate failed");
    VKX_HIP(hipEventRecord(e, earlier));
    VKX_HIP(hipStreamWaitEvent(later, e, 0));
    ctx->order_events.push_back(e);
    return VKX_OK;
}

// Error exits of the calls that spread their work over the context's side streams: `main_stream` continues after everything
// queued on the side streams so far and is the context's launch stream again, so that the next call on the context cannot race
// the kernels the failed call left in flight.  Never touches the message of the failure being reported (no vkx_set_error here).
void vkx_ctx_join_streams(vkx_ctx *ctx, hipStream_t main_stream)
{
    vkx_device_guard guard(ctx);
    ctx->stream = main_stream;
    for (hipStream_t s : ctx->copy_stream) {
        if (!s || s == main_stream) continue;
        hipEvent_t e = take_order_event(ctx);
        if (!e || hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(main_stream, e, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(s);       // no event to order them with: wait on the host
        }
        if (e) ctx->order_events.push_back(e);
    }
}

// The staged paths run on the compute stream: a pending lattices-ready mark (lattices built on another stream) is consumed
// there, so that it neither goes unheeded nor outlives the lattices it spoke of.
int vkx_chain_consume_lattices_mark(vkx_ctx *ctx)
{
    if (!ctx->lattices_armed) return VKX_OK;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipStreamWaitEvent(ctx->stream, ctx->lattices_ready, 0));
    ctx->lattices_armed = false;
    return VKX_OK;
}

VKX_EXPORT int vkx_chain_lattices_ready(vkx_ctx *ctx)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    vkx_device_guard guard(ctx);
    if (!ctx->lattices_ready) VKX_HIP(hipEventCreateWithFlags(&ctx->lattices_ready, hipEventDisableTiming));
    VKX_HIP(hipEventRecord(ctx->lattices_ready, ctx->stream));
    ctx->lattices_armed = true;
    return VKX_OK;
}

VKX_EXPORT int vkx_ctx_sync_stream(vkx_ctx *ctx, int stream)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    int rc;
    hipStream_t st = vkx_stream_by_id(ctx, stream, &rc);
    if (rc) return rc;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipStreamSynchronize(st));
    return VKX_OK;
}

VKX_EXPORT int vkx_event_record(vkx_ctx *ctx, int stream, void **event)
{
    VKX_REQUIRE(ctx && event, "NULL argument");
    int rc;
    hipStream_t st = vkx_stream_by_id(ctx, stream, &rc);
    if (rc) return rc;
    vkx_device_guard guard(ctx);
    hipEvent_t e = nullptr;
    VKX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t err = hipEventRecord(e, st);
    if (err != hipSuccess) {
        (void)hipEventDestroy(e);
        vkx_set_error("hipEventRecord failed: %s", hipGetErrorString(err));
        return VKX_ERR_HIP;
    }
    *event = (void *)e;
    return VKX_OK;
}

VKX_EXPORT int vkx_event_wait(vkx_ctx *ctx, void *event)
{
    VKX_REQUIRE(ctx && event, "NULL argument");
    vkx_device_guard guard(ctx);
    VKX_HIP(hipEventSynchronize((hipEvent_t)event));
    VKX_HIP(hipEventDestroy((hipEvent_t)event));
    return VKX_OK;
}

// ---- per-kernel timing -------------------------------------------------------------------------------------
static hipEvent_t take_event(vkx_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

vkx_timed::vkx_timed(vkx_ctx *c, const char *kernel_name, bool major) : guard(c), ctx(c), slot(-1)
{
    if (!ctx || !ctx->timing || (ctx->timing == 2 && !major)) return;
    int id = -1;
    for (size_t i = 0; i < ctx->timing_names.size(); i++)
        if (ctx->timing_names[i] == kernel_name) { id = (int)i; break; }
    if (id < 0) {
        ctx->timing_names.emplace_back(kernel_name);
        ctx->timing_ms.push_back(0.0);
        ctx->timing_count.push_back(0);
        id = (int)ctx->timing_names.size() - 1;
    }
    vkx_ctx::TimedLaunch l{id, take_event(ctx), take_event(ctx)};
    if (!l.start || !l.stop) return;
    (void)hipEventRecord(l.start, ctx->stream);
    ctx->launches.push_back(l);
    slot = (int)ctx->launches.size() - 1;
}

vkx_timed::~vkx_timed()
{
    if (slot >= 0) (void)hipEventRecord(ctx->launches[slot].stop, ctx->stream);
}

VKX_EXPORT int vkx_ctx_set_timing(vkx_ctx *ctx, int enabled)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->timing = enabled == 2 ? 2 : (enabled != 0);
    return VKX_OK;
}

// Synchronises the stream, folds every recorded launch into the per-kernel totals and returns their number.
VKX_EXPORT int vkx_ctx_collect_timings(vkx_ctx *ctx, int *n_kernels)
{
    VKX_REQUIRE(ctx && n_kernels, "NULL argument");
    vkx_device_guard guard(ctx);
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &l : ctx->launches) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, l.start, l.stop) == hipSuccess) {
            ctx->timing_ms[l.name_id] += ms;
            ctx->timing_count[l.name_id] += 1;
        }
        ctx->event_pool.push_back(l.start);
        ctx->event_pool.push_back(l.stop);
    }
    ctx->launches.clear();
    *n_kernels = (int)ctx->timing_names.size();
    return VKX_OK;
}

VKX_EXPORT int vkx_ctx_get_timing(vkx_ctx *ctx, int index, const char **name, double *total_ms, long long *launches)
{
    VKX_REQUIRE(ctx && name && total_ms && launches, "NULL argument");
    VKX_REQUIRE(index >= 0 && index < (int)ctx->timing_names.size(), "index out of range");
    *name = ctx->timing_names[index].c_str();
    *total_ms = ctx->timing_ms[index];
    *launches = ctx->timing_count[index];
    return VKX_OK;
}

VKX_EXPORT int vkx_ctx_reset_timings(vkx_ctx *ctx)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    int n = 0;
    int rc = vkx_ctx_collect_timings(ctx, &n);
    if (rc) return rc;
    ctx->timing_names.clear();
    ctx->timing_ms.clear();
    ctx->timing_count.clear();
    return VKX_OK;
}
