// Internal definitions shared by the translation units of libvkx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <limits.h>
#include <string>
#include <vector>

#include "../../include/vkx.h"

#define VKX_EXPORT extern "C" __attribute__((visibility("default")))

void vkx_set_error(const char *fmt, ...);

#define VKX_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            vkx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                       \
            return VKX_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define VKX_REQUIRE(cond, msg)                                     \
    do {                                                           \
        if (!(cond)) {                                             \
            vkx_set_error("%s: requirement failed: %s", __func__, msg); \
            return VKX_ERR_INVALID;                                \
        }                                                          \
    } while (0)

#define VKX_LAUNCH_CHECK()                                                   \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            vkx_set_error("kernel launch failed in %s: %s", __func__, hipGetErrorString(e__)); \
            return VKX_ERR_HIP;                                              \
        }                                                                    \
    } while (0)

// A grow-only device scratch slot (owner maps, cell tables, staging for the host entry points).
struct vkx_scratch {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct vkx_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    vkx_scratch owner;    // int32 [dh, dw]
    vkx_scratch paint_owner;          // the ownership raster of vkx_paint_polys*_dev: all zero between calls (the resolve kernel clears what it reads)
    size_t paint_owner_zeroed = 0;    // ... bytes of it known to be zero
    vkx_scratch cells;    // CellRec [n_cells]
    vkx_scratch misc;     // small parameter blocks (layers, element descriptors)
    vkx_scratch tables;   // constant lookup tables (HSV division LUTs), uploaded once
    bool tables_ready = false;
    vkx_scratch stage[6]; // staging planes of the host-pointer entry points
    vkx_scratch chain[3]; // ping-pong planes of the batched chain entry point; [2]: a tile buffer expanded to its int16 plane
    vkx_scratch chain_cells, chain_bins, chain_misc;   // cell records / tile bins / descriptors of the fused chain: its setup kernels
                                                       // run on the side stream while the shared slots above serve the compute stream
    hipEvent_t chain_setup_done = nullptr; // the side stream past the setup kernels of the current chain call
    hipEvent_t chain_done = nullptr;       // the pixel kernel of the last chain call (reads the three slots above)
    hipEvent_t lattices_ready = nullptr;   // vkx_chain_lattices_ready: the point of a stream the lattices are complete at
    bool lattices_armed = false;           // ... for the NEXT chain call only (a stale mark must not outlive the lattices it spoke of)
    vkx_scratch noise_table;          // int16 [65536] inverse-CDF table of vkx_noise_normal_i16 for noise_table_std
    double noise_table_std = 0.0;
    bool noise_table_fits8 = false;
    vkx_scratch np_tabs;              // jump constants + ziggurat tables of the numpy streams (nprand.hip), uploaded once
    vkx_scratch noise_rows;           // tiled noise of the fused chain: (row, tile column) -> slot offset records (fused.hip)
    vkx_scratch np_work[2];           // tile arrays of the numpy streams: the chunks of a call alternate (nprand.hip)
    vkx_scratch mls_work;                     // batched similarity_mls states (mls.hip): descriptors, handle tables, projected positions
    vkx_scratch camera_work;                  // camera states (camera.hip): descriptors, results, the depth values of the cubic curve
    vkx_scratch fog_work;                     // fog field (fog.hip): raw draws + the float64 centres of a level; a glass round's temporaries
    vkx_scratch glass_win;                    // glass shuffle: the winner plane of a round's scatter (uint64 [h, w], zero between rounds)
    vkx_scratch pz_tabs, pz_work, pz_draws;   // rng.poisson on the device (poisson.hip): per-lam constants; block plan; raw draws + E rows
    bool pz_tabs_ready = false;

    // Host-array pipelines: two copy streams next to the compute stream (created on first use), a pool of events that
    // order them, and a page-locked ring through which the launch descriptors of the tile kernels reach the device
    // without a stream synchronisation (a launch returns while its descriptors are still in flight).
    hipStream_t copy_stream[2] = {nullptr, nullptr};   // [0] host -> device, [1] device -> host
    std::vector<hipEvent_t> order_events;              // reusable events of vkx_ctx_order / vkx_event_record
    unsigned char *desc_ring = nullptr;
    size_t desc_cap = 0, desc_off = 0;

    // The tap tables of the last few CUBIC / LANCZOS4 resize geometries (PageResizingStep resizes seven elements with one
    // geometry: the tables are built and uploaded for the first one only).
    struct ResizeTabs {
        int key[6] = {-1, -1, -1, -1, -1, -1};    // taps, fixed point?, sh, sw, dh, dw
        vkx_scratch buf;
        size_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // the tables inside buf (xofs, xcoef, yofs, ycoef for the tap kernels)
        std::vector<int> yofs;                    // host side: row offsets (tile planning) or other small metadata
        unsigned long stamp = 0;
    };
    ResizeTabs resize_tabs[6];
    unsigned long resize_clock = 0;

    // Optional per-kernel timing with HIP events recorded on the launch stream (vkx_ctx_set_timing).
    int timing = 0;                           // 0 off, 1 every kernel, 2 only the large kernels (VKX_TIMED_MAJOR)
    struct TimedLaunch { int name_id; hipEvent_t start, stop; };
    std::vector<TimedLaunch> launches;       // recorded, not yet folded into the totals
    std::vector<hipEvent_t> event_pool;      // reusable events
    std::vector<std::string> timing_names;
    std::vector<double> timing_ms;
    std::vector<long long> timing_count;
};

// The current device is per-thread state: a ctx used from a thread other than its creator (or after the caller
// switched devices, e.g. torch.cuda.set_device) must see its own device while it launches on ctx->stream.
// Saves, sets and restores; free when the device is already current.
struct vkx_device_guard {
    int prev = -1;
    bool switched = false;
    explicit vkx_device_guard(const vkx_ctx *ctx);
    ~vkx_device_guard();
    vkx_device_guard(const vkx_device_guard &) = delete;
    vkx_device_guard &operator=(const vkx_device_guard &) = delete;
};

// RAII scope around ONE kernel launch: binds the ctx device for the launch and records a start / stop event pair on
// the ctx stream when timing is on.
struct vkx_timed {
    vkx_device_guard guard;
    vkx_ctx *ctx;
    int slot;
    vkx_timed(vkx_ctx *ctx, const char *kernel_name, bool major = false);
    ~vkx_timed();
};
#define VKX_TIMED(ctx, name) vkx_timed timed_scope__(ctx, name)
#define VKX_TIMED_MAJOR(ctx, name) vkx_timed timed_scope__(ctx, name, true)

int vkx_scratch_reserve(vkx_ctx *ctx, vkx_scratch *s, size_t bytes);
// `bytes` of page-locked host memory that stays untouched until everything queued on ctx->stream so far has run
// (ring allocation; wraps around with one stream synchronisation)
int vkx_desc_ring_take(vkx_ctx *ctx, size_t bytes, void **hptr);
// Small records between page-locked host memory and the device ON the compute stream, by a one-workgroup kernel that
// reads / writes the host memory through its device mapping: a hipMemcpyAsync of a few hundred bytes queues on the copy
// engines behind the multi-megabyte plane transfers of the other lanes of a host pipeline and stalls the kernels after
// it for the length of those transfers.  `bytes` a multiple of 4.  to_host falls back to hipMemcpyAsync when `host` is
// not mapped page-locked memory.
int vkx_small_to_device(vkx_ctx *ctx, void *dev, const void *ring_host, size_t bytes);
int vkx_small_to_host(vkx_ctx *ctx, void *host, const void *dev, size_t bytes);
hipStream_t vkx_stream_by_id(vkx_ctx *ctx, int id, int *rc);
const void *vkx_ring_device_ptr(const void *ring_host);              // a ring block as kernels address it (mapped host memory), or nullptr
int vkx_stream_order(vkx_ctx *ctx, hipStream_t later, hipStream_t earlier);
void vkx_ctx_join_streams(vkx_ctx *ctx, hipStream_t main_stream);   // error exits of multi-stream calls: main after the side streams, ctx->stream = main
int vkx_chain_consume_lattices_mark(vkx_ctx *ctx);                   // staged chain paths: the compute stream waits for a pending lattices-ready mark
// out[i] = next_double of the PCG64 stream (state, inc) at its (i + 1)-th step (poisson.hip); asynchronous on the ctx stream
int vkx_pcg64_doubles_dev(vkx_ctx *ctx, const uint64_t *state, const uint64_t *inc, long long M, double *out);

// Plane copies between host and device staging: hipMemcpy2DAsync is an order of magnitude slower than a linear copy on
// this stack (12 ms instead of 1 ms for a 2048^2 RGB plane), so planes whose rows follow each other without gaps -- every
// numpy array the binding passes -- travel as ONE linear copy; only genuinely pitched planes take the 2D call.
static inline hipError_t vkx_copy_plane(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t row_bytes,
                                        size_t rows, hipMemcpyKind kind, hipStream_t stream)
{
    if (rows == 0 || row_bytes == 0) return hipSuccess;
    if (rows == 1 || (dst_pitch == row_bytes && src_pitch == row_bytes))
        return hipMemcpyAsync(dst, src, row_bytes * rows, kind, stream);
    return hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, row_bytes, rows, kind, stream);
}

static inline unsigned vkx_blocks(size_t n, unsigned per_block)
{
    size_t b = (n + per_block - 1) / per_block;
    return (unsigned)(b < 1 ? 1 : b);
}

// ---------------------------------------------------------------------------------------------
// Device helpers.  Everything here must agree bit-for-bit with oracle/vkx_oracle.c; the library
// is compiled with -ffp-contract=off so a*b+c is never fused unless written as fma().
// ---------------------------------------------------------------------------------------------
namespace vkd {

// OpenCV cvRound on x86: round-half-even, "integer indefinite" outside int32 / NaN.
__device__ __forceinline__ int cv_round(float v)
{
    if (!(fabsf(v) < 2147483648.f)) return INT_MIN;   // -2^31 itself converts to INT_MIN either way
    return __float2int_rn(v);
}
__device__ __forceinline__ int cv_round(double v)
{
    if (!(v >= -2147483648.5 && v < 2147483647.5)) return INT_MIN;
    return __double2int_rn(v);
}
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int clamp_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// One destination pixel of cv::remapBilinear on uint8, BORDER_CONSTANT 0.  X, Y are the source
// coordinate in 1/32 px.  The int16 weight table of OpenCV is (32-fy)(32-fx)*32 etc. (with the
// {32767,0,0,1} quirk at fy=fx=0); (sum*32 + 2^14) >> 15 == (sum + 512) >> 10 for every entry,
// the quirk entry included, so the table never needs to be materialised.
// PTR: `const uint8_t *` or the same in an explicit address space (a kernel that holds its planes as global-address-space
// pointers keeps FLAT instructions out of its code that way).
template <int CN, typename PTR = const uint8_t *>
__device__ __forceinline__ void sample_u8(PTR src, int sh, int sw, ptrdiff_t sstride, int X, int Y, uint8_t *out)
{
    const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
    const int fx = X & 31, fy = Y & 31;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
#pragma unroll
        for (int k = 0; k < CN; k++) out[k] = 0;
        return;
    }
    const bool x0 = sx >= 0, x1 = sx + 1 < sw, y0 = sy >= 0, y1 = sy + 1 < sh;
    const PTR r0 = src + (ptrdiff_t)sy * sstride + (ptrdiff_t)sx * CN;
    const PTR r1 = r0 + sstride;
    const int w00 = (32 - fy) * (32 - fx), w01 = (32 - fy) * fx, w10 = fy * (32 - fx), w11 = fy * fx;
#pragma unroll
    for (int k = 0; k < CN; k++) {
        const int v0 = (x0 && y0) ? r0[k] : 0;
        const int v1 = (x1 && y0) ? r0[CN + k] : 0;
        const int v2 = (x0 && y1) ? r1[k] : 0;
        const int v3 = (x1 && y1) ? r1[CN + k] : 0;
        out[k] = (uint8_t)((v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + 512) >> 10);
    }
}

template <typename PTR = const float *>
__device__ __forceinline__ float sample_f32(PTR src, int sh, int sw, ptrdiff_t sstride_el, int X, int Y)
{
    const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
    const int fx = X & 31, fy = Y & 31;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return 0.f;
    const bool x0 = sx >= 0, x1 = sx + 1 < sw, y0 = sy >= 0, y1 = sy + 1 < sh;
    const PTR r0 = src + (ptrdiff_t)sy * sstride_el + sx;
    const PTR r1 = r0 + sstride_el;
    const float ax = fx * (1.f / 32), ay = fy * (1.f / 32);
    const float bx = 1.f - ax, by = 1.f - ay;
    const float w0 = by * bx, w1 = by * ax, w2 = ay * bx, w3 = ay * ax; // exact products (5 bit x 5 bit)
    const float v0 = (x0 && y0) ? r0[0] : 0.f;
    const float v1 = (x1 && y0) ? r0[1] : 0.f;
    const float v2 = (x0 && y1) ? r1[0] : 0.f;
    const float v3 = (x1 && y1) ? r1[1] : 0.f;
    const float p0 = v0 * w0, p1 = v1 * w1, p2 = v2 * w2, p3 = v3 * w3;
    return ((p0 + p1) + p2) + p3;
}

} // namespace vkd

// cross-translation-unit helpers
int vkx_hsv_tables(vkx_ctx *ctx, const void **out);                      // photo.hip
int vkx_gaussian_kernel_q8_host(int n, double sigma, uint16_t *kq);      // photo.hip
int vkx_chain_fused_try(vkx_ctx *ctx, const vkx_chain_item *items, int n_items);  // fused.hip
// the steps of the fused chain, for callers that queue them on streams of their choice (chain.hip); fused.hip
struct vkx_chain_plan;
int vkx_chain_plan_build(vkx_ctx *ctx, const vkx_chain_item *items, int n_items, vkx_chain_plan **out);   // VKX_ERR_UNSUPPORTED: not for the fused path
void vkx_chain_plan_free(vkx_chain_plan *p);
int vkx_chain_plan_setup_aside(vkx_ctx *ctx, vkx_chain_plan *p, hipEvent_t *done);      // side stream; *done (may be NULL: ran on ctx->stream): what the pixel kernels wait for
int vkx_chain_plan_noise_rows(vkx_ctx *ctx, vkx_chain_plan *p, int first, int count);   // on ctx->stream
int vkx_chain_plan_tiles(vkx_ctx *ctx, vkx_chain_plan *p, int first, int count);        // on ctx->stream
int vkx_chain_mark_done(vkx_ctx *ctx, hipStream_t stream);
// one chunk of VKX_NP_NORMAL_TILES jobs (nprand.hip): begin = tile states + draw pass, finish = carries, walks, tables; each on
// ctx->stream as it is when called.  `slot`: which of the two tile-array scratch slots the chunk uses.
struct vkx_np_chunk;
int vkx_np_jobs_check(const vkx_np_job *jobs, int n_jobs, const vkx_np_result *results_host);
int vkx_np_chunk_begin(vkx_ctx *ctx, const vkx_np_job *jobs, int n_jobs, vkx_np_result *results_host, int slot, vkx_np_chunk **out);
int vkx_np_chunk_finish(vkx_ctx *ctx, vkx_np_chunk *c);
void vkx_np_chunk_free(vkx_np_chunk *c);
int vkx_tile_remap_try(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw, const int32_t *src_vertices,
                       const int32_t *dst_vertices, int rows, int cols, int dh, int dw);  // fused.hip
// the tile buffer of a VKX_NP_NORMAL_TILES job of n samples (nprand.hip; read by the fused chain kernel)
struct vkx_np_tiles_shape {
    long long n_tiles;
    int slot_elems;
    size_t table_offset, slots_offset, bytes;
    double samples_per_tile;     // expectation: the first guess of "which tile holds sample i"
};
vkx_np_tiles_shape vkx_np_tiles_shape_of(long long n);
