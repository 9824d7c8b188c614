// Host-pointer entry points: stage the caller's numpy buffers through ctx scratch, run the *_dev
// implementation on the ctx stream, copy the results back and synchronise.  Nothing here computes pixels.
#include "vkx_internal.h"

#include <stdlib.h>
#include <string.h>

namespace {

// Collects the planes of one call, packs them into a single device allocation (stage[0]) and moves them with
// hipMemcpy2DAsync so arbitrary host row pitches are honoured.
class HostStage {
public:
    explicit HostStage(vkx_ctx *ctx) : ctx_(ctx) {}

    // returns the plane id; device pitch is row_bytes (tightly packed)
    int add(const void *host_in, void *host_out, size_t row_bytes, int rows, ptrdiff_t host_pitch)
    {
        Plane p;
        p.in = host_in; p.out = host_out; p.row_bytes = row_bytes; p.rows = rows; p.pitch = host_pitch;
        p.off = total_;
        total_ += (row_bytes * (size_t)(rows > 0 ? rows : 0) + 255) & ~(size_t)255;
        planes_.push_back(p);
        return (int)planes_.size() - 1;
    }

    // copy_aside: the call is asynchronous (nothing is read back): its staging copy may go to the copy stream (see below)
    int commit(bool copy_aside = false)
    {
        int rc = vkx_scratch_reserve(ctx_, &ctx_->stage[0], total_ ? total_ : 256);
        if (rc) return rc;
        base_ = (uint8_t *)ctx_->stage[0].ptr;
        // The input planes are gathered in the page-locked descriptor ring and travel as ONE copy per run of neighbours (normally one
        // run): a copy from pageable memory is staged by the runtime anyway -- in chunks, each a copy KERNEL on the compute queue (a C4
        // page's two full-page score maps were 9 such dispatches) --, while one copy out of page-locked memory goes to a DMA engine and
        // leaves the compute queue to the kernels (of this process and of the other workers sharing the GPU).  Beyond 48 MB of inputs
        // the planes go directly.
        constexpr size_t kRingMax = (size_t)48 << 20;
        size_t in_total = 0;
        for (auto &p : planes_)
            if (p.in && p.row_bytes && p.rows > 0) in_total += (p.row_bytes * (size_t)p.rows + 255) & ~(size_t)255;
        uint8_t *ring = nullptr;
        if (in_total > 0 && in_total <= kRingMax) {
            void *r = nullptr;
            if ((rc = vkx_desc_ring_take(ctx_, in_total, &r))) return rc;
            ring = (uint8_t *)r;
        }
        size_t ring_off = 0, run_dev = 0, run_ring = 0, run_bytes = 0;
        // A page's worth of planes (>= 256 KB) is copied on the context's host -> device copy stream, ordered after what the compute stream
        // has queued (the staging block may still be read) and before what it queues next.  In line on the compute stream, behind
        // kernels, the runtime executes the copy as a copy KERNEL (8 MB: 0.2 ms of the compute queue per page); on a stream of its own it
        // goes to a DMA engine: kernel time per C4 page 0.74 -> 0.52 ms, eight workers sharing the GPU 1 202 -> 1 563 pages/s
        // (profiles/r6h0_ / r6h1_page_dispatches.txt).  VKX_STAGE_COPY_STREAM=0 keeps it in line.
        static const bool aside = [] { const char *e = getenv("VKX_STAGE_COPY_STREAM"); return !(e && e[0] == '0'); }();
        hipStream_t copy_stream = ctx_->stream;
        // (only for calls that return without reading anything back: behind a synchronous call -- similarity_mls.distort on one 2048^2
        //  image -- the DMA engine's start-up latency is what the caller waits for: 1.5 -> 2.1 ms per call)
        if (aside && copy_aside && ring && in_total >= ((size_t)256 << 10)) {
            int src = VKX_OK;
            hipStream_t cs = vkx_stream_by_id(ctx_, VKX_STREAM_COPY_IN, &src);
            if (src == VKX_OK && cs && vkx_stream_order(ctx_, cs, ctx_->stream) == VKX_OK) copy_stream = cs;
        }
        auto flush = [&]() -> hipError_t {
            if (!run_bytes) return hipSuccess;
            const hipError_t e = hipMemcpyAsync(base_ + run_dev, ring + run_ring, run_bytes, hipMemcpyHostToDevice, copy_stream);
            run_bytes = 0;
            return e;
        };
        for (auto &p : planes_) {
            if (!p.in || p.row_bytes == 0 || p.rows <= 0) continue;
            const size_t bytes = p.row_bytes * (size_t)p.rows, padded = (bytes + 255) & ~(size_t)255;
            if (ring) {
                // device offsets of consecutive planes are contiguous (add() pads to 256 like the ring does)
                if (run_bytes && run_dev + run_bytes != p.off) VKX_HIP(flush());
                if (!run_bytes) { run_dev = p.off; run_ring = ring_off; }
                if ((size_t)p.pitch == p.row_bytes || p.rows == 1) memcpy(ring + ring_off, p.in, bytes);
                else
                    for (int r = 0; r < p.rows; r++) memcpy(ring + ring_off + (size_t)r * p.row_bytes, (const uint8_t *)p.in + (ptrdiff_t)r * p.pitch, p.row_bytes);
                ring_off += padded;
                run_bytes += padded;
                continue;
            }
            VKX_HIP(flush());
            // a contiguous plane (the normal numpy case) is ONE linear copy: the 2-D form moves row by row and runs at a
            // fraction of the link (15 ms instead of 0.5 ms for a 2048^2 RGB page and its result)
            if ((size_t)p.pitch == p.row_bytes || p.rows == 1)
                VKX_HIP(hipMemcpyAsync(base_ + p.off, p.in, bytes, hipMemcpyHostToDevice, ctx_->stream));
            else
                VKX_HIP(hipMemcpy2DAsync(base_ + p.off, p.row_bytes, p.in, (size_t)p.pitch, p.row_bytes, (size_t)p.rows,
                                         hipMemcpyHostToDevice, ctx_->stream));
        }
        VKX_HIP(flush());
        if (copy_stream != ctx_->stream) return vkx_stream_order(ctx_, ctx_->stream, copy_stream);
        return VKX_OK;
    }

    // Input planes that the kernel reads ONCE (the layers of a composite: every pixel of a plane is touched by one lane): gathered in
    // the page-locked ring and read there, in place, over the link -- no copy to device memory at all.  The link carries each byte
    // once either way; what goes is the copy's dispatches (a C4 page staged 13 MB of layer planes with 9 runtime copy kernels) and,
    // for pageable sources, the runtime's own staging pass.  false: too large for the ring or not mappable (use commit()).
    bool commit_mapped()
    {
        if (total_ == 0 || total_ > ((size_t)48 << 20)) return false;
        for (auto &p : planes_)
            if (p.out) return false;             // outputs need device memory + finish()
        void *r = nullptr;
        if (vkx_desc_ring_take(ctx_, total_, &r) != VKX_OK) return false;
        uint8_t *mapped = (uint8_t *)const_cast<void *>(vkx_ring_device_ptr(r));
        if (!mapped) return false;
        uint8_t *ring = (uint8_t *)r;
        for (auto &p : planes_) {
            if (!p.in || p.row_bytes == 0 || p.rows <= 0) continue;
            if ((size_t)p.pitch == p.row_bytes || p.rows == 1) memcpy(ring + p.off, p.in, p.row_bytes * (size_t)p.rows);
            else
                for (int row = 0; row < p.rows; row++) memcpy(ring + p.off + (size_t)row * p.row_bytes, (const uint8_t *)p.in + (ptrdiff_t)row * p.pitch, p.row_bytes);
        }
        base_ = mapped;
        return true;
    }

    template <class T> T *dev(int id) const { return id < 0 ? nullptr : (T *)(base_ + planes_[id].off); }
    size_t total_bytes() const { return total_; }

    int finish()
    {
        for (auto &p : planes_) {
            if (!p.out || p.row_bytes == 0 || p.rows <= 0) continue;
            if ((size_t)p.pitch == p.row_bytes || p.rows == 1)
                VKX_HIP(hipMemcpyAsync(p.out, base_ + p.off, p.row_bytes * (size_t)p.rows, hipMemcpyDeviceToHost, ctx_->stream));
            else
                VKX_HIP(hipMemcpy2DAsync(p.out, (size_t)p.pitch, base_ + p.off, p.row_bytes, p.row_bytes, (size_t)p.rows,
                                         hipMemcpyDeviceToHost, ctx_->stream));
        }
        VKX_HIP(hipStreamSynchronize(ctx_->stream));
        return VKX_OK;
    }

private:
    struct Plane {
        const void *in;
        void *out;
        size_t row_bytes;
        int rows;
        ptrdiff_t pitch;
        size_t off;
    };
    vkx_ctx *ctx_;
    std::vector<Plane> planes_;
    size_t total_ = 0;
    uint8_t *base_ = nullptr;
};

#define VKX_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)

} // namespace

VKX_EXPORT int vkx_remap_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                            const float *map_x, const float *map_y, ptrdiff_t map_stride_el, uint8_t *dst, int dh,
                            int dw, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && map_x && map_y && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh >= 0 && dw >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)sw * cn, sh, src_stride);
    const int mx = st.add(map_x, nullptr, (size_t)dw * 4, dh, map_stride_el * 4);
    const int my = st.add(map_y, nullptr, (size_t)dw * 4, dh, map_stride_el * 4);
    const int d = st.add(nullptr, dst, (size_t)dw * cn, dh, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_remap_u8_dev(ctx, st.dev<uint8_t>(s), sh, sw, cn, (ptrdiff_t)sw * cn, st.dev<float>(mx),
                             st.dev<float>(my), dw, st.dev<uint8_t>(d), dh, dw, (ptrdiff_t)dw * cn));
    return st.finish();
}

VKX_EXPORT int vkx_remap_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                             const float *map_x, const float *map_y, ptrdiff_t map_stride_el, float *dst, int dh, int dw,
                             ptrdiff_t dst_stride_el)
{
    VKX_REQUIRE(ctx && src && map_x && map_y && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh >= 0 && dw >= 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)sw * 4, sh, src_stride_el * 4);
    const int mx = st.add(map_x, nullptr, (size_t)dw * 4, dh, map_stride_el * 4);
    const int my = st.add(map_y, nullptr, (size_t)dw * 4, dh, map_stride_el * 4);
    const int d = st.add(nullptr, dst, (size_t)dw * 4, dh, dst_stride_el * 4);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_remap_f32_dev(ctx, st.dev<float>(s), sh, sw, sw, st.dev<float>(mx), st.dev<float>(my), dw,
                              st.dev<float>(d), dh, dw, dw));
    return st.finish();
}

#define VKX_WARP_HOST(NAME, MLEN)                                                                                   \
    VKX_EXPORT int NAME##_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,          \
                             const double M[MLEN], uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride)                \
    {                                                                                                                 \
        VKX_REQUIRE(ctx && src && M && dst, "NULL argument");                                                         \
        VKX_REQUIRE(sh > 0 && sw > 0 && dh >= 0 && dw >= 0 && cn > 0, "bad shape");                                   \
        HostStage st(ctx);                                                                                            \
        const int s = st.add(src, nullptr, (size_t)sw * cn, sh, src_stride);                                          \
        const int d = st.add(nullptr, dst, (size_t)dw * cn, dh, dst_stride);                                          \
        VKX_TRY(st.commit());                                                                                         \
        VKX_TRY(NAME##_u8_dev(ctx, st.dev<uint8_t>(s), sh, sw, cn, (ptrdiff_t)sw * cn, M, st.dev<uint8_t>(d), dh, dw, \
                              (ptrdiff_t)dw * cn));                                                                   \
        return st.finish();                                                                                           \
    }                                                                                                                 \
    VKX_EXPORT int NAME##_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,                \
                              const double M[MLEN], float *dst, int dh, int dw, ptrdiff_t dst_stride_el)              \
    {                                                                                                                 \
        VKX_REQUIRE(ctx && src && M && dst, "NULL argument");                                                         \
        VKX_REQUIRE(sh > 0 && sw > 0 && dh >= 0 && dw >= 0, "bad shape");                                             \
        HostStage st(ctx);                                                                                            \
        const int s = st.add(src, nullptr, (size_t)sw * 4, sh, src_stride_el * 4);                                    \
        const int d = st.add(nullptr, dst, (size_t)dw * 4, dh, dst_stride_el * 4);                                    \
        VKX_TRY(st.commit());                                                                                         \
        VKX_TRY(NAME##_f32_dev(ctx, st.dev<float>(s), sh, sw, sw, M, st.dev<float>(d), dh, dw, dw));                  \
        return st.finish();                                                                                           \
    }

VKX_WARP_HOST(vkx_warp_affine, 6)
VKX_WARP_HOST(vkx_warp_perspective, 9)

VKX_EXPORT int vkx_grid_to_map(vkx_ctx *ctx, const int32_t *src_vertices, const int32_t *dst_vertices, int rows,
                               int cols, int dh, int dw, float *map_x, float *map_y, ptrdiff_t map_stride_el,
                               int32_t *owner)
{
    VKX_REQUIRE(ctx && src_vertices && dst_vertices && map_x && map_y, "NULL argument");
    VKX_REQUIRE(rows >= 2 && cols >= 2 && dh > 0 && dw > 0, "bad shape");
    HostStage st(ctx);
    const size_t vbytes = (size_t)rows * cols * 2 * sizeof(int32_t);
    const int sv = st.add(src_vertices, nullptr, vbytes, 1, (ptrdiff_t)vbytes);
    const int dv = st.add(dst_vertices, nullptr, vbytes, 1, (ptrdiff_t)vbytes);
    const int mx = st.add(nullptr, map_x, (size_t)dw * 4, dh, map_stride_el * 4);
    const int my = st.add(nullptr, map_y, (size_t)dw * 4, dh, map_stride_el * 4);
    const int ow = owner ? st.add(nullptr, owner, (size_t)dw * 4, dh, (ptrdiff_t)dw * 4) : -1;
    VKX_TRY(st.commit());
    VKX_TRY(vkx_grid_to_map_dev(ctx, st.dev<int32_t>(sv), st.dev<int32_t>(dv), rows, cols, dh, dw, st.dev<float>(mx),
                                st.dev<float>(my), dw, st.dev<int32_t>(ow)));
    return st.finish();
}

VKX_EXPORT int vkx_grid_remap(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw,
                              const int32_t *src_vertices, const int32_t *dst_vertices, int rows, int cols, int dh, int dw)
{
    VKX_REQUIRE(ctx && elems && src_vertices && dst_vertices, "NULL argument");
    VKX_REQUIRE(n_elems >= 1 && n_elems <= 4, "1..4 elements per call");
    VKX_REQUIRE(rows >= 2 && cols >= 2 && dh > 0 && dw > 0 && sh > 0 && sw > 0, "bad shape");
    HostStage st(ctx);
    const size_t vbytes = (size_t)rows * cols * 2 * sizeof(int32_t);
    const int sv = st.add(src_vertices, nullptr, vbytes, 1, (ptrdiff_t)vbytes);
    const int dv = st.add(dst_vertices, nullptr, vbytes, 1, (ptrdiff_t)vbytes);
    int sid[4], did[4];
    for (int i = 0; i < n_elems; i++) {
        const vkx_elem &e = elems[i];
        VKX_REQUIRE(e.src && e.dst && e.cn >= 1 && e.cn <= 4, "bad element");
        const size_t esz = e.is_f32 ? 4 : 1;
        sid[i] = st.add(e.src, nullptr, (size_t)sw * e.cn * esz, sh, e.src_stride * (ptrdiff_t)esz);
        did[i] = st.add(nullptr, e.dst, (size_t)dw * e.cn * esz, dh, e.dst_stride * (ptrdiff_t)esz);
    }
    VKX_TRY(st.commit());
    vkx_elem de[4];
    for (int i = 0; i < n_elems; i++) {
        de[i] = elems[i];
        de[i].src = st.dev<uint8_t>(sid[i]);
        de[i].dst = st.dev<uint8_t>(did[i]);
        de[i].src_stride = (ptrdiff_t)sw * elems[i].cn;
        de[i].dst_stride = (ptrdiff_t)dw * elems[i].cn;
    }
    VKX_TRY(vkx_grid_remap_dev(ctx, de, n_elems, sh, sw, st.dev<int32_t>(sv), st.dev<int32_t>(dv), rows, cols, dh, dw));
    return st.finish();
}

VKX_EXPORT int vkx_remap_multi(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw, const float *map_x,
                               const float *map_y, ptrdiff_t map_stride_el, int dh, int dw)
{
    VKX_REQUIRE(ctx && elems && map_x && map_y, "NULL argument");
    VKX_REQUIRE(n_elems >= 1 && n_elems <= 8, "1..8 elements per call");
    VKX_REQUIRE(dh >= 0 && dw >= 0 && sh > 0 && sw > 0, "bad shape");
    HostStage st(ctx);
    const int mx = st.add(map_x, nullptr, (size_t)dw * 4, dh, map_stride_el * 4);
    const int my = st.add(map_y, nullptr, (size_t)dw * 4, dh, map_stride_el * 4);
    int sid[8], did[8];
    for (int i = 0; i < n_elems; i++) {
        const vkx_elem &e = elems[i];
        VKX_REQUIRE(e.src && e.dst && e.cn >= 1 && e.cn <= 4, "bad element");
        const size_t esz = e.is_f32 ? 4 : 1;
        sid[i] = st.add(e.src, nullptr, (size_t)sw * e.cn * esz, sh, e.src_stride * (ptrdiff_t)esz);
        did[i] = st.add(nullptr, e.dst, (size_t)dw * e.cn * esz, dh, e.dst_stride * (ptrdiff_t)esz);
    }
    VKX_TRY(st.commit());
    vkx_elem de[8];
    for (int i = 0; i < n_elems; i++) {
        de[i] = elems[i];
        de[i].src = st.dev<uint8_t>(sid[i]);
        de[i].dst = st.dev<uint8_t>(did[i]);
        de[i].src_stride = (ptrdiff_t)sw * elems[i].cn;
        de[i].dst_stride = (ptrdiff_t)dw * elems[i].cn;
    }
    VKX_TRY(vkx_remap_multi_dev(ctx, de, n_elems, sh, sw, st.dev<float>(mx), st.dev<float>(my), dw, dh, dw));
    return st.finish();
}

VKX_EXPORT int vkx_gaussian_blur_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                    int ksize, double sigma, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_gaussian_blur_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, ksize, sigma,
                                     st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_color_shift_rgb(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta,
                                   uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, dst, (size_t)w * 3, h, src_stride);
    (void)dst_stride;
    VKX_REQUIRE(src_stride == dst_stride, "host color_shift needs equal source / destination pitch");
    VKX_TRY(st.commit());
    VKX_TRY(vkx_color_shift_rgb_dev(ctx, st.dev<uint8_t>(s), h, w, (ptrdiff_t)w * 3, delta, st.dev<uint8_t>(s),
                                    (ptrdiff_t)w * 3));
    return st.finish();
}

VKX_EXPORT int vkx_cvt_rgb_hsv_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int to_hsv,
                                  uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * 3, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * 3, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_cvt_rgb_hsv_u8_dev(ctx, st.dev<uint8_t>(s), h, w, (ptrdiff_t)w * 3, to_hsv, st.dev<uint8_t>(d),
                                   (ptrdiff_t)w * 3));
    return st.finish();
}

VKX_EXPORT int vkx_mean_shift_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int delta,
                                 int has_threshold, int threshold, int cycle, unsigned channel_mask, uint8_t *dst,
                                 ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_mean_shift_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, delta, has_threshold, threshold,
                                  cycle, channel_mask, st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_add_noise_i16(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                 const int16_t *noise, ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && noise && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int n = st.add(noise, nullptr, (size_t)w * cn * 2, h, noise_stride_el * 2);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_add_noise_i16_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, st.dev<int16_t>(n),
                                  (ptrdiff_t)w * cn, st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_line_streak_u8(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int thickness,
                                  int gap, int dash_thickness, int dash_gap, const uint8_t color[4], double alpha,
                                  int enable_vert, int enable_hori)
{
    VKX_REQUIRE(ctx && img, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(img, img, (size_t)w * cn, h, stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_line_streak_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, thickness, gap, dash_thickness,
                                   dash_gap, color, alpha, enable_vert, enable_hori));
    return st.finish();
}

// Device page, host layer planes (the text lines of a page assembled onto a device-resident image): the planes are staged
// like vkx_fill_u8's, the page neither travels nor is waited for.  Asynchronous on the ctx stream.
VKX_EXPORT int vkx_fill_u8_dev_host_layers(vkx_ctx *ctx, uint8_t *dst_dev, int h, int w, int cn, ptrdiff_t dst_stride,
                                           const vkx_layer *layers, int n_layers)
{
    VKX_REQUIRE(ctx && dst_dev, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    VKX_REQUIRE(n_layers >= 0 && (n_layers == 0 || layers), "bad layer list");
    HostStage st(ctx);
    std::vector<int> mid(n_layers, -1), aid(n_layers, -1), vid(n_layers, -1);
    constexpr int kOnDevice = VKX_LAYER_MASK_ON_DEVICE | VKX_LAYER_ALPHA_ON_DEVICE | VKX_LAYER_VALUE_ON_DEVICE;
    // A layer selected by its alpha plane alone (fill_np_array: np_mask = alpha > 0, element/box.py:329-331) touches nothing in rows whose
    // alpha is <= 0 throughout: the box shrinks to the rows that select something (a page-sized score map with a bounding-box line or a
    // barcode in it -- page_assembler.py:167-177 -- is 4 MB of zeros around a few KB), so neither the CPU staging pass nor the link
    // carries them.  A dense plane costs the scan a few elements at its first and last row.
    std::vector<vkx_layer> cropped(layers, layers + n_layers);
    for (int i = 0; i < n_layers; i++) {
        vkx_layer &l = cropped[i];
        if (!l.alpha || l.mask || (l.mode & VKX_LAYER_ALPHA_ON_DEVICE) || l.height <= 0 || l.width <= 0) continue;
        auto selects = [&](int row) {
            const float *a = l.alpha + (ptrdiff_t)row * l.alpha_stride_el;
            // an all-zero row first, as one OR over its words (no early exit: the compiler vectorises it; a page-sized score map is
            // 4 MB of exactly that), then the exact test for rows that hold anything
            uint32_t any = 0;
            for (int x = 0; x < l.width; x++) {
                uint32_t bits;
                memcpy(&bits, a + x, 4);
                any |= bits;
            }
            if (!any) return false;
            for (int x = 0; x < l.width; x++)
                if (a[x] > 0.0f) return true;
            return false;
        };
        int r0 = 0, r1 = l.height;
        while (r0 < r1 && !selects(r0)) r0++;
        while (r1 > r0 && !selects(r1 - 1)) r1--;
        if (r0 == 0 && r1 == l.height) continue;
        if (r0 == r1) { l.height = 0; continue; }          // selects nothing: dropped by the composite (to_layer_dev)
        l.alpha += (ptrdiff_t)r0 * l.alpha_stride_el;
        if (l.value) l.value += (ptrdiff_t)r0 * l.value_stride;
        l.up += r0;
        l.height = r1 - r0;
    }
    layers = cropped.data();
    for (int i = 0; i < n_layers; i++) {
        const vkx_layer &l = layers[i];
        VKX_REQUIRE(l.height >= 0 && l.width >= 0, "bad layer box");
        if (l.mask && !(l.mode & VKX_LAYER_MASK_ON_DEVICE)) mid[i] = st.add(l.mask, nullptr, (size_t)l.width, l.height, l.mask_stride);
        if (l.alpha && !(l.mode & VKX_LAYER_ALPHA_ON_DEVICE)) aid[i] = st.add(l.alpha, nullptr, (size_t)l.width * 4, l.height, l.alpha_stride_el * 4);
        if (l.value && !(l.mode & VKX_LAYER_VALUE_ON_DEVICE)) vid[i] = st.add(l.value, nullptr, (size_t)l.width * cn, l.height, l.value_stride);
    }
    // Where the kernel finds the planes: a few KB (a mask strip, a small box) are read in place in the mapped ring -- no copy at all;
    // a page's worth (MBs) is copied to device memory by a DMA engine (commit(): one copy out of the ring) and read from HBM: the
    // composite kernel holding CUs for 0.6 ms while 16 MB cross the link was the largest kernel of a page (profiles/r6c_page_dispatches.txt)
    // and, with several workers on the GPU, compute-queue time is what the workers share.  VKX_LAYERS_MAPPED=1 / 0 forces one way.
    static const int map_env = [] { const char *e = getenv("VKX_LAYERS_MAPPED"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    const bool mapped = map_env >= 0 ? map_env != 0 : st.total_bytes() <= ((size_t)64 << 10);
    if (!mapped || !st.commit_mapped()) VKX_TRY(st.commit(true));
    std::vector<vkx_layer> dl(layers, layers + n_layers);
    for (int i = 0; i < n_layers; i++) {
        if (mid[i] >= 0) { dl[i].mask = st.dev<uint8_t>(mid[i]); dl[i].mask_stride = layers[i].width; }
        if (aid[i] >= 0) { dl[i].alpha = st.dev<float>(aid[i]); dl[i].alpha_stride_el = layers[i].width; }
        if (vid[i] >= 0) { dl[i].value = st.dev<uint8_t>(vid[i]); dl[i].value_stride = (ptrdiff_t)layers[i].width * cn; }
        dl[i].mode &= ~kOnDevice;
    }
    return vkx_fill_u8_dev(ctx, dst_dev, h, w, cn, dst_stride, dl.data(), n_layers);
}

VKX_EXPORT int vkx_fill_u8(vkx_ctx *ctx, uint8_t *dst, int h, int w, int cn, ptrdiff_t dst_stride,
                           const vkx_layer *layers, int n_layers)
{
    VKX_REQUIRE(ctx && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    VKX_REQUIRE(n_layers >= 0 && (n_layers == 0 || layers), "bad layer list");
    HostStage st(ctx);
    const int d = st.add(dst, dst, (size_t)w * cn, h, dst_stride);
    std::vector<int> mid(n_layers, -1), aid(n_layers, -1), vid(n_layers, -1);
    for (int i = 0; i < n_layers; i++) {
        const vkx_layer &l = layers[i];
        VKX_REQUIRE(l.height >= 0 && l.width >= 0, "bad layer box");
        if (l.mask) mid[i] = st.add(l.mask, nullptr, (size_t)l.width, l.height, l.mask_stride);
        if (l.alpha) aid[i] = st.add(l.alpha, nullptr, (size_t)l.width * 4, l.height, l.alpha_stride_el * 4);
        if (l.value) vid[i] = st.add(l.value, nullptr, (size_t)l.width * cn, l.height, l.value_stride);
    }
    VKX_TRY(st.commit());
    std::vector<vkx_layer> dl(layers, layers + n_layers);
    for (int i = 0; i < n_layers; i++) {
        dl[i].mask = st.dev<uint8_t>(mid[i]);
        dl[i].mask_stride = layers[i].width;
        dl[i].alpha = st.dev<float>(aid[i]);
        dl[i].alpha_stride_el = layers[i].width;
        dl[i].value = st.dev<uint8_t>(vid[i]);
        dl[i].value_stride = (ptrdiff_t)layers[i].width * cn;
    }
    VKX_TRY(vkx_fill_u8_dev(ctx, st.dev<uint8_t>(d), h, w, cn, (ptrdiff_t)w * cn, dl.data(), n_layers));
    return st.finish();
}

VKX_EXPORT int vkx_fill_f32(vkx_ctx *ctx, float *dst, int h, int w, ptrdiff_t dst_stride_el,
                            const vkx_layer_f32 *layers, int n_layers)
{
    VKX_REQUIRE(ctx && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0, "bad shape");
    VKX_REQUIRE(n_layers >= 0 && (n_layers == 0 || layers), "bad layer list");
    HostStage st(ctx);
    const int d = st.add(dst, dst, (size_t)w * 4, h, dst_stride_el * 4);
    std::vector<int> mid(n_layers, -1), aid(n_layers, -1), vid(n_layers, -1);
    for (int i = 0; i < n_layers; i++) {
        const vkx_layer_f32 &l = layers[i];
        VKX_REQUIRE(l.height >= 0 && l.width >= 0, "bad layer box");
        if (l.mask) mid[i] = st.add(l.mask, nullptr, (size_t)l.width, l.height, l.mask_stride);
        if (l.alpha) aid[i] = st.add(l.alpha, nullptr, (size_t)l.width * 4, l.height, l.alpha_stride_el * 4);
        if (l.value) vid[i] = st.add(l.value, nullptr, (size_t)l.width * 4, l.height, l.value_stride_el * 4);
    }
    VKX_TRY(st.commit());
    std::vector<vkx_layer_f32> dl(layers, layers + n_layers);
    for (int i = 0; i < n_layers; i++) {
        dl[i].mask = st.dev<uint8_t>(mid[i]);
        dl[i].mask_stride = layers[i].width;
        dl[i].alpha = st.dev<float>(aid[i]);
        dl[i].alpha_stride_el = layers[i].width;
        dl[i].value = st.dev<float>(vid[i]);
        dl[i].value_stride_el = layers[i].width;
    }
    VKX_TRY(vkx_fill_f32_dev(ctx, st.dev<float>(d), h, w, w, dl.data(), n_layers));
    return st.finish();
}

VKX_EXPORT int vkx_resize_cubic_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                                   uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)sw * cn, sh, src_stride);
    const int d = st.add(nullptr, dst, (size_t)dw * cn, dh, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_resize_cubic_u8_dev(ctx, st.dev<uint8_t>(s), sh, sw, cn, (ptrdiff_t)sw * cn, st.dev<uint8_t>(d), dh, dw,
                                    (ptrdiff_t)dw * cn));
    return st.finish();
}

VKX_EXPORT int vkx_resize_cubic_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el, float *dst,
                                    int dh, int dw, ptrdiff_t dst_stride_el)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)sw * 4, sh, src_stride_el * 4);
    const int d = st.add(nullptr, dst, (size_t)dw * 4, dh, dst_stride_el * 4);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_resize_cubic_f32_dev(ctx, st.dev<float>(s), sh, sw, sw, st.dev<float>(d), dh, dw, dw));
    return st.finish();
}

VKX_EXPORT int vkx_resize_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el, float *dst, int dh, int dw,
                              ptrdiff_t dst_stride_el, int interpolation)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)sw * 4, sh, src_stride_el * 4);
    const int d = st.add(nullptr, dst, (size_t)dw * 4, dh, dst_stride_el * 4);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_resize_f32_dev(ctx, st.dev<float>(s), sh, sw, sw, st.dev<float>(d), dh, dw, dw, interpolation));
    return st.finish();
}

VKX_EXPORT int vkx_filter2d_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                               const float *kernel_host, int kh, int kw, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_filter2d_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, kernel_host, kh, kw, st.dev<uint8_t>(d),
                                (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_pointwise_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int op, int p0,
                                int p1, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_pointwise_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, op, p0, p1, channel_mask,
                                 st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_impulse_noise_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                    const uint8_t *selector, ptrdiff_t selector_stride, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && selector && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int m = st.add(selector, nullptr, (size_t)w, h, selector_stride);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_impulse_noise_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, st.dev<uint8_t>(m), w,
                                     st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_speckle_noise_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                    const double *noise, ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && noise && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int n = st.add(noise, nullptr, (size_t)w * cn * 8, h, noise_stride_el * 8);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_speckle_noise_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, st.dev<double>(n),
                                     (ptrdiff_t)w * cn, st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_cvt_color_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int code, uint8_t *dst,
                                ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0, "bad shape");
    const int scn = code == VKX_CVT_GRAY2RGB || code == VKX_CVT_GRAY2RGBA ? 1 : (code == VKX_CVT_RGBA2RGB || code == VKX_CVT_RGBA2GRAY ? 4 : 3);
    const int dcn = code == VKX_CVT_RGB2GRAY || code == VKX_CVT_RGBA2GRAY ? 1 : (code == VKX_CVT_RGB2RGBA || code == VKX_CVT_GRAY2RGBA ? 4 : 3);
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * scn, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * dcn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_cvt_color_u8_dev(ctx, st.dev<uint8_t>(s), h, w, (ptrdiff_t)w * scn, code, st.dev<uint8_t>(d),
                                 (ptrdiff_t)w * dcn));
    return st.finish();
}

VKX_EXPORT int vkx_blend_u8(vkx_ctx *ctx, const uint8_t *a, ptrdiff_t a_stride, const uint8_t *b, ptrdiff_t b_stride, int h, int w,
                            int cn, double w0, double w1, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && a && b && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn >= 1 && cn <= 4, "bad shape");
    HostStage st(ctx);
    const int ia = st.add(a, nullptr, (size_t)w * cn, h, a_stride);
    const int ib = st.add(b, nullptr, (size_t)w * cn, h, b_stride);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_blend_u8_dev(ctx, st.dev<uint8_t>(ia), (ptrdiff_t)w * cn, st.dev<uint8_t>(ib), (ptrdiff_t)w * cn, h, w, cn, w0, w1,
                             channel_mask, st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_fog_f32_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const float *weight,
                              ptrdiff_t weight_stride_el, const float *fog, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && weight && fog && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn >= 1 && cn <= 4, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int m = st.add(weight, nullptr, (size_t)w * 4, h, weight_stride_el * 4);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_fog_f32_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, st.dev<float>(m), w, fog, st.dev<uint8_t>(d),
                               (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_brightness_shift_rgb(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta,
                                        uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * 3, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * 3, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_brightness_shift_rgb_dev(ctx, st.dev<uint8_t>(s), h, w, (ptrdiff_t)w * 3, delta, st.dev<uint8_t>(d),
                                         (ptrdiff_t)w * 3));
    return st.finish();
}

VKX_EXPORT int vkx_color_balance_rgb(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, double ratio,
                                     uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * 3, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * 3, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_color_balance_rgb_dev(ctx, st.dev<uint8_t>(s), h, w, (ptrdiff_t)w * 3, ratio, st.dev<uint8_t>(d),
                                      (ptrdiff_t)w * 3));
    return st.finish();
}

VKX_EXPORT int vkx_histogram_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int32_t *hist)
{
    VKX_REQUIRE(ctx && src && hist, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn >= 1 && cn <= 4, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int d = st.add(nullptr, hist, sizeof(int32_t) * 256 * cn, 1, sizeof(int32_t) * 256 * cn);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_histogram_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, st.dev<int32_t>(d)));
    return st.finish();
}

VKX_EXPORT int vkx_apply_lut_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                const uint8_t *lut_host, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst && lut_host, "NULL argument");
    VKX_REQUIRE(h >= 0 && w >= 0 && cn >= 1 && cn <= 4, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_apply_lut_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, lut_host, channel_mask,
                                 st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

VKX_EXPORT int vkx_gather_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                             const int32_t *pos_y, const int32_t *pos_x, ptrdiff_t pos_stride_el, uint8_t *dst, int dh, int dw,
                             ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && pos_y && pos_x && dst, "NULL argument");
    VKX_REQUIRE(sh >= 0 && sw >= 0 && dh >= 0 && dw >= 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)sw * cn, sh, src_stride);
    const int py = st.add(pos_y, nullptr, (size_t)dw * 4, dh, pos_stride_el * 4);
    const int px = st.add(pos_x, nullptr, (size_t)dw * 4, dh, pos_stride_el * 4);
    const int d = st.add(nullptr, dst, (size_t)dw * cn, dh, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_gather_u8_dev(ctx, st.dev<uint8_t>(s), sh, sw, cn, (ptrdiff_t)sw * cn, st.dev<int32_t>(py), st.dev<int32_t>(px),
                              dw, st.dev<uint8_t>(d), dh, dw, (ptrdiff_t)dw * cn));
    return st.finish();
}

VKX_EXPORT int vkx_resize_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride, uint8_t *dst,
                             int dh, int dw, ptrdiff_t dst_stride, int interpolation)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)sw * cn, sh, src_stride);
    const int d = st.add(nullptr, dst, (size_t)dw * cn, dh, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_resize_u8_dev(ctx, st.dev<uint8_t>(s), sh, sw, cn, (ptrdiff_t)sw * cn, st.dev<uint8_t>(d), dh, dw,
                              (ptrdiff_t)dw * cn, interpolation));
    return st.finish();
}

VKX_EXPORT int vkx_saturate_i64_u8(vkx_ctx *ctx, const int64_t *src, size_t n, uint8_t *dst)
{
    VKX_REQUIRE(ctx && (n == 0 || (src && dst)), "NULL argument");
    if (n == 0) return VKX_OK;
    HostStage st(ctx);
    const int s = st.add(src, nullptr, n * 8, 1, (ptrdiff_t)(n * 8));
    const int d = st.add(nullptr, dst, n, 1, (ptrdiff_t)n);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_saturate_i64_u8_dev(ctx, st.dev<int64_t>(s), n, st.dev<uint8_t>(d)));
    return st.finish();
}

VKX_EXPORT int vkx_zoom_in_blur_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                                   const int32_t *sizes_hw_host, int n_sizes, double alpha, uint8_t *dst, ptrdiff_t dst_stride)
{
    VKX_REQUIRE(ctx && src && dst, "NULL argument");
    VKX_REQUIRE(h > 0 && w > 0 && cn > 0, "bad shape");
    HostStage st(ctx);
    const int s = st.add(src, nullptr, (size_t)w * cn, h, src_stride);
    const int d = st.add(nullptr, dst, (size_t)w * cn, h, dst_stride);
    VKX_TRY(st.commit());
    VKX_TRY(vkx_zoom_in_blur_u8_dev(ctx, st.dev<uint8_t>(s), h, w, cn, (ptrdiff_t)w * cn, sizes_hw_host, n_sizes, alpha,
                                    st.dev<uint8_t>(d), (ptrdiff_t)w * cn));
    return st.finish();
}

