// The batched geometric + photometric chain on device-resident RGB images:
//   image-grid remap (camera_* / similarity_mls state) -> gaussian_blur -> color_shift -> gaussion_noise -> line_streak,
// i.e. BASELINE config 3.  Every item is an independent image with its own grid, destination size and
// parameters (ragged batch); stages with a disabled parameter are skipped.
//
// Two implementations with identical pixels: the tile-fused kernel of fused.hip (preferred) and, for shapes it
// declines or under VKX_CHAIN_STAGED=1, the individual kernels of grid.hip / photo.hip below, ping-ponging through two
// ctx-owned planes so that only the final stage writes the caller's destination.
#include "vkx_internal.h"

#include <stdlib.h>
#include <algorithm>
#include <memory>
#include <unordered_map>
#include <vector>

VKX_EXPORT int vkx_chain_rgb_batch_dev(vkx_ctx *ctx, const vkx_chain_item *items, int n_items)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    VKX_REQUIRE(n_items >= 0 && (n_items == 0 || items), "bad item list");
    size_t max_plane = 0;
    for (int i = 0; i < n_items; i++) {
        const vkx_chain_item &it = items[i];
        VKX_REQUIRE(it.src && it.dst && it.src_vertices && it.dst_vertices, "NULL plane in chain item");
        VKX_REQUIRE(it.sh > 0 && it.sw > 0 && it.dh > 0 && it.dw > 0, "bad shape in chain item");
        if (it.streak_enabled) {
            VKX_REQUIRE(it.streak_thickness + it.streak_gap > 0, "streak thickness + gap must be positive");
            if (it.streak_alpha < 0.0 || it.streak_alpha > 1.0) {
                vkx_set_error("alpha=%g is invalid.", it.streak_alpha);
                return VKX_ERR_INVALID;
            }
        }
        const size_t bytes = (size_t)it.dh * it.dw * 3;
        if (bytes > max_plane) max_plane = bytes;
    }
    // Preferred: the tile-fused kernel (fused.hip).  VKX_CHAIN_STAGED=1 forces the per-stage kernels (A/B runs).
    static const bool staged_only = [] { const char *e = getenv("VKX_CHAIN_STAGED"); return e && e[0] == '1'; }();
    if (!staged_only) {
        const int frc = vkx_chain_fused_try(ctx, items, n_items);
        if (frc != VKX_ERR_UNSUPPORTED) return frc;
    }
    // the staged kernels run on the compute stream: lattices marked ready on another stream (vkx_camera_states_dev) first
    if (int mrc = vkx_chain_consume_lattices_mark(ctx)) return mrc;
    for (int k = 0; k < 2; k++) {
        int rc = vkx_scratch_reserve(ctx, &ctx->chain[k], max_plane);
        if (rc) return rc;
    }
    for (int i = 0; i < n_items; i++) {
        const vkx_chain_item &it = items[i];
        const bool blur = it.blur_ksize > 1;
        const bool hue = it.hue_enabled != 0;
        const bool noise = it.noise != nullptr;
        const int stages_after_remap = (blur ? 1 : 0) + (hue ? 1 : 0) + (noise ? 1 : 0);
        const ptrdiff_t tmp_stride = (ptrdiff_t)it.dw * 3;
        int remaining = stages_after_remap;
        int flip = 0;
        // output plane of the next stage: the caller's destination for the last one, a scratch plane otherwise
        auto next_out = [&](uint8_t *&out, ptrdiff_t &stride) {
            if (remaining == 0) { out = it.dst; stride = it.dst_stride; }
            else { out = (uint8_t *)ctx->chain[flip].ptr; stride = tmp_stride; flip ^= 1; }
        };
        uint8_t *cur = nullptr;
        ptrdiff_t cur_stride = 0;
        next_out(cur, cur_stride);
        vkx_elem e;
        e.src = it.src; e.dst = cur; e.src_stride = it.src_stride; e.dst_stride = cur_stride; e.cn = 3; e.is_f32 = 0;
        int rc = vkx_grid_remap_dev(ctx, &e, 1, it.sh, it.sw, it.src_vertices, it.dst_vertices, it.rows, it.cols, it.dh,
                                    it.dw);
        if (rc) return rc;
        if (blur) {
            remaining--;
            uint8_t *out; ptrdiff_t ostride;
            next_out(out, ostride);
            rc = vkx_gaussian_blur_u8_dev(ctx, cur, it.dh, it.dw, 3, cur_stride, it.blur_ksize, it.blur_sigma, out, ostride);
            if (rc) return rc;
            cur = out; cur_stride = ostride;
        }
        if (hue) {
            remaining--;
            uint8_t *out; ptrdiff_t ostride;
            next_out(out, ostride);
            rc = vkx_color_shift_rgb_dev(ctx, cur, it.dh, it.dw, cur_stride, it.hue_delta, out, ostride);
            if (rc) return rc;
            cur = out; cur_stride = ostride;
        }
        if (noise) {
            remaining--;
            uint8_t *out; ptrdiff_t ostride;
            next_out(out, ostride);
            const int16_t *plane = it.noise;
            ptrdiff_t plane_stride = it.noise_stride_el;
            if (it.noise_tiled) {      // the generator's tile buffer: as the plane it stands for
                const long long n = (long long)it.dh * it.dw * 3;
                if ((rc = vkx_scratch_reserve(ctx, &ctx->chain[2], (size_t)n * 2))) return rc;
                if ((rc = vkx_np_tiles_expand_dev(ctx, it.noise, n, (int16_t *)ctx->chain[2].ptr))) return rc;
                plane = (const int16_t *)ctx->chain[2].ptr;
                plane_stride = (ptrdiff_t)it.dw * 3;
            }
            rc = vkx_add_noise_i16_dev(ctx, cur, it.dh, it.dw, 3, cur_stride, plane, plane_stride, out, ostride);
            if (rc) return rc;
            cur = out; cur_stride = ostride;
        }
        if (it.streak_enabled) {   // in place on the caller's destination, which the last stage above has written
            rc = vkx_line_streak_u8_dev(ctx, it.dst, it.dh, it.dw, 3, it.dst_stride, it.streak_thickness, it.streak_gap,
                                        it.streak_dash_thickness, it.streak_dash_gap, it.streak_color, it.streak_alpha,
                                        it.streak_enable_vert, it.streak_enable_hori);
            if (rc) return rc;
        }
    }
    return VKX_OK;
}

// The numpy streams of a batch's gaussion_noise members AND its chain in one call, with the dependencies known to the library:
//   jobs      VKX_NP_NORMAL_TILES jobs; job k draws the tile buffer that is the `noise` of exactly one item (matched by pointer);
//   items     as for vkx_chain_rgb_batch_dev.
// Same results as vkx_np_draw_batch_dev(jobs) followed by vkx_chain_rgb_batch_dev(items).  What changes is where the small kernels
// run.  Either call alone is a sequence on one stream: tile states, DRAW (VALU bound, 7 ms per 256 pages), carry resolution, walk
// of the rare tiles, tile tables; then descriptor prologue, cell setup, Jacobi cells, noise row records, the PIXEL kernel (9 ms)
// -- eight microsecond kernels of a few workgroups each between and before the two large ones, 0.8 ms of a 16.9 ms step during
// which most of the device idles.  Here the batch is cut into chunks of images:
//   compute stream   draw(0) draw(1) .. draw(C-1) | pixels(0) pixels(1) .. pixels(C-1)
//   aux stream              post(0)  post(1) ..        post(C-1)           post(c) = resolve + walk + tables + noise rows of chunk c
//   side stream      prologue + cell setup + Jacobi cells of the whole batch (depends on the lattices only)
// post(c) runs under draw(c + 1) -- the last one under pixels(0) --, the setup under draw(0).
VKX_EXPORT int vkx_chain_rgb_batch_np_dev(vkx_ctx *ctx, const vkx_chain_item *items, int n_items, const vkx_np_job *jobs, int n_jobs,
                                          vkx_np_result *results_host)
{
    VKX_REQUIRE(ctx != nullptr, "ctx is NULL");
    VKX_REQUIRE(n_items >= 0 && (n_items == 0 || items), "bad item list");
    if (n_jobs <= 0) return vkx_chain_rgb_batch_dev(ctx, items, n_items);
    int rc = vkx_np_jobs_check(jobs, n_jobs, results_host);
    if (rc) return rc;
    // job -> item by the tile buffer; the jobs of a chunk must be a contiguous run of the job list
    std::unordered_map<const void *, int> item_of;
    for (int i = 0; i < n_items; i++)
        if (items[i].noise && items[i].noise_tiled) item_of[(const void *)items[i].noise] = i;
    bool pipelined = (jobs[0].kind & 0xff) == VKX_NP_NORMAL_TILES;
    std::vector<int> job_item((size_t)n_jobs, -1);
    for (int k = 0; k < n_jobs && pipelined; k++) {
        auto it = item_of.find(jobs[k].dst);
        if (it == item_of.end() || (k > 0 && it->second <= job_item[k - 1])) pipelined = false;    // unknown buffer / not in item order
        else job_item[k] = it->second;
    }
    static const bool staged_only = [] { const char *e = getenv("VKX_CHAIN_STAGED"); return e && e[0] == '1'; }();
    static const int want_chunks = [] { const char *e = getenv("VKX_CHAIN_CHUNKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2; }();
    static const int overlap = [] { const char *e = getenv("VKX_CHAIN_OVERLAP"); return e ? atoi(e) : 0; }();
    vkx_chain_plan *raw = nullptr;
    if (pipelined && !staged_only) {
        for (int i = 0; i < n_items; i++) {      // the checks of vkx_chain_rgb_batch_dev
            const vkx_chain_item &it = items[i];
            VKX_REQUIRE(it.src && it.dst && it.src_vertices && it.dst_vertices, "NULL plane in chain item");
            VKX_REQUIRE(it.sh > 0 && it.sw > 0 && it.dh > 0 && it.dw > 0, "bad shape in chain item");
            if (it.streak_enabled) {
                VKX_REQUIRE(it.streak_thickness + it.streak_gap > 0, "streak thickness + gap must be positive");
                if (it.streak_alpha < 0.0 || it.streak_alpha > 1.0) {
                    vkx_set_error("alpha=%g is invalid.", it.streak_alpha);
                    return VKX_ERR_INVALID;
                }
            }
        }
        rc = vkx_chain_plan_build(ctx, items, n_items, &raw);
        if (rc && rc != VKX_ERR_UNSUPPORTED) return rc;
    }
    if (!raw) {       // shapes the fused path does not take, other job kinds: one call after the other
        if ((rc = vkx_chain_consume_lattices_mark(ctx))) return rc;
        if ((rc = vkx_np_draw_batch_dev(ctx, jobs, n_jobs, results_host))) return rc;
        return vkx_chain_rgb_batch_dev(ctx, items, n_items);
    }
    std::unique_ptr<vkx_chain_plan, void (*)(vkx_chain_plan *)> plan(raw, vkx_chain_plan_free);
    // chunks of at least 8 jobs, of about equally many jobs; chunk c owns the items up to (excluding) the first item of chunk c + 1's jobs
    const int n_chunks = std::max(1, std::min(want_chunks, n_jobs / 8));
    struct Chunk { int job0, njobs, item0, nitems; vkx_np_chunk *np; hipEvent_t posted; };
    std::vector<Chunk> chunks((size_t)n_chunks);
    for (int c = 0; c < n_chunks; c++) {
        Chunk &ch = chunks[c];
        ch.job0 = (int)((long long)n_jobs * c / n_chunks);
        ch.njobs = (int)((long long)n_jobs * (c + 1) / n_chunks) - ch.job0;
        ch.item0 = c == 0 ? 0 : job_item[ch.job0];
        ch.np = nullptr; ch.posted = nullptr;
    }
    for (int c = 0; c < n_chunks; c++) chunks[c].nitems = (c + 1 < n_chunks ? chunks[c + 1].item0 : n_items) - chunks[c].item0;
    struct Cleanup {
        std::vector<Chunk> &chunks;
        ~Cleanup() { for (Chunk &ch : chunks) { if (ch.np) vkx_np_chunk_free(ch.np); if (ch.posted) (void)hipEventDestroy(ch.posted); } }
    } cleanup{chunks};

    vkx_device_guard guard(ctx);
    hipStream_t main_stream = ctx->stream;
    // The three-stream schedule.  Every exit of it -- a failing chunk included -- leaves the context with ctx->stream == main_stream and
    // main_stream ordered after whatever reached aux / side: a caller that goes on using the context after an error cannot race the
    // kernels this call left in flight (tests/test_gpu_chain_errors.py).
    const char *fail_env = getenv("VKX_DEBUG_FAIL_CHUNK");      // read per call: tests set and clear it around one call
    const int fail_chunk = fail_env && fail_env[0] ? atoi(fail_env) : -1;
    auto schedule = [&]() -> int {
        int rc = VKX_OK;
        hipStream_t aux = vkx_stream_by_id(ctx, VKX_STREAM_COPY_IN, &rc);
        if (rc) return rc;
        hipStream_t side = vkx_stream_by_id(ctx, VKX_STREAM_COPY_OUT, &rc);
        if (rc) return rc;
        // what the caller queued before this call (the sources may be its product)
        hipEvent_t entry = nullptr;
        if (overlap) {
            VKX_HIP(hipEventCreateWithFlags(&entry, hipEventDisableTiming));
            hipError_t e = hipEventRecord(entry, main_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(side, entry, 0);
            (void)hipEventDestroy(entry);       // (deferred until the event has completed)
            VKX_HIP(e);
        }
        // side stream: the cell setup of the whole batch (the compute stream is ordered after it by the call)
        hipEvent_t setup_done = nullptr;
        if ((rc = vkx_chain_plan_setup_aside(ctx, raw, &setup_done))) return rc;
        if (setup_done) VKX_HIP(hipStreamWaitEvent(aux, setup_done, 0));     // the noise row records read the descriptors the prologue copies
        for (int c = 0; c < n_chunks; c++) {
            Chunk &ch = chunks[c];
            // the scratch slot's previous tenant (chunk c - 2) must have been posted before its tile arrays are overwritten
            if (c >= 2) VKX_HIP(hipStreamWaitEvent(main_stream, chunks[c - 2].posted, 0));
            if ((rc = vkx_np_chunk_begin(ctx, jobs + ch.job0, ch.njobs, results_host + ch.job0, c & 1, &ch.np))) return rc;
            if ((rc = vkx_stream_order(ctx, aux, main_stream))) return rc;
            ctx->stream = aux;
            rc = vkx_np_chunk_finish(ctx, ch.np);
            if (!rc) rc = vkx_chain_plan_noise_rows(ctx, raw, ch.item0, ch.nitems);
            ctx->stream = main_stream;
            if (rc) return rc;
            VKX_HIP(hipEventCreateWithFlags(&ch.posted, hipEventDisableTiming));
            VKX_HIP(hipEventRecord(ch.posted, aux));
            if (c == fail_chunk) {       // fault injection (tests): a failure with draw(c) in flight on the compute stream and post(c) on aux
                vkx_set_error("injected failure in chunk %d (VKX_DEBUG_FAIL_CHUNK)", c);
                return VKX_ERR_INVALID;
            }
        }
        // VKX_CHAIN_OVERLAP=1: the pixel kernels on the side stream, so that pixels(c) shares the device with draw(c + 1 ..)
        hipStream_t pix = overlap ? side : main_stream;
        if (setup_done && pix != side) VKX_HIP(hipStreamWaitEvent(pix, setup_done, 0));
        for (int c = 0; c < n_chunks; c++) {
            Chunk &ch = chunks[c];
            VKX_HIP(hipStreamWaitEvent(pix, ch.posted, 0));
            ctx->stream = pix;
            rc = vkx_chain_plan_tiles(ctx, raw, ch.item0, ch.nitems);
            ctx->stream = main_stream;
            if (rc) return rc;
        }
        if ((rc = vkx_chain_mark_done(ctx, pix))) return rc;
        if (pix != main_stream && (rc = vkx_stream_order(ctx, main_stream, pix))) return rc;
        return VKX_OK;
    };
    rc = schedule();
    if (rc) vkx_ctx_join_streams(ctx, main_stream);
    return rc;
}
