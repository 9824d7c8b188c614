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
