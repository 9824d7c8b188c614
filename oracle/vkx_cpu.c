/* CPU twins of the product's C-ABI entry points on the headline path (last bullet): the SAME signatures as
 * include/vkx.h -- `vkx_X(ctx, ...)` becomes `vkx_cpu_X(ctx, ...)`, host pointers, `ctx` ignored (may be NULL) -- implemented by the
 * oracle's restatement of the reference arithmetic (oracle/vkx_oracle.c: each vko_* function cites the reference lines it follows).
 * A caller bound to libvkx.so can be pointed at this library to get the reference's CPU arithmetic through the very same calls;
 * tests/test_cpu_twins.py drives both through one ctypes signature table.
 *
 * TEST INFRASTRUCTURE, like everything under oracle/: the product never loads this library (tests/test_cabi_symbols.py).
 * Built without HIP (gcc, C99) into the oracle's shared object. */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vkx.h"          /* the structs, the error codes -- and the signatures the twins repeat */

#define VKC_API __attribute__((visibility("default")))

/* oracle/vkx_oracle.c */
int vko_remap_u8(const uint8_t *, int, int, int, ptrdiff_t, const float *, const float *, ptrdiff_t, uint8_t *, int, int, ptrdiff_t);
int vko_remap_f32(const float *, int, int, ptrdiff_t, const float *, const float *, ptrdiff_t, float *, int, int, ptrdiff_t);
int vko_warp_affine_coords(const double Mfwd[6], int dh, int dw, int *X, int *Y);
int vko_sample_fixed_u8(const uint8_t *, int, int, int, ptrdiff_t, const int *, const int *, uint8_t *, int, int, ptrdiff_t);
int vko_grid_to_map(const int32_t *, const int32_t *, int, int, int, int, int, float *, float *, int32_t *);
int vko_gaussian_blur_u8(const uint8_t *, int, int, int, ptrdiff_t, int, double, uint8_t *, ptrdiff_t);
int vko_color_shift_rgb(const uint8_t *, size_t, int, uint8_t *);
int vko_add_noise_i16(const uint8_t *, const int16_t *, size_t, uint8_t *);
int vko_fill_u8_mode(uint8_t *, int, int, int, ptrdiff_t, int, int, int, int, const uint8_t *, ptrdiff_t, const float *, ptrdiff_t, double,
                     const uint8_t *, ptrdiff_t, const uint8_t *, int);
int vko_line_streak_u8(uint8_t *, int, int, int, ptrdiff_t, int, int, int, int, const uint8_t *, double, int, int);

VKC_API int vkx_cpu_remap_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride, const float *map_x,
                             const float *map_y, ptrdiff_t map_stride_el, uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride)
{
    (void)ctx;
    return vko_remap_u8(src, sh, sw, cn, src_stride, map_x, map_y, map_stride_el, dst, dh, dw, dst_stride) ? VKX_ERR_INVALID : VKX_OK;
}

VKC_API int vkx_cpu_remap_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el, const float *map_x,
                              const float *map_y, ptrdiff_t map_stride_el, float *dst, int dh, int dw, ptrdiff_t dst_stride_el)
{
    (void)ctx;
    return vko_remap_f32(src, sh, sw, src_stride_el, map_x, map_y, map_stride_el, dst, dh, dw, dst_stride_el) ? VKX_ERR_INVALID : VKX_OK;
}

VKC_API int vkx_cpu_warp_affine_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride, const double M[6],
                                   uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride)
{
    (void)ctx;
    int *X = (int *)malloc(sizeof(int) * 2 * (size_t)dh * (size_t)dw);
    if (!X) return VKX_ERR_NOMEM;
    int *Y = X + (size_t)dh * dw;
    int rc = vko_warp_affine_coords(M, dh, dw, X, Y);
    if (!rc) rc = vko_sample_fixed_u8(src, sh, sw, cn, src_stride, X, Y, dst, dh, dw, dst_stride);
    free(X);
    return rc ? VKX_ERR_INVALID : VKX_OK;
}

/* the maps are dense here: map_stride_el == dw (the oracle writes contiguous planes) */
VKC_API int vkx_cpu_grid_to_map(vkx_ctx *ctx, const int32_t *src_vertices, const int32_t *dst_vertices, int rows, int cols, int dh, int dw,
                                float *map_x, float *map_y, ptrdiff_t map_stride_el, int32_t *owner)
{
    (void)ctx;
    if (map_stride_el != dw) return VKX_ERR_UNSUPPORTED;
    return vko_grid_to_map(src_vertices, dst_vertices, rows, cols, dh, dw, 0 /* the parity definition: closed form, Jacobi for degenerate cells */,
                           map_x, map_y, owner) < 0 ? VKX_ERR_INVALID : VKX_OK;
}

VKC_API int vkx_cpu_grid_remap(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw, const int32_t *src_vertices,
                               const int32_t *dst_vertices, int rows, int cols, int dh, int dw)
{
    (void)ctx;
    float *mx = (float *)malloc(sizeof(float) * 2 * (size_t)dh * (size_t)dw);
    if (!mx) return VKX_ERR_NOMEM;
    float *my = mx + (size_t)dh * dw;
    int rc = vko_grid_to_map(src_vertices, dst_vertices, rows, cols, dh, dw, 0, mx, my, NULL) < 0;
    for (int e = 0; e < n_elems && !rc; e++) {
        const vkx_elem *el = &elems[e];
        if (el->is_f32) rc = vko_remap_f32((const float *)el->src, sh, sw, el->src_stride, mx, my, dw, (float *)el->dst, dh, dw, el->dst_stride);
        else rc = vko_remap_u8((const uint8_t *)el->src, sh, sw, el->cn, el->src_stride, mx, my, dw, (uint8_t *)el->dst, dh, dw, el->dst_stride);
    }
    free(mx);
    return rc ? VKX_ERR_INVALID : VKX_OK;
}

VKC_API int vkx_cpu_gaussian_blur_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int ksize, double sigma,
                                     uint8_t *dst, ptrdiff_t dst_stride)
{
    (void)ctx;
    return vko_gaussian_blur_u8(src, h, w, cn, src_stride, ksize, sigma, dst, dst_stride) ? VKX_ERR_INVALID : VKX_OK;
}

VKC_API int vkx_cpu_color_shift_rgb(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta, uint8_t *dst,
                                    ptrdiff_t dst_stride)
{
    (void)ctx;
    for (int y = 0; y < h; y++)
        if (vko_color_shift_rgb(src + (ptrdiff_t)y * src_stride, (size_t)w, delta, dst + (ptrdiff_t)y * dst_stride)) return VKX_ERR_INVALID;
    return VKX_OK;
}

VKC_API int vkx_cpu_add_noise_i16(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const int16_t *noise,
                                  ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride)
{
    (void)ctx;
    for (int y = 0; y < h; y++)
        vko_add_noise_i16(src + (ptrdiff_t)y * src_stride, noise + (ptrdiff_t)y * noise_stride_el, (size_t)w * cn, dst + (ptrdiff_t)y * dst_stride);
    return VKX_OK;
}

VKC_API int vkx_cpu_line_streak_u8(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int thickness, int gap,
                                   int dash_thickness, int dash_gap, const uint8_t color[4], double alpha, int enable_vert, int enable_hori)
{
    (void)ctx;
    return vko_line_streak_u8(img, h, w, cn, stride, thickness, gap, dash_thickness, dash_gap, color, alpha, enable_vert, enable_hori)
               ? VKX_ERR_INVALID : VKX_OK;
}

VKC_API int vkx_cpu_fill_u8(vkx_ctx *ctx, uint8_t *dst, int h, int w, int cn, ptrdiff_t dst_stride, const vkx_layer *layers, int n_layers)
{
    (void)ctx;
    for (int i = 0; i < n_layers; i++) {
        const vkx_layer *L = &layers[i];
        if (vko_fill_u8_mode(dst, h, w, cn, dst_stride, L->up, L->left, L->height, L->width, L->mask, L->mask_stride, L->alpha,
                             L->alpha_stride_el, L->alpha_scalar, L->value, L->value_stride, L->value_const, L->mode))
            return VKX_ERR_INVALID;
    }
    return VKX_OK;
}

/* vkx_chain_rgb_batch_dev's twin on HOST arrays: every pointer of an item is a host pointer; `noise` is an int16 plane (a
 * generator's tile buffer is a device format: VKX_ERR_UNSUPPORTED).  Stage order and skipping rules are the product's. */
VKC_API int vkx_cpu_chain_rgb_batch(vkx_ctx *ctx, const vkx_chain_item *items, int n_items)
{
    (void)ctx;
    for (int i = 0; i < n_items; i++) {
        const vkx_chain_item *it = &items[i];
        if (!it->src || !it->dst || !it->src_vertices || !it->dst_vertices || it->dh <= 0 || it->dw <= 0) return VKX_ERR_INVALID;
        if (it->noise && it->noise_tiled) return VKX_ERR_UNSUPPORTED;
        const size_t px = (size_t)it->dh * it->dw;
        float *mx = (float *)malloc(sizeof(float) * 2 * px);
        uint8_t *a = (uint8_t *)malloc(px * 3), *b = (uint8_t *)malloc(px * 3);
        if (!mx || !a || !b) { free(mx); free(a); free(b); return VKX_ERR_NOMEM; }
        float *my = mx + px;
        const ptrdiff_t step = (ptrdiff_t)it->dw * 3;
        int rc = vko_grid_to_map(it->src_vertices, it->dst_vertices, it->rows, it->cols, it->dh, it->dw, 0, mx, my, NULL) < 0;
        if (!rc) rc = vko_remap_u8(it->src, it->sh, it->sw, 3, it->src_stride, mx, my, it->dw, a, it->dh, it->dw, step);
        uint8_t *cur = a, *other = b;
        if (!rc && it->blur_ksize > 1) {
            rc = vko_gaussian_blur_u8(cur, it->dh, it->dw, 3, step, it->blur_ksize, it->blur_sigma, other, step);
            uint8_t *t = cur; cur = other; other = t;
        }
        if (!rc && it->hue_enabled) rc = vko_color_shift_rgb(cur, px, it->hue_delta, cur);
        if (!rc && it->noise)
            for (int y = 0; y < it->dh; y++)
                vko_add_noise_i16(cur + (ptrdiff_t)y * step, it->noise + (ptrdiff_t)y * it->noise_stride_el, (size_t)it->dw * 3, cur + (ptrdiff_t)y * step);
        if (!rc && it->streak_enabled)
            rc = vko_line_streak_u8(cur, it->dh, it->dw, 3, step, it->streak_thickness, it->streak_gap, it->streak_dash_thickness,
                                    it->streak_dash_gap, it->streak_color, it->streak_alpha, it->streak_enable_vert, it->streak_enable_hori);
        for (int y = 0; y < it->dh && !rc; y++) memcpy(it->dst + (ptrdiff_t)y * it->dst_stride, cur + (ptrdiff_t)y * step, (size_t)step);
        free(mx); free(a); free(b);
        if (rc) return VKX_ERR_INVALID;
    }
    return VKX_OK;
}
