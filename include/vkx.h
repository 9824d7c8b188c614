/*
 * vkx.h -- C ABI of libvkx.so: the MI355X (gfx950) implementation of vkit's per-pixel
 * distortion hot path.  This header is the drop-in boundary: it is what a binding on the
 * reference side (ctypes) loads instead of the cv2 / numpy calls cited
 * at each entry point.  Paths are relative to the reference's vkit/ package.
 *
 * Conventions
 *   - Every function returns 0 (VKX_OK) or a negative error code; vkx_last_error() returns a
 *     thread-local human readable message for the last failure on the calling thread.
 *   - A vkx_ctx owns one HIP stream and all device scratch of one (process, GPU) pair.  Calls
 *     on one ctx must be serialised by the caller.  ctypes releases the GIL around calls.
 *   - `*_dev` entry points take DEVICE pointers, enqueue on the ctx stream and return
 *     immediately (vkx_ctx_sync to wait).  Entry points without the suffix take HOST
 *     pointers, stage through ctx scratch and are synchronous; the library never retains a
 *     host pointer past the call.
 *   - Images are row-major, channel-interleaved; `*_stride` is the row pitch in BYTES for
 *     uint8 planes and in ELEMENTS for float32 / int16 / int32 planes (`*_stride_el`).
 *   - Integer and byte results are bit-exact with oracle/ (the CPU restatement of the
 *     reference's numpy/OpenCV arithmetic); float32 results (ScoreMap) are bit-exact too,
 *     the stated tolerance against cv2 itself is 2 ulp.
 */
#ifndef VKX_H_
#define VKX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKX_OK 0
#define VKX_ERR_INVALID (-1)     /* bad argument */
#define VKX_ERR_HIP (-2)         /* HIP runtime failure (message has the hipError string) */
#define VKX_ERR_NOMEM (-3)
#define VKX_ERR_UNSUPPORTED (-4)
#define VKX_ERR_OUT_OF_LATTICE (-5) /* a point falls outside the lattice cells (the reference raises IndexError there) */
#define VKX_ERR_DIVIDE (-6)         /* a division by zero the reference raises FloatingPointError for */

typedef struct vkx_ctx vkx_ctx;

/* ---- context, stream, memory -------------------------------------------------------- */
int vkx_version(void);
const char *vkx_last_error(void);
int vkx_device_count(int *count);
/* "0000:c1:00.0" of visible device `device` (hipDeviceGetPCIBusId): process placement next to the GPU's NUMA node (vkit_amd/shard.py) */
int vkx_device_pci_bus_id(int device, char *buf, int len);
int vkx_ctx_create(int device, vkx_ctx **out);
int vkx_ctx_destroy(vkx_ctx *ctx);
int vkx_ctx_sync(vkx_ctx *ctx);
/* Borrow an external hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL restores
 * the ctx's own stream. */
int vkx_ctx_set_stream(vkx_ctx *ctx, void *hip_stream);
void *vkx_ctx_stream(vkx_ctx *ctx);
int vkx_malloc(vkx_ctx *ctx, size_t bytes, void **dptr);
int vkx_free(vkx_ctx *ctx, void *dptr);
int vkx_upload(vkx_ctx *ctx, void *dptr, const void *hptr, size_t bytes);   /* synchronous */
int vkx_download(vkx_ctx *ctx, void *hptr, const void *dptr, size_t bytes); /* synchronous */
int vkx_memset(vkx_ctx *ctx, void *dptr, int value, size_t bytes);          /* async */

/* ---- page-locked host memory, copy streams, ordering ----------------------------------------
 * For callers that keep host arrays flowing (the reference hands numpy arrays in and out of every operator,
 * mechanism/distortion/interface.py:824-912; its process pool, utility/pool.py:65-96, is the only overlap it has):
 * a ctx owns a host->device and a device->host copy stream next to its compute stream.  vkx_upload_async /
 * vkx_download_async enqueue on those; vkx_ctx_order(later, earlier) makes `later` wait for everything queued on
 * `earlier` at the time of the call; vkx_event_record / vkx_event_wait let the host wait for one point of one stream.
 * Copies overlap compute only from / to page-locked memory (vkx_host_alloc). */
#define VKX_STREAM_COMPUTE 0
#define VKX_STREAM_COPY_IN 1
#define VKX_STREAM_COPY_OUT 2
int vkx_host_alloc(vkx_ctx *ctx, size_t bytes, void **hptr);
int vkx_host_free(vkx_ctx *ctx, void *hptr);
int vkx_upload_async(vkx_ctx *ctx, void *dptr, const void *hptr, size_t bytes);
int vkx_download_async(vkx_ctx *ctx, void *hptr, const void *dptr, size_t bytes);
/* one copy on the named stream of the ctx: to_device 0 = device -> host, 1 = host -> device, 2 = device -> device */
int vkx_memcpy_async(vkx_ctx *ctx, int stream, void *dst, const void *src, size_t bytes, int to_device);
int vkx_ctx_order(vkx_ctx *ctx, int later_stream, int earlier_stream);
int vkx_ctx_sync_stream(vkx_ctx *ctx, int stream);
int vkx_event_record(vkx_ctx *ctx, int stream, void **event);
int vkx_event_wait(vkx_ctx *ctx, void *event);   /* waits on the host and releases the event */

/* ---- backward-map bilinear remap ----------------------------------------------------
 * cv.remap(src, map_x, map_y, cv.INTER_LINEAR), BORDER_CONSTANT 0
 *   mechanism/distortion/geometric/grid_rendering/grid_blender.py:60 (Image), :80 (Mask,
 *   bilinear on 0/1 bytes), :70 (ScoreMap).  cn in {1,3,4} for uint8. */
int vkx_remap_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                     const float *map_x, const float *map_y, ptrdiff_t map_stride_el,
                     uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride);
int vkx_remap_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                      const float *map_x, const float *map_y, ptrdiff_t map_stride_el,
                      float *dst, int dh, int dw, ptrdiff_t dst_stride_el);
int vkx_remap_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                 const float *map_x, const float *map_y, ptrdiff_t map_stride_el,
                 uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride);
int vkx_remap_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                  const float *map_x, const float *map_y, ptrdiff_t map_stride_el,
                  float *dst, int dh, int dw, ptrdiff_t dst_stride_el);

/* ---- affine / perspective warps -----------------------------------------------------
 * cv.warpAffine(mat, trans_mat, dsize) / cv.warpPerspective(mat, trans_mat, dsize)
 *   mechanism/distortion/geometric/affine.py:38-43 (affine_mat), used by :430-456.
 * M is the FORWARD matrix (row-major 2x3 / 3x3, float32-valued doubles), dsize = (dw, dh). */
int vkx_warp_affine_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                           const double M[6], uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride);
int vkx_warp_affine_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                            const double M[6], float *dst, int dh, int dw, ptrdiff_t dst_stride_el);
int vkx_warp_perspective_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                                const double M[9], uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride);
int vkx_warp_perspective_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                                 const double M[9], float *dst, int dh, int dw, ptrdiff_t dst_stride_el);
int vkx_warp_affine_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                       const double M[6], uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride);
int vkx_warp_affine_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                        const double M[6], float *dst, int dh, int dw, ptrdiff_t dst_stride_el);
int vkx_warp_perspective_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                            const double M[9], uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride);
int vkx_warp_perspective_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                             const double M[9], float *dst, int dh, int dw, ptrdiff_t dst_stride_el);

/* ---- image-grid distortions: grid -> dense map, and the fused grid remap ------------
 * ImageGrid.generate_remap_params  grid_rendering/type.py:209-261 with get_inv_trans_mat
 * (:182-197, cv.getPerspectiveTransform DECOMP_SVD) and the per-cell cv.fillPoly raster
 * (:199-207, element/polygon.py:70-77).  Vertices: int32 [rows, cols, 2] as (x, y), the ROUNDED
 * grid points of the source / destination ImageGrid.  Unfilled pixels map to (0, 0).
 * owner (optional): int32 [dh, dw], 1 + row-major index of the cell that wrote the pixel. */
int vkx_grid_to_map_dev(vkx_ctx *ctx, const int32_t *src_vertices, const int32_t *dst_vertices, int rows,
                        int cols, int dh, int dw, float *map_x, float *map_y, ptrdiff_t map_stride_el,
                        int32_t *owner);
int vkx_grid_to_map(vkx_ctx *ctx, const int32_t *src_vertices, const int32_t *dst_vertices, int rows,
                    int cols, int dh, int dw, float *map_x, float *map_y, ptrdiff_t map_stride_el,
                    int32_t *owner);

/* Shared-grid multi-element remap (blend_src_to_dst_{image,mask,score_map},
 * grid_blender.py:54-81 through one DistortionStateImageGridBased): the dense map is never
 * written to memory; every element is gathered in the same pass. */
typedef struct vkx_elem {
    const void *src;      /* uint8 [sh, sw, cn] or float32 [sh, sw] */
    void *dst;            /* same type, [dh, dw(, cn)] */
    ptrdiff_t src_stride; /* bytes (uint8) / elements (float32) */
    ptrdiff_t dst_stride;
    int32_t cn;           /* 1, 3, 4 for uint8; 1 for float32 */
    int32_t is_f32;       /* 0: uint8, 1: float32 */
} vkx_elem;
int vkx_grid_remap_dev(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw,
                       const int32_t *src_vertices, const int32_t *dst_vertices, int rows, int cols,
                       int dh, int dw);
/* The same elements through an explicit dense map shared by all of them (cv.remap per element with one
 * (map_x, map_y) pair, grid_blender.py:54-81 when the caller keeps the map of generate_remap_params). */
int vkx_remap_multi_dev(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw, const float *map_x,
                        const float *map_y, ptrdiff_t map_stride_el, int dh, int dw);
int vkx_remap_multi(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw, const float *map_x,
                    const float *map_y, ptrdiff_t map_stride_el, int dh, int dw);
int vkx_grid_remap(vkx_ctx *ctx, const vkx_elem *elems, int n_elems, int sh, int sw,
                   const int32_t *src_vertices, const int32_t *dst_vertices, int rows, int cols,
                   int dh, int dw);

/* ---- photometric members ------------------------------------------------------------ */
/* cv.GaussianBlur(mat, (k, k), sigma), BORDER_REFLECT_101  photometric/blur.py:54-69 */
int vkx_gaussian_blur_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                             int ksize, double sigma, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_gaussian_blur_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                         int ksize, double sigma, uint8_t *dst, ptrdiff_t dst_stride);
/* color_shift on RGB: cvtColor RGB2HSV_FULL, H = (H + delta) mod 256, HSV2RGB_FULL
 *   photometric/color.py:93-116, element/image.py:188-202,771-814.  In place allowed. */
int vkx_color_shift_rgb_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta,
                            uint8_t *dst, ptrdiff_t dst_stride);
int vkx_color_shift_rgb(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta,
                        uint8_t *dst, ptrdiff_t dst_stride);
/* cv.cvtColor(.., COLOR_RGB2HSV_FULL / COLOR_HSV2RGB_FULL)  element/image.py:794-808 */
int vkx_cvt_rgb_hsv_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int to_hsv,
                           uint8_t *dst, ptrdiff_t dst_stride);
int vkx_cvt_rgb_hsv_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int to_hsv,
                       uint8_t *dst, ptrdiff_t dst_stride);
/* _mean_shift  photometric/color.py:32-55 + photometric/opt.py:41-57.  channel_mask 0 = all. */
int vkx_mean_shift_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                          int delta, int has_threshold, int threshold, int cycle, unsigned channel_mask,
                          uint8_t *dst, ptrdiff_t dst_stride);
int vkx_mean_shift_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                      int delta, int has_threshold, int threshold, int cycle, unsigned channel_mask,
                      uint8_t *dst, ptrdiff_t dst_stride);
/* gaussion_noise_image tail: clip(int16(px) + noise, 0, 255)  photometric/noise.py:51-53.
 * noise: int16 [h, w, cn] = round(rng.normal(0, std, shape)), C order. */
int vkx_add_noise_i16_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                          const int16_t *noise, ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_add_noise_i16(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                      const int16_t *noise, ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride);
/* line_streak_image  photometric/streak.py:56-99 (+ :24-41), in place. */
int vkx_line_streak_u8_dev(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int thickness,
                           int gap, int dash_thickness, int dash_gap, const uint8_t color[4], double alpha,
                           int enable_vert, int enable_hori);
int vkx_line_streak_u8(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int thickness,
                       int gap, int dash_thickness, int dash_gap, const uint8_t color[4], double alpha,
                       int enable_vert, int enable_hori);

/* ellipse_streak  photometric/streak.py:283-337.  vkx_ellipse_mask_u8: the loop of
 * cv.ellipse(mask, (cx, cy), axes[i], angle 0, arc 0..360, color 1, thickness) calls (:312-324; LINE_8, shift 0) drawn
 * onto `mask` (uint8 [h, w]; touched pixels become 1, the others keep their value).  axes_host: HOST int32
 * [n_ellipses, 2] as (box.width // 2, box.height // 2).  thickness in 1 .. 32767 (filled ellipses are not on the path).
 * vkx_ellipse_streak_u8: that mask on a cleared plane, then Mask.fill_image(image, color, alpha) in place. */
int vkx_ellipse_mask_u8_dev(vkx_ctx *ctx, uint8_t *mask, ptrdiff_t mask_stride, int h, int w, int cx, int cy,
                            const int32_t *axes_host, int n_ellipses, int thickness);
int vkx_ellipse_mask_u8(vkx_ctx *ctx, uint8_t *mask, ptrdiff_t mask_stride, int h, int w, int cx, int cy,
                        const int32_t *axes_host, int n_ellipses, int thickness);
int vkx_ellipse_streak_u8_dev(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int cx, int cy,
                              const int32_t *axes_host, int n_ellipses, int thickness, const uint8_t color[4],
                              double alpha);
int vkx_ellipse_streak_u8(vkx_ctx *ctx, uint8_t *img, int h, int w, int cn, ptrdiff_t stride, int cx, int cy,
                          const int32_t *axes_host, int n_ellipses, int thickness, const uint8_t color[4],
                          double alpha);

/* cv.cvtColor on uint8 images, the codes Image.to_target_mode_image uses (element/image.py:188-202,771-814).
 * HSL images of the reference are HLS with the last two channels swapped on the host. */
#define VKX_CVT_RGB2HSV_FULL 0
#define VKX_CVT_HSV2RGB_FULL 1
#define VKX_CVT_RGB2HLS_FULL 2
#define VKX_CVT_HLS2RGB_FULL 3
#define VKX_CVT_RGB2GRAY 4   /* [h, w, 3] -> [h, w]; not in place */
#define VKX_CVT_GRAY2RGB 5   /* [h, w] -> [h, w, 3]; not in place */
#define VKX_CVT_RGBA2RGB 6   /* [h, w, 4] -> [h, w, 3]: alpha dropped; not in place */
#define VKX_CVT_RGB2RGBA 7   /* [h, w, 3] -> [h, w, 4]: alpha 255 */
#define VKX_CVT_GRAY2RGBA 8  /* [h, w] -> [h, w, 4] */
#define VKX_CVT_RGBA2GRAY 9  /* [h, w, 4] -> [h, w] */
int vkx_cvt_color_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int code,
                         uint8_t *dst, ptrdiff_t dst_stride);
int vkx_cvt_color_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int code,
                     uint8_t *dst, ptrdiff_t dst_stride);
/* color_balance on any image mode  photometric/color.py:371-396: dst = uint8(clip(w0 * a + w1 * b, 0, 255)) on the channels of
 * channel_mask (0 = all), b elsewhere; a = the grey version of the image in the image's own mode, b = the image, w0 = 1 - ratio,
 * w1 = ratio (float32 products rounded separately, truncation). */
int vkx_blend_u8_dev(vkx_ctx *ctx, const uint8_t *a, ptrdiff_t a_stride, const uint8_t *b, ptrdiff_t b_stride, int h, int w,
                     int cn, double w0, double w1, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_blend_u8(vkx_ctx *ctx, const uint8_t *a, ptrdiff_t a_stride, const uint8_t *b, ptrdiff_t b_stride, int h, int w, int cn,
                 double w0, double w1, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride);
/* fog with fractional fog values (GRAYSCALE images)  photometric/effect.py:192-205: dst = uint8(clip((1 - m) * px + m * fog[c]))
 * with the float32 weight plane m [h, w]; `fog`: cn float32 values on the host. */
int vkx_fog_f32_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const float *weight,
                       ptrdiff_t weight_stride_el, const float *fog, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_fog_f32_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const float *weight,
                   ptrdiff_t weight_stride_el, const float *fog, uint8_t *dst, ptrdiff_t dst_stride);
/* brightness_shift on RGB with the default HSL intermediate  photometric/color.py:125-160: RGB2HLS_FULL,
 * L = clip(L + delta), HLS2RGB_FULL, one pass.  In place allowed. */
int vkx_brightness_shift_rgb_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta,
                                 uint8_t *dst, ptrdiff_t dst_stride);
int vkx_brightness_shift_rgb(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, int delta,
                             uint8_t *dst, ptrdiff_t dst_stride);
/* color_balance on RGB  photometric/color.py:364-397: uint8(clip(fl32(1 - ratio) * gray + fl32(ratio) * px)),
 * gray = RGB2GRAY of the pixel.  In place allowed. */
int vkx_color_balance_rgb_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, double ratio,
                              uint8_t *dst, ptrdiff_t dst_stride);
int vkx_color_balance_rgb(vkx_ctx *ctx, const uint8_t *src, int h, int w, ptrdiff_t src_stride, double ratio,
                          uint8_t *dst, ptrdiff_t dst_stride);

/* complement / posterization / channel_permutation  photometric/color.py:299-357, 423-432 (numpy only).
 *   VKX_POINT_COMPLEMENT  p0 = threshold or -1 (none), p1 = enable_threshold_lte: v = 255 - v where selected
 *   VKX_POINT_POSTERIZE   p0 = num_bits in [0, 7]: v &= (0xFF >> p0) << p0
 *   VKX_POINT_PERMUTE     p0 = permutation, 2 bits per output channel: out[c] = in[(p0 >> 2c) & 3]; not in place
 * channel_mask 0 = all channels (ignored by PERMUTE). */
#define VKX_POINT_COMPLEMENT 0
#define VKX_POINT_POSTERIZE 1
#define VKX_POINT_PERMUTE 2
int vkx_pointwise_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int op, int p0,
                         int p1, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_pointwise_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int op, int p0,
                     int p1, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride);

/* The two halves of boundary_equalization / histogram_equalization  photometric/color.py:214-285:
 *   vkx_histogram_u8  per-channel histogram int32 [cn][256] (exact integer reduction; device pointer for *_dev)
 *   vkx_apply_lut_u8  dst[c] = lut[c][src[c]] on the channels of channel_mask (0 = all); lut is a HOST uint8 [cn][256]
 * Between them the host turns the histogram into the table: min / max and the float32 scale for the boundary
 * equalisation (numpy arithmetic of :222-243 per distinct value), cv.equalizeHist's cumulative table for the other. */
int vkx_histogram_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int32_t *hist);
int vkx_histogram_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, int32_t *hist);
/* std_shift's mean (photometric/color.py:165-210): sums_host [n_sel] = the float32 sums numpy forms inside np.mean of the float32
 * copy, value for value, including the roundings once a running sum has left the exactly representable range (2^24):
 *   sequential = 0  np.mean(mat) of a single plane, and the channels picked with mat[:, :, channels] (numpy lays that copy out
 *                   channel first): per channel, pieces of 8 192 contiguous elements summed exactly, accumulated in float32 one
 *                   after the other;
 *   sequential = 1  np.mean(mat.reshape(-1, C), axis=0) of ALL C >= 2 channels of an interleaved image: a sequential float32
 *                   accumulation over the pixels, per channel.
 * The mean is sums / float32(h * w).  h * w <= 2^22.  Synchronous (the caller builds its tables from the values); *_dev: src is a
 * device plane. */
int vkx_sum_f32_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const int32_t *channels_host,
                       int n_sel, int sequential, float *sums_host);
int vkx_sum_f32_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride, const int32_t *channels_host,
                   int n_sel, int sequential, float *sums_host);
int vkx_apply_lut_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                         const uint8_t *lut_host, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride);
/* The same look-up for up to 8 dense single-channel device planes, each with its own HOST table uint8 [256], in one launch (the four masks
 * PageResizingStep binarises before and after their resize: Mask.to_resized_mask, element/mask.py:454-479). */
typedef struct vkx_lut_plane {
    const uint8_t *src;   /* device, dense, 16-byte aligned */
    uint8_t *dst;         /* device, dense, 16-byte aligned (may equal src) */
    size_t n_bytes;
    const uint8_t *lut_host;
} vkx_lut_plane;
int vkx_apply_lut_u8_planes_dev(vkx_ctx *ctx, const vkx_lut_plane *planes, int n_planes);
int vkx_apply_lut_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                     const uint8_t *lut_host, unsigned channel_mask, uint8_t *dst, ptrdiff_t dst_stride);

/* mat[pos_y, pos_x] -- numpy advanced indexing with two int32 index planes [dh, dw]: the pixel shuffle of
 * glass_blur  photometric/blur.py:204-250 (the planes come from the caller's numpy Generator stream).  Entries outside
 * the source are an error (VKX_ERR_INVALID). */
int vkx_gather_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                      const int32_t *pos_y, const int32_t *pos_x, ptrdiff_t pos_stride_el, uint8_t *dst, int dh, int dw,
                      ptrdiff_t dst_stride);
int vkx_gather_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                  const int32_t *pos_y, const int32_t *pos_x, ptrdiff_t pos_stride_el, uint8_t *dst, int dh, int dw,
                  ptrdiff_t dst_stride);

/* poisson_noise  photometric/noise.py:81-91: np.clip(samples, 0, 255).astype(uint8) of the int64 rng.poisson draws
 * (the draws themselves are the caller's numpy Generator stream; n values, flat). */
int vkx_saturate_i64_u8_dev(vkx_ctx *ctx, const int64_t *src, size_t n, uint8_t *dst);
int vkx_saturate_i64_u8(vkx_ctx *ctx, const int64_t *src, size_t n, uint8_t *dst);

/* impulse_noise  photometric/noise.py:125-150: selector uint8 [h, w] (0 keep, 1 salt = 255, 2 pepper = 0 on every
 * channel of the pixel), drawn by the caller's numpy Generator (rng.choice). */
int vkx_impulse_noise_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                             const uint8_t *selector, ptrdiff_t selector_stride, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_impulse_noise_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                         const uint8_t *selector, ptrdiff_t selector_stride, uint8_t *dst, ptrdiff_t dst_stride);

/* speckle_noise  photometric/noise.py:172-183: uint8(clip(px + px * noise, 0, 255)) evaluated in float64, noise
 * float64 [h, w, cn] = rng.normal(0, std, shape) of the caller's numpy Generator. */
int vkx_speckle_noise_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                             const double *noise, ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_speckle_noise_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                         const double *noise, ptrdiff_t noise_stride_el, uint8_t *dst, ptrdiff_t dst_stride);

/* ---- alpha composite ----------------------------------------------------------------
 * fill_np_array  element/opt.py:118-209 reached through Box.fill_np_array element/box.py:311-340
 * (Box.fill_image :394-416, Mask.fill_image element/mask.py:601-612, ScoreMap.fill_image
 * element/score_map.py:678-687); layer order = PageAssemblerStep.run
 * pipeline/text_detection/page_assembler.py:155-236.  Layers are applied in order, in place.
 * All plane pointers of a layer live in the same memory space as dst (device for *_dev). */
typedef struct vkx_layer {
    int32_t up, left, height, width; /* box inside dst */
    const uint8_t *mask;             /* optional [height, width] uint8: selected where > 0 */
    ptrdiff_t mask_stride;
    const float *alpha;              /* optional [height, width] float32; selects alpha > 0 if !mask */
    ptrdiff_t alpha_stride_el;
    double alpha_scalar;             /* used when alpha == NULL; must be in [0, 1] */
    const uint8_t *value;            /* optional [height, width, cn] uint8 */
    ptrdiff_t value_stride;
    uint8_t value_const[4];          /* used when value == NULL */
    int32_t mode;                    /* VKX_FILL_*: keep_max_value / keep_min_value, element/opt.py:150-158 */
} vkx_layer;
#define VKX_FILL_PLAIN 0
#define VKX_FILL_KEEP_MAX 1 /* write only where dst < value; acts when alpha is the scalar 1.0, like the reference */
#define VKX_FILL_KEEP_MIN 2 /* write only where dst > value */
int vkx_fill_u8_dev(vkx_ctx *ctx, uint8_t *dst, int h, int w, int cn, ptrdiff_t dst_stride,
                    const vkx_layer *layers, int n_layers);
int vkx_fill_u8(vkx_ctx *ctx, uint8_t *dst, int h, int w, int cn, ptrdiff_t dst_stride,
                const vkx_layer *layers, int n_layers);
/* device page, HOST layer planes (page_assembler.py:155-236 onto a device-resident page): the planes are staged for the
 * call -- gathered in the context's page-locked ring and read THERE by the composite kernel (every plane byte is read once: no
 * copy to device memory, no copy dispatch) --, the page stays where it is; asynchronous.  A plane whose VKX_LAYER_*_ON_DEVICE bit
 * is or-ed into `mode` already lives in device memory and is used where it is (fill_page_inactive_region: a device mask selects,
 * the host bottom layer is the value). */
#define VKX_LAYER_MASK_ON_DEVICE 0x100
#define VKX_LAYER_ALPHA_ON_DEVICE 0x200
#define VKX_LAYER_VALUE_ON_DEVICE 0x400
int vkx_fill_u8_dev_host_layers(vkx_ctx *ctx, uint8_t *dst_dev, int h, int w, int cn, ptrdiff_t dst_stride,
                                const vkx_layer *layers_host_planes, int n_layers);
/* The layer lists of n_pages equally shaped device destinations in ONE launch (a batch of pages assembled together,
 * PageAssemblerStep.run per page of the batch): page p takes layers[layer_begin[p] .. layer_begin[p + 1]) in order, with the
 * pixels vkx_fill_u8_dev(dsts[p], ...) would produce.  dsts_host: HOST array of n_pages device pointers; layer_begin_host:
 * HOST int32 [n_pages + 1], layer_begin[0] == 0. */
int vkx_fill_u8_batch_dev(vkx_ctx *ctx, uint8_t *const *dsts_host, int n_pages, int h, int w, int cn, ptrdiff_t dst_stride,
                          const vkx_layer *layers, const int32_t *layer_begin_host);

/* float32 destinations: ScoreMap fills (Box.fill_score_map element/box.py:368-392, Mask.fill_score_map
 * element/mask.py:575-599, Polygon.fill_score_map element/polygon.py:475-487), the label height maps of
 * pipeline/text_detection/page_distortion.py:163-314. */
typedef struct vkx_layer_f32 {
    int32_t up, left, height, width;
    const uint8_t *mask;
    ptrdiff_t mask_stride;
    const float *alpha;
    ptrdiff_t alpha_stride_el;
    double alpha_scalar;
    const float *value;              /* optional [height, width] float32 */
    ptrdiff_t value_stride_el;
    float value_const;
    int32_t mode;
} vkx_layer_f32;
int vkx_fill_f32_dev(vkx_ctx *ctx, float *dst, int h, int w, ptrdiff_t dst_stride_el,
                     const vkx_layer_f32 *layers, int n_layers);
int vkx_fill_f32(vkx_ctx *ctx, float *dst, int h, int w, ptrdiff_t dst_stride_el,
                 const vkx_layer_f32 *layers, int n_layers);

/* ---- bicubic resize ------------------------------------------------------------------
 * cv.resize(mat, (dw, dh), interpolation=cv.INTER_CUBIC): Image.to_resized_image element/image.py:836-852
 * (bottom layer of fill_page_inactive_region, pipeline/text_detection/page_distortion.py:146-161),
 * Mask.to_resized_mask element/mask.py:454-479 (on the 0/255 plane), ScoreMap.to_resized_score_map
 * element/score_map.py:616-640.  Keys cubic A = -0.75, replicated border, 11-bit fixed-point coefficients and
 * (sum + 2^21) >> 22 rounding for uint8 (OpenCV's scalar path); float32 sums left to right. */
int vkx_resize_cubic_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                            uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride);
int vkx_resize_cubic_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                             float *dst, int dh, int dw, ptrdiff_t dst_stride_el);
int vkx_resize_cubic_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                        uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride);
int vkx_resize_cubic_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el,
                         float *dst, int dh, int dw, ptrdiff_t dst_stride_el);

/* cv.resize with the interpolation codes of cv2 (NEAREST 0, LINEAR 1, CUBIC 2) on uint8: pixelation
 * photometric/effect.py:61-79 shrinks with INTER_LINEAR and grows back with INTER_NEAREST.  LINEAR: 11-bit
 * coefficients, OpenCV's two-stage vertical rounding, an exact 2 x 2 shrink is the INTER_AREA box. */
#define VKX_INTER_NEAREST 0
#define VKX_INTER_LINEAR 1
#define VKX_INTER_CUBIC 2
/* ... and the interpolations PageResizingStep samples for the page Image, Masks and ScoreMaps
 * (pipeline/text_detection/page_resizing.py:110-181, sample_cv_resize_interpolation utility/opt.py:125-148):
 * LANCZOS4 (8 x 8 taps, 11-bit coefficients / float32), LINEAR_EXACT (8.8 fixed point on uint8; float32 has no
 * bit-exact path in cv.resize and falls back to LINEAR), NEAREST_EXACT (16.16 index steps), AREA (shrinking only: box
 * sums for integer factors, fractional cell weights in float32 otherwise). */
#define VKX_INTER_AREA 3
#define VKX_INTER_LANCZOS4 4
#define VKX_INTER_LINEAR_EXACT 5
#define VKX_INTER_NEAREST_EXACT 6
int vkx_resize_u8_dev(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                      uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride, int interpolation);
int vkx_resize_u8(vkx_ctx *ctx, const uint8_t *src, int sh, int sw, int cn, ptrdiff_t src_stride,
                  uint8_t *dst, int dh, int dw, ptrdiff_t dst_stride, int interpolation);

/* cv.filter2D(image, -1, kernel) on uint8 with a float32 kernel of up to 15 x 15 taps (HOST pointer, row-major):
 * defocus_blur / motion_blur  photometric/blur.py:85-192.  Correlation anchored at the kernel centre,
 * BORDER_REFLECT_101, the non-zero taps in row-major order accumulated in float32 (products and sums rounded
 * separately), cvRound + saturate. */
int vkx_filter2d_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                        const float *kernel_host, int kh, int kw, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_filter2d_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                    const float *kernel_host, int kh, int kw, uint8_t *dst, ptrdiff_t dst_stride);

/* the same on a float32 plane (ScoreMap.to_resized_score_map element/score_map.py:616-640); strides in elements */
int vkx_resize_f32_dev(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el, float *dst, int dh, int dw,
                       ptrdiff_t dst_stride_el, int interpolation);
int vkx_resize_f32(vkx_ctx *ctx, const float *src, int sh, int sw, ptrdiff_t src_stride_el, float *dst, int dh, int dw,
                   ptrdiff_t dst_stride_el, int interpolation);

/* zoom_in_blur  photometric/blur.py:264-316: the image plus the centred crops of its INTER_CUBIC enlargements to the
 * sizes of sizes_hw_host (HOST int32 [n, 2] as (height, width), every size >= the image) are summed in uint16, then
 * dst = uint8(clip((1 - alpha) * px + alpha * rint(sum / (n + 1)))) in float64. */
int vkx_zoom_in_blur_u8_dev(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                            const int32_t *sizes_hw_host, int n_sizes, double alpha, uint8_t *dst, ptrdiff_t dst_stride);
int vkx_zoom_in_blur_u8(vkx_ctx *ctx, const uint8_t *src, int h, int w, int cn, ptrdiff_t src_stride,
                        const int32_t *sizes_hw_host, int n_sizes, double alpha, uint8_t *dst, ptrdiff_t dst_stride);

/* ---- point projection through the lattice ------------------------------------------------
 * FuncImageGridBased.func_point, grid_rendering/interface.py:194-216, for a batch of points (the reference loops
 * point by point, distortion/interface.py:638-661): cell = (y // grid_size, x // grid_size) of the ROUNDED point,
 * trans_mat = cv.getPerspectiveTransform(src quad, dst quad) of that cell (grid_rendering/type.py:166-180),
 * (x', y') = trans_mat . (smooth_x, smooth_y, 1) dehomogenised, all in float64.  Everything is HOST memory:
 * src_vertices / dst_vertices int32 [rows, cols, 2] (x, y); pts_xy int32 [n, 2] rounded (x, y); pts_smooth_xy
 * float64 [n, 2]; out_xy float64 [n, 2].  A point whose cell index falls outside the (rows-1) x (cols-1) cells is an
 * error (VKX_ERR_OUT_OF_LATTICE; the reference raises IndexError there). */
int vkx_grid_project_points(vkx_ctx *ctx, const int32_t *src_vertices, const int32_t *dst_vertices, int rows, int cols,
                            int grid_size, const int32_t *pts_xy, const double *pts_smooth_xy, int n, double *out_xy);

/* ---- similarity_mls lattice construction --------------------------------------------------
 * SimilarityMlsPointProjector.project_point, geometric/mls.py:38-135, for every vertex of the source lattice in one
 * launch (the reference projects vertex by vertex through create_dst_image_grid_and_shift_amounts_and_resize_ratios,
 * grid_rendering/grid_creator.py:44-115).  float32 arithmetic in the reference's order of operations, including the
 * accumulation orders of numpy's reductions and of np.matmul (csrc/mls.hip lists them).
 *   src_handles / dst_handles                float32 [n_handles, 2] (x, y): the INTEGER handle positions
 *                                            (PointTuple.to_smooth_np_array, element/point.py:251-252)
 *   src_handles_smooth / dst_handles_smooth  float64 [n_handles, 2]: a vertex exactly on a source handle maps to that
 *                                            handle's target (mls.py:57-61; the last duplicate wins like the dict)
 *   vertices_xy, out_xy                      float64 [n_vertices, 2]
 * Any number of handles (tables beyond 2048 handles are read through the caches instead of LDS).  Note that the reference's
 * own result depends on the BLAS kernels of its host for the weighted centroids (sgemv): this entry point follows numpy 2.2 +
 * OpenBLAS 0.3.29 on an AVX-512 host, the machine tests/golden was generated on.  VKX_ERR_DIVIDE where the reference's np.errstate(divide='raise') fires (a vertex on an integer
 * handle position that is not an exact handle hit).  _dev: device pointers, `status` int32 (zeroed by the caller)
 * receives 1 + the index of such a vertex. */
int vkx_mls_project_dev(vkx_ctx *ctx, const float *src_handles, const float *dst_handles,
                        const double *src_handles_smooth, const double *dst_handles_smooth, int n_handles,
                        const double *vertices_xy, int n_vertices, double *out_xy, int32_t *status);
int vkx_mls_project(vkx_ctx *ctx, const float *src_handles, const float *dst_handles,
                    const double *src_handles_smooth, const double *dst_handles_smooth, int n_handles,
                    const double *vertices_xy, int n_vertices, double *out_xy);

/* ---- polygon rasterisation -----------------------------------------------------------
 * cv.fillPoly(zeros((h, w), uint8), [pts], 1): PolygonInternals.np_mask element/polygon.py:70-77
 * (Bresenham LINE_8 outline + even-odd scanline spans).  pts: HOST int32 [npts, 2] as (x, y), all
 * inside the mask.  *_dev: mask is a device plane that is OR-ed into; host variant: mask is
 * overwritten with the 0/1 raster. */
int vkx_fill_poly_mask_u8_dev(vkx_ctx *ctx, const int32_t *pts_host, int npts, uint8_t *mask, int h, int w,
                              ptrdiff_t mask_stride);
int vkx_fill_poly_mask_u8(vkx_ctx *ctx, const int32_t *pts_host, int npts, uint8_t *mask, int h, int w,
                          ptrdiff_t mask_stride);

/* Ordered paint of many polygons into one label plane pair -- the label rasterisation after distortion,
 * pipeline/text_detection/page_distortion.py:163-314 (for polygon in order: polygon.fill_mask(mask) /
 * polygon.fill_score_map(score_map, value)), element/polygon.py:458-487.  Polygon i covers the cv.fillPoly
 * raster of its vertices; where polygons overlap the LAST one in the list wins, exactly like the sequential
 * fills.  Pixels inside any polygon get mask = 1 and score = values[winner]; other pixels are untouched.
 * pts_host: HOST int32 [total_pts, 2] (x, y) in plane coordinates (parts outside the plane are clipped);
 * poly_offsets_host: HOST int32 [n_polys + 1]; values_host: HOST float32 [n_polys] (required with score).
 * mask / score: either may be NULL.  A polygon with more than 64 crossings on one scanline is refused
 * (VKX_ERR_UNSUPPORTED; paint such polygons one by one through vkx_fill_poly_mask_u8 + vkx_fill_*).
 * _dev: asynchronous on the ctx stream when no polygon has more than 64 vertices (none can then exceed the crossing limit: the
 * tables travel through the context's page-locked ring); with larger polygons the call waits for the overflow flag. */
int vkx_paint_polys_dev(vkx_ctx *ctx, const int32_t *pts_host, const int32_t *poly_offsets_host, int n_polys,
                        const float *values_host, uint8_t *mask, ptrdiff_t mask_stride, float *score,
                        ptrdiff_t score_stride_el, int h, int w);
int vkx_paint_polys(vkx_ctx *ctx, const int32_t *pts_host, const int32_t *poly_offsets_host, int n_polys,
                    const float *values_host, uint8_t *mask, ptrdiff_t mask_stride, float *score,
                    ptrdiff_t score_stride_el, int h, int w);
/* The same paint into FRESH planes: mask / score are uninitialised device memory and every pixel of them is written (mask = 0 /
 * score = 0 outside every polygon) -- what the label rasterisation needs (it starts from np.zeros planes, page_distortion.py:201,
 * :283) without a memset dispatch per plane. */
int vkx_paint_polys_fresh_dev(vkx_ctx *ctx, const int32_t *pts_host, const int32_t *poly_offsets_host, int n_polys,
                              const float *values_host, uint8_t *mask, ptrdiff_t mask_stride, float *score,
                              ptrdiff_t score_stride_el, int h, int w);
/* ... and the label plane sets of ONE page in one call (page_distortion.py:163-314 paints four: text-line mask + height map, char
 * mask, seal-impression char mask, char height map): up to 8 sets of one plane shape, each with its own polygon list, values and
 * fresh output planes; the same pixels as n_sets vkx_paint_polys_fresh_dev calls, with one table copy and three kernels for all. */
typedef struct vkx_paint_set {
    const int32_t *pts_host;          /* HOST int32 [total_pts, 2] (x, y) */
    const int32_t *poly_offsets_host; /* HOST int32 [n_polys + 1] */
    int32_t n_polys;
    const float *values_host;         /* HOST float32 [n_polys]; required with score */
    uint8_t *mask;                    /* device [h, w] or NULL */
    ptrdiff_t mask_stride;
    float *score;                     /* device [h, w] or NULL */
    ptrdiff_t score_stride_el;
} vkx_paint_set;
int vkx_paint_poly_sets_fresh_dev(vkx_ctx *ctx, const vkx_paint_set *sets, int n_sets, int h, int w);

/* ---- the numpy Generator streams of the noise operators, drawn on the device -------------------------------------
 * photometric/noise.py:44-54 (gaussion_noise), :160-190 (speckle_noise), :100-157 (impulse_noise) draw from the
 * caller's numpy Generator (PCG64).  A job = one stream: the generator's 128-bit state and increment
 * (rng.bit_generator.state['state']) and the number of samples.  The device produces the SAME values numpy would
 * (PCG64 jump-ahead + the 256-layer ziggurat with its variable number of raw draws per sample resolved by a two-pass
 * scan), and reports how many raw 64-bit draws the call consumed, so that the caller can advance its generator
 * (state' = pcg64 jump by `draws`).  Kinds:
 *   VKX_NP_NORMAL_I16     dst int16 [n]  = np.round(rng.normal(0, scale, n)).astype(int16)
 *   VKX_NP_NORMAL_ADD_U8  dst uint8 [n]  = clip(int16(src) + the plane above, 0, 255)          (gaussion_noise, fused)
 *   VKX_NP_SPECKLE_U8     dst uint8 [n]  = uint8(clip(src + src * rng.normal(0, scale, n), 0, 255)) in float64
 *   VKX_NP_CHOICE3_U8     dst uint8 [n]  = rng.choice((0, 1, 2), n, p) as #{k : cdf[k] <= rng.random()}, cdf = cumsum(p) / sum
 *   VKX_NP_IMPULSE_U8     dst uint8 [n, cn] = src with salt (255) where the selector is 1, pepper (0) where it is 2
 *   VKX_NP_NORMAL_TILES   dst = a TILE BUFFER of vkx_np_tiles_layout(n) bytes (256-byte aligned) standing for the int16 plane of
 *                         VKX_NP_NORMAL_I16 without ever forming it: the generator works in tiles of 3072 raw draws, and a tile's
 *                         samples stay where the draw pass left them, squeezed together in the tile's slot; a table gives the
 *                         index of every tile's first sample.  vkx_chain_item.noise takes such a buffer (noise_tiled = 1): the
 *                         chain kernel looks its samples up in the slots, so gaussion_noise costs one 2-byte read per sample
 *                         instead of a placement pass (2 bytes per raw draw read, 2 bytes per sample written) plus that read.
 *                         vkx_np_tiles_expand_dev turns a buffer into the plane it stands for.
 * src / dst are dense device arrays (vkx_np_draw_batch_dev: asynchronous on the ctx stream, results_host is valid after a
 * synchronisation).  exp() / log1p() of the device differ from glibc's in the last bits: a decision or an
 * emitted integer that could depend on them sets VKX_NP_AMBIGUOUS in the job's result (the caller then redraws that plane
 * with numpy itself; expected < 1e-6 per 2048^2 plane); VKX_NP_SHORT = the provisioned raw draws did not yield n samples
 * (same remedy; never observed). */
#define VKX_NP_NORMAL_I16 0
#define VKX_NP_NORMAL_ADD_U8 1
#define VKX_NP_SPECKLE_U8 2
#define VKX_NP_CHOICE3_U8 3
#define VKX_NP_IMPULSE_U8 4
#define VKX_NP_NORMAL_TILES 5
#define VKX_NP_DEBUG_WIDE_MARGIN 0x100 /* or-ed into kind: every wedge test counts as ambiguous (exercises the fallback in tests) */
#define VKX_NP_AMBIGUOUS 1u
#define VKX_NP_SHORT 2u
typedef struct vkx_np_job {
    uint64_t state[2];   /* PCG64 state, low / high word */
    uint64_t inc[2];     /* PCG64 increment, low / high word */
    int64_t n;           /* samples (normal kinds) or elements (uniform kinds) */
    int32_t kind, cn;
    double scale;        /* std of the normal kinds */
    double cdf[3];       /* uniform kinds */
    const void *src;     /* device */
    void *dst;           /* device */
} vkx_np_job;
typedef struct vkx_np_result {
    unsigned long long draws;    /* raw 64-bit draws consumed */
    unsigned long long samples;  /* samples the provisioned draws yielded (>= n unless VKX_NP_SHORT) */
    uint32_t flags, reserved;
} vkx_np_result;
int vkx_np_draw_batch_dev(vkx_ctx *ctx, const vkx_np_job *jobs_host, int n_jobs, vkx_np_result *results_host);
/* Tile buffer of a VKX_NP_NORMAL_TILES job of n samples: `bytes` in all; a 16-byte header (uint32 tiles, uint32 slot elements,
 * uint64 samples the tiles hold), at table_offset [n_tiles + 1] x (uint32 index of the tile's first sample, uint32 first valid
 * element of its slot), at slots_offset n_tiles slots of slot_elems int16.  Sample i of the plane, with
 * table[t].first <= i < table[t + 1].first, is slot t's element table[t].skip + i - table[t].first; samples i + 1 and i + 2 follow
 * it in the same slot (a slot ends with the first two samples of its successor).  Any output pointer may be NULL. */
int vkx_np_tiles_layout(int64_t n, int64_t *n_tiles, int64_t *slot_elems, int64_t *table_offset, int64_t *slots_offset, int64_t *bytes);
/* dst int16 [n] (device, 8-byte aligned) = the plane a finished tile buffer stands for; asynchronous on the ctx stream */
int vkx_np_tiles_expand_dev(vkx_ctx *ctx, const void *tiles, int64_t n, int16_t *dst);
/* one job whose src / dst are HOST arrays; synchronous */
int vkx_np_draw(vkx_ctx *ctx, const vkx_np_job *job, vkx_np_result *result_host);

/* rng.poisson(lam) with lam = the bytes of a uint8 array (vkit poisson_noise, photometric/noise.py:81-90: `rng.poisson(mat as float32)`
 * + clip to uint8), value for value numpy 2.2.6's random_poisson per element in C order (distributions.c: nothing for lam 0, the
 * multiplication method below 10, PTRS from 10 up), drawn on the device from the PCG64 stream (state, inc) although every element takes
 * a data-dependent number of draws: windows of candidate stream positions per block of 32 elements, resolved superblock by superblock
 * (vkit_amd/csrc/poisson.hip).  dst uint8 [n] = min(sample, 255); *consumed_host = the raw 64-bit draws numpy would have made.
 * *flags_host != 0: the result is NOT to be used and the stream not to be moved (the caller draws with numpy on the host): a PTRS
 * comparison fell within 2e-13 of equality (device log vs glibc log), a start left its 6-sigma window, or the stream would need more than 2^29 raw draws.
 * Synchronous.  _dev: device arrays, src 16-byte aligned. */
#define VKX_NP_POISSON_AMBIGUOUS 1u
#define VKX_NP_POISSON_TABLE 2u     /* not raised any more: k + 1 beyond the loggam table takes random_loggam with the device log */
#define VKX_NP_POISSON_WINDOW 4u
#define VKX_NP_POISSON_MISMATCH 8u
#define VKX_NP_POISSON_DRAWS 16u
#define VKX_NP_POISSON_SIZE 32u   /* the stream would take more than 2^29 raw draws (4 GB of scratch) */
int vkx_np_poisson_u8_dev(vkx_ctx *ctx, const uint64_t *state, const uint64_t *inc, const uint8_t *src, long long n, uint8_t *dst,
                          long long *consumed_host, unsigned *flags_host);
int vkx_np_poisson_u8(vkx_ctx *ctx, const uint64_t *state, const uint64_t *inc, const uint8_t *src_host, long long n, uint8_t *dst_host,
                      long long *consumed_host, unsigned *flags_host);
/* out[x - 1] = numpy's random_loggam(x), x = 1 .. n, as the library tabulates it on the host (tests compare it with libnpyrandom.a) */
int vkx_np_poisson_loggam_table(double *out, int n);

/* The fog density field of `fog` (reference photometric/effect.py:89-216) on the device: the diamond-square lattice of
 * generate_diamond_square_mask -- (2^levels + 1)^2 float32, field 4-byte aligned device memory whose four corners the caller drew
 * (corners_host: [0, 0], [0, -1], [-1, -1], [-1, 0]) -- filled level by level with numpy's float32 / float64 roundings from the PCG64
 * stream (state, inc) positioned AFTER the corner draws; noise_weight_host[l] = roughness ** l as Python computes it.
 * *consumed_host = the raw draws taken (the caller moves its generator, then draws the crop offsets).  Asynchronous on the ctx stream.
 * vkx_fog_stretch_f32_dev: mask float32 [h, w] = the crop (up, left) stretched like the reference: x - min, / max, * float32(span),
 * + float32(lo).  (vkit_amd/csrc/fog.hip) */
int vkx_fog_field_f32_dev(vkx_ctx *ctx, const uint64_t *state, const uint64_t *inc, int levels, const double *noise_weight_host,
                          const float *corners_host, float *field, long long *consumed_host);
int vkx_fog_stretch_f32_dev(vkx_ctx *ctx, const float *field, int size, int up, int left, int h, int w, double span, double lo, float *mask);

/* glass_blur's shuffle planes (reference photometric/blur.py:204-250) kept on the device: pos_y / pos_x int32 [h, w] dense planes, the
 * operands of vkx_gather_u8_dev.  vkx_glass_init_dev: the identity (and the context's scratch for the rounds).  vkx_glass_round_dev: one
 * round of swaps -- centres at rows r0 + i pitch (i < n_rows), columns c0 + j pitch (j < n_cols); centre k = i n_cols + j exchanges its
 * entry with the one at clip(its current source position + (jump_y[k], jump_x[k])) with numpy's semantics of
 * `pos[centres], pos[to] = pos[to], pos[centres]` (right sides first, the last centre in C order wins a shared `to`).  The jumps are host
 * arrays (the caller's rng.integers draws); synchronous on return.  (vkit_amd/csrc/fog.hip) */
int vkx_glass_init_dev(vkx_ctx *ctx, int32_t *pos_y, int32_t *pos_x, int h, int w);
int vkx_glass_round_dev(vkx_ctx *ctx, int32_t *pos_y, int32_t *pos_x, int h, int w, int r0, int c0, int pitch, int n_rows, int n_cols,
                        const int32_t *jump_y_host, const int32_t *jump_x_host);

/* ---- throughput-mode noise plane ---------------------------------------------------------------
 * gaussion_noise (photometric/noise.py:44-54) adds np.round(rng.normal(0, std, shape)) drawn from the caller's numpy
 * Generator; the parity path takes that int16 plane from the caller (vkx_add_noise_i16, vkx_chain_item.noise).
 * vkx_noise_normal_i16 draws a plane with the same DISTRIBUTION on the device -- not the same values, which are a
 * function of numpy's bit stream: sample s of the flat C-order plane takes the (s & 3)-th 16-bit uniform of the
 * Philox2x32-10 block with counter (s >> 2, seed >> 32) and key = seed low word, and maps it through a 65536-entry
 * inverse-CDF table of round(N(0, std)) (vkx_noise_normal_table builds that table on the host, for checks).
 * Separately labelled: a caller opts in.
 * dst: int16 [h, w, cn], cn in 1..4; _dev: device plane (8-byte aligned when dense: stride_el == w * cn), asynchronous;
 * host variant: host plane, synchronous. */
int vkx_noise_normal_table(double std, int16_t *table_host /* [65536] */);
int vkx_noise_normal_i16_dev(vkx_ctx *ctx, int16_t *dst, ptrdiff_t stride_el, int h, int w, int cn, double std, uint64_t seed);
/* the planes of a batch that share `std` in one launch (each with its own seed): the table is staged once per workgroup */
typedef struct vkx_noise_plane {
    int16_t *dst;          /* device */
    ptrdiff_t stride_el;
    int h, w, cn, reserved;
    uint64_t seed;
} vkx_noise_plane;
int vkx_noise_normal_i16_batch_dev(vkx_ctx *ctx, const vkx_noise_plane *planes_host, int n_planes, double std);
int vkx_noise_normal_i16(vkx_ctx *ctx, int16_t *dst, ptrdiff_t stride_el, int h, int w, int cn, double std, uint64_t seed);

/* ---- batched geometric + photometric chain (device resident) ---------------------------
 * RandomDistortion's geometric stage followed by photometric members on one page image
 * (mechanism/distortion_policy/random_distortion.py:190-203 applied through
 * Distortion.distort, mechanism/distortion/interface.py:824-912), for a ragged batch of
 * independent images: image-grid remap -> gaussian_blur -> color_shift -> gaussion_noise -> line_streak
 * (photometric/streak.py:56-99); stages with a disabled parameter are skipped.
 * `items` is a HOST array; every pointer inside is a DEVICE pointer. */
typedef struct vkx_chain_item {
    const uint8_t *src;            /* uint8 [sh, sw, 3] */
    uint8_t *dst;                  /* uint8 [dh, dw, 3] */
    ptrdiff_t src_stride, dst_stride;
    int32_t sh, sw, dh, dw;
    const int32_t *src_vertices;   /* int32 [rows, cols, 2] (x, y) */
    const int32_t *dst_vertices;
    int32_t rows, cols;
    const int16_t *noise;          /* optional int16 [dh, dw, 3]; NULL = no noise stage.  noise_tiled: the tile buffer of a
                                      VKX_NP_NORMAL_TILES job of dh * dw * 3 samples instead (noise_stride_el unused) */
    ptrdiff_t noise_stride_el;
    double blur_sigma;
    int32_t blur_ksize;            /* <= 1 = no blur stage */
    int32_t hue_delta;
    int32_t hue_enabled;           /* 0 = no color_shift stage */
    int32_t streak_enabled;        /* 0 = no line_streak stage */
    int32_t streak_thickness, streak_gap, streak_dash_thickness, streak_dash_gap;
    int32_t streak_enable_vert, streak_enable_hori;
    uint8_t streak_color[4];
    int32_t noise_tiled;           /* 0: `noise` is a plane */
    double streak_alpha;           /* in [0, 1] */
} vkx_chain_item;
int vkx_chain_rgb_batch_dev(vkx_ctx *ctx, const vkx_chain_item *items, int n_items);
/* The numpy streams of a batch's gaussion_noise members and its chain in ONE call: jobs = VKX_NP_NORMAL_TILES jobs, each drawing the
 * tile buffer that is the `noise` (noise_tiled = 1) of one item, in item order.  Same results as vkx_np_draw_batch_dev(jobs) followed
 * by vkx_chain_rgb_batch_dev(items); the library cuts the batch into chunks of images and runs the microsecond kernels of either
 * call (carry resolution, tile tables, cell setup, noise row records: eight launches of a few workgroups) on the context's two other
 * streams, under the draw pass of the next chunk or the pixel kernel of the first one (vkit_amd/csrc/chain.hip).  Jobs of another
 * kind, or items the fused kernel declines, make it exactly the two calls.  Asynchronous on the ctx stream. */
int vkx_chain_rgb_batch_np_dev(vkx_ctx *ctx, const vkx_chain_item *items, int n_items, const vkx_np_job *jobs_host, int n_jobs,
                               vkx_np_result *results_host);
/* The cell setup of a chain call (one lane per lattice cell: homographies, edge tables, tile bins) reads nothing but the vertex
 * lattices and runs on a side stream of the context.  By default it starts after everything queued on the ctx stream before the
 * chain call.  A caller whose lattices are complete earlier says so: this call marks the current point of the ctx stream, and the
 * setup of the NEXT chain call waits for that point only (e.g. it then runs under a page composite queued between the mark and the
 * chain call).  One chain call per mark: a mark never outlives the lattices it spoke of. */
int vkx_chain_lattices_ready(vkx_ctx *ctx);

/* ---- camera-model states built on the device ------------------------------------------------------------------------
 * The state of camera_plane_only / camera_cubic_curve (mechanism/distortion/geometric/camera.py:58-265, 324-423: CameraModel,
 * the 2-D -> 3-D strategies, cv.Rodrigues / cv.projectPoints; grid_rendering/grid_creator.py:44-115: source lattice, projection,
 * shift by the rounded minimum; element/point.py:31-47: rounding) for a BATCH of configs: the dozen scalars of a state on the host
 * in C (vkx_camera_model_host: float32 / float64 exactly where numpy computes in them, libm's sin / cos / tan like Python's math
 * module), the per-vertex work on the device, one workgroup per state.  The vertex lattices come out bit for bit the host
 * operator's (tests/golden/camera_states.npz).  vkit_amd/csrc/camera.hip */
#define VKX_CAMERA_PLANE_ONLY 0
#define VKX_CAMERA_CUBIC_CURVE 1
typedef struct vkx_camera_config {
    int32_t kind, height, width, grid_size;
    double rotation_unit_vec[3];
    double rotation_theta;
    double focal_length, camera_distance;   /* 0 = not given: completed from the shape like the reference */
    double principal_point[3];
    int32_t principal_point_len;            /* 0 = not given; 2 or 3 */
    int32_t reserved;
    double curve_alpha, curve_beta, curve_direction, curve_scale;     /* VKX_CAMERA_CUBIC_CURVE */
} vkx_camera_config;
typedef struct vkx_camera_model {           /* what the per-vertex kernel consumes; exposed for the parity tests */
    double R[9], t[3];                      /* float64 Rodrigues matrix of the float32 rotation vector; float32 translation as double */
    double fx, fy, cx, cy;
    float a0, a1, along_min, along_range;   /* cubic curve: first row of the float32 direction matrix, extent of the page along it */
    double poly[4], curve_scale;
    int32_t rows, cols;                     /* lattice shape */
    int32_t points_f32, reserved;           /* the projected points take float32 (plane only: float32 3-D points) */
} vkx_camera_model;
#define VKX_GRID_STATE_NAN 1u               /* a projected vertex is NaN (the reference raises ValueError in Point's round()) */
#define VKX_GRID_STATE_INF 2u               /* ... infinite (OverflowError) */
#define VKX_GRID_STATE_RANGE 4u             /* the lattice leaves int32 */
#define VKX_GRID_STATE_DIVIDE 8u            /* similarity_mls: a vertex on an integer handle position (the reference raises FloatingPointError) */
typedef struct vkx_grid_state {
    int32_t rows, cols, dh, dw;             /* lattice shape; result shape = extent of the destination lattice */
    int32_t shift_y, shift_x;               /* DistortionStateImageGridBased.shift_amount_* */
    uint32_t flags, reserved;
} vkx_grid_state;
int vkx_camera_model_host(const vkx_camera_config *config, vkx_camera_model *out);
/* src_vertices / dst_vertices: HOST arrays of n DEVICE pointers to int32 [rows, cols, 2] (x, y) lattices (rows / cols of a config:
 * vkx_camera_model_host).  stream: VKX_STREAM_*: on which of the context's streams the states are built -- a side stream builds
 * the states of the next batch while the compute stream still runs the current one.  states_host (page-locked memory for an
 * asynchronous copy) is valid after vkx_ctx_sync_stream(ctx, stream).  The chain calls that follow start their cell setup after
 * this call's kernel (it records the lattices-ready point of vkx_chain_lattices_ready). */
int vkx_camera_states_dev(vkx_ctx *ctx, const vkx_camera_config *configs_host, int n, int32_t *const *src_vertices,
                          int32_t *const *dst_vertices, vkx_grid_state *states_host, int stream);

/* SimilarityMlsState (geometric/mls.py:140-157; SimilarityMlsPointProjector.project_point :38-135 per lattice vertex, then
 * grid_rendering/grid_creator.py:44-115 with resize_as_src = False, the operator's default) for a BATCH of configs: handle tables are
 * HOST arrays -- float32 [n, 2] integer positions (PointTuple.to_smooth_np_array) and float64 [n, 2] smooth positions of the source and
 * destination handles --, the lattices DEVICE buffers as for vkx_camera_states_dev, to which everything else said there applies.
 * (vkit_amd/csrc/mls.hip) */
typedef struct vkx_mls_config {
    int32_t height, width, grid_size, n_handles;
    const float *src_handles, *dst_handles;
    const double *src_handles_smooth, *dst_handles_smooth;
} vkx_mls_config;
int vkx_mls_states_dev(vkx_ctx *ctx, const vkx_mls_config *configs_host, int n, int32_t *const *src_vertices,
                       int32_t *const *dst_vertices, vkx_grid_state *states_host, int stream);

/* ---- per-kernel timing -----------------------------------------------------------------
 * When enabled, every kernel launch is bracketed by a hipEvent pair recorded on the ctx
 * stream; vkx_ctx_collect_timings synchronises and folds them into per-kernel totals.
 * enabled = 2: only the large kernels (k_chain_fused, k_np_draw) are bracketed -- an event pair costs a few microseconds of
 * stream time, which a step of a dozen microsecond kernels notices. */
int vkx_ctx_set_timing(vkx_ctx *ctx, int enabled);
int vkx_ctx_collect_timings(vkx_ctx *ctx, int *n_kernels);
int vkx_ctx_get_timing(vkx_ctx *ctx, int index, const char **name, double *total_ms, long long *launches);
int vkx_ctx_reset_timings(vkx_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* VKX_H_ */
