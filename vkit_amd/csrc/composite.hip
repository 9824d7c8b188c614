// fill_np_array (element/opt.py:118-209) as an ordered list of box-clipped layers blended in place on gfx950:
// the text-layer alpha composite of PageAssemblerStep.run (pipeline/text_detection/page_assembler.py:155-236).
// One launch per layer keeps the reference's layer order on the stream; lanes cover 64 consecutive pixels of a
// box row so destination traffic is coalesced and untouched pixels are neither read nor written.
#include "vkx_internal.h"

namespace {

template <typename T> struct LayerDev {
    int up, left, height, width;
    const uint8_t *mask;
    ptrdiff_t mask_stride;
    const float *alpha;
    ptrdiff_t alpha_stride;
    const T *value;
    ptrdiff_t value_stride; // elements of T per box row
    float alpha_scalar;
    int copy; // scalar alpha == 1.0
    int mode; // VKX_FILL_*: honoured in the copy branch only (element/opt.py:150-158)
    T value_const[4];
};

// uint8: trunc(fl32(1-a)*fl32(d) + fl32(a*fl32(v))), products rounded separately (-ffp-contract=off);
// float32: the same expression without the final conversion.
__device__ __forceinline__ uint8_t blend_px(float w0, float w1, uint8_t d, uint8_t v)
{
    const float t0 = w0 * (float)d, t1 = w1 * (float)v;
    return (uint8_t)(t0 + t1);
}
__device__ __forceinline__ float blend_px(float w0, float w1, float d, float v)
{
    const float t0 = w0 * d, t1 = w1 * v;
    return t0 + t1;
}

template <typename T, int CN>
__global__ void __launch_bounds__(256) k_fill(T *dst, ptrdiff_t dstride, LayerDev<T> L)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= L.width || y >= L.height) return;
    const float a = L.alpha ? L.alpha[(ptrdiff_t)y * L.alpha_stride + x] : L.alpha_scalar;
    const bool sel = L.mask ? L.mask[(ptrdiff_t)y * L.mask_stride + x] > 0 : (L.alpha ? a > 0.0f : true);
    if (!sel) return;
    T *d = dst + (ptrdiff_t)(L.up + y) * dstride + (ptrdiff_t)(L.left + x) * CN;
    const T *v = L.value ? L.value + (ptrdiff_t)y * L.value_stride + (ptrdiff_t)x * CN : nullptr;
    if (L.copy) {
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const T val = v ? v[c] : L.value_const[c];
            if (L.mode == VKX_FILL_PLAIN || (L.mode == VKX_FILL_KEEP_MAX ? d[c] < val : d[c] > val)) d[c] = val;
        }
    } else {
        const float w1 = a, w0 = 1.0f - w1;
#pragma unroll
        for (int c = 0; c < CN; c++) d[c] = blend_px(w0, w1, d[c], v ? v[c] : L.value_const[c]);
    }
}

template <typename LAYER>
int check_layers(const LAYER *layers, int n_layers, int h, int w)
{
    for (int i = 0; i < n_layers; i++) {
        const LAYER &l = layers[i];
        if (l.height < 0 || l.width < 0 || l.up < 0 || l.left < 0 || l.up + l.height > h || l.left + l.width > w) {
            vkx_set_error("layer %d: box (up=%d left=%d h=%d w=%d) outside the %dx%d destination", i, l.up, l.left,
                          l.height, l.width, h, w);
            return VKX_ERR_INVALID;
        }
        if (!l.alpha && (l.alpha_scalar < 0.0 || l.alpha_scalar > 1.0)) {
            vkx_set_error("alpha=%g is invalid.", l.alpha_scalar);
            return VKX_ERR_INVALID;
        }
        if (l.mode < VKX_FILL_PLAIN || l.mode > VKX_FILL_KEEP_MIN) {
            vkx_set_error("layer %d: unknown fill mode %d", i, l.mode);
            return VKX_ERR_INVALID;
        }
    }
    return VKX_OK;
}

} // namespace

VKX_EXPORT int vkx_fill_u8_dev(vkx_ctx *ctx, uint8_t *dst, int h, int w, int cn, ptrdiff_t dst_stride,
                               const vkx_layer *layers, int n_layers)
{
    VKX_REQUIRE(ctx && dst, "NULL argument");
    VKX_REQUIRE(n_layers >= 0 && (n_layers == 0 || layers), "bad layer list");
    VKX_REQUIRE(cn == 1 || cn == 3 || cn == 4, "1, 3 or 4 channels");
    int rc = check_layers(layers, n_layers, h, w);
    if (rc) return rc;
    for (int i = 0; i < n_layers; i++) {
        const vkx_layer &l = layers[i];
        if (l.height == 0 || l.width == 0) continue;
        if (!l.alpha && l.alpha_scalar == 0.0) continue; // element/opt.py:143-144
        LayerDev<uint8_t> L;
        L.up = l.up; L.left = l.left; L.height = l.height; L.width = l.width;
        L.mask = l.mask; L.mask_stride = l.mask_stride;
        L.alpha = l.alpha; L.alpha_stride = l.alpha_stride_el;
        L.value = l.value; L.value_stride = l.value_stride;
        L.alpha_scalar = (float)l.alpha_scalar;
        L.copy = !l.alpha && l.alpha_scalar == 1.0;
        L.mode = l.mode;
        for (int c = 0; c < 4; c++) L.value_const[c] = l.value_const[c];
        dim3 block(64, 4), grid(vkx_blocks(l.width, 64), vkx_blocks(l.height, 4));
        switch (cn) {
        case 1: { VKX_TIMED(ctx, "k_fill"); k_fill<uint8_t, 1><<<grid, block, 0, ctx->stream>>>(dst, dst_stride, L); } break;
        case 3: { VKX_TIMED(ctx, "k_fill"); k_fill<uint8_t, 3><<<grid, block, 0, ctx->stream>>>(dst, dst_stride, L); } break;
        default: { VKX_TIMED(ctx, "k_fill"); k_fill<uint8_t, 4><<<grid, block, 0, ctx->stream>>>(dst, dst_stride, L); } break;
        }
        VKX_LAUNCH_CHECK();
    }
    return VKX_OK;
}

VKX_EXPORT int vkx_fill_f32_dev(vkx_ctx *ctx, float *dst, int h, int w, ptrdiff_t dst_stride_el,
                                const vkx_layer_f32 *layers, int n_layers)
{
    VKX_REQUIRE(ctx && dst, "NULL argument");
    VKX_REQUIRE(n_layers >= 0 && (n_layers == 0 || layers), "bad layer list");
    int rc = check_layers(layers, n_layers, h, w);
    if (rc) return rc;
    for (int i = 0; i < n_layers; i++) {
        const vkx_layer_f32 &l = layers[i];
        if (l.height == 0 || l.width == 0) continue;
        if (!l.alpha && l.alpha_scalar == 0.0) continue;
        LayerDev<float> L;
        L.up = l.up; L.left = l.left; L.height = l.height; L.width = l.width;
        L.mask = l.mask; L.mask_stride = l.mask_stride;
        L.alpha = l.alpha; L.alpha_stride = l.alpha_stride_el;
        L.value = l.value; L.value_stride = l.value_stride_el;
        L.alpha_scalar = (float)l.alpha_scalar;
        L.copy = !l.alpha && l.alpha_scalar == 1.0;
        L.mode = l.mode;
        L.value_const[0] = l.value_const;
        dim3 block(64, 4), grid(vkx_blocks(l.width, 64), vkx_blocks(l.height, 4));
        { VKX_TIMED(ctx, "k_fill"); k_fill<float, 1><<<grid, block, 0, ctx->stream>>>(dst, dst_stride_el, L); }
        VKX_LAUNCH_CHECK();
    }
    return VKX_OK;
}
