// cv.cvtColor RGB <-> HSV_FULL on uint8 (element/image.py:188-202,771-814; photometric/color.py:93-116), shared by the
// single-stage kernels (photo.hip) and the fused chain kernel (fused.hip).  Must agree bit for bit with
// oracle/vkx_oracle.c: integer LUT division forwards, the scalar float32 HSV2RGB_native formula (no FMA) backwards.
#ifndef VKX_COLOR_H_
#define VKX_COLOR_H_

#include "vkx_internal.h"

namespace vkd {

// sdiv[i] = cvRound((255 << 12) / i), hdiv[i] = cvRound((256 << 12) / (6 i)): both < 2^24, and diff, |hh| < 2^11, so the
// 24-bit multiplies (full rate) are exact.
// WRAP_H: leave the hue as computed, valid modulo 256 (callers that add a shift and reduce modulo 256 anyway).
// Otherwise the negative half is moved up by 256; the quotient then lies in 0..255 by construction (|hh| <= 5 diff before
// the division, so at most 213 after it, at least -43): saturate_cast has nothing to clamp.
template <bool WRAP_H = false>
__device__ __forceinline__ void rgb2hsv_full(const int *sdiv, const int *hdiv, int r, int g, int b, int &H, int &S, int &V)
{
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    S = (__mul24(diff, sdiv[v]) + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    // |hh| < 2^11 and hdiv < 2^24: the signed 24-bit multiply-add is exact (the intrinsic, not an asm statement with a
    // scalar operand: see mad_u24_ks in fused.hip)
    hh = __mul24(hh, hdiv[diff]) + (1 << 11);
    hh >>= 12;
    if (!WRAP_H) hh += hh < 0 ? 256 : 0;
    H = hh;
    V = v;
}

// Branch free: with s == 0 every candidate below is fv * 1.0f == fv, which is the reference's grey shortcut; H < 256
// keeps the sector in 0..5.  sector table (b, g, r): 0 (t1,t3,t0) 1 (t1,t0,t2) 2 (t3,t0,t1) 3 (t0,t2,t1) 4 (t0,t1,t3)
// 5 (t2,t1,t0) = (t1, odd ? t0 : t3, odd ? t2 : t0) rotated left by sector / 2.
__device__ __forceinline__ void hsv2rgb_full(int H, int S, int V, int &r, int &g, int &b)
{
    const float s = S * (1.0f / 255.0f);
    const float fv = V * (1.0f / 255.0f);
    // h = H * 6 / 256 is exact in float32 (6 / 256 = 3 / 128, H * 6 < 2^11): its floor and fraction in integers
    const int h6 = __mul24(H, 6);
    const int sector = h6 >> 8;
    const float h = (float)(h6 & 255) * (1.0f / 256);
    const bool odd = sector & 1;
    const int rot = sector >> 1;
    // t2 = fv (1 - s h) serves the odd sectors, t3 = fv (1 - s (1 - h)) the even ones: only the one in use is evaluated
    const float t0 = fv;
    const float t1 = fv * (1.f - s);
    const float tt = fv * (1.f - s * (odd ? h : 1.f - h));
    const float u0 = t1, u1 = odd ? t0 : tt, u2 = odd ? tt : t0;
    const float fb = rot == 0 ? u0 : (rot == 1 ? u1 : u2);
    const float fg = rot == 0 ? u1 : (rot == 1 ? u2 : u0);
    const float fr = rot == 0 ? u2 : (rot == 1 ? u0 : u1);
    // 0 <= f <= 1, so round-half-even needs neither the cvRound range check nor the saturate_cast clamp
    r = __float2int_rn(fr * 255.0f);
    g = __float2int_rn(fg * 255.0f);
    b = __float2int_rn(fb * 255.0f);
}

// The hue shift of the fused chain kernel: the same arithmetic as rgb2hsv_full + hsv2rgb_full with the shifted hue, arranged
// for the instruction mix of gfx950 (simple integer / float32 adds and multiplies issue at twice the rate of converts,
// compares, selects and 24-bit multiplies there):
//   * t0 * 255 rounds to V itself (V / 255 is within one ulp of the quotient, times 255 within one ulp of V): not evaluated;
//   * cvRound(t * 255) as  fl(fl(t * 255) + 1.5 * 2^23): the sum is rounded to an integer, ties to even, exactly like cvRound
//     of the rounded product, and the integer sits in the low byte of the float's bit pattern;
//   * the sector's channel assignment is ONE byte permute with a selector looked up by sector (`sel`, kHsvSelectors in LDS)
//     instead of nine selects: sources are byte 0 = q/t, byte 1 = V (one dword) and byte 4 = p (the other).
// Returns r | g << 8 | b << 16.
//   sector table (b, g, r): 0 (p, t, V) 1 (p, V, q) 2 (t, V, p) 3 (V, q, p) 4 (V, p, t) 5 (q, p, V)
__device__ constexpr uint32_t kHsvSelectors[8] = {0x0c040001u, 0x0c040100u, 0x0c000104u, 0x0c010004u,
                                                  0x0c010400u, 0x0c000401u, 0x0c0c0c0cu, 0x0c0c0c0cu};

__device__ __forceinline__ uint32_t hue_shift_packed(const int *sdiv, const int *hdiv, const uint32_t *sel, int delta,
                                                     int r, int g, int b)
{
    int H, S, V;
    rgb2hsv_full<true>(sdiv, hdiv, r, g, b, H, S, V);
    const uint32_t h6 = __umul24((uint32_t)(H + delta) & 255u, 6u);   // python modulo 256, then h = H * 6 / 256 exactly
    const float s = (float)S * (1.0f / 255.0f);
    const float fv = (float)V * (1.0f / 255.0f);
    const float h = (float)(h6 & 255u) * (1.0f / 256);
    const float x = (h6 & 256u) ? h : 1.f - h;              // odd sectors interpolate with h, even ones with 1 - h
    const float t1 = fv * (1.f - s);
    const float tt = fv * (1.f - s * x);
    const float m1 = t1 * 255.0f + 12582912.0f;             // products and sums round separately (-ffp-contract=off)
    const float mt = tt * 255.0f + 12582912.0f;
    const uint32_t lo = __float_as_uint(mt) | ((uint32_t)V << 8);   // byte 0 = q / t, byte 1 = V (bits 8..15 of mt are 0)
    return __builtin_amdgcn_perm(__float_as_uint(m1), lo, sel[h6 >> 8]);
}

} // namespace vkd

#endif // VKX_COLOR_H_
