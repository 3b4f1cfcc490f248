// csrc/metal_common.h — device helpers for the `metal` numerics family: the float arithmetic the
// reference's Metal shaders perform on unorm8 samples.  Shared by the
// convertscale metal path, videofilter, compositor and deinterlace kernels.
//
// Restates (does not copy) reference common/vfmetalshaders.m:40-168 (matrices, yuvToRGB, rgbaToNV12 /
// rgbaToI420) and convertscale/metalconvertscale_shaders.h:151-269 (packed YUV fetch / store).
// The CPU twin used by the tests is oracle/metalref.c; both are compiled with -ffp-contract=off and
// evaluate every expression in the same order — with explicit fmaf () in the same places (interpolation, colour
// matrices, blur sums: what MSL's fast-math would contract; IEEE fma is deterministic on both sides) — so they
// normally agree bit for bit (tests allow +-1 LSB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vfhip.h"

namespace vfhip {
namespace metal {

struct Img {                 // one video frame in device memory
  const uint8_t *p[3];
  int s[3];
  int w, h, fmt;
  int m709;                  // reference rule: BT.709 iff matrix == BT709, else BT.601 (vfmetaltextureutil.m:35-41)
};
struct OutImg {
  uint8_t *p[3];
  int s[3];
  int w, h, fmt;
  int m709;
};

struct F4 { float r, g, b, a; };

__device__ __forceinline__ float un8 (uint32_t v) { return (float) v * (1.0f / 255.0f); }
__device__ __forceinline__ float clamp01 (float x) { return fminf (fmaxf (x, 0.0f), 1.0f); }
// unorm8 write: clamp, x255, round to nearest even.  v_cvt_pk_u8_f32 does the rounding, the saturation to [0, 255] AND
// the insertion into a byte lane in one instruction (checked against rintf(clamp) on 215 k values incl. all ties and
// out-of-range inputs: tools/ubench/cvt_pk_test.hip) — 1/3 of the issue cycles of max/min/rndne/cvt/shift/or.
__device__ __forceinline__ uint32_t quant8 (float x) { return __builtin_amdgcn_cvt_pk_u8_f32 (x * 255.0f, 0u, 0u); }
__device__ __forceinline__ int iclamp (int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// limited-range YCbCr -> RGB (reference yuvToRGB); the matrix's zero entries are skipped (x + 0*u == x for finite u)
__device__ __forceinline__ F4 yuv_to_rgb (float y, float cb, float cr, int m709)
{
  const float yy = y - 16.0f / 255.0f, u = cb - 128.0f / 255.0f, v = cr - 128.0f / 255.0f;
  F4 o;
  const float ly = 1.164383f * yy;
  if (m709) {
    o.r = fmaf (1.792741f, v, ly);
    o.g = fmaf (-0.532909f, v, fmaf (-0.213249f, u, ly));
    o.b = fmaf (2.112402f, u, ly);
  } else {
    o.r = fmaf (1.596027f, v, ly);
    o.g = fmaf (-0.812968f, v, fmaf (-0.391762f, u, ly));
    o.b = fmaf (2.017232f, u, ly);
  }
  o.r = clamp01 (o.r); o.g = clamp01 (o.g); o.b = clamp01 (o.b); o.a = 1.0f;
  return o;
}

// RGB -> limited-range YCbCr, unclamped (reference bt601/709_rgb_matrix)
__device__ __forceinline__ void rgb_to_yuv (float r, float g, float b, int m709, float *y, float *u, float *v)
{
  if (m709) {
    *y = fmaf (0.062007f, b, fmaf (0.614231f, g, 0.182586f * r)) + 16.0f / 255.0f;
    *u = fmaf (0.439216f, b, fmaf (-0.338572f, g, -0.100644f * r)) + 128.0f / 255.0f;
    *v = fmaf (-0.040274f, b, fmaf (-0.398942f, g, 0.439216f * r)) + 128.0f / 255.0f;
  } else {
    *y = fmaf (0.097906f, b, fmaf (0.504129f, g, 0.256788f * r)) + 16.0f / 255.0f;
    *u = fmaf (0.439216f, b, fmaf (-0.290993f, g, -0.148223f * r)) + 128.0f / 255.0f;
    *v = fmaf (-0.071427f, b, fmaf (-0.367788f, g, 0.439216f * r)) + 128.0f / 255.0f;
  }
}

// ---- texture sampling ---------------------------------------------------
struct Taps { int i0, i1; float f; };
__device__ __forceinline__ Taps lin_taps_px (int n, float x)           // x in texel units, already minus .5
{
  const float fl = floorf (x);
  Taps t; t.f = x - fl;
  const int i = (int) fl;
  t.i0 = iclamp (i, 0, n - 1); t.i1 = iclamp (i + 1, 0, n - 1);
  return t;
}
__device__ __forceinline__ Taps lin_taps (int n, float coord) { return lin_taps_px (n, coord * (float) n - 0.5f); }   // coord normalised 0..1
__device__ __forceinline__ int near_tap (int n, float coord) { return iclamp ((int) floorf (coord * (float) n), 0, n - 1); }
__device__ __forceinline__ float lerp2 (float a, float b, float f) { return fmaf (b - a, f, a); }

// one channel of a plane with `bpt` bytes per texel
__device__ __forceinline__ float plane_taps (const uint8_t *p, int stride, int bpt, int ch, Taps tx, Taps ty)
{
  const uint8_t *r0 = p + (size_t) ty.i0 * stride, *r1 = p + (size_t) ty.i1 * stride;
  const float a = lerp2 (un8 (r0[tx.i0 * bpt + ch]), un8 (r0[tx.i1 * bpt + ch]), tx.f);
  const float b = lerp2 (un8 (r1[tx.i0 * bpt + ch]), un8 (r1[tx.i1 * bpt + ch]), tx.f);
  return lerp2 (a, b, ty.f);
}
__device__ __forceinline__ float plane_linear (const uint8_t *p, int stride, int bpt, int ch, int W, int H, float u, float v)
{
  return plane_taps (p, stride, bpt, ch, lin_taps (W, u), lin_taps (H, v));
}
__device__ __forceinline__ float plane_nearest (const uint8_t *p, int stride, int bpt, int ch, int W, int H, float u, float v)
{
  return un8 (p[(size_t) near_tap (H, v) * stride + near_tap (W, u) * bpt + ch]);
}
__device__ __forceinline__ float plane_sample (const uint8_t *p, int stride, int bpt, int ch, int W, int H, float u, float v, bool linear)
{
  return linear ? plane_linear (p, stride, bpt, ch, W, H, u, v) : plane_nearest (p, stride, bpt, ch, W, H, u, v);
}

// sample any of the six input formats at normalised (u, v) -> logical RGBA float
// RGBA / BGRA and NV12 chroma fetch a texel's bytes together (one dword / one (U, V) pair per tap instead of a byte load per channel and
// tap: 4 instead of 16 loads per RGBA sample) and then interpolate each byte exactly like plane_taps: same values, same operations.
__device__ __forceinline__ F4 sample_rgba (const Img &im, float u, float v, bool linear)
{
  typedef uint32_t __attribute__ ((aligned (1))) u32_any;
  typedef uint16_t __attribute__ ((aligned (1))) u16_any;
  F4 o;
  switch (im.fmt) {
    case VFHIP_FORMAT_BGRA: case VFHIP_FORMAT_RGBA: {
      float c[4];
      if (linear) {
        const Taps tx = lin_taps (im.w, u), ty = lin_taps (im.h, v);
        const uint8_t *r0 = im.p[0] + (size_t) ty.i0 * im.s[0], *r1 = im.p[0] + (size_t) ty.i1 * im.s[0];
        const uint32_t t00 = *reinterpret_cast<const u32_any *> (r0 + 4 * tx.i0), t10 = *reinterpret_cast<const u32_any *> (r0 + 4 * tx.i1);
        const uint32_t t01 = *reinterpret_cast<const u32_any *> (r1 + 4 * tx.i0), t11 = *reinterpret_cast<const u32_any *> (r1 + 4 * tx.i1);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float a = lerp2 (un8 ((t00 >> (8 * k)) & 0xffu), un8 ((t10 >> (8 * k)) & 0xffu), tx.f);
          const float b = lerp2 (un8 ((t01 >> (8 * k)) & 0xffu), un8 ((t11 >> (8 * k)) & 0xffu), tx.f);
          c[k] = lerp2 (a, b, ty.f);
        }
      } else {
        const uint32_t t = *reinterpret_cast<const u32_any *> (im.p[0] + (size_t) near_tap (im.h, v) * im.s[0] + 4 * near_tap (im.w, u));
#pragma unroll
        for (int k = 0; k < 4; k++) c[k] = un8 ((t >> (8 * k)) & 0xffu);
      }
      const bool rgba = im.fmt == VFHIP_FORMAT_RGBA;
      o.r = rgba ? c[0] : c[2]; o.g = c[1]; o.b = rgba ? c[2] : c[0]; o.a = c[3];
      return o;
    }
    case VFHIP_FORMAT_NV12: {
      const int cw = (im.w + 1) / 2, chh = (im.h + 1) / 2;
      const float y = plane_sample (im.p[0], im.s[0], 1, 0, im.w, im.h, u, v, linear);
      float cb, cr;
      if (linear) {
        const Taps tx = lin_taps (cw, u), ty = lin_taps (chh, v);
        const uint8_t *r0 = im.p[1] + (size_t) ty.i0 * im.s[1], *r1 = im.p[1] + (size_t) ty.i1 * im.s[1];
        const uint32_t t00 = *reinterpret_cast<const u16_any *> (r0 + 2 * tx.i0), t10 = *reinterpret_cast<const u16_any *> (r0 + 2 * tx.i1);
        const uint32_t t01 = *reinterpret_cast<const u16_any *> (r1 + 2 * tx.i0), t11 = *reinterpret_cast<const u16_any *> (r1 + 2 * tx.i1);
        cb = lerp2 (lerp2 (un8 (t00 & 0xffu), un8 (t10 & 0xffu), tx.f), lerp2 (un8 (t01 & 0xffu), un8 (t11 & 0xffu), tx.f), ty.f);
        cr = lerp2 (lerp2 (un8 (t00 >> 8), un8 (t10 >> 8), tx.f), lerp2 (un8 (t01 >> 8), un8 (t11 >> 8), tx.f), ty.f);
      } else {
        const uint32_t t = *reinterpret_cast<const u16_any *> (im.p[1] + (size_t) near_tap (chh, v) * im.s[1] + 2 * near_tap (cw, u));
        cb = un8 (t & 0xffu); cr = un8 (t >> 8);
      }
      return yuv_to_rgb (y, cb, cr, im.m709);
    }
    case VFHIP_FORMAT_I420: {
      const int cw = (im.w + 1) / 2, chh = (im.h + 1) / 2;
      const float y = plane_sample (im.p[0], im.s[0], 1, 0, im.w, im.h, u, v, linear);
      const float cb = plane_sample (im.p[1], im.s[1], 1, 0, cw, chh, u, v, linear);
      const float cr = plane_sample (im.p[2], im.s[2], 1, 0, cw, chh, u, v, linear);
      return yuv_to_rgb (y, cb, cr, im.m709);
    }
    default: {   // UYVY / YUY2: always nearest macro-pixel (reference metalconvertscalerenderer.m:184-185)
      const int tw = im.w / 2;                           // macro-pixels; an odd last column is dropped (:420)
      const float texw = (float) tw, fullw = texw * 2.0f;
      const float px = u * fullw;
      const float mx = floorf (px / 2.0f);
      const float sub = px - mx * 2.0f;
      const int tx = near_tap (tw, (mx + 0.5f) / texw), ty = near_tap (im.h, v);
      const uint8_t *t = im.p[0] + (size_t) ty * im.s[0] + 4 * tx;
      float y, cb, cr;
      if (im.fmt == VFHIP_FORMAT_UYVY) { cb = un8 (t[0]); cr = un8 (t[2]); y = sub < 1.0f ? un8 (t[1]) : un8 (t[3]); }
      else { cb = un8 (t[1]); cr = un8 (t[3]); y = sub < 1.0f ? un8 (t[0]) : un8 (t[2]); }
      return yuv_to_rgb (y, cb, cr, im.m709);
    }
  }
}

// 1:1 fetch at pixel (x, y) of an image whose full-resolution planes match the output grid (filter, deinterlace,
// unscaled compositor pads): exact luma / RGBA texel; 4:2:0 chroma either bilinear at texel coordinate
// 0.5*x - 0.25 (what a linear sampler sees at texcoord (x+.5)/W on the half-size plane) or the nearest texel x/2
// (the deinterlace input pass, filter::nearest).  SURVEY.md Appendix B item 2.
__device__ __forceinline__ F4 fetch_1to1 (const Img &im, int x, int y, bool chroma_linear)
{
  x = iclamp (x, 0, im.w - 1); y = iclamp (y, 0, im.h - 1);
  if (im.fmt == VFHIP_FORMAT_BGRA || im.fmt == VFHIP_FORMAT_RGBA) {
    const uint32_t t = *reinterpret_cast<const uint32_t *> (im.p[0] + (size_t) y * im.s[0] + 4 * x);
    F4 o;
    o.g = un8 ((t >> 8) & 0xff); o.a = un8 (t >> 24);
    if (im.fmt == VFHIP_FORMAT_RGBA) { o.r = un8 (t & 0xff); o.b = un8 ((t >> 16) & 0xff); }
    else { o.b = un8 (t & 0xff); o.r = un8 ((t >> 16) & 0xff); }
    return o;
  }
  const int cw = (im.w + 1) / 2, chh = (im.h + 1) / 2;
  const float Y = un8 (im.p[0][(size_t) y * im.s[0] + x]);
  float cb, cr;
  if (chroma_linear) {
    const Taps tx = lin_taps_px (cw, 0.5f * (float) x - 0.25f), ty = lin_taps_px (chh, 0.5f * (float) y - 0.25f);
    if (im.fmt == VFHIP_FORMAT_NV12) {
      // the (U, V) pair of a tap as one 16-bit load (4 loads instead of 8), then plane_taps' interpolation on each byte
      typedef uint16_t __attribute__ ((aligned (1))) u16_any;
      const uint8_t *r0 = im.p[1] + (size_t) ty.i0 * im.s[1], *r1 = im.p[1] + (size_t) ty.i1 * im.s[1];
      const uint32_t t00 = *reinterpret_cast<const u16_any *> (r0 + 2 * tx.i0), t10 = *reinterpret_cast<const u16_any *> (r0 + 2 * tx.i1);
      const uint32_t t01 = *reinterpret_cast<const u16_any *> (r1 + 2 * tx.i0), t11 = *reinterpret_cast<const u16_any *> (r1 + 2 * tx.i1);
      cb = lerp2 (lerp2 (un8 (t00 & 0xffu), un8 (t10 & 0xffu), tx.f), lerp2 (un8 (t01 & 0xffu), un8 (t11 & 0xffu), tx.f), ty.f);
      cr = lerp2 (lerp2 (un8 (t00 >> 8), un8 (t10 >> 8), tx.f), lerp2 (un8 (t01 >> 8), un8 (t11 >> 8), tx.f), ty.f);
    }
    else { cb = plane_taps (im.p[1], im.s[1], 1, 0, tx, ty); cr = plane_taps (im.p[2], im.s[2], 1, 0, tx, ty); }
  } else {
    const int cx = iclamp (x >> 1, 0, cw - 1), cy = iclamp (y >> 1, 0, chh - 1);
    if (im.fmt == VFHIP_FORMAT_NV12) { cb = un8 (im.p[1][(size_t) cy * im.s[1] + 2 * cx]); cr = un8 (im.p[1][(size_t) cy * im.s[1] + 2 * cx + 1]); }
    else { cb = un8 (im.p[1][(size_t) cy * im.s[1] + cx]); cr = un8 (im.p[2][(size_t) cy * im.s[2] + cx]); }
  }
  return yuv_to_rgb (Y, cb, cr, im.m709);
}

__device__ __forceinline__ uint32_t pack_rgba8 (uint32_t r, uint32_t g, uint32_t b, uint32_t a) { return r | (g << 8) | (b << 16) | (a << 24); }
__device__ __forceinline__ uint32_t quant_rgba8 (F4 c)
{
  uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32 (c.r * 255.0f, 0u, 0u);
  q = __builtin_amdgcn_cvt_pk_u8_f32 (c.g * 255.0f, 1u, q);
  q = __builtin_amdgcn_cvt_pk_u8_f32 (c.b * 255.0f, 2u, q);
  return __builtin_amdgcn_cvt_pk_u8_f32 (c.a * 255.0f, 3u, q);
}
__device__ __forceinline__ F4 unpack_rgba8 (uint32_t q)
{
  F4 o; o.r = un8 (q & 0xff); o.g = un8 ((q >> 8) & 0xff); o.b = un8 ((q >> 16) & 0xff); o.a = un8 (q >> 24);
  return o;
}

// ---- output epilogue: one thread owns a 2x2 block of logical RGBA8 pixels -----------------------------
// q[dy][dx] must be edge-clamped duplicates when the block hangs over the right / bottom edge.
// Replaces the reference's separate RGBA->NV12 / I420 / UYVY / YUY2 compute passes and the render-target
// read-back (common/vfmetalshaders.m:90-168, convertscale/metalconvertscale_shaders.h:202-269) — the 8-bit
// intermediate stays in registers instead of making two more trips through memory.
__device__ __forceinline__ void store_block (const OutImg &o, int bx, int by, const uint32_t q[2][2])
{
  const int x0 = 2 * bx, y0 = 2 * by;
  switch (o.fmt) {
    case VFHIP_FORMAT_BGRA: case VFHIP_FORMAT_RGBA: {
#pragma unroll
      for (int dy = 0; dy < 2; dy++) {
        if (y0 + dy >= o.h) break;
        uint32_t *row = reinterpret_cast<uint32_t *> (o.p[0] + (size_t) (y0 + dy) * o.s[0]);
        uint32_t v0 = q[dy][0], v1 = q[dy][1];
        if (o.fmt == VFHIP_FORMAT_BGRA) {
          v0 = __builtin_amdgcn_perm (0u, v0, 0x03000102u);           // swap bytes 0 and 2
          v1 = __builtin_amdgcn_perm (0u, v1, 0x03000102u);
        }
        // the pair goes out as ONE 8-byte store when the address allows: a dword store per pixel leaves every store
        // instruction half-coalesced (lane stride 8 bytes)
        // ... and non-temporal: the frame is not read again, and keeping it out of L2 leaves the inputs (compositor pads, the
        // filter's LUT) there — A/B on one box: C4 43.7 k -> 48.2 k frames/s, C3 15.0 k -> 16.3 k
        if (x0 + 1 < o.w && ((reinterpret_cast<uintptr_t> (row) & 7) == 0)) {
          typedef uint32_t v2u __attribute__ ((ext_vector_type (2)));
          const v2u pair = { v0, v1 };
          __builtin_nontemporal_store (pair, reinterpret_cast<v2u *> (row + x0));
        }
        else { row[x0] = v0; if (x0 + 1 < o.w) row[x0 + 1] = v1; }
      }
      return;
    }
    case VFHIP_FORMAT_NV12: case VFHIP_FORMAT_I420: {
      float sr = 0.0f, sg = 0.0f, sb = 0.0f;
      uint32_t yq[2][2];
#pragma unroll
      for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
          const F4 c = unpack_rgba8 (q[dy][dx]);
          sr += c.r; sg += c.g; sb += c.b;
          float y, u, v; rgb_to_yuv (c.r, c.g, c.b, o.m709, &y, &u, &v);
          yq[dy][dx] = quant8 (y);
        }
      // luma: the two pixels of a row go out as ONE 2-byte store when the address allows (byte stores cost a full
      // store instruction each: measured 3x on the deinterlacer)
#pragma unroll
      for (int dy = 0; dy < 2; dy++) {
        if (y0 + dy >= o.h) break;
        uint8_t *d = o.p[0] + (size_t) (y0 + dy) * o.s[0] + x0;
        if (x0 + 1 < o.w && !((uintptr_t) d & 1)) __builtin_nontemporal_store ((uint16_t) (yq[dy][0] | (yq[dy][1] << 8)), reinterpret_cast<uint16_t *> (d));   // non-temporal like the RGB pairs (C5: +4 %)
        else { d[0] = (uint8_t) yq[dy][0]; if (x0 + 1 < o.w) d[1] = (uint8_t) yq[dy][1]; }
      }
      sr *= 0.25f; sg *= 0.25f; sb *= 0.25f;
      float y, u, v; rgb_to_yuv (sr, sg, sb, o.m709, &y, &u, &v);
      if (o.fmt == VFHIP_FORMAT_NV12) {
        uint8_t *d = o.p[1] + (size_t) by * o.s[1] + 2 * bx;
        const uint32_t U = quant8 (u), V = quant8 (v);
        if (!((uintptr_t) d & 1)) __builtin_nontemporal_store ((uint16_t) (U | (V << 8)), reinterpret_cast<uint16_t *> (d));
        else { d[0] = (uint8_t) U; d[1] = (uint8_t) V; }
      } else {
        o.p[1][(size_t) by * o.s[1] + bx] = (uint8_t) quant8 (u);
        o.p[2][(size_t) by * o.s[2] + bx] = (uint8_t) quant8 (v);
      }
      return;
    }
    default: {   // UYVY / YUY2: one macro-pixel per block row.  The reference dispatches width/2 macro-pixels
      // (metalconvertscale_shaders.h:210) and leaves the half macro-pixel of an odd width UNWRITTEN (whatever the
      // pool buffer held); here it is written from the edge pixel (callers clamp q to the last column — the clamp
      // the reference's own p1 carries at :217), so every byte of the output is defined.
      if (bx >= (o.w + 1) / 2) return;
#pragma unroll
      for (int dy = 0; dy < 2; dy++) {
        if (y0 + dy >= o.h) break;
        const F4 c0 = unpack_rgba8 (q[dy][0]), c1 = unpack_rgba8 (q[dy][1]);
        float ya, ua, va, yb, ub, vb;
        rgb_to_yuv (c0.r, c0.g, c0.b, o.m709, &ya, &ua, &va);
        rgb_to_yuv (c1.r, c1.g, c1.b, o.m709, &yb, &ub, &vb);
        const uint32_t U = quant8 ((ua + ub) * 0.5f), V = quant8 ((va + vb) * 0.5f), Y0 = quant8 (ya), Y1 = quant8 (yb);
        uint32_t *d = reinterpret_cast<uint32_t *> (o.p[0] + (size_t) (y0 + dy) * o.s[0]) + bx;
        *d = o.fmt == VFHIP_FORMAT_UYVY ? pack_rgba8 (U, Y0, V, Y1) : pack_rgba8 (Y0, U, Y1, V);
      }
      return;
    }
  }
}

// ---- 4 x 2 pixels per lane on 4:2:0 frames: the same values as fetch_1to1 (linear chroma) / store_block, with dword accesses ---------------
// fetch420_quad: pixels (x0 .. x0+3, y0 .. y0+1) of an NV12 / I420 frame, x0 = 4 * xq, y0 = 2 * by, as fetch_1to1 (im, x, y, true) returns them.
// A linear sampler at texcoord (x + .5) / W sits at 0.5 x - 0.25 on the half-size plane: an even column 2k takes chroma columns (k - 1, k) with
// weight .75 on the second, an odd one (k, k + 1) with .25 — so the eight pixels read chroma columns 2xq - 1 .. 2xq + 2 of rows by - 1 .. by + 1
// (edge-clamped like lin_taps_px): three 8-byte windows (NV12; 4-byte per plane for I420) instead of 32 two-byte taps, every texel converted once,
// the interpolation in fetch_1to1's order (horizontal, then vertical) with its weights.  Contract: W % 4 == 0, W >= 8, even H, 4-byte aligned
// luma rows, 2-byte aligned NV12 chroma rows.
__device__ __forceinline__ void fetch420_quad (const Img &im, int xq, int by, F4 out[2][4])
{
  typedef uint2 __attribute__ ((aligned (2))) uint2_a2;
  typedef uint32_t __attribute__ ((aligned (1))) uint32_a1;
  const int cw = im.w >> 1, chh = im.h >> 1;
  const int cbase = iclamp (2 * xq - 1, 0, cw - 4);
  uint32_t sel = 0;                                        // byte t: window index of chroma column clamp (2xq - 1 + t)
#pragma unroll
  for (int t = 0; t < 4; t++) sel |= (uint32_t) (iclamp (2 * xq - 1 + t, 0, cw - 1) - cbase) << (8 * t);
  const int jr[3] = { max (by - 1, 0), by, min (by + 1, chh - 1) };
  float cu[3][4], cv[3][4];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    uint32_t uq, vq;                                       // [c0 c1 c2 c3] of this row
    if (im.fmt == VFHIP_FORMAT_NV12) {
      const uint2 w = *reinterpret_cast<const uint2_a2 *> (im.p[1] + ((uint32_t) jr[r] * (uint32_t) im.s[1] + 2u * (uint32_t) cbase));
      const uint32_t su = sel << 1;                        // pair i = bytes (2i, 2i + 1)
      uq = __builtin_amdgcn_perm (w.y, w.x, su); vq = __builtin_amdgcn_perm (w.y, w.x, su + 0x01010101u);
    } else {
      const uint32_t wu = *reinterpret_cast<const uint32_a1 *> (im.p[1] + ((uint32_t) jr[r] * (uint32_t) im.s[1] + (uint32_t) cbase));
      const uint32_t wv = *reinterpret_cast<const uint32_a1 *> (im.p[2] + ((uint32_t) jr[r] * (uint32_t) im.s[2] + (uint32_t) cbase));
      uq = __builtin_amdgcn_perm (0u, wu, sel); vq = __builtin_amdgcn_perm (0u, wv, sel);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) { cu[r][t] = un8 ((uq >> (8 * t)) & 0xffu); cv[r][t] = un8 ((vq >> (8 * t)) & 0xffu); }
  }
  float hu[3][4], hv[3][4];                                // horizontal taps of the four pixel columns, per chroma row
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int dx = 0; dx < 4; dx++) {
      const int ta = (dx + 1) >> 1;                        // dx 0: (0, 1) .75 | 1: (1, 2) .25 | 2: (1, 2) .75 | 3: (2, 3) .25
      const float f = (dx & 1) ? 0.25f : 0.75f;
      hu[r][dx] = lerp2 (cu[r][ta], cu[r][ta + 1], f); hv[r][dx] = lerp2 (cv[r][ta], cv[r][ta + 1], f);
    }
#pragma unroll
  for (int dy = 0; dy < 2; dy++) {
    const uint32_t yq = *reinterpret_cast<const uint32_t *> (im.p[0] + ((uint32_t) (2 * by + dy) * (uint32_t) im.s[0] + 4u * (uint32_t) xq));
#pragma unroll
    for (int dx = 0; dx < 4; dx++) {
      // row 2j leans on chroma rows (j - 1, j) with .75, row 2j + 1 on (j, j + 1) with .25
      const float cb = dy ? lerp2 (hu[1][dx], hu[2][dx], 0.25f) : lerp2 (hu[0][dx], hu[1][dx], 0.75f);
      const float cr = dy ? lerp2 (hv[1][dx], hv[2][dx], 0.25f) : lerp2 (hv[0][dx], hv[1][dx], 0.75f);
      out[dy][dx] = yuv_to_rgb (un8 ((yq >> (8 * dx)) & 0xffu), cb, cr, im.m709);
    }
  }
}

// the same 4 x 2 pixels of any of the four element formats (BGRA / RGBA: one 16-byte load per row; contract: 16-byte aligned rows)
__device__ __forceinline__ void fetch_quad (const Img &im, int xq, int by, F4 out[2][4])
{
  if (im.fmt == VFHIP_FORMAT_NV12 || im.fmt == VFHIP_FORMAT_I420) { fetch420_quad (im, xq, by, out); return; }
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  const bool rgba = im.fmt == VFHIP_FORMAT_RGBA;
#pragma unroll
  for (int dy = 0; dy < 2; dy++) {
    const v4u t = *(reinterpret_cast<const v4u *> (im.p[0] + (size_t) (2 * by + dy) * im.s[0]) + xq);
#pragma unroll
    for (int dx = 0; dx < 4; dx++) {
      F4 o;
      o.g = un8 ((t[dx] >> 8) & 0xff); o.a = un8 (t[dx] >> 24);
      if (rgba) { o.r = un8 (t[dx] & 0xff); o.b = un8 ((t[dx] >> 16) & 0xff); }
      else { o.b = un8 (t[dx] & 0xff); o.r = un8 ((t[dx] >> 16) & 0xff); }
      out[dy][dx] = o;
    }
  }
}

// store_quad: what store_block (o, 2 * xq, by, left half) + store_block (o, 2 * xq + 1, by, right half) write, as one dword per luma row and one
// (NV12) chroma dword, or two 16-byte RGBA rows.  Contract: BGRA / RGBA / NV12 / I420 output, W % 4 == 0, even H, rows aligned for those stores.
__device__ __forceinline__ void store_quad (const OutImg &o, int xq, int by, const uint32_t q[2][4])
{
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  if (o.fmt == VFHIP_FORMAT_BGRA || o.fmt == VFHIP_FORMAT_RGBA) {
#pragma unroll
    for (int dy = 0; dy < 2; dy++) {
      v4u v = { q[dy][0], q[dy][1], q[dy][2], q[dy][3] };
      if (o.fmt == VFHIP_FORMAT_BGRA) {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __builtin_amdgcn_perm (0u, v[k], 0x03000102u);
      }
      __builtin_nontemporal_store (v, reinterpret_cast<v4u *> (o.p[0] + (size_t) (2 * by + dy) * o.s[0]) + xq);
    }
    return;
  }
  uint32_t yrow[2] = { 0, 0 }, uu[2], vv[2];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
      for (int dx = 0; dx < 2; dx++) {
        const F4 px = unpack_rgba8 (q[dy][2 * c + dx]);
        sr += px.r; sg += px.g; sb += px.b;
        float y, u, v; rgb_to_yuv (px.r, px.g, px.b, o.m709, &y, &u, &v);
        yrow[dy] |= quant8 (y) << (8 * (2 * c + dx));
      }
    sr *= 0.25f; sg *= 0.25f; sb *= 0.25f;
    float y, u, v; rgb_to_yuv (sr, sg, sb, o.m709, &y, &u, &v);
    uu[c] = quant8 (u); vv[c] = quant8 (v);
  }
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
    __builtin_nontemporal_store (yrow[dy], reinterpret_cast<uint32_t *> (o.p[0] + (size_t) (2 * by + dy) * o.s[0]) + xq);
  if (o.fmt == VFHIP_FORMAT_NV12)
    __builtin_nontemporal_store (uu[0] | (vv[0] << 8) | (uu[1] << 16) | (vv[1] << 24), reinterpret_cast<uint32_t *> (o.p[1] + (size_t) by * o.s[1]) + xq);
  else {
    *reinterpret_cast<uint16_t *> (o.p[1] + (size_t) by * o.s[1] + 2 * xq) = (uint16_t) (uu[0] | (uu[1] << 8));
    *reinterpret_cast<uint16_t *> (o.p[2] + (size_t) by * o.s[2] + 2 * xq) = (uint16_t) (vv[0] | (vv[1] << 8));
  }
}

// host helpers: VfHipFrame -> device image descriptors
// frame k of a batch: every plane pointer advanced by k * frame pitch (wave-uniform, scalar arithmetic)
__device__ __forceinline__ Img img_at (const Img &im, size_t off) { Img r = im; r.p[0] += off; r.p[1] += off; r.p[2] += off; return r; }
__device__ __forceinline__ OutImg out_at (const OutImg &im, size_t off) { OutImg r = im; r.p[0] += off; r.p[1] += off; r.p[2] += off; return r; }

// the contract of fetch_quad / store_quad for a (frame, batch pitch) pair: BGRA / RGBA / NV12 / I420, W % 4 == 0, W >= 8, even H, rows aligned for the
// 16-byte (RGB), dword (luma, NV12 chroma out), 2-byte (NV12 chroma in, I420 chroma out) accesses
static inline bool quad_frame_ok (const VfHipFrame *f, size_t pitch, bool is_output)
{
  const int fmt = f->info.format, w = f->info.width, h = f->info.height;
  if ((w & 3) || (h & 1) || w < 8) return false;
  const uintptr_t a0 = (uintptr_t) f->data[0] | (uintptr_t) f->stride[0] | (uintptr_t) pitch;
  if (fmt == VFHIP_FORMAT_BGRA || fmt == VFHIP_FORMAT_RGBA) return !(a0 & 15);
  if (fmt != VFHIP_FORMAT_NV12 && fmt != VFHIP_FORMAT_I420) return false;
  if (a0 & 3) return false;
  const uintptr_t a1 = (uintptr_t) f->data[1] | (uintptr_t) f->stride[1], a2 = (uintptr_t) f->data[2] | (uintptr_t) f->stride[2];
  if (fmt == VFHIP_FORMAT_NV12) return !(a1 & (is_output ? 3 : 1));
  return is_output ? !((a1 | a2) & 1) : true;
}

static inline Img make_img (const VfHipFrame *f)
{
  Img im {};
  for (int k = 0; k < 3; k++) { im.p[k] = (const uint8_t *) f->data[k]; im.s[k] = f->stride[k]; }
  im.w = f->info.width; im.h = f->info.height; im.fmt = f->info.format; im.m709 = f->info.color_matrix == VFHIP_MATRIX_BT709;
  return im;
}
static inline OutImg make_out (const VfHipFrame *f)
{
  OutImg im {};
  for (int k = 0; k < 3; k++) { im.p[k] = (uint8_t *) f->data[k]; im.s[k] = f->stride[k]; }
  im.w = f->info.width; im.h = f->info.height; im.fmt = f->info.format; im.m709 = f->info.color_matrix == VFHIP_MATRIX_BT709;
  return im;
}

}  // namespace metal
}  // namespace vfhip
