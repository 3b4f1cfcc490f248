// csrc/convertscale_ntap_kernels.h — method=bicubic of vfhipconvertscale: GstVideoScaler's 6-bit n-tap passes on 4 x u8
// pixels (videoscale method=catrom; rules and how they were pinned: oracle/gst114.c "videoscale method=catrom").
// The reference has no bicubic (convertscale/gstvfmetalconvertscale.m:81-85); north_star names it.
//
// Like GStreamer, the element converts at the INPUT size first (the existing gst-exact conversion kernels into an RGBA
// intermediate) and then scales in two separable passes over 8-bit lines, each out = clamp((sum p_l t_l + 32) >> 6):
//   k_ntap_v : one lane = one pixel (4 bytes) of an output row; the row's n taps (source row, weight) are wave-uniform
//   k_ntap_h : one lane = one output pixel; its n taps (source column, weight) come from the per-column table
// Correct-first kernels (three passes over HBM for a converting scale); the fused 2-tap kernels stay the headline path.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vfhip {

struct NtapParams {
  const uint8_t *in; int is;
  uint8_t *out; int os;
  int w, h;                 // size of the pass's output in pixels
  int n;                    // taps per output sample
  const int2 *tab;          // [out samples][n] of {source index, 6-bit weight}
};

// The sums are accumulated in f32: every term is an 8-bit sample times a 6-bit signed weight and |sum| < 2^15, far inside
// the 24-bit range in which f32 integer arithmetic is exact; v_cvt_f32_ubyteN extracts and converts a byte in one
// instruction and v_fma_f32 issues at full rate.
struct Acc4 { float a0, a1, a2, a3; };
__device__ __forceinline__ void ntap_acc (Acc4 &a, uint32_t v, int tap)
{
  const float t = (float) tap;
  a.a0 = fmaf ((float) (v & 0xff), t, a.a0); a.a1 = fmaf ((float) ((v >> 8) & 0xff), t, a.a1);
  a.a2 = fmaf ((float) ((v >> 16) & 0xff), t, a.a2); a.a3 = fmaf ((float) (v >> 24), t, a.a3);
}

// clamp ((sum + 32) >> 6, 0, 255) per channel, packed.  Done in f32 as well — floor (sum / 64 + .5) is exact, and
// v_cvt_pk_u8_f32 saturates and inserts the byte.  NOT written as integer shift + min/max: hipcc (ROCm 7.2) fuses that
// pattern into gfx950's v_ashr_pk_u8_i32 and then ORs the other two channels into the result's upper half, which the
// instruction does not clear (it kept the bits of its first source: every pixel whose first channel summed negative came
// out with 255 in the third) — found by the golden-vector tests.
__device__ __forceinline__ uint32_t ntap_finish (const Acc4 &a)
{
  uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32 (floorf (fmaf (a.a0, 0.015625f, 0.5f)), 0u, 0u);
  q = __builtin_amdgcn_cvt_pk_u8_f32 (floorf (fmaf (a.a1, 0.015625f, 0.5f)), 1u, q);
  q = __builtin_amdgcn_cvt_pk_u8_f32 (floorf (fmaf (a.a2, 0.015625f, 0.5f)), 2u, q);
  return __builtin_amdgcn_cvt_pk_u8_f32 (floorf (fmaf (a.a3, 0.015625f, 0.5f)), 3u, q);
}

__global__ __launch_bounds__ (256) void k_ntap_v (const NtapParams p)
{
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = __builtin_amdgcn_readfirstlane ((int) (blockIdx.y * 4 + threadIdx.y));      // one row per wave
  if (y >= p.h || x >= p.w) return;
  const int2 *t = p.tab + (size_t) y * p.n;
  Acc4 a = { 0.0f, 0.0f, 0.0f, 0.0f };
  for (int l = 0; l < p.n; l++) {
    const int2 e = t[l];
    ntap_acc (a, *reinterpret_cast<const uint32_t *> (p.in + (size_t) e.x * p.is + 4 * x), e.y);
  }
  *reinterpret_cast<uint32_t *> (p.out + (size_t) y * p.os + 4 * x) = ntap_finish (a);
}

__global__ __launch_bounds__ (256) void k_ntap_h (const NtapParams p)
{
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (y >= p.h || x >= p.w) return;
  const int2 *t = p.tab + (size_t) x * p.n;
  const uint8_t *row = p.in + (size_t) y * p.is;
  Acc4 a = { 0.0f, 0.0f, 0.0f, 0.0f };
  for (int l = 0; l < p.n; l++) {
    const int2 e = t[l];
    ntap_acc (a, *reinterpret_cast<const uint32_t *> (row + 4 * e.x), e.y);
  }
  *reinterpret_cast<uint32_t *> (p.out + (size_t) y * p.os + 4 * x) = ntap_finish (a);
}

// the tile kernel's form: the weight already a float (converted once when the tap table is staged in LDS), and — OPAQUE — no
// per-tap arithmetic on the fourth channel.  A source without alpha converts to A = 255 in every pixel, so every input of one pass's
// sum carries the SAME alpha `ain` (255 in the first pass; in the second, the first pass's value for that row / column, which depends
// only on the first pass's tap sum) and the filtered alpha is ((ain * sum of the taps) + 32) >> 6.  The tap sum is carried instead of
// assumed: GStreamer's 6-bit taps do not always sum to 64 (at 3:1 some columns sum to 63 and videoscale really outputs A = 251 there —
// found by tools/fuzz_gst_exact.py against the pinned oracle), so a constant 255 would be wrong.
template <bool OPAQUE>
__device__ __forceinline__ void ntap_accf (Acc4 &a, uint32_t v, float t)
{
  a.a0 = fmaf ((float) (v & 0xff), t, a.a0); a.a1 = fmaf ((float) ((v >> 8) & 0xff), t, a.a1);
  a.a2 = fmaf ((float) ((v >> 16) & 0xff), t, a.a2);
  if (OPAQUE) a.a3 += t; else a.a3 = fmaf ((float) (v >> 24), t, a.a3);
}
// `ain`: OPAQUE only — the alpha every input of this sum carried (see above)
template <bool OPAQUE>
__device__ __forceinline__ uint32_t ntap_finishf (const Acc4 &a, float ain)
{
  uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32 (floorf (fmaf (a.a0, 0.015625f, 0.5f)), 0u, 0u);
  q = __builtin_amdgcn_cvt_pk_u8_f32 (floorf (fmaf (a.a1, 0.015625f, 0.5f)), 1u, q);
  q = __builtin_amdgcn_cvt_pk_u8_f32 (floorf (fmaf (a.a2, 0.015625f, 0.5f)), 2u, q);
  return __builtin_amdgcn_cvt_pk_u8_f32 (floorf (fmaf (OPAQUE ? a.a3 * ain : a.a3, 0.015625f, 0.5f)), 3u, q);
}

// ---- k_cs_cubic_tile: conversion + both n-tap passes fused per output tile ----------------------------------------------
// One workgroup = a 64 x 16 tile of output pixels.  The source region the tile's taps reach (for 2:1: 135 x 39 pixels) is
// converted ONCE into LDS as 8-bit RGBA (cs_tap: the gst-exact per-pixel conversion of k_cs_generic — GStreamer converts at
// the input size first), the first pass runs LDS -> LDS, the second LDS -> HBM, in GstVideoScaler's order; the 8-bit
// rounding between the passes is kept, so the result is the three-pass path's bit for bit while every input byte is read
// about 1.3 times and nothing intermediate touches HBM.  The host falls back to the three-pass path when a tile's source
// region does not fit the LDS arrays (down-scales beyond ~2.1:1 horizontally or ~2.5:1 vertically).
constexpr int CT_TW = 64, CT_TH = 16, CT_RW = 152, CT_RH = 48, CT_MAXN = 12;       // CT_RW: a multiple of 8 (aligned 8-column groups) with room for the alignment slack

// Eight adjacent source pixels (columns gx .. gx+7, gx % 8 == 0) of NV12 row cy, converted with the packed ORC pipeline of
// the 2-tap kernels (load_craw / hfilter / orc_pair: one 8-byte luma load, two chroma rows with their neighbour pairs)
// instead of eight scalar cs_tap calls.  Same arithmetic (horizontal then vertical chroma filter, mulhs matrix).
// Requires 8-byte aligned planes / strides and gx + 8 <= in_w.
template <bool COSITED>
__device__ __forceinline__ void cs_convert8_nv12 (const CsParams &p, const uint8_t *const in[3], int gx, int cy, uint32_t out[8])
{
  const int cw = (p.in_w + 1) >> 1, chh = (p.in_h + 1) >> 1;
  const int j = cy >> 1, jn = (cy & 1) ? min (j + 1, chh - 1) : max (j - 1, 0);
  // neighbour pairs, edge-replicated: the right one of the last chroma group / the left one of the first are their own
  const uint32_t roff = (gx / 2 + 4 < cw) ? 8u : 6u, loff = gx > 0 ? 2u : 0u;
  const CRow a = hfilter<COSITED> (load_craw<COSITED> (in[1], (uint32_t) j * (uint32_t) p.is[1] + (uint32_t) gx, roff, loff));
  const CRow b = hfilter<COSITED> (load_craw<COSITED> (in[1], (uint32_t) jn * (uint32_t) p.is[1] + (uint32_t) gx, roff, loff));
  const uint32_t X = 0x80808080u;
  const uint32_t e01 = filt31_u8 (a.e01, b.e01) ^ X, e23 = filt31_u8 (a.e23, b.e23) ^ X, o01 = filt31_u8 (a.o01, b.o01) ^ X, o23 = filt31_u8 (a.o23, b.o23) ^ X;
  const uint2 yv = *reinterpret_cast<const uint2 *> (in[0] + (size_t) cy * p.is[0] + gx);
  const uint32_t y0 = yv.x ^ X, y1 = yv.y ^ X;
  const int bias = 128 << 16;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const uint32_t sel = (n & 1) ? 0x03030202u : 0x01010000u;
    const uint32_t yn = n < 2 ? y0 : y1, en = n < 2 ? e01 : e23, on = n < 2 ? o01 : o23;
    uint32_t bb, gg, rr;
    orc_pair (perm_b32 (0u, yn, sel), perm_b32 (0u, en, sel), perm_b32 (0u, on, sel), p.c, bias, bb, gg, rr);
    const uint32_t x = p.out_rgba ? rr : bb, z = p.out_rgba ? bb : rr;        // byte 0 / byte 2 channel
    const uint32_t xg = perm_b32 (gg, x, 0x05010400u);                          // [x_e, g_e, x_o, g_o]
    const uint32_t za = perm_b32 (0xffffffffu, z, 0x07010700u);                 // [z_e, ff, z_o, ff]
    out[2 * n] = perm_b32 (za, xg, 0x05040100u);                                // [x_e, g_e, z_e, ff]
    out[2 * n + 1] = perm_b32 (za, xg, 0x07060302u);                            // [x_o, g_o, z_o, ff]
  }
}

// cs_convert8_nv12 for k_cs_cubic_dot (load and arithmetic apart, so that the next item's loads can be in flight): the same eight pixels as three CHANNEL dwords per four pixels (lo: pixels 0-3, hi: 4-7; channel index = byte
// position in the output pixel) straight from orc_pair's planar [even, odd] byte pairs — the RGBA interleave above and the de-interleave the planes
// would need cancel (32 v_perm per eight pixels); alpha is not produced (a source without alpha: the kernel derives A from the tap sums)
struct Cv8Raw { CRaw a, b; uint2 y; };           // the raw bytes behind eight NV12 pixels: chroma rows j and its vertical neighbour, the luma
template <bool COSITED>
__device__ __forceinline__ Cv8Raw cs_load8_nv12 (const CsParams &p, const uint8_t *const in[3], int gx, int cy)
{
  const int cw = (p.in_w + 1) >> 1, chh = (p.in_h + 1) >> 1;
  const int j = cy >> 1, jn = (cy & 1) ? min (j + 1, chh - 1) : max (j - 1, 0);
  const uint32_t roff = (gx / 2 + 4 < cw) ? 8u : 6u, loff = gx > 0 ? 2u : 0u;
  Cv8Raw r;
  r.a = load_craw<COSITED> (in[1], (uint32_t) j * (uint32_t) p.is[1] + (uint32_t) gx, roff, loff);
  r.b = load_craw<COSITED> (in[1], (uint32_t) jn * (uint32_t) p.is[1] + (uint32_t) gx, roff, loff);
  r.y = *reinterpret_cast<const uint2 *> (in[0] + (size_t) cy * p.is[0] + gx);
  return r;
}
template <bool COSITED>
__device__ __forceinline__ void cs_convert8_nv12_planar (const CsParams &p, const Cv8Raw &raw, uint32_t lo[3], uint32_t hi[3])
{
  const CRow a = hfilter<COSITED> (raw.a), b = hfilter<COSITED> (raw.b);
  const uint32_t X = 0x80808080u;
  const uint32_t e01 = filt31_u8 (a.e01, b.e01) ^ X, e23 = filt31_u8 (a.e23, b.e23) ^ X, o01 = filt31_u8 (a.o01, b.o01) ^ X, o23 = filt31_u8 (a.o23, b.o23) ^ X;
  const uint32_t y0 = raw.y.x ^ X, y1 = raw.y.y ^ X;
  const int bias = 128 << 16;
  uint32_t bb[4], gg[4], rr[4];
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const uint32_t sel = (n & 1) ? 0x03030202u : 0x01010000u;
    const uint32_t yn = n < 2 ? y0 : y1, en = n < 2 ? e01 : e23, on = n < 2 ? o01 : o23;
    orc_pair (perm_b32 (0u, yn, sel), perm_b32 (0u, en, sel), perm_b32 (0u, on, sel), p.c, bias, bb[n], gg[n], rr[n]);
  }
  const uint32_t *x = p.out_rgba ? rr : bb, *z = p.out_rgba ? bb : rr;          // byte 0 / byte 2 channel
  lo[0] = perm_b32 (x[1], x[0], 0x05040100u); hi[0] = perm_b32 (x[3], x[2], 0x05040100u);
  lo[1] = perm_b32 (gg[1], gg[0], 0x05040100u); hi[1] = perm_b32 (gg[3], gg[2], 0x05040100u);
  lo[2] = perm_b32 (z[1], z[0], 0x05040100u); hi[2] = perm_b32 (z[3], z[2], 0x05040100u);
}

// the same for I420: GStreamer's I420 fast path replicates the chroma sample of row cy >> 1 to its two columns and two rows (nearest:
// oracle/gst114.c gst114_yuv420_to_rgb, planar = 1), so a group is one 8-byte luma load and one 4-byte load from each chroma plane.
// Requires 8-byte aligned luma and 4-byte aligned chroma planes / strides, gx % 8 == 0 and gx + 8 <= in_w.
__device__ __forceinline__ void cs_convert8_i420 (const CsParams &p, const uint8_t *const in[3], int gx, int cy, uint32_t out[8])
{
  const uint32_t X = 0x80808080u;
  const size_t j = (size_t) (cy >> 1);
  const uint32_t U4 = *reinterpret_cast<const uint32_t *> (in[1] + j * p.is[1] + (gx >> 1)) ^ X;
  const uint32_t V4 = *reinterpret_cast<const uint32_t *> (in[2] + j * p.is[2] + (gx >> 1)) ^ X;
  const uint2 yv = *reinterpret_cast<const uint2 *> (in[0] + (size_t) cy * p.is[0] + gx);
  const uint32_t y0 = yv.x ^ X, y1 = yv.y ^ X;
  const int bias = 128 << 16;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const uint32_t sel = (n & 1) ? 0x03030202u : 0x01010000u;
    const uint32_t uv = perm_b32 (V4, U4, (uint32_t) (((4 + n) << 24) | ((4 + n) << 16) | (n << 8) | n));       // [U_n U_n V_n V_n]
    uint32_t bb, gg, rr;
    orc_pair (perm_b32 (0u, n < 2 ? y0 : y1, sel), uv, uv, p.c, bias, bb, gg, rr);
    const uint32_t x = p.out_rgba ? rr : bb, z = p.out_rgba ? bb : rr;        // byte 0 / byte 2 channel
    const uint32_t xg = perm_b32 (gg, x, 0x05010400u);                          // [x_e, g_e, x_o, g_o]
    const uint32_t za = perm_b32 (0xffffffffu, z, 0x07010700u);                 // [z_e, ff, z_o, ff]
    out[2 * n] = perm_b32 (za, xg, 0x05040100u);                                // [x_e, g_e, z_e, ff]
    out[2 * n + 1] = perm_b32 (za, xg, 0x07060302u);                            // [x_o, g_o, z_o, ff]
  }
}

// ... and for packed 4:2:2 (UYVY: U Y0 V Y1, YUY2: Y0 U Y1 V): eight pixels are one 16-byte load; the chroma pairs and the luma bytes
// come apart with four byte permutes, the row's chroma is up-sampled horizontally with NV12's rule (hfilter; there is no vertical
// step: oracle/gst114.c gst114_packed422_to_rgb), then the ORC matrix.  The neighbour chroma pairs are the macro-pixels 4 bytes before
// and 16 bytes after the group, replicated at the ends of the row.  Requires 16-byte aligned rows, gx % 8 == 0 and gx + 8 <= in_w.
template <bool YUY2, bool COSITED>
__device__ __forceinline__ void cs_convert8_packed (const CsParams &p, const uint8_t *const in[3], int gx, int cy, uint32_t out[8])
{
  const uint32_t X = 0x80808080u;
  const uint8_t *row = in[0] + (size_t) cy * p.is[0] + 2 * gx;
  const uint4 m = *reinterpret_cast<const uint4 *> (row);
  const uint32_t csel = YUY2 ? 0x07050301u : 0x06040200u, ysel = YUY2 ? 0x06040200u : 0x07050301u, psel = YUY2 ? 0x0c0c0301u : 0x0c0c0200u;
  CRaw r;
  r.v.x = perm_b32 (m.y, m.x, csel); r.v.y = perm_b32 (m.w, m.z, csel);              // [U0V0U1V1][U2V2U3V3]
  const bool last = gx + 8 >= p.in_w, first = gx == 0;
  const uint32_t rd = *reinterpret_cast<const uint32_t *> (row + (last ? 12 : 16)), ld = *reinterpret_cast<const uint32_t *> (row - (first ? 0 : 4));
  r.right = perm_b32 (0u, rd, psel);                                                 // the next macro-pixel's (U, V); the row's last one is its own
  r.left = perm_b32 (0u, ld, psel);
  const CRow c = hfilter<COSITED> (r);
  const uint32_t e01 = c.e01 ^ X, e23 = c.e23 ^ X, o01 = c.o01 ^ X, o23 = c.o23 ^ X;
  const uint32_t y0 = perm_b32 (m.y, m.x, ysel) ^ X, y1 = perm_b32 (m.w, m.z, ysel) ^ X;
  const int bias = 128 << 16;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const uint32_t sel = (n & 1) ? 0x03030202u : 0x01010000u;
    const uint32_t yn = n < 2 ? y0 : y1, en = n < 2 ? e01 : e23, on = n < 2 ? o01 : o23;
    uint32_t bb, gg, rr;
    orc_pair (perm_b32 (0u, yn, sel), perm_b32 (0u, en, sel), perm_b32 (0u, on, sel), p.c, bias, bb, gg, rr);
    const uint32_t x = p.out_rgba ? rr : bb, z = p.out_rgba ? bb : rr;        // byte 0 / byte 2 channel
    const uint32_t xg = perm_b32 (gg, x, 0x05010400u);                          // [x_e, g_e, x_o, g_o]
    const uint32_t za = perm_b32 (0xffffffffu, z, 0x07010700u);                 // [z_e, ff, z_o, ff]
    out[2 * n] = perm_b32 (za, xg, 0x05040100u);                                // [x_e, g_e, z_e, ff]
    out[2 * n + 1] = perm_b32 (za, xg, 0x07060302u);                            // [x_o, g_o, z_o, ff]
  }
}

// ---- k_cs_yuv_same<FMT, COSITED>: NV12 / I420 / UYVY / YUY2 -> BGRA / RGBA at the SAME size -----------------------------------------------------
// The element as a plain converter (a decoder's NV12 / I420 to RGB for display or inference, no scaling): videoscale passes through and
// what is left is videoconvert's chroma up-sampling and matrix.  k_cs_taps ran this shape as a 2-tap scale with unit weights — four
// conversions per output pixel, byte-wise: 12.1 us for a 1080p NV12 frame, 49.5 us for 2160p (0.12 of the roofline).  Here one lane =
// eight adjacent pixels of one row: cs_convert8_nv12 (one 8-byte luma load, the two chroma rows of the pixel's vertical filter with
// their neighbour pairs, the packed ORC pipeline of the 2:1 kernel), cs_convert8_i420 or cs_convert8_packed, and two 16-byte non-temporal stores.  A
// wave covers 512 consecutive pixels of a row; the chroma rows are fetched by the luma rows that lean on them (L2 hits).
// Contract (checked by the host): width % 8 == 0, 8-byte aligned planes / strides / pitch (I420 chroma: 4-byte; packed: 16-byte), 16-byte aligned output.
// FMT: 0 NV12, 1 I420, 2 UYVY, 3 YUY2
template <int FMT, bool COSITED>
__global__ __launch_bounds__ (256) void k_cs_yuv_same (const CsParams p)
{
  const int groups = p.in_w >> 3;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * p.in_h) return;
  const int row = t / groups, g = t - row * groups;
  const size_t fo = (size_t) blockIdx.y * p.in_pitch;
  const uint8_t *in[3] = { p.in[0] + fo, FMT <= 1 ? p.in[1] + fo : nullptr, FMT == 1 ? p.in[2] + fo : nullptr };
  uint32_t px[8];
  if (FMT == 1) cs_convert8_i420 (p, in, 8 * g, row, px);
  else if (FMT == 0) cs_convert8_nv12<COSITED> (p, in, 8 * g, row, px);
  else cs_convert8_packed<FMT == 3, COSITED> (p, in, 8 * g, row, px);
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  v4u *d = reinterpret_cast<v4u *> (p.out + (size_t) blockIdx.y * p.out_pitch + (size_t) row * p.os) + 2 * g;
  const v4u a = { px[0], px[1], px[2], px[3] }, b = { px[4], px[5], px[6], px[7] };
  __builtin_nontemporal_store (a, d);
  __builtin_nontemporal_store (b, d + 1);
}

// ---- k_cs_rgb_same: BGRA / RGBA -> BGRA / RGBA at the SAME size (a copy, or the R <-> B swap videoconvert makes of it; videoscale passes through
// whatever the method): 16 bytes per lane each way.  The shape reached k_cs_bilinear_tile (7.4 us per 1080p frame for what is a 16.6 MB copy).
// Contract (host): width % 4 == 0, 16-byte aligned rows / pitches on both sides.
__global__ __launch_bounds__ (256) void k_cs_rgb_same (const CsParams p, int swap)
{
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  const int groups = p.in_w >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= groups * p.in_h) return;
  const int row = t / groups, g = t - row * groups;
  v4u v = *(reinterpret_cast<const v4u *> (p.in[0] + (size_t) blockIdx.y * p.in_pitch + (size_t) row * p.is[0]) + g);
  if (swap) {
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = perm_b32 (0u, v[k], 0x03000102u);
  }
  __builtin_nontemporal_store (v, reinterpret_cast<v4u *> (p.out + (size_t) blockIdx.y * p.out_pitch + (size_t) row * p.os) + g);
}

struct CubicTileParams {
  CsParams cs;                       // input planes / strides / matrix / formats for cs_tap; in_pitch, out_pitch for batches
  uint8_t *out; int os;
  int ow, oh, nh, nv, vfirst;
  int fast_nv12;                     // NV12 input with 8-byte aligned planes / strides: convert in 8-column groups (cs_convert8_nv12)
  const int2 *tab_h, *tab_v;         // [ow][nh], [oh][nv] of {source index, 6-bit weight}; nh / nv == 0: no scaling on that axis
  int tiles_x, tiles_y, n_tiles, n_chunk;   // the launch: tiles per row / column of a frame, tiles in all, tiles per XCD (ceil (n_tiles / 8))
};

template <int THREADS, bool OPAQUE>
__global__ __launch_bounds__ (THREADS) void k_cs_cubic_tile (const CubicTileParams p)
{
  __shared__ __attribute__ ((aligned (16))) uint32_t reg[CT_RH][CT_RW];   // converted source region
  __shared__ uint32_t tmp[CT_RH * CT_TW];                             // first-pass result: [CT_TH][rw] (V first) or [rh][CT_TW] (H first)
  __shared__ int2 lth[CT_TW * CT_MAXN], ltv[CT_TH * CT_MAXN];         // this tile's tap tables (LDS reads instead of per-lane global loads)
  const int tid = threadIdx.x;
  // XCD-aware tile order (1-D grid of 8 * n_chunk blocks): blocks b and b + 8 share an XCD and its L2, so block b takes tile (b % 8) * n_chunk + b / 8
  // — each XCD one contiguous run of tiles (x fastest, then y, then frame).  Neighbouring tiles share the source lines their regions overlap in and
  // both halves of every 128-byte line a 64-pixel-wide tile covers half of; dealt out round-robin they sat in different L2s and the input was fetched
  // 3.4 times (profiles/r03s_pmc_elements_summary.json).  Speed only: any mapping computes the same bytes.
  const int bt = (int) (blockIdx.x & 7u) * p.n_chunk + (int) (blockIdx.x >> 3);
  if (bt >= p.n_tiles) return;
  const int per = p.tiles_x * p.tiles_y, bz = bt / per, brem = bt - bz * per, by = brem / p.tiles_x, bx = brem - by * p.tiles_x;
  const int x0 = bx * CT_TW, y0 = by * CT_TH;
  const int tw = min (CT_TW, p.ow - x0), th = min (CT_TH, p.oh - y0);
  const uint8_t *in[3] = { p.cs.in[0] + (size_t) bz * p.cs.in_pitch,
                           p.cs.in[1] ? p.cs.in[1] + (size_t) bz * p.cs.in_pitch : nullptr,
                           p.cs.in[2] ? p.cs.in[2] + (size_t) bz * p.cs.in_pitch : nullptr };
  uint8_t *out = p.out + (size_t) bz * p.cs.out_pitch;
  // source region of this tile (tables hold absolute, edge-clamped, non-decreasing indices)
  int cx0 = p.nh ? p.tab_h[(size_t) x0 * p.nh].x : x0;
  const int cx1 = p.nh ? p.tab_h[(size_t) (x0 + tw - 1) * p.nh + p.nh - 1].x : x0 + tw - 1;
  const int ry0 = p.nv ? p.tab_v[(size_t) y0 * p.nv].x : y0, ry1 = p.nv ? p.tab_v[(size_t) (y0 + th - 1) * p.nv + p.nv - 1].x : y0 + th - 1;
  const int rh = ry1 - ry0 + 1;
  int rw = cx1 - cx0 + 1;
  // horizontal taps tap-major ([l][tx]): the lanes of a wave are consecutive tx, so tap l of 64 columns is 64 consecutive int2 —
  // column-major ([tx][l], 8 * nh bytes apart) put every lane of a wave on the same four LDS banks
  // staged as { index relative to the region, weight as float bits }: no subtraction and no int -> float conversion per tap
  const int ga0 = p.fast_nv12 ? (cx0 & ~7) : cx0;                      // the region's first column (see below)
  for (int i = tid; i < tw * p.nh; i += THREADS) {
    const int tx = i / p.nh, l = i - tx * p.nh;
    const int2 e = p.tab_h[(size_t) x0 * p.nh + i];
    lth[l * CT_TW + tx] = make_int2 (e.x - ga0, __float_as_int ((float) e.y));
  }
  for (int i = tid; i < th * p.nv; i += THREADS) { const int2 e = p.tab_v[(size_t) y0 * p.nv + i]; ltv[i] = make_int2 (e.x - ry0, __float_as_int ((float) e.y)); }
  if (p.fast_nv12) {
    // region widened to whole 8-column groups; groups that would cross the right image edge fall back to cs_tap
    const int ga = cx0 & ~7, groups = ((cx1 + 1 - ga) + 7) >> 3;
    for (int i = tid; i < groups * rh; i += THREADS) {
      const int ry = i / groups, g = i - ry * groups, gx = ga + 8 * g;
      uint32_t px8[8];
      if (gx + 8 <= p.cs.in_w) {
        if (p.cs.cosited) cs_convert8_nv12<true> (p.cs, in, gx, ry0 + ry, px8);
        else cs_convert8_nv12<false> (p.cs, in, gx, ry0 + ry, px8);
      } else {
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
          int px[4];
          cs_tap (p.cs, in, min (gx + k, p.cs.in_w - 1), ry0 + ry, px);
          px8[k] = (uint32_t) px[0] | ((uint32_t) px[1] << 8) | ((uint32_t) px[2] << 16) | ((uint32_t) px[3] << 24);
        }
      }
      uint4 *d = reinterpret_cast<uint4 *> (&reg[ry][8 * g]);
      d[0] = make_uint4 (px8[0], px8[1], px8[2], px8[3]); d[1] = make_uint4 (px8[4], px8[5], px8[6], px8[7]);
    }
    cx0 = ga; rw = 8 * groups;
  } else {
    for (int i = tid; i < rw * rh; i += THREADS) {
      const int ry = i / rw, rx = i - ry * rw;
      int px[4];
      cs_tap (p.cs, in, cx0 + rx, ry0 + ry, px);
      reg[ry][rx] = (uint32_t) px[0] | ((uint32_t) px[1] << 8) | ((uint32_t) px[2] << 16) | ((uint32_t) px[3] << 24);
    }
  }
  __syncthreads ();
  if (p.nh && p.nv) {
    if (p.vfirst) {                                                    // tmp[ty][rx] = vertical taps over the region's columns
      for (int i = tid; i < th * rw; i += THREADS) {
        const int ty = i / rw, rx = i - ty * rw;
        const int2 *t = ltv + ty * p.nv;
        Acc4 a = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (int l = 0; l < p.nv; l++) ntap_accf<OPAQUE> (a, reg[t[l].x][rx], __int_as_float (t[l].y));
        tmp[ty * rw + rx] = ntap_finishf<OPAQUE> (a, 255.0f);
      }
    } else {                                                           // tmp[ry][tx] = horizontal taps over the region's rows
      for (int i = tid; i < rh * tw; i += THREADS) {
        const int ry = i / tw, tx = i - ry * tw;
        const int2 *t = lth + tx;
        Acc4 a = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (int l = 0; l < p.nh; l++) ntap_accf<OPAQUE> (a, reg[ry][t[l * CT_TW].x], __int_as_float (t[l * CT_TW].y));
        tmp[ry * CT_TW + tx] = ntap_finishf<OPAQUE> (a, 255.0f);
      }
    }
    __syncthreads ();
  }
  for (int i = tid; i < th * tw; i += THREADS) {
    const int ty = i / tw, tx = i - ty * tw;
    Acc4 a = { 0.0f, 0.0f, 0.0f, 0.0f };
    uint32_t q;
    if (p.nh && p.nv) {
      float ain;
      if (p.vfirst) {
        const int2 *t = lth + tx;
        ain = (float) (tmp[ty * rw + t[0].x] >> 24);
        for (int l = 0; l < p.nh; l++) ntap_accf<OPAQUE> (a, tmp[ty * rw + t[l * CT_TW].x], __int_as_float (t[l * CT_TW].y));
      } else {
        const int2 *t = ltv + ty * p.nv;
        ain = (float) (tmp[t[0].x * CT_TW + tx] >> 24);
        for (int l = 0; l < p.nv; l++) ntap_accf<OPAQUE> (a, tmp[t[l].x * CT_TW + tx], __int_as_float (t[l].y));
      }
      q = ntap_finishf<OPAQUE> (a, ain);
    } else if (p.nh) {
      const int2 *t = lth + tx;
      for (int l = 0; l < p.nh; l++) ntap_accf<OPAQUE> (a, reg[ty][t[l * CT_TW].x], __int_as_float (t[l * CT_TW].y));
      q = ntap_finishf<OPAQUE> (a, 255.0f);
    } else if (p.nv) {
      const int2 *t = ltv + ty * p.nv;
      for (int l = 0; l < p.nv; l++) ntap_accf<OPAQUE> (a, reg[t[l].x][tx], __int_as_float (t[l].y));
      q = ntap_finishf<OPAQUE> (a, 255.0f);
    } else q = reg[ty][tx];
    *reinterpret_cast<uint32_t *> (out + (size_t) (y0 + ty) * p.os + 4 * (x0 + tx)) = q;
  }
}

// ---- k_cs_cubic_dot: the tile kernel with the passes as int8 dot products (round 3) ---------------------------------------------------------
// k_cs_cubic_tile evaluates a tap as three v_cvt_f32_ubyte + three v_fma (21 issue cycles per tap and pixel) and reads its sample AND its table
// entry from LDS for every tap: ~210 of its ~520 instructions per output pixel.  A pass is sum (sample * 6-bit tap) over a WINDOW of adjacent
// samples of one channel, so here the converted region lives in LDS as byte PLANES (one per channel, row-major, samples biased by -128 to int8)
// and a pass is v_dot4c_i32_i8 on four samples at a time:
//   * the host turns each output's tap list {index, weight} into a window: start s (the first tap's index), W = 4, 8 or 12 weights as int8 (taps
//     that the edge clamp put on the same sample merged, the rest 0), and the tap sum; with acc0 = 128 * sum + 32 the dot products give
//     sum (sample * tap) + 32 exactly, then >> 6 and saturate to a byte — GstVideoScaler's clamp ((sum + 32) >> 6), bit for bit;
//   * horizontal pass: a window starts at any byte: W / 4 + 1 dwords, v_alignbyte_b32 by s & 3, W / 4 dot products (16 cycles per channel at W = 8);
//   * vertical pass: four adjacent columns at once — one dword per window row, a 4 x 4 byte transpose (8 v_perm) per four rows turns them into
//     four column dwords, four dot products (24 cycles per channel and pixel at W = 8);
//   * the conversion (cs_convert8_nv12's eight pixels of a row, or cs_tap) is de-interleaved with the same 4 x 4 transpose and written as one
//     8-byte store per channel; the first pass writes the planes the second reads (four results per dword store), the second pass writes pixels.
// A source without alpha converts to A = 255 everywhere: three planes, and A of the output from the two tap sums (the rule of k_cs_cubic_tile:
// GStreamer's taps do not always sum to 64).  Same results as k_cs_cubic_tile and the three-pass path (tests compare all three with the real
// element's vectors); used when every tile's windows fit the planes and every merged weight fits int8 (else k_cs_cubic_tile).
constexpr int CD_TW = 64, CD_TH = 16;
constexpr int CD_RW = 176, CD_RH = 52;            // plane row stride in bytes (a multiple of 8) / plane rows: the region of a tile incl. alignment and window slack
constexpr int CD_P2 = CD_RH * CD_TW;              // first-pass result per channel: [CD_TH][CD_RW] (V first) or [CD_RH][CD_TW] (H first), whichever is larger
constexpr int CD_WT = 6;                          // dwords per output in a window table: start, tap sum, three dwords of int8 weights, alpha of an opaque source after this pass
static_assert (CD_TH * CD_RW <= CD_P2 && CD_RW % 8 == 0, "plane geometry");

struct CubicDotParams {
  CsParams cs;                       // input planes / strides / matrix / formats for the conversion; in_pitch, out_pitch for batches
  uint8_t *out; int os;
  int ow, oh, vfirst, fast_nv12;
  const uint32_t *win_h, *win_v;     // [ow][CD_WT], [oh][CD_WT]
  int wh, wv;                        // window widths: 4, 8 or 12
  int tiles_x, tiles_y, n_tiles, n_chunk;
};

// 4 x 4 byte transpose: (a, b, c, d) = four dwords of four bytes -> o[j] = { a.byte j, b.byte j, c.byte j, d.byte j }.  Pixels -> channel dwords,
// window rows -> column dwords, channel dwords -> pixels: the same eight v_perm
__device__ __forceinline__ void cd_transpose4 (uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t o[4])
{
  const uint32_t t0 = perm_b32 (b, a, 0x05010400u), t1 = perm_b32 (b, a, 0x07030602u), t2 = perm_b32 (d, c, 0x05010400u), t3 = perm_b32 (d, c, 0x07030602u);
  o[0] = perm_b32 (t2, t0, 0x05040100u); o[1] = perm_b32 (t2, t0, 0x07060302u); o[2] = perm_b32 (t3, t1, 0x05040100u); o[3] = perm_b32 (t3, t1, 0x07060302u);
}
// four sums -> four bytes clamp (sum >> 6): 16-bit pairs through v_sat_pk_u8_i16 (|sum >> 6| < 2^15 by far: |sum| < 2^15 already)
__device__ __forceinline__ uint32_t cd_pack4 (int a0, int a1, int a2, int a3)
{
  const uint32_t p01 = sat_pk_u8_i16 (perm_b32 ((uint32_t) (a1 >> 6), (uint32_t) (a0 >> 6), 0x05040100u));
  const uint32_t p23 = sat_pk_u8_i16 (perm_b32 ((uint32_t) (a3 >> 6), (uint32_t) (a2 >> 6), 0x05040100u));
  return perm_b32 (p23, p01, 0x05040100u);
}
// window along a row: `row` = the plane row, wt = the output's table entry (start relative to the row's first byte); NG = W / 4 dwords of weights
template <int NG>
__device__ __forceinline__ int cd_hwin_n (const uint8_t *row, const uint32_t *wt)
{
  const uint32_t s = wt[0];
  const uint32_t *d = reinterpret_cast<const uint32_t *> (row + (s & ~3u));
  int acc = (int) wt[1] * 128 + 32;
  uint32_t lo = d[0];
#pragma unroll
  for (int g = 0; g < NG; g++) {
    const uint32_t hi = d[g + 1];
    acc = __builtin_amdgcn_sdot4 ((int) alignbyte (hi, lo, s & 3u), (int) wt[2 + g], acc, false);
    lo = hi;
  }
  return acc;
}
__device__ __forceinline__ int cd_hwin (const uint8_t *row, const uint32_t *wt, int W)        // W is wave-uniform: one scalar branch
{
  return W == 8 ? cd_hwin_n<2> (row, wt) : (W == 4 ? cd_hwin_n<1> (row, wt) : cd_hwin_n<3> (row, wt));
}
// window down four adjacent columns (byte offset col4, a multiple of 4) of a plane with `stride` bytes per row
template <int NG>
__device__ __forceinline__ void cd_vwin4_n (const uint8_t *plane, int stride, int col4, const uint32_t *wt, int acc[4])
{
  const uint8_t *r = plane + (int) wt[0] * stride + col4;
  acc[0] = acc[1] = acc[2] = acc[3] = (int) wt[1] * 128 + 32;
#pragma unroll
  for (int g = 0; g < NG; g++) {
    uint32_t col[4];
    cd_transpose4 (*reinterpret_cast<const uint32_t *> (r), *reinterpret_cast<const uint32_t *> (r + stride),
                   *reinterpret_cast<const uint32_t *> (r + 2 * stride), *reinterpret_cast<const uint32_t *> (r + 3 * stride), col);
    const int w = (int) wt[2 + g];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_sdot4 ((int) col[j], w, acc[j], false);
    r += 4 * stride;
  }
}
__device__ __forceinline__ void cd_vwin4 (const uint8_t *plane, int stride, int col4, const uint32_t *wt, int W, int acc[4])
{
  if (W == 8) cd_vwin4_n<2> (plane, stride, col4, wt, acc);
  else if (W == 4) cd_vwin4_n<1> (plane, stride, col4, wt, acc);
  else cd_vwin4_n<3> (plane, stride, col4, wt, acc);
}

// i / d for the kernel's item indices without an integer division (~20 instructions): exact for d <= 64-ish divisors and quotients < 1000 —
// (i + .5) * fl (1 / d) is off by < 1e-4 there and (i + .5) / d is at least .5 / d away from every integer
__device__ __forceinline__ int cd_div (int i, float inv) { return (int) (((float) i + 0.5f) * inv); }

// ONLY_FAST (host: NV12 through the packed pipeline and in_w % 8 == 0, so that every 8-pixel group is inside the frame or outside it): the instantiation
// without cs_tap's general conversion — with both in one kernel the lane index spilled to scratch at 64 VGPRs (four workgroups per CU)
template <int NCH, bool ONLY_FAST>
__global__ __launch_bounds__ (512, ONLY_FAST ? 8 : 6) void k_cs_cubic_dot (const CubicDotParams p)
{
  __shared__ __attribute__ ((aligned (16))) uint8_t P1[NCH][CD_RH * CD_RW + 16];    // converted region, one plane per channel (byte order of the output pixel)
  __shared__ __attribute__ ((aligned (16))) uint8_t P2[NCH][CD_P2 + 16];            // first-pass result
  __shared__ uint32_t lwh[CD_TW][CD_WT], lwv[CD_TH][CD_WT];
  const int tid = threadIdx.x;
  const int bt = (int) (blockIdx.x & 7u) * p.n_chunk + (int) (blockIdx.x >> 3);       // XCD-aware tile order (k_cs_cubic_tile)
  if (bt >= p.n_tiles) return;
  const int per = p.tiles_x * p.tiles_y, bz = bt / per, brem = bt - bz * per, by = brem / p.tiles_x, bx = brem - by * p.tiles_x;
  const int x0 = bx * CD_TW, y0 = by * CD_TH;
  const int tw = min (CD_TW, p.ow - x0), th = min (CD_TH, p.oh - y0);
  const uint8_t *in[3] = { p.cs.in[0] + (size_t) bz * p.cs.in_pitch,
                           p.cs.in[1] ? p.cs.in[1] + (size_t) bz * p.cs.in_pitch : nullptr,
                           p.cs.in[2] ? p.cs.in[2] + (size_t) bz * p.cs.in_pitch : nullptr };
  uint8_t *out = p.out + (size_t) bz * p.cs.out_pitch;
  // the region: columns from the first window's start (down to a multiple of 8: whole conversion groups, dword-aligned windows) to the last window's
  // end, rows likewise; the tables staged with starts relative to it (entries beyond a partial tile repeat the last one: harmless work)
  const int ga = (int) p.win_h[(size_t) x0 * CD_WT] & ~7, ry0 = (int) p.win_v[(size_t) y0 * CD_WT];
  const int cols = (int) p.win_h[(size_t) (x0 + tw - 1) * CD_WT] + p.wh - ga, rows = (int) p.win_v[(size_t) (y0 + th - 1) * CD_WT] + p.wv - ry0;
  const int groups = (cols + 7) >> 3;
  for (int i = tid; i < CD_TW * CD_WT; i += 512) {
    const int tx = i / CD_WT, k = i - tx * CD_WT;
    const uint32_t v = p.win_h[(size_t) (x0 + min (tx, tw - 1)) * CD_WT + k];
    lwh[tx][k] = k ? v : v - (uint32_t) ga;
  }
  for (int i = tid; i < CD_TH * CD_WT; i += 512) {
    const int ty = i / CD_WT, k = i - ty * CD_WT;
    const uint32_t v = p.win_v[(size_t) (y0 + min (ty, th - 1)) * CD_WT + k];
    lwv[ty][k] = k ? v : v - (uint32_t) ry0;
  }
  // conversion: eight pixels of a row per item -> NCH x 8 bytes.  Rows / groups beyond the frame are not converted: their weights are 0.
  // NV12 through the packed ORC pipeline with the NEXT item's seven loads issued before this item's arithmetic (a lane has one or two items:
  // without this the second one's HBM latency was the phase's tail); everything else pixel by pixel (cs_tap)
  const float inv_groups = 1.0f / (float) groups;
  const int n_items = groups * rows;
  auto convert = [&] (auto cosited_tag) {
    constexpr bool COS = decltype (cosited_tag)::value;
    auto where = [&] (int i, int &ry, int &g, int &gx, int &cy) { ry = cd_div (i, inv_groups); g = i - ry * groups; gx = ga + 8 * g; cy = ry0 + ry; };
    auto is_fast = [&] (int gx, int cy) { return NCH == 3 && (ONLY_FAST || p.fast_nv12) && cy < p.cs.in_h && gx + 8 <= p.cs.in_w; };
    int i = tid, ry = 0, g = 0, gx = 0, cy = 0;
    Cv8Raw cur {};
    if (i < n_items) { where (i, ry, g, gx, cy); if (is_fast (gx, cy)) cur = cs_load8_nv12<COS> (p.cs, in, gx, cy); }
    while (i < n_items) {
      const int in_ = i + 512;
      int nry = 0, ng = 0, ngx = 0, ncy = 0;
      Cv8Raw nxt {};
      if (in_ < n_items) { where (in_, nry, ng, ngx, ncy); if (is_fast (ngx, ncy)) nxt = cs_load8_nv12<COS> (p.cs, in, ngx, ncy); }
      if (cy < p.cs.in_h && gx < p.cs.in_w) {
        uint32_t lo[4], hi[4];
        if (ONLY_FAST || is_fast (gx, cy)) cs_convert8_nv12_planar<COS> (p.cs, cur, lo, hi);
        else {
          uint32_t px8[8];
#pragma unroll 1
          for (int k = 0; k < 8; k++) {
            int px[4];
            cs_tap (p.cs, in, min (gx + k, p.cs.in_w - 1), cy, px);
            px8[k] = (uint32_t) px[0] | ((uint32_t) px[1] << 8) | ((uint32_t) px[2] << 16) | ((uint32_t) px[3] << 24);
          }
          cd_transpose4 (px8[0], px8[1], px8[2], px8[3], lo);
          cd_transpose4 (px8[4], px8[5], px8[6], px8[7], hi);
        }
#pragma unroll
        for (int c = 0; c < NCH; c++)
          *reinterpret_cast<uint2 *> (&P1[c][ry * CD_RW + 8 * g]) = make_uint2 (lo[c] ^ 0x80808080u, hi[c] ^ 0x80808080u);
      }
      cur = nxt; i = in_; ry = nry; g = ng; gx = ngx; cy = ncy;
    }
  };
  if (p.cs.cosited) convert (std::true_type ()); else convert (std::false_type ());
  __syncthreads ();
  const uint32_t X = 0x80808080u;
  if (p.vfirst) {
    // pass 1, vertical: (channel, output row, four region columns) -> P2[c][ty][rx]
    const int q = groups * 2;                                         // dwords per region row
    const float inv_q = 1.0f / (float) q;
    for (int i = tid; i < NCH * th * q; i += 512) {
      const int cty = cd_div (i, inv_q), rx4 = i - cty * q;         // (channel, row) pairs: c * th + ty
      const int c = cty >= 2 * th ? (cty >= 3 * th ? 3 : 2) : (cty >= th ? 1 : 0), ty = cty - c * th;
      int acc[4];
      cd_vwin4 (P1[c], CD_RW, 4 * rx4, lwv[ty], p.wv, acc);
      *reinterpret_cast<uint32_t *> (&P2[c][ty * CD_RW + 4 * rx4]) = cd_pack4 (acc[0], acc[1], acc[2], acc[3]) ^ X;
    }
    __syncthreads ();
    // pass 2, horizontal: one output pixel per item
    const float inv_tw = 1.0f / (float) tw;
    for (int i = tid; i < th * tw; i += 512) {
      const int ty = cd_div (i, inv_tw), tx = i - ty * tw;
      int a[4];
#pragma unroll
      for (int c = 0; c < NCH; c++) a[c] = cd_hwin (&P2[c][ty * CD_RW], lwh[tx], p.wh);
      if (NCH == 3) a[3] = (int) lwv[ty][5] * (int) lwh[tx][1] + 32;                       // alpha after the vertical pass x the horizontal tap sum
      *reinterpret_cast<uint32_t *> (out + (size_t) (y0 + ty) * p.os + 4 * (x0 + tx)) = cd_pack4 (a[0], a[1], a[2], a[3]);
    }
  } else {
    // pass 1, horizontal: (channel, region row, four output columns) -> P2[c][ry][tx]
    for (int i = tid; i < NCH * rows * (CD_TW / 4); i += 512) {
      const int cry = i / (CD_TW / 4), tx4 = i - cry * (CD_TW / 4);       // (channel, region row) pairs: c * rows + ry
      const int c = cry >= 2 * rows ? (cry >= 3 * rows ? 3 : 2) : (cry >= rows ? 1 : 0), ry = cry - c * rows;
      const uint8_t *row = &P1[c][ry * CD_RW];
      int a[4];
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] = cd_hwin (row, lwh[4 * tx4 + k], p.wh);
      *reinterpret_cast<uint32_t *> (&P2[c][ry * CD_TW + 4 * tx4]) = cd_pack4 (a[0], a[1], a[2], a[3]) ^ X;
    }
    __syncthreads ();
    // pass 2, vertical: four output pixels of a row per item
    for (int i = tid; i < th * (CD_TW / 4); i += 512) {
      const int ty = i / (CD_TW / 4), tx4 = i - ty * (CD_TW / 4);
      if (4 * tx4 >= tw) continue;
      uint32_t ch[4];
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        int acc[4];
        cd_vwin4 (P2[c], CD_TW, 4 * tx4, lwv[ty], p.wv, acc);
        ch[c] = cd_pack4 (acc[0], acc[1], acc[2], acc[3]);
      }
      if (NCH == 3) {
        const int sv = (int) lwv[ty][1];
        ch[3] = cd_pack4 ((int) lwh[4 * tx4][5] * sv + 32, (int) lwh[4 * tx4 + 1][5] * sv + 32, (int) lwh[4 * tx4 + 2][5] * sv + 32, (int) lwh[4 * tx4 + 3][5] * sv + 32);
      }
      uint32_t px[4];
      cd_transpose4 (ch[0], ch[1], ch[2], ch[3], px);
      uint32_t *o = reinterpret_cast<uint32_t *> (out + (size_t) (y0 + ty) * p.os + 4 * (x0 + 4 * tx4));
#pragma unroll
      for (int k = 0; k < 4; k++) if (4 * tx4 + k < tw) o[k] = px[k];
    }
  }
}

// ---- k_cs_bilinear_tile: conversion + GStreamer's two 2-tap passes fused per output tile (bilinear at any ratio the tile holds) -------
// k_cs_taps / k_cs_generic evaluate an output pixel from four converted taps: 4 conversions (with their chroma filters) per OUTPUT
// pixel whatever the ratio — 16 x the source pixels at 2 x up-scaling, ~220 VALU instructions per output pixel, 42 us for 1080p -> 2160p.
// GStreamer converts at the input size and then scales the 8-bit RGBA lines, vertical pass first iff in_h > out_h + 2, with an 8-bit
// rounding between the passes; this kernel does exactly that per 64 x TH output tile: the source region the tile's taps reach is
// converted ONCE into LDS (aligned 8-pixel groups through the packed ORC pipeline for NV12, cs_tap otherwise), the first pass runs
// LDS -> LDS, the second LDS -> HBM.  Both passes work on whole RGBA dwords as two u16 pairs:
//   vertical  (a * (256 - w) + b * w + 128) >> 8   ==  a + (((b - a) * w + 128) >> 8)   (the sum stays below 2^16),
//   horizontal (a * (256 - f) + b * f) >> 8,  xa = (x * hinc) >> 16, f = ((x * hinc) >> 8) & 255
// — the arithmetic of k_cs_generic, bit for bit.  Used where it pays: no minification on either axis (the region is then smaller than
// the tile: at 2 x up-scaling 4 x fewer conversions than output pixels) with 64 x 32 tiles; for down-scales the per-pixel kernels'
// ~1.6 x more arithmetic costs less than the tile's three barriers (measured: 1080p -> 720p 5.3 vs 5.4 us, I420 / RGB inputs slower).
__device__ __forceinline__ uint32_t bl_vtap (uint32_t a, uint32_t b, uint32_t ww, uint32_t wm)
{
  const u16x2 al = as_u16x2 (a & 0x00ff00ffu), ah = as_u16x2 ((a >> 8) & 0x00ff00ffu), bl = as_u16x2 (b & 0x00ff00ffu), bh = as_u16x2 ((b >> 8) & 0x00ff00ffu);
  const u16x2 tl = bl * as_u16x2 (ww) + (al * as_u16x2 (wm) + as_u16x2 (0x00800080u));
  const u16x2 th = bh * as_u16x2 (ww) + (ah * as_u16x2 (wm) + as_u16x2 (0x00800080u));
  return ((as_u32 (tl) >> 8) & 0x00ff00ffu) | (as_u32 (th) & 0xff00ff00u);
}
__device__ __forceinline__ uint32_t bl_htap (uint32_t a, uint32_t b, uint32_t fw, uint32_t fm)
{
  const u16x2 al = as_u16x2 (a & 0x00ff00ffu), ah = as_u16x2 ((a >> 8) & 0x00ff00ffu), bl = as_u16x2 (b & 0x00ff00ffu), bh = as_u16x2 ((b >> 8) & 0x00ff00ffu);
  const u16x2 tl = bl * as_u16x2 (fw) + al * as_u16x2 (fm);
  const u16x2 th = bh * as_u16x2 (fw) + ah * as_u16x2 (fm);
  return ((as_u32 (tl) >> 8) & 0x00ff00ffu) | (as_u32 (th) & 0xff00ff00u);
}

// ---- k_cs_rgb_taps_strip<VFIRST, ROWS>: BGRA / RGBA -> BGRA / RGBA, 2-tap bilinear with minification on an axis (a capture or a render scaled down) ----
// k_cs_generic gathered four dwords per output pixel and ran the two passes channel by channel with its format switches: 6.2 us for 1080p -> 720p.
// Like k_cs_taps_strip: the two taps of a row are adjacent pixels — ONE 8-byte window per source row (half the gathers) —, a lane walks ROWS
// output rows of its column with the column-only work done once, the rows' taps as scalar loads up front and the next row's windows in flight;
// both passes on whole pixels as two u16 pairs (bl_vtap / bl_htap: the tile kernel's arithmetic, k_cs_generic's bit for bit), the R <-> B swap
// of a format change applied to the finished pixel (the passes treat the channels alike).
// Contract (host): bilinear, no borders, in_w >= 2, adjacent vertical taps, 4-byte aligned rows.
template <bool VFIRST, int ROWS>
__global__ __launch_bounds__ (256) void k_cs_rgb_taps_strip (const CsParams p, int swap)
{
  typedef uint2 __attribute__ ((aligned (4))) uint2_a4;
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int strip = __builtin_amdgcn_readfirstlane ((int) (blockIdx.y * 4 + threadIdx.y));
  const int y0 = strip * ROWS, yend = min (y0 + ROWS, p.out_h);
  if (x >= p.out_w || y0 >= p.out_h) return;
  const uint8_t *ip = p.in[0] + (size_t) blockIdx.z * p.in_pitch;
  uint8_t *op = p.out + (size_t) blockIdx.z * p.out_pitch + 4 * (size_t) x;
  const int4 *vt = reinterpret_cast<const int4 *> (p.vtab);
  int xa = x, xb = x, f = 0;
  if (p.hscale_on) {
    const uint32_t tt = (uint32_t) x * p.hinc;
    xa = min ((int) (tt >> 16), p.in_w - 1); f = (int) ((tt >> 8) & 0xff); xb = min (xa + 1, p.in_w - 1);
  }
  const int base = min (xa, p.in_w - 2);                            // the 8-byte window [base, base + 1] holds both taps
  const bool a_hi = xa != base, b_hi = xb != base;
  const uint32_t fw = (uint32_t) f | ((uint32_t) f << 16), fm = 0x01000100u - fw;
  auto load = [&] (int i0, int i1, uint2 &a, uint2 &b) {
    a = *reinterpret_cast<const uint2_a4 *> (ip + ((uint32_t) i0 * (uint32_t) p.is[0] + 4u * (uint32_t) base));
    b = *reinterpret_cast<const uint2_a4 *> (ip + ((uint32_t) i1 * (uint32_t) p.is[0] + 4u * (uint32_t) base));
  };
  auto compute = [&] (uint2 r0, uint2 r1, int w, int y) {
    const uint32_t a0 = a_hi ? r0.y : r0.x, b0 = b_hi ? r0.y : r0.x, a1 = a_hi ? r1.y : r1.x, b1 = b_hi ? r1.y : r1.x;
    const uint32_t ww = (uint32_t) w | ((uint32_t) w << 16), wm = 0x01000100u - ww;
    uint32_t q;
    if (VFIRST) q = bl_htap (bl_vtap (a0, a1, ww, wm), bl_vtap (b0, b1, ww, wm), fw, fm);
    else q = bl_vtap (bl_htap (a0, b0, fw, fm), bl_htap (a1, b1, fw, fm), ww, wm);
    if (swap) q = perm_b32 (0u, q, 0x03000102u);
    __builtin_nontemporal_store (q, reinterpret_cast<uint32_t *> (op + (uint32_t) y * (uint32_t) p.os));
  };
  int4 tp[ROWS + 1];                                                  // scalar loads, all before the first store (see k_cs_taps_strip)
#pragma unroll
  for (int r = 0; r <= ROWS; r++) tp[r] = vt[min (y0 + r, p.out_h - 1)];
  uint2 A[2], B[2];
  load (tp[0].x, tp[0].y, A[0], B[0]);
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    if (y0 + r >= yend) break;
    load (tp[r + 1].x, tp[r + 1].y, A[(r + 1) & 1], B[(r + 1) & 1]);
    __builtin_amdgcn_sched_barrier (0);
    compute (A[r & 1], B[r & 1], tp[r].z, y0 + r);
  }
}

template <int THREADS, int TH>
__global__ __launch_bounds__ (THREADS) void k_cs_bilinear_tile (const CsParams p, int fast)
{
  __shared__ __attribute__ ((aligned (16))) uint32_t reg[CT_RH][CT_RW];   // converted source region
  __shared__ uint32_t tmp[CT_RH * CT_TW];                             // first-pass result: [TH][rw] (V first) or [rh][CT_TW] (H first)
  __shared__ int lxa[CT_TW], lxb[CT_TW];                              // this tile's horizontal taps: region-relative columns ...
  __shared__ uint32_t lfw[CT_TW];                                     // ... and the weight f
  __shared__ int lv0[TH], lv1[TH];                              // vertical taps: region-relative rows ...
  __shared__ uint32_t lvw[TH];                                     // ... and the weight w
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * CT_TW, y0 = blockIdx.y * TH;
  const int ow = p.out_w, oh = p.out_h;
  const int tw = min (CT_TW, ow - x0), th = min (TH, oh - y0);
  const uint8_t *in[3] = { p.in[0] + (size_t) blockIdx.z * p.in_pitch,
                           p.in[1] ? p.in[1] + (size_t) blockIdx.z * p.in_pitch : nullptr,
                           p.in[2] ? p.in[2] + (size_t) blockIdx.z * p.in_pitch : nullptr };
  uint8_t *out = p.out + (size_t) blockIdx.z * p.out_pitch;
  auto hx = [&] (int x, int &xa, int &xb, int &f) {
    xa = x; xb = x; f = 0;
    if (p.hscale_on) {
      const uint32_t t = (uint32_t) x * p.hinc;
      xa = min ((int) (t >> 16), p.in_w - 1); f = (int) ((t >> 8) & 0xff); xb = min (xa + 1, p.in_w - 1);
    }
  };
  // source region of this tile (both tap sequences are non-decreasing)
  int cx0, cx1, t0, t1;
  hx (x0, cx0, t0, t1); hx (x0 + tw - 1, t0, cx1, t1);
  const int ry0 = p.vtab[4 * y0], ry1 = p.vtab[4 * (y0 + th - 1) + 1];
  const int rh = ry1 - ry0 + 1;
  int rw = cx1 - cx0 + 1;
  const int ga0 = fast ? (cx0 & ~7) : cx0;                            // the region's first column (see below)
  for (int i = tid; i < tw; i += THREADS) { int xa, xb, f; hx (x0 + i, xa, xb, f); lxa[i] = xa - ga0; lxb[i] = xb - ga0; lfw[i] = (uint32_t) f; }
  for (int i = tid; i < th; i += THREADS) { lv0[i] = p.vtab[4 * (y0 + i)] - ry0; lv1[i] = p.vtab[4 * (y0 + i) + 1] - ry0; lvw[i] = (uint32_t) p.vtab[4 * (y0 + i) + 2]; }
  if (fast) {
    // `fast`: the input meets the alignment contract of its 8-pixel converter (1 NV12, 2 I420, 3 UYVY, 4 YUY2).  The region is widened to
    // whole 8-column groups; groups that would cross the right image edge fall back to cs_tap
    const int groups = ((cx1 + 1 - ga0) + 7) >> 3;
    for (int i = tid; i < groups * rh; i += THREADS) {
      const int ry = i / groups, g = i - ry * groups, gx = ga0 + 8 * g;
      uint32_t px8[8];
      if (gx + 8 <= p.in_w) {
        if (fast == 2) cs_convert8_i420 (p, in, gx, ry0 + ry, px8);
        else if (fast == 1) { if (p.cosited) cs_convert8_nv12<true> (p, in, gx, ry0 + ry, px8); else cs_convert8_nv12<false> (p, in, gx, ry0 + ry, px8); }
        else if (fast == 3) { if (p.cosited) cs_convert8_packed<false, true> (p, in, gx, ry0 + ry, px8); else cs_convert8_packed<false, false> (p, in, gx, ry0 + ry, px8); }
        else { if (p.cosited) cs_convert8_packed<true, true> (p, in, gx, ry0 + ry, px8); else cs_convert8_packed<true, false> (p, in, gx, ry0 + ry, px8); }
      } else {
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
          int px[4];
          cs_tap (p, in, min (gx + k, p.in_w - 1), ry0 + ry, px);
          px8[k] = (uint32_t) px[0] | ((uint32_t) px[1] << 8) | ((uint32_t) px[2] << 16) | ((uint32_t) px[3] << 24);
        }
      }
      uint4 *d = reinterpret_cast<uint4 *> (&reg[ry][8 * g]);
      d[0] = make_uint4 (px8[0], px8[1], px8[2], px8[3]); d[1] = make_uint4 (px8[4], px8[5], px8[6], px8[7]);
    }
    rw = 8 * groups;
  } else {
    for (int i = tid; i < rw * rh; i += THREADS) {
      const int ry = i / rw, rx = i - ry * rw;
      int px[4];
      cs_tap (p, in, cx0 + rx, ry0 + ry, px);
      reg[ry][rx] = (uint32_t) px[0] | ((uint32_t) px[1] << 8) | ((uint32_t) px[2] << 16) | ((uint32_t) px[3] << 24);
    }
  }
  __syncthreads ();
  if (p.vfirst || !p.hscale_on) {
    // vertical first: tmp[ty][rx] over the region's columns, then the horizontal taps along each row
    for (int i = tid; i < th * rw; i += THREADS) {
      const int ty = i / rw, rx = i - ty * rw;
      const uint32_t w = lvw[ty], ww = w | (w << 16);
      tmp[ty * rw + rx] = bl_vtap (reg[lv0[ty]][rx], reg[lv1[ty]][rx], ww, 0x01000100u - ww);
    }
    __syncthreads ();
    // four adjacent pixels per lane, one 16-byte store (dword stores when the frame is not 16-byte aligned)
    const bool vec = !(((uintptr_t) out | (uintptr_t) p.os) & 15);
    for (int i = tid; i < th * (CT_TW / 4); i += THREADS) {
      const int ty = i / (CT_TW / 4), tx = 4 * (i - ty * (CT_TW / 4));
      if (tx >= tw) continue;
      uint32_t v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int txk = min (tx + k, tw - 1);
        uint32_t q = tmp[ty * rw + lxa[txk]];
        if (p.hscale_on) { const uint32_t f = lfw[txk], fw = f | (f << 16); q = bl_htap (q, tmp[ty * rw + lxb[txk]], fw, 0x01000100u - fw); }
        v[k] = q;
      }
      uint32_t *d = reinterpret_cast<uint32_t *> (out + (size_t) (y0 + ty) * p.os + 4 * (x0 + tx));
      if (vec && tx + 3 < tw) {
        typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
        const v4u q = { v[0], v[1], v[2], v[3] };
        __builtin_nontemporal_store (q, reinterpret_cast<v4u *> (d));
      } else {
        d[0] = v[0];
        if (tx + 1 < tw) d[1] = v[1];
        if (tx + 2 < tw) d[2] = v[2];
        if (tx + 3 < tw) d[3] = v[3];
      }
    }
  } else {
    // horizontal first: tmp[ry][tx] over the region's rows, then the vertical taps down each column
    for (int i = tid; i < rh * tw; i += THREADS) {
      const int ry = i / tw, tx = i - ry * tw;
      const uint32_t f = lfw[tx], fw = f | (f << 16);
      tmp[ry * CT_TW + tx] = bl_htap (reg[ry][lxa[tx]], reg[ry][lxb[tx]], fw, 0x01000100u - fw);
    }
    __syncthreads ();
    // four adjacent pixels per lane: two 16-byte LDS reads, one 16-byte store (dword stores when the frame is not 16-byte aligned)
    const bool vec = !(((uintptr_t) out | (uintptr_t) p.os) & 15);
    for (int i = tid; i < th * (CT_TW / 4); i += THREADS) {
      const int ty = i / (CT_TW / 4), tx = 4 * (i - ty * (CT_TW / 4));
      if (tx >= tw) continue;
      const uint32_t w = lvw[ty], ww = w | (w << 16), wm = 0x01000100u - ww;
      const uint4 a = *reinterpret_cast<const uint4 *> (&tmp[lv0[ty] * CT_TW + tx]), b = *reinterpret_cast<const uint4 *> (&tmp[lv1[ty] * CT_TW + tx]);
      const uint4 v = make_uint4 (bl_vtap (a.x, b.x, ww, wm), bl_vtap (a.y, b.y, ww, wm), bl_vtap (a.z, b.z, ww, wm), bl_vtap (a.w, b.w, ww, wm));
      uint32_t *d = reinterpret_cast<uint32_t *> (out + (size_t) (y0 + ty) * p.os + 4 * (x0 + tx));
      if (vec && tx + 3 < tw) {
        typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
        const v4u q = { v.x, v.y, v.z, v.w };
        __builtin_nontemporal_store (q, reinterpret_cast<v4u *> (d));
      } else {
        d[0] = v.x;
        if (tx + 1 < tw) d[1] = v.y;
        if (tx + 2 < tw) d[2] = v.z;
        if (tx + 3 < tw) d[3] = v.w;
      }
    }
  }
}


}  // namespace vfhip
