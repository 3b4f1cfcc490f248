// csrc/convertscale_planar_kernels.h — gst-exact cells of vfhipconvertscale whose OUTPUT is 4:2:0 (NV12 / I420) or packed
// 4:2:2 (UYVY / YUY2).
//
// GStreamer's `videoconvert ! videoscale` does these in two steps and so do we (bit-exact, oracle/gst114.c
// gst114_rgb_to_yuv420 / gst114_*_packed422 / gst114_scale_plane*; rules pinned by probing the real 1.14 elements):
//   stage 1  videoconvert at the INPUT size: BGRA/RGBA -> NV12/I420/UYVY/YUY2 (8-bit integer matrix, chroma averaged
//            vertically then horizontally), NV12 <-> I420 re-packing, 4:2:0 <-> packed 4:2:2 (GStreamer's fast paths
//            for I420, the generic up / down-sampling path for NV12), UYVY <-> YUY2 swizzle;
//   stage 2  videoscale plane by plane: luma / I420 chroma as 1 x u8 (edge-aligned 16.16 horizontal taps, or pair
//            averaging when exactly halved), NV12 chroma as 2 x u8 (centre-aligned, 6-bit taps), vertical 8-bit
//            centre-aligned taps, pass order per plane; a packed frame as three interleaved lines; method=nearest
//            through the same tables with a zero second tap; method=bicubic through n-tap tables (catrom on luma and
//            packed lines, un-limited LINEAR taps on planar chroma).
// Replaces reference rgbaToNV12 / rgbaToI420 (common/vfmetalshaders.m:90-168) and rgbaToUYVY / rgbaToYUY2
// (convertscale/metalconvertscale_shaders.h:202-269) for numerics=gst-exact.
// Every kernel takes a batch (blockIdx.z = frame).  Access width is what these byte kernels live on: dword / 16-bit loads
// and dword stores wherever rows are aligned, byte-wise twins otherwise (DESIGN.md section 5.2 has the measurements).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vfhip {

struct Rgb2YuvParams {
  size_t in_pitch, out_pitch;
  const uint8_t *in; int is;
  uint8_t *y, *u, *v; int ys, us, vs;     // NV12: u = uv plane, v unused
  int w, h, in_rgba, planar, cosited;
  int c[9];
};

__global__ __launch_bounds__ (256) void k_rgb_to_yuv420 (const Rgb2YuvParams p0)
{
  Rgb2YuvParams p = p0;
  p.in += (size_t) blockIdx.z * p.in_pitch;
  p.y += (size_t) blockIdx.z * p.out_pitch; p.u += (size_t) blockIdx.z * p.out_pitch; if (p.v) p.v += (size_t) blockIdx.z * p.out_pitch;
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;      // chroma sample (k, j)
  const int cw = (p.w + 1) >> 1, chh = (p.h + 1) >> 1;
  if (k >= cw || j >= chh) return;
  const int ro = p.in_rgba ? 0 : 2, bo = 2 - ro;
  const int x = 2 * k;
  // vertical averages (a+b+1)>>1 of the converted chroma at columns x-1, x, x+1 (edge rules: see oracle/gst114.c)
  int xs[3];
  xs[1] = x;
  if (p.cosited) { xs[0] = x > 0 ? x - 1 : 0; xs[2] = (k == cw - 1 && k > 0) ? x : min (x + 1, p.w - 1); }   // the last sample ignores its right neighbour, unless it is also the first
  else { xs[0] = x; xs[2] = min (x + 1, p.w - 1); }
  int su[3], sv[3];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    int au = 1, av = 1;
#pragma unroll
    for (int d = 0; d < 2; d++) {
      const int yy = min (2 * j + d, p.h - 1);
      const uint8_t *px = p.in + (size_t) yy * p.is + 4 * xs[t];
      const int r = px[ro], g = px[1], b = px[bo];
      au += ((p.c[3] * r + p.c[4] * g + p.c[5] * b) >> 8) + 128;
      av += ((p.c[6] * r + p.c[7] * g + p.c[8] * b) >> 8) + 128;
    }
    su[t] = au >> 1; sv[t] = av >> 1;
  }
  int U, V;
  if (p.cosited) { U = (su[0] + 2 * su[1] + su[2] + 2) >> 2; V = (sv[0] + 2 * sv[1] + sv[2] + 2) >> 2; }
  else { U = (su[1] + su[2] + 1) >> 1; V = (sv[1] + sv[2] + 1) >> 1; }
  if (p.planar) { p.u[(size_t) j * p.us + k] = (uint8_t) U; p.v[(size_t) j * p.vs + k] = (uint8_t) V; }
  else { p.u[(size_t) j * p.us + 2 * k] = (uint8_t) U; p.u[(size_t) j * p.us + 2 * k + 1] = (uint8_t) V; }
  // luma of the 2x2 block
#pragma unroll
  for (int d = 0; d < 2; d++)
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int xx = x + e, yy = 2 * j + d;
      if (xx < p.w && yy < p.h) {
        const uint8_t *px = p.in + (size_t) yy * p.is + 4 * xx;
        p.y[(size_t) yy * p.ys + xx] = (uint8_t) (((p.c[0] * px[ro] + p.c[1] * px[1] + p.c[2] * px[bo]) >> 8) + 16);
      }
    }
}

// Fast variant for 4-byte-aligned RGB rows: every pixel is ONE dword load and every matrix row one or two `v_dot4_u32_u8`
// (coefficients packed per byte lane on the host, positive and negative parts apart: the dot product is the same exact
// integer as the three multiply-adds of k_rgb_to_yuv420, so the bytes are identical).  Same lane = chroma sample mapping.
struct Rgb2YuvFastParams {
  Rgb2YuvParams b;
  uint32_t cy, cup, cun, cvp, cvn;     // packed coefficients in the input's byte order (alpha lane 0)
};

__device__ __forceinline__ void rgb_chroma (const Rgb2YuvFastParams &q, uint32_t px, int &u, int &v)
{
  u = (((int) __builtin_amdgcn_udot4 (px, q.cup, 0u, false) - (int) __builtin_amdgcn_udot4 (px, q.cun, 0u, false)) >> 8) + 128;
  v = (((int) __builtin_amdgcn_udot4 (px, q.cvp, 0u, false) - (int) __builtin_amdgcn_udot4 (px, q.cvn, 0u, false)) >> 8) + 128;
}

__global__ __launch_bounds__ (256) void k_rgb_to_yuv420_fast (const Rgb2YuvFastParams q)
{
  Rgb2YuvParams p = q.b;
  p.in += (size_t) blockIdx.z * p.in_pitch;
  p.y += (size_t) blockIdx.z * p.out_pitch; p.u += (size_t) blockIdx.z * p.out_pitch; if (p.v) p.v += (size_t) blockIdx.z * p.out_pitch;
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  const int cw = (p.w + 1) >> 1, chh = (p.h + 1) >> 1;
  if (k >= cw || j >= chh) return;
  const int x = 2 * k, xr = min (x + 1, p.w - 1), xl = max (x - 1, 0);
  const int y0 = 2 * j, y1 = min (2 * j + 1, p.h - 1);
  const uint32_t *r0 = reinterpret_cast<const uint32_t *> (p.in + (size_t) y0 * p.is), *r1 = reinterpret_cast<const uint32_t *> (p.in + (size_t) y1 * p.is);
  const uint32_t c0 = r0[x], c1 = r1[x], e0 = r0[xr], e1 = r1[xr];
  int u0, v0, u1, v1;
  // vertical pair averages of the converted chroma at the centre and right columns (and left, co-sited)
  rgb_chroma (q, c0, u0, v0); rgb_chroma (q, c1, u1, v1);
  const int cu = (u0 + u1 + 1) >> 1, cv = (v0 + v1 + 1) >> 1;
  rgb_chroma (q, e0, u0, v0); rgb_chroma (q, e1, u1, v1);
  int ru = (u0 + u1 + 1) >> 1, rv = (v0 + v1 + 1) >> 1;
  int U, V;
  if (p.cosited) {
    if (k == cw - 1 && k > 0) { ru = cu; rv = cv; }        // the last sample ignores its right neighbour, unless it is also the first
    int lu = cu, lv = cv;
    if (x > 0) {
      rgb_chroma (q, r0[xl], u0, v0); rgb_chroma (q, r1[xl], u1, v1);
      lu = (u0 + u1 + 1) >> 1; lv = (v0 + v1 + 1) >> 1;
    }
    U = (lu + 2 * cu + ru + 2) >> 2; V = (lv + 2 * cv + rv + 2) >> 2;
  } else { U = (cu + ru + 1) >> 1; V = (cv + rv + 1) >> 1; }
  if (p.planar) { p.u[(size_t) j * p.us + k] = (uint8_t) U; p.v[(size_t) j * p.vs + k] = (uint8_t) V; }
  else {
    uint8_t *d = p.u + (size_t) j * p.us + 2 * k;
    if (((uintptr_t) d & 1) == 0) *reinterpret_cast<uint16_t *> (d) = (uint16_t) (U | (V << 8));
    else { d[0] = (uint8_t) U; d[1] = (uint8_t) V; }
  }
  // luma of the 2x2 block: the pair of a row goes out as one 2-byte store when it can
  const uint32_t ya = (__builtin_amdgcn_udot4 (c0, q.cy, 0u, false) >> 8) + 16, yb = (__builtin_amdgcn_udot4 (e0, q.cy, 0u, false) >> 8) + 16;
  const uint32_t yc = (__builtin_amdgcn_udot4 (c1, q.cy, 0u, false) >> 8) + 16, yd = (__builtin_amdgcn_udot4 (e1, q.cy, 0u, false) >> 8) + 16;
  uint8_t *d0 = p.y + (size_t) y0 * p.ys + x, *d1 = p.y + (size_t) (2 * j + 1) * p.ys + x;
  const bool two = x + 1 < p.w;
  if (two && ((uintptr_t) d0 & 1) == 0) *reinterpret_cast<uint16_t *> (d0) = (uint16_t) (ya | (yb << 8));
  else { d0[0] = (uint8_t) ya; if (two) d0[1] = (uint8_t) yb; }
  if (2 * j + 1 < p.h) {
    if (two && ((uintptr_t) d1 & 1) == 0) *reinterpret_cast<uint16_t *> (d1) = (uint16_t) (yc | (yd << 8));
    else { d1[0] = (uint8_t) yc; if (two) d1[1] = (uint8_t) yd; }
  }
}

struct RepackParams {
  size_t in_pitch, out_pitch;
  const uint8_t *iy, *iu, *iv; int iys, ius, ivs;
  uint8_t *oy, *ou, *ov; int oys, ous, ovs;
  int w, h, in_planar, out_planar;
};

__global__ __launch_bounds__ (256) void k_repack_420 (const RepackParams p0)
{
  RepackParams p = p0;
  p.iy += (size_t) blockIdx.z * p.in_pitch; p.iu += (size_t) blockIdx.z * p.in_pitch; if (p.iv) p.iv += (size_t) blockIdx.z * p.in_pitch;
  p.oy += (size_t) blockIdx.z * p.out_pitch; p.ou += (size_t) blockIdx.z * p.out_pitch; if (p.ov) p.ov += (size_t) blockIdx.z * p.out_pitch;
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  const int cw = (p.w + 1) >> 1, chh = (p.h + 1) >> 1;
  if (k >= cw || j >= chh) return;
  uint8_t U, V;
  if (p.in_planar) { U = p.iu[(size_t) j * p.ius + k]; V = p.iv[(size_t) j * p.ivs + k]; }
  else { U = p.iu[(size_t) j * p.ius + 2 * k]; V = p.iu[(size_t) j * p.ius + 2 * k + 1]; }
  if (p.out_planar) { p.ou[(size_t) j * p.ous + k] = U; p.ov[(size_t) j * p.ovs + k] = V; }
  else { p.ou[(size_t) j * p.ous + 2 * k] = U; p.ou[(size_t) j * p.ous + 2 * k + 1] = V; }
#pragma unroll
  for (int d = 0; d < 2; d++)
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int xx = 2 * k + e, yy = 2 * j + d;
      if (xx < p.w && yy < p.h) p.oy[(size_t) yy * p.oys + xx] = p.iy[(size_t) yy * p.iys + xx];
    }
}

// the same with 16-byte accesses: a lane moves 16 luma columns of two rows and the eight chroma samples under them (NV12 side: one 16-byte
// load / store of interleaved pairs; I420 side: 8 bytes per plane), the (de)interleave is two v_perm per dword.  NV12 <-> I420 at the same size is
// what the element does between a decoder and an encoder that disagree on the layout: 3.8 -> see DESIGN §5.2 per 1080p frame.
// Contract (host): w % 16 == 0, even h, 16-byte aligned luma and NV12 chroma rows, 8-byte aligned I420 chroma rows.
__global__ __launch_bounds__ (256) void k_repack_420_vec (const RepackParams p0)
{
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  typedef uint32_t v2u __attribute__ ((ext_vector_type (2)));
  RepackParams p = p0;
  p.iy += (size_t) blockIdx.z * p.in_pitch; p.iu += (size_t) blockIdx.z * p.in_pitch; if (p.iv) p.iv += (size_t) blockIdx.z * p.in_pitch;
  p.oy += (size_t) blockIdx.z * p.out_pitch; p.ou += (size_t) blockIdx.z * p.out_pitch; if (p.ov) p.ov += (size_t) blockIdx.z * p.out_pitch;
  const int g = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (16 * g >= p.w || 2 * j >= p.h) return;
  const v4u y0 = *(reinterpret_cast<const v4u *> (p.iy + (size_t) (2 * j) * p.iys) + g), y1 = *(reinterpret_cast<const v4u *> (p.iy + (size_t) (2 * j + 1) * p.iys) + g);
  v2u u, v;                                      // eight U and eight V samples
  if (p.in_planar) {
    u = *(reinterpret_cast<const v2u *> (p.iu + (size_t) j * p.ius) + g); v = *(reinterpret_cast<const v2u *> (p.iv + (size_t) j * p.ivs) + g);
  } else {
    const v4u c = *(reinterpret_cast<const v4u *> (p.iu + (size_t) j * p.ius) + g);
    u.x = __builtin_amdgcn_perm (c.y, c.x, 0x06040200u); u.y = __builtin_amdgcn_perm (c.w, c.z, 0x06040200u);
    v.x = __builtin_amdgcn_perm (c.y, c.x, 0x07050301u); v.y = __builtin_amdgcn_perm (c.w, c.z, 0x07050301u);
  }
  __builtin_nontemporal_store (y0, reinterpret_cast<v4u *> (p.oy + (size_t) (2 * j) * p.oys) + g);
  __builtin_nontemporal_store (y1, reinterpret_cast<v4u *> (p.oy + (size_t) (2 * j + 1) * p.oys) + g);
  if (p.out_planar) {
    __builtin_nontemporal_store (u, reinterpret_cast<v2u *> (p.ou + (size_t) j * p.ous) + g);
    __builtin_nontemporal_store (v, reinterpret_cast<v2u *> (p.ov + (size_t) j * p.ovs) + g);
  } else {
    const v4u c = { __builtin_amdgcn_perm (v.x, u.x, 0x05010400u), __builtin_amdgcn_perm (v.x, u.x, 0x07030602u),
                    __builtin_amdgcn_perm (v.y, u.y, 0x05010400u), __builtin_amdgcn_perm (v.y, u.y, 0x07030602u) };
    __builtin_nontemporal_store (c, reinterpret_cast<v4u *> (p.ou + (size_t) j * p.ous) + g);
  }
}

// ---- stage 1 for YUV -> YUV with a MATRIX change (NV12 / I420 / UYVY / YUY2 either side) or NV12 <-> I420 with a SITING change ----
// videoconvert's generic path (oracle/gst114.c gst114_yuv_to_yuv; 50 real-pipeline vectors, tests/golden/convertscale_gst114_remat.npz):
//   same siting and the same subsampling on both sides: every luma sample is matrixed with its nearest chroma sample, every output
//     chroma sample is the matrixed input chroma sample;
//   otherwise: chroma up-sampled to 4:4:4 with the input siting (horizontal, then — 4:2:0 only — vertical 3:1 over an even number
//     of lines), the 8-bit matrix out = clamp8 (((a Y + b U + c V) >> 8) + d) per sample, chroma down-sampled with the output
//     siting (vertical pair average for 4:2:0, then horizontal).
// One lane = one OUTPUT chroma sample and the luma samples under it.  A rare cell: written for exactness, not for bandwidth.
struct YuvRematParams {
  size_t in_pitch, out_pitch;
  const uint8_t *iy, *iu, *iv; int iys, iystep, ics, icstep, in420;      // luma x of row y: iy[y * iys + x * iystep]; chroma k of row j: iu[j * ics + k * icstep]
  uint8_t *oy, *ou, *ov; int oys, oystep, ocs, ocstep, out420;
  int w, h, cos_in, cos_out, remat, same;
  int t[12];                                                             // rows Y, U, V x (a, b, c, d)
};

__device__ __forceinline__ int remat_row (const int *r, int y, int u, int v) { return min (max (((r[0] * y + r[1] * u + r[2] * v) >> 8) + r[3], 0), 255); }

__global__ __launch_bounds__ (256) void k_yuv_to_yuv (const YuvRematParams p0)
{
  YuvRematParams p = p0;
  p.iy += (size_t) blockIdx.z * p.in_pitch; p.iu += (size_t) blockIdx.z * p.in_pitch; p.iv += (size_t) blockIdx.z * p.in_pitch;
  p.oy += (size_t) blockIdx.z * p.out_pitch; p.ou += (size_t) blockIdx.z * p.out_pitch; p.ov += (size_t) blockIdx.z * p.out_pitch;
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  const int w = p.w, h = p.h, cw = (w + 1) >> 1;
  const int ich = p.in420 ? (h + 1) >> 1 : h, och = p.out420 ? (h + 1) >> 1 : h;
  if (k >= cw || j >= och) return;
  const int rows = p.out420 ? 2 : 1, y0 = p.out420 ? 2 * j : j;
  if (p.same) {
    const int U = p.iu[(size_t) j * p.ics + k * p.icstep], V = p.iv[(size_t) j * p.ics + k * p.icstep];
    for (int d = 0; d < rows; d++)
      for (int e = 0; e < 2; e++) {
        const int x = 2 * k + e, y = y0 + d;
        if (x < w && y < h) { const int Y = p.iy[(size_t) y * p.iys + x * p.iystep]; p.oy[(size_t) y * p.oys + x * p.oystep] = (uint8_t) (p.remat ? remat_row (p.t, Y, U, V) : Y); }
      }
    p.ou[(size_t) j * p.ocs + k * p.ocstep] = (uint8_t) (p.remat ? remat_row (p.t + 4, 0, U, V) : U);
    p.ov[(size_t) j * p.ocs + k * p.ocstep] = (uint8_t) (p.remat ? remat_row (p.t + 8, 0, U, V) : V);
    return;
  }
  // horizontally up-sampled input chroma of chroma row jj at full-resolution column x (GStreamer's rule for the input siting)
  auto hup = [&] (const uint8_t *c, int jj, int x) {
    const int kk = x >> 1;
    const uint8_t *row = c + (size_t) jj * p.ics;
    const int c0 = row[kk * p.icstep], cm = row[max (kk - 1, 0) * p.icstep], cp = row[min (kk + 1, cw - 1) * p.icstep];
    if (p.cos_in) return (x & 1) ? (c0 + cp + 1) >> 1 : c0;
    return (x & 1) ? (3 * c0 + cp + 2) >> 2 : (3 * c0 + cm + 2) >> 2;
  };
  // 4:4:4 chroma at (x, y); y may be the phantom line h of an odd 4:2:0 frame
  auto up = [&] (const uint8_t *c, int x, int y) {
    if (!p.in420) return hup (c, min (y, h - 1), x);
    const int jj = y >> 1, jn = (y & 1) ? min (jj + 1, ich - 1) : max (jj - 1, 0);
    return (3 * hup (c, jj, x) + hup (c, jn, x) + 2) >> 2;
  };
  // columns the output chroma sample taps: co-sited (l, 2k, r) with weights 1 2 1, else (2k, 2k + 1) averaged
  const int xc = 2 * k, xl = max (xc - 1, 0), xr = p.cos_out ? ((k == cw - 1 && k > 0) ? xc : min (xc + 1, w - 1)) : min (xc + 1, w - 1);
  int fu[3], fv[3];                                     // vertically reduced matrixed chroma at columns l, c, r
  for (int q = 0; q < 3; q++) {
    if (q == 0 && !p.cos_out) { fu[0] = fv[0] = 0; continue; }
    const int x = q == 0 ? xl : (q == 1 ? xc : xr);
    int su = 0, sv = 0;
    for (int d = 0; d < rows; d++) {
      const int U = up (p.iu, x, y0 + d), V = up (p.iv, x, y0 + d);
      su += p.remat ? remat_row (p.t + 4, 0, U, V) : U; sv += p.remat ? remat_row (p.t + 8, 0, U, V) : V;
    }
    fu[q] = p.out420 ? (su + 1) >> 1 : su; fv[q] = p.out420 ? (sv + 1) >> 1 : sv;
  }
  const int U = p.cos_out ? (fu[0] + 2 * fu[1] + fu[2] + 2) >> 2 : (fu[1] + fu[2] + 1) >> 1;
  const int V = p.cos_out ? (fv[0] + 2 * fv[1] + fv[2] + 2) >> 2 : (fv[1] + fv[2] + 1) >> 1;
  p.ou[(size_t) j * p.ocs + k * p.ocstep] = (uint8_t) U;
  p.ov[(size_t) j * p.ocs + k * p.ocstep] = (uint8_t) V;
  for (int d = 0; d < rows; d++)
    for (int e = 0; e < 2; e++) {
      const int x = 2 * k + e, y = y0 + d;
      if (x >= w || y >= h) continue;
      const int Y = p.iy[(size_t) y * p.iys + x * p.iystep];
      p.oy[(size_t) y * p.oys + x * p.oystep] = (uint8_t) (p.remat ? remat_row (p.t, Y, up (p.iu, x, y), up (p.iv, x, y)) : Y);
    }
}

// one plane of n interleaved u8 components
struct PlaneScaleParams {
  const uint8_t *in; int is;
  uint8_t *out; int os;
  int w, h, ow, oh, n;
  int istep, ostep;        // bytes between consecutive samples of this plane (n for a plane of its own; 2 / 4 for the
                           // luma / chroma lines interleaved in a packed 4:2:2 frame); hmode 1 / 2 need istep == 1
  size_t in_pitch, out_pitch;   // batch: frame k of the launch at base + k * pitch (blockIdx.z)
  int vec;                 // source rows are 4-byte aligned: hmode 0 / 2 read dwords
  int hmode;               // 0: no horizontal scaling, 1: edge-aligned 16.16 (1 x u8), 2: pair average (1 x u8), 3: 2-tap table (6-bit), 4: n-tap table, 5: like 3 with the exact-half table (2k, 2k+1, 32), 6: nearest (htab)
  int vmode;               // 0: no vertical scaling, 1: 2-tap 8-bit (vtab), 2: n-tap 6-bit (vnt), 3: nearest (vtab)
  int vfirst;
  uint32_t hinc;
  const int *vtab;         // vmode 1: oh * {i0, i1, w, 0}
  const int *htab;         // hmode 3: ow * {i0, i1, t, 0}
  const int2 *hnt, *vnt;   // hmode 4 / vmode 2: out * n * {source index, 6-bit tap}
  int nh, nv;
};

// One pass each, written over an accessor so that they compose in either order (GstVideoScaler runs the vertical pass
// first iff in_h > out_h + n_taps_v, and every pass rounds to 8 bits):
//   hpass (p, x, at): output column x from at (i) = the sample of source column i (raw, or already vertically scaled)
//   vpass (p, y, at): output row y from at (r) = the sample of source row r (raw, or already horizontally scaled)
template <class At>
__device__ __forceinline__ int hpass (const PlaneScaleParams &p, int x, At at)
{
  switch (p.hmode) {
    case 0: return at (x);
    case 1: {                                        // 2-tap, edge-aligned 16.16 increment (1 x u8)
      const uint32_t t = (uint32_t) x * p.hinc;
      const int i = min ((int) (t >> 16), p.w - 1), f = (int) ((t >> 8) & 0xff), i1 = min (i + 1, p.w - 1);
      return (at (i) * (256 - f) + at (i1) * f) >> 8;
    }
    case 2: return (at (2 * x) + at (2 * x + 1) + 1) >> 1;      // exactly halved (1 x u8)
    case 3: case 5: {                                // 2-tap, centre-aligned 6-bit table (also nearest: second tap 0); 5 = the table is (2k, 2k+1, 32)
      const int4 e = reinterpret_cast<const int4 *> (p.htab)[x];          // one 16-byte load per table entry
      return (at (e.x) * (64 - e.z) + at (e.y) * e.z + 32) >> 6;
    }
    case 6: return at (p.htab[4 * x]);               // nearest: one source column
    default: {                                       // n taps, 6-bit (catrom; un-limited linear)
      int acc = 32;
      for (int l = 0; l < p.nh; l++) { const int2 e = p.hnt[x * p.nh + l]; acc += at (e.x) * e.y; }
      return min (max (acc >> 6, 0), 255);
    }
  }
}

template <class At>
__device__ __forceinline__ int vpass (const PlaneScaleParams &p, int y, At at)
{
  switch (p.vmode) {
    case 0: return at (y);
    case 1: {                                        // 2-tap, 8-bit: only the second tap is used
      const int4 e = reinterpret_cast<const int4 *> (p.vtab)[y];
      const int a = at (e.x), b = at (e.y);
      return a + (((b - a) * e.z + 128) >> 8);
    }
    case 3: return at (p.vtab[4 * y]);               // nearest: one source row
    default: {
      int acc = 32;
      for (int l = 0; l < p.nv; l++) { const int2 e = p.vnt[y * p.nv + l]; acc += at (e.x) * e.y; }
      return min (max (acc >> 6, 0), 255);
    }
  }
}

// one output sample (x, y), component c, of a plane whose first sample is at `base`
__device__ __forceinline__ int plane_sample (const PlaneScaleParams &p, const uint8_t *base, int x, int y, int c)
{
  auto raw = [&] (int r, int i) { return (int) base[(size_t) r * p.is + p.istep * i + c]; };
  if (p.vfirst) return hpass (p, x, [&] (int i) { return vpass (p, y, [&] (int r) { return raw (r, i); }); });
  return vpass (p, y, [&] (int r) { return hpass (p, x, [&] (int i) { return raw (r, i); }); });
}

// a plane of its own (n = istep = ostep: 4:2:0 luma / chroma planes).  One lane = G groups of FOUR consecutive output bytes
// of a row (4 samples of a 1 x u8 plane, 2 samples of NV12's 2 x u8 plane), stored as dwords when the address allows:
// byte stores cost a full store instruction each.  blockIdx.z = frame of the batch.
// F = which fast paths this instantiation carries (each kernel stays small: the gather paths are latency-bound and live on
// occupancy): 0 = contiguous source bytes (hmode 0 / 2 / 5), 1 = 2-tap gathers (hmode 1 / 3), 2 = none (n-tap tables)
template <int F>
__device__ __forceinline__ uint32_t plane_group (const PlaneScaleParams &p, const uint8_t *r0, const uint8_t *r1, int wt, bool v2, bool vecok, int bx, int wb, int y)
{
  const bool fast = vecok && bx + 3 < wb;
  uint32_t v = 0;
  if (F == 0 && fast && p.hmode == 0) {
    // no horizontal pass: four consecutive bytes of each source row are one dword
    const uint32_t a4 = *reinterpret_cast<const uint32_t *> (r0 + bx);
    if (!v2) v = a4;
    else {
      const uint32_t b4 = *reinterpret_cast<const uint32_t *> (r1 + bx);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int a = (a4 >> (8 * k)) & 0xff, b = (b4 >> (8 * k)) & 0xff;
        v |= (uint32_t) (a + (((b - a) * wt + 128) >> 8)) << (8 * k);
      }
    }
  } else if (F == 0 && fast && p.hmode == 2) {
    // exactly halved 1 x u8 plane: eight consecutive source bytes per row = two dwords
    const uint32_t *s0 = reinterpret_cast<const uint32_t *> (r0 + 2 * bx), *s1 = reinterpret_cast<const uint32_t *> (r1 + 2 * bx);
    const uint32_t a8[2] = { s0[0], s0[1] };
    uint32_t b8[2] = { a8[0], a8[1] };
    if (v2) { b8[0] = s1[0]; b8[1] = s1[1]; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int sh = 16 * (k & 1);
      const int a0 = (a8[k >> 1] >> sh) & 0xff, a1 = (a8[k >> 1] >> (sh + 8)) & 0xff, b0 = (b8[k >> 1] >> sh) & 0xff, b1 = (b8[k >> 1] >> (sh + 8)) & 0xff;
      int r;
      if (!v2) r = (a0 + a1 + 1) >> 1;
      else if (p.vfirst) { const int l = a0 + (((b0 - a0) * wt + 128) >> 8), m = a1 + (((b1 - a1) * wt + 128) >> 8); r = (l + m + 1) >> 1; }
      else { const int l = (a0 + a1 + 1) >> 1, m = (b0 + b1 + 1) >> 1; r = l + (((m - l) * wt + 128) >> 8); }
      v |= (uint32_t) r << (8 * k);
    }
  } else if (F == 0 && fast && p.hmode == 5) {
    // exactly halved 2 x u8 plane (NV12 chroma): the table is (2k, 2k+1, 32) for every k, so the two source pairs of an
    // output pair are one dword and the taps are constants
    const uint32_t *s0 = reinterpret_cast<const uint32_t *> (r0 + 2 * bx), *s1 = reinterpret_cast<const uint32_t *> (r1 + 2 * bx);
    const uint32_t a8[2] = { s0[0], s0[1] };
    uint32_t b8[2] = { a8[0], a8[1] };
    if (v2) { b8[0] = s1[0]; b8[1] = s1[1]; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int sh = 8 * (k & 1);
      const int a0 = (a8[k >> 1] >> sh) & 0xff, a1 = (a8[k >> 1] >> (sh + 16)) & 0xff, b0 = (b8[k >> 1] >> sh) & 0xff, b1 = (b8[k >> 1] >> (sh + 16)) & 0xff;
      int r;
      if (!v2) r = (a0 * 32 + a1 * 32 + 32) >> 6;
      else if (p.vfirst) { const int l = a0 + (((b0 - a0) * wt + 128) >> 8), m = a1 + (((b1 - a1) * wt + 128) >> 8); r = (l * 32 + m * 32 + 32) >> 6; }
      else { const int l = (a0 * 32 + a1 * 32 + 32) >> 6, m = (b0 * 32 + b1 * 32 + 32) >> 6; r = l + (((m - l) * wt + 128) >> 8); }
      v |= (uint32_t) r << (8 * k);
    }
  } else if (F == 1 && fast && p.hmode == 3 && p.n == 2) {
    // NV12 chroma with table taps: a U/V pair is one 16-bit load, a table entry one 16-byte load
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int4 e = reinterpret_cast<const int4 *> (p.htab)[(bx >> 1) + k];
      const uint16_t *q0 = reinterpret_cast<const uint16_t *> (r0), *q1 = reinterpret_cast<const uint16_t *> (r1);
      const uint32_t a0 = q0[e.x], a1 = q0[e.y];
      uint32_t b0 = a0, b1 = a1;
      if (v2) { b0 = q1[e.x]; b1 = q1[e.y]; }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int l0 = (a0 >> (8 * c)) & 0xff, l1 = (a1 >> (8 * c)) & 0xff, m0 = (b0 >> (8 * c)) & 0xff, m1 = (b1 >> (8 * c)) & 0xff;
        int r;
        if (!v2) r = (l0 * (64 - e.z) + l1 * e.z + 32) >> 6;
        else if (p.vfirst) { const int l = l0 + (((m0 - l0) * wt + 128) >> 8), m = l1 + (((m1 - l1) * wt + 128) >> 8); r = (l * (64 - e.z) + m * e.z + 32) >> 6; }
        else { const int l = (l0 * (64 - e.z) + l1 * e.z + 32) >> 6, m = (m0 * (64 - e.z) + m1 * e.z + 32) >> 6; r = l + (((m - l) * wt + 128) >> 8); }
        v |= (uint32_t) r << (16 * k + 8 * c);
      }
    }
  } else if (F == 1 && fast && p.hmode == 1) {
    // edge-aligned 16.16 taps on a 1 x u8 plane: the two taps are neighbouring bytes -> one (unaligned) 16-bit load per row
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t t = (uint32_t) (bx + k) * p.hinc;
      const int i = min ((int) (t >> 16), p.w - 1), f = (int) ((t >> 8) & 0xff);
      uint16_t pa, pb;
      if (i + 1 < p.w) { __builtin_memcpy (&pa, r0 + i, 2); __builtin_memcpy (&pb, r1 + i, 2); }
      else { pa = (uint16_t) (r0[i] * 0x101u); pb = (uint16_t) (r1[i] * 0x101u); }
      const int a0 = pa & 0xff, a1 = pa >> 8, b0 = pb & 0xff, b1 = pb >> 8;
      int r;
      if (!v2) r = (a0 * (256 - f) + a1 * f) >> 8;
      else if (p.vfirst) { const int l = a0 + (((b0 - a0) * wt + 128) >> 8), m = a1 + (((b1 - a1) * wt + 128) >> 8); r = (l * (256 - f) + m * f) >> 8; }
      else { const int l = (a0 * (256 - f) + a1 * f) >> 8, m = (b0 * (256 - f) + b1 * f) >> 8; r = l + (((m - l) * wt + 128) >> 8); }
      v |= (uint32_t) r << (8 * k);
    }
  } else {
#pragma unroll 1                                   // (rolled: unrolled it sets the register count of every instantiation)
    for (int k = 0; k < 4; k++) {
      const int b = min (bx + k, wb - 1);
      const int x = p.n == 2 ? b >> 1 : b, c = p.n == 2 ? b & 1 : 0;
      v |= (uint32_t) plane_sample (p, p.in, x, y, c) << (8 * k);
    }
  }
  return v;
}

template <int G, int F>
__global__ __launch_bounds__ (256) void k_scale_plane (const PlaneScaleParams p0)
{
  PlaneScaleParams p = p0;
  p.in += (size_t) blockIdx.z * p.in_pitch; p.out += (size_t) blockIdx.z * p.out_pitch;
  const int bx0 = 4 * G * (blockIdx.x * 64 + threadIdx.x), y = blockIdx.y * 4 + threadIdx.y, wb = p.n * p.ow;
  if (bx0 >= wb || y >= p.oh) return;
  int i0 = y, i1 = y, wt = 0;
  const bool v2 = p.vmode == 1, vecok = p.vec && p.vmode <= 1;      // the dword / 16-bit paths cover the 2-tap modes
  if (v2) { i0 = p.vtab[4 * y]; i1 = p.vtab[4 * y + 1]; wt = p.vtab[4 * y + 2]; }
  const uint8_t *r0 = p.in + (size_t) i0 * p.is, *r1 = p.in + (size_t) i1 * p.is;
  uint32_t v[G];
#pragma unroll
  for (int g = 0; g < G; g++) v[g] = bx0 + 4 * g < wb ? plane_group<F> (p, r0, r1, wt, v2, vecok, bx0 + 4 * g, wb, y) : 0u;
  uint8_t *d = p.out + (size_t) y * p.os + bx0;
  if (G == 2 && bx0 + 7 < wb && ((uintptr_t) d & 7) == 0) { *reinterpret_cast<uint2 *> (d) = make_uint2 (v[0], v[G - 1]); return; }
#pragma unroll
  for (int g = 0; g < G; g++) {
    const int bx = bx0 + 4 * g;
    if (bx >= wb) break;
    if (bx + 3 < wb && ((uintptr_t) (d + 4 * g) & 3) == 0) *reinterpret_cast<uint32_t *> (d + 4 * g) = v[g];
    else for (int k = 0; k < 4 && bx + k < wb; k++) d[4 * g + k] = (uint8_t) (v[g] >> (8 * k));
  }
}

// ---- n-tap passes as two kernels through an intermediate plane (method=bicubic on a plane that is scaled both ways): n_v + n_h
// loads per sample instead of the n_v x n_h of the composed form above.  Same arithmetic: every pass clamp ((sum + 32) >> 6).
struct PlaneTapParams {
  const uint8_t *in; int is;
  uint8_t *out; int os;
  int wb, rows;            // output row bytes (n * samples) and output rows of THIS pass
  int n;                   // components per sample (horizontal pass)
  const int2 *tab; int nt; // out * nt * {source index, 6-bit tap}
  size_t in_pitch, out_pitch;
  int vec;                 // source rows 4-byte aligned (vertical pass: dword loads)
};

__device__ __forceinline__ void store4 (uint8_t *d, uint32_t v, int bx, int wb)
{
  if (bx + 3 < wb && ((uintptr_t) d & 3) == 0) *reinterpret_cast<uint32_t *> (d) = v;
  else for (int k = 0; k < 4 && bx + k < wb; k++) d[k] = (uint8_t) (v >> (8 * k));
}

__global__ __launch_bounds__ (256) void k_plane_vtap (const PlaneTapParams p)
{
  const int bx = 4 * (blockIdx.x * 64 + threadIdx.x), y = blockIdx.y * 4 + threadIdx.y;
  if (bx >= p.wb || y >= p.rows) return;
  const uint8_t *in = p.in + (size_t) blockIdx.z * p.in_pitch;
  int acc[4] = { 32, 32, 32, 32 };
  const bool wide = p.vec && bx + 3 < p.wb;
  for (int l = 0; l < p.nt; l++) {
    const int2 e = p.tab[y * p.nt + l];
    const uint8_t *row = in + (size_t) e.x * p.is + bx;
    if (wide) {
      const uint32_t v = *reinterpret_cast<const uint32_t *> (row);
#pragma unroll
      for (int k = 0; k < 4; k++) acc[k] += (int) ((v >> (8 * k)) & 0xff) * e.y;
    } else {
      for (int k = 0; k < 4 && bx + k < p.wb; k++) acc[k] += row[k] * e.y;
    }
  }
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) v |= (uint32_t) min (max (acc[k] >> 6, 0), 255) << (8 * k);
  store4 (p.out + (size_t) blockIdx.z * p.out_pitch + (size_t) y * p.os + bx, v, bx, p.wb);
}

__global__ __launch_bounds__ (256) void k_plane_htap (const PlaneTapParams p)
{
  const int bx = 4 * (blockIdx.x * 64 + threadIdx.x), y = blockIdx.y * 4 + threadIdx.y;
  if (bx >= p.wb || y >= p.rows) return;
  const uint8_t *row = p.in + (size_t) blockIdx.z * p.in_pitch + (size_t) y * p.is;
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int b = min (bx + k, p.wb - 1), x = p.n == 2 ? b >> 1 : b, c = p.n == 2 ? b & 1 : 0;
    int acc = 32;
    for (int l = 0; l < p.nt; l++) { const int2 e = p.tab[x * p.nt + l]; acc += row[e.x * p.n + c] * e.y; }
    v |= (uint32_t) min (max (acc >> 6, 0), 255) << (8 * k);
  }
  store4 (p.out + (size_t) blockIdx.z * p.out_pitch + (size_t) y * p.os + bx, v, bx, p.wb);
}

// add-borders with an RGB output whose rectangle another kernel writes (method=bicubic): the colour everywhere else
struct RgbBorderParams { uint8_t *out; int os; size_t pitch; int w, h, rx, ry, rw, rh; uint32_t colour; };

__global__ __launch_bounds__ (256) void k_border_fill_rgb (const RgbBorderParams p)
{
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= p.w || y >= p.h || (x >= p.rx && x < p.rx + p.rw && y >= p.ry && y < p.ry + p.rh)) return;
  uint8_t *d = p.out + (size_t) blockIdx.z * p.pitch + (size_t) y * p.os + 4 * (size_t) x;
  if (((uintptr_t) d & 3) == 0) *reinterpret_cast<uint32_t *> (d) = p.colour;
  else { d[0] = (uint8_t) p.colour; d[1] = (uint8_t) (p.colour >> 8); d[2] = (uint8_t) (p.colour >> 16); d[3] = (uint8_t) (p.colour >> 24); }
}

// add-borders with a YUV output: every sample OUTSIDE the destination rectangle gets the border colour (already through the
// RGB -> YUV matrix).  One lane = one column pair (x, x+1) of one row; the rectangle sits on chroma-sample boundaries.
struct BorderFillParams {
  uint8_t *p[3]; int s[3];
  int fmt, w, h, rx, ry, rw, rh;
  int yuv[3];
  size_t pitch;
};

__global__ __launch_bounds__ (256) void k_border_fill_yuv (const BorderFillParams p)
{
  const int k = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, x = 2 * k;
  if (x >= p.w || y >= p.h) return;
  if (x >= p.rx && x < p.rx + p.rw && y >= p.ry && y < p.ry + p.rh) return;       // (x even, rx / rw even: the pair is in or out as one)
  const size_t f = (size_t) blockIdx.z * p.pitch;
  const uint8_t Y = (uint8_t) p.yuv[0], U = (uint8_t) p.yuv[1], V = (uint8_t) p.yuv[2];
  if (p.fmt == VFHIP_FORMAT_UYVY || p.fmt == VFHIP_FORMAT_YUY2) {
    uint8_t *d = p.p[0] + f + (size_t) y * p.s[0] + 4 * k;
    if (p.fmt == VFHIP_FORMAT_YUY2) { d[0] = Y; d[1] = U; d[2] = Y; d[3] = V; } else { d[0] = U; d[1] = Y; d[2] = V; d[3] = Y; }
    return;
  }
  uint8_t *d = p.p[0] + f + (size_t) y * p.s[0] + x;
  d[0] = Y; if (x + 1 < p.w) d[1] = Y;
  if (!(y & 1)) {                                            // the chroma sample of this 2x2 block (ry / rh even: in or out as one)
    if (p.fmt == VFHIP_FORMAT_NV12) { uint8_t *c = p.p[1] + f + (size_t) (y >> 1) * p.s[1] + 2 * k; c[0] = U; c[1] = V; }
    else { p.p[1][f + (size_t) (y >> 1) * p.s[1] + k] = U; p.p[2][f + (size_t) (y >> 1) * p.s[2] + k] = V; }
  }
}

// videoscale on a packed 4:2:2 frame: one lane = one output macro-pixel (Y0 U Y1 V in the frame's byte order), the three
// interleaved lines each with their own tables; pl[0] = luma (step 2), pl[1] = U, pl[2] = V (step 4)
struct PackedScaleParams { PlaneScaleParams pl[3]; int yo, uo, vo; int fast; };   // fast: every line 2-tap-table scaled horizontally, <= 2 taps vertically

__global__ __launch_bounds__ (256) void k_scale_packed422 (const PackedScaleParams q)
{
  const int k = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (k >= q.pl[1].ow || y >= q.pl[0].oh) return;
  const size_t fin = (size_t) blockIdx.z * q.pl[0].in_pitch, fout = (size_t) blockIdx.z * q.pl[0].out_pitch;
  // (constant indices only: a dynamically indexed parameter struct or byte array ends up in scratch memory)
  uint32_t Y0, Y1, U, V;
  if (q.fast) {
    // 2-tap tables both ways: the two taps of a luma sample lie in ONE 4-byte window of the row, the U and V taps of a chroma
    // sample in ONE 8-byte window (two neighbouring macro-pixels) -> 6 loads per output macro-pixel instead of 16
    const PlaneScaleParams &py = q.pl[0];
    int r0i = y, r1i = y, wt = 0;
    if (py.vmode == 1) { const int4 e = reinterpret_cast<const int4 *> (py.vtab)[y]; r0i = e.x; r1i = e.y; wt = e.z; }
    const uint8_t *base = py.in - q.yo + fin;                          // first byte of the frame's rows (pl[0].in points at the first luma byte)
    const uint8_t *r0 = base + (size_t) r0i * py.is, *r1 = base + (size_t) r1i * py.is;
    const bool vf = py.vfirst != 0, vs = py.vmode == 1;
    auto mix = [&] (int a0, int a1, int b0, int b1, int t) {           // a = taps in row r0, b = in row r1
      if (!vs) return (a0 * (64 - t) + a1 * t + 32) >> 6;
      if (vf) { const int l = a0 + (((b0 - a0) * wt + 128) >> 8), m = a1 + (((b1 - a1) * wt + 128) >> 8); return (l * (64 - t) + m * t + 32) >> 6; }
      const int l = (a0 * (64 - t) + a1 * t + 32) >> 6, m = (b0 * (64 - t) + b1 * t + 32) >> 6;
      return l + (((m - l) * wt + 128) >> 8);
    };
    auto luma = [&] (int x) -> uint32_t {
      const int4 e = reinterpret_cast<const int4 *> (py.htab)[x];
      if (e.y != e.x + 1) return (uint32_t) plane_sample (py, py.in + fin, x, y, 0);      // clamped at the row end
      uint32_t a, b;
      __builtin_memcpy (&a, r0 + 2 * e.x, 4); __builtin_memcpy (&b, r1 + 2 * e.x, 4);
      const int sh = 8 * q.yo;
      return (uint32_t) mix ((a >> sh) & 0xff, (a >> (sh + 16)) & 0xff, (b >> sh) & 0xff, (b >> (sh + 16)) & 0xff, e.z);
    };
    Y0 = luma (2 * k);
    Y1 = 2 * k + 1 < py.ow ? luma (2 * k + 1) : Y0;                    // spare slot of an odd width
    const int4 e = reinterpret_cast<const int4 *> (q.pl[1].htab)[k];
    if (e.y != e.x + 1) {
      U = (uint32_t) plane_sample (q.pl[1], q.pl[1].in + fin, k, y, 0);
      V = (uint32_t) plane_sample (q.pl[2], q.pl[2].in + fin, k, y, 0);
    } else {
      uint2 a, b;
      __builtin_memcpy (&a, r0 + 4 * e.x, 8); __builtin_memcpy (&b, r1 + 4 * e.x, 8);
      const int su = 8 * q.uo, sv = 8 * q.vo;
      U = (uint32_t) mix ((a.x >> su) & 0xff, (a.y >> su) & 0xff, (b.x >> su) & 0xff, (b.y >> su) & 0xff, e.z);
      V = (uint32_t) mix ((a.x >> sv) & 0xff, (a.y >> sv) & 0xff, (b.x >> sv) & 0xff, (b.y >> sv) & 0xff, e.z);
    }
  } else {
    Y0 = (uint32_t) plane_sample (q.pl[0], q.pl[0].in + fin, 2 * k, y, 0);
    Y1 = 2 * k + 1 < q.pl[0].ow ? (uint32_t) plane_sample (q.pl[0], q.pl[0].in + fin, 2 * k + 1, y, 0) : Y0;   // spare slot of an odd width
    U = (uint32_t) plane_sample (q.pl[1], q.pl[1].in + fin, k, y, 0);
    V = (uint32_t) plane_sample (q.pl[2], q.pl[2].in + fin, k, y, 0);
  }
  uint8_t *d = q.pl[0].out + fout + (size_t) y * q.pl[0].os + 4 * k;
  const uint32_t v = q.yo == 0 ? (Y0 | U << 8 | Y1 << 16 | V << 24) : (U | Y0 << 8 | V << 16 | Y1 << 24);
  if (((uintptr_t) d & 3) == 0) *reinterpret_cast<uint32_t *> (d) = v;
  else { d[0] = (uint8_t) v; d[1] = (uint8_t) (v >> 8); d[2] = (uint8_t) (v >> 16); d[3] = (uint8_t) (v >> 24); }
}

// ---- packed 4:2:2 (UYVY / YUY2) outputs and packed -> 4:2:0: videoconvert's conversions at the input size ------------
// (oracle/gst114.c gst114_rgb_to_packed422 / gst114_yuv420_to_packed422 / gst114_packed422_swizzle /
// gst114_packed422_to_yuv420; every rule pinned by probing the real 1.14 element, tests/golden/convertscale_gst114_packedout.npz)

// horizontally up-sampled chroma at full-resolution column x of a row of cw samples `step` bytes apart
__device__ __forceinline__ int chroma_up_h (const uint8_t *row, int step, int cw, int x, int cosited)
{
  const int k = x >> 1;
  const int c0 = row[k * step];
  if (cosited) return (x & 1) ? (c0 + row[min (k + 1, cw - 1) * step] + 1) >> 1 : c0;
  return (x & 1) ? (3 * c0 + row[min (k + 1, cw - 1) * step] + 2) >> 2 : (3 * c0 + row[max (k - 1, 0) * step] + 2) >> 2;
}

// the three full-resolution columns the horizontal 2:1 down-sampling of chroma sample k reads, and its result
__device__ __forceinline__ void chroma_down_cols (int k, int cw, int w, int cosited, int xs[3])
{
  const int x = 2 * k;
  xs[1] = x;
  if (cosited) { xs[0] = x > 0 ? x - 1 : 0; xs[2] = (k == cw - 1 && k > 0) ? x : min (x + 1, w - 1); }
  else { xs[0] = x; xs[2] = min (x + 1, w - 1); }
}
__device__ __forceinline__ int chroma_down (const int s[3], int cosited)
{
  return cosited ? (s[0] + 2 * s[1] + s[2] + 2) >> 2 : (s[1] + s[2] + 1) >> 1;
}

// Full-resolution chroma at columns x-1, x, x+1 (x = 2k) from the three samples cm, c0, cp of one row (already clamped at
// the row ends): the same values chroma_up_h gives, without re-reading the row
__device__ __forceinline__ void chroma_up3 (int cm, int c0, int cp, int cosited, int &L, int &C, int &R)
{
  if (cosited) { C = c0; R = (c0 + cp + 1) >> 1; L = (cm + c0 + 1) >> 1; }
  else { C = (3 * c0 + cm + 2) >> 2; R = (3 * c0 + cp + 2) >> 2; L = (3 * cm + c0 + 2) >> 2; }
}
// horizontal 2:1 down-sampling of chroma sample k from the columns L (x-1), C (x), R (x+1), with chroma_down_cols' edge rules
__device__ __forceinline__ int chroma_down3 (int L, int C, int R, int k, int cw, int w, int cosited)
{
  if (2 * k + 1 > w - 1) R = C;                        // odd width: the last column pairs with itself
  if (!cosited) return (C + R + 1) >> 1;
  if (k == cw - 1 && k > 0) R = C;                     // the last sample ignores its right neighbour, unless it is also the first
  if (k == 0) L = C;
  return (L + 2 * C + R + 2) >> 2;
}

struct ToPackedParams {
  size_t in_pitch, out_pitch;
  const uint8_t *in[3]; int is[3];
  uint8_t *out; int os;
  int w, h, in_fmt, out_yuy2, cosited_in, cosited_out;
  int vec;                 // NV12: rows 2-byte aligned (a U/V or luma pair is one 16-bit load); RGB: rows 4-byte aligned (dword pixels)
  int c[9];                // RGB -> YUV matrix (RGB inputs)
  uint32_t cy, cup, cun, cvp, cvn;     // the same coefficients packed per byte lane of the input, positive and negative parts apart
};

__device__ __forceinline__ uint32_t pack_macro (int yuy2, int Y0, int Y1, int U, int V)
{
  return yuy2 ? (uint32_t) Y0 | (uint32_t) U << 8 | (uint32_t) Y1 << 16 | (uint32_t) V << 24
              : (uint32_t) U | (uint32_t) Y0 << 8 | (uint32_t) V << 16 | (uint32_t) Y1 << 24;
}

// macro-pixel k of row y (two luma samples + U + V) of the packed output; p's pointers already at the frame
__device__ __forceinline__ uint32_t to_packed_one (const ToPackedParams &p, int k, int y)
{
  const int cw = (p.w + 1) >> 1;
  const int x = 2 * k, x1 = min (x + 1, p.w - 1);
  int Y0, Y1, U, V;
  switch (p.in_fmt) {
    case VFHIP_FORMAT_BGRA: case VFHIP_FORMAT_RGBA: {
      const int ro = p.in_fmt == VFHIP_FORMAT_RGBA ? 0 : 2, bo = 2 - ro;
      const uint8_t *row = p.in[0] + (size_t) y * p.is[0];
      int xs[3], su[3], sv[3];
      chroma_down_cols (k, cw, p.w, p.cosited_out, xs);
      if (p.vec) {
        // 4-byte-aligned rows: one dword load per pixel, every matrix row one or two v_dot4_u32_u8 (like k_rgb_to_yuv420_fast)
        const uint32_t *r32 = reinterpret_cast<const uint32_t *> (row);
        const uint32_t pc = r32[x], pr = r32[x1];
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const uint32_t px = t == 1 ? pc : (xs[t] == x ? pc : (xs[t] == x1 ? pr : r32[xs[t]]));
          su[t] = (((int) __builtin_amdgcn_udot4 (px, p.cup, 0u, false) - (int) __builtin_amdgcn_udot4 (px, p.cun, 0u, false)) >> 8) + 128;
          sv[t] = (((int) __builtin_amdgcn_udot4 (px, p.cvp, 0u, false) - (int) __builtin_amdgcn_udot4 (px, p.cvn, 0u, false)) >> 8) + 128;
        }
        U = chroma_down (su, p.cosited_out); V = chroma_down (sv, p.cosited_out);
        Y0 = (int) (__builtin_amdgcn_udot4 (pc, p.cy, 0u, false) >> 8) + 16;
        Y1 = (int) (__builtin_amdgcn_udot4 (pr, p.cy, 0u, false) >> 8) + 16;
        break;
      }
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const uint8_t *px = row + 4 * xs[t];
        const int r = px[ro], g = px[1], b = px[bo];
        su[t] = ((p.c[3] * r + p.c[4] * g + p.c[5] * b) >> 8) + 128;
        sv[t] = ((p.c[6] * r + p.c[7] * g + p.c[8] * b) >> 8) + 128;
      }
      U = chroma_down (su, p.cosited_out); V = chroma_down (sv, p.cosited_out);
      const uint8_t *a = row + 4 * x, *b = row + 4 * x1;
      Y0 = ((p.c[0] * a[ro] + p.c[1] * a[1] + p.c[2] * a[bo]) >> 8) + 16;
      Y1 = ((p.c[0] * b[ro] + p.c[1] * b[1] + p.c[2] * b[bo]) >> 8) + 16;
      break;
    }
    case VFHIP_FORMAT_I420: {          // fast path: chroma row y >> 1 as it is
      const int j = y >> 1;
      U = p.in[1][(size_t) j * p.is[1] + k]; V = p.in[2][(size_t) j * p.is[2] + k];
      Y0 = p.in[0][(size_t) y * p.is[0] + x]; Y1 = p.in[0][(size_t) y * p.is[0] + x1];
      break;
    }
    case VFHIP_FORMAT_NV12: {          // generic path: up h (input siting), up v (3:1 with the nearer row), down h (output siting)
      const int chh = (p.h + 1) >> 1, j = y >> 1, jn = (y & 1) ? min (j + 1, chh - 1) : max (j - 1, 0);
      const uint8_t *r0 = p.in[1] + (size_t) j * p.is[1], *r1 = p.in[1] + (size_t) jn * p.is[1];
      if (p.vec) {
        // three U/V pairs per chroma row, each one 16-bit load; the full-resolution columns x-1, x, x+1 come from them
        const int km = max (k - 1, 0), kp = min (k + 1, cw - 1);
        const uint16_t *q0 = reinterpret_cast<const uint16_t *> (r0), *q1 = reinterpret_cast<const uint16_t *> (r1);
        const uint32_t a[3] = { q0[km], q0[k], q0[kp] }, b[3] = { q1[km], q1[k], q1[kp] };
        int s[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
          int L0, C0, R0, L1, C1, R1;
          chroma_up3 ((a[0] >> (8 * c)) & 0xff, (a[1] >> (8 * c)) & 0xff, (a[2] >> (8 * c)) & 0xff, p.cosited_in, L0, C0, R0);
          chroma_up3 ((b[0] >> (8 * c)) & 0xff, (b[1] >> (8 * c)) & 0xff, (b[2] >> (8 * c)) & 0xff, p.cosited_in, L1, C1, R1);
          s[c] = chroma_down3 ((3 * L0 + L1 + 2) >> 2, (3 * C0 + C1 + 2) >> 2, (3 * R0 + R1 + 2) >> 2, k, cw, p.w, p.cosited_out);
        }
        U = s[0]; V = s[1];
        const uint8_t *yr = p.in[0] + (size_t) y * p.is[0] + x;
        if (x + 1 < p.w) { const uint32_t yy = *reinterpret_cast<const uint16_t *> (yr); Y0 = yy & 0xff; Y1 = yy >> 8; }
        else Y0 = Y1 = yr[0];
        break;
      }
      int xs[3], su[3], sv[3];
      chroma_down_cols (k, cw, p.w, p.cosited_out, xs);
#pragma unroll
      for (int t = 0; t < 3; t++) {
        su[t] = (3 * chroma_up_h (r0, 2, cw, xs[t], p.cosited_in) + chroma_up_h (r1, 2, cw, xs[t], p.cosited_in) + 2) >> 2;
        sv[t] = (3 * chroma_up_h (r0 + 1, 2, cw, xs[t], p.cosited_in) + chroma_up_h (r1 + 1, 2, cw, xs[t], p.cosited_in) + 2) >> 2;
      }
      U = chroma_down (su, p.cosited_out); V = chroma_down (sv, p.cosited_out);
      Y0 = p.in[0][(size_t) y * p.is[0] + x]; Y1 = p.in[0][(size_t) y * p.is[0] + x1];
      break;
    }
    default: {                         // UYVY <-> YUY2: byte swizzle
      const uint8_t *m = p.in[0] + (size_t) y * p.is[0] + 4 * k;
      const int yo = p.in_fmt == VFHIP_FORMAT_YUY2 ? 0 : 1;
      Y0 = m[yo]; Y1 = x + 1 < p.w ? m[yo + 2] : m[yo]; U = m[1 - yo]; V = m[3 - yo];
      break;
    }
  }
  return pack_macro (p.out_yuy2, Y0, Y1, U, V);
}

__device__ __forceinline__ void store_macro (uint8_t *d, uint32_t v)
{
  if (((uintptr_t) d & 3) == 0) *reinterpret_cast<uint32_t *> (d) = v;
  else { d[0] = (uint8_t) v; d[1] = (uint8_t) (v >> 8); d[2] = (uint8_t) (v >> 16); d[3] = (uint8_t) (v >> 24); }
}

// one lane = one macro-pixel
__global__ __launch_bounds__ (256) void k_to_packed422 (const ToPackedParams p0)
{
  ToPackedParams p = p0;
  for (int t = 0; t < 3; t++) if (p.in[t]) p.in[t] += (size_t) blockIdx.z * p.in_pitch;
  p.out += (size_t) blockIdx.z * p.out_pitch;
  const int k = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (k >= ((p.w + 1) >> 1) || y >= p.h) return;
  store_macro (p.out + (size_t) y * p.os + 4 * k, to_packed_one (p, k, y));
}

struct FromPackedParams {
  size_t in_pitch, out_pitch;
  const uint8_t *in; int is;
  uint8_t *y, *u, *v; int ys, us, vs;     // NV12: u = uv plane, v unused
  int w, h, in_yuy2, planar, cosited_in, cosited_out;
  int vec;                 // packed rows are 4-byte aligned: a macro-pixel is one dword load
};

// one lane = one 4:2:0 chroma sample (and the 2x2 luma block under it)
__global__ __launch_bounds__ (256) void k_packed422_to_420 (const FromPackedParams p0)
{
  FromPackedParams p = p0;
  p.in += (size_t) blockIdx.z * p.in_pitch;
  p.y += (size_t) blockIdx.z * p.out_pitch; p.u += (size_t) blockIdx.z * p.out_pitch; if (p.v) p.v += (size_t) blockIdx.z * p.out_pitch;
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  const int cw = (p.w + 1) >> 1, chh = (p.h + 1) >> 1;
  if (k >= cw || j >= chh) return;
  const int yo = p.in_yuy2 ? 0 : 1, uo = 1 - yo, vo = 3 - yo;
  const uint8_t *r0 = p.in + (size_t) (2 * j) * p.is, *r1 = p.in + (size_t) min (2 * j + 1, p.h - 1) * p.is;
  int U, V;
  if (p.planar) {                      // fast path: vertical pair average, no horizontal step
    U = (r0[4 * k + uo] + r1[4 * k + uo] + 1) >> 1; V = (r0[4 * k + vo] + r1[4 * k + vo] + 1) >> 1;
    p.u[(size_t) j * p.us + k] = (uint8_t) U; p.v[(size_t) j * p.vs + k] = (uint8_t) V;
  } else if (p.vec) {                  // generic path on whole macro-pixels: three dword loads per row give U, V and the luma block
    const int km = max (k - 1, 0), kp = min (k + 1, cw - 1);
    const uint32_t *q0 = reinterpret_cast<const uint32_t *> (r0), *q1 = reinterpret_cast<const uint32_t *> (r1);
    const uint32_t a[3] = { q0[km], q0[k], q0[kp] }, b[3] = { q1[km], q1[k], q1[kp] };
    int s[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int sh = 8 * (c ? vo : uo);
      int L0, C0, R0, L1, C1, R1;
      chroma_up3 ((a[0] >> sh) & 0xff, (a[1] >> sh) & 0xff, (a[2] >> sh) & 0xff, p.cosited_in, L0, C0, R0);
      chroma_up3 ((b[0] >> sh) & 0xff, (b[1] >> sh) & 0xff, (b[2] >> sh) & 0xff, p.cosited_in, L1, C1, R1);
      s[c] = chroma_down3 ((L0 + L1 + 1) >> 1, (C0 + C1 + 1) >> 1, (R0 + R1 + 1) >> 1, k, cw, p.w, p.cosited_out);
    }
    uint8_t *d = p.u + (size_t) j * p.us + 2 * k;
    if (((uintptr_t) d & 1) == 0) *reinterpret_cast<uint16_t *> (d) = (uint16_t) (s[0] | (s[1] << 8));
    else { d[0] = (uint8_t) s[0]; d[1] = (uint8_t) s[1]; }
    // luma: the two samples of each row from the macro-pixels already loaded, one 2-byte store per row when it can
    const bool two = 2 * k + 1 < p.w;
#pragma unroll
    for (int dd = 0; dd < 2; dd++) {
      const int yy = 2 * j + dd;
      if (yy >= p.h) break;
      const uint32_t m = dd ? b[1] : a[1];
      const uint32_t l0 = (m >> (8 * yo)) & 0xff, l1 = (m >> (8 * yo + 16)) & 0xff;
      uint8_t *o = p.y + (size_t) yy * p.ys + 2 * k;
      if (two && ((uintptr_t) o & 1) == 0) *reinterpret_cast<uint16_t *> (o) = (uint16_t) (l0 | (l1 << 8));
      else { o[0] = (uint8_t) l0; if (two) o[1] = (uint8_t) l1; }
    }
    return;
  } else {                             // generic path: up h (input siting), vertical pair average, down h (output siting)
    int xs[3], su[3], sv[3];
    chroma_down_cols (k, cw, p.w, p.cosited_out, xs);
#pragma unroll
    for (int t = 0; t < 3; t++) {
      su[t] = (chroma_up_h (r0 + uo, 4, cw, xs[t], p.cosited_in) + chroma_up_h (r1 + uo, 4, cw, xs[t], p.cosited_in) + 1) >> 1;
      sv[t] = (chroma_up_h (r0 + vo, 4, cw, xs[t], p.cosited_in) + chroma_up_h (r1 + vo, 4, cw, xs[t], p.cosited_in) + 1) >> 1;
    }
    U = chroma_down (su, p.cosited_out); V = chroma_down (sv, p.cosited_out);
    p.u[(size_t) j * p.us + 2 * k] = (uint8_t) U; p.u[(size_t) j * p.us + 2 * k + 1] = (uint8_t) V;
  }
#pragma unroll
  for (int d = 0; d < 2; d++)
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int xx = 2 * k + e, yy = 2 * j + d;
      if (xx < p.w && yy < p.h) p.y[(size_t) yy * p.ys + xx] = p.in[(size_t) yy * p.is + 2 * xx + yo];
    }
}

}  // namespace vfhip
