// csrc/convertscale_planar_kernels.h — gst-exact cells of vfhipconvertscale whose OUTPUT is 4:2:0 (NV12 / I420).
//
// GStreamer's `videoconvert ! videoscale` does these in two steps and so do we (bit-exact, oracle/gst114.c
// gst114_rgb_to_yuv420 / gst114_scale_plane; rules pinned by probing the real 1.14 elements):
//   stage 1  videoconvert at the INPUT size: BGRA/RGBA -> NV12/I420 (8-bit integer matrix, chroma averaged vertically
//            then horizontally), or NV12 <-> I420 re-packing;
//   stage 2  videoscale plane by plane: luma / I420 chroma as 1 x u8 (edge-aligned 16.16 horizontal taps, or pair
//            averaging when exactly halved), NV12 chroma as 2 x u8 (centre-aligned, 6-bit taps), vertical 8-bit
//            centre-aligned taps, pass order per plane.
// Replaces reference rgbaToNV12 / rgbaToI420 (common/vfmetalshaders.m:90-168) for numerics=gst-exact.
// These kernels are correct-first (one sample per lane); the headline path is k_cs_nv12_half.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vfhip {

struct Rgb2YuvParams {
  const uint8_t *in; int is;
  uint8_t *y, *u, *v; int ys, us, vs;     // NV12: u = uv plane, v unused
  int w, h, in_rgba, planar, cosited;
  int c[9];
};

__global__ __launch_bounds__ (256) void k_rgb_to_yuv420 (const Rgb2YuvParams p)
{
  const int k = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;      // chroma sample (k, j)
  const int cw = (p.w + 1) >> 1, chh = (p.h + 1) >> 1;
  if (k >= cw || j >= chh) return;
  const int ro = p.in_rgba ? 0 : 2, bo = 2 - ro;
  const int x = 2 * k;
  // vertical averages (a+b+1)>>1 of the converted chroma at columns x-1, x, x+1 (edge rules: see oracle/gst114.c)
  int xs[3];
  xs[1] = x;
  if (p.cosited) { xs[0] = x > 0 ? x - 1 : 0; xs[2] = (k == cw - 1) ? x : min (x + 1, p.w - 1); }
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

struct RepackParams {
  const uint8_t *iy, *iu, *iv; int iys, ius, ivs;
  uint8_t *oy, *ou, *ov; int oys, ous, ovs;
  int w, h, in_planar, out_planar;
};

__global__ __launch_bounds__ (256) void k_repack_420 (const RepackParams p)
{
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

// one plane of n interleaved u8 components
struct PlaneScaleParams {
  const uint8_t *in; int is;
  uint8_t *out; int os;
  int w, h, ow, oh, n;
  int hmode;               // 0: no horizontal scaling, 1: edge-aligned 16.16 (1 x u8), 2: pair average (1 x u8), 3: table (2 x u8, 6-bit taps)
  int vscale_on, vfirst;
  uint32_t hinc;
  const int *vtab;         // oh * {i0, i1, w, 0}
  const int *htab;         // hmode 3: ow * {i0, i1, t, 0}
};

__device__ __forceinline__ int plane_htap (const PlaneScaleParams &p, const uint8_t *row, int x, int c)
{
  switch (p.hmode) {
    case 0: return row[p.n * x + c];
    case 1: {
      const uint32_t t = (uint32_t) x * p.hinc;
      const int i = min ((int) (t >> 16), p.w - 1), f = (int) ((t >> 8) & 0xff), i1 = min (i + 1, p.w - 1);
      return (row[i] * (256 - f) + row[i1] * f) >> 8;
    }
    case 2: return (row[2 * x] + row[2 * x + 1] + 1) >> 1;
    default: {
      const int i0 = p.htab[4 * x], i1 = p.htab[4 * x + 1], t = p.htab[4 * x + 2];
      return (row[p.n * i0 + c] * (64 - t) + row[p.n * i1 + c] * t + 32) >> 6;
    }
  }
}

__global__ __launch_bounds__ (256) void k_scale_plane (const PlaneScaleParams p)
{
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= p.ow || y >= p.oh) return;
  int i0 = y, i1 = y, wt = 0;
  if (p.vscale_on) { i0 = p.vtab[4 * y]; i1 = p.vtab[4 * y + 1]; wt = p.vtab[4 * y + 2]; }
  const uint8_t *r0 = p.in + (size_t) i0 * p.is, *r1 = p.in + (size_t) i1 * p.is;
  for (int c = 0; c < p.n; c++) {
    int v;
    if (!p.vscale_on) v = plane_htap (p, r0, x, c);
    else if (p.hmode == 0) { const int a = r0[p.n * x + c], b = r1[p.n * x + c]; v = a + (((b - a) * wt + 128) >> 8); }
    else if (p.vfirst) {
      // vertical first: the horizontal taps run on vertically scaled samples -> scale each source column the tap touches
      switch (p.hmode) {
        case 1: {
          const uint32_t t = (uint32_t) x * p.hinc;
          const int i = min ((int) (t >> 16), p.w - 1), f = (int) ((t >> 8) & 0xff), j1 = min (i + 1, p.w - 1);
          const int a = r0[i] + (((r1[i] - r0[i]) * wt + 128) >> 8), b = r0[j1] + (((r1[j1] - r0[j1]) * wt + 128) >> 8);
          v = (a * (256 - f) + b * f) >> 8; break;
        }
        case 2: {
          const int a = r0[2 * x] + (((r1[2 * x] - r0[2 * x]) * wt + 128) >> 8), b = r0[2 * x + 1] + (((r1[2 * x + 1] - r0[2 * x + 1]) * wt + 128) >> 8);
          v = (a + b + 1) >> 1; break;
        }
        default: {
          const int j0 = p.htab[4 * x], j1 = p.htab[4 * x + 1], t = p.htab[4 * x + 2];
          const int a0 = r0[p.n * j0 + c], a1 = r1[p.n * j0 + c], b0 = r0[p.n * j1 + c], b1 = r1[p.n * j1 + c];
          const int a = a0 + (((a1 - a0) * wt + 128) >> 8), b = b0 + (((b1 - b0) * wt + 128) >> 8);
          v = (a * (64 - t) + b * t + 32) >> 6; break;
        }
      }
    } else {
      const int a = plane_htap (p, r0, x, c), b = plane_htap (p, r1, x, c);
      v = a + (((b - a) * wt + 128) >> 8);
    }
    p.out[(size_t) y * p.os + p.n * x + c] = (uint8_t) v;
  }
}

}  // namespace vfhip
