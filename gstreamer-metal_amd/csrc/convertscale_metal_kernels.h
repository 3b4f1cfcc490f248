// csrc/convertscale_metal_kernels.h — `numerics=metal` path of vfhipconvertscale: the reference's float
// pipeline (render pass at output size into an 8-bit target, then RGBA->YUV) for all 6x6 format cells,
// fused into one kernel.  Restates convertscale/metalconvertscale_shaders.h:48-269 +
// metalconvertscalerenderer.m:137-166,353-448 (viewport, clear colour, nearest-only packed YUV inputs).
#pragma once
#include "metal_common.h"
#include "vfhip_internal.h"

namespace vfhip {

struct CsMetalParams {
  metal::Img in;
  metal::OutImg out;
  size_t in_pitch, out_pitch;
  float rx, ry, rw, rh;        // letterbox quad in output pixels (float, centred)
  int linear;
  uint32_t border_rgba8;       // clear colour, logical RGBA8
};

__global__ __launch_bounds__ (256) void k_cs_metal (const CsMetalParams p)
{
  const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
  if (2 * bx >= p.out.w || 2 * by >= p.out.h) return;
  metal::Img in = p.in;
  metal::OutImg out = p.out;
  for (int k = 0; k < 3; k++) {
    if (in.p[k]) in.p[k] += (size_t) blockIdx.z * p.in_pitch;
    if (out.p[k]) out.p[k] += (size_t) blockIdx.z * p.out_pitch;
  }
  uint32_t q[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++) {
      const int x = min (2 * bx + dx, out.w - 1), y = min (2 * by + dy, out.h - 1);
      const float cx = (float) x + 0.5f, cy = (float) y + 0.5f;
      if (cx >= p.rx && cx < p.rx + p.rw && cy >= p.ry && cy < p.ry + p.rh) {
        const float u = (cx - p.rx) / p.rw, v = (cy - p.ry) / p.rh;
        q[dy][dx] = metal::quant_rgba8 (metal::sample_rgba (in, u, v, p.linear != 0));
      } else {
        q[dy][dx] = p.border_rgba8;
      }
    }
  metal::store_block (out, bx, by, q);
}

static inline int cs_metal_launch (const VfHipVideoInfo &ii, const VfHipVideoInfo &oi, int method, int add_borders,
    uint32_t border_argb, const VfHipFrame *in, VfHipFrame *out, size_t in_pitch, size_t out_pitch, int n_frames, hipStream_t s)
{
  CsMetalParams p {};
  for (int k = 0; k < 3; k++) { p.in.p[k] = (const uint8_t *) in->data[k]; p.in.s[k] = in->stride[k]; p.out.p[k] = (uint8_t *) out->data[k]; p.out.s[k] = out->stride[k]; }
  p.in.w = ii.width; p.in.h = ii.height; p.in.fmt = ii.format; p.in.m709 = ii.color_matrix == VFHIP_MATRIX_BT709;
  p.out.w = oi.width; p.out.h = oi.height; p.out.fmt = oi.format; p.out.m709 = oi.color_matrix == VFHIP_MATRIX_BT709;
  p.in_pitch = in_pitch; p.out_pitch = out_pitch;
  // reference -_computeViewportWithAddBorders: (metalconvertscalerenderer.m:137-166): NDC quad scaled about the centre
  float sx = 1.0f, sy = 1.0f;
  if (add_borders && ii.width > 0 && ii.height > 0) {
    const float src = (float) ii.width / (float) ii.height, dst = (float) oi.width / (float) oi.height;
    if (src > dst) sy = dst / src; else sx = src / dst;
  }
  p.rw = (float) oi.width * sx; p.rh = (float) oi.height * sy;
  p.rx = ((float) oi.width - p.rw) * 0.5f; p.ry = ((float) oi.height - p.rh) * 0.5f;
  // packed YUV inputs are always fetched nearest (metalconvertscalerenderer.m:184-185)
  p.linear = method == VFHIP_SCALE_BILINEAR;
  const uint32_t a = border_argb >> 24, r = (border_argb >> 16) & 0xff, g = (border_argb >> 8) & 0xff, b = border_argb & 0xff;
  p.border_rgba8 = r | (g << 8) | (b << 16) | (a << 24);
  const int bw = (oi.width + 1) / 2, bh = (oi.height + 1) / 2;
  dim3 grid ((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4), (unsigned) n_frames);
  hipLaunchKernelGGL (k_cs_metal, grid, dim3 (64, 4), 0, s, p);
  VFHIP_CHECK_HIP (hipGetLastError ());
  return VFHIP_OK;
}

}  // namespace vfhip
