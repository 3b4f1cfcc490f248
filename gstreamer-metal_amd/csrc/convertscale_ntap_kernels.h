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

__device__ __forceinline__ uint32_t ntap_finish (int a0, int a1, int a2, int a3)
{
  a0 = min (max ((a0 + 32) >> 6, 0), 255); a1 = min (max ((a1 + 32) >> 6, 0), 255);
  a2 = min (max ((a2 + 32) >> 6, 0), 255); a3 = min (max ((a3 + 32) >> 6, 0), 255);
  return (uint32_t) a0 | ((uint32_t) a1 << 8) | ((uint32_t) a2 << 16) | ((uint32_t) a3 << 24);
}

__global__ __launch_bounds__ (256) void k_ntap_v (const NtapParams p)
{
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = __builtin_amdgcn_readfirstlane ((int) (blockIdx.y * 4 + threadIdx.y));      // one row per wave
  if (y >= p.h || x >= p.w) return;
  const int2 *t = p.tab + (size_t) y * p.n;
  int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int l = 0; l < p.n; l++) {
    const int2 e = t[l];
    const uint32_t v = *reinterpret_cast<const uint32_t *> (p.in + (size_t) e.x * p.is + 4 * x);
    a0 += (int) (v & 0xff) * e.y; a1 += (int) ((v >> 8) & 0xff) * e.y; a2 += (int) ((v >> 16) & 0xff) * e.y; a3 += (int) (v >> 24) * e.y;
  }
  *reinterpret_cast<uint32_t *> (p.out + (size_t) y * p.os + 4 * x) = ntap_finish (a0, a1, a2, a3);
}

__global__ __launch_bounds__ (256) void k_ntap_h (const NtapParams p)
{
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (y >= p.h || x >= p.w) return;
  const int2 *t = p.tab + (size_t) x * p.n;
  const uint8_t *row = p.in + (size_t) y * p.is;
  int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int l = 0; l < p.n; l++) {
    const int2 e = t[l];
    const uint32_t v = *reinterpret_cast<const uint32_t *> (row + 4 * e.x);
    a0 += (int) (v & 0xff) * e.y; a1 += (int) ((v >> 8) & 0xff) * e.y; a2 += (int) ((v >> 16) & 0xff) * e.y; a3 += (int) (v >> 24) * e.y;
  }
  *reinterpret_cast<uint32_t *> (p.out + (size_t) y * p.os + 4 * x) = ntap_finish (a0, a1, a2, a3);
}

}  // namespace vfhip
