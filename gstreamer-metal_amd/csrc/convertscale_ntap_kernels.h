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

}  // namespace vfhip
