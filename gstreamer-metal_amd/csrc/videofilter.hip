// csrc/videofilter.hip — vfhip_videofilter_* : the 15-property single-pass video filter.
// Mirrors MetalVideoFilterRenderer (reference videofilter/metalvideofilterrenderer.{h,m}) and restates
// applyColorAdjustments / hash12 / rgbToHsv / hsvToRgb / filterFragment* / blurHorizontal / blurVertical /
// unsharpMask of videofilter/metalvideofilter_shaders.h:63-328 (`metal` numerics).
//
// The reference runs 1 render pass + (sharpness != 0) 3 compute passes + 1 RGBA->YUV pass, every one a full
// trip through an 8-bit texture (metalvideofilterrenderer.m:523-681).  Here:
//   k_vf_point : sharpness == 0 — one kernel, each lane owns a 2x2 block (needed by the 4:2:0 store epilogue);
//   k_vf_sharp : sharpness != 0 — one kernel, a 128x56 output tile per workgroup; the colour-adjusted tile plus
//                its 4-pixel halo is quantised to 8 bits into LDS exactly where the reference wrote its render
//                target, the 9-tap horizontal and vertical Gaussians (sliding windows in registers) and the unsharp
//                mask run out of LDS (with the reference's 8-bit requantisation between passes), and only the final
//                frame goes to HBM.
#include "vfhip_internal.h"
#include "metal_common.h"
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

using namespace vfhip;

namespace vfhip {

// The uniforms of the colour stages folded on the host (vf_fold) for the fast path, see color_fast ().
enum { VF_ON_HUE = 1, VF_ON_GAMMA = 2, VF_ON_B = 4, VF_ON_KEY = 8, VF_ON_KEY_STEP = 16, VF_ON_VIG = 32, VF_ON_NOISE = 64, VF_ON_LUT16 = 128 };
struct VfFast {
  float a[3][3], ao[3];            // brightness, contrast, saturation: one affine map
  float b[3][3], bo[3];            // sepia mix, invert: one affine map
  float hue_shift, inv_gamma;      // hue / 2 pi; 1 / gamma
  float key_r, key_g, key_b, key_e0, key_scale, key_bias;     // smoothstep argument t = clamp (dist * scale + bias); step: dist < e0 ? 0 : 1
  float vig, noise_gain;           // vignette amount; noise * .5
  float fw, fh, lut_nm1, lut_n;    // frame size, LUT size - 1, LUT size, as floats
  uint32_t on;                     // VF_ON_*
};

struct VfParams {
  metal::Img in;
  metal::OutImg out;
  VfHipVideoFilterParams u;
  VfFast f;
  const float4 *lut;               // fp32 cells (96 bytes), the exact path's table
  const uint4 *lut16;              // fp16 cells (64 bytes), the fast path's table; null when the table does not fit fp16
  int lut_size;
  int quad_in;                     // k_vf_sharp: a 4:2:0 input that meets metal::fetch420_quad's contract (the region fill takes 4 x 2 pixels at a time)
  size_t in_pitch, out_pitch;      // batch: frame blockIdx.z at base + z * pitch, frame_index + z
};

// this launch's frame of the batch
__device__ __forceinline__ VfParams vf_frame (const VfParams &p)
{
  VfParams q = p;
  q.in = metal::img_at (p.in, blockIdx.z * p.in_pitch);
  q.out = metal::out_at (p.out, blockIdx.z * p.out_pitch);
  q.u.frame_index += blockIdx.z;
  return q;
}

using metal::F4;
using metal::clamp01;

__device__ __forceinline__ float fractf_ (float x) { return x - floorf (x); }
__device__ __forceinline__ float mixf (float a, float b, float t) { return fmaf (b - a, t, a); }
__device__ __forceinline__ float stepf (float e, float x) { return x < e ? 0.0f : 1.0f; }
__device__ __forceinline__ float smoothstepf (float e0, float e1, float x)
{
  if (!(e0 < e1)) return stepf (e0, x);          // MSL leaves e0 >= e1 undefined (smoothness = 0): defined as step
  const float inv = 1.0f / (e1 - e0);            // wave-uniform edges: the division is scalar-rate work hoisted out of the pixel loop
  const float t = clamp01 ((x - e0) * inv);
  return t * t * (3.0f - 2.0f * t);
}
__device__ __forceinline__ float hash12 (float px, float py, uint32_t frame)
{
  const float fo = (float) frame * 0.00137f;
  float x = fractf_ (px * 0.1031f + fo), y = fractf_ (py * 0.1031f + fo), z = fractf_ (px * 0.1031f + fo);
  const float d = x * (y + 33.33f) + y * (z + 33.33f) + z * (x + 33.33f);
  x += d; y += d; z += d;
  return fractf_ ((x + y) * z);
}
// 1 / x for a positive normal x: exponent-flip seed + three Newton steps of two fma each — the same fixed sequence as
// oracle/metalref.c vf_rcp (relative error < 6e-8).  7 full-rate instructions where an IEEE division is ~10, two of them quarter-rate;
// MSL's fast-math division is a reciprocal approximation times the numerator anyway.
__device__ __forceinline__ float vf_rcp (float x)
{
  float r = __uint_as_float (0x7EF311C7u - __float_as_uint (x));
  r = fmaf (r, fmaf (-x, r, 1.0f), r);
  r = fmaf (r, fmaf (-x, r, 1.0f), r);
  r = fmaf (r, fmaf (-x, r, 1.0f), r);
  return r;
}
__device__ __forceinline__ void rgb_to_hsv (float r, float g, float b, float *h, float *s, float *v)
{
  const float Kx = 0.0f, Ky = -1.0f / 3.0f, Kz = 2.0f / 3.0f, Kw = -1.0f;
  const float t1 = stepf (b, g);
  const float px = mixf (b, g, t1), py = mixf (g, b, t1), pz = mixf (Kw, Kx, t1), pw = mixf (Kz, Ky, t1);
  const float t2 = stepf (px, r);
  const float qx = mixf (px, r, t2), qy = mixf (py, py, t2), qz = mixf (pw, pz, t2), qw = mixf (r, px, t2);
  const float d = qx - fminf (qw, qy);
  const float e = 1.0e-10f;
  *h = fabsf (qz + (qw - qy) * vf_rcp (6.0f * d + e));
  *s = d * vf_rcp (qx + e);
  *v = qx;
}
__device__ __forceinline__ void hsv_to_rgb (float h, float s, float v, float *r, float *g, float *b)
{
  const float pr = fabsf (fractf_ (h + 1.0f) * 6.0f - 3.0f);
  const float pg = fabsf (fractf_ (h + 2.0f / 3.0f) * 6.0f - 3.0f);
  const float pb = fabsf (fractf_ (h + 1.0f / 3.0f) * 6.0f - 3.0f);
  *r = v * mixf (1.0f, clamp01 (pr - 1.0f), s);
  *g = v * mixf (1.0f, clamp01 (pg - 1.0f), s);
  *b = v * mixf (1.0f, clamp01 (pb - 1.0f), s);
}


// pow(x, y) for x in [1e-4, 1], y > 0 (gamma stage): the same fixed sequence of IEEE single-precision operations as
// oracle/metalref.c vf_powf — atanh-series log2 on the reduced mantissa, degree-7 exp2, explicit fma steps — so both
// sides agree bit for bit (libm and OCML powf do not, and MSL's fast-math pow is not correctly rounded anyway:
// SURVEY.md Appendix B item 8).
__device__ __forceinline__ float vf_powf (float x, float y)
{
  const uint32_t ux = __float_as_uint (x);
  int e = (int) (ux >> 23) - 127;
  float m = __uint_as_float ((ux & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float t = (m - 1.0f) * vf_rcp (m + 1.0f), t2 = t * t;
  float p = 0.11111111f;
  p = fmaf (p, t2, 0.14285714f); p = fmaf (p, t2, 0.2f); p = fmaf (p, t2, 0.33333333f); p = fmaf (p, t2, 1.0f);
  const float l2 = fmaf (t * p, 2.88539008f, (float) e);
  const float z = y * l2;
  if (z < -126.0f) return 0.0f;
  const float zi = floorf (z + 0.5f), f = (z - zi) * 0.69314718f;
  float q = 1.98412698e-4f;
  q = fmaf (q, f, 1.38888889e-3f); q = fmaf (q, f, 8.33333333e-3f); q = fmaf (q, f, 4.16666667e-2f); q = fmaf (q, f, 0.16666667f);
  q = fmaf (q, f, 0.5f); q = fmaf (q, f, 1.0f); q = fmaf (q, f, 1.0f);
  return __uint_as_float (__float_as_uint (q) + ((uint32_t) (int) zi << 23));
}

// applyColorAdjustments (metalvideofilter_shaders.h:92-155), fixed order.  Every `if` tests a wave-uniform parameter:
// a disabled stage costs one scalar branch.
__device__ __forceinline__ F4 color_adjust (F4 c, const VfHipVideoFilterParams &u, float tu, float tv, int W, int H)
{
  float r = c.r, g = c.g, b = c.b, a = c.a;
  r += u.brightness; g += u.brightness; b += u.brightness;
  r = fmaf (r - 0.5f, u.contrast, 0.5f); g = fmaf (g - 0.5f, u.contrast, 0.5f); b = fmaf (b - 0.5f, u.contrast, 0.5f);
  const float lum = fmaf (b, 0.0722f, fmaf (g, 0.7152f, r * 0.2126f));
  r = mixf (lum, r, u.saturation); g = mixf (lum, g, u.saturation); b = mixf (lum, b, u.saturation);
  if (fabsf (u.hue) > 0.001f) {
    float h, s, v;
    rgb_to_hsv (clamp01 (r), clamp01 (g), clamp01 (b), &h, &s, &v);
    h = fractf_ (h + u.hue / (2.0f * 3.14159265358979323846f));
    hsv_to_rgb (h, s, v, &r, &g, &b);
  }
  const float ig = 1.0f / u.gamma;
  r = fminf (fmaxf (r, 0.0001f), 1.0f); g = fminf (fmaxf (g, 0.0001f), 1.0f); b = fminf (fmaxf (b, 0.0001f), 1.0f);
  if (ig != 1.0f) { r = vf_powf (r, ig); g = vf_powf (g, ig); b = vf_powf (b, ig); }      // pow (x, 1) == x exactly
  if (u.sepia > 0.001f) {
    const float sr = fmaf (b, 0.189f, fmaf (g, 0.769f, r * 0.393f));
    const float sg = fmaf (b, 0.168f, fmaf (g, 0.686f, r * 0.349f));
    const float sb = fmaf (b, 0.131f, fmaf (g, 0.534f, r * 0.272f));
    r = mixf (r, sr, u.sepia); g = mixf (g, sg, u.sepia); b = mixf (b, sb, u.sepia);
  }
  if (u.invert) { r = 1.0f - r; g = 1.0f - g; b = 1.0f - b; }
  if (u.chroma_key_enabled) {
    const float dr = r - u.chroma_key_r, dg = g - u.chroma_key_g, db = b - u.chroma_key_b;
    const float dist = sqrtf (dr * dr + dg * dg + db * db);
    a *= smoothstepf (u.chroma_key_tolerance, u.chroma_key_tolerance + u.chroma_key_smoothness, dist);
  }
  if (u.vignette > 0.001f) {
    const float cx = tu - 0.5f, cy = tv - 0.5f;
    const float dist = sqrtf (cx * cx + cy * cy) * 1.414f;
    const float vig = 1.0f - smoothstepf (0.5f, 1.0f, dist) * u.vignette;
    r *= vig; g *= vig; b *= vig;
  }
  if (u.noise > 0.001f) {
    float n = hash12 (tu * (float) W, tv * (float) H, u.frame_index);
    n = (n - 0.5f) * u.noise * 0.5f;
    r += n; g += n; b += n;
  }
  F4 o; o.r = clamp01 (r); o.g = clamp01 (g); o.b = clamp01 (b); o.a = a;
  return o;
}

// trilinear 3D LUT, coordinate c*(N-1)/N + .5/N (metalvideofilter_shaders.h:188-194).  The table is re-laid at upload
// (vf_upload_lut) CELL-MAJOR: entry (b, g, r) holds all eight corners of the cell whose low corner it is — per channel k
// two float4 { L(r,g,b), L(r+1,g,b), L(r,g+1,b), L(r+1,g+1,b) }.k for blue levels b and b+1, the +1 indices clamped like the
// sampler's second tap — 96 contiguous bytes.  With the plain [b][g][r] table the eight corners of a pixel sit on eight
// different cache lines, random colours pull ~1 KB of lines per pixel through L1 for 96 useful bytes, and that traffic —
// not arithmetic, not HBM — bounded the stage (42 of 49 us per 1080p frame, profiles/r02d_vf_ablation.jsonl); cell-major
// it is one or two lines.  Same values, same interpolation order as the plain table.
constexpr int VF_LUT_CELL = 6;          // float4 per cell
__device__ __forceinline__ void lut_sample (const float4 *cells, int N, F4 &c)
{
  const float scale = (float) (N - 1) / (float) N, offset = 0.5f / (float) N;
  const metal::Taps tx = metal::lin_taps (N, fmaf (c.r, scale, offset)), ty = metal::lin_taps (N, fmaf (c.g, scale, offset)),
                    tz = metal::lin_taps (N, fmaf (c.b, scale, offset));
  const uint32_t n = (uint32_t) N;
  const float4 *cell = cells + (((uint32_t) tz.i0 * n + (uint32_t) ty.i0) * n + (uint32_t) tx.i0) * (uint32_t) VF_LUT_CELL;
  using metal::lerp2;
  float o[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float4 lo = cell[2 * k], hi = cell[2 * k + 1];
    o[k] = lerp2 (lerp2 (lerp2 (lo.x, lo.y, tx.f), lerp2 (lo.z, lo.w, tx.f), ty.f), lerp2 (lerp2 (hi.x, hi.y, tx.f), lerp2 (hi.z, hi.w, tx.f), ty.f), tz.f);
  }
  c.r = o[0]; c.g = o[1]; c.b = o[2];
}


// ---- the fast colour path (round 3; default) -----------------------------------------------------------------------
// The reference compiles its MSL with default options (common/vfmetaldevice.m:87-93: options nil, i.e. fast-math on), so pow, /, sqrt,
// length and distance of applyColorAdjustments (videofilter/metalvideofilter_shaders.h:92-155) ARE hardware approximations there.  This path
// does the same with CDNA's: v_log_f32 / v_exp_f32 for the gamma pow, v_rcp_f32 for the divisions, v_sqrt_f32 for the distances (1 ulp each,
// 6 issue cycles per wave against ~75 for vf_powf), and folds what is affine in the uniforms on the host (vf_fold): brightness, contrast and
// saturation are ONE 3 x 3 map + offset, the sepia mix and the inversion another.  Same stages, same order, same clamps; every intermediate
// differs from the exact sequence by a few ulp (1e-6), which moves an 8-bit result only where it sits within that of a rounding
// boundary: tests compare this path with oracle/metalref.c at +-1 LSB and bound the fraction of bytes that differ.  color_adjust () above
// — the oracle's sequence operation for operation — stays as the exact path (VFHIP_VF_EXACT=1), which the tests hold to the oracle too.
// Issue-rate facts used (tools/ubench/valu_rate3.hip, profiles/r03a_valu_rate3.txt): add / mul / fma with or without the clamp modifier 2 cycles
// per wave64; min / max / med3 / cndmask / compare / fract / floor / every cvt 3.1; log / exp / rcp / sqrt 6.
__device__ __forceinline__ float fast_fract (float x) { return __builtin_amdgcn_fractf (x); }
// A VALU instruction with an SGPR operand issues in 3.9 cycles per wave64 on gfx950, the same instruction on VGPRs (or with an inline / literal constant) in
// 2.2 (tools/ubench/valu_occ.hip, profiles/r03j_valu_occ.txt: v_add / v_mul / v_fmac / v_fma alike).  The compiler keeps every wave-uniform value — the
// folded matrices, the blur weights — in SGPRs, so ~60 of a pixel's ~200 instructions and every one of the blur's 54 multiply-adds paid that.  in_vgpr ()
// hands the value over in a VGPR (one v_mov per lane, hoisted out of the pixel loops); `on`, which only steers branches, stays scalar.
__device__ __forceinline__ float in_vgpr (float x) { float v; asm ("v_mov_b32 %0, %1" : "=v"(v) : "s"(x)); return v; }

__device__ __forceinline__ F4 color_fast (F4 c, const VfFast &f, float tu, float tv, uint32_t frame)
{
  // brightness, contrast, saturation (affine), then the clamp every later stage starts with (folds into the last fma)
  float r = clamp01 (fmaf (f.a[0][0], c.r, fmaf (f.a[0][1], c.g, fmaf (f.a[0][2], c.b, f.ao[0]))));
  float g = clamp01 (fmaf (f.a[1][0], c.r, fmaf (f.a[1][1], c.g, fmaf (f.a[1][2], c.b, f.ao[1]))));
  float b = clamp01 (fmaf (f.a[2][0], c.r, fmaf (f.a[2][1], c.g, fmaf (f.a[2][2], c.b, f.ao[2]))));
  float a = c.a;
  if (f.on & VF_ON_HUE) {
    // rgbToHsv: the step / mix selections of the shader are max / min of the sorted channels plus two selected constants
    const float px = fmaxf (g, b), py = fminf (g, b);
    const bool gb = !(g < b), rp = !(r < px);
    const float pz = gb ? 0.0f : -1.0f, pw = gb ? -1.0f / 3.0f : 2.0f / 3.0f;
    const float qx = fmaxf (r, px), qw = fminf (r, px), qz = rp ? pz : pw;
    const float d = qx - fminf (qw, py);
    const float h0 = fabsf (fmaf (qw - py, __builtin_amdgcn_rcpf (fmaf (6.0f, d, 1.0e-10f)), qz));
    const float vs = d * __builtin_amdgcn_rcpf (qx + 1.0e-10f) * qx;             // v * s
    // hsvToRgb on the rotated hue: v * mix (1, clamp (p - 1), s) = (v - v s) + v s * clamp (p - 1)
    const float h = fast_fract (h0 + f.hue_shift);
    const float base = qx - vs;
    const float pr = fabsf (fmaf (h, 6.0f, -3.0f));                                // fract (h + 1) == h
    const float pg = fabsf (fmaf (fast_fract (h + 2.0f / 3.0f), 6.0f, -3.0f));
    const float pb = fabsf (fmaf (fast_fract (h + 1.0f / 3.0f), 6.0f, -3.0f));
    r = fmaf (vs, clamp01 (pr - 1.0f), base); g = fmaf (vs, clamp01 (pg - 1.0f), base); b = fmaf (vs, clamp01 (pb - 1.0f), base);
  }
  r = fmaxf (r, 0.0001f); g = fmaxf (g, 0.0001f); b = fmaxf (b, 0.0001f);         // (<= 1 already: clamped above / v, s <= 1)
  if (f.on & VF_ON_GAMMA) {
    r = __builtin_amdgcn_exp2f (f.inv_gamma * __builtin_amdgcn_logf (r));
    g = __builtin_amdgcn_exp2f (f.inv_gamma * __builtin_amdgcn_logf (g));
    b = __builtin_amdgcn_exp2f (f.inv_gamma * __builtin_amdgcn_logf (b));
  }
  if (f.on & VF_ON_B) {
    const float r1 = fmaf (f.b[0][0], r, fmaf (f.b[0][1], g, fmaf (f.b[0][2], b, f.bo[0])));
    const float g1 = fmaf (f.b[1][0], r, fmaf (f.b[1][1], g, fmaf (f.b[1][2], b, f.bo[1])));
    const float b1 = fmaf (f.b[2][0], r, fmaf (f.b[2][1], g, fmaf (f.b[2][2], b, f.bo[2])));
    r = r1; g = g1; b = b1;
  }
  if (f.on & VF_ON_KEY) {
    const float dr = r - f.key_r, dg = g - f.key_g, db = b - f.key_b;
    const float dist = __builtin_amdgcn_sqrtf (fmaf (db, db, fmaf (dg, dg, dr * dr)));
    if (f.on & VF_ON_KEY_STEP) a = dist < f.key_e0 ? 0.0f : a;
    else {
      const float t = clamp01 (fmaf (dist, f.key_scale, f.key_bias));
      a *= (t * t) * fmaf (-2.0f, t, 3.0f);
    }
  }
  if (f.on & VF_ON_VIG) {
    const float cx = tu - 0.5f, cy = tv - 0.5f;
    const float q = __builtin_amdgcn_sqrtf (fmaf (cx, cx, cy * cy));
    const float t = clamp01 (fmaf (q, 2.0f * 1.414f, -1.0f));                       // smoothstep (.5, 1, q * 1.414)
    const float vig = fmaf ((t * t) * fmaf (-2.0f, t, 3.0f), -f.vig, 1.0f);
    r *= vig; g *= vig; b *= vig;
  }
  if (f.on & VF_ON_NOISE) {
    // the hash amplifies ulp differences of its coordinates a hundredfold: its operations stay the oracle's, in its order (only fract is the instruction)
    const float fo = (float) frame * 0.00137f;
    const float hx = tu * f.fw, hy = tv * f.fh;
    float x = fast_fract (hx * 0.1031f + fo), y = fast_fract (hy * 0.1031f + fo);
    const float d = x * (y + 33.33f) + y * (x + 33.33f) + x * (x + 33.33f);     // (z == x: the shader's p3.z is p.x again)
    x += d; y += d;
    const float n = (fast_fract ((x + y) * x) - 0.5f) * f.noise_gain;
    r += n; g += n; b += n;
  }
  F4 o; o.r = clamp01 (r); o.g = clamp01 (g); o.b = clamp01 (b); o.a = a;
  return o;
}

__device__ __forceinline__ VfFast fast_in_vgprs (const VfFast &f)
{
  VfFast v = f;                                  // (a disabled stage's uniforms are never read: they stay where they are, no v_mov spent on them)
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) v.a[i][j] = in_vgpr (f.a[i][j]);
    v.ao[i] = in_vgpr (f.ao[i]);
  }
  if (f.on & VF_ON_HUE) v.hue_shift = in_vgpr (f.hue_shift);
  if (f.on & VF_ON_GAMMA) v.inv_gamma = in_vgpr (f.inv_gamma);
  if (f.on & VF_ON_B)
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) v.b[i][j] = in_vgpr (f.b[i][j]);
      v.bo[i] = in_vgpr (f.bo[i]);
    }
  if (f.on & VF_ON_KEY) {
    v.key_r = in_vgpr (f.key_r); v.key_g = in_vgpr (f.key_g); v.key_b = in_vgpr (f.key_b);
    v.key_e0 = in_vgpr (f.key_e0); v.key_scale = in_vgpr (f.key_scale); v.key_bias = in_vgpr (f.key_bias);
  }
  if (f.on & VF_ON_VIG) v.vig = in_vgpr (f.vig);
  if (f.on & VF_ON_NOISE) { v.noise_gain = in_vgpr (f.noise_gain); v.fw = in_vgpr (f.fw); v.fh = in_vgpr (f.fh); }
  if (f.on & VF_ON_LUT16) { v.lut_nm1 = in_vgpr (f.lut_nm1); v.lut_n = in_vgpr (f.lut_n); }
  return v;
}

template <bool FAST> __device__ __forceinline__ VfParams vf_frame_t (const VfParams &p)
{
  VfParams q = vf_frame (p);
  if (FAST) q.f = fast_in_vgprs (p.f);
  return q;
}

// The fast path's LUT: 64-byte cells (one L2 line per pixel; 3 x 16 bytes read), fp16, per channel the eight coefficients of the cell's
// trilinear polynomial  k0 + fx kx + fy (ky + fx kxy) + fz (kz + fx kxz + fy (kyz + fx kxyz))  of the table's RESIDUAL against the identity
// (entry - lattice coordinate), times 255: trilinear interpolation reproduces the identity exactly, so  out * 255 = 255 c + poly (f),
// and what fp16 rounds (2^-12 relative) is the residual — small for grading LUTs, zero for an identity table — not the value.  A gather of
// random cells is bound by L2 requests, not by arithmetic (tools/ubench/lut_gather.hip, profiles/r03a_lut_gather.txt: 96-byte fp32 cells
// 15.0 us per 1080p frame, 48-byte cells 11.3 — they straddle lines —, 64-byte aligned cells with three parts read 8.9); v_fma_mix_f32 feeds
// the halves to an fp32 fma without a conversion.  Returns r, g, b TIMES 255 (the quantiser's multiply is folded in).
typedef _Float16 h8 __attribute__ ((ext_vector_type (8)));
__device__ __forceinline__ void lut_sample16 (const uint4 *cells, float nm1, float fn, F4 &c)
{
  const float x = c.r * nm1, y = c.g * nm1, z = c.b * nm1;               // lattice coordinate: c * (N-1)/N + .5/N, times N, minus .5
  const float fx = fast_fract (x), fy = fast_fract (y), fz = fast_fract (z);
  const float idx = fmaf (fmaf (z - fz, fn, y - fy), fn, x - fx);       // exact in fp32: < 2^24 (c == 1: cell N - 1, whose +1 corners are itself)
  const uint4 *cell = cells + (uint32_t) idx * 4u;
  const float in[3] = { c.r, c.g, c.b };
  float o[3];
  // the three parts of the cell go out together, whatever the register pressure says: the scheduler otherwise serialises them under pressure
  // (load, wait, use, load ...: three L2 latencies per pixel instead of one — the sharpening kernel ran 1.6x slower that way)
  const uint4 part[3] = { cell[0], cell[1], cell[2] };
  __builtin_amdgcn_sched_barrier (0);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint4 raw = part[k];
    h8 q;
    __builtin_memcpy (&q, &raw, 16);
    const float A = fmaf (fx, (float) q[1], (float) q[0]), B = fmaf (fx, (float) q[3], (float) q[2]);
    const float C = fmaf (fx, (float) q[5], (float) q[4]), D = fmaf (fx, (float) q[7], (float) q[6]);
    o[k] = fmaf (in[k], 255.0f, fmaf (fz, fmaf (fy, D, C), fmaf (fy, B, A)));
  }
  c.r = o[0]; c.g = o[1]; c.b = o[2];
}

// pass 1 of the reference for one texel value: adjustments -> LUT -> 8-bit target.  FAST: color_fast + the fp16 table; otherwise the oracle's sequence.
template <bool FAST> __device__ __forceinline__ uint32_t vf_shade (const VfParams &p, F4 c, float tu, float tv)
{
  if (FAST) {
    c = color_fast (c, p.f, tu, tv, p.u.frame_index);
    if (p.lut16) {
      lut_sample16 (p.lut16, p.f.lut_nm1, p.f.lut_n, c);                 // r, g, b come back times 255
      uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32 (c.r, 0u, 0u);
      q = __builtin_amdgcn_cvt_pk_u8_f32 (c.g, 1u, q);
      q = __builtin_amdgcn_cvt_pk_u8_f32 (c.b, 2u, q);
      return __builtin_amdgcn_cvt_pk_u8_f32 (c.a * 255.0f, 3u, q);
    }
  } else c = color_adjust (c, p.u, tu, tv, p.out.w, p.out.h);
  if (p.lut) lut_sample (p.lut, p.lut_size, c);
  return metal::quant_rgba8 (c);
}
// ... for one pixel: sample (exact texel, linear chroma) first
template <bool FAST> __device__ __forceinline__ uint32_t vf_pass1 (const VfParams &p, int x, int y)
{
  x = metal::iclamp (x, 0, p.out.w - 1); y = metal::iclamp (y, 0, p.out.h - 1);
  const float tu = ((float) x + 0.5f) * (1.0f / (float) p.out.w), tv = ((float) y + 0.5f) * (1.0f / (float) p.out.h);      // wave-uniform reciprocals (oracle: inv_w, inv_h)
  return vf_shade<FAST> (p, metal::fetch_1to1 (p.in, x, y, true), tu, tv);
}
// an RGBA / BGRA texel as the sampler returns it
__device__ __forceinline__ F4 vf_texel_rgb (uint32_t t, bool rgba)
{
  F4 c;
  c.g = metal::un8 ((t >> 8) & 0xff); c.a = metal::un8 (t >> 24);
  if (rgba) { c.r = metal::un8 (t & 0xff); c.b = metal::un8 ((t >> 16) & 0xff); }
  else { c.b = metal::un8 (t & 0xff); c.r = metal::un8 ((t >> 16) & 0xff); }
  return c;
}

// ... for an RGBA / BGRA texel in its 32 bits.  FAST: the launch has folded the texel's 1 / 255 into the first affine map (vf_launch_kernels: bytes
// kernels), so the channels enter as byte values; without a chroma key the alpha byte passes through untouched (rint (a / 255 * 255) == a).
template <bool FAST> __device__ __forceinline__ uint32_t vf_shade_texel (const VfParams &p, uint32_t t, bool rgba, float tu, float tv)
{
  if (!FAST) return vf_shade<false> (p, vf_texel_rgb (t, rgba), tu, tv);
  F4 c;
  c.g = (float) ((t >> 8) & 0xff);
  if (rgba) { c.r = (float) (t & 0xff); c.b = (float) ((t >> 16) & 0xff); }
  else { c.b = (float) (t & 0xff); c.r = (float) ((t >> 16) & 0xff); }
  const bool key = p.f.on & VF_ON_KEY;
  c.a = key ? metal::un8 (t >> 24) : 0.0f;
  const uint32_t q = vf_shade<true> (p, c, tu, tv);
  return key ? q : (q & 0x00ffffffu) | (t & 0xff000000u);
}

template <bool FAST> __global__ __launch_bounds__ (256) void k_vf_point (const VfParams pp)
{
  const VfParams p = vf_frame (pp);          // (uniforms stay in SGPRs here: with the 4:2:0 sampler's registers the VGPR copies cost more occupancy than the operands cost issue cycles)
  const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
  if (2 * bx >= p.out.w || 2 * by >= p.out.h) return;
  uint32_t q[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++) q[dy][dx] = vf_pass1<FAST> (p, 2 * bx + dx, 2 * by + dy);
  metal::store_block (p.out, bx, by, q);
}

// ------------------------------------------------------------------------------------------------------------------
// k_vf_sharp: sharpness != 0.  One workgroup (512 lanes) per 128 x 56 output tile:
//   0. RGBA / BGRA inputs: the raw texels of the tile and its 4-pixel halo (136 x 64, clamped to the image like the blur's reads) go to LDS
//      first, the seventeen loads of a lane all in flight — pass 1 then reads LDS, and no lane ever waits for HBM between two pixels (round 2
//      loaded a texel, shaded it, loaded the next: with sharpening only, the kernel spent half its time in those seventeen latencies);
//   1. pass 1 (colour adjustments + LUT — by far the most expensive part) for the region, 1.21x redundancy (the first version's 64x16
//      tile: 1.69x, 128x32: 1.33x), quantised to 8 bits into LDS exactly where the reference wrote its render target;
//   2. horizontal 9-tap Gaussian, LDS -> LDS: a lane owns one ROW of the region (64 rows = one wave) and a run of 16
//      columns; it slides along the row, so every texel is unpacked once per run (24 unpacks for 16 outputs instead of
//      144) and a wave reads one column at a time — conflict-free with the odd row stride;
//   3. vertical 9-tap + unsharp mask: a lane owns one COLUMN and a run of 14 rows, sliding down (22 unpacks for 14
//      outputs); RGB outputs go straight to HBM from here (coalesced dwords), YUV outputs pass through LDS once more
//      for the 2x2-block store epilogue.
// The alpha byte is not blurred: the unsharp mask keeps the pass-1 alpha (metalvideofilter_shaders.h:318-327), so the
// blurred alpha of the reference's two temporaries is never read.
// LDS: 64 x 137 + 64 x 129 dwords = 68 KB -> two workgroups (16 waves) per CU.
constexpr int VF_HALO = 4;
constexpr int VF_TW = 128, VF_TH = 56, VF_THREADS = 512;
constexpr int VF_RW = VF_TW + 2 * VF_HALO, VF_RH = VF_TH + 2 * VF_HALO;      // 136 x 64
constexpr int VF_RS = VF_RW + 1, VF_HS = VF_TW + 1;                          // odd LDS row strides (dwords)
constexpr int VF_HRUN = 16, VF_VRUN = 14;
constexpr int VF_PER = VF_RW * VF_RH / VF_THREADS;                           // region pixels per lane: 17
static_assert (VF_RH == 64 && VF_THREADS == VF_RH * (VF_TW / VF_HRUN) && VF_THREADS == VF_TW * (VF_TH / VF_VRUN) && VF_PER * VF_THREADS == VF_RW * VF_RH &&
               VF_THREADS == 4 * 128 && VF_RW == 128 + 8 && VF_THREADS == 8 * VF_RH && VF_PER == 17, "tile / lane mapping");
__constant__ float kBlurW[9] = { 0.028532f, 0.067234f, 0.124009f, 0.179044f, 0.20236f, 0.179044f, 0.124009f, 0.067234f, 0.028532f };

struct F3 { float r, g, b; };
// FAST: the blur and the unsharp mask work on the byte VALUES (0 .. 255 as floats) instead of value / 255: the same weighted sums up to the last ulp, and
// every unpack loses its multiply by 1 / 255, every quantisation its multiply by 255 (v_cvt_pk_u8_f32 rounds to nearest even and saturates either way)
template <bool FAST> __device__ __forceinline__ F3 unpack_rgb (uint32_t q)
{
  F3 o;
  if (FAST) { o.r = (float) (q & 0xff); o.g = (float) ((q >> 8) & 0xff); o.b = (float) ((q >> 16) & 0xff); }
  else { o.r = metal::un8 (q & 0xff); o.g = metal::un8 ((q >> 8) & 0xff); o.b = metal::un8 ((q >> 16) & 0xff); }
  return o;
}
template <bool FAST> __device__ __forceinline__ uint32_t quant_rgb (float r, float g, float b)
{
  const float k = FAST ? 1.0f : 255.0f;
  uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32 (FAST ? r : r * k, 0u, 0u);
  q = __builtin_amdgcn_cvt_pk_u8_f32 (FAST ? g : g * k, 1u, q);
  return __builtin_amdgcn_cvt_pk_u8_f32 (FAST ? b : b * k, 2u, q);
}
__device__ __forceinline__ F3 unpack_rgb8 (uint32_t q) { F3 o; o.r = metal::un8 (q & 0xff); o.g = metal::un8 ((q >> 8) & 0xff); o.b = metal::un8 ((q >> 16) & 0xff); return o; }
__device__ __forceinline__ uint32_t quant_rgb8 (float r, float g, float b)
{
  uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32 (r * 255.0f, 0u, 0u);
  q = __builtin_amdgcn_cvt_pk_u8_f32 (g * 255.0f, 1u, q);
  return __builtin_amdgcn_cvt_pk_u8_f32 (b * 255.0f, 2u, q);
}

// k_vf_point_rgba4: sharpness == 0, RGBA / BGRA in and out, 16-byte aligned rows (the filter on a decoded-to-RGB or rendered stream: by far the
// most common way the element is used).  No 4:2:0 output means no 2 x 2 blocks: a lane takes four adjacent pixels of one row as ONE 16-byte
// load and one 16-byte non-temporal store (k_vf_point moves 8 bytes per access); per pixel exactly vf_pass1's operations.
template <bool FAST> __global__ __launch_bounds__ (256) void k_vf_point_rgba4 (const VfParams pp)
{
  const VfParams p = vf_frame_t<FAST> (pp);
  const int x4 = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (4 * x4 >= p.out.w || y >= p.out.h) return;                     // out.w is a multiple of 4
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  const v4u t = *reinterpret_cast<const v4u *> (p.in.p[0] + (size_t) y * p.in.s[0] + 16 * (size_t) x4);
  const bool rgba_in = p.in.fmt == VFHIP_FORMAT_RGBA, bgra_out = p.out.fmt == VFHIP_FORMAT_BGRA;
  const float tv = ((float) y + 0.5f) * (1.0f / (float) p.out.h);
  v4u o;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int x = 4 * x4 + i;
    const float tu = ((float) x + 0.5f) * (1.0f / (float) p.out.w);
    const uint32_t q = vf_shade_texel<FAST> (p, t[i], rgba_in, tu, tv);               // metal::fetch_1to1's RGBA / BGRA texel
    o[i] = bgra_out ? __builtin_amdgcn_perm (0u, q, 0x03000102u) : q;
  }
  __builtin_nontemporal_store (o, reinterpret_cast<v4u *> (p.out.p[0] + (size_t) y * p.out.s[0]) + x4);
}

// k_vf_point_quad: sharpness == 0 with a 4:2:0 frame on either side (a decoder's frames through the filter, to an encoder's 4:2:0 or to RGB; RGB
// into 4:2:0): a lane takes 4 x 2 pixels.  k_vf_point's 2 x 2 block cost 20 one- and two-byte loads on a 4:2:0 input (the linear chroma sampler of
// the reference: four taps per pixel) and two-byte stores; here the block's chroma neighbourhood is three window loads, the luma two dwords
// (metal::fetch_quad), and the output goes out as dwords / 16-byte rows (metal::store_quad).  Per pixel exactly vf_pass1's operations on exactly
// its inputs: NV12 -> NV12 1080p 7.5 -> 5.05 us, NV12 -> BGRA 6.9 -> 4.3.
template <bool FAST> __global__ __launch_bounds__ (256) void k_vf_point_quad (const VfParams pp)
{
  const VfParams p = vf_frame (pp);          // (uniforms stay in SGPRs here: with the 4:2:0 sampler's registers the VGPR copies cost more occupancy than the operands cost issue cycles)
  const int xq = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
  if (4 * xq >= p.out.w || 2 * by >= p.out.h) return;                 // W % 4 == 0, even H
  F4 c[2][4];
  metal::fetch_quad (p.in, xq, by, c);
  uint32_t q[2][4];
  const float inv_w = 1.0f / (float) p.out.w, inv_h = 1.0f / (float) p.out.h;
#pragma unroll
  for (int dy = 0; dy < 2; dy++) {
    const float tv = ((float) (2 * by + dy) + 0.5f) * inv_h;
#pragma unroll
    for (int dx = 0; dx < 4; dx++) q[dy][dx] = vf_shade<FAST> (p, c[dy][dx], ((float) (4 * xq + dx) + 0.5f) * inv_w, tv);
  }
  metal::store_quad (p.out, xq, by, q);
}

// One workgroup per tile, tiles numbered x fastest, then y, then frame.  (Round 3 also tried PERSISTENT workgroups — two per CU walking the tiles,
// each loading its next tile's raw texels into registers during the vertical pass — and the kernel without its barriers: both within 1 % of this
// version, profiles/r03j_sharp_experiments.txt; the time is not in load latency or barrier skew.)
struct VfTile { int x0, y0; uint32_t frame; };
__device__ __forceinline__ VfTile vf_tile (int t, int tiles_x, int tiles_y)
{
  const int per = tiles_x * tiles_y, f = t / per, r = t - f * per, ty = r / tiles_x;
  VfTile o; o.frame = (uint32_t) f; o.y0 = ty * VF_TH; o.x0 = (r - ty * tiles_x) * VF_TW;
  return o;
}
template <bool FAST, bool STAGED> __global__ __launch_bounds__ (VF_THREADS, 4) void k_vf_sharp (const VfParams pp, int tiles_x, int tiles_y, int n_tiles, int n_chunk)
{
  __shared__ uint32_t rt[VF_RH * VF_RS];       // pass-1 render target, tile + halo (clamped to the image like the blur's reads)
  __shared__ uint32_t hb[VF_RH * VF_HS];       // horizontal blur (8-bit, like _blurTemp); reused for the result of YUV outputs
  const int tid = threadIdx.x;
  const int w = pp.out.w, h = pp.out.h;
  const float inv_w = 1.0f / (float) w, inv_h = 1.0f / (float) h;
  // STAGED: an RGBA / BGRA input (the launch decides; the two kinds of input are two kernels so that neither carries the other's registers)
  // staged inputs, lane mapping: the region's first 128 columns as a walk down ONE column (rows tid / 128, + 4, ...: 16 texels; a wave = 64 consecutive
  // texels of a row), its last 8 columns one texel per lane (64 rows x 8).  So 16 of a lane's 17 pixels share their x: the texture coordinate, the
  // vignette's cx^2 and the LDS column are loop invariants.
  const int cx = tid & 127, cy = tid >> 7, ex = 128 + (tid & 7), ey = tid >> 3;
  // XCD-aware tile order: workgroups are dealt round-robin over the chip's 8 XCDs (blocks b and b + 8 share one, each XCD has its own L2), so block b
  // takes tile (b % 8) * chunk + b / 8: every XCD works through ONE contiguous run of tiles — whole frames of a batch — and the halo rows and columns
  // neighbouring tiles share are fetched into one L2 instead of up to four (PMC, 64-frame launches: input fetched 1.64x -> see DESIGN §5.3).
  // The grid is rounded up to 8 * chunk blocks; the surplus ones have no tile.  Placement is a speed matter only: any mapping gives the same bytes.
  const int bt = (int) (blockIdx.x & 7u) * n_chunk + (int) (blockIdx.x >> 3);
  if (bt >= n_tiles) return;
  const VfTile T = vf_tile (bt, tiles_x, tiles_y);
  const int x0 = T.x0, y0 = T.y0;
  VfParams p = pp;
  p.in = metal::img_at (pp.in, T.frame * pp.in_pitch);
  p.out = metal::out_at (pp.out, T.frame * pp.out_pitch);
  p.u.frame_index += T.frame;
  if (FAST) p.f = fast_in_vgprs (pp.f);
  float bw[9];                                  // the blur weights in VGPRs (in_vgpr: SGPR operands halve the multiply-adds' issue rate); symmetric
#pragma unroll
  for (int k = 0; k < 5; k++) bw[k] = bw[8 - k] = in_vgpr (kBlurW[k]);
  if (!STAGED && p.quad_in) {
    // 4:2:0 input: the region (its corner sits on the 4 x 2 grid) in quads — the chroma neighbourhood of eight pixels as three window loads instead
    // of 32 taps; quads off the frame's edge (the halo of an edge tile) are clamped duplicates and go pixel by pixel
    for (int i = tid; i < (VF_RW / 4) * (VF_RH / 2); i += VF_THREADS) {
      const int qx = i % (VF_RW / 4), qy = i / (VF_RW / 4);
      const int gx = x0 - VF_HALO + 4 * qx, gy = y0 - VF_HALO + 2 * qy;
      uint32_t *d = rt + (2 * qy) * VF_RS + 4 * qx;
      if (gx >= 0 && gx + 3 < w && gy >= 0 && gy + 1 < h) {
        F4 c[2][4];
        metal::fetch420_quad (p.in, gx >> 2, gy >> 1, c);
#pragma unroll
        for (int dy = 0; dy < 2; dy++) {
          const float tv = ((float) (gy + dy) + 0.5f) * inv_h;
#pragma unroll
          for (int dx = 0; dx < 4; dx++) d[dy * VF_RS + dx] = vf_shade<FAST> (p, c[dy][dx], ((float) (gx + dx) + 0.5f) * inv_w, tv);
        }
      } else {
#pragma unroll 1
        for (int k = 0; k < 8; k++) d[(k >> 2) * VF_RS + (k & 3)] = vf_pass1<FAST> (p, gx + (k & 3), gy + (k >> 2));
      }
    }
  } else if (STAGED) {
    // step 0: the tile's raw texels -> LDS, every load of the lane issued before the first is used
    {
      uint32_t raw[VF_PER];
      const int gx = metal::iclamp (x0 - VF_HALO + cx, 0, w - 1);
#pragma unroll
      for (int k = 0; k < VF_PER - 1; k++)
        raw[k] = *reinterpret_cast<const uint32_t *> (p.in.p[0] + (size_t) metal::iclamp (y0 - VF_HALO + cy + 4 * k, 0, h - 1) * p.in.s[0] + 4 * (size_t) gx);
      raw[VF_PER - 1] = *reinterpret_cast<const uint32_t *> (p.in.p[0] + (size_t) metal::iclamp (y0 - VF_HALO + ey, 0, h - 1) * p.in.s[0] + 4 * (size_t) metal::iclamp (x0 - VF_HALO + ex, 0, w - 1));
#pragma unroll
      for (int k = 0; k < VF_PER - 1; k++) rt[(cy + 4 * k) * VF_RS + cx] = raw[k];
      rt[ey * VF_RS + ex] = raw[VF_PER - 1];
    }
    // step 1: pass 1 in place, each lane on the texels it staged itself (no barrier needed in between)
    const bool rgba_in = p.in.fmt == VFHIP_FORMAT_RGBA;
    const float wm1 = in_vgpr ((float) (w - 1)), hm1 = in_vgpr ((float) (h - 1)), inv_wv = in_vgpr (inv_w), inv_hv = in_vgpr (inv_h);
    uint32_t *d = rt + cy * VF_RS + cx;
    float fx = (float) (x0 - VF_HALO + cx), fy = (float) (y0 - VF_HALO + cy);
#pragma unroll 1
    for (int k = 0; k < VF_PER; k++) {
      if (k == VF_PER - 1) { d = rt + ey * VF_RS + ex; fx = (float) (x0 - VF_HALO + ex); fy = (float) (y0 - VF_HALO + ey); }
      // (the oracle's texture coordinate, operation for operation: the noise hash amplifies an ulp of it a thousandfold)
      const float tu = (__builtin_amdgcn_fmed3f (fx, 0.0f, wm1) + 0.5f) * inv_wv, tv = (__builtin_amdgcn_fmed3f (fy, 0.0f, hm1) + 0.5f) * inv_hv;
      *d = vf_shade_texel<FAST> (p, *d, rgba_in, tu, tv);
      d += 4 * VF_RS; fy += 4.0f;
    }
  } else
  for (int i = tid; i < VF_RW * VF_RH; i += VF_THREADS) {
    const int rx = i % VF_RW, ry = i / VF_RW;
    rt[ry * VF_RS + rx] = vf_pass1<FAST> (p, x0 - VF_HALO + rx, y0 - VF_HALO + ry);
  }
  __syncthreads ();
  // horizontal pass.  Reads clamp to the IMAGE (not the region): region column rx holds image column clamp(x0-4+rx)
  // (vf_pass1 clamps), so an unclamped region read is the clamped image read.
  {
    const int row = tid & (VF_RH - 1), c0 = (tid >> 6) * VF_HRUN;
    const uint32_t *src = rt + row * VF_RS + c0;
    F3 px[VF_HRUN + 8];
#pragma unroll
    for (int k = 0; k < VF_HRUN + 8; k++) px[k] = unpack_rgb<FAST> (src[k]);
#pragma unroll
    for (int j = 0; j < VF_HRUN; j++) {
      float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; k++) { sr = fmaf (px[j + k].r, bw[k], sr); sg = fmaf (px[j + k].g, bw[k], sg); sb = fmaf (px[j + k].b, bw[k], sb); }
      hb[row * VF_HS + c0 + j] = quant_rgb<FAST> (sr, sg, sb);
    }
  }
  __syncthreads ();
  const float amount = p.u.sharpness;
  const int col = tid & (VF_TW - 1), r0 = (tid >> 7) * VF_VRUN;
  // the vertical pass + unsharp mask of this lane's column run; emit (j, rgba8) takes each result as it is finished
  auto vertical = [&] (auto &&emit) {
    const uint32_t *src = hb + r0 * VF_HS + col;
    F3 px[VF_VRUN + 8];
#pragma unroll
    for (int k = 0; k < VF_VRUN + 8; k++) px[k] = unpack_rgb<FAST> (src[k * VF_HS]);
#pragma unroll
    for (int j = 0; j < VF_VRUN; j++) {
      float sr = 0.0f, sg = 0.0f, sb = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; k++) { sr = fmaf (px[j + k].r, bw[k], sr); sg = fmaf (px[j + k].g, bw[k], sg); sb = fmaf (px[j + k].b, bw[k], sb); }
      // _blurResult is 8-bit as well.  The unsharp mask itself stays in the oracle's normalised arithmetic in both paths: with both operands 8-bit
      // values and amount = .5 every other result is an exact tie in byte units, and the tie must fall the way value / 255 arithmetic makes it fall
      // (the byte-domain version of this step differed from the oracle in 10 % of the bytes, all of them such ties)
      F3 b;
      if (FAST) {                                                       // (the weights sum to < 1: rint of the byte-domain sum needs no clamp)
        b.r = __builtin_rintf (sr) * (1.0f / 255.0f); b.g = __builtin_rintf (sg) * (1.0f / 255.0f); b.b = __builtin_rintf (sb) * (1.0f / 255.0f);
      } else b = unpack_rgb8 (quant_rgb8 (sr, sg, sb));
      const uint32_t oq = rt[(r0 + j + VF_HALO) * VF_RS + col + VF_HALO];
      const F3 o = unpack_rgb8 (oq);
      float rr, rg, rb;
      if (amount > 0.0f) {
        rr = clamp01 (fmaf (o.r - b.r, amount, o.r)); rg = clamp01 (fmaf (o.g - b.g, amount, o.g)); rb = clamp01 (fmaf (o.b - b.b, amount, o.b));
      } else {
        const float t = fabsf (amount);
        rr = mixf (o.r, b.r, t); rg = mixf (o.g, b.g, t); rb = mixf (o.b, b.b, t);
      }
      emit (j, quant_rgb8 (rr, rg, rb) | (oq & 0xff000000u));           // alpha: the pass-1 value, untouched
    }
  };
  if (p.out.fmt == VFHIP_FORMAT_BGRA || p.out.fmt == VFHIP_FORMAT_RGBA) {
    // RGB outputs: a wave's 64 lanes are 64 consecutive pixels of one row -> 256-byte coalesced dword stores, each as soon as its value exists
    const int gx = x0 + col;
    const bool bgra = p.out.fmt == VFHIP_FORMAT_BGRA;
    uint8_t *o0 = p.out.p[0] + (size_t) (y0 + r0) * p.out.s[0] + 4 * (size_t) gx;
    vertical ([&] (int j, uint32_t v) {
      if (gx < w && y0 + r0 + j < h)
        __builtin_nontemporal_store (bgra ? __builtin_amdgcn_perm (0u, v, 0x03000102u) : v, reinterpret_cast<uint32_t *> (o0 + (size_t) j * p.out.s[0]));
    });
  } else {
    // the unsharp result (8-bit) goes to the tile's own rows of rt: each lane overwrites exactly the pass-1 texels it has just read (oq), nobody else's
    vertical ([&] (int j, uint32_t v) { rt[(r0 + j + VF_HALO) * VF_RS + col + VF_HALO] = v; });
    __syncthreads ();
    // store epilogue: 2x2 blocks of the tile
    for (int i = tid; i < (VF_TW / 2) * (VF_TH / 2); i += VF_THREADS) {
      const int lbx = i % (VF_TW / 2), lby = i / (VF_TW / 2);
      const int gx = x0 + 2 * lbx, gy = y0 + 2 * lby;
      if (gx >= w || gy >= h) continue;
      uint32_t q[2][2];
#pragma unroll
      for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
          const int lx = min (gx + dx, w - 1) - x0, ly = min (gy + dy, h - 1) - y0;      // edge-clamped duplicates
          q[dy][dx] = rt[(ly + VF_HALO) * VF_RS + lx + VF_HALO];
        }
      metal::store_block (p.out, gx / 2, gy / 2, q);
    }
  }
}

}  // namespace vfhip

struct VfHipVideoFilter {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo in {}, out {};
  float4 *d_lut = nullptr;          // fp32 cells (exact path; the fast path's fallback for tables fp16 cannot hold)
  uint4 *d_lut16 = nullptr;         // fp16 residual cells (fast path)
  int lut_size = 0;
  Flights fl;                       // pipelined host path (submit / wait)
};

// the fast path's folded uniforms (VfFast), in double, rounded once
static VfFast vf_fold (const VfHipVideoFilterParams &u, int w, int hh, int lut_size)
{
  VfFast f {};
  const double k = u.contrast, b0 = u.brightness, s = u.saturation, t = k * b0 - 0.5 * k + 0.5;
  const double wl[3] = { 0.2126, 0.7152, 0.0722 }, sw = wl[0] + wl[1] + wl[2];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) f.a[i][j] = (float) ((1.0 - s) * wl[j] * k + (i == j ? s * k : 0.0));
    f.ao[i] = (float) (s * t + (1.0 - s) * t * sw);
  }
  if (fabsf (u.hue) > 0.001f) { f.on |= VF_ON_HUE; f.hue_shift = (float) ((double) u.hue / (2.0 * 3.14159265358979323846)); }
  f.inv_gamma = 1.0f / u.gamma;
  if (f.inv_gamma != 1.0f) f.on |= VF_ON_GAMMA;
  const double sep[3][3] = { { 0.393, 0.769, 0.189 }, { 0.349, 0.686, 0.168 }, { 0.272, 0.534, 0.131 } };
  const double ps = u.sepia > 0.001f ? (double) u.sepia : 0.0, sign = u.invert ? -1.0 : 1.0;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) f.b[i][j] = (float) (sign * ((i == j ? 1.0 - ps : 0.0) + ps * sep[i][j]));
    f.bo[i] = u.invert ? 1.0f : 0.0f;
  }
  if (ps > 0.0 || u.invert) f.on |= VF_ON_B;
  if (u.chroma_key_enabled) {
    f.on |= VF_ON_KEY;
    f.key_r = u.chroma_key_r; f.key_g = u.chroma_key_g; f.key_b = u.chroma_key_b;
    const float e0 = u.chroma_key_tolerance, e1 = u.chroma_key_tolerance + u.chroma_key_smoothness;
    f.key_e0 = e0;
    if (!(e0 < e1)) f.on |= VF_ON_KEY_STEP;
    else { const double inv = 1.0 / ((double) e1 - (double) e0); f.key_scale = (float) inv; f.key_bias = (float) (-(double) e0 * inv); }
  }
  if (u.vignette > 0.001f) { f.on |= VF_ON_VIG; f.vig = u.vignette; }
  if (u.noise > 0.001f) { f.on |= VF_ON_NOISE; f.noise_gain = u.noise * 0.5f; }
  f.fw = (float) w; f.fh = (float) hh;
  f.lut_nm1 = (float) (lut_size - 1); f.lut_n = (float) lut_size;
  return f;
}

template <bool FAST> static int vf_launch_kernels (VfParams p, const VfHipFrame *in, const VfHipFrame *out, int w, int hh, int n_frames, hipStream_t s)
{
  const VfHipVideoFilterParams *prm = &p.u;
  const bool rgb_in = in->info.format == VFHIP_FORMAT_RGBA || in->info.format == VFHIP_FORMAT_BGRA;
  // the kernels that shade RGBA / BGRA texels straight from their bytes (vf_shade_texel) get the texel's 1 / 255 folded into the first affine map
  auto bytes_in = [&p] () { for (auto &row : p.f.a) for (float &v : row) v = (float) ((double) v / 255.0); };
  if (prm->sharpness < -0.001f || prm->sharpness > 0.001f) {
    if (FAST && rgb_in && !p.quad_in) bytes_in ();
    const int tiles_x = (w + VF_TW - 1) / VF_TW, tiles_y = (hh + VF_TH - 1) / VF_TH;
    const long long n_tiles = (long long) tiles_x * tiles_y * n_frames;
    if (n_tiles > 0x7fffff00ll) return set_error (VFHIP_ERR_INVALID, "videofilter: %lld tiles in one batch (at most 2^31): split the batch", n_tiles);
    const int n_chunk = (int) ((n_tiles + 7) / 8);                               // tiles per XCD (k_vf_sharp's tile order)
    const dim3 grid ((unsigned) (8 * n_chunk));
    if (rgb_in && !p.quad_in) hipLaunchKernelGGL ((k_vf_sharp<FAST, true>), grid, dim3 (VF_THREADS), 0, s, p, tiles_x, tiles_y, (int) n_tiles, n_chunk);
    else hipLaunchKernelGGL ((k_vf_sharp<FAST, false>), grid, dim3 (VF_THREADS), 0, s, p, tiles_x, tiles_y, (int) n_tiles, n_chunk);
  } else {
    const bool rgb_io = rgb_in && (out->info.format == VFHIP_FORMAT_RGBA || out->info.format == VFHIP_FORMAT_BGRA);
    const uintptr_t al = (uintptr_t) in->data[0] | (uintptr_t) in->stride[0] | (uintptr_t) p.in_pitch | (uintptr_t) out->data[0] | (uintptr_t) out->stride[0] | (uintptr_t) p.out_pitch;
    if (rgb_io && !(w & 3) && !(al & 15) && getenv ("VFHIP_VF_BLOCKS") == nullptr) {
      dim3 grid ((unsigned) ((w / 4 + 63) / 64), (unsigned) ((hh + 3) / 4), (unsigned) n_frames);
      if (FAST) bytes_in ();
      hipLaunchKernelGGL (k_vf_point_rgba4<FAST>, grid, dim3 (64, 4), 0, s, p);
    } else if (getenv ("VFHIP_VF_BLOCKS") == nullptr && metal::quad_frame_ok (in, p.in_pitch, false) && metal::quad_frame_ok (out, p.out_pitch, true)) {
      dim3 grid ((unsigned) ((w / 4 + 63) / 64), (unsigned) ((hh / 2 + 3) / 4), (unsigned) n_frames);
      hipLaunchKernelGGL (k_vf_point_quad<FAST>, grid, dim3 (64, 4), 0, s, p);
    } else {
      const int bw = (w + 1) / 2, bh = (hh + 1) / 2;
      dim3 grid ((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4), (unsigned) n_frames);
      hipLaunchKernelGGL (k_vf_point<FAST>, grid, dim3 (64, 4), 0, s, p);
    }
  }
  return VFHIP_OK;
}

static int vf_launch (VfHipVideoFilter *h, const VfHipFrame *in, VfHipFrame *out, const VfHipVideoFilterParams *prm, hipStream_t s,
    int n_frames = 1, size_t in_pitch = 0, size_t out_pitch = 0)
{
  VfParams p {};
  p.in_pitch = in_pitch; p.out_pitch = out_pitch;
  p.in = metal::make_img (in); p.out = metal::make_out (out);
  p.u = *prm; p.lut = h->d_lut; p.lut_size = h->lut_size;
  p.quad_in = (in->info.format == VFHIP_FORMAT_NV12 || in->info.format == VFHIP_FORMAT_I420) && metal::quad_frame_ok (in, in_pitch, false) && getenv ("VFHIP_VF_BLOCKS") == nullptr;
  const int w = h->out.width, hh = h->out.height;
  // the fast path (hardware transcendentals, folded uniforms, fp16 table) unless VFHIP_VF_EXACT asks for the oracle's operation sequence;
  // VFHIP_VF_LUT32 keeps the fast path on the fp32 table (A/B of the table alone)
  if (getenv ("VFHIP_VF_EXACT") == nullptr) {
    p.f = vf_fold (*prm, w, hh, h->lut_size);
    p.lut16 = getenv ("VFHIP_VF_LUT32") == nullptr ? h->d_lut16 : nullptr;
    if (p.lut16) p.f.on |= VF_ON_LUT16;
    if (int rc = vf_launch_kernels<true> (p, in, out, w, hh, n_frames, s)) return rc;
  } else if (int rc = vf_launch_kernels<false> (p, in, out, w, hh, n_frames, s)) return rc;
  VFHIP_CHECK_HIP (hipGetLastError ());
  return VFHIP_OK;
}

static int vf_check (VfHipVideoFilter *h, const VfHipFrame *in, const VfHipFrame *out, const VfHipVideoFilterParams *prm)
{
  if (!h || !prm) return set_error (VFHIP_ERR_INVALID, "null argument");
  if (!h->configured) return set_error (VFHIP_ERR_NOT_CONFIGURED, "videofilter: process before configure");
  if (!(prm->gamma > 0.0f)) return set_error (VFHIP_ERR_INVALID, "gamma must be > 0");
  int rc = check_frame (in, &h->in, "input");
  if (rc) return rc;
  return check_frame (out, &h->out, "output");
}

static int vf_upload_lut (VfHipVideoFilter *h, const float *rgba, int size)
{
  if (size < 2 || size > 64) return set_error (VFHIP_ERR_INVALID, "LUT size %d outside 2..64", size);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  // cell-major layout (lut_sample): N^3 cells x 3 channels x 2 blue levels x { (r,g), (r+1,g), (r,g+1), (r+1,g+1) }, +1 clamped
  const size_t n = (size_t) size;
  std::vector<float> faces (n * n * n * VF_LUT_CELL * 4);
  for (size_t b = 0; b < n; b++)
    for (size_t g = 0; g < n; g++)
      for (size_t r = 0; r < n; r++) {
        const size_t r1 = r + 1 < n ? r + 1 : r, g1 = g + 1 < n ? g + 1 : g, b1 = b + 1 < n ? b + 1 : b;
        float *d = &faces[((b * n + g) * n + r) * VF_LUT_CELL * 4];
        for (size_t k = 0; k < 3; k++)
          for (size_t z = 0; z < 2; z++) {
            const size_t bb = z ? b1 : b;
            float *e = d + (2 * k + z) * 4;
            e[0] = rgba[((bb * n + g) * n + r) * 4 + k]; e[1] = rgba[((bb * n + g) * n + r1) * 4 + k];
            e[2] = rgba[((bb * n + g1) * n + r) * 4 + k]; e[3] = rgba[((bb * n + g1) * n + r1) * 4 + k];
          }
      }
  const size_t bytes = faces.size () * sizeof (float);
  float4 *d = nullptr;
  VFHIP_CHECK_HIP (dev_malloc (&d, bytes));
  hipError_t e = upload_in_stream (d, faces.data (), bytes, h->st.s_compute);
  if (e != hipSuccess) { (void) hipFree (d); return set_error (VFHIP_ERR_HIP, "LUT upload failed: %s", hipGetErrorString (e)); }
  // the fast path's table (lut_sample16): per cell and channel the eight coefficients of the trilinear polynomial of 255 * (entry - lattice
  // coordinate), fp16, 64-byte cells (the last 16 bytes unused: a cell never straddles an L2 line).  Computed in double.  A table with an entry
  // fp16 cannot hold (non-finite, or a coefficient beyond its range) gets no such table and the fast path reads the fp32 cells instead.
  std::vector<uint16_t> half (n * n * n * 32, 0);
  bool fits = true;
  auto to_half = [&fits] (double v) -> uint16_t {
    if (!(fabs (v) <= 65504.0)) { fits = false; return 0; }
    const _Float16 hv = (_Float16) v;                       // round to nearest even
    uint16_t bits; memcpy (&bits, &hv, 2);
    return bits;
  };
  for (size_t b = 0; b < n && fits; b++)
    for (size_t g = 0; g < n; g++)
      for (size_t r = 0; r < n; r++) {
        const size_t i1[3] = { r + 1 < n ? r + 1 : r, g + 1 < n ? g + 1 : g, b + 1 < n ? b + 1 : b }, i0[3] = { r, g, b };
        uint16_t *cell = &half[((b * n + g) * n + r) * 32];
        for (size_t k = 0; k < 3; k++) {
          double R[2][2][2];                                  // residual at the corners [dz][dy][dx]
          for (int dz = 0; dz < 2; dz++)
            for (int dy = 0; dy < 2; dy++)
              for (int dx = 0; dx < 2; dx++) {
                const size_t ix = dx ? i1[0] : i0[0], iy = dy ? i1[1] : i0[1], iz = dz ? i1[2] : i0[2];
                const size_t lattice[3] = { ix, iy, iz };
                R[dz][dy][dx] = 255.0 * ((double) rgba[((iz * n + iy) * n + ix) * 4 + k] - (double) lattice[k] / (double) (n - 1));
              }
          // (a clamped +1 corner repeats the cell's own lattice point: its residual difference is zero, and so is the weight it gets — f == 0 there)
          const double k0 = R[0][0][0], kx = R[0][0][1] - k0, ky = R[0][1][0] - k0, kz = R[1][0][0] - k0;
          const double kxy = R[0][1][1] - R[0][0][1] - R[0][1][0] + k0, kxz = R[1][0][1] - R[0][0][1] - R[1][0][0] + k0, kyz = R[1][1][0] - R[0][1][0] - R[1][0][0] + k0;
          const double kxyz = R[1][1][1] - R[1][1][0] - R[1][0][1] - R[0][1][1] + R[1][0][0] + R[0][1][0] + R[0][0][1] - k0;
          const double co[8] = { k0, kx, ky, kxy, kz, kxz, kyz, kxyz };
          for (int q = 0; q < 8; q++) cell[8 * k + q] = to_half (co[q]);
        }
      }
  uint4 *d16 = nullptr;
  if (fits) {
    e = dev_malloc (&d16, half.size () * sizeof (uint16_t));
    if (e == hipSuccess) e = upload_in_stream (d16, half.data (), half.size () * sizeof (uint16_t), h->st.s_compute);
    if (e != hipSuccess) { (void) hipFree (d); if (d16) (void) hipFree (d16); return set_error (VFHIP_ERR_HIP, "LUT upload failed: %s", hipGetErrorString (e)); }
  }
  // swap after the device is idle for this handle's streams: a frame in flight may still read the old table
  (void) hipStreamSynchronize (h->st.s_compute);
  if (h->d_lut) (void) hipFree (h->d_lut);
  if (h->d_lut16) (void) hipFree (h->d_lut16);
  h->d_lut = d; h->d_lut16 = d16; h->lut_size = size;
  return VFHIP_OK;
}

extern "C" {

VfHipVideoFilter *vfhip_videofilter_new (int device)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  VfHipVideoFilter *h = new (std::nothrow) VfHipVideoFilter ();
  if (!h) { set_error (VFHIP_ERR_NOMEM, "out of memory"); return nullptr; }
  h->dev = d;
  if (h->st.init (d) != VFHIP_OK) { delete h; return nullptr; }
  return h;
}

int vfhip_videofilter_configure (VfHipVideoFilter *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out)
{
  if (!h || !in || !out) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "configure with %d submitted frame(s) still in flight: wait for them first", h->fl.count);
  if (in->width <= 0 || in->height <= 0 || in->width > 32768 || in->height > 32768)
    return set_error (VFHIP_ERR_INVALID, "bad frame size %dx%d", in->width, in->height);
  if (in->width != out->width || in->height != out->height)
    return set_error (VFHIP_ERR_INVALID, "videofilter does not scale (%dx%d -> %dx%d)", in->width, in->height, out->width, out->height);
  // pad templates of the reference: BGRA, RGBA, NV12, I420 (videofilter/gstvfmetalvideofilter.m:53-65)
  if (in->format < VFHIP_FORMAT_BGRA || in->format > VFHIP_FORMAT_I420 || out->format < VFHIP_FORMAT_BGRA || out->format > VFHIP_FORMAT_I420)
    return set_error (VFHIP_ERR_UNSUPPORTED, "videofilter: format not supported");
  h->in = *in; h->out = *out; h->configured = true;
  return VFHIP_OK;
}

int vfhip_videofilter_process (VfHipVideoFilter *h, const VfHipFrame *in, VfHipFrame *out, const VfHipVideoFilterParams *prm)
{
  int rc = vf_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "frames submitted with vfhip_videofilter_submit are still in flight");
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  VfHipFrame din, dout;
  if ((rc = upload_frame (h->st, 0, in, &din))) return rc;
  if ((rc = output_frame (h->st, 1, &h->out, out, &dout))) return rc;
  VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if ((rc = vf_launch (h, &din, &dout, prm, h->st.s_compute))) return rc;
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  return download_frame (h->st, 1, &dout, out);
}

int vfhip_videofilter_submit (VfHipVideoFilter *h, const VfHipFrame *in, VfHipFrame *out, const VfHipVideoFilterParams *prm)
{
  int rc = vf_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  const VfHipVideoFilterParams p = *prm;              // the launch takes the parameters by value: nothing of `prm` is kept
  return flights_submit (h->st, h->fl, &h->out, in, out,
      [h, &p] (const VfHipFrame *di, VfHipFrame *dout, hipStream_t s) { return vf_launch (h, di, dout, &p, s); });
}

int vfhip_videofilter_wait (VfHipVideoFilter *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null handle");
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return flights_wait (h->st, h->fl);
}

int vfhip_videofilter_in_flight (VfHipVideoFilter *h)
{
  if (!h) return 0;
  std::lock_guard<std::mutex> lk (h->mu);
  return h->fl.count;
}

int vfhip_videofilter_process_device (VfHipVideoFilter *h, const VfHipFrame *in, VfHipFrame *out,
    const VfHipVideoFilterParams *prm, void *stream)
{
  int rc = vf_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return vf_launch (h, in, out, prm, stream ? (hipStream_t) stream : h->st.s_compute);
}

int vfhip_videofilter_process_device_batch (VfHipVideoFilter *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, const VfHipVideoFilterParams *prm, void *stream)
{
  int rc = vf_check (h, in0, out0, prm);
  if (rc) return rc;
  if (n_frames < 1 || n_frames > 65535) return set_error (VFHIP_ERR_INVALID, "n_frames %d outside 1..65535", n_frames);
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return vf_launch (h, in0, out0, prm, stream ? (hipStream_t) stream : h->st.s_compute, n_frames, in_frame_pitch, out_frame_pitch);
}

int vfhip_videofilter_set_lut (VfHipVideoFilter *h, const float *rgba, int size)
{
  if (!h || !rgba) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  return vf_upload_lut (h, rgba, size);
}

// .cube and .png LUT files: parsed on the host (host_parsers.hip), uploaded like vfhip_videofilter_set_lut
int vfhip_videofilter_load_lut (VfHipVideoFilter *h, const char *path)
{
  if (!h || !path) return set_error (VFHIP_ERR_INVALID, "null argument");
  const size_t n = strlen (path);
  std::vector<float> lut;
  int size = 0, rc;
  if (n >= 4 && strcasecmp (path + n - 4, ".png") == 0) rc = parse_png_lut (path, lut, &size);
  else if (n >= 5 && strcasecmp (path + n - 5, ".cube") == 0) rc = parse_cube_lut (path, lut, &size);
  else return set_error (VFHIP_ERR_UNSUPPORTED, "LUT files must be .cube or .png (got %s)", path);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  return vf_upload_lut (h, lut.data (), size);
}

void vfhip_videofilter_clear_lut (VfHipVideoFilter *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  (void) hipStreamSynchronize (h->st.s_compute);
  if (h->d_lut) (void) hipFree (h->d_lut);
  if (h->d_lut16) (void) hipFree (h->d_lut16);
  h->d_lut = nullptr; h->d_lut16 = nullptr; h->lut_size = 0;
}

int vfhip_videofilter_lut_size (VfHipVideoFilter *h) { return h ? h->lut_size : 0; }

void vfhip_videofilter_cleanup (VfHipVideoFilter *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  flights_abandon (h->st, h->fl);
  for (auto &b : h->st.slots) { if (b.host) (void) hipHostFree (b.host); if (b.devp) (void) hipFree (b.devp); }
  h->st.slots.clear ();
  h->configured = false;              // the LUT survives cleanup like the reference's _lutTexture (a property, not a caps resource)
}

void vfhip_videofilter_free (VfHipVideoFilter *h)
{
  if (!h) return;
  vfhip_videofilter_cleanup (h);
  (void) hipSetDevice (h->dev->ordinal);
  if (h->d_lut) (void) hipFree (h->d_lut);
  if (h->d_lut16) (void) hipFree (h->d_lut16);
  h->st.destroy ();
  delete h;
}

}  // extern "C"
