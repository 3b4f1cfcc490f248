// csrc/deinterlace.hip — vfhip_deinterlace_* : bob / weave / linear / greedy-H deinterlacer.
// Mirrors MetalDeinterlaceRenderer (reference deinterlace/metaldeinterlacerenderer.{h,m}) and restates the
// kernels of deinterlace/metaldeinterlace_shaders.h:45-218 (`metal` numerics: float on unorm8, 8-bit RGBA
// intermediates).  What the reference does in 4-5 passes over 8-bit RGBA textures (YUV->RGBA render pass, the
// deinterlace compute pass, RGBA->YUV compute pass, a blit for the history and two command-buffer waits,
// metaldeinterlacerenderer.m:295-413) is ONE kernel here: the 8-bit RGBA intermediate of a pixel is a pure
// function of the input bytes, so it is recomputed in registers for the (at most three) taps a pixel needs,
// and the history is the previous INPUT frame in its native format (12.4 MB for NV12 2160p instead of 33 MB).
#include "vfhip_internal.h"
#include "metal_common.h"
#include <cmath>
#include <cstdlib>

using namespace vfhip;

namespace vfhip {

struct DeintParams {
  metal::Img cur, prev;      // prev.p[0] == nullptr: no history
  metal::OutImg out;
  int method, tff;
  float threshold;
  // batch of consecutive frames of ONE stream: frame z at base + z * pitch; its history is frame z-1 of the batch
  // (frame 0: `prev`, the handle's stored history)
  size_t in_pitch, out_pitch;
};

__device__ __forceinline__ DeintParams deint_frame (const DeintParams &p, unsigned z)
{
  DeintParams q = p;
  q.cur = metal::img_at (p.cur, z * p.in_pitch);
  q.out = metal::out_at (p.out, z * p.out_pitch);
  if (z > 0) q.prev = metal::img_at (p.cur, (z - 1) * p.in_pitch);
  return q;
}

// the reference's _inputRGBA texel: YUV -> RGB with NEAREST chroma, quantised to 8 bits; RGBA bytes pass through
__device__ __forceinline__ uint32_t deint_input_rgba8 (const metal::Img &im, int x, int y)
{
  return metal::quant_rgba8 (metal::fetch_1to1 (im, x, y, false));
}

__device__ __forceinline__ uint32_t deint_pixel (const DeintParams &p, int x, int y)
{
  const int h = p.out.h;
  const bool top = (y & 1) == 0;
  const bool keep = p.tff ? top : !top;
  const uint32_t c = deint_input_rgba8 (p.cur, x, y);
  if (keep) return c;
  int method = p.method;
  if ((method == VFHIP_DEINTERLACE_WEAVE || method == VFHIP_DEINTERLACE_GREEDYH) && !p.prev.p[0]) method = VFHIP_DEINTERLACE_BOB;
  if (method == VFHIP_DEINTERLACE_WEAVE) return deint_input_rgba8 (p.prev, x, y);
  bool bob = true;
  uint32_t pq = 0;
  if (method == VFHIP_DEINTERLACE_GREEDYH) {
    pq = deint_input_rgba8 (p.prev, x, y);
    const metal::F4 cl = metal::unpack_rgba8 (c), pl = metal::unpack_rgba8 (pq);
    const float dr = cl.r - pl.r, dg = cl.g - pl.g, db = cl.b - pl.b;
    const float motion = sqrtf (dr * dr + dg * dg + db * db);
    bob = !(motion < p.threshold);
  }
  if (!bob) return pq;
  const int above = y > 0 ? y - 1 : 0, below = y < h - 1 ? y + 1 : h - 1;
  const metal::F4 a = metal::unpack_rgba8 (deint_input_rgba8 (p.cur, x, above)), b = metal::unpack_rgba8 (deint_input_rgba8 (p.cur, x, below));
  metal::F4 o;
  o.r = (a.r + b.r) * 0.5f; o.g = (a.g + b.g) * 0.5f; o.b = (a.b + b.b) * 0.5f; o.a = (a.a + b.a) * 0.5f;
  return metal::quant_rgba8 (o);
}

__global__ __launch_bounds__ (256) void k_deinterlace (const DeintParams pp)
{
  const DeintParams p = deint_frame (pp, blockIdx.z);
  const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
  if (2 * bx >= p.out.w || 2 * by >= p.out.h) return;
  uint32_t q[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++)
      q[dy][dx] = deint_pixel (p, min (2 * bx + dx, p.out.w - 1), min (2 * by + dy, p.out.h - 1));
  metal::store_block (p.out, bx, by, q);
}

// ---- 4:2:0 inputs: column strips with a sliding window -------------------------------------------------------
// One lane owns one chroma column (2 pixels wide) and walks down a strip of rows two at a time.  The 8-bit RGBA
// intermediate of each source row is computed ONCE and carried in registers to serve as the "above" / "below" tap of
// its neighbours (k_deinterlace recomputes it up to three times); the previous frame is only converted on the lines
// that need it.  Same arithmetic, same results; ~1.5 instead of ~2.5 YUV->RGB conversions per pixel.
struct Q2 { uint32_t a, b; };          // logical RGBA8 of pixels (2k, y) and (2k+1, y)

template <bool PLANAR>
__device__ __forceinline__ Q2 deint_row_rgba8 (const metal::Img &im, int k, int y)
{
  const int x0 = 2 * k, x1 = min (2 * k + 1, im.w - 1);
  const uint8_t *yr = im.p[0] + (size_t) y * im.s[0];
  const int cy = y >> 1;
  typedef uint16_t __attribute__ ((aligned (1))) u16_any;        // 2-byte loads at any address (one instruction instead of two)
  const uint32_t yy = (x1 != x0) ? (uint32_t) *reinterpret_cast<const u16_any *> (yr + x0) : (uint32_t) yr[x0] * 0x0101u;
  const uint32_t Y0 = yy & 0xffu, Y1 = yy >> 8;
  uint32_t U, V;
  if (PLANAR) { U = im.p[1][(size_t) cy * im.s[1] + k]; V = im.p[2][(size_t) cy * im.s[2] + k]; }
  else { const uint32_t c = *reinterpret_cast<const u16_any *> (im.p[1] + (size_t) cy * im.s[1] + 2 * k); U = c & 0xffu; V = c >> 8; }
  const float cb = metal::un8 (U), cr = metal::un8 (V);
  Q2 q;
  q.a = metal::quant_rgba8 (metal::yuv_to_rgb (metal::un8 (Y0), cb, cr, im.m709));
  q.b = metal::quant_rgba8 (metal::yuv_to_rgb (metal::un8 (Y1), cb, cr, im.m709));
  return q;
}

__device__ __forceinline__ uint32_t deint_bob8 (uint32_t above, uint32_t below)
{
  const metal::F4 a = metal::unpack_rgba8 (above), b = metal::unpack_rgba8 (below);
  metal::F4 o;
  o.r = (a.r + b.r) * 0.5f; o.g = (a.g + b.g) * 0.5f; o.b = (a.b + b.b) * 0.5f; o.a = (a.a + b.a) * 0.5f;
  return metal::quant_rgba8 (o);
}

// output pixel of a discarded-field line given its own, above, below and previous-frame RGBA8
__device__ __forceinline__ uint32_t deint_other8 (int method, float thr, uint32_t cur, uint32_t above, uint32_t below, uint32_t prev)
{
  if (method == VFHIP_DEINTERLACE_WEAVE) return prev;
  if (method == VFHIP_DEINTERLACE_GREEDYH) {
    const metal::F4 cl = metal::unpack_rgba8 (cur), pl = metal::unpack_rgba8 (prev);
    const float dr = cl.r - pl.r, dg = cl.g - pl.g, db = cl.b - pl.b;
    if (sqrtf (dr * dr + dg * dg + db * db) < thr) return prev;
  }
  return deint_bob8 (above, below);
}

constexpr int DEINT_ROWS = 8;

template <bool PLANAR>
__global__ __launch_bounds__ (256) void k_deinterlace_420 (const DeintParams pp)
{
  const DeintParams p = deint_frame (pp, blockIdx.y);
  const int cw = (p.out.w + 1) >> 1, h = p.out.h;
  const int strips = (h + DEINT_ROWS - 1) / DEINT_ROWS;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= cw * strips) return;
  const int strip = t / cw, k = t - strip * cw;
  const int y0 = strip * DEINT_ROWS, yend = min (y0 + DEINT_ROWS, h);
  int method = p.method;
  if ((method == VFHIP_DEINTERLACE_WEAVE || method == VFHIP_DEINTERLACE_GREEDYH) && !p.prev.p[0]) method = VFHIP_DEINTERLACE_BOB;
  const bool need_prev = method == VFHIP_DEINTERLACE_WEAVE || method == VFHIP_DEINTERLACE_GREEDYH;
  Q2 qa = deint_row_rgba8<PLANAR> (p.cur, k, max (y0 - 1, 0));      // row above the pair
  Q2 qb = deint_row_rgba8<PLANAR> (p.cur, k, y0);                    // first row of the pair
  for (int y = y0; y < yend; y += 2) {
    const int r1 = min (y + 1, h - 1), r2 = min (y + 2, h - 1);
    const Q2 qc = deint_row_rgba8<PLANAR> (p.cur, k, r1);            // second row of the pair
    const Q2 qd = deint_row_rgba8<PLANAR> (p.cur, k, r2);            // row below the pair (= first row of the next pair)
    const bool keep0 = p.tff ? true : false;                          // y is even: top field
    uint32_t q[2][2];
    if (keep0) {                                                       // row y kept, row y+1 reconstructed
      q[0][0] = qb.a; q[0][1] = qb.b;
      Q2 pv = { 0u, 0u };
      if (need_prev) pv = deint_row_rgba8<PLANAR> (p.prev, k, r1);
      q[1][0] = deint_other8 (method, p.threshold, qc.a, qb.a, qd.a, pv.a);
      q[1][1] = deint_other8 (method, p.threshold, qc.b, qb.b, qd.b, pv.b);
    } else {                                                           // row y reconstructed (above = y-1 clamped), row y+1 kept
      Q2 pv = { 0u, 0u };
      if (need_prev) pv = deint_row_rgba8<PLANAR> (p.prev, k, y);
      q[0][0] = deint_other8 (method, p.threshold, qb.a, qa.a, qc.a, pv.a);
      q[0][1] = deint_other8 (method, p.threshold, qb.b, qa.b, qc.b, pv.b);
      q[1][0] = qc.a; q[1][1] = qc.b;
    }
    if (y + 1 >= h) { q[1][0] = q[0][0]; q[1][1] = q[0][1]; }         // odd height: edge-clamped duplicate for the 2x2 mean
    if (2 * k + 1 >= p.out.w) { q[0][1] = q[0][0]; q[1][1] = q[1][0]; }
    metal::store_block (p.out, k, y >> 1, q);
    qa = qc; qb = qd;
  }
}

// ---- 4:2:0 in and out, width % 4 == 0, even height: four pixels per lane, float intermediates ------------------------
// k_deinterlace_420q.  Same values as the kernels above (and as the reference's three passes), organised for the
// machine: a lane owns FOUR adjacent pixels (one dword of luma, two chroma columns) and walks a strip of rows in pairs.
//   * The reference's 8-bit RGBA intermediate lives in registers as the float an 8-bit texel reads back as — byte / 255
//     with the byte obtained by x255, round-to-nearest-even (`quantf`) — so nothing is packed to bytes and unpacked again
//     between the input pass, the method pass and the RGB -> YUV pass (each row used to be unpacked up to three times).
//   * Every source row is converted once per strip and carried as the above / below tap of its neighbours.
//   * greedy-H compares the squared distance with the smallest float whose correctly rounded square root reaches the
//     threshold (computed on the host: `motion2_limit`), which decides exactly like sqrt (d2) < threshold without the
//     square root.
//   * Dword loads / stores on luma and on NV12 chroma, 32-bit offsets from wave-uniform plane bases.
struct Rgb4 { float r[4], g[4], b[4]; };
// the two matrices of metal_common.h as wave-uniform coefficient sets (selected once per kernel: no per-pixel branch on m709)
struct YuvCoef { float rv, gu, gv, bu; };
struct RgbCoef { float yr, yg, yb, ur, ug, ub, vr, vg, vb; };
__device__ __forceinline__ YuvCoef yuv_coef (int m709)
{
  YuvCoef k;
  k.rv = m709 ? 1.792741f : 1.596027f; k.gu = m709 ? -0.213249f : -0.391762f; k.gv = m709 ? -0.532909f : -0.812968f; k.bu = m709 ? 2.112402f : 2.017232f;
  return k;
}
__device__ __forceinline__ RgbCoef rgb_coef (int m709)
{
  RgbCoef k;
  k.yr = m709 ? 0.182586f : 0.256788f; k.yg = m709 ? 0.614231f : 0.504129f; k.yb = m709 ? 0.062007f : 0.097906f;
  k.ur = m709 ? -0.100644f : -0.148223f; k.ug = m709 ? -0.338572f : -0.290993f; k.ub = 0.439216f;
  k.vr = 0.439216f; k.vg = m709 ? -0.398942f : -0.367788f; k.vb = m709 ? -0.040274f : -0.071427f;
  return k;
}

__device__ __forceinline__ float quantf01 (float x)      // what an 8-bit unorm texel written with x in [0, 1] reads back as
{
  return __builtin_rintf (x * 255.0f) * (1.0f / 255.0f);
}
// clamp01 (fmaf (a, b, c)) in ONE instruction: the clamp output modifier of v_fma_f32 saturates the correctly rounded fma result
// to [0, 1] (the compiler emits fma + v_max ... clamp: a fifth of this kernel's instructions were such clamps)
__device__ __forceinline__ float fma_sat (float a, float b, float c)
{
  float d;
  asm ("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <bool PLANAR>
__device__ __forceinline__ Rgb4 deint_row4 (const uint8_t *yp, const uint8_t *up, const uint8_t *vp, uint32_t ys, uint32_t cs, const YuvCoef &k, uint32_t q, int y)
{
  const uint32_t Y4 = *reinterpret_cast<const uint32_t *> (yp + (__umul24 ((uint32_t) y, ys) + 4u * q));
  uint32_t U0, V0, U1, V1;
  if (PLANAR) {
    const uint32_t co = __umul24 ((uint32_t) (y >> 1), cs) + 2u * q;
    const uint32_t u2 = *reinterpret_cast<const uint16_t *> (up + co), v2 = *reinterpret_cast<const uint16_t *> (vp + co);
    U0 = u2 & 0xffu; U1 = u2 >> 8; V0 = v2 & 0xffu; V1 = v2 >> 8;
  } else {
    const uint32_t c4 = *reinterpret_cast<const uint32_t *> (up + (__umul24 ((uint32_t) (y >> 1), cs) + 4u * q));
    U0 = c4 & 0xffu; V0 = (c4 >> 8) & 0xffu; U1 = (c4 >> 16) & 0xffu; V1 = c4 >> 24;
  }
  const float cb[2] = { metal::un8 (U0), metal::un8 (U1) }, cr[2] = { metal::un8 (V0), metal::un8 (V1) };
  Rgb4 o;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    // metal::yuv_to_rgb with the coefficients in registers (same operations, same order)
    const float ly = 1.164383f * (metal::un8 ((Y4 >> (8 * i)) & 0xffu) - 16.0f / 255.0f);
    const float u = cb[i >> 1] - 128.0f / 255.0f, v = cr[i >> 1] - 128.0f / 255.0f;
    o.r[i] = quantf01 (fma_sat (k.rv, v, ly)); o.g[i] = quantf01 (fma_sat (k.gv, v, fmaf (k.gu, u, ly))); o.b[i] = quantf01 (fma_sat (k.bu, u, ly));
  }
  return o;
}

// rows y (o0) and y+1 (o1) of one lane's four columns -> NV12 / I420 (the reference's rgbaToNV12 / rgbaToI420 pass:
// luma per pixel, chroma from the mean of each 2x2 block, summed in the reference's order)
template <bool PLANAR>
__device__ __forceinline__ void deint_store4 (const metal::OutImg &o, const RgbCoef &k, uint32_t q, int y, const Rgb4 &o0, const Rgb4 &o1)
{
  uint32_t l0 = 0, l1 = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    // metal::rgb_to_yuv (luma row), coefficients in registers
    const float Y0 = fmaf (k.yb, o0.b[i], fmaf (k.yg, o0.g[i], k.yr * o0.r[i])) + 16.0f / 255.0f;
    const float Y1 = fmaf (k.yb, o1.b[i], fmaf (k.yg, o1.g[i], k.yr * o1.r[i])) + 16.0f / 255.0f;
    l0 = __builtin_amdgcn_cvt_pk_u8_f32 (Y0 * 255.0f, (uint32_t) i, l0);
    l1 = __builtin_amdgcn_cvt_pk_u8_f32 (Y1 * 255.0f, (uint32_t) i, l1);
  }
  const uint32_t lo = __umul24 ((uint32_t) y, (uint32_t) o.s[0]) + 4u * q;
  __builtin_nontemporal_store (l0, reinterpret_cast<uint32_t *> (o.p[0] + lo));
  __builtin_nontemporal_store (l1, reinterpret_cast<uint32_t *> (o.p[0] + (lo + (uint32_t) o.s[0])));
  uint32_t uu[2], vv[2];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    sr += o0.r[2 * c]; sg += o0.g[2 * c]; sb += o0.b[2 * c];
    sr += o0.r[2 * c + 1]; sg += o0.g[2 * c + 1]; sb += o0.b[2 * c + 1];
    sr += o1.r[2 * c]; sg += o1.g[2 * c]; sb += o1.b[2 * c];
    sr += o1.r[2 * c + 1]; sg += o1.g[2 * c + 1]; sb += o1.b[2 * c + 1];
    sr *= 0.25f; sg *= 0.25f; sb *= 0.25f;
    const float U = fmaf (k.ub, sb, fmaf (k.ug, sg, k.ur * sr)) + 128.0f / 255.0f;
    const float V = fmaf (k.vb, sb, fmaf (k.vg, sg, k.vr * sr)) + 128.0f / 255.0f;
    uu[c] = metal::quant8 (U); vv[c] = metal::quant8 (V);
  }
  if (PLANAR) {
    const uint32_t co = __umul24 ((uint32_t) (y >> 1), (uint32_t) o.s[1]) + 2u * q;
    *reinterpret_cast<uint16_t *> (o.p[1] + co) = (uint16_t) (uu[0] | (uu[1] << 8));
    *reinterpret_cast<uint16_t *> (o.p[2] + (__umul24 ((uint32_t) (y >> 1), (uint32_t) o.s[2]) + 2u * q)) = (uint16_t) (vv[0] | (vv[1] << 8));
  } else {
    const uint32_t co = __umul24 ((uint32_t) (y >> 1), (uint32_t) o.s[1]) + 4u * q;
    __builtin_nontemporal_store (uu[0] | (vv[0] << 8) | (uu[1] << 16) | (vv[1] << 24), reinterpret_cast<uint32_t *> (o.p[1] + co));
  }
}

// the reconstructed line: own pixel `cur`, its neighbours in the kept field, the previous frame's pixel
template <int METHOD>
__device__ __forceinline__ Rgb4 deint_recon4 (const Rgb4 &cur, const Rgb4 &above, const Rgb4 &below, const Rgb4 &prev, float m2_limit)
{
  if (METHOD == VFHIP_DEINTERLACE_WEAVE) return prev;
  Rgb4 o;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    // the mean of two texel values is inside [0, 1]: no clamp
    const float br = quantf01 ((above.r[i] + below.r[i]) * 0.5f), bg = quantf01 ((above.g[i] + below.g[i]) * 0.5f), bb = quantf01 ((above.b[i] + below.b[i]) * 0.5f);
    bool still = false;                                   // greedy-H: no motion against the previous frame -> weave (selects, no divergent branch)
    if (METHOD == VFHIP_DEINTERLACE_GREEDYH) {
      const float dr = cur.r[i] - prev.r[i], dg = cur.g[i] - prev.g[i], db = cur.b[i] - prev.b[i];
      still = dr * dr + dg * dg + db * db < m2_limit;
    }
    o.r[i] = still ? prev.r[i] : br; o.g[i] = still ? prev.g[i] : bg; o.b[i] = still ? prev.b[i] : bb;
  }
  return o;
}

constexpr int DEINTQ_ROWS = 8;

template <bool PLANAR, bool TFF, int METHOD>
__global__ __launch_bounds__ (256) void k_deinterlace_420q (const DeintParams pp, float m2_limit)
{
  const DeintParams p = deint_frame (pp, blockIdx.y);
  const int quads = p.out.w >> 2, h = p.out.h;
  const int strips = (h + DEINTQ_ROWS - 1) / DEINTQ_ROWS;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= quads * strips) return;
  const int strip = t / quads;
  const uint32_t q = (uint32_t) (t - strip * quads);
  const int y0 = strip * DEINTQ_ROWS, yend = min (y0 + DEINTQ_ROWS, h);
  constexpr bool NEED_PREV = METHOD == VFHIP_DEINTERLACE_WEAVE || METHOD == VFHIP_DEINTERLACE_GREEDYH;
  // frame 0 of a stream (or of a batch on a fresh handle) has no history: weave / greedy-H fall back to bob for that frame
  // only (wave-uniform: blockIdx.y picks the frame), the rest of the batch uses its predecessor in the batch
  const bool hist = p.prev.p[0] != nullptr;
  const uint8_t *cy = p.cur.p[0], *cu = p.cur.p[1], *cv = p.cur.p[2];
  const uint8_t *py = p.prev.p[0], *pu = p.prev.p[1], *pv = p.prev.p[2];
  const uint32_t ys = (uint32_t) p.cur.s[0], cs = (uint32_t) p.cur.s[1], pys = (uint32_t) p.prev.s[0], pcs = (uint32_t) p.prev.s[1];
  const YuvCoef kc = yuv_coef (p.cur.m709), kp = yuv_coef (p.prev.m709);
  const RgbCoef ko = rgb_coef (p.out.m709);
  // TFF: even rows are kept, odd rows reconstructed from the kept rows above (y) and below (y + 2);
  // BFF: odd rows are kept, even rows reconstructed from the kept rows above (y - 1) and below (y + 1).
  Rgb4 carry = deint_row4<PLANAR> (cy, cu, cv, ys, cs, kc, q, TFF ? y0 : max (y0 - 1, 0));
  for (int y = y0; y < yend; y += 2) {
    Rgb4 prev {};
    if (TFF) {
      Rgb4 cur {};
      if (METHOD == VFHIP_DEINTERLACE_GREEDYH && hist) cur = deint_row4<PLANAR> (cy, cu, cv, ys, cs, kc, q, y + 1);      // only the motion test reads it
      const Rgb4 below = deint_row4<PLANAR> (cy, cu, cv, ys, cs, kc, q, min (y + 2, h - 1));
      if (NEED_PREV && hist) prev = deint_row4<PLANAR> (py, pu, pv, pys, pcs, kp, q, y + 1);
      const Rgb4 rec = (!NEED_PREV || hist) ? deint_recon4<METHOD> (cur, carry, below, prev, m2_limit)
                                            : deint_recon4<VFHIP_DEINTERLACE_BOB> (cur, carry, below, prev, m2_limit);
      deint_store4<PLANAR> (p.out, ko, q, y, carry, rec);
      carry = below;
    } else {
      Rgb4 cur {};
      if (METHOD == VFHIP_DEINTERLACE_GREEDYH && hist) cur = deint_row4<PLANAR> (cy, cu, cv, ys, cs, kc, q, y);
      const Rgb4 kept = deint_row4<PLANAR> (cy, cu, cv, ys, cs, kc, q, y + 1);
      if (NEED_PREV && hist) prev = deint_row4<PLANAR> (py, pu, pv, pys, pcs, kp, q, y);
      const Rgb4 rec = (!NEED_PREV || hist) ? deint_recon4<METHOD> (cur, carry, kept, prev, m2_limit)
                                            : deint_recon4<VFHIP_DEINTERLACE_BOB> (cur, carry, kept, prev, m2_limit);
      deint_store4<PLANAR> (p.out, ko, q, y, rec, kept);
      carry = kept;
    }
  }
}

}  // namespace vfhip

struct VfHipDeinterlace {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo info {};
  // host path: three upload slots used in turn (the previous one is the history; with two frames in flight the third is
  // the only one no kernel can still be reading), outputs in slots 3 / 4
  unsigned seq = 0;
  Flights fl;                       // pipelined host path (submit / wait)
  bool has_prev = false;
  VfHipFrame prev_dev {};           // device-side previous input frame (host path: staging slot; device path: hist buffer)
  void *hist[2] = { nullptr, nullptr }; size_t hist_bytes = 0;   // device path: ping-pong history images
  int hist_cur = 0;
};

// smallest float d with sqrtf (d) >= thr: (sqrtf (m2) < thr) == (m2 < d) for every m2 >= 0, because the correctly rounded
// square root is monotone.  thr <= 0 -> 0 (never below), NaN -> NaN (never below), beyond sqrt (FLT_MAX) -> +inf.
static float motion2_limit (float thr)
{
  if (!(thr > 0.0f)) return thr != thr ? thr : 0.0f;
  float d = thr * thr;
  if (std::isinf (d)) return d;
  while (d > 0.0f && sqrtf (d) >= thr) d = nextafterf (d, 0.0f);
  while (sqrtf (d) < thr) d = nextafterf (d, INFINITY);
  return d;
}

// k_deinterlace_420q's contract: 4:2:0 frame, width % 4 == 0, even height, every plane (and the batch pitches) aligned
// for the dword / 16-bit accesses it makes
static bool deint_quad_ok (const VfHipVideoInfo &info, const VfHipFrame *cur, const VfHipFrame *prev, const VfHipFrame *out, size_t in_pitch, size_t out_pitch)
{
  static const bool enabled = [] { const char *e = getenv ("VFHIP_DEINT_QUAD"); return !e || atoi (e) != 0; } ();     // test / A-B knob
  if (!enabled) return false;
  const bool nv12 = info.format == VFHIP_FORMAT_NV12, i420 = info.format == VFHIP_FORMAT_I420;
  if (!(nv12 || i420) || (info.width & 3) || (info.height & 1) || info.width < 4 || info.height < 2) return false;
  if ((in_pitch | out_pitch) & 3) return false;
  auto ok = [&] (const VfHipFrame *f) {
    if (!f) return true;
    if (((uintptr_t) f->data[0] | (uintptr_t) f->stride[0]) & 3) return false;
    const uintptr_t cm = nv12 ? 3 : 1;
    if (((uintptr_t) f->data[1] | (uintptr_t) f->stride[1]) & cm) return false;
    if (i420 && (((uintptr_t) f->data[2] | (uintptr_t) f->stride[2]) & 1)) return false;
    if (i420 && f->stride[1] != f->stride[2]) return false;       // one chroma offset serves U and V on the input side
    return true;
  };
  return ok (cur) && ok (prev) && ok (out);
}

static int deint_launch (VfHipDeinterlace *h, const VfHipFrame *cur, const VfHipFrame *prev, VfHipFrame *out,
    const VfHipDeinterlaceParams *prm, hipStream_t s, int n_frames = 1, size_t in_pitch = 0, size_t out_pitch = 0)
{
  DeintParams p {};
  p.in_pitch = in_pitch; p.out_pitch = out_pitch;
  p.cur = metal::make_img (cur);
  if (prev) p.prev = metal::make_img (prev);
  p.out = metal::make_out (out);
  p.method = prm->method; p.tff = prm->top_field_first != 0; p.threshold = prm->motion_threshold;
  const int bw = (h->info.width + 1) / 2, bh = (h->info.height + 1) / 2;
  if (deint_quad_ok (h->info, cur, prev, out, in_pitch, out_pitch)) {
    // the whole frame in dwords: k_deinterlace_420q, one instantiation per (layout, field order, method)
    int method = p.method;
    if ((method == VFHIP_DEINTERLACE_WEAVE || method == VFHIP_DEINTERLACE_GREEDYH) && !p.prev.p[0] && n_frames == 1) method = VFHIP_DEINTERLACE_BOB;   // no history at all
    if (method == VFHIP_DEINTERLACE_LINEAR) method = VFHIP_DEINTERLACE_BOB;                    // the reference's linear IS bob (shaders.h:134-148)
    const int strips = (h->info.height + DEINTQ_ROWS - 1) / DEINTQ_ROWS;
    dim3 grid ((unsigned) (((size_t) (h->info.width / 4) * strips + 255) / 256), (unsigned) n_frames);
    const float lim = motion2_limit (p.threshold);
    const bool planar = h->info.format == VFHIP_FORMAT_I420;
#define VF_DQ(PL, TF, M) hipLaunchKernelGGL ((k_deinterlace_420q<PL, TF, M>), grid, dim3 (256), 0, s, p, lim)
#define VF_DQ_M(PL, TF) do { if (method == VFHIP_DEINTERLACE_BOB) VF_DQ (PL, TF, VFHIP_DEINTERLACE_BOB); else if (method == VFHIP_DEINTERLACE_WEAVE) VF_DQ (PL, TF, VFHIP_DEINTERLACE_WEAVE); \
                              else VF_DQ (PL, TF, VFHIP_DEINTERLACE_GREEDYH); } while (0)
    if (planar) { if (p.tff) VF_DQ_M (true, true); else VF_DQ_M (true, false); }
    else { if (p.tff) VF_DQ_M (false, true); else VF_DQ_M (false, false); }
#undef VF_DQ_M
#undef VF_DQ
  } else if (h->info.format == VFHIP_FORMAT_NV12 || h->info.format == VFHIP_FORMAT_I420) {
    const int strips = (h->info.height + DEINT_ROWS - 1) / DEINT_ROWS;
    dim3 grid ((unsigned) (((size_t) bw * strips + 255) / 256), (unsigned) n_frames);
    if (h->info.format == VFHIP_FORMAT_I420) hipLaunchKernelGGL (k_deinterlace_420<true>, grid, dim3 (256), 0, s, p);
    else hipLaunchKernelGGL (k_deinterlace_420<false>, grid, dim3 (256), 0, s, p);
  } else {
    dim3 grid ((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4), (unsigned) n_frames);
    hipLaunchKernelGGL (k_deinterlace, grid, dim3 (64, 4), 0, s, p);
  }
  VFHIP_CHECK_HIP (hipGetLastError ());
  return VFHIP_OK;
}

static int deint_check (VfHipDeinterlace *h, const VfHipFrame *in, const VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  if (!h || !prm) return set_error (VFHIP_ERR_INVALID, "null argument");
  if (!h->configured) return set_error (VFHIP_ERR_NOT_CONFIGURED, "deinterlace: process before configure");
  if (prm->method < 0 || prm->method > 3) return set_error (VFHIP_ERR_INVALID, "bad deinterlace method %d", prm->method);
  int rc = check_frame (in, &h->info, "input");
  if (rc) return rc;
  return check_frame (out, &h->info, "output");
}

static int deint_device_locked (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0, size_t in_pitch, size_t out_pitch, int n_frames,
    const VfHipDeinterlaceParams *prm, hipStream_t s);

extern "C" {

VfHipDeinterlace *vfhip_deinterlace_new (int device)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  VfHipDeinterlace *h = new (std::nothrow) VfHipDeinterlace ();
  if (!h) { set_error (VFHIP_ERR_NOMEM, "out of memory"); return nullptr; }
  h->dev = d;
  if (h->st.init (d) != VFHIP_OK) { delete h; return nullptr; }
  return h;
}

int vfhip_deinterlace_configure (VfHipDeinterlace *h, const VfHipVideoInfo *info)
{
  if (!h || !info) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "configure with %d submitted frame(s) still in flight: wait for them first", h->fl.count);
  if (info->width <= 0 || info->height <= 0 || info->width > 32768 || info->height > 32768)
    return set_error (VFHIP_ERR_INVALID, "bad frame size %dx%d", info->width, info->height);
  // pad template of the reference: BGRA, RGBA, NV12, I420 (deinterlace/gstvfmetaldeinterlace.m:43-55)
  if (info->format < VFHIP_FORMAT_BGRA || info->format > VFHIP_FORMAT_I420)
    return set_error (VFHIP_ERR_UNSUPPORTED, "deinterlace: format %d not supported", info->format);
  h->info = *info;
  h->has_prev = false;               // reference resets the history on reconfigure (metaldeinterlacerenderer.m:180)
  h->configured = true;
  return VFHIP_OK;
}

int vfhip_deinterlace_reset (VfHipDeinterlace *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  h->has_prev = false;
  return VFHIP_OK;
}

// one frame onto the handle's streams: upload (or use in place) -> kernel with the history -> download queued behind it
static int deint_submit_locked (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  if (h->fl.count >= 2) return set_error (VFHIP_ERR_INVALID, "two frames are already in flight: call vfhip_deinterlace_wait first");
  const int k = (h->fl.head + h->fl.count) & 1;
  const size_t in_slot = h->seq % 3, out_slot = 3 + (size_t) k;
  VfHipFrame din, dout;
  int rc;
  if ((rc = upload_frame (h->st, in_slot, in, &din))) return rc;
  if ((rc = output_frame (h->st, out_slot, &h->info, out, &dout))) return rc;
  VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if (in->flags & VFHIP_FRAME_FLAG_DEVICE) {
    // device-resident input (memory:HIPMemory buffer): it belongs to the caller and may be recycled after this frame, so
    // the history is copied device-to-device like in the process_device path
    if ((rc = deint_device_locked (h, &din, &dout, 0, 0, 1, prm, h->st.s_compute))) return rc;
  } else {
    if ((rc = deint_launch (h, &din, h->has_prev ? &h->prev_dev : nullptr, &dout, prm, h->st.s_compute))) return rc;
    h->prev_dev = din; h->has_prev = true;              // the staging slot of this frame is the next frame's history
  }
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  h->fl.f[k].out = *out;
  if ((rc = download_begin (h->st, out_slot, &h->fl.f[k].out, h->fl.f[k].staged, h->st.ev_done[k]))) return rc;
  h->fl.count++; h->seq++;
  return VFHIP_OK;
}

static int deint_wait_locked (VfHipDeinterlace *h)
{
  if (h->fl.count == 0) return set_error (VFHIP_ERR_INVALID, "no frame in flight");
  const int k = h->fl.head;
  h->fl.head ^= 1; h->fl.count--;
  return download_finish (h->st, 3 + (size_t) k, &h->fl.f[k].out, h->fl.f[k].staged, h->st.ev_done[k]);
}

int vfhip_deinterlace_process (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  int rc = deint_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "frames submitted with vfhip_deinterlace_submit are still in flight");
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  if ((rc = deint_submit_locked (h, in, out, prm))) return rc;
  return deint_wait_locked (h);
}

int vfhip_deinterlace_submit (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  int rc = deint_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return deint_submit_locked (h, in, out, prm);
}

int vfhip_deinterlace_wait (VfHipDeinterlace *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null handle");
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return deint_wait_locked (h);
}

int vfhip_deinterlace_in_flight (VfHipDeinterlace *h)
{
  if (!h) return 0;
  std::lock_guard<std::mutex> lk (h->mu);
  return h->fl.count;
}

// history = the previous input frame, kept in one of two internal device images and filled by a stream-ordered
// device-to-device copy (the reference blits _inputRGBA -> _prevFrameRGBA, :394-405).  Writing the history from
// inside the kernel was measured 3x SLOWER (2-byte stores per lane: 9.2 k vs 27.3 k frames/s on NV12 2160p).
// In a batch the history of frame k is frame k-1 of the batch itself; only the LAST frame is copied.
static int deint_device_locked (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0, size_t in_pitch, size_t out_pitch, int n_frames,
    const VfHipDeinterlaceParams *prm, hipStream_t s)
{
  int rc;
  size_t total = 0, off[VFHIP_MAX_PLANES] = { 0 };
  const int np = format_n_planes (in0->info.format);
  for (int p = 0; p < np; p++) { off[p] = total; total += (frame_plane_bytes (in0, p) + 255) / 256 * 256; }
  if (h->hist_bytes < total) {
    for (int k = 0; k < 2; k++) { if (h->hist[k]) (void) hipFree (h->hist[k]); h->hist[k] = nullptr; }
    h->hist_bytes = 0; h->has_prev = false;
    for (int k = 0; k < 2; k++) VFHIP_CHECK_HIP (dev_malloc (&h->hist[k], total));
    h->hist_bytes = total;
  }
  VfHipFrame next = *in0;
  const int nxt = 1 - h->hist_cur;
  for (int p = 0; p < np; p++) next.data[p] = (uint8_t *) h->hist[nxt] + off[p];
  if ((rc = deint_launch (h, in0, h->has_prev ? &h->prev_dev : nullptr, out0, prm, s, n_frames, in_pitch, out_pitch))) return rc;
  const size_t last = (size_t) (n_frames - 1) * in_pitch;
  for (int p = 0; p < np; p++)
    VFHIP_CHECK_HIP (hipMemcpyAsync (next.data[p], (const uint8_t *) in0->data[p] + last, frame_plane_bytes (in0, p), hipMemcpyDeviceToDevice, s));
  h->prev_dev = next; h->hist_cur = nxt;
  h->has_prev = true;
  return VFHIP_OK;
}

static int deint_device (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0, size_t in_pitch, size_t out_pitch, int n_frames,
    const VfHipDeinterlaceParams *prm, void *stream)
{
  int rc = deint_check (h, in0, out0, prm);
  if (rc) return rc;
  if (n_frames < 1 || n_frames > 65535) return set_error (VFHIP_ERR_INVALID, "n_frames %d outside 1..65535", n_frames);
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return deint_device_locked (h, in0, out0, in_pitch, out_pitch, n_frames, prm, stream ? (hipStream_t) stream : h->st.s_compute);
}

int vfhip_deinterlace_process_device (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out,
    const VfHipDeinterlaceParams *prm, void *stream)
{
  return deint_device (h, in, out, 0, 0, 1, prm, stream);
}

int vfhip_deinterlace_process_device_batch (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, const VfHipDeinterlaceParams *prm, void *stream)
{
  return deint_device (h, in0, out0, in_frame_pitch, out_frame_pitch, n_frames, prm, stream);
}

void vfhip_deinterlace_cleanup (VfHipDeinterlace *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  flights_abandon (h->st, h->fl);
  for (int k = 0; k < 2; k++) { if (h->hist[k]) (void) hipFree (h->hist[k]); h->hist[k] = nullptr; }
  h->hist_bytes = 0;
  for (auto &b : h->st.slots) { if (b.host) (void) hipHostFree (b.host); if (b.devp) (void) hipFree (b.devp); }
  h->st.slots.clear ();
  h->has_prev = false; h->configured = false;       // reference: -cleanup drops the history (metaldeinterlacerenderer.m:422)
}

void vfhip_deinterlace_free (VfHipDeinterlace *h)
{
  if (!h) return;
  vfhip_deinterlace_cleanup (h);
  h->st.destroy ();
  delete h;
}

}  // extern "C"
