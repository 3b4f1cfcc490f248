/* oracle/metalref.c — TEST INFRASTRUCTURE ONLY (never linked into libvfhip, never on the product path).
 *
 * CPU restatement, in plain C float arithmetic, of what the reference's Metal shaders compute for the four
 * hot-path elements, following SURVEY.md Appendix B ("Metal semantics a CPU restatement must model"):
 *   convertscale : convertscale/metalconvertscale_shaders.h:48-269, metalconvertscalerenderer.m:137-166,353-485
 *   deinterlace  : deinterlace/metaldeinterlace_shaders.h:45-218, metaldeinterlacerenderer.m:204-293,326-405
 *   videofilter  : videofilter/metalvideofilter_shaders.h:63-328, metalvideofilterrenderer.m:523-681
 *   compositor   : compositor/metalcomprenderer.m:51-122,199-239,377-513
 *   YUV outputs  : common/vfmetalshaders.m:40-168
 *
 * PARITY UNPINNED: the reference cannot be built or run on Linux (Objective-C + Metal.framework), its tests hold
 * no pixel values, and GStreamer's CPU compositor/deinterlace are absent from this container.
 * Sampler weights, unorm rounding and fast-math builtins live in Apple's driver; this file fixes them as
 * Appendix B states (float weights, round-to-nearest-even unorm8 writes, IEEE sqrt / divide — except the per-pixel divisions of the video filter's colour stages, which like its pow are fixed sequences of IEEE operations: vf_rcp, vf_powf).  The HIP kernels are checked
 * against THIS file to +-1 LSB; nothing here has been compared with real Metal output.
 *
 * Built with -ffp-contract=off so that every expression rounds exactly as written (the HIP side does the same).
 * Multiply-adds that MSL's default fast-math would contract are written as explicit fmaf () — linear interpolation
 * a + (b - a) * f, the colour matrices, the blur sums, the polynomial steps of vf_powf — in the SAME places on both
 * sides (csrc/metal_common.h, csrc/videofilter.hip ...), so the two still agree bit for bit: IEEE fma is exact and
 * deterministic on x86 (vfmadd / glibc) and on gfx950 (v_fma_f32) alike.
 */
#include "metalref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float r, g, b, a; } F4;

static inline float un8 (uint32_t v) { return (float) v * (1.0f / 255.0f); }
static inline float clamp01 (float x) { return fminf (fmaxf (x, 0.0f), 1.0f); }
static inline uint32_t quant8 (float x) { return (uint32_t) lrintf (clamp01 (x) * 255.0f); }   /* RNE under the default rounding mode */
static inline int iclamp (int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float lerp2 (float a, float b, float f) { return fmaf (b - a, f, a); }
static inline uint32_t pack_rgba8 (uint32_t r, uint32_t g, uint32_t b, uint32_t a) { return r | (g << 8) | (b << 16) | (a << 24); }
static inline uint32_t quant_rgba8 (F4 c) { return pack_rgba8 (quant8 (c.r), quant8 (c.g), quant8 (c.b), quant8 (c.a)); }
static inline F4 unpack_rgba8 (uint32_t q) { F4 o = { un8 (q & 0xff), un8 ((q >> 8) & 0xff), un8 ((q >> 16) & 0xff), un8 (q >> 24) }; return o; }

static F4 yuv_to_rgb (float y, float cb, float cr, int m709)
{
  const float yy = y - 16.0f / 255.0f, u = cb - 128.0f / 255.0f, v = cr - 128.0f / 255.0f;
  F4 o;
  const float ly = 1.164383f * yy;                       /* the matrices' zero entries contribute nothing (finite inputs) */
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

static void rgb_to_yuv (float r, float g, float b, int m709, float *y, float *u, float *v)
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

/* ---- sampling ------------------------------------------------------------------------------------- */
typedef struct { int i0, i1; float f; } Taps;
static Taps lin_taps_px (int n, float x)              /* x in texel units, already minus .5 */
{
  const float fl = floorf (x);
  Taps t; t.f = x - fl;
  const int i = (int) fl;
  t.i0 = iclamp (i, 0, n - 1); t.i1 = iclamp (i + 1, 0, n - 1);
  return t;
}
static Taps lin_taps (int n, float coord) { return lin_taps_px (n, coord * (float) n - 0.5f); }
static int near_tap (int n, float coord) { return iclamp ((int) floorf (coord * (float) n), 0, n - 1); }

static float plane_taps (const uint8_t *p, int stride, int bpt, int ch, Taps tx, Taps ty)
{
  const uint8_t *r0 = p + (size_t) ty.i0 * stride, *r1 = p + (size_t) ty.i1 * stride;
  const float a = lerp2 (un8 (r0[tx.i0 * bpt + ch]), un8 (r0[tx.i1 * bpt + ch]), tx.f);
  const float b = lerp2 (un8 (r1[tx.i0 * bpt + ch]), un8 (r1[tx.i1 * bpt + ch]), tx.f);
  return lerp2 (a, b, ty.f);
}
static float plane_sample (const uint8_t *p, int stride, int bpt, int ch, int W, int H, float u, float v, int linear)
{
  if (linear) return plane_taps (p, stride, bpt, ch, lin_taps (W, u), lin_taps (H, v));
  return un8 (p[(size_t) near_tap (H, v) * stride + near_tap (W, u) * bpt + ch]);
}

static F4 sample_rgba (const MrImg *im, float u, float v, int linear)
{
  F4 o;
  switch (im->fmt) {
    case MR_BGRA: case MR_RGBA: {
      const int ro = im->fmt == MR_RGBA ? 0 : 2;
      o.r = plane_sample (im->p[0], im->s[0], 4, ro, im->w, im->h, u, v, linear);
      o.g = plane_sample (im->p[0], im->s[0], 4, 1, im->w, im->h, u, v, linear);
      o.b = plane_sample (im->p[0], im->s[0], 4, 2 - ro, im->w, im->h, u, v, linear);
      o.a = plane_sample (im->p[0], im->s[0], 4, 3, im->w, im->h, u, v, linear);
      return o;
    }
    case MR_NV12: {
      const int cw = (im->w + 1) / 2, ch = (im->h + 1) / 2;
      const float y = plane_sample (im->p[0], im->s[0], 1, 0, im->w, im->h, u, v, linear);
      const float cb = plane_sample (im->p[1], im->s[1], 2, 0, cw, ch, u, v, linear);
      const float cr = plane_sample (im->p[1], im->s[1], 2, 1, cw, ch, u, v, linear);
      return yuv_to_rgb (y, cb, cr, im->m709);
    }
    case MR_I420: {
      const int cw = (im->w + 1) / 2, ch = (im->h + 1) / 2;
      const float y = plane_sample (im->p[0], im->s[0], 1, 0, im->w, im->h, u, v, linear);
      const float cb = plane_sample (im->p[1], im->s[1], 1, 0, cw, ch, u, v, linear);
      const float cr = plane_sample (im->p[2], im->s[2], 1, 0, cw, ch, u, v, linear);
      return yuv_to_rgb (y, cb, cr, im->m709);
    }
    default: {
      const int tw = im->w / 2;
      const float texw = (float) tw, fullw = texw * 2.0f;
      const float px = u * fullw;
      const float mx = floorf (px / 2.0f);
      const float sub = px - mx * 2.0f;
      const int tx = near_tap (tw, (mx + 0.5f) / texw), ty = near_tap (im->h, v);
      const uint8_t *t = im->p[0] + (size_t) ty * im->s[0] + 4 * tx;
      float y, cb, cr;
      if (im->fmt == MR_UYVY) { cb = un8 (t[0]); cr = un8 (t[2]); y = sub < 1.0f ? un8 (t[1]) : un8 (t[3]); }
      else { cb = un8 (t[1]); cr = un8 (t[3]); y = sub < 1.0f ? un8 (t[0]) : un8 (t[2]); }
      return yuv_to_rgb (y, cb, cr, im->m709);
    }
  }
}

/* 1:1 fetch at pixel (x, y) of an image whose full-resolution planes match the output grid: exact luma / RGBA
 * texel; `chroma_linear`: 4:2:0 chroma bilinear at texel coordinate 0.5*x - 0.25 (the linear sampler of the filter /
 * compositor shaders), else nearest chroma texel x/2 (the deinterlace input pass). */
static F4 fetch_1to1 (const MrImg *im, int x, int y, int chroma_linear)
{
  x = iclamp (x, 0, im->w - 1); y = iclamp (y, 0, im->h - 1);
  if (im->fmt == MR_BGRA || im->fmt == MR_RGBA) {
    const uint8_t *t = im->p[0] + (size_t) y * im->s[0] + 4 * x;
    const int ro = im->fmt == MR_RGBA ? 0 : 2;
    F4 o = { un8 (t[ro]), un8 (t[1]), un8 (t[2 - ro]), un8 (t[3]) };
    return o;
  }
  const int cw = (im->w + 1) / 2, ch = (im->h + 1) / 2;
  const float Y = un8 (im->p[0][(size_t) y * im->s[0] + x]);
  float cb, cr;
  if (chroma_linear) {
    const Taps tx = lin_taps_px (cw, 0.5f * (float) x - 0.25f), ty = lin_taps_px (ch, 0.5f * (float) y - 0.25f);
    if (im->fmt == MR_NV12) { cb = plane_taps (im->p[1], im->s[1], 2, 0, tx, ty); cr = plane_taps (im->p[1], im->s[1], 2, 1, tx, ty); }
    else { cb = plane_taps (im->p[1], im->s[1], 1, 0, tx, ty); cr = plane_taps (im->p[2], im->s[2], 1, 0, tx, ty); }
  } else {
    const int cx = iclamp (x >> 1, 0, cw - 1), cy = iclamp (y >> 1, 0, ch - 1);
    if (im->fmt == MR_NV12) { cb = un8 (im->p[1][(size_t) cy * im->s[1] + 2 * cx]); cr = un8 (im->p[1][(size_t) cy * im->s[1] + 2 * cx + 1]); }
    else { cb = un8 (im->p[1][(size_t) cy * im->s[1] + cx]); cr = un8 (im->p[2][(size_t) cy * im->s[2] + cx]); }
  }
  return yuv_to_rgb (Y, cb, cr, im->m709);
}

/* ---- output: logical RGBA8 image (w*h uint32, r in byte 0) -> any of the six formats -------------------- */
static void store_image (const MrImg *o, const uint32_t *q)
{
  const int w = o->w, h = o->h;
#define Q(x, y) q[(size_t) iclamp ((y), 0, h - 1) * w + iclamp ((x), 0, w - 1)]
  switch (o->fmt) {
    case MR_BGRA: case MR_RGBA:
      for (int y = 0; y < h; y++) {
        uint32_t *row = (uint32_t *) (o->p[0] + (size_t) y * o->s[0]);
        for (int x = 0; x < w; x++) {
          uint32_t v = q[(size_t) y * w + x];
          if (o->fmt == MR_BGRA) v = (v & 0xff00ff00u) | ((v & 0xff) << 16) | ((v >> 16) & 0xff);
          row[x] = v;
        }
      }
      break;
    case MR_NV12: case MR_I420:
      for (int by = 0; by < (h + 1) / 2; by++)
        for (int bx = 0; bx < (w + 1) / 2; bx++) {
          float sr = 0.0f, sg = 0.0f, sb = 0.0f;
          for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++) {
              const int x = 2 * bx + dx, y = 2 * by + dy;
              const F4 c = unpack_rgba8 (Q (x, y));
              sr += c.r; sg += c.g; sb += c.b;
              if (x < w && y < h) {
                float Y, U, V; rgb_to_yuv (c.r, c.g, c.b, o->m709, &Y, &U, &V);
                o->p[0][(size_t) y * o->s[0] + x] = (uint8_t) quant8 (Y);
              }
            }
          sr *= 0.25f; sg *= 0.25f; sb *= 0.25f;
          float Y, U, V; rgb_to_yuv (sr, sg, sb, o->m709, &Y, &U, &V);
          if (o->fmt == MR_NV12) { uint8_t *d = o->p[1] + (size_t) by * o->s[1] + 2 * bx; d[0] = (uint8_t) quant8 (U); d[1] = (uint8_t) quant8 (V); }
          else { o->p[1][(size_t) by * o->s[1] + bx] = (uint8_t) quant8 (U); o->p[2][(size_t) by * o->s[2] + bx] = (uint8_t) quant8 (V); }
        }
      break;
    default:   /* an odd width's last half macro-pixel: unwritten by the reference (width/2 threads, shaders.h:210); written
                  here from the clamped edge pixel (the reference's p1 clamp, :217) so the output is fully defined */
      for (int y = 0; y < h; y++)
        for (int bx = 0; bx < (w + 1) / 2; bx++) {
          const F4 c0 = unpack_rgba8 (Q (2 * bx, y)), c1 = unpack_rgba8 (Q (2 * bx + 1, y));
          float ya, ua, va, yb, ub, vb;
          rgb_to_yuv (c0.r, c0.g, c0.b, o->m709, &ya, &ua, &va);
          rgb_to_yuv (c1.r, c1.g, c1.b, o->m709, &yb, &ub, &vb);
          const uint32_t U = quant8 ((ua + ub) * 0.5f), V = quant8 ((va + vb) * 0.5f), Y0 = quant8 (ya), Y1 = quant8 (yb);
          uint32_t *d = (uint32_t *) (o->p[0] + (size_t) y * o->s[0]) + bx;
          *d = o->fmt == MR_UYVY ? pack_rgba8 (U, Y0, V, Y1) : pack_rgba8 (Y0, U, Y1, V);
        }
      break;
  }
#undef Q
}

/* ---- convertscale ------------------------------------------------------------------------------------ */
int metalref_convertscale (const MrImg *in, const MrImg *out, int linear, int add_borders, uint32_t border_argb)
{
  const int w = out->w, h = out->h;
  uint32_t *q = malloc ((size_t) w * h * 4);
  if (!q) return -2;
  float sx = 1.0f, sy = 1.0f;
  if (add_borders && in->w > 0 && in->h > 0) {
    const float src = (float) in->w / (float) in->h, dst = (float) w / (float) h;
    if (src > dst) sy = dst / src; else sx = src / dst;
  }
  const float rw = (float) w * sx, rh = (float) h * sy;
  const float rx = ((float) w - rw) * 0.5f, ry = ((float) h - rh) * 0.5f;
  const uint32_t ba = border_argb >> 24, br = (border_argb >> 16) & 0xff, bg = (border_argb >> 8) & 0xff, bb = border_argb & 0xff;
  const uint32_t border = br | (bg << 8) | (bb << 16) | (ba << 24);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const float cx = (float) x + 0.5f, cy = (float) y + 0.5f;
      if (cx >= rx && cx < rx + rw && cy >= ry && cy < ry + rh)
        q[(size_t) y * w + x] = quant_rgba8 (sample_rgba (in, (cx - rx) / rw, (cy - ry) / rh, linear));
      else
        q[(size_t) y * w + x] = border;
    }
  store_image (out, q);
  free (q);
  return 0;
}

/* ---- deinterlace ------------------------------------------------------------------------------------- */
int metalref_deinterlace (const MrImg *cur, const MrImg *prev, const MrImg *out, int method, int tff, float threshold)
{
  const int w = out->w, h = out->h;
  uint32_t *c = malloc ((size_t) w * h * 4), *p = malloc ((size_t) w * h * 4), *q = malloc ((size_t) w * h * 4);
  if (!c || !p || !q) { free (c); free (p); free (q); return -2; }
  /* input pass: 8-bit RGBA intermediate (nearest chroma); RGBA inputs are copied verbatim */
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      c[(size_t) y * w + x] = quant_rgba8 (fetch_1to1 (cur, x, y, 0));
      if (prev) p[(size_t) y * w + x] = quant_rgba8 (fetch_1to1 (prev, x, y, 0));
    }
  if ((method == MR_DEINT_WEAVE || method == MR_DEINT_GREEDYH) && !prev) method = MR_DEINT_BOB;   /* no history yet */
  for (int y = 0; y < h; y++) {
    const int top = (y % 2) == 0;
    const int keep = tff ? top : !top;
    const int above = y > 0 ? y - 1 : 0, below = y < h - 1 ? y + 1 : h - 1;
    for (int x = 0; x < w; x++) {
      const size_t i = (size_t) y * w + x;
      if (keep) { q[i] = c[i]; continue; }
      const F4 a = unpack_rgba8 (c[(size_t) above * w + x]), b = unpack_rgba8 (c[(size_t) below * w + x]);
      const F4 bob = { (a.r + b.r) * 0.5f, (a.g + b.g) * 0.5f, (a.b + b.b) * 0.5f, (a.a + b.a) * 0.5f };
      if (method == MR_DEINT_BOB || method == MR_DEINT_LINEAR) q[i] = quant_rgba8 (bob);     /* linear == bob (:148) */
      else if (method == MR_DEINT_WEAVE) q[i] = p[i];
      else {
        const F4 cl = unpack_rgba8 (c[i]), pl = unpack_rgba8 (p[i]);
        const float dr = cl.r - pl.r, dg = cl.g - pl.g, db = cl.b - pl.b;
        const float motion = sqrtf (dr * dr + dg * dg + db * db);
        q[i] = motion < threshold ? p[i] : quant_rgba8 (bob);
      }
    }
  }
  store_image (out, q);
  free (c); free (p); free (q);
  return 0;
}

/* ---- videofilter -------------------------------------------------------------------------------------- */
static inline float fractf (float x) { return x - floorf (x); }
static inline float mixf (float a, float b, float t) { return fmaf (b - a, t, a); }
static inline float stepf (float e, float x) { return x < e ? 0.0f : 1.0f; }
static inline float smoothstepf (float e0, float e1, float x)
{
  if (!(e0 < e1)) return stepf (e0, x);                 /* undefined in MSL for e0 >= e1: define as step (Appendix B item 8) */
  const float inv = 1.0f / (e1 - e0);                   /* the edges are uniforms: one reciprocal per frame, a multiply per pixel */
  const float t = clamp01 ((x - e0) * inv);
  return t * t * (3.0f - 2.0f * t);
}
static float hash12 (float px, float py, uint32_t frame)
{
  const float fo = (float) frame * 0.00137f;
  float x = fractf (px * 0.1031f + fo), y = fractf (py * 0.1031f + fo), z = fractf (px * 0.1031f + fo);
  const float d = x * (y + 33.33f) + y * (z + 33.33f) + z * (x + 33.33f);
  x += d; y += d; z += d;
  return fractf ((x + y) * z);
}
/* 1 / x for a positive normal x — the divisions of the colour stages.  MSL compiles `a / b` under fast-math to a * rcp (b) with a
 * hardware reciprocal approximation; an IEEE division is ten instructions on the GPU (two of them quarter-rate) and buys nothing an
 * 8-bit result can show.  So the restatement defines its own reciprocal in plain IEEE operations in a fixed order, like vf_powf:
 * the exponent-flip seed (relative error < 12 %) and three Newton steps r += r * (1 - x * r), each as two fma (relative error
 * < 6e-8, checked on 400 k values from 1e-10 to 10).  The HIP kernel evaluates the identical sequence: both sides agree bit for bit. */
static inline float vf_rcp (float x)
{
  uint32_t u; memcpy (&u, &x, 4);
  u = 0x7EF311C7u - u;
  float r; memcpy (&r, &u, 4);
  r = fmaf (r, fmaf (-x, r, 1.0f), r);
  r = fmaf (r, fmaf (-x, r, 1.0f), r);
  r = fmaf (r, fmaf (-x, r, 1.0f), r);
  return r;
}
static void rgb_to_hsv (float r, float g, float b, float *h, float *s, float *v)
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
static void hsv_to_rgb (float h, float s, float v, float *r, float *g, float *b)
{
  const float pr = fabsf (fractf (h + 1.0f) * 6.0f - 3.0f);
  const float pg = fabsf (fractf (h + 2.0f / 3.0f) * 6.0f - 3.0f);
  const float pb = fabsf (fractf (h + 1.0f / 3.0f) * 6.0f - 3.0f);
  *r = v * mixf (1.0f, clamp01 (pr - 1.0f), s);
  *g = v * mixf (1.0f, clamp01 (pg - 1.0f), s);
  *b = v * mixf (1.0f, clamp01 (pb - 1.0f), s);
}


/* pow(x, y) for x in [1e-4, 1], y > 0 — the gamma stage.  MSL's fast-math pow is not correctly rounded and neither
 * libm nor OCML agree with it or with each other bit for bit, so the restatement defines its own: log2 by an atanh
 * series on the mantissa reduced to [sqrt(.5), sqrt(2)), exp2 by a degree-7 polynomial on [-.5, .5], both in plain
 * IEEE single-precision operations in a fixed order (relative error ~2e-7, far below 1/255).  The HIP kernel
 * evaluates the identical sequence, so the two sides agree bit for bit. */
static inline float vf_powf (float x, float y)
{
  uint32_t ux; memcpy (&ux, &x, 4);
  int e = (int) (ux >> 23) - 127;
  uint32_t um = (ux & 0x007fffffu) | 0x3f800000u;
  float m; memcpy (&m, &um, 4);
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float t = (m - 1.0f) * vf_rcp (m + 1.0f), t2 = t * t;
  float p = 0.11111111f;
  p = fmaf (p, t2, 0.14285714f); p = fmaf (p, t2, 0.2f); p = fmaf (p, t2, 0.33333333f); p = fmaf (p, t2, 1.0f);
  const float l2 = fmaf (t * p, 2.88539008f, (float) e);     /* 2 / ln 2 */
  const float z = y * l2;
  if (z < -126.0f) return 0.0f;
  const float zi = floorf (z + 0.5f), f = (z - zi) * 0.69314718f;
  float q = 1.98412698e-4f;
  q = fmaf (q, f, 1.38888889e-3f); q = fmaf (q, f, 8.33333333e-3f); q = fmaf (q, f, 4.16666667e-2f); q = fmaf (q, f, 0.16666667f);
  q = fmaf (q, f, 0.5f); q = fmaf (q, f, 1.0f); q = fmaf (q, f, 1.0f);
  uint32_t uq; memcpy (&uq, &q, 4);
  uq += (uint32_t) (int32_t) zi << 23;              /* unsigned shift: zi is negative for x < 1 */
  float r; memcpy (&r, &uq, 4);
  return r;
}

static F4 color_adjust (F4 c, const MrFilterParams *u, float tu, float tv, int W, int H)
{
  float r = c.r, g = c.g, b = c.b, a = c.a;
  r += u->brightness; g += u->brightness; b += u->brightness;
  r = fmaf (r - 0.5f, u->contrast, 0.5f); g = fmaf (g - 0.5f, u->contrast, 0.5f); b = fmaf (b - 0.5f, u->contrast, 0.5f);
  const float lum = fmaf (b, 0.0722f, fmaf (g, 0.7152f, r * 0.2126f));
  r = mixf (lum, r, u->saturation); g = mixf (lum, g, u->saturation); b = mixf (lum, b, u->saturation);
  if (fabsf (u->hue) > 0.001f) {
    float h, s, v;
    rgb_to_hsv (clamp01 (r), clamp01 (g), clamp01 (b), &h, &s, &v);
    h = fractf (h + u->hue / (2.0f * 3.14159265358979323846f));
    hsv_to_rgb (h, s, v, &r, &g, &b);
  }
  const float ig = 1.0f / u->gamma;
  r = fminf (fmaxf (r, 0.0001f), 1.0f); g = fminf (fmaxf (g, 0.0001f), 1.0f); b = fminf (fmaxf (b, 0.0001f), 1.0f);
  if (ig != 1.0f) { r = vf_powf (r, ig); g = vf_powf (g, ig); b = vf_powf (b, ig); }      /* pow (x, 1) == x exactly */
  if (u->sepia > 0.001f) {
    const float sr = fmaf (b, 0.189f, fmaf (g, 0.769f, r * 0.393f));
    const float sg = fmaf (b, 0.168f, fmaf (g, 0.686f, r * 0.349f));
    const float sb = fmaf (b, 0.131f, fmaf (g, 0.534f, r * 0.272f));
    r = mixf (r, sr, u->sepia); g = mixf (g, sg, u->sepia); b = mixf (b, sb, u->sepia);
  }
  if (u->invert) { r = 1.0f - r; g = 1.0f - g; b = 1.0f - b; }
  if (u->chroma_key_enabled) {
    const float dr = r - u->key_r, dg = g - u->key_g, db = b - u->key_b;
    const float dist = sqrtf (dr * dr + dg * dg + db * db);
    a *= smoothstepf (u->key_tolerance, u->key_tolerance + u->key_smoothness, dist);
  }
  if (u->vignette > 0.001f) {
    const float cx = tu - 0.5f, cy = tv - 0.5f;
    const float dist = sqrtf (cx * cx + cy * cy) * 1.414f;
    const float vig = 1.0f - smoothstepf (0.5f, 1.0f, dist) * u->vignette;
    r *= vig; g *= vig; b *= vig;
  }
  if (u->noise > 0.001f) {
    float n = hash12 (tu * (float) W, tv * (float) H, u->frame_index);
    n = (n - 0.5f) * u->noise * 0.5f;
    r += n; g += n; b += n;
  }
  F4 o = { clamp01 (r), clamp01 (g), clamp01 (b), a };
  return o;
}

static void lut_sample (const float *lut, int N, float cr, float cg, float cb, float *r, float *g, float *b)
{
  const float scale = (float) (N - 1) / (float) N, offset = 0.5f / (float) N;
  const Taps tx = lin_taps (N, cr * scale + offset), ty = lin_taps (N, cg * scale + offset), tz = lin_taps (N, cb * scale + offset);
  float o[3];
  for (int k = 0; k < 3; k++) {
#define L(x, y, z) lut[(((size_t) (z) * N + (y)) * N + (x)) * 4 + k]
    const float c00 = lerp2 (L (tx.i0, ty.i0, tz.i0), L (tx.i1, ty.i0, tz.i0), tx.f);
    const float c10 = lerp2 (L (tx.i0, ty.i1, tz.i0), L (tx.i1, ty.i1, tz.i0), tx.f);
    const float c01 = lerp2 (L (tx.i0, ty.i0, tz.i1), L (tx.i1, ty.i0, tz.i1), tx.f);
    const float c11 = lerp2 (L (tx.i0, ty.i1, tz.i1), L (tx.i1, ty.i1, tz.i1), tx.f);
#undef L
    o[k] = lerp2 (lerp2 (c00, c10, ty.f), lerp2 (c01, c11, ty.f), tz.f);
  }
  *r = o[0]; *g = o[1]; *b = o[2];
}

static const float kBlurW[9] = { 0.028532f, 0.067234f, 0.124009f, 0.179044f, 0.20236f, 0.179044f, 0.124009f, 0.067234f, 0.028532f };

int metalref_videofilter (const MrImg *in, const MrImg *out, const MrFilterParams *p, const float *lut, int lut_size)
{
  const int w = out->w, h = out->h;
  uint32_t *rt = malloc ((size_t) w * h * 4), *t1 = malloc ((size_t) w * h * 4), *t2 = malloc ((size_t) w * h * 4);
  if (!rt || !t1 || !t2) { free (rt); free (t1); free (t2); return -2; }
  /* the texture coordinate of a pixel centre, (x + .5) / size, as a multiplication by the once-divided reciprocal (the rasteriser's
   * interpolation is not specified to the bit either way; the kernel does the same) */
  const float inv_w = 1.0f / (float) w, inv_h = 1.0f / (float) h;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const float tu = ((float) x + 0.5f) * inv_w, tv = ((float) y + 0.5f) * inv_h;
      F4 c = fetch_1to1 (in, x, y, 1);
      c = color_adjust (c, p, tu, tv, w, h);
      if (lut && lut_size >= 2) lut_sample (lut, lut_size, c.r, c.g, c.b, &c.r, &c.g, &c.b);
      rt[(size_t) y * w + x] = quant_rgba8 (c);
    }
  const uint32_t *fin = rt;
  if (p->sharpness < -0.001f || p->sharpness > 0.001f) {
    for (int pass = 0; pass < 2; pass++) {
      const uint32_t *src = pass == 0 ? rt : t1; uint32_t *dst = pass == 0 ? t1 : t2;
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
          F4 s = { 0, 0, 0, 0 };
          for (int i = 0; i < 9; i++) {
            const int xx = pass == 0 ? iclamp (x + i - 4, 0, w - 1) : x, yy = pass == 0 ? y : iclamp (y + i - 4, 0, h - 1);
            const F4 c = unpack_rgba8 (src[(size_t) yy * w + xx]);
            s.r = fmaf (c.r, kBlurW[i], s.r); s.g = fmaf (c.g, kBlurW[i], s.g); s.b = fmaf (c.b, kBlurW[i], s.b); s.a = fmaf (c.a, kBlurW[i], s.a);
          }
          dst[(size_t) y * w + x] = quant_rgba8 (s);
        }
    }
    const float amount = p->sharpness;
    for (size_t i = 0; i < (size_t) w * h; i++) {
      const F4 o = unpack_rgba8 (rt[i]), b = unpack_rgba8 (t2[i]);
      F4 r;
      if (amount > 0.0f) {
        r.r = clamp01 (fmaf (o.r - b.r, amount, o.r)); r.g = clamp01 (fmaf (o.g - b.g, amount, o.g)); r.b = clamp01 (fmaf (o.b - b.b, amount, o.b));
      } else {
        const float t = fabsf (amount);
        r.r = mixf (o.r, b.r, t); r.g = mixf (o.g, b.g, t); r.b = mixf (o.b, b.b, t);
      }
      r.a = o.a;
      t1[i] = quant_rgba8 (r);
    }
    fin = t1;
  }
  store_image (out, fin);
  free (rt); free (t1); free (t2);
  return 0;
}

/* ---- compositor --------------------------------------------------------------------------------------- */
int metalref_compositor (const MrPad *pads, int n, int background, const MrImg *out)
{
  const int w = out->w, h = out->h;
  uint32_t *q = malloc ((size_t) w * h * 4);
  if (!q) return -2;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t v;
      switch (background) {
        case MR_BG_BLACK: v = 0xff000000u; break;
        case MR_BG_WHITE: v = 0xffffffffu; break;
        case MR_BG_TRANSPARENT: v = 0u; break;
        default: {
          const float tu = ((float) x + 0.5f) / (float) w, tv = ((float) y + 0.5f) / (float) h;
          const int px = (int) (tu * (float) w), py = (int) (tv * (float) h);
          const float gray = ((px / 8) + (py / 8)) % 2 ? 0.75f : 0.5f;
          const F4 c = { gray, gray, gray, 1.0f };
          v = quant_rgba8 (c);
        }
      }
      q[(size_t) y * w + x] = v;
    }
  for (int k = 0; k < n; k++) {
    const MrPad *pd = &pads[k];
    if (pd->width <= 0 || pd->height <= 0) continue;
    const float alpha = (float) pd->alpha;
    const int x0 = iclamp (pd->xpos, 0, w), x1 = iclamp (pd->xpos + pd->width, 0, w);
    const int y0 = iclamp (pd->ypos, 0, h), y1 = iclamp (pd->ypos + pd->height, 0, h);
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) {
        const float tu = (((float) x + 0.5f) - (float) pd->xpos) / (float) pd->width;
        const float tv = (((float) y + 0.5f) - (float) pd->ypos) / (float) pd->height;
        /* an unscaled pad samples texel centres: the linear sampler returns the exact texel (Appendix B item 2),
         * 4:2:0 chroma still interpolates at its .25/.75 phases */
        const int unscaled = pd->width == pd->img.w && pd->height == pd->img.h;
        F4 s = unscaled ? fetch_1to1 (&pd->img, x - pd->xpos, y - pd->ypos, 1) : sample_rgba (&pd->img, tu, tv, 1);
        s.a *= alpha; s.r *= s.a; s.g *= s.a; s.b *= s.a;
        const F4 d = unpack_rgba8 (q[(size_t) y * w + x]);
        F4 o;
        if (pd->blend == MR_BLEND_SOURCE) o = s;
        else if (pd->blend == MR_BLEND_ADD) { o.r = s.r + d.r; o.g = s.g + d.g; o.b = s.b + d.b; o.a = s.a + d.a; }
        else { const float k1 = 1.0f - s.a; o.r = fmaf (d.r, k1, s.r); o.g = fmaf (d.g, k1, s.g); o.b = fmaf (d.b, k1, s.b); o.a = fmaf (d.a, k1, s.a); }
        q[(size_t) y * w + x] = quant_rgba8 (o);
      }
  }
  store_image (out, q);
  free (q);
  return 0;
}

/* ---- transform ---------------------------------------------------------------------------------------- */
/* UV matrix of the eight methods, column-major [m00 m10 m01 m11] (transform/metaltransformrenderer.m:44-104) */
static void transform_matrix (int method, float m[4])
{
  static const float T[8][4] = {
    {  1,  0,  0,  1 }, {  0, -1,  1,  0 }, { -1,  0,  0, -1 }, {  0,  1, -1,  0 },
    { -1,  0,  0,  1 }, {  1,  0,  0, -1 }, {  0,  1,  1,  0 }, {  0, -1, -1,  0 },
  };
  for (int k = 0; k < 4; k++) m[k] = T[method & 7][k];
}

int metalref_transform (const MrImg *in, const MrImg *out, int method, int crop_top, int crop_bottom, int crop_left, int crop_right)
{
  const int w = out->w, h = out->h;
  uint32_t *q = malloc ((size_t) w * h * 4);
  if (!q) return -2;
  /* crop folded into the UV transform (metaltransformrenderer.m:265-293) */
  const float cl = (float) crop_left / (float) in->w, cr = (float) crop_right / (float) in->w;
  const float ct = (float) crop_top / (float) in->h, cb = (float) crop_bottom / (float) in->h;
  const float sx = 1.0f - cl - cr, sy = 1.0f - ct - cb, ox = (cl - cr) * 0.5f, oy = (ct - cb) * 0.5f;
  float t[4]; transform_matrix (method, t);
  const float m0 = t[0] * sx, m1 = t[1] * sx, m2 = t[2] * sy, m3 = t[3] * sy;
  const float offx = t[0] * ox + t[2] * oy + 0.0f, offy = t[1] * ox + t[3] * oy + 0.0f;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float tx = ((float) x + 0.5f) / (float) w, ty = ((float) y + 0.5f) / (float) h;
      tx -= 0.5f; ty -= 0.5f;
      const float ux = m0 * tx + m2 * ty, uy = m1 * tx + m3 * ty;
      tx = ux + (0.5f + offx); ty = uy + (0.5f + offy);
      if (tx < 0.0f || tx > 1.0f || ty < 0.0f || ty > 1.0f) q[(size_t) y * w + x] = 0xff000000u;   /* opaque black (:72-74) */
      else q[(size_t) y * w + x] = quant_rgba8 (sample_rgba (in, tx, ty, 1));
    }
  store_image (out, q);
  free (q);
  return 0;
}


/* ---- overlay: video sampled 1:1, the image bilinear inside its rectangle, rgb = mix (video, image, image.a * alpha)
 * (overlayFragmentRGBA / NV12 / I420, overlay/metaloverlay_shaders.h:60-151).  Quirk kept: the reference's decoder hands
 * the shader PREMULTIPLIED colours which it then mixes as if they were straight (metaloverlayrenderer.m:214-219). ---- */
int metalref_overlay (const MrImg *in, const MrImg *out, const MrImg *ov, float ox, float oy, float ow, float oh, float alpha)
{
  const int w = out->w, h = out->h;
  uint32_t *q = malloc ((size_t) w * h * 4);
  if (!q) return -2;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      F4 v = fetch_1to1 (in, x, y, 1);
      if (ov) {
        const float tu = ((float) x + 0.5f) / (float) w, tv = ((float) y + 0.5f) / (float) h;
        const float px = tu * (float) w, py = tv * (float) h;
        if (px >= ox && px < ox + ow && py >= oy && py < oy + oh) {
          const F4 o = sample_rgba (ov, (px - ox) / ow, (py - oy) / oh, 1);
          const float a = o.a * alpha;
          v.r = v.r + (o.r - v.r) * a; v.g = v.g + (o.g - v.g) * a; v.b = v.b + (o.b - v.b) * a;
        }
      }
      q[(size_t) y * w + x] = quant_rgba8 (v);
    }
  store_image (out, q);
  free (q);
  return 0;
}
