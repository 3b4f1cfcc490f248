/* oracle/gst114.c — TEST INFRASTRUCTURE ONLY (never linked into libvfhip, never on the product path).
 *
 * CPU restatement of the arithmetic GStreamer 1.14.0's `videoconvert ! videoscale` pipeline applies
 * on the north-star path NV12/I420 -> BGRA/RGBA (+ 2-tap bilinear / nearest scale).  This is the
 * parity oracle BASELINE.json's north_star names ("match GStreamer's CPU videoconvert/videoscale
 * ... within +-1 LSB"); the reference plugin (/root/reference, Objective-C + Metal) cannot be built
 * on Linux, and its own tests pin no pixel values.
 *
 * PINNED: every rule below was checked byte-for-byte against the real GStreamer 1.14.0 elements in
 * the build container (tools/gen_goldens.py regenerates tests/golden npz files from them; the not-gpu
 * test-suite replays those fixtures through this file).  Rules (SURVEY.md §8c "Pinned arithmetic"):
 *   1. chroma 4:2:0 -> 4:4:4: horizontal then vertical, edge replicated, integer shifts;
 *   2. ORC AYUV->ARGB matrix with mulhs() on byte-splatted int16 samples;
 *   3. bilinear: 8-bit 2-tap; vertical pass first iff in_h > out_h + 2, else horizontal first;
 *      vertical taps centre-aligned, horizontal taps edge-aligned with a truncating 16.16 increment;
 *   4. nearest: floor(((x + .5) / out) * in) evaluated in IEEE double in exactly that order
 *      (200 -> 100 picks source 28, not 29, for x = 14: 0.145 * 200 = 28.999999999999996).
 * What it replaces in the reference: the float shader pipeline convertScaleFragmentNV12/I420/RGBA
 * (convertscale/metalconvertscale_shaders.h:71-148) + yuvToRGB (common/vfmetalshaders.m:40-79),
 * whose arithmetic differs from GStreamer's CPU path (SURVEY.md finding 3).
 */
#include "gst114.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads used by the OpenMP loops (bench.py cpu_baseline leg) */
int gst114_set_threads (int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads (n);
  return omp_get_max_threads ();
#else
  (void) n; return 1;
#endif
}

static const int COEF[3][5] = {
  { 298, 409, 516, -100, -208 },   /* bt601  */
  { 298, 459, 541,  -55, -136 },   /* bt709  */
  { 298, 430, 548,  -48, -167 },   /* bt2020 */
};

static inline int splat16 (int x)          /* int16((s<<8)|(s&0xff)), s = int8(x-128) */
{
  int b = (x ^ 0x80) & 0xff;
  int v = b * 257;
  return v >= 32768 ? v - 65536 : v;
}
static inline int mulhs (int a, int c) { return (a * c) >> 16; }   /* arithmetic shift == floor */
static inline int sat16 (int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
static inline int sat8 (int v) { return v < -128 ? -128 : v > 127 ? 127 : v; }
static inline int clampi (int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

void gst114_yuv_to_rgb (int matrix, int Y, int U, int V, int *r, int *g, int *b)
{
  const int *p = COEF[matrix];
  int wy = mulhs (splat16 (Y), p[0]);
  *r = sat8 (sat16 (wy + mulhs (splat16 (V), p[1]))) + 128;
  *b = sat8 (sat16 (wy + mulhs (splat16 (U), p[2]))) + 128;
  *g = sat8 (sat16 (sat16 (wy + mulhs (splat16 (U), p[3])) + mulhs (splat16 (V), p[4]))) + 128;
}

/* horizontal 2x chroma upsample of one row of `cw` samples (stride `cs` bytes between samples)
 * into `w` samples. */
static void upsample_h (const uint8_t *c, int cs, int cw, int w, int cosited, uint8_t *out)
{
  for (int x = 0; x < w; x++) {
    int k = x >> 1;
    int c0 = c[k * cs];
    int cm = c[clampi (k - 1, 0, cw - 1) * cs];
    int cp = c[clampi (k + 1, 0, cw - 1) * cs];
    int v;
    if (cosited) v = (x & 1) ? (c0 + cp + 1) >> 1 : c0;
    else         v = (x & 1) ? (3 * c0 + cp + 2) >> 2 : (3 * c0 + cm + 2) >> 2;
    out[x] = (uint8_t) v;
  }
}

/* full-resolution YUV 4:2:0 -> packed 8-bit RGB.  `planar`: 0 = NV12 (u,v interleaved in uv plane),
 * 1 = I420 (separate u and v planes, chroma nearest-replicated: GStreamer's I420 fast path). */
int gst114_yuv420_to_rgb (const uint8_t *yp, int ys, const uint8_t *up, int us, const uint8_t *vp, int vs,
    int planar, int w, int h, int matrix, int cosited, int out_format, uint8_t *out, int os)
{
  if (w <= 0 || h <= 0 || matrix < 0 || matrix > 2) return -1;
  int cw = (w + 1) / 2, ch = (h + 1) / 2;
  int ro = out_format == GST114_RGBA ? 0 : 2, bo = 2 - ro;
  uint8_t *hu = malloc ((size_t) ch * w), *hv = malloc ((size_t) ch * w);
  if (!hu || !hv) { free (hu); free (hv); return -2; }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < ch; j++) {
    if (planar) {
      for (int x = 0; x < w; x++) { hu[(size_t) j * w + x] = up[(size_t) j * us + (x >> 1)]; hv[(size_t) j * w + x] = vp[(size_t) j * vs + (x >> 1)]; }
    } else {
      upsample_h (up + (size_t) j * us, 2, cw, w, cosited, hu + (size_t) j * w);
      upsample_h (up + (size_t) j * us + 1, 2, cw, w, cosited, hv + (size_t) j * w);
    }
  }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    int j = y >> 1;
    int jn = (y & 1) ? clampi (j + 1, 0, ch - 1) : clampi (j - 1, 0, ch - 1);
    const uint8_t *u0 = hu + (size_t) j * w, *u1 = hu + (size_t) jn * w;
    const uint8_t *v0 = hv + (size_t) j * w, *v1 = hv + (size_t) jn * w;
    uint8_t *o = out + (size_t) y * os;
    for (int x = 0; x < w; x++) {
      int U, V, r, g, b;
      if (planar) { U = u0[x]; V = v0[x]; }
      else { U = (3 * u0[x] + u1[x] + 2) >> 2; V = (3 * v0[x] + v1[x] + 2) >> 2; }
      gst114_yuv_to_rgb (matrix, yp[(size_t) y * ys + x], U, V, &r, &g, &b);
      o[4 * x + ro] = (uint8_t) r; o[4 * x + 1] = (uint8_t) g; o[4 * x + bo] = (uint8_t) b; o[4 * x + 3] = 255;
    }
  }
  free (hu); free (hv);
  return 0;
}

/* ---- videoscale, 4 x u8 ---------------------------------------------------------------- */

void gst114_vtaps (int in_h, int out_h, int y, int *i0, int *i1, int *w)
{
  /* centre-aligned; weight quantised to 8 bits */
  double p = (y + 0.5) * in_h / out_h - 0.5;
  int i = (int) __builtin_floor (p);
  *w = (int) __builtin_floor ((p - i) * 256.0 + 0.5);
  *i0 = clampi (i, 0, in_h - 1); *i1 = clampi (i + 1, 0, in_h - 1);
}

uint32_t gst114_hinc (int in_w, int out_w)
{
  if (out_w <= 1) return 0;
  return (uint32_t) ((((uint64_t) (in_w - 1)) << 16) / (uint64_t) (out_w - 1)) - 1;
}

int gst114_nearest_index (int in, int out, int j)
{
  volatile double t = ((double) j + 0.5) / (double) out;   /* volatile: keep the two roundings separate */
  volatile double p = t * (double) in;
  int i = (int) __builtin_floor (p);
  return clampi (i, 0, in - 1);
}

static void vscale (const uint8_t *in, int is, int w, int h, uint8_t *out, int os, int oh)
{
#pragma omp parallel for schedule(static)
  for (int y = 0; y < oh; y++) {
    int i0, i1, wt; gst114_vtaps (h, oh, y, &i0, &i1, &wt);
    const uint8_t *s1 = in + (size_t) i0 * is, *s2 = in + (size_t) i1 * is;
    uint8_t *o = out + (size_t) y * os;
    for (int k = 0; k < 4 * w; k++) o[k] = (uint8_t) (s1[k] + ((((int) s2[k] - (int) s1[k]) * wt + 128) >> 8));
  }
}

static void hscale (const uint8_t *in, int is, int w, int h, uint8_t *out, int os, int ow)
{
  uint32_t inc = gst114_hinc (w, ow);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    const uint8_t *s = in + (size_t) y * is; uint8_t *o = out + (size_t) y * os;
    for (int x = 0; x < ow; x++) {
      uint32_t t = (uint32_t) x * inc; int i = t >> 16, f = (t >> 8) & 0xff;
      int i1 = i + 1 < w ? i + 1 : w - 1;
      for (int c = 0; c < 4; c++) o[4 * x + c] = (uint8_t) ((s[4 * i + c] * (256 - f) + s[4 * i1 + c] * f) >> 8);
    }
  }
}

int gst114_scale_4u8 (const uint8_t *in, int is, int w, int h, uint8_t *out, int os, int ow, int oh, int method)
{
  if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0) return -1;
  if (method == GST114_NEAREST) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < oh; y++) {
      int sy = gst114_nearest_index (h, oh, y);
      for (int x = 0; x < ow; x++) {
        int sx = gst114_nearest_index (w, ow, x);
        memcpy (out + (size_t) y * os + 4 * x, in + (size_t) sy * is + 4 * sx, 4);
      }
    }
    return 0;
  }
  if (ow == w && oh == h) { for (int y = 0; y < h; y++) memcpy (out + (size_t) y * os, in + (size_t) y * is, 4 * (size_t) w); return 0; }
  if (ow == w) { vscale (in, is, w, h, out, os, oh); return 0; }
  if (oh == h) { hscale (in, is, w, h, out, os, ow); return 0; }
  if (h > oh + 2) {                 /* vertical first */
    uint8_t *tmp = malloc ((size_t) oh * w * 4); if (!tmp) return -2;
    vscale (in, is, w, h, tmp, w * 4, oh);
    hscale (tmp, w * 4, w, oh, out, os, ow);
    free (tmp);
  } else {                          /* horizontal first */
    uint8_t *tmp = malloc ((size_t) h * ow * 4); if (!tmp) return -2;
    hscale (in, is, w, h, tmp, ow * 4, ow);
    vscale (tmp, ow * 4, ow, h, out, os, oh);
    free (tmp);
  }
  return 0;
}

int gst114_default_matrix (int height) { return height >= 2160 ? GST114_BT2020 : height > 576 ? GST114_BT709 : GST114_BT601; }
int gst114_default_cosited (int height) { return height > 576; }

int gst114_convertscale_yuv420 (const uint8_t *yp, int ys, const uint8_t *up, int us, const uint8_t *vp, int vs,
    int planar, int w, int h, int matrix, int cosited, int out_format, int method,
    uint8_t *out, int os, int ow, int oh)
{
  uint8_t *full = malloc ((size_t) w * h * 4); if (!full) return -2;
  int rc = gst114_yuv420_to_rgb (yp, ys, up, us, vp, vs, planar, w, h, matrix, cosited, out_format, full, w * 4);
  if (rc == 0) rc = gst114_scale_4u8 (full, w * 4, w, h, out, os, ow, oh, method);
  free (full);
  return rc;
}
