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

/* packed 4:2:2 (UYVY: U Y0 V Y1, YUY2: Y0 U Y1 V) -> packed 8-bit RGB: the chroma row is up-sampled horizontally with the
 * same rule as NV12's (co-sited / 3:1), there is no vertical step, then the ORC matrix.  Pinned by probing videoconvert
 * (chroma ramps under chroma-site none / jpeg / mpeg2) and by tests/golden/convertscale_gst114_packed.npz. */
int gst114_packed422_to_rgb (const uint8_t *in, int is, int yuy2, int w, int h, int matrix, int cosited, int out_format, uint8_t *out, int os)
{
  if (w <= 0 || h <= 0 || matrix < 0 || matrix > 2) return -1;
  const int cw = (w + 1) / 2, yo = yuy2 ? 0 : 1, uo = yuy2 ? 1 : 0, vo = yuy2 ? 3 : 2;
  const int ro = out_format == GST114_RGBA ? 0 : 2, bo = 2 - ro;
  int oom = 0;                                          /* a failed per-row scratch allocation: the row is skipped, the call fails (-2), nothing is dereferenced */
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    const uint8_t *row = in + (size_t) y * is;
    uint8_t *hu = malloc ((size_t) w), *hv = malloc ((size_t) w);
    if (!hu || !hv) {
      free (hu); free (hv);
#pragma omp atomic write
      oom = 1;
      continue;
    }
    upsample_h (row + uo, 4, cw, w, cosited, hu);
    upsample_h (row + vo, 4, cw, w, cosited, hv);
    uint8_t *o = out + (size_t) y * os;
    for (int x = 0; x < w; x++) {
      int r, g, b;
      gst114_yuv_to_rgb (matrix, row[2 * x + yo], hu[x], hv[x], &r, &g, &b);
      o[4 * x + ro] = (uint8_t) r; o[4 * x + 1] = (uint8_t) g; o[4 * x + bo] = (uint8_t) b; o[4 * x + 3] = 255;
    }
    free (hu); free (hv);
  }
  return oom ? -2 : 0;
}

int gst114_convertscale_packed422 (const uint8_t *in, int is, int yuy2, int w, int h, int matrix, int cosited, int out_format, int method,
    uint8_t *out, int os, int ow, int oh)
{
  uint8_t *full = malloc ((size_t) w * h * 4); if (!full) return -2;
  int rc = gst114_packed422_to_rgb (in, is, yuy2, w, h, matrix, cosited, out_format, full, w * 4);
  if (rc == 0) rc = gst114_scale_4u8 (full, w * 4, w, h, out, os, ow, oh, method);
  free (full);
  return rc;
}

/* ---- videoscale, 4 x u8 ---------------------------------------------------------------- */

/* GstVideoResampler's 2-tap linear set-up, quantised the way GstVideoScaler does it: position x = ((j+.5)/out)*in - .5 in
 * IEEE double IN THAT ORDER, clamped to [0, in-1]; weights 1-|x-i| normalised; integer taps floor(off + w * 2^prec) with a
 * bisection on `off` (from .5, at most 64 steps) until they sum to 2^prec.  The bisection is what decides exact .5 ties:
 * there is no `off` that splits them, the search ends a few ulps below .5 and the LARGER tap rounds up through double
 * rounding (probed on 67 -> 64, 21 -> 448, 67 -> 256 ...: plain round-half-up is wrong on every tie). */
void gst114_linear_taps (int in, int out, int j, int prec, int *i0, int *i1, int *t0, int *t1)
{
  double x = ((j + 0.5) / out) * in - 0.5;
  x = x < 0.0 ? 0.0 : (x > in - 1.0 ? in - 1.0 : x);
  const int xi = (int) __builtin_floor (x);
  double w0 = 1.0 - __builtin_fabs (x - xi), w1 = 1.0 - __builtin_fabs (x - (xi + 1));
  if (w0 < 0.0) w0 = 0.0;
  if (w1 < 0.0) w1 = 0.0;
  const double sum = w0 + w1, m0 = w0 / sum, m1 = w1 / sum, mul = (double) (1 << prec);
  double lo = 0.0, hi = 1.0, off = 0.5;
  int a = 0, b = 0;
  for (int it = 0; it < 64; it++) {
    a = (int) __builtin_floor (off + m0 * mul); b = (int) __builtin_floor (off + m1 * mul);
    if (a + b == (1 << prec)) break;
    if (lo == hi) break;
    if (a + b < (1 << prec)) { if (off > lo) lo = off; off += (hi - lo) / 2; }
    else { if (off < hi) hi = off; off -= (hi - lo) / 2; }
  }
  *i0 = clampi (xi, 0, in - 1); *i1 = clampi (xi + 1, 0, in - 1); *t0 = a; *t1 = b;
}

void gst114_vtaps (int in_h, int out_h, int y, int *i0, int *i1, int *w)
{
  /* centre-aligned; 8-bit taps; the line function uses the second tap only: s1 + (((s2 - s1) * w + 128) >> 8) */
  int t0;
  gst114_linear_taps (in_h, out_h, y, 8, i0, i1, &t0, w);
}

uint32_t gst114_hinc (int in_w, int out_w)
{
  if (out_w <= 1 || in_w <= 1) return 0;           /* a one-sample line is replicated (the formula below would wrap to -1) */
  return (uint32_t) ((((uint64_t) (in_w - 1)) << 16) / (uint64_t) (out_w - 1)) - 1;
}

int gst114_scale_4u8 (const uint8_t *in, int is, int w, int h, uint8_t *out, int os, int ow, int oh, int method);

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

/* ---- videoscale method=catrom ("bicubic"): GstVideoResampler's cubic set-up + GstVideoScaler's 6-bit n-tap path.
 * Not in the reference (it has no bicubic, gstvfmetalconvertscale.m:81-85) but named by north_star; pinned by probing the
 * real element (impulse responses give every 6-bit tap exactly; random frames confirm arithmetic and pass order;
 * tests/golden/convertscale_gst114_bicubic.npz):
 *   scale = in / out; fx = min (1, 1 / scale); n_taps = ceil (4 / fx); fx = 4 / n_taps        (envelope 2)
 *   x = clamp (((j + .5) / out) * in - .5, 0, in - 1) in exactly this order; first tap at floor (x) - (n_taps - 1) / 2
 *   w_l = k ((x - x_l) * fx), Mitchell-Netravali form with b = 0, c = .5, normalised by their sum; taps that fall outside
 *   the line are ADDED to the edge tap in double precision; then t_l = floor (offset + 64 w_l) with the offset found by
 *   bisection from .5 so that the taps sum to 64 (<= 64 steps);
 *   each pass: out = clamp ((sum p_l t_l + 32) >> 6, 0, 255) on u8 lines; vertical pass first iff in_h > out_h + n_taps_v. */
static double cubic_k (double a)
{
  const double b = 0.0, c = 0.5;
  a = a < 0 ? -a : a;
  const double a2 = a * a, a3 = a2 * a;
  if (a <= 1.0) return ((12.0 - 9.0 * b - 6.0 * c) * a3 + (-18.0 + 12.0 * b + 6.0 * c) * a2 + (6.0 - 2.0 * b)) / 6.0;
  if (a <= 2.0) return ((-b - 6.0 * c) * a3 + (6.0 * b + 30.0 * c) * a2 + (-12.0 * b - 48.0 * c) * a + (8.0 * b + 24.0 * c)) / 6.0;
  return 0.0;
}

static int cubic_n_taps (int in, int out)
{
  const double scale = (double) in / (double) out;
  const double fx = scale > 1.0 ? 1.0 / scale : 1.0;
  return (int) __builtin_ceil (2.0 * 2.0 / fx);
}

/* n_taps of the LINEAR method without a tap limit (what videoscale's catrom leaves the chroma planes of a planar frame
 * with): envelope 1 -> ceil (2 / fx) */
static int linear_n_taps (int in, int out)
{
  const double scale = (double) in / (double) out;
  const double fx = scale > 1.0 ? 1.0 / scale : 1.0;
  return (int) __builtin_ceil (2.0 * 1.0 / fx);
}

/* pinned domain: the line is at least as long as the filter (n_taps <= in) and n_taps <= 64; GstVideoResampler's handling
 * of shorter lines (it truncates the filter) is not restated.  kernel: 0 = cubic (catrom), 1 = linear (1 - |a|). */
static int resampler_taps (int kernel, int in, int out, int *idx, int *taps, int max_entries)
{
  const int n = kernel ? linear_n_taps (in, out) : cubic_n_taps (in, out);
  if (n > 64 || n > in || (long) n * out > max_entries) return -1;
  const double fx = (kernel ? 2.0 * 1.0 : 2.0 * 2.0) / n;
  for (int j = 0; j < out; j++) {
    double x = ((j + 0.5) / out) * in - 0.5;        /* this order: the quotient first (it decides which way exact .5 ties fall) */
    x = x < 0.0 ? 0.0 : (x > in - 1.0 ? in - 1.0 : x);
    const int xi = (int) __builtin_floor (x) - (n - 1) / 2;
    double w[64], sum = 0.0, m[64];
    int pos[64], cnt = 0;
    for (int l = 0; l < n; l++) {
      const double a = (x - (xi + l)) * fx;
      if (kernel) { const double aa = a < 0 ? -a : a; w[l] = aa < 1.0 ? 1.0 - aa : 0.0; }
      else w[l] = cubic_k (a);
      sum += w[l];
    }
    for (int l = 0; l < n; l++) {                    /* merge clamped taps (in tap order: left edge first) */
      const int k = clampi (xi + l, 0, in - 1);
      if (cnt > 0 && pos[cnt - 1] == k) m[cnt - 1] += w[l] / sum;
      else { pos[cnt] = k; m[cnt] = w[l] / sum; cnt++; }
    }
    double lo = 0.0, hi = 1.0, off = 0.5;
    for (int it = 0; it < 64; it++) {
      int s = 0;
      for (int l = 0; l < cnt; l++) s += (int) __builtin_floor (off + m[l] * 64.0);
      if (s == 64) break;
      if (lo == hi) break;
      if (s < 64) { if (off > lo) lo = off; off += (hi - lo) / 2; }
      else { if (off < hi) hi = off; off -= (hi - lo) / 2; }
    }
    for (int l = 0; l < n; l++) {
      idx[j * n + l] = l < cnt ? pos[l] : pos[cnt - 1];
      taps[j * n + l] = l < cnt ? (int) __builtin_floor (off + m[l] * 64.0) : 0;
    }
  }
  return n;
}

int gst114_cubic_taps (int in, int out, int *idx, int *taps, int max_entries) { return resampler_taps (0, in, out, idx, taps, max_entries); }
int gst114_linear_ntaps (int in, int out, int *idx, int *taps, int max_entries) { return resampler_taps (1, in, out, idx, taps, max_entries); }

static int cubic_pass (const uint8_t *in, int is, int w, int h, uint8_t *out, int os, int on, int vertical)
{
  const int len = vertical ? h : w;
  const int n = cubic_n_taps (len, on);
  int *idx = malloc (sizeof (int) * (size_t) n * on), *tp = malloc (sizeof (int) * (size_t) n * on);
  if (!idx || !tp || gst114_cubic_taps (len, on, idx, tp, n * on) < 0) { free (idx); free (tp); return -2; }
  if (vertical) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < on; y++)
      for (int k = 0; k < 4 * w; k++) {
        int acc = 0;
        for (int l = 0; l < n; l++) acc += in[(size_t) idx[y * n + l] * is + k] * tp[y * n + l];
        out[(size_t) y * os + k] = (uint8_t) clampi ((acc + 32) >> 6, 0, 255);
      }
  } else {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
      for (int x = 0; x < on; x++)
        for (int c = 0; c < 4; c++) {
          int acc = 0;
          for (int l = 0; l < n; l++) acc += in[(size_t) y * is + 4 * idx[x * n + l] + c] * tp[x * n + l];
          out[(size_t) y * os + 4 * x + c] = (uint8_t) clampi ((acc + 32) >> 6, 0, 255);
        }
  }
  free (idx); free (tp);
  return 0;
}

static int scale_cubic_4u8 (const uint8_t *in, int is, int w, int h, uint8_t *out, int os, int ow, int oh)
{
  if ((ow != w && (cubic_n_taps (w, ow) > w || cubic_n_taps (w, ow) > 64)) || (oh != h && (cubic_n_taps (h, oh) > h || cubic_n_taps (h, oh) > 64)))
    return -3;                                         /* outside the pinned domain */
  if (ow == w && oh == h) { for (int y = 0; y < h; y++) memcpy (out + (size_t) y * os, in + (size_t) y * is, 4 * (size_t) w); return 0; }
  if (ow == w) return cubic_pass (in, is, w, h, out, os, oh, 1);
  if (oh == h) return cubic_pass (in, is, w, h, out, os, ow, 0);
  int rc;
  if (h > oh + cubic_n_taps (h, oh)) {                /* vertical first */
    uint8_t *tmp = malloc ((size_t) oh * w * 4); if (!tmp) return -2;
    rc = cubic_pass (in, is, w, h, tmp, w * 4, oh, 1);
    if (!rc) rc = cubic_pass (tmp, w * 4, w, oh, out, os, ow, 0);
    free (tmp);
  } else {
    uint8_t *tmp = malloc ((size_t) h * ow * 4); if (!tmp) return -2;
    rc = cubic_pass (in, is, w, h, tmp, ow * 4, ow, 0);
    if (!rc) rc = cubic_pass (tmp, ow * 4, ow, h, out, os, oh, 1);
    free (tmp);
  }
  return rc;
}

int gst114_scale_4u8 (const uint8_t *in, int is, int w, int h, uint8_t *out, int os, int ow, int oh, int method)
{
  if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0) return -1;
  if (method == GST114_BICUBIC) return scale_cubic_4u8 (in, is, w, h, out, os, ow, oh);
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


/* ---- RGB -> YUV 4:2:0 (videoconvert BGRA/RGBA -> NV12/I420), pinned by probing the real element:
 *   Y = ((cy . rgb) >> 8) + 16, U = ((cu . rgb) >> 8) + 128, V likewise, arithmetic shifts, 8-bit coefficients;
 *   4:4:4 -> 4:2:0: vertical (a+b+1)>>1 first (an odd last row pairs with itself), then horizontal:
 *   not co-sited: (a+b+1)>>1 (an odd last column pairs with itself); h-co-sited (mpeg2): (l + 2c + r + 2)>>2 with the
 *   left tap of the first sample AND the right tap of the LAST sample replaced by the centre (GStreamer quirk: the last
 *   sample ignores its right neighbour even when it exists — except on a two-sample line, where the only chroma sample is
 *   first and last and does use it). */
static const int RGB2YUV[3][9] = {
  {  66, 129,  25,  -38,  -74, 112,  112,  -94, -18 },   /* bt601  */
  {  47, 157,  16,  -26,  -87, 112,  112, -102, -10 },   /* bt709  */
  {  58, 149,  13,  -31,  -81, 112,  112, -103,  -9 },   /* bt2020 */
};

int gst114_rgb_to_yuv420 (const uint8_t *in, int is, int in_format, int w, int h, int matrix, int cosited,
    int planar, uint8_t *yp, int ys, uint8_t *up, int us, uint8_t *vp, int vs)
{
  if (w <= 0 || h <= 0 || matrix < 0 || matrix > 2) return -1;
  const int *c = RGB2YUV[matrix];
  const int ro = in_format == GST114_RGBA ? 0 : 2, bo = 2 - ro;
  const int cw = (w + 1) / 2, ch = (h + 1) / 2;
  int *vu = malloc ((size_t) ch * w * sizeof (int)), *vv = malloc ((size_t) ch * w * sizeof (int));
  if (!vu || !vv) { free (vu); free (vv); return -2; }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < ch; j++) {
    for (int x = 0; x < w; x++) {
      int su = 1, sv = 1;
      for (int d = 0; d < 2; d++) {
        const int y = 2 * j + d < h ? 2 * j + d : h - 1;
        const uint8_t *px = in + (size_t) y * is + 4 * x;
        const int r = px[ro], g = px[1], b = px[bo];
        if (2 * j + d < h) yp[(size_t) y * ys + x] = (uint8_t) (((c[0] * r + c[1] * g + c[2] * b) >> 8) + 16);
        su += ((c[3] * r + c[4] * g + c[5] * b) >> 8) + 128;
        sv += ((c[6] * r + c[7] * g + c[8] * b) >> 8) + 128;
      }
      vu[(size_t) j * w + x] = su >> 1; vv[(size_t) j * w + x] = sv >> 1;
    }
  }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < ch; j++) {
    const int *ru = vu + (size_t) j * w, *rv = vv + (size_t) j * w;
    for (int k = 0; k < cw; k++) {
      int U, V;
      const int x = 2 * k;
      if (cosited) {
        const int l = x > 0 ? x - 1 : 0, r = (k == cw - 1 && k > 0) ? x : (x + 1 < w ? x + 1 : w - 1);
        U = (ru[l] + 2 * ru[x] + ru[r] + 2) >> 2; V = (rv[l] + 2 * rv[x] + rv[r] + 2) >> 2;
      } else {
        const int r = x + 1 < w ? x + 1 : w - 1;
        U = (ru[x] + ru[r] + 1) >> 1; V = (rv[x] + rv[r] + 1) >> 1;
      }
      if (planar) { up[(size_t) j * us + k] = (uint8_t) U; vp[(size_t) j * vs + k] = (uint8_t) V; }
      else { up[(size_t) j * us + 2 * k] = (uint8_t) U; up[(size_t) j * us + 2 * k + 1] = (uint8_t) V; }
    }
  }
  free (vu); free (vv);
  return 0;
}

static void down_h_row (const int *c, int w, int cosited, int *out);

/* ---- videoconvert YUV -> YUV when the colour matrix and / or the chroma siting change (NV12 / I420 / UYVY / YUY2) ------------
 * Pinned by probing the real element (flat-colour cubes for the matrix, random frames of every parity for the resampling;
 * tests/golden/convertscale_gst114_remat.npz):
 *   matrix: GstVideoConverter's 8-bit path, out = clamp8 (((a Y + b U + c V) >> 8) + d) per sample with the integer rows
 *     below (coefficient = rint (256 m) of the combined limited-range matrix, d = floor of its offset);
 *   same siting on both sides: no chroma resampling at all — each luma sample is matrixed with its nearest-replicated
 *     chroma sample, each output chroma sample is the matrixed input chroma sample (the luma coefficient of the chroma rows is 0);
 *   siting change: chroma is up-sampled to 4:4:4 with the INPUT siting (horizontal, then vertical 3:1 — over an EVEN number
 *     of lines: an odd frame's phantom last line is filtered too), matrixed per sample (or left alone when only the siting
 *     changes: NV12 <-> I420), and down-sampled with the OUTPUT siting (vertical pair average, then horizontal). */
static const int YUV2YUV[3][3][12] = {           /* [matrix in][matrix out][row (Y, U, V) x (a, b, c, d)] */
  { { 0 }, { 256, -30, -53, 41,   0, 261, 29, -18,   0, 19, 262, -13 }, { 256, -32, -29, 30,   0, 259, 16, -10,   0, 22, 264, -15 } },   /* bt601 -> */
  { { 256, 25, 49, -38,   0, 253, -28, 15,   0, -19, 252, 11 }, { 0 }, { 256, -4, 24, -10,   0, 255, -13, 7,   0, 3, 257, -2 } },        /* bt709 -> */
  { { 256, 30, 26, -28,   0, 255, -15, 8,   0, -22, 250, 13 }, { 256, 5, -24, 9,   0, 257, 13, -8,   0, -3, 255, 1 }, { 0 } },           /* bt2020 -> */
};
static inline int mat8 (const int *r, int y, int u, int v) { return clampi (((r[0] * y + r[1] * u + r[2] * v) >> 8) + r[3], 0, 255); }

/* Any of NV12 / I420 / UYVY / YUY2 on either side, one size.  Sample addressing: luma sample x of row y at yp[y * ys + x * ystep],
 * chroma sample k of chroma row j at up / vp[j * cs + k * cstep] (NV12: cstep 2; I420: 1; packed: ystep 2, cstep 4, cs == ys, chroma row
 * == luma row); `in420` / `out420` say whether chroma is vertically subsampled.
 * The same rules hold for the packed formats (probed: 45 random packed <-> packed / 4:2:0 frames with a matrix change, 0 differing
 * bytes): "same siting, no resampling" needs the same subsampling on both sides as well; everything else takes the up-sample ->
 * matrix -> down-sample path, 4:2:2 chroma having no vertical step. */
int gst114_yuv_to_yuv (const uint8_t *yp, int ys, int ystep, const uint8_t *up, const uint8_t *vp, int cs, int cstep, int in420, int w, int h,
    int matrix_in, int cosited_in, int matrix_out, int cosited_out,
    uint8_t *oy, int oys, int oystep, uint8_t *ou, uint8_t *ov, int ocs, int ocstep, int out420)
{
  if (w <= 0 || h <= 0 || matrix_in < 0 || matrix_in > 2 || matrix_out < 0 || matrix_out > 2) return -1;
  const int cw = (w + 1) / 2, ch = in420 ? (h + 1) / 2 : h, och = out420 ? (h + 1) / 2 : h;
  const int hp = in420 ? 2 * ch : (out420 ? 2 * och : h);       /* full-resolution chroma lines, an even number when either side is 4:2:0 */
  const int remat = matrix_in != matrix_out;
  const int *t = YUV2YUV[matrix_in][matrix_out];
  if (cosited_in == cosited_out && !in420 == !out420) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        const int j = in420 ? y >> 1 : y;
        const int Y = yp[(size_t) y * ys + x * ystep], U = up[(size_t) j * cs + (x >> 1) * cstep], V = vp[(size_t) j * cs + (x >> 1) * cstep];
        oy[(size_t) y * oys + x * oystep] = (uint8_t) (remat ? mat8 (t, Y, U, V) : Y);
      }
    for (int j = 0; j < ch; j++)
      for (int k = 0; k < cw; k++) {
        const int U = up[(size_t) j * cs + k * cstep], V = vp[(size_t) j * cs + k * cstep];
        ou[(size_t) j * ocs + k * ocstep] = (uint8_t) (remat ? mat8 (t + 4, 0, U, V) : U);
        ov[(size_t) j * ocs + k * ocstep] = (uint8_t) (remat ? mat8 (t + 8, 0, U, V) : V);
      }
    return 0;
  }
  uint8_t *hu = malloc ((size_t) ch * w), *hv = malloc ((size_t) ch * w);
  int *fu = malloc ((size_t) hp * w * sizeof (int)), *fv = malloc ((size_t) hp * w * sizeof (int));
  int *du = malloc ((size_t) cw * sizeof (int)), *dv = malloc ((size_t) cw * sizeof (int)), *vu = malloc ((size_t) w * sizeof (int)), *vv = malloc ((size_t) w * sizeof (int));
  if (!hu || !hv || !fu || !fv || !du || !dv || !vu || !vv) { free (hu); free (hv); free (fu); free (fv); free (du); free (dv); free (vu); free (vv); return -2; }
  for (int j = 0; j < ch; j++) {
    upsample_h (up + (size_t) j * cs, cstep, cw, w, cosited_in, hu + (size_t) j * w);
    upsample_h (vp + (size_t) j * cs, cstep, cw, w, cosited_in, hv + (size_t) j * w);
  }
  for (int y = 0; y < hp; y++) {                       /* 4:4:4 chroma of every line, the phantom line of an odd height included */
    const int yl = y < h ? y : h - 1;
    for (int x = 0; x < w; x++) {
      int U, V;
      if (in420) {
        const int j = y >> 1, jn = (y & 1) ? clampi (j + 1, 0, ch - 1) : clampi (j - 1, 0, ch - 1);
        U = (3 * hu[(size_t) j * w + x] + hu[(size_t) jn * w + x] + 2) >> 2; V = (3 * hv[(size_t) j * w + x] + hv[(size_t) jn * w + x] + 2) >> 2;
      } else { U = hu[(size_t) yl * w + x]; V = hv[(size_t) yl * w + x]; }
      const int Y = yp[(size_t) yl * ys + x * ystep];
      if (y < h) oy[(size_t) y * oys + x * oystep] = (uint8_t) (remat ? mat8 (t, Y, U, V) : Y);
      fu[(size_t) y * w + x] = remat ? mat8 (t + 4, 0, U, V) : U;
      fv[(size_t) y * w + x] = remat ? mat8 (t + 8, 0, U, V) : V;
    }
  }
  for (int j = 0; j < och; j++) {
    for (int x = 0; x < w; x++) {
      if (out420) {
        vu[x] = (fu[(size_t) (2 * j) * w + x] + fu[(size_t) (2 * j + 1) * w + x] + 1) >> 1;
        vv[x] = (fv[(size_t) (2 * j) * w + x] + fv[(size_t) (2 * j + 1) * w + x] + 1) >> 1;
      } else { vu[x] = fu[(size_t) j * w + x]; vv[x] = fv[(size_t) j * w + x]; }
    }
    down_h_row (vu, w, cosited_out, du); down_h_row (vv, w, cosited_out, dv);
    for (int k = 0; k < cw; k++) { ou[(size_t) j * ocs + k * ocstep] = (uint8_t) du[k]; ov[(size_t) j * ocs + k * ocstep] = (uint8_t) dv[k]; }
  }
  free (hu); free (hv); free (fu); free (fv); free (du); free (dv); free (vu); free (vv);
  return 0;
}

/* ---- videoscale on planar 8-bit data (NV12 / I420 -> same format), pinned by probing the real element:
 *   vertical: the same 8-bit centre-aligned 2-tap as for 4 x u8;
 *   horizontal, 1 x u8 planes (Y, I420 chroma): the edge-aligned 16.16 path of the 4 x u8 case, EXCEPT when the plane is
 *   exactly halved horizontally and vertically unchanged or exactly halved: then pairs average, (a+b+1)>>1;
 *   horizontal, 2 x u8 plane (NV12 chroma): centre-aligned with 6-bit taps: t = round(frac*64), (a(64-t) + b t + 32)>>6;
 *   pass order per plane: vertical first iff plane_in_h > plane_out_h + 2. */
/* centre-aligned 2-tap with 6-bit taps (GstVideoScaler's generic n-tap u8 line function at 2 taps) */
static void hscale_centre6 (const uint8_t *in, int is, int w, int h, int n, uint8_t *out, int os, int ow)
{
  for (int x = 0; x < ow; x++) {
    int i0, i1, t0, t1;
    gst114_linear_taps (w, ow, x, 6, &i0, &i1, &t0, &t1);
    for (int y = 0; y < h; y++)
      for (int c = 0; c < n; c++)
        out[(size_t) y * os + n * x + c] = (uint8_t) clampi ((in[(size_t) y * is + n * i0 + c] * t0 + in[(size_t) y * is + n * i1 + c] * t1 + 32) >> 6, 0, 255);
  }
}

static void hscale_plane (const uint8_t *in, int is, int w, int h, int n, uint8_t *out, int os, int ow, int half)
{
  if (n == 1 && half) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < ow; x++) out[(size_t) y * os + x] = (uint8_t) ((in[(size_t) y * is + 2 * x] + in[(size_t) y * is + 2 * x + 1] + 1) >> 1);
    return;
  }
  if (n == 1) {
    const uint32_t inc = gst114_hinc (w, ow);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < ow; x++) {
        const uint32_t t = (uint32_t) x * inc; const int i = t >> 16, f = (t >> 8) & 0xff, i1 = i + 1 < w ? i + 1 : w - 1;
        out[(size_t) y * os + x] = (uint8_t) ((in[(size_t) y * is + i] * (256 - f) + in[(size_t) y * is + i1] * f) >> 8);
      }
    return;
  }
  hscale_centre6 (in, is, w, h, n, out, os, ow);
}

static void vscale_plane (const uint8_t *in, int is, int wb, int h, uint8_t *out, int os, int oh)
{
  for (int y = 0; y < oh; y++) {
    int i0, i1, wt; gst114_vtaps (h, oh, y, &i0, &i1, &wt);
    const uint8_t *s1 = in + (size_t) i0 * is, *s2 = in + (size_t) i1 * is;
    for (int k = 0; k < wb; k++) out[(size_t) y * os + k] = (uint8_t) (s1[k] + ((((int) s2[k] - (int) s1[k]) * wt + 128) >> 8));
  }
}

/* one plane of `n` interleaved u8 components, w x h samples -> ow x oh samples */
int gst114_scale_plane (const uint8_t *in, int is, int w, int h, int n, uint8_t *out, int os, int ow, int oh)
{
  if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0 || (n != 1 && n != 2)) return -1;
  const int half = (w == 2 * ow) && (oh == h || h == 2 * oh);
  if (ow == w && oh == h) { for (int y = 0; y < h; y++) memcpy (out + (size_t) y * os, in + (size_t) y * is, (size_t) n * w); return 0; }
  if (ow == w) { vscale_plane (in, is, n * w, h, out, os, oh); return 0; }
  if (oh == h) { hscale_plane (in, is, w, h, n, out, os, ow, half); return 0; }
  if (h > oh + 2) {
    uint8_t *tmp = malloc ((size_t) oh * w * n); if (!tmp) return -2;
    vscale_plane (in, is, n * w, h, tmp, n * w, oh);
    hscale_plane (tmp, n * w, w, oh, n, out, os, ow, half);
    free (tmp);
  } else {
    uint8_t *tmp = malloc ((size_t) h * ow * n); if (!tmp) return -2;
    hscale_plane (in, is, w, h, n, tmp, n * ow, ow, half);
    vscale_plane (tmp, n * ow, n * ow, h, out, os, oh);
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

/* ---- packed 4:2:2 OUTPUTS and packed -> 4:2:0 (videoconvert at the input size, then videoscale on the packed frame),
 * pinned by probing the real elements (tools/gen_goldens.py packedout, tests/golden/convertscale_gst114_packedout.npz):
 *   RGB -> packed: the RGB -> YUV matrix of the 4:2:0 path and only the HORIZONTAL half of its chroma step (output siting);
 *   I420 -> packed: fast path, Y copied, chroma row y>>1 copied (no interpolation at all);
 *   NV12 -> packed: generic path, chroma up-sampled to 4:4:4 (horizontal rule of the INPUT siting, then the 3:1 vertical
 *     rule), then the horizontal down-sampling rule of the OUTPUT siting (note: videoconvert copies colorimetry AND
 *     chroma-site from its input when the output caps carry no colorimetry);
 *   UYVY <-> YUY2: byte swizzle;
 *   packed -> I420: fast path, Y copied, chroma rows averaged in pairs (a+b+1)>>1 (an odd last row pairs with itself);
 *   packed -> NV12: generic path, horizontal up-sample (input siting), vertical pair average, horizontal down-sample
 *     (output siting).
 *   videoscale on a packed frame (2-tap): Y (w samples), U and V ((w+1)/2 samples each) are three interleaved lines
 *   scaled like NV12's chroma plane: centre-aligned 6-bit taps horizontally, the 8-bit 2-tap vertically over every byte,
 *   vertical first iff in_h > out_h + 2.  GStreamer bug NOT restated: when the vertical pass runs first AND a horizontal
 *   pass follows, it covers 2*w bytes only, so for an ODD width the last V sample of the intermediate line is never
 *   written and the V outputs that tap it carry whatever the temporary line held (0 after a bare videoscale, garbage
 *   after videoconvert).  This restatement uses the properly scaled sample; comparisons with the real element skip
 *   those V outputs (tests/test_oracle_golden.py: undefined_v_mask). */
static inline void pk_offsets (int yuy2, int *yo, int *uo, int *vo) { *yo = yuy2 ? 0 : 1; *uo = yuy2 ? 1 : 0; *vo = yuy2 ? 3 : 2; }

static void down_h_row (const int *c, int w, int cosited, int *out)
{
  const int cw = (w + 1) / 2;
  for (int k = 0; k < cw; k++) {
    const int x = 2 * k;
    if (cosited) {
      const int l = x > 0 ? x - 1 : 0, r = (k == cw - 1 && k > 0) ? x : (x + 1 < w ? x + 1 : w - 1);
      out[k] = (c[l] + 2 * c[x] + c[r] + 2) >> 2;
    } else {
      const int r = x + 1 < w ? x + 1 : w - 1;
      out[k] = (c[x] + c[r] + 1) >> 1;
    }
  }
}

/* Y row + cw-sample U / V rows -> one packed row; an odd width's spare luma slot repeats the last luma sample */
static void pk_store_row (uint8_t *o, int yuy2, int w, const uint8_t *y, const int *u, const int *v)
{
  int yo, uo, vo; pk_offsets (yuy2, &yo, &uo, &vo);
  const int cw = (w + 1) / 2;
  for (int k = 0; k < cw; k++) {
    o[4 * k + yo] = y[2 * k];
    o[4 * k + yo + 2] = y[2 * k + 1 < w ? 2 * k + 1 : w - 1];
    o[4 * k + uo] = (uint8_t) u[k]; o[4 * k + vo] = (uint8_t) v[k];
  }
}

int gst114_rgb_to_packed422 (const uint8_t *in, int is, int in_format, int w, int h, int matrix, int cosited, int yuy2, uint8_t *out, int os)
{
  if (w <= 0 || h <= 0 || matrix < 0 || matrix > 2) return -1;
  const int *c = RGB2YUV[matrix];
  const int ro = in_format == GST114_RGBA ? 0 : 2, bo = 2 - ro, cw = (w + 1) / 2;
  int oom = 0;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    uint8_t *Y = malloc ((size_t) w);
    int *u = malloc (sizeof (int) * (size_t) (2 * w + 2 * cw));
    if (!Y || !u) {
      free (Y); free (u);
#pragma omp atomic write
      oom = 1;
      continue;
    }
    int *v = u + w, *du = v + w, *dv = du + cw;
    for (int x = 0; x < w; x++) {
      const uint8_t *px = in + (size_t) y * is + 4 * x;
      const int r = px[ro], g = px[1], b = px[bo];
      Y[x] = (uint8_t) (((c[0] * r + c[1] * g + c[2] * b) >> 8) + 16);
      u[x] = ((c[3] * r + c[4] * g + c[5] * b) >> 8) + 128;
      v[x] = ((c[6] * r + c[7] * g + c[8] * b) >> 8) + 128;
    }
    down_h_row (u, w, cosited, du); down_h_row (v, w, cosited, dv);
    pk_store_row (out + (size_t) y * os, yuy2, w, Y, du, dv);
    free (Y); free (u);
  }
  return oom ? -2 : 0;
}

int gst114_yuv420_to_packed422 (const uint8_t *yp, int ys, const uint8_t *up, int us, const uint8_t *vp, int vs,
    int planar, int w, int h, int cosited_in, int cosited_out, int yuy2, uint8_t *out, int os)
{
  if (w <= 0 || h <= 0) return -1;
  const int cw = (w + 1) / 2, ch = (h + 1) / 2;
  uint8_t *hu = NULL, *hv = NULL;
  if (!planar) {
    hu = malloc ((size_t) ch * w); hv = malloc ((size_t) ch * w);
    if (!hu || !hv) { free (hu); free (hv); return -2; }
    for (int j = 0; j < ch; j++) {
      upsample_h (up + (size_t) j * us, 2, cw, w, cosited_in, hu + (size_t) j * w);
      upsample_h (up + (size_t) j * us + 1, 2, cw, w, cosited_in, hv + (size_t) j * w);
    }
  }
  int oom = 0;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < h; y++) {
    const int j = y >> 1;
    int *u = malloc (sizeof (int) * (size_t) (2 * w + 2 * cw));
    if (!u) {
#pragma omp atomic write
      oom = 1;
      continue;
    }
    int *v = u + w, *du = v + w, *dv = du + cw;
    if (planar) {
      for (int k = 0; k < cw; k++) { du[k] = up[(size_t) j * us + k]; dv[k] = vp[(size_t) j * vs + k]; }
    } else {
      const int jn = (y & 1) ? clampi (j + 1, 0, ch - 1) : clampi (j - 1, 0, ch - 1);
      for (int x = 0; x < w; x++) {
        u[x] = (3 * hu[(size_t) j * w + x] + hu[(size_t) jn * w + x] + 2) >> 2;
        v[x] = (3 * hv[(size_t) j * w + x] + hv[(size_t) jn * w + x] + 2) >> 2;
      }
      down_h_row (u, w, cosited_out, du); down_h_row (v, w, cosited_out, dv);
    }
    pk_store_row (out + (size_t) y * os, yuy2, w, yp + (size_t) y * ys, du, dv);
    free (u);
  }
  free (hu); free (hv);
  return oom ? -2 : 0;
}

int gst114_packed422_swizzle (const uint8_t *in, int is, int in_yuy2, int w, int h, int out_yuy2, uint8_t *out, int os)
{
  if (w <= 0 || h <= 0) return -1;
  const int cw = (w + 1) / 2;
  for (int y = 0; y < h; y++)
    for (int k = 0; k < cw; k++) {
      const uint8_t *s = in + (size_t) y * is + 4 * k; uint8_t *d = out + (size_t) y * os + 4 * k;
      if (in_yuy2 == out_yuy2) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; }
      else { d[0] = s[1]; d[1] = s[0]; d[2] = s[3]; d[3] = s[2]; }
    }
  return 0;
}

int gst114_packed422_to_yuv420 (const uint8_t *in, int is, int yuy2, int w, int h, int cosited_in, int cosited_out, int planar,
    uint8_t *yp, int ys, uint8_t *up, int us, uint8_t *vp, int vs)
{
  if (w <= 0 || h <= 0) return -1;
  int yo, uo, vo; pk_offsets (yuy2, &yo, &uo, &vo);
  const int cw = (w + 1) / 2, ch = (h + 1) / 2;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) yp[(size_t) y * ys + x] = in[(size_t) y * is + 2 * x + yo];
  int oom = 0;
#pragma omp parallel for schedule(static)
  for (int j = 0; j < ch; j++) {
    const uint8_t *r0 = in + (size_t) (2 * j) * is, *r1 = in + (size_t) (2 * j + 1 < h ? 2 * j + 1 : h - 1) * is;
    if (planar) {
      for (int k = 0; k < cw; k++) {
        up[(size_t) j * us + k] = (uint8_t) ((r0[4 * k + uo] + r1[4 * k + uo] + 1) >> 1);
        vp[(size_t) j * vs + k] = (uint8_t) ((r0[4 * k + vo] + r1[4 * k + vo] + 1) >> 1);
      }
    } else {
      uint8_t *a = malloc ((size_t) 4 * w);
      int *u = malloc (sizeof (int) * (size_t) (2 * w + 2 * cw));
      if (!a || !u) {
        free (a); free (u);
#pragma omp atomic write
      oom = 1;
      continue;
    }
      uint8_t *b = a + w, *c = b + w, *d = c + w;
      int *v = u + w, *du = v + w, *dv = du + cw;
      upsample_h (r0 + uo, 4, cw, w, cosited_in, a); upsample_h (r1 + uo, 4, cw, w, cosited_in, b);
      upsample_h (r0 + vo, 4, cw, w, cosited_in, c); upsample_h (r1 + vo, 4, cw, w, cosited_in, d);
      for (int x = 0; x < w; x++) { u[x] = (a[x] + b[x] + 1) >> 1; v[x] = (c[x] + d[x] + 1) >> 1; }
      down_h_row (u, w, cosited_out, du); down_h_row (v, w, cosited_out, dv);
      for (int k = 0; k < cw; k++) { up[(size_t) j * us + 2 * k] = (uint8_t) du[k]; up[(size_t) j * us + 2 * k + 1] = (uint8_t) dv[k]; }
      free (a); free (u);
    }
  }
  return oom ? -2 : 0;
}

/* videoscale method=bilinear on a packed 4:2:2 frame */
int gst114_scale_packed422 (const uint8_t *in, int is, int yuy2, int w, int h, uint8_t *out, int os, int ow, int oh)
{
  if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0) return -1;
  int yo, uo, vo; pk_offsets (yuy2, &yo, &uo, &vo);
  const int cw = (w + 1) / 2, cow = (ow + 1) / 2;
  const int vfirst = h > oh + 2, hs = ow != w, vs_ = oh != h;
  /* three planes of ints, processed in the pinned order */
  const int mh = h > oh ? h : oh, mw = w > ow ? w : ow;
  uint8_t *A = malloc ((size_t) 3 * mh * mw), *B = malloc ((size_t) 3 * mh * mw);
  if (!A || !B) { free (A); free (B); return -2; }
  for (int p = 0; p < 3; p++) {
    const int pw = p ? cw : w, pow_ = p ? cow : ow, off = p == 0 ? yo : p == 1 ? uo : vo, step = p ? 4 : 2;
    uint8_t *a = A + (size_t) p * mh * mw, *b = B + (size_t) p * mh * mw;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < pw; x++) a[(size_t) y * mw + x] = in[(size_t) y * is + step * x + off];
    int cur_w = pw, cur_h = h;
    uint8_t *src = a, *dst = b;
    for (int pass = 0; pass < 2; pass++) {
      const int vertical = (pass == 0) == (vfirst != 0);
      if (vertical && vs_) {
        vscale_plane (src, mw, cur_w, cur_h, dst, mw, oh);
        cur_h = oh;
        uint8_t *t = src; src = dst; dst = t;
      } else if (!vertical && hs) {
        hscale_centre6 (src, mw, cur_w, cur_h, 1, dst, mw, pow_);
        cur_w = pow_;
        uint8_t *t = src; src = dst; dst = t;
      }
    }
    for (int y = 0; y < oh; y++)
      for (int x = 0; x < pow_; x++) out[(size_t) y * os + step * x + off] = src[(size_t) y * mw + x];
  }
  if (ow & 1)                                            /* spare luma slot of an odd width: GStreamer repeats the last luma sample */
    for (int y = 0; y < oh; y++) out[(size_t) y * os + 2 * ow + yo] = out[(size_t) y * os + 2 * (ow - 1) + yo];
  free (A); free (B);
  return 0;
}

/* ---- videoscale method=nearest-neighbour on YUV frames (pinned by probing: every plane / interleaved line takes the
 * nearest index of its own size, gst114_nearest_index) */
int gst114_scale_plane_nearest (const uint8_t *in, int is, int w, int h, int n, uint8_t *out, int os, int ow, int oh)
{
  if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0 || n < 1) return -1;
  for (int y = 0; y < oh; y++) {
    const uint8_t *row = in + (size_t) gst114_nearest_index (h, oh, y) * is;
    for (int x = 0; x < ow; x++) {
      const int i = gst114_nearest_index (w, ow, x);
      for (int c = 0; c < n; c++) out[(size_t) y * os + n * x + c] = row[n * i + c];
    }
  }
  return 0;
}

int gst114_scale_packed422_nearest (const uint8_t *in, int is, int yuy2, int w, int h, uint8_t *out, int os, int ow, int oh)
{
  if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0) return -1;
  int yo, uo, vo; pk_offsets (yuy2, &yo, &uo, &vo);
  const int cw = (w + 1) / 2, cow = (ow + 1) / 2;
  for (int y = 0; y < oh; y++) {
    const uint8_t *row = in + (size_t) gst114_nearest_index (h, oh, y) * is;
    uint8_t *o = out + (size_t) y * os;
    for (int x = 0; x < ow; x++) o[2 * x + yo] = row[2 * gst114_nearest_index (w, ow, x) + yo];
    if (ow & 1) o[2 * ow + yo] = o[2 * (ow - 1) + yo];
    for (int k = 0; k < cow; k++) {
      const int i = gst114_nearest_index (cw, cow, k);
      o[4 * k + uo] = row[4 * i + uo]; o[4 * k + vo] = row[4 * i + vo];
    }
  }
  return 0;
}

/* ---- videoscale method=catrom on YUV frames (pinned by 48 real-pipeline vectors, tests/golden/convertscale_gst114_yuvcubic.npz):
 *   the luma plane and all three interleaved lines of a packed frame take the catrom n-tap passes of the 4 x u8 case;
 *   the CHROMA planes of a planar frame (I420 U / V, NV12 UV) do not: GstVideoConverter scales them with its LINEAR
 *   method without videoscale's 2-tap limit, n_taps = ceil (2 * max (1, in / out)).  Each pass picks its line function by
 *   its own tap count: 2 taps -> the pinned 2-tap functions (vertical 8-bit; horizontal edge-aligned 16.16 for 1 x u8,
 *   centre-aligned 6-bit for 2 x u8), more -> the generic n-tap function (6-bit taps, clamp ((sum + 32) >> 6));
 *   vertical pass first iff in_h > out_h + n_taps_v.  Same domain: every scaled line is at least as long as its filter. */
static int ntap_line_pass (int kernel, const uint8_t *in, int is, int istep, int w, int h, int n, uint8_t *out, int os, int ostep, int on, int vertical)
{
  const int len = vertical ? h : w;
  const int nt = kernel ? linear_n_taps (len, on) : cubic_n_taps (len, on);
  if (kernel && nt == 2) {                            /* the special 2-tap line functions (n = istep = ostep here) */
    if (vertical) vscale_plane (in, is, n * w, h, out, os, on);
    else hscale_plane (in, is, w, h, n, out, os, on, 0);
    return 0;
  }
  int *idx = malloc (sizeof (int) * (size_t) nt * on), *tp = malloc (sizeof (int) * (size_t) nt * on);
  if (!idx || !tp || resampler_taps (kernel, len, on, idx, tp, nt * on) < 0) { free (idx); free (tp); return -3; }
  const int oh_ = vertical ? on : h, ow_ = vertical ? w : on;
  for (int y = 0; y < oh_; y++)
    for (int x = 0; x < ow_; x++)
      for (int c = 0; c < n; c++) {
        int acc = 0;
        for (int l = 0; l < nt; l++)
          acc += (vertical ? in[(size_t) idx[y * nt + l] * is + istep * x + c] : in[(size_t) y * is + istep * idx[x * nt + l] + c]) * (vertical ? tp[y * nt + l] : tp[x * nt + l]);
        out[(size_t) y * os + ostep * x + c] = (uint8_t) clampi ((acc + 32) >> 6, 0, 255);
      }
  free (idx); free (tp);
  return 0;
}

static int ntap_ok (int kernel, int in, int out) { const int n = kernel ? linear_n_taps (in, out) : cubic_n_taps (in, out); return in == out || (n <= in && n <= 64); }

/* one line set: n components per sample, samples istep / ostep bytes apart (the linear kernel is only used on planes of
 * their own: istep = ostep = n) */
static int scale_line_ntap (int kernel, const uint8_t *in, int is, int istep, int w, int h, int n, uint8_t *out, int os, int ostep, int ow, int oh)
{
  if (!ntap_ok (kernel, w, ow) || !ntap_ok (kernel, h, oh)) return -3;
  if (ow == w && oh == h) {
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) for (int c = 0; c < n; c++) out[(size_t) y * os + ostep * x + c] = in[(size_t) y * is + istep * x + c];
    return 0;
  }
  if (ow == w) return ntap_line_pass (kernel, in, is, istep, w, h, n, out, os, ostep, oh, 1);
  if (oh == h) return ntap_line_pass (kernel, in, is, istep, w, h, n, out, os, ostep, ow, 0);
  int rc;
  const int nv = kernel ? linear_n_taps (h, oh) : cubic_n_taps (h, oh);
  if (h > oh + nv) {
    uint8_t *tmp = malloc ((size_t) oh * w * n); if (!tmp) return -2;
    rc = ntap_line_pass (kernel, in, is, istep, w, h, n, tmp, w * n, n, oh, 1);
    if (!rc) rc = ntap_line_pass (kernel, tmp, w * n, n, w, oh, n, out, os, ostep, ow, 0);
    free (tmp);
  } else {
    uint8_t *tmp = malloc ((size_t) h * ow * n); if (!tmp) return -2;
    rc = ntap_line_pass (kernel, in, is, istep, w, h, n, tmp, ow * n, n, ow, 0);
    if (!rc) rc = ntap_line_pass (kernel, tmp, ow * n, n, ow, h, n, out, os, ostep, oh, 1);
    free (tmp);
  }
  return rc;
}
#define scale_line_cubic(in, is, istep, w, h, n, out, os, ostep, ow, oh) scale_line_ntap (0, in, is, istep, w, h, n, out, os, ostep, ow, oh)

/* `chroma`: a chroma plane of a planar frame (linear kernel); otherwise the luma plane (catrom) */
int gst114_scale_plane_cubic (const uint8_t *in, int is, int w, int h, int n, uint8_t *out, int os, int ow, int oh, int chroma)
{
  if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0 || (n != 1 && n != 2)) return -1;
  return scale_line_ntap (chroma ? 1 : 0, in, is, n, w, h, n, out, os, n, ow, oh);
}

int gst114_scale_packed422_cubic (const uint8_t *in, int is, int yuy2, int w, int h, uint8_t *out, int os, int ow, int oh)
{
  if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0) return -1;
  int yo, uo, vo; pk_offsets (yuy2, &yo, &uo, &vo);
  const int cw = (w + 1) / 2, cow = (ow + 1) / 2;
  int rc = scale_line_cubic (in + yo, is, 2, w, h, 1, out + yo, os, 2, ow, oh);
  if (!rc) rc = scale_line_cubic (in + uo, is, 4, cw, h, 1, out + uo, os, 4, cow, oh);
  if (!rc) rc = scale_line_cubic (in + vo, is, 4, cw, h, 1, out + vo, os, 4, cow, oh);
  if (!rc && (ow & 1)) for (int y = 0; y < oh; y++) out[(size_t) y * os + 2 * ow + yo] = out[(size_t) y * os + 2 * (ow - 1) + yo];
  return rc;
}
