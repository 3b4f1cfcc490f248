// csrc/convertscale.hip — vfhip_convertscale_* : host side of the fused colourspace-convert + scale.
// Mirrors MetalConvertScaleRenderer (reference convertscale/metalconvertscalerenderer.{h,m}):
//   -init                               -> vfhip_convertscale_new
//   -configureWithInputInfo:...         -> vfhip_convertscale_configure   (:226-330)
//   -processFrame:output:               -> vfhip_convertscale_process     (:332-512)
//   -cleanup                            -> vfhip_convertscale_cleanup     (:514-528)
#include "vfhip_internal.h"
#include "convertscale_kernels.h"
#include "convertscale_ntap_kernels.h"
#include "convertscale_metal_kernels.h"
#include "convertscale_planar_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

using namespace vfhip;


// videoconvert's RGB -> YUV 8-bit integer matrices (oracle/gst114.c RGB2YUV, pinned against the real element)
// videoconvert's 8-bit YUV -> YUV matrices, [matrix in][matrix out][row (Y, U, V) x (a, b, c, d)]: out = clamp8 (((a Y + b U + c V) >> 8) + d)
// (probed on the real element: oracle/gst114.c YUV2YUV)
static const int kYuv2Yuv[3][3][12] = {
  { { 0 }, { 256, -30, -53, 41,   0, 261, 29, -18,   0, 19, 262, -13 }, { 256, -32, -29, 30,   0, 259, 16, -10,   0, 22, 264, -15 } },
  { { 256, 25, 49, -38,   0, 253, -28, 15,   0, -19, 252, 11 }, { 0 }, { 256, -4, 24, -10,   0, 255, -13, 7,   0, 3, 257, -2 } },
  { { 256, 30, 26, -28,   0, 255, -15, 8,   0, -22, 250, 13 }, { 256, 5, -24, 9,   0, 257, 13, -8,   0, -3, 255, 1 }, { 0 } },
};
static const int kRgb2Yuv[3][9] = {
  {  66, 129,  25,  -38,  -74, 112,  112,  -94, -18 },   // bt601
  {  47, 157,  16,  -26,  -87, 112,  112, -102, -10 },   // bt709
  {  58, 149,  13,  -31,  -81, 112,  112, -103,  -9 },   // bt2020
};

struct PlaneCfg {                 // stage-2 set-up of one output plane (gst-exact, 4:2:0 outputs)
  int w = 0, h = 0, ow = 0, oh = 0, n = 1, hmode = 0, vmode = 0, vfirst = 0, nh = 0, nv = 0;
  int step = 1, off = 0;                    // packed 4:2:2: bytes between samples, offset of the first one
  uint32_t hinc = 0;
  int *d_vtab = nullptr, *d_htab = nullptr;
  int2 *d_hnt = nullptr, *d_vnt = nullptr;   // n-tap tables (method=bicubic)
};

static const int kOrcCoef[3][5] = {
  { 298, 409, 516, -100, -208 },   // bt601
  { 298, 459, 541,  -55, -136 },   // bt709
  { 298, 430, 548,  -48, -167 },   // bt2020
};

struct VfHipConvertScale {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo in {}, out {};
  int method = 0, add_borders = 0, numerics = 0;
  uint32_t border_color = 0;
  int rx = 0, ry = 0, rw = 0, rh = 0;
  int *d_vtab = nullptr, *d_htab = nullptr;
  int vfirst = 1, hscale_on = 0;
  uint32_t hinc = 0;
  enum Kernel { K_NONE, K_HALF, K_GENERIC, K_TAPS, K_METAL, K_STAGED, K_NTAP, K_SAME, K_BLTILE } kernel = K_NONE;
  int bl_th = 32;                   // K_BLTILE: tile height (32, or 16 when a 32-row tile's source region does not fit)
  int strip_rows = 4, strip_fill = 2;   // k_cs_taps_strip: rows per lane (1 = k_cs_taps), waves per SIMD a strip launch must give; $VFHIP_TAPS_ROWS / $VFHIP_TAPS_FILL at configure (A/B and test knobs)
  bool rgb_same = false;            // BGRA / RGBA -> BGRA / RGBA at the same size, no borders: k_cs_rgb_same when the frames meet its alignment contract
  bool taps_adjacent = false;       // bilinear: every row's two vertical taps are the same or adjacent source rows (k_cs_taps_strip's contract)
  Kernel same_fallback = K_NONE;    // K_SAME: what runs instead when a frame misses k_cs_yuv_same's alignment contract
  const char *kernel_name = "none";
  // K_STAGED: videoconvert at the input size into `mid` (when the format changes), then per-plane videoscale
  PlaneCfg plane[3];
  int n_out_planes = 0;
  bool need_convert = false, need_scale = false;
  bool remat = false;               // stage 1 is videoconvert's generic YUV -> YUV path (matrix and / or siting change): k_yuv_to_yuv
  bool lb = false;                                    // staged path with borders: the planes are written through a sub-rectangle view
  void *nt_tmp = nullptr; size_t nt_tmp_bytes = 0;    // intermediate plane of the two-pass n-tap path (YUV outputs, method=bicubic)
  void *mid = nullptr; size_t mid_bytes = 0; int mid_frames = 0;   // mid_bytes: one intermediate frame; mid holds mid_frames of them
  // K_NTAP (method=bicubic): optional conversion at the input size by a private child handle, then the n-tap passes
  VfHipConvertScale *conv = nullptr;
  int2 *d_nt_h = nullptr, *d_nt_v = nullptr; int nt_h = 0, nt_v = 0;
  void *nt_mid0 = nullptr, *nt_mid1 = nullptr;
  bool nt_tile = false;             // every tile's source region fits k_cs_cubic_tile's LDS arrays
  bool nt_dot = false;              // ... and k_cs_cubic_dot's planes, with every merged weight an int8: the dot-product tile kernel runs
  uint32_t *d_win_h = nullptr, *d_win_v = nullptr; int win_wh = 4, win_wv = 4;   // its window tables ([out][CD_WT]) and window widths
  Flights fl;                       // pipelined host path (submit / wait)
};

static void free_tables (VfHipConvertScale *h)
{
  if (h->d_vtab) (void) hipFree (h->d_vtab);
  if (h->d_htab) (void) hipFree (h->d_htab);
  h->d_vtab = h->d_htab = nullptr;
  for (auto &pc : h->plane) {
    if (pc.d_vtab) (void) hipFree (pc.d_vtab);
    if (pc.d_htab) (void) hipFree (pc.d_htab);
    if (pc.d_hnt) (void) hipFree (pc.d_hnt);
    if (pc.d_vnt) (void) hipFree (pc.d_vnt);
    pc = PlaneCfg ();
  }
  if (h->mid) (void) hipFree (h->mid);
  h->mid = nullptr; h->mid_bytes = 0; h->mid_frames = 0;
  if (h->nt_tmp) (void) hipFree (h->nt_tmp);
  h->nt_tmp = nullptr; h->nt_tmp_bytes = 0;
  if (h->d_nt_h) (void) hipFree (h->d_nt_h);
  if (h->d_nt_v) (void) hipFree (h->d_nt_v);
  if (h->d_win_h) (void) hipFree (h->d_win_h);
  if (h->d_win_v) (void) hipFree (h->d_win_v);
  h->d_win_h = h->d_win_v = nullptr; h->nt_dot = false;
  if (h->nt_mid0) (void) hipFree (h->nt_mid0);
  if (h->nt_mid1) (void) hipFree (h->nt_mid1);
  h->d_nt_h = h->d_nt_v = nullptr; h->nt_mid0 = h->nt_mid1 = nullptr; h->nt_h = h->nt_v = 0;
  if (h->conv) { vfhip_convertscale_free (h->conv); h->conv = nullptr; }
}

// ---- method=bicubic: GstVideoResampler's cubic set-up (b = 0, c = .5), the same double-precision operations in the same
// order as oracle/gst114.c gst114_cubic_taps (which is pinned against the real element) -------------------------------
static double cubic_k (double a)
{
  const double b = 0.0, c = 0.5;
  a = a < 0 ? -a : a;
  const double a2 = a * a, a3 = a2 * a;
  if (a <= 1.0) return ((12.0 - 9.0 * b - 6.0 * c) * a3 + (-18.0 + 12.0 * b + 6.0 * c) * a2 + (6.0 - 2.0 * b)) / 6.0;
  if (a <= 2.0) return ((-b - 6.0 * c) * a3 + (6.0 * b + 30.0 * c) * a2 + (-12.0 * b - 48.0 * c) * a + (8.0 * b + 24.0 * c)) / 6.0;
  return 0.0;
}

// GstVideoScaler's integer taps: floor(off + w * 2^prec) with a bisection on `off` (from .5, at most 64 steps) until they
// sum to 2^prec.  Exact .5 ties have no such `off`: the search ends a few ulps below .5 and double rounding lifts the
// LARGER tap (probed on the real element, oracle/gst114.c gst114_linear_taps; round-half-up is wrong on every tie).
static double tap_offset (const double *m, int cnt, int prec)
{
  const double mul = (double) (1 << prec);
  double lo = 0.0, hi = 1.0, off = 0.5;
  for (int it = 0; it < 64; it++) {
    int s = 0;
    for (int l = 0; l < cnt; l++) s += (int) std::floor (off + m[l] * mul);
    if (s == (1 << prec)) break;
    if (lo == hi) break;
    if (s < (1 << prec)) { if (off > lo) lo = off; off += (hi - lo) / 2; }
    else { if (off < hi) hi = off; off -= (hi - lo) / 2; }
  }
  return off;
}

// 2-tap linear set-up for output sample j of an in -> out line: source indices and the two integer taps
static void linear_taps (int in, int out, int j, int prec, int *i0, int *i1, int *t0, int *t1)
{
  double x = ((j + 0.5) / out) * in - 0.5;          // this order: the quotient first
  x = x < 0.0 ? 0.0 : (x > in - 1.0 ? in - 1.0 : x);
  const int xi = (int) std::floor (x);
  double w0 = 1.0 - std::fabs (x - xi), w1 = 1.0 - std::fabs (x - (xi + 1));
  if (w0 < 0.0) w0 = 0.0;
  if (w1 < 0.0) w1 = 0.0;
  const double sum = w0 + w1, m[2] = { w0 / sum, w1 / sum };
  const double off = tap_offset (m, 2, prec), mul = (double) (1 << prec);
  *i0 = xi < 0 ? 0 : (xi > in - 1 ? in - 1 : xi);
  *i1 = xi + 1 > in - 1 ? in - 1 : xi + 1;
  *t0 = (int) std::floor (off + m[0] * mul); *t1 = (int) std::floor (off + m[1] * mul);
}

static int cubic_n_taps (int in, int out)
{
  const double scale = (double) in / (double) out;
  const double fx = scale > 1.0 ? 1.0 / scale : 1.0;
  return (int) std::ceil (2.0 * 2.0 / fx);
}

static bool cubic_in_domain (int in, int out) { const int n = cubic_n_taps (in, out); return in == out || (n <= in && n <= 64); }

// un-limited LINEAR method (what videoscale's catrom leaves the chroma planes of a planar frame with): envelope 1
static int linear_n_taps (int in, int out)
{
  const double scale = (double) in / (double) out;
  const double fx = scale > 1.0 ? 1.0 / scale : 1.0;
  return (int) std::ceil (2.0 * 1.0 / fx);
}

// kernel: 0 = cubic (catrom), 1 = linear (1 - |a|)
static int ntap_table (int kernel, int in, int out, std::vector<int2> &tab)
{
  const int n = kernel ? linear_n_taps (in, out) : cubic_n_taps (in, out);
  const double fx = (kernel ? 2.0 : 4.0) / n;
  tab.assign ((size_t) n * out, make_int2 (0, 0));
  for (int j = 0; j < out; j++) {
    double x = ((j + 0.5) / out) * in - 0.5;        // this order: the quotient first (it decides which way exact .5 ties fall)
    x = x < 0.0 ? 0.0 : (x > in - 1.0 ? in - 1.0 : x);
    const int xi = (int) std::floor (x) - (n - 1) / 2;
    double w[64], sum = 0.0, m[64];
    int pos[64], cnt = 0;
    for (int l = 0; l < n; l++) {
      const double a = (x - (xi + l)) * fx;
      w[l] = kernel ? (std::fabs (a) < 1.0 ? 1.0 - std::fabs (a) : 0.0) : cubic_k (a);
      sum += w[l];
    }
    for (int l = 0; l < n; l++) {                    // taps outside the line are added to the edge tap, in double
      const int k = xi + l < 0 ? 0 : (xi + l > in - 1 ? in - 1 : xi + l);
      if (cnt > 0 && pos[cnt - 1] == k) m[cnt - 1] += w[l] / sum;
      else { pos[cnt] = k; m[cnt] = w[l] / sum; cnt++; }
    }
    const double off = tap_offset (m, cnt, 6);      // rounding offset that makes the 6-bit taps sum to 64
    for (int l = 0; l < n; l++)
      tab[(size_t) j * n + l] = l < cnt ? make_int2 (pos[l], (int) std::floor (off + m[l] * 64.0)) : make_int2 (pos[cnt - 1], 0);
  }
  return n;
}

static int cubic_table (int in, int out, std::vector<int2> &tab) { return ntap_table (0, in, out, tab); }

// the stream of the handle being configured (configure holds the handle's mutex; one handle per thread at a time)
static thread_local hipStream_t t_upload_stream = nullptr;

static int upload_int2 (const std::vector<int2> &v, int2 **dst)
{
  VFHIP_CHECK_HIP (dev_malloc (dst, v.size () * sizeof (int2)));
  VFHIP_CHECK_HIP (upload_in_stream (*dst, v.data (), v.size () * sizeof (int2), t_upload_stream));
  return VFHIP_OK;
}

static int upload_ints (const std::vector<int> &v, int **dst)
{
  *dst = nullptr;
  if (v.empty ()) return VFHIP_OK;
  VFHIP_CHECK_HIP (dev_malloc (dst, v.size () * sizeof (int)));
  VFHIP_CHECK_HIP (upload_in_stream (*dst, v.data (), v.size () * sizeof (int), t_upload_stream));
  return VFHIP_OK;
}

// GstVideoScaler's 2-tap vertical set-up (8-bit weights, centre aligned)
static void vertical_taps (int in_h, int out_h, std::vector<int> &vt)
{
  vt.assign ((size_t) out_h * 4, 0);
  for (int y = 0; y < out_h; y++) {
    int i0 = y, i1 = y, w = 0;
    if (out_h != in_h) { int t0; linear_taps (in_h, out_h, y, 8, &i0, &i1, &t0, &w); }   // the line function uses the second tap only
    vt[4 * y] = i0; vt[4 * y + 1] = i1; vt[4 * y + 2] = w;
  }
}

// stage-2 configuration of one plane (rules: oracle/gst114.c gst114_scale_plane)
static int nearest_index (int in, int out, int j);

// `nearest`: every tap table holds the nearest source index with a zero second tap (the 2-tap formulas then return the sample)
// `kernel`: -1 = videoscale's bilinear (2 taps), 0 = catrom, 1 = GstVideoConverter's un-limited LINEAR (planar chroma under
// catrom); with 0 / 1 each pass picks its line function by its own tap count (2 -> the 2-tap modes, more -> n-tap tables)
static int setup_plane (PlaneCfg &pc, int w, int h, int ow, int oh, int n, bool table = false, bool nearest = false, int kernel = -1)
{
  pc = PlaneCfg ();
  pc.w = w; pc.h = h; pc.ow = ow; pc.oh = oh; pc.n = n; pc.step = n;
  pc.vmode = oh != h ? 1 : 0; pc.vfirst = h > oh + 2;
  std::vector<int> vt, ht;
  vertical_taps (h, oh, vt);
  bool h2tap = true;
  if (kernel >= 0) {
    auto taps = [kernel] (int in, int out) { return kernel ? linear_n_taps (in, out) : cubic_n_taps (in, out); };
    for (int d = 0; d < 2; d++) {
      const int in = d ? h : w, out = d ? oh : ow;
      if (in != out && (taps (in, out) > in || taps (in, out) > 64))
        return set_error (VFHIP_ERR_UNSUPPORTED, "method=bicubic: a %d -> %d line is shorter than its %d-tap filter (or it has more than 64 taps)", in, out, taps (in, out));
    }
    if (oh != h && !(kernel == 1 && taps (h, oh) == 2)) {
      std::vector<int2> tv;
      pc.nv = ntap_table (kernel, h, oh, tv); pc.vmode = 2;
      int rc = upload_int2 (tv, &pc.d_vnt); if (rc) return rc;
    }
    pc.vfirst = h > oh + (oh != h ? taps (h, oh) : 2);
    if (ow != w && !(kernel == 1 && taps (w, ow) == 2)) {
      std::vector<int2> th;
      pc.nh = ntap_table (kernel, w, ow, th); pc.hmode = 4; h2tap = false;
      int rc = upload_int2 (th, &pc.d_hnt); if (rc) return rc;
    }
  }
  if (nearest) {
    for (int y = 0; y < oh; y++) { vt[4 * y] = vt[4 * y + 1] = nearest_index (h, oh, y); vt[4 * y + 2] = 0; }
    pc.vfirst = 0;
    if (oh != h) pc.vmode = 3;
    if (ow != w) {
      pc.hmode = 6;
      ht.assign ((size_t) ow * 4, 0);
      for (int x = 0; x < ow; x++) ht[4 * x] = ht[4 * x + 1] = nearest_index (w, ow, x);
    }
  } else if (!h2tap) { /* n-tap table set above */ }
  else if (ow == w) pc.hmode = 0;
  else if (n == 1 && !table && kernel < 0 && w == 2 * ow && (oh == h || h == 2 * oh)) pc.hmode = 2;
  else if (n == 1 && !table) { pc.hmode = 1; pc.hinc = (ow > 1 && w > 1) ? (uint32_t) ((((uint64_t) (w - 1)) << 16) / (uint64_t) (ow - 1)) - 1 : 0; }   // a one-sample line is replicated
  else {
    pc.hmode = 3;
    ht.assign ((size_t) ow * 4, 0);
    for (int x = 0; x < ow; x++) {
      int t0;
      linear_taps (w, ow, x, 6, &ht[4 * x], &ht[4 * x + 1], &t0, &ht[4 * x + 2]);
      if (t0 + ht[4 * x + 2] != 64) return set_error (VFHIP_ERR_UNSUPPORTED, "6-bit taps of column %d (%d -> %d) do not sum to 64", x, w, ow);
    }
    if (n == 2 && w == 2 * ow) {                    // exactly halved: every entry is (2k, 2k+1, 32) -> the kernel's dword path
      bool half = true;
      for (int x = 0; x < ow && half; x++) half = ht[4 * x] == 2 * x && ht[4 * x + 1] == 2 * x + 1 && ht[4 * x + 2] == 32;
      if (half) pc.hmode = 5;
    }
  }
  int rc = upload_ints (vt, &pc.d_vtab);
  if (rc) return rc;
  return upload_ints (ht, &pc.d_htab);
}

// GStreamer 1.14 nearest-neighbour source index: floor(((j + .5) / out) * in) in IEEE double, in that order.
static int nearest_index (int in, int out, int j)   // (declared above setup_plane)
{
  volatile double t = ((double) j + 0.5) / (double) out;
  volatile double p = t * (double) in;
  int i = (int) std::floor (p);
  return i < 0 ? 0 : (i > in - 1 ? in - 1 : i);
}

// inner rectangle when add-borders is set: aspect-preserving, centred (reference
// -_computeViewportWithAddBorders:, metalconvertscalerenderer.m:137-166)
static void compute_rect (VfHipConvertScale *h)
{
  const int iw = h->in.width, ih = h->in.height, ow = h->out.width, oh = h->out.height;
  h->rx = h->ry = 0; h->rw = ow; h->rh = oh;
  if (!h->add_borders || iw == 0 || ih == 0) return;
  const float src = (float) iw / (float) ih, dst = (float) ow / (float) oh;
  if (src > dst) { h->rh = (int) std::lround ((double) oh * (dst / src)); }
  else { h->rw = (int) std::lround ((double) ow * (src / dst)); }
  if (h->rw < 1) h->rw = 1;
  if (h->rh < 1) h->rh = 1;
  if (h->rw > ow) h->rw = ow;
  if (h->rh > oh) h->rh = oh;
  h->rx = (ow - h->rw) / 2; h->ry = (oh - h->rh) / 2;
}

static uint32_t border_in_output_order (uint32_t argb, int out_format)
{
  const uint32_t a = argb >> 24, r = (argb >> 16) & 0xff, g = (argb >> 8) & 0xff, b = argb & 0xff;
  if (out_format == VFHIP_FORMAT_RGBA) return r | (g << 8) | (b << 16) | (a << 24);
  return b | (g << 8) | (r << 16) | (a << 24);
}

static int env_int (const char *name, int dflt)
{
  const char *e = getenv (name);
  return e && *e ? atoi (e) : dflt;
}

static void launch_half (CsParams &p, int n_frames, int n_cu, hipStream_t s)
{
  const int cgpr = p.out_w / 4;
  auto bpf = [&] (int rows) { return (cgpr * ((p.out_h + rows - 1) / rows) + 255) / 256; };
  // rows per lane: long strips amortise the 2-row chroma prologue; short ones keep a small batch wide enough
  const size_t want = (size_t) n_cu * 4;                       // blocks for >= 16 waves per CU
  int rows = (size_t) bpf (16) * n_frames >= want ? 16 : ((size_t) bpf (8) * n_frames >= want ? 8 : 4);
  const int r = env_int ("VFHIP_HALF_ROWS", 0);                // tuning knobs (tools/gpu_ab.sh)
  if (r == 4 || r == 8 || r == 16) rows = r;
  p.half_rows = rows;
  dim3 grid ((unsigned) (bpf (rows) * n_frames));
  if (p.in_fmt == VFHIP_FORMAT_I420) {
    if (p.out_rgba) hipLaunchKernelGGL ((k_cs_i420_half<true>), grid, dim3 (256), 0, s, p);
    else hipLaunchKernelGGL ((k_cs_i420_half<false>), grid, dim3 (256), 0, s, p);
    return;
  }
  if (p.cosited) {
    if (p.out_rgba) hipLaunchKernelGGL ((k_cs_nv12_half<true, true>), grid, dim3 (256), 0, s, p);
    else hipLaunchKernelGGL ((k_cs_nv12_half<true, false>), grid, dim3 (256), 0, s, p);
  } else {
    if (p.out_rgba) hipLaunchKernelGGL ((k_cs_nv12_half<false, true>), grid, dim3 (256), 0, s, p);
    else hipLaunchKernelGGL ((k_cs_nv12_half<false, false>), grid, dim3 (256), 0, s, p);
  }
}

extern "C" {

VfHipConvertScale *vfhip_convertscale_new (int device)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  VfHipConvertScale *h = new (std::nothrow) VfHipConvertScale ();
  if (!h) { set_error (VFHIP_ERR_NOMEM, "out of memory"); return nullptr; }
  h->dev = d;
  if (h->st.init (d) != VFHIP_OK) { delete h; return nullptr; }
  return h;
}

int vfhip_convertscale_configure (VfHipConvertScale *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out,
    int method, int add_borders, uint32_t border_color, int numerics)
{
  if (!h || !in || !out) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "configure with %d submitted frame(s) still in flight: wait for them first", h->fl.count);
  t_upload_stream = h->st.s_compute;
  if (in->width <= 0 || in->height <= 0 || out->width <= 0 || out->height <= 0 || in->width > 32768 || in->height > 32768 ||
      out->width > 32768 || out->height > 32768)
    return set_error (VFHIP_ERR_INVALID, "bad frame size %dx%d -> %dx%d", in->width, in->height, out->width, out->height);
  if (format_n_planes (in->format) < 0 || format_n_planes (out->format) < 0) return VFHIP_ERR_INVALID;
  if (method != VFHIP_SCALE_BILINEAR && method != VFHIP_SCALE_NEAREST && method != VFHIP_SCALE_BICUBIC)
    return set_error (VFHIP_ERR_INVALID, "bad method %d", method);
  if (numerics != VFHIP_NUMERICS_GST_EXACT && numerics != VFHIP_NUMERICS_METAL && numerics != VFHIP_NUMERICS_GST_EXACT_STRICT)
    return set_error (VFHIP_ERR_INVALID, "bad numerics %d", numerics);
  const bool strict = numerics == VFHIP_NUMERICS_GST_EXACT_STRICT;
  if (strict) numerics = VFHIP_NUMERICS_GST_EXACT;
  if (in->color_matrix < 0 || in->color_matrix > 2 || out->color_matrix < 0 || out->color_matrix > 2)
    return set_error (VFHIP_ERR_INVALID, "bad colour matrix");
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  h->configured = false; h->lb = false; h->rgb_same = false;
  free_tables (h);
  h->in = *in; h->out = *out; h->method = method; h->add_borders = add_borders ? 1 : 0;
  h->border_color = border_color; h->numerics = numerics;
  compute_rect (h);

  const bool in_420_or_rgb = in->format == VFHIP_FORMAT_NV12 || in->format == VFHIP_FORMAT_I420 ||
                             in->format == VFHIP_FORMAT_BGRA || in->format == VFHIP_FORMAT_RGBA;
  const bool in_packed = in->format == VFHIP_FORMAT_UYVY || in->format == VFHIP_FORMAT_YUY2;   // pinned for RGB outputs only
  const bool out_rgb = out->format == VFHIP_FORMAT_BGRA || out->format == VFHIP_FORMAT_RGBA;
  // gst-exact covers the cells whose GStreamer arithmetic is pinned; every other cell of
  // the 6x6 matrix runs the reference-shader (metal) arithmetic.
  const bool in_yuv = in->format == VFHIP_FORMAT_NV12 || in->format == VFHIP_FORMAT_I420;
  const bool out_420 = out->format == VFHIP_FORMAT_NV12 || out->format == VFHIP_FORMAT_I420;
  // 4:2:0 outputs: pinned for the 2-tap method, no borders, and (YUV -> YUV) an unchanged matrix (no re-matrixing step)
  // packed 4:2:2 outputs (from all six formats) and packed -> 4:2:0 under the same conditions.  Not pinned: a 2-pixel-wide
  // packed frame scaled horizontally (GStreamer 1.14 emits out-of-line garbage for it) -> metal arithmetic
  const bool out_packed = out->format == VFHIP_FORMAT_UYVY || out->format == VFHIP_FORMAT_YUY2;
  const bool any_yuv_in = in_yuv || in_packed;
  // add-borders with a YUV output: the reference's centred rectangle (compute_rect) receives exactly what the two-step path
  // gives at the rectangle's size, the rest is the border colour through the RGB -> YUV matrix — when the rectangle sits on
  // chroma-sample boundaries (even x / width; even y / height as well for 4:2:0); otherwise metal arithmetic
  const bool lb = h->add_borders && (h->rw != out->width || h->rh != out->height);
  const bool lb_ok = !lb || (!((h->rx | h->rw) & 1) && (out_packed || !((h->ry | h->rh) & 1)));
  // YUV -> YUV with a matrix change (any of the four YUV formats either side), or NV12 <-> I420 with a siting change: stage 1 is
  // videoconvert's generic path (k_yuv_to_yuv).  With one matrix the siting may differ where GStreamer either ignores it (the same
  // format on both sides: videoconvert passes through; I420 <-> packed fast paths; the UYVY <-> YUY2 swizzle) or resamples with
  // both (NV12 <-> packed: the conversion kernels take both).
  const bool remat = numerics == VFHIP_NUMERICS_GST_EXACT && any_yuv_in && (out_420 || out_packed) &&
                     (in->color_matrix != out->color_matrix || (in_yuv && out_420 && in->format != out->format && in->chroma_site != out->chroma_site));
  h->remat = false;
  const bool staged = numerics == VFHIP_NUMERICS_GST_EXACT && lb_ok &&
                      ((in_420_or_rgb && out_420) || out_packed || (in_packed && out_420)) &&
                      !(out_packed && in->width == 2 && h->rw != 2);
  // RGB -> RGB at the same size (whatever the method: videoscale passes through): a copy or the R <-> B swap, chosen at launch when the frames are aligned
  h->rgb_same = numerics == VFHIP_NUMERICS_GST_EXACT && (in->format == VFHIP_FORMAT_BGRA || in->format == VFHIP_FORMAT_RGBA) && out_rgb &&
                in->width == out->width && in->height == out->height && h->rw == out->width && h->rh == out->height && !(in->width & 3) && getenv ("VFHIP_NO_SAME") == nullptr;
  if (method == VFHIP_SCALE_BICUBIC && !staged) {
    const int iw = in->width, ih = in->height, ow = h->rw, oh = h->rh;      // the destination rectangle (= the frame without borders)
    if (numerics != VFHIP_NUMERICS_GST_EXACT || !(in_420_or_rgb || in_packed) || !out_rgb)
      return set_error (VFHIP_ERR_UNSUPPORTED, "method=bicubic needs numerics=gst-exact (YUV outputs: the same matrix and chroma siting on both sides, borders only on chroma-sample boundaries)");
    if (!cubic_in_domain (iw, ow) || !cubic_in_domain (ih, oh))
      return set_error (VFHIP_ERR_UNSUPPORTED, "method=bicubic: %dx%d -> %dx%d has a line shorter than its filter (or more than 64 taps)", iw, ih, ow, oh);
    std::vector<int2> th_, tv_;
    if (ow != iw) { h->nt_h = cubic_table (iw, ow, th_); int rc = upload_int2 (th_, &h->d_nt_h); if (rc) return rc; }
    if (oh != ih) { h->nt_v = cubic_table (ih, oh, tv_); int rc = upload_int2 (tv_, &h->d_nt_v); if (rc) return rc; }
    h->vfirst = ih > oh + h->nt_v ? 1 : 0;           // GstVideoScaler pass order for n taps (oracle/gst114.c)
    // fused tile kernel when every 64 x 16 output tile's source region fits its LDS arrays
    auto span = [] (const std::vector<int2> &tb, int n, int out, int tile, int in) {
      if (!n) return tile < in ? tile : in;
      int m = 0;
      for (int o0 = 0; o0 < out; o0 += tile) {
        const int o1 = (o0 + tile < out ? o0 + tile : out) - 1;
        const int sp = tb[(size_t) o1 * n + n - 1].x - tb[(size_t) o0 * n].x + 1;
        if (sp > m) m = sp;
      }
      return m;
    };
    const int rw = span (th_, h->nt_h, ow, CT_TW, iw), rh = span (tv_, h->nt_v, oh, CT_TH, ih);
    const int rwa = rw + 14;                          // 8-column alignment slack of the NV12 fast conversion on both sides
    h->nt_tile = h->nt_h <= CT_MAXN && h->nt_v <= CT_MAXN && rwa <= CT_RW && rh <= CT_RH && (h->vfirst ? CT_TH * rwa : rh * CT_TW) <= CT_RH * CT_TW;
    if (const char *e = getenv ("VFHIP_CUBIC_TILE")) h->nt_tile = h->nt_tile && atoi (e) != 0;        // tuning / test knob
    // k_cs_cubic_dot's window tables: per output { start, tap sum, W int8 weights (taps the edge clamp put on one sample merged), alpha of an opaque
    // source after this pass }; an unscaled axis is the identity window { o, 64, [64, 0, 0, 0] }.  The kernel runs when every tile's windows fit its
    // planes and every merged weight is an int8
    {
      auto windows = [] (const std::vector<int2> &tb, int n, int out, std::vector<uint32_t> &w, int &W) {
        W = n ? 4 * ((n + 3) / 4) : 4;
        w.assign ((size_t) out * CD_WT, 0u);
        bool ok = W <= 12;
        for (int o = 0; o < out && ok; o++) {
          int wt[12] = { 0 }, sum = 0, s0 = o;
          if (n) {
            s0 = tb[(size_t) o * n].x;
            for (int l = 0; l < n; l++) {
              const int k = tb[(size_t) o * n + l].x - s0;
              if (k < 0 || k >= W) { ok = false; break; }
              wt[k] += tb[(size_t) o * n + l].y; sum += tb[(size_t) o * n + l].y;
            }
          } else { wt[0] = 64; sum = 64; }
          uint32_t *e = &w[(size_t) o * CD_WT];
          e[0] = (uint32_t) s0; e[1] = (uint32_t) sum;
          for (int k = 0; k < 12; k++) {
            if (wt[k] < -128 || wt[k] > 127) ok = false;
            e[2 + k / 4] |= (uint32_t) (wt[k] & 0xff) << (8 * (k & 3));
          }
          const int a = (255 * sum + 32) >> 6;
          e[5] = (uint32_t) (a < 0 ? 0 : (a > 255 ? 255 : a));
          if (sum < -255 || sum > 255) ok = false;                      // (the kernel's 128 * sum + 32 and alpha * sum stay far inside 32 bits; sanity only)
        }
        return ok;
      };
      std::vector<uint32_t> wh_, wv_;
      bool ok = windows (th_, h->nt_h, ow, wh_, h->win_wh) && windows (tv_, h->nt_v, oh, wv_, h->win_wv);
      // every tile's region: columns [first start & ~7, last start + W), rows [first start, last start + W)
      for (int x0 = 0; x0 < ow && ok; x0 += CD_TW) {
        const int x1 = std::min (x0 + CD_TW, ow) - 1;
        const int cols = (int) wh_[(size_t) x1 * CD_WT] + h->win_wh - ((int) wh_[(size_t) x0 * CD_WT] & ~7);
        if (((cols + 7) & ~7) > CD_RW - 8) ok = false;
      }
      for (int y0 = 0; y0 < oh && ok; y0 += CD_TH) {
        const int y1 = std::min (y0 + CD_TH, oh) - 1;
        if ((int) wv_[(size_t) y1 * CD_WT] + h->win_wv - (int) wv_[(size_t) y0 * CD_WT] > CD_RH) ok = false;
      }
      if (const char *e = getenv ("VFHIP_CUBIC_DOT")) ok = ok && atoi (e) != 0;                     // test / A-B knob: 0 = k_cs_cubic_tile
      h->nt_dot = ok && h->nt_tile;
      if (h->nt_dot) {
        VFHIP_CHECK_HIP (dev_malloc (&h->d_win_h, wh_.size () * sizeof (uint32_t)));
        VFHIP_CHECK_HIP (upload_in_stream (h->d_win_h, wh_.data (), wh_.size () * sizeof (uint32_t), t_upload_stream));
        VFHIP_CHECK_HIP (dev_malloc (&h->d_win_v, wv_.size () * sizeof (uint32_t)));
        VFHIP_CHECK_HIP (upload_in_stream (h->d_win_v, wv_.data (), wv_.size () * sizeof (uint32_t), t_upload_stream));
      }
    }
    if (!h->nt_tile) {                                // three-pass fallback: conversion at the input size by a child handle, then the passes
      if (in->format != out->format) {
        h->conv = vfhip_convertscale_new (h->dev->ordinal);
        if (!h->conv) return VFHIP_ERR_HIP;
        VfHipVideoInfo mid = *out;
        mid.width = iw; mid.height = ih;
        int rc = vfhip_convertscale_configure (h->conv, in, &mid, VFHIP_SCALE_BILINEAR, 0, 0, VFHIP_NUMERICS_GST_EXACT);
        if (rc) return rc;
        VFHIP_CHECK_HIP (dev_malloc (&h->nt_mid0, (size_t) iw * ih * 4 + 256));
      }
      if (h->nt_h && h->nt_v) VFHIP_CHECK_HIP (dev_malloc (&h->nt_mid1, (h->vfirst ? (size_t) iw * oh : (size_t) ow * ih) * 4 + 256));
    }
    h->kernel = VfHipConvertScale::K_NTAP; h->kernel_name = h->nt_dot ? "k_cs_cubic_dot" : (h->nt_tile ? "k_cs_cubic_tile" : "k_cs_ntap");
    h->configured = true;
    return VFHIP_OK;
  }
  if (staged) {
    const int iw = in->width, ih = in->height, ow = h->rw, oh = h->rh;      // the destination rectangle (= the frame without borders)
    h->lb = lb;
    h->remat = remat;
    h->need_convert = in->format != out->format || remat;
    h->need_scale = iw != ow || ih != oh;
    h->n_out_planes = out->format == VFHIP_FORMAT_NV12 ? 2 : 3;
    if (out_packed) {
      // videoscale on a packed frame: three interleaved lines (luma every 2 bytes, U and V every 4), each scaled like
      // NV12's chroma plane (6-bit table taps), vertical pass over every byte
      const int yuy2 = out->format == VFHIP_FORMAT_YUY2;
      const bool nn = method == VFHIP_SCALE_NEAREST;
      const int kern = method == VFHIP_SCALE_BICUBIC ? 0 : -1;          // catrom runs on all three interleaved lines
      int rc = setup_plane (h->plane[0], iw, ih, ow, oh, 1, true, nn, kern);
      if (!rc) rc = setup_plane (h->plane[1], (iw + 1) / 2, ih, (ow + 1) / 2, oh, 1, true, nn, kern);
      if (!rc) rc = setup_plane (h->plane[2], (iw + 1) / 2, ih, (ow + 1) / 2, oh, 1, true, nn, kern);
      if (rc) return rc;
      h->plane[0].step = 2; h->plane[0].off = yuy2 ? 0 : 1;
      h->plane[1].step = 4; h->plane[1].off = yuy2 ? 1 : 0;
      h->plane[2].step = 4; h->plane[2].off = yuy2 ? 3 : 2;
      if (kern < 0) for (int k = 0; k < 3; k++) h->plane[k].vfirst = !nn && ih > oh + 2;
      if (h->need_convert && h->need_scale) {
        h->mid_bytes = (((size_t) 4 * ((iw + 1) / 2) + 15) / 16 * 16) * ih + 1024;
        VFHIP_CHECK_HIP (dev_malloc (&h->mid, h->mid_bytes));
        h->mid_frames = 1;
      }
      h->kernel = VfHipConvertScale::K_STAGED; h->kernel_name = "k_cs_staged_422";
      h->configured = true;
      return VFHIP_OK;
    }
    const bool nn = method == VFHIP_SCALE_NEAREST;
    // method=bicubic: catrom on the luma plane; the chroma planes get GstVideoConverter's un-limited LINEAR taps (oracle/gst114.c)
    const int ky = method == VFHIP_SCALE_BICUBIC ? 0 : -1, kc = method == VFHIP_SCALE_BICUBIC ? 1 : -1;
    int rc = setup_plane (h->plane[0], iw, ih, ow, oh, 1, nn, nn, ky);
    if (rc) return rc;
    if (out->format == VFHIP_FORMAT_NV12) rc = setup_plane (h->plane[1], (iw + 1) / 2, (ih + 1) / 2, (ow + 1) / 2, (oh + 1) / 2, 2, nn, nn, kc);
    else { rc = setup_plane (h->plane[1], (iw + 1) / 2, (ih + 1) / 2, (ow + 1) / 2, (oh + 1) / 2, 1, nn, nn, kc);
           if (!rc) rc = setup_plane (h->plane[2], (iw + 1) / 2, (ih + 1) / 2, (ow + 1) / 2, (oh + 1) / 2, 1, nn, nn, kc); }
    if (rc) return rc;
    if (h->need_convert && h->need_scale) {        // intermediate frame: output format at the input size
      const size_t ys = ((size_t) iw + 15) / 16 * 16, cs = ((size_t) 2 * ((iw + 1) / 2) + 15) / 16 * 16;
      h->mid_bytes = ys * ih + 3 * cs * ((ih + 1) / 2) + 1024;
      VFHIP_CHECK_HIP (dev_malloc (&h->mid, h->mid_bytes));
      h->mid_frames = 1;
    }
    h->kernel = VfHipConvertScale::K_STAGED; h->kernel_name = "k_cs_staged_420";
    h->configured = true;
    return VFHIP_OK;
  }
  const bool exact = numerics == VFHIP_NUMERICS_GST_EXACT && (in_420_or_rgb || in_packed) && out_rgb;
  if (!exact) {
    // gst-exact asked for on a cell without pinned GStreamer arithmetic: never silently (vfhip_convertscale_numerics_in_effect,
    // the element's warning), and not at all under gst-exact-strict
    if (strict)
      return set_error (VFHIP_ERR_UNSUPPORTED, "numerics=gst-exact-strict: no pinned GStreamer arithmetic for this cell (format %d %dx%d matrix %d site %d -> format %d %dx%d matrix %d site %d%s)",
          in->format, in->width, in->height, in->color_matrix, in->chroma_site, out->format, out->width, out->height, out->color_matrix, out->chroma_site,
          h->add_borders ? ", borders" : "");
    h->kernel = VfHipConvertScale::K_METAL; h->kernel_name = "k_cs_metal";
    h->configured = true;
    return VFHIP_OK;
  }

  // tap tables for the destination rectangle (host double arithmetic identical to GstVideoResampler's 2-tap/1-tap set-up)
  std::vector<int> vt ((size_t) h->rh * 4, 0), ht;
  h->taps_adjacent = false;
  { const char *e = getenv ("VFHIP_TAPS_ROWS"); h->strip_rows = (e && atoi (e) < 4) ? 1 : 4; e = getenv ("VFHIP_TAPS_FILL"); h->strip_fill = e ? std::max (0, atoi (e)) : 2; }
  if (method == VFHIP_SCALE_NEAREST) {
    for (int y = 0; y < h->rh; y++) vt[4 * y] = nearest_index (in->height, h->rh, y);
    ht.resize (h->rw);
    for (int x = 0; x < h->rw; x++) ht[x] = nearest_index (in->width, h->rw, x);
  } else {
    h->taps_adjacent = true;
    for (int y = 0; y < h->rh; y++) {
      int i0 = y, i1 = y, w = 0;
      if (h->rh != in->height) { int t0; linear_taps (in->height, h->rh, y, 8, &i0, &i1, &t0, &w); }
      vt[4 * y] = i0; vt[4 * y + 1] = i1; vt[4 * y + 2] = w;
      if (i1 - i0 < 0 || i1 - i0 > 1) h->taps_adjacent = false;
    }
  }
  h->hscale_on = (method == VFHIP_SCALE_BILINEAR && h->rw != in->width) ? 1 : 0;
  h->hinc = (h->rw > 1 && in->width > 1) ? (uint32_t) ((((uint64_t) (in->width - 1)) << 16) / (uint64_t) (h->rw - 1)) - 1 : 0;   // a one-pixel line is replicated
  h->vfirst = in->height > h->rh + 2 ? 1 : 0;      // GstVideoScaler pass order (oracle/gst114.c rule 3)
  VFHIP_CHECK_HIP (dev_malloc (&h->d_vtab, vt.size () * sizeof (int)));
  VFHIP_CHECK_HIP (upload_in_stream (h->d_vtab, vt.data (), vt.size () * sizeof (int), h->st.s_compute));
  if (!ht.empty ()) {
    VFHIP_CHECK_HIP (dev_malloc (&h->d_htab, ht.size () * sizeof (int)));
    VFHIP_CHECK_HIP (upload_in_stream (h->d_htab, ht.data (), ht.size () * sizeof (int), h->st.s_compute));
  }

  const bool half = in_yuv && method == VFHIP_SCALE_BILINEAR && !h->add_borders &&
                    in->width == 2 * out->width && in->height == 2 * out->height && (out->width % 4) == 0 && out->height >= 3;
  const bool taps = in_yuv && method == VFHIP_SCALE_BILINEAR && in->width >= 8;
  // conversion only: NV12 / I420 / UYVY / YUY2 at the output's size, nothing to scale and no borders (bilinear or nearest: videoscale passes through either way)
  const bool same = any_yuv_in && method != VFHIP_SCALE_BICUBIC && in->width == out->width && in->height == out->height &&
                    h->rw == out->width && h->rh == out->height && (in->width % 8) == 0 && in->width >= 16 && getenv ("VFHIP_NO_SAME") == nullptr;
  if (half) { h->kernel = VfHipConvertScale::K_HALF; h->kernel_name = in->format == VFHIP_FORMAT_I420 ? "k_cs_i420_half" : "k_cs_nv12_half"; }
  else if (same) {
    h->kernel = VfHipConvertScale::K_SAME; h->kernel_name = in->format == VFHIP_FORMAT_I420 ? "k_cs_i420_same" : (in->format == VFHIP_FORMAT_NV12 ? "k_cs_nv12_same" : (in->format == VFHIP_FORMAT_UYVY ? "k_cs_uyvy_same" : "k_cs_yuy2_same"));
    h->same_fallback = taps ? VfHipConvertScale::K_TAPS : VfHipConvertScale::K_GENERIC;
  }
  else if (taps) { h->kernel = VfHipConvertScale::K_TAPS; h->kernel_name = "k_cs_taps"; }
  else { h->kernel = VfHipConvertScale::K_GENERIC; h->kernel_name = "k_cs_generic"; }
  // bilinear without minification (up-scales, one axis only, conversion at the same size from RGB): the source region of a 64 x 32 output
  // tile is smaller than the tile — k_cs_bilinear_tile converts each source pixel once per tile instead of four times per output pixel ...
  // ... and, for NV12 and the packed 4:2:2 inputs (whose region is converted eight pixels at a time), every down-scale whose tile regions fit the
  // LDS arrays as well (to ~2.2 : 1): NV12 1080p -> 720p 5.46 -> 4.59 us, UYVY 10.3 -> 4.5 us.  I420 (k_cs_taps<I420> is cheap: nearest chroma;
  // 3.9 vs 4.3 us, 3.3 in strips) and RGB inputs keep the per-pixel kernels for down-scales.
  // ... except NV12 minified on BOTH axes: k_cs_taps_strip (shared chroma rows, strips of four rows) beats the tile there since round 2
  // (1080p -> 720p 3.46 vs 5.03 us, 2160p -> 1920x1200 9.2 vs 14.4; profiles/r02x_taps_strip_ab.txt); one-axis down-scales stay with the tile
  const bool strip_shape = in->format == VFHIP_FORMAT_NV12 && taps && h->taps_adjacent && h->hscale_on && out->width < in->width && out->height < in->height;
  const bool bl_down = ((in->format == VFHIP_FORMAT_NV12 && !strip_shape) || in_packed) && in->width >= 16;
  if ((h->kernel == VfHipConvertScale::K_TAPS || h->kernel == VfHipConvertScale::K_GENERIC) && method == VFHIP_SCALE_BILINEAR &&
      h->rw == out->width && h->rh == out->height && h->rx == 0 && h->ry == 0 && ((out->width >= in->width && out->height >= in->height) || bl_down) &&
      getenv ("VFHIP_NO_BILINEAR_TILE") == nullptr) {
    const int iw = in->width, ow = out->width, oh = out->height;
    auto xa_of = [&] (int x) { return h->hscale_on ? std::min ((int) (((uint32_t) x * h->hinc) >> 16), iw - 1) : x; };
    int rwm = 0;
    for (int x0 = 0; x0 < ow; x0 += CT_TW) {
      const int x1 = std::min (x0 + CT_TW, ow) - 1;
      rwm = std::max (rwm, std::min (xa_of (x1) + (h->hscale_on ? 1 : 0), iw - 1) - xa_of (x0) + 1);
    }
    const int rwa = rwm + 14;                          // 8-column alignment slack of the NV12 fast conversion on both sides
    const bool vf = h->vfirst || !h->hscale_on;
    for (int th : { 32, 16 }) {
      int rhm = 0;
      for (int y0 = 0; y0 < oh; y0 += th) {
        const int y1 = std::min (y0 + th, oh) - 1;
        rhm = std::max (rhm, vt[4 * y1 + 1] - vt[4 * y0] + 1);
      }
      if (rwa <= CT_RW && rhm <= CT_RH && (vf ? th * rwa : rhm * CT_TW) <= CT_RH * CT_TW) {
        h->kernel = VfHipConvertScale::K_BLTILE; h->kernel_name = "k_cs_bilinear_tile"; h->bl_th = th;
        break;
      }
    }
  }
  h->configured = true;
  return VFHIP_OK;
}

const char *vfhip_convertscale_kernel_name (VfHipConvertScale *h) { return h ? (h->rgb_same ? "k_cs_rgb_same" : h->kernel_name) : "none"; }

int vfhip_convertscale_numerics_in_effect (VfHipConvertScale *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  if (!h->configured) return set_error (VFHIP_ERR_NOT_CONFIGURED, "convertscale: not configured");
  return h->kernel == VfHipConvertScale::K_METAL ? VFHIP_NUMERICS_METAL : VFHIP_NUMERICS_GST_EXACT;
}

static int validate_frames (VfHipConvertScale *h, const VfHipFrame *in, const VfHipFrame *out)
{
  if (!h || !in || !out) return set_error (VFHIP_ERR_INVALID, "null argument");
  if (!h->configured) return set_error (VFHIP_ERR_NOT_CONFIGURED, "convertscale: process before configure");
  if (in->info.format != h->in.format || in->info.width != h->in.width || in->info.height != h->in.height ||
      out->info.format != h->out.format || out->info.width != h->out.width || out->info.height != h->out.height)
    return set_error (VFHIP_ERR_INVALID, "frame does not match the configured caps");
  for (int p = 0; p < format_n_planes (in->info.format); p++)
    if (!in->data[p] || in->stride[p] < plane_width_bytes (in->info.format, p, in->info.width))
      return set_error (VFHIP_ERR_INVALID, "input plane %d: null pointer or short stride", p);
  for (int p = 0; p < format_n_planes (out->info.format); p++)
    if (!out->data[p] || out->stride[p] < plane_width_bytes (out->info.format, p, out->info.width))
      return set_error (VFHIP_ERR_INVALID, "output plane %d: null pointer or short stride", p);
  return VFHIP_OK;
}

// gst-exact, 4:2:0 output: stage 1 (format change at the input size) + stage 2 (per-plane scale)
// gst-exact cells with YUV outputs: videoconvert at the input size (stage 1), then videoscale (stage 2); every kernel
// takes the whole batch (blockIdx.z = frame)
static int staged_launch (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out, size_t in_pitch, size_t out_pitch, int n_frames, hipStream_t s)
{
  const int iw = h->in.width, ih = h->in.height;
  const bool out_planar = h->out.format == VFHIP_FORMAT_I420;
  const bool out_packed = h->out.format == VFHIP_FORMAT_UYVY || h->out.format == VFHIP_FORMAT_YUY2;
  VfHipFrame view = *out;
  if (h->lb) {
    // borders first (every sample outside the rectangle), then the two-step path writes the rectangle through a view of the frame
    BorderFillParams b {};
    const uint32_t argb = h->border_color;
    const int r = (argb >> 16) & 0xff, g = (argb >> 8) & 0xff, bl = argb & 0xff;
    const int *c = kRgb2Yuv[h->out.color_matrix];
    b.yuv[0] = ((c[0] * r + c[1] * g + c[2] * bl) >> 8) + 16; b.yuv[1] = ((c[3] * r + c[4] * g + c[5] * bl) >> 8) + 128; b.yuv[2] = ((c[6] * r + c[7] * g + c[8] * bl) >> 8) + 128;
    for (int k = 0; k < 3; k++) { b.p[k] = (uint8_t *) out->data[k]; b.s[k] = out->stride[k]; }
    b.fmt = h->out.format; b.w = h->out.width; b.h = h->out.height; b.rx = h->rx; b.ry = h->ry; b.rw = h->rw; b.rh = h->rh; b.pitch = out_pitch;
    dim3 grid ((unsigned) (((h->out.width + 1) / 2 + 63) / 64), (unsigned) ((h->out.height + 3) / 4), (unsigned) n_frames);
    hipLaunchKernelGGL (k_border_fill_yuv, grid, dim3 (64, 4), 0, s, b);
    VFHIP_CHECK_HIP (hipGetLastError ());
    if (out_packed) view.data[0] = (uint8_t *) out->data[0] + (size_t) h->ry * out->stride[0] + 2 * h->rx;
    else {
      view.data[0] = (uint8_t *) out->data[0] + (size_t) h->ry * out->stride[0] + h->rx;
      view.data[1] = (uint8_t *) out->data[1] + (size_t) (h->ry / 2) * out->stride[1] + (out_planar ? h->rx / 2 : h->rx);
      if (out_planar) view.data[2] = (uint8_t *) out->data[2] + (size_t) (h->ry / 2) * out->stride[2] + h->rx / 2;
    }
    out = &view;
  }
  const bool in_packed = h->in.format == VFHIP_FORMAT_UYVY || h->in.format == VFHIP_FORMAT_YUY2;
  const unsigned nz = (unsigned) n_frames;
  // where stage 1 writes / stage 2 reads: the output itself (no scaling), the input itself (no conversion), or `mid`
  VfHipFrame mid {};
  size_t mid_pitch = 0;
  mid.info = h->out; mid.info.width = iw; mid.info.height = ih;
  if (h->need_convert && h->need_scale) {
    if (h->mid_frames < n_frames) {              // intermediate frames of the batch (grown on demand; hipFree waits for the device)
      if (h->mid) (void) hipFree (h->mid);
      h->mid = nullptr; h->mid_frames = 0;
      VFHIP_CHECK_HIP (dev_malloc (&h->mid, h->mid_bytes * (size_t) n_frames));
      h->mid_frames = n_frames;
    }
    mid_pitch = h->mid_bytes;
    uint8_t *b = (uint8_t *) h->mid;
    if (out_packed) { mid.data[0] = b; mid.stride[0] = (int) (((size_t) 4 * ((iw + 1) / 2) + 15) / 16 * 16); }
    else {
      const size_t ys = ((size_t) iw + 15) / 16 * 16, cs = ((size_t) 2 * ((iw + 1) / 2) + 15) / 16 * 16;
      mid.data[0] = b; mid.stride[0] = (int) ys;
      mid.data[1] = b + ys * ih; mid.stride[1] = (int) cs;
      mid.data[2] = b + ys * ih + cs * ((ih + 1) / 2); mid.stride[2] = (int) cs;
    }
  } else if (h->need_convert) { mid = *out; mid_pitch = out_pitch; }
  else { mid = *in; mid_pitch = in_pitch; }
  const int cw = (iw + 1) / 2, chh = (ih + 1) / 2;
  if (h->need_convert) {
    dim3 grid ((unsigned) ((cw + 63) / 64), (unsigned) (((out_packed ? ih : chh) + 3) / 4), nz);
    if (h->remat) {
      YuvRematParams p {};
      p.in_pitch = in_pitch; p.out_pitch = mid_pitch;
      auto spec = [] (const VfHipFrame &f, int fmt, const uint8_t **y, const uint8_t **u, const uint8_t **v, int *ys, int *ystep, int *cs, int *cstep, int *is420) {
        const uint8_t *b0 = (const uint8_t *) f.data[0];
        if (fmt == VFHIP_FORMAT_NV12) { *y = b0; *ys = f.stride[0]; *ystep = 1; *u = (const uint8_t *) f.data[1]; *v = *u + 1; *cs = f.stride[1]; *cstep = 2; *is420 = 1; }
        else if (fmt == VFHIP_FORMAT_I420) { *y = b0; *ys = f.stride[0]; *ystep = 1; *u = (const uint8_t *) f.data[1]; *v = (const uint8_t *) f.data[2]; *cs = f.stride[1]; *cstep = 1; *is420 = 1; }
        else { const int yuy2 = fmt == VFHIP_FORMAT_YUY2; *y = b0 + (yuy2 ? 0 : 1); *u = b0 + (yuy2 ? 1 : 0); *v = b0 + (yuy2 ? 3 : 2); *ys = *cs = f.stride[0]; *ystep = 2; *cstep = 4; *is420 = 0; }
      };
      const uint8_t *oy, *ou, *ov;
      spec (*in, h->in.format, &p.iy, &p.iu, &p.iv, &p.iys, &p.iystep, &p.ics, &p.icstep, &p.in420);
      spec (mid, h->out.format, &oy, &ou, &ov, &p.oys, &p.oystep, &p.ocs, &p.ocstep, &p.out420);
      if (h->in.format == VFHIP_FORMAT_I420 && in->stride[2] != in->stride[1]) return set_error (VFHIP_ERR_UNSUPPORTED, "I420 input with different U and V strides");
      if (h->out.format == VFHIP_FORMAT_I420 && mid.stride[2] != mid.stride[1]) return set_error (VFHIP_ERR_UNSUPPORTED, "I420 output with different U and V strides");
      p.oy = const_cast<uint8_t *> (oy); p.ou = const_cast<uint8_t *> (ou); p.ov = const_cast<uint8_t *> (ov);
      p.w = iw; p.h = ih;
      p.cos_in = h->in.chroma_site == VFHIP_CHROMA_SITE_H_COSITED; p.cos_out = h->out.chroma_site == VFHIP_CHROMA_SITE_H_COSITED;
      p.remat = h->in.color_matrix != h->out.color_matrix;
      p.same = p.cos_in == p.cos_out && p.in420 == p.out420;
      for (int k = 0; k < 12; k++) p.t[k] = kYuv2Yuv[h->in.color_matrix][h->out.color_matrix][k];
      hipLaunchKernelGGL (k_yuv_to_yuv, grid, dim3 (64, 4), 0, s, p);
    } else if (out_packed) {
      ToPackedParams p {};
      p.in_pitch = in_pitch; p.out_pitch = mid_pitch;
      for (int k = 0; k < 3; k++) { p.in[k] = (const uint8_t *) in->data[k]; p.is[k] = in->stride[k]; }
      p.out = (uint8_t *) mid.data[0]; p.os = mid.stride[0];
      p.w = iw; p.h = ih; p.in_fmt = h->in.format; p.out_yuy2 = h->out.format == VFHIP_FORMAT_YUY2;
      p.cosited_in = h->in.chroma_site == VFHIP_CHROMA_SITE_H_COSITED; p.cosited_out = h->out.chroma_site == VFHIP_CHROMA_SITE_H_COSITED;
      for (int k = 0; k < 9; k++) p.c[k] = kRgb2Yuv[h->out.color_matrix][k];
      p.vec = h->in.format == VFHIP_FORMAT_NV12 && getenv ("VFHIP_PLANE_SCALAR") == nullptr &&
              (((uintptr_t) p.in[0] | (uintptr_t) p.in[1] | (uintptr_t) p.is[0] | (uintptr_t) p.is[1] | (uintptr_t) in_pitch) & 1) == 0;
      if (h->in.format == VFHIP_FORMAT_BGRA || h->in.format == VFHIP_FORMAT_RGBA) {
        const bool rgba = h->in.format == VFHIP_FORMAT_RGBA;
        auto pack = [&] (int r, int g, int b, bool neg) {
          auto part = [&] (int c) { return (uint32_t) (neg ? (c < 0 ? -c : 0) : (c > 0 ? c : 0)); };
          return (rgba ? part (r) : part (b)) | part (g) << 8 | (rgba ? part (b) : part (r)) << 16;
        };
        p.cy = pack (p.c[0], p.c[1], p.c[2], false);
        p.cup = pack (p.c[3], p.c[4], p.c[5], false); p.cun = pack (p.c[3], p.c[4], p.c[5], true);
        p.cvp = pack (p.c[6], p.c[7], p.c[8], false); p.cvn = pack (p.c[6], p.c[7], p.c[8], true);
        p.vec = (((uintptr_t) p.in[0] | (uintptr_t) p.is[0] | (uintptr_t) in_pitch) & 3) == 0 && getenv ("VFHIP_RGB2YUV_SCALAR") == nullptr;
      }
      // (tried: two macro-pixels per lane with the four U/V pairs as one unaligned 8-byte window — 3 loads per 8 output bytes
      // instead of 14 — was SLOWER, 168 k vs 214 k frames/s on NV12 1080p -> UYVY: misaligned 8-byte loads, half the lanes)
      hipLaunchKernelGGL (k_to_packed422, grid, dim3 (64, 4), 0, s, p);
    } else if (h->in.format == VFHIP_FORMAT_BGRA || h->in.format == VFHIP_FORMAT_RGBA) {
      Rgb2YuvParams p {};
      p.in_pitch = in_pitch; p.out_pitch = mid_pitch;
      p.in = (const uint8_t *) in->data[0]; p.is = in->stride[0];
      p.y = (uint8_t *) mid.data[0]; p.ys = mid.stride[0];
      p.u = (uint8_t *) mid.data[1]; p.us = mid.stride[1];
      p.v = (uint8_t *) mid.data[2]; p.vs = mid.stride[2];
      p.w = iw; p.h = ih; p.in_rgba = h->in.format == VFHIP_FORMAT_RGBA; p.planar = out_planar;
      p.cosited = h->out.chroma_site == VFHIP_CHROMA_SITE_H_COSITED;
      for (int k = 0; k < 9; k++) p.c[k] = kRgb2Yuv[h->out.color_matrix][k];
      if ((((uintptr_t) p.in | (uintptr_t) p.is | (uintptr_t) in_pitch) & 3) == 0 && getenv ("VFHIP_RGB2YUV_SCALAR") == nullptr) {
        // dword pixel loads + v_dot4: coefficients packed in the input's byte order, positive and negative parts apart
        Rgb2YuvFastParams q {};
        q.b = p;
        auto pack = [&] (int r, int g, int b, bool neg) {
          auto part = [&] (int c) { return (uint32_t) (neg ? (c < 0 ? -c : 0) : (c > 0 ? c : 0)); };
          const uint32_t lo = p.in_rgba ? part (r) : part (b), hi = p.in_rgba ? part (b) : part (r);
          return lo | part (g) << 8 | hi << 16;
        };
        q.cy = pack (p.c[0], p.c[1], p.c[2], false);
        q.cup = pack (p.c[3], p.c[4], p.c[5], false); q.cun = pack (p.c[3], p.c[4], p.c[5], true);
        q.cvp = pack (p.c[6], p.c[7], p.c[8], false); q.cvn = pack (p.c[6], p.c[7], p.c[8], true);
        hipLaunchKernelGGL (k_rgb_to_yuv420_fast, grid, dim3 (64, 4), 0, s, q);
      } else
      hipLaunchKernelGGL (k_rgb_to_yuv420, grid, dim3 (64, 4), 0, s, p);
    } else if (in_packed) {
      FromPackedParams p {};
      p.in_pitch = in_pitch; p.out_pitch = mid_pitch;
      p.in = (const uint8_t *) in->data[0]; p.is = in->stride[0];
      p.y = (uint8_t *) mid.data[0]; p.ys = mid.stride[0];
      p.u = (uint8_t *) mid.data[1]; p.us = mid.stride[1];
      p.v = (uint8_t *) mid.data[2]; p.vs = mid.stride[2];
      p.w = iw; p.h = ih; p.in_yuy2 = h->in.format == VFHIP_FORMAT_YUY2; p.planar = out_planar;
      p.cosited_in = h->in.chroma_site == VFHIP_CHROMA_SITE_H_COSITED; p.cosited_out = h->out.chroma_site == VFHIP_CHROMA_SITE_H_COSITED;
      p.vec = (((uintptr_t) p.in | (uintptr_t) p.is | (uintptr_t) in_pitch) & 3) == 0 && getenv ("VFHIP_PLANE_SCALAR") == nullptr;
      hipLaunchKernelGGL (k_packed422_to_420, grid, dim3 (64, 4), 0, s, p);
    } else {
      RepackParams p {};
      p.in_pitch = in_pitch; p.out_pitch = mid_pitch;
      p.iy = (const uint8_t *) in->data[0]; p.iys = in->stride[0];
      p.iu = (const uint8_t *) in->data[1]; p.ius = in->stride[1];
      p.iv = (const uint8_t *) in->data[2]; p.ivs = in->stride[2];
      p.oy = (uint8_t *) mid.data[0]; p.oys = mid.stride[0];
      p.ou = (uint8_t *) mid.data[1]; p.ous = mid.stride[1];
      p.ov = (uint8_t *) mid.data[2]; p.ovs = mid.stride[2];
      p.w = iw; p.h = ih; p.in_planar = h->in.format == VFHIP_FORMAT_I420; p.out_planar = out_planar;
      // 16-byte accesses when the frames allow (k_repack_420_vec's contract)
      const uintptr_t ya = (uintptr_t) p.iy | (uintptr_t) p.iys | (uintptr_t) p.oy | (uintptr_t) p.oys | (uintptr_t) in_pitch | (uintptr_t) mid_pitch;
      const uintptr_t ica = (uintptr_t) p.iu | (uintptr_t) p.ius | (p.in_planar ? (uintptr_t) p.iv | (uintptr_t) p.ivs : 0);
      const uintptr_t oca = (uintptr_t) p.ou | (uintptr_t) p.ous | (p.out_planar ? (uintptr_t) p.ov | (uintptr_t) p.ovs : 0);
      if (!(iw & 15) && !(ih & 1) && !(ya & 15) && !(ica & (p.in_planar ? 7 : 15)) && !(oca & (p.out_planar ? 7 : 15)) && getenv ("VFHIP_PLANE_SCALAR") == nullptr) {
        dim3 vg ((unsigned) ((iw / 16 + 63) / 64), (unsigned) ((chh + 3) / 4), nz);
        hipLaunchKernelGGL (k_repack_420_vec, vg, dim3 (64, 4), 0, s, p);
      } else
      hipLaunchKernelGGL (k_repack_420, grid, dim3 (64, 4), 0, s, p);
    }
    VFHIP_CHECK_HIP (hipGetLastError ());
  }
  if (!(h->need_scale || !h->need_convert)) return VFHIP_OK;
  auto fill = [&] (PlaneScaleParams &p, const PlaneCfg &pc) {
    p.in_pitch = mid_pitch; p.out_pitch = out_pitch;
    p.w = pc.w; p.h = pc.h; p.ow = pc.ow; p.oh = pc.oh; p.n = pc.n; p.istep = p.ostep = pc.step; p.hmode = pc.hmode;
    p.vmode = pc.vmode; p.vfirst = pc.vfirst; p.hinc = pc.hinc; p.vtab = pc.d_vtab; p.htab = pc.d_htab;
    p.hnt = pc.d_hnt; p.vnt = pc.d_vnt; p.nh = pc.nh; p.nv = pc.nv;
  };
  if (out_packed) {
    PackedScaleParams q {};
    for (int k = 0; k < 3; k++) {
      fill (q.pl[k], h->plane[k]);
      q.pl[k].n = 1;
      q.pl[k].in = (const uint8_t *) mid.data[0] + h->plane[k].off; q.pl[k].is = mid.stride[0];
      q.pl[k].out = (uint8_t *) out->data[0]; q.pl[k].os = out->stride[0];
    }
    q.yo = h->plane[0].off; q.uo = h->plane[1].off; q.vo = h->plane[2].off;
    q.fast = h->plane[0].hmode == 3 && h->plane[1].hmode == 3 && h->plane[2].hmode == 3 && h->plane[0].vmode <= 1 && getenv ("VFHIP_PLANE_SCALAR") == nullptr;
    dim3 grid ((unsigned) ((h->plane[1].ow + 63) / 64), (unsigned) ((h->plane[0].oh + 3) / 4), nz);
    hipLaunchKernelGGL (k_scale_packed422, grid, dim3 (64, 4), 0, s, q);
    VFHIP_CHECK_HIP (hipGetLastError ());
    return VFHIP_OK;
  }
  for (int k = 0; k < h->n_out_planes; k++) {
    const PlaneCfg &pc = h->plane[k];
    PlaneScaleParams p {};
    fill (p, pc);
    p.in = (const uint8_t *) mid.data[k]; p.is = mid.stride[k];
    p.out = (uint8_t *) out->data[k]; p.os = out->stride[k];
    p.vec = (((uintptr_t) p.in | (uintptr_t) p.is | (uintptr_t) mid_pitch) & 3) == 0 && getenv ("VFHIP_PLANE_SCALAR") == nullptr;
    if (pc.vmode == 2 && pc.hmode == 4 && getenv ("VFHIP_PLANE_COMPOSED") == nullptr) {
      // method=bicubic, both directions: two passes through an intermediate plane (n_v + n_h loads per sample, not n_v x n_h)
      const int wb_in = pc.n * pc.w, wb_out = pc.n * pc.ow;
      const size_t ts = ((size_t) (pc.vfirst ? wb_in : wb_out) + 15) / 16 * 16, tbytes = ts * (size_t) (pc.vfirst ? pc.oh : pc.h);
      if (h->nt_tmp_bytes < tbytes * (size_t) n_frames) {
        if (h->nt_tmp) (void) hipFree (h->nt_tmp);
        h->nt_tmp = nullptr; h->nt_tmp_bytes = 0;
        VFHIP_CHECK_HIP (dev_malloc (&h->nt_tmp, tbytes * (size_t) n_frames));
        h->nt_tmp_bytes = tbytes * (size_t) n_frames;
      }
      PlaneTapParams a {}, b {};
      a.in = p.in; a.is = p.is; a.in_pitch = mid_pitch; a.out = (uint8_t *) h->nt_tmp; a.os = (int) ts; a.out_pitch = tbytes; a.n = pc.n;
      b.in = (const uint8_t *) h->nt_tmp; b.is = (int) ts; b.in_pitch = tbytes; b.out = p.out; b.os = p.os; b.out_pitch = out_pitch; b.n = pc.n;
      a.vec = (((uintptr_t) a.in | (uintptr_t) a.is | (uintptr_t) a.in_pitch) & 3) == 0; b.vec = 1;
      if (pc.vfirst) { a.wb = wb_in; a.rows = pc.oh; a.tab = pc.d_vnt; a.nt = pc.nv; b.wb = wb_out; b.rows = pc.oh; b.tab = pc.d_hnt; b.nt = pc.nh; }
      else { a.wb = wb_out; a.rows = pc.h; a.tab = pc.d_hnt; a.nt = pc.nh; b.wb = wb_out; b.rows = pc.oh; b.tab = pc.d_vnt; b.nt = pc.nv; }
      dim3 ga ((unsigned) ((a.wb + 255) / 256), (unsigned) ((a.rows + 3) / 4), nz), gb ((unsigned) ((b.wb + 255) / 256), (unsigned) ((b.rows + 3) / 4), nz);
      if (pc.vfirst) { hipLaunchKernelGGL (k_plane_vtap, ga, dim3 (64, 4), 0, s, a); hipLaunchKernelGGL (k_plane_htap, gb, dim3 (64, 4), 0, s, b); }
      else { hipLaunchKernelGGL (k_plane_htap, ga, dim3 (64, 4), 0, s, a); hipLaunchKernelGGL (k_plane_vtap, gb, dim3 (64, 4), 0, s, b); }
      VFHIP_CHECK_HIP (hipGetLastError ());
      continue;
    }
    // contiguous source bytes (no horizontal pass, or an exact half): 8 output bytes per lane; 2-tap gathers and n-tap tables:
    // 4 bytes per lane, each family in its own small kernel
    const bool twotap = p.vec && pc.vmode <= 1;
    if (twotap && (pc.hmode == 0 || pc.hmode == 2 || pc.hmode == 5) && getenv ("VFHIP_PLANE_G1") == nullptr) {
      dim3 grid ((unsigned) ((pc.n * pc.ow + 511) / 512), (unsigned) ((pc.oh + 3) / 4), nz);
      hipLaunchKernelGGL ((k_scale_plane<2, 0>), grid, dim3 (64, 4), 0, s, p);
    } else {
      dim3 grid ((unsigned) ((pc.n * pc.ow + 255) / 256), (unsigned) ((pc.oh + 3) / 4), nz);
      if (twotap && (pc.hmode == 0 || pc.hmode == 2 || pc.hmode == 5)) hipLaunchKernelGGL ((k_scale_plane<1, 0>), grid, dim3 (64, 4), 0, s, p);
      else if (twotap && (pc.hmode == 1 || pc.hmode == 3)) hipLaunchKernelGGL ((k_scale_plane<1, 1>), grid, dim3 (64, 4), 0, s, p);
      else hipLaunchKernelGGL ((k_scale_plane<1, 2>), grid, dim3 (64, 4), 0, s, p);
    }
    VFHIP_CHECK_HIP (hipGetLastError ());
  }
  return VFHIP_OK;
}

// launch on device frames; caller holds the mutex
static int launch_device (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out, size_t in_pitch, size_t out_pitch,
    int n_frames, hipStream_t s);

// method=bicubic: [conversion at the input size ->] n-tap pass -> n-tap pass, in GstVideoScaler's order
static int ntap_launch (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out, hipStream_t s)
{
  const int iw = h->in.width, ih = h->in.height, ow = h->rw, oh = h->rh;      // `out` is the view of the destination rectangle
  const uint8_t *src = (const uint8_t *) in->data[0];
  int ss = in->stride[0];
  if (h->conv) {
    VfHipFrame mid {};
    mid.info = h->conv->out;
    mid.data[0] = h->nt_mid0; mid.stride[0] = iw * 4;
    int rc = launch_device (h->conv, in, &mid, 0, 0, 1, s);
    if (rc) return rc;
    src = (const uint8_t *) h->nt_mid0; ss = iw * 4;
  }
  auto pass = [&] (bool vertical, const uint8_t *pin, int pis, uint8_t *pout, int pos, int w, int hh) {
    NtapParams p {};
    p.in = pin; p.is = pis; p.out = pout; p.os = pos; p.w = w; p.h = hh;
    p.n = vertical ? h->nt_v : h->nt_h; p.tab = vertical ? h->d_nt_v : h->d_nt_h;
    dim3 grid ((unsigned) ((w + 63) / 64), (unsigned) ((hh + 3) / 4));
    if (vertical) hipLaunchKernelGGL (k_ntap_v, grid, dim3 (64, 4), 0, s, p);
    else hipLaunchKernelGGL (k_ntap_h, grid, dim3 (64, 4), 0, s, p);
  };
  uint8_t *dst = (uint8_t *) out->data[0];
  const int ds = out->stride[0];
  if (!h->nt_h && !h->nt_v) {                         // same size: the conversion is everything (or a plain copy)
    VFHIP_CHECK_HIP (hipMemcpy2DAsync (dst, (size_t) ds, src, (size_t) ss, (size_t) iw * 4, (size_t) ih, hipMemcpyDeviceToDevice, s));
  } else if (!h->nt_h) pass (true, src, ss, dst, ds, iw, oh);
  else if (!h->nt_v) pass (false, src, ss, dst, ds, ow, ih);
  else if (h->vfirst) {
    pass (true, src, ss, (uint8_t *) h->nt_mid1, iw * 4, iw, oh);
    pass (false, (const uint8_t *) h->nt_mid1, iw * 4, dst, ds, ow, oh);
  } else {
    pass (false, src, ss, (uint8_t *) h->nt_mid1, ow * 4, ow, ih);
    pass (true, (const uint8_t *) h->nt_mid1, ow * 4, dst, ds, ow, oh);
  }
  VFHIP_CHECK_HIP (hipGetLastError ());
  return VFHIP_OK;
}

static int launch_device (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out, size_t in_pitch, size_t out_pitch,
    int n_frames, hipStream_t s)
{
  if (n_frames <= 0) return VFHIP_OK;
  if (n_frames > 65535) return set_error (VFHIP_ERR_INVALID, "batch of %d frames exceeds 65535", n_frames);
  if (h->rgb_same && !(((uintptr_t) in->data[0] | (uintptr_t) in->stride[0] | (uintptr_t) in_pitch | (uintptr_t) out->data[0] | (uintptr_t) out->stride[0] | (uintptr_t) out_pitch) & 15)) {
    CsParams p {};
    p.in[0] = (const uint8_t *) in->data[0]; p.is[0] = in->stride[0]; p.out = (uint8_t *) out->data[0]; p.os = out->stride[0];
    p.in_pitch = in_pitch; p.out_pitch = out_pitch; p.in_w = h->in.width; p.in_h = h->in.height;
    dim3 grid ((unsigned) (((size_t) (p.in_w >> 2) * p.in_h + 255) / 256), (unsigned) n_frames);
    hipLaunchKernelGGL (k_cs_rgb_same, grid, dim3 (256), 0, s, p, h->in.format != h->out.format ? 1 : 0);
    VFHIP_CHECK_HIP (hipGetLastError ());
    return VFHIP_OK;
  }
  VfHipFrame rect_view;
  if (h->kernel == VfHipConvertScale::K_NTAP && (h->rw != h->out.width || h->rh != h->out.height)) {
    // add-borders: the border colour everywhere outside the rectangle, then the bicubic path writes the rectangle through a view
    RgbBorderParams b {};
    b.out = (uint8_t *) out->data[0]; b.os = out->stride[0]; b.pitch = out_pitch;
    b.w = h->out.width; b.h = h->out.height; b.rx = h->rx; b.ry = h->ry; b.rw = h->rw; b.rh = h->rh;
    b.colour = border_in_output_order (h->border_color, h->out.format);
    dim3 grid ((unsigned) ((b.w + 63) / 64), (unsigned) ((b.h + 3) / 4), (unsigned) n_frames);
    hipLaunchKernelGGL (k_border_fill_rgb, grid, dim3 (64, 4), 0, s, b);
    VFHIP_CHECK_HIP (hipGetLastError ());
    rect_view = *out;
    rect_view.data[0] = (uint8_t *) out->data[0] + (size_t) h->ry * out->stride[0] + 4 * (size_t) h->rx;
    out = &rect_view;
  }
  if (h->kernel == VfHipConvertScale::K_NTAP && h->nt_dot) {
    CubicDotParams t {};
    for (int k = 0; k < 3; k++) { t.cs.in[k] = (const uint8_t *) in->data[k]; t.cs.is[k] = in->stride[k]; }
    t.cs.in_pitch = in_pitch; t.cs.out_pitch = out_pitch;
    t.cs.in_w = h->in.width; t.cs.in_h = h->in.height; t.cs.in_fmt = h->in.format; t.cs.out_rgba = h->out.format == VFHIP_FORMAT_RGBA;
    for (int k = 0; k < 5; k++) t.cs.c[k] = kOrcCoef[h->in.color_matrix][k];
    t.cs.cosited = h->in.chroma_site == VFHIP_CHROMA_SITE_H_COSITED;
    t.out = (uint8_t *) out->data[0]; t.os = out->stride[0];
    t.ow = h->rw; t.oh = h->rh; t.vfirst = h->vfirst;
    t.win_h = h->d_win_h; t.win_v = h->d_win_v; t.wh = h->win_wh; t.wv = h->win_wv;
    {
      const uintptr_t a = (uintptr_t) t.cs.in[0] | (uintptr_t) t.cs.in[1] | (uintptr_t) t.cs.is[0] | (uintptr_t) t.cs.is[1] | (uintptr_t) in_pitch;
      t.fast_nv12 = h->in.format == VFHIP_FORMAT_NV12 && !(a & 7) && h->in.width >= 16 && getenv ("VFHIP_CUBIC_SCALAR") == nullptr;
    }
    t.tiles_x = (t.ow + CD_TW - 1) / CD_TW; t.tiles_y = (t.oh + CD_TH - 1) / CD_TH;
    const long long nt = (long long) t.tiles_x * t.tiles_y * n_frames;
    if (nt > 0x7fffff00ll) return set_error (VFHIP_ERR_INVALID, "bicubic: too many tiles in one batch");
    t.n_tiles = (int) nt; t.n_chunk = (t.n_tiles + 7) / 8;
    const bool opaque = h->in.format != VFHIP_FORMAT_BGRA && h->in.format != VFHIP_FORMAT_RGBA;
    if (opaque && t.fast_nv12 && !(t.cs.in_w & 7)) hipLaunchKernelGGL ((k_cs_cubic_dot<3, true>), dim3 ((unsigned) (8 * t.n_chunk)), dim3 (512), 0, s, t);
    else if (opaque) hipLaunchKernelGGL ((k_cs_cubic_dot<3, false>), dim3 ((unsigned) (8 * t.n_chunk)), dim3 (512), 0, s, t);
    else hipLaunchKernelGGL ((k_cs_cubic_dot<4, false>), dim3 ((unsigned) (8 * t.n_chunk)), dim3 (512), 0, s, t);
    VFHIP_CHECK_HIP (hipGetLastError ());
    return VFHIP_OK;
  }
  if (h->kernel == VfHipConvertScale::K_NTAP && h->nt_tile) {
    CubicTileParams t {};
    for (int k = 0; k < 3; k++) { t.cs.in[k] = (const uint8_t *) in->data[k]; t.cs.is[k] = in->stride[k]; }
    t.cs.in_pitch = in_pitch; t.cs.out_pitch = out_pitch;
    t.cs.in_w = h->in.width; t.cs.in_h = h->in.height; t.cs.in_fmt = h->in.format; t.cs.out_rgba = h->out.format == VFHIP_FORMAT_RGBA;
    for (int k = 0; k < 5; k++) t.cs.c[k] = kOrcCoef[h->in.color_matrix][k];
    t.cs.cosited = h->in.chroma_site == VFHIP_CHROMA_SITE_H_COSITED;
    t.out = (uint8_t *) out->data[0]; t.os = out->stride[0];
    t.ow = h->rw; t.oh = h->rh; t.nh = h->nt_h; t.nv = h->nt_v; t.vfirst = h->vfirst;
    t.tab_h = h->d_nt_h; t.tab_v = h->d_nt_v;
    {
      const uintptr_t a = (uintptr_t) t.cs.in[0] | (uintptr_t) t.cs.in[1] | (uintptr_t) t.cs.is[0] | (uintptr_t) t.cs.is[1] | (uintptr_t) in_pitch;
      t.fast_nv12 = h->in.format == VFHIP_FORMAT_NV12 && !(a & 7) && h->in.width >= 16 && getenv ("VFHIP_CUBIC_SCALAR") == nullptr;
    }
    t.tiles_x = (t.ow + CT_TW - 1) / CT_TW; t.tiles_y = (t.oh + CT_TH - 1) / CT_TH;
    const long long nt = (long long) t.tiles_x * t.tiles_y * n_frames;
    if (nt > 0x7fffff00ll) return set_error (VFHIP_ERR_INVALID, "bicubic: too many tiles in one batch");
    t.n_tiles = (int) nt; t.n_chunk = (t.n_tiles + 7) / 8;
    dim3 grid ((unsigned) (8 * t.n_chunk));
    // 512 lanes share one tile's 48 KB of LDS: 3 workgroups = 24 waves per CU (256 lanes: 12 waves, latency-bound)
    // (measured on C2 bicubic: 256 lanes 25.2 k frames/s, 512 lanes 28.6 k, 1024 lanes 19.6 k)
    // a source format without alpha converts to A = 255 everywhere: the tile kernel then filters three channels (ntap_accf)
    const bool opaque = h->in.format != VFHIP_FORMAT_BGRA && h->in.format != VFHIP_FORMAT_RGBA;
    if (opaque) hipLaunchKernelGGL ((k_cs_cubic_tile<512, true>), grid, dim3 (512), 0, s, t);
    else hipLaunchKernelGGL ((k_cs_cubic_tile<512, false>), grid, dim3 (512), 0, s, t);
    VFHIP_CHECK_HIP (hipGetLastError ());
    return VFHIP_OK;
  }
  if (h->kernel == VfHipConvertScale::K_NTAP) {
    for (int k = 0; k < n_frames; k++) {
      VfHipFrame fi = *in, fo = *out;
      for (int p = 0; p < 3; p++) {
        if (fi.data[p]) fi.data[p] = (uint8_t *) fi.data[p] + (size_t) k * in_pitch;
        if (fo.data[p]) fo.data[p] = (uint8_t *) fo.data[p] + (size_t) k * out_pitch;
      }
      int rc = ntap_launch (h, &fi, &fo, s);
      if (rc) return rc;
    }
    return VFHIP_OK;
  }
  if (h->kernel == VfHipConvertScale::K_STAGED) return staged_launch (h, in, out, in_pitch, out_pitch, n_frames, s);
  if (h->kernel == VfHipConvertScale::K_METAL)
    return cs_metal_launch (h->in, h->out, h->method, h->add_borders, h->border_color, in, out, in_pitch, out_pitch, n_frames, s);
  CsParams p {};
  for (int k = 0; k < 3; k++) { p.in[k] = (const uint8_t *) in->data[k]; p.is[k] = in->stride[k]; }
  p.out = (uint8_t *) out->data[0]; p.os = out->stride[0];
  p.in_pitch = in_pitch; p.out_pitch = out_pitch;
  p.in_w = h->in.width; p.in_h = h->in.height; p.out_w = h->out.width; p.out_h = h->out.height;
  p.rx = h->rx; p.ry = h->ry; p.rw = h->rw; p.rh = h->rh;
  p.in_fmt = h->in.format; p.out_rgba = h->out.format == VFHIP_FORMAT_RGBA;
  for (int k = 0; k < 5; k++) p.c[k] = kOrcCoef[h->in.color_matrix][k];
  p.cosited = h->in.chroma_site == VFHIP_CHROMA_SITE_H_COSITED;
  p.nearest = h->method == VFHIP_SCALE_NEAREST; p.vfirst = h->vfirst; p.hscale_on = h->hscale_on; p.hinc = h->hinc;
  p.vtab = h->d_vtab; p.htab = h->d_htab;
  p.border = border_in_output_order (h->border_color, h->out.format);

  bool half = h->kernel == VfHipConvertScale::K_HALF;
  if (half) {
    // vector-access preconditions of the fast path; otherwise the generic kernel computes the same bytes
    // 8-byte luma loads (both formats), 8-byte chroma loads from the NV12 UV plane, 4-byte ones from the I420 U / V planes,
    // 16-byte stores
    const bool i420 = p.in_fmt == VFHIP_FORMAT_I420;
    const uintptr_t luma = (uintptr_t) p.in[0] | (uintptr_t) p.is[0] | (uintptr_t) in_pitch;
    const uintptr_t chroma = (uintptr_t) p.in[1] | (uintptr_t) p.is[1] | (i420 ? (uintptr_t) p.in[2] | (uintptr_t) p.is[2] : 0);
    const uintptr_t b = (uintptr_t) p.out | (uintptr_t) p.os | (uintptr_t) out_pitch;
    if ((luma & 7) || (chroma & (i420 ? 3 : 7)) || (b & 15)) half = false;
  }
  bool same = h->kernel == VfHipConvertScale::K_SAME;
  if (same) {
    const bool i420 = p.in_fmt == VFHIP_FORMAT_I420, packed = p.in_fmt == VFHIP_FORMAT_UYVY || p.in_fmt == VFHIP_FORMAT_YUY2;
    const uintptr_t luma = (uintptr_t) p.in[0] | (uintptr_t) p.is[0] | (uintptr_t) in_pitch;
    const uintptr_t chroma = packed ? 0 : ((uintptr_t) p.in[1] | (uintptr_t) p.is[1] | (i420 ? (uintptr_t) p.in[2] | (uintptr_t) p.is[2] : 0));
    const uintptr_t b = (uintptr_t) p.out | (uintptr_t) p.os | (uintptr_t) out_pitch;
    if ((luma & (packed ? 15 : 7)) || (chroma & (i420 ? 3 : 7)) || (b & 15)) same = false;      // the generic kernels compute the same bytes
  }
  if (h->kernel == VfHipConvertScale::K_BLTILE) {
    // the 8-pixel converters' alignment contracts (k_cs_yuv_same); a frame that misses its format's converts its regions pixel by pixel
    const bool i420 = p.in_fmt == VFHIP_FORMAT_I420, packed = p.in_fmt == VFHIP_FORMAT_UYVY || p.in_fmt == VFHIP_FORMAT_YUY2;
    const uintptr_t luma = (uintptr_t) p.in[0] | (uintptr_t) p.is[0] | (uintptr_t) in_pitch;
    const uintptr_t chroma = packed ? 0 : ((uintptr_t) p.in[1] | (uintptr_t) p.is[1] | (i420 ? (uintptr_t) p.in[2] | (uintptr_t) p.is[2] : 0));
    int fast = 0;
    if ((p.in_fmt == VFHIP_FORMAT_NV12 || i420 || packed) && p.in_w >= 16 && !(luma & (packed ? 15 : 7)) && !(chroma & (i420 ? 3 : 7)) &&
        getenv ("VFHIP_CUBIC_SCALAR") == nullptr)
      fast = p.in_fmt == VFHIP_FORMAT_NV12 ? 1 : (i420 ? 2 : (p.in_fmt == VFHIP_FORMAT_UYVY ? 3 : 4));
    dim3 grid ((unsigned) ((p.out_w + CT_TW - 1) / CT_TW), (unsigned) ((p.out_h + h->bl_th - 1) / h->bl_th), (unsigned) n_frames);
    if (h->bl_th == 32) hipLaunchKernelGGL ((k_cs_bilinear_tile<512, 32>), grid, dim3 (512), 0, s, p, fast);
    else hipLaunchKernelGGL ((k_cs_bilinear_tile<512, 16>), grid, dim3 (512), 0, s, p, fast);
  } else if (half) {
    launch_half (p, n_frames, h->dev->n_cu, s);
  } else if (same) {
    dim3 grid ((unsigned) (((size_t) (p.in_w >> 3) * p.in_h + 255) / 256), (unsigned) n_frames);
    const int f = p.in_fmt == VFHIP_FORMAT_NV12 ? 0 : (p.in_fmt == VFHIP_FORMAT_I420 ? 1 : (p.in_fmt == VFHIP_FORMAT_UYVY ? 2 : 3));
#define VF_SAME(F) { if (p.cosited) hipLaunchKernelGGL ((k_cs_yuv_same<F, true>), grid, dim3 (256), 0, s, p); else hipLaunchKernelGGL ((k_cs_yuv_same<F, false>), grid, dim3 (256), 0, s, p); }
    if (f == 0) VF_SAME (0) else if (f == 1) hipLaunchKernelGGL ((k_cs_yuv_same<1, false>), grid, dim3 (256), 0, s, p); else if (f == 2) VF_SAME (2) else VF_SAME (3)
#undef VF_SAME
  } else {
    dim3 grid ((unsigned) ((p.out_w + 63) / 64), (unsigned) ((p.out_h + 3) / 4), (unsigned) n_frames);
    // window loads need >= 2 luma columns and >= 4 chroma pairs per row; tiny frames and nearest / RGB inputs use k_cs_generic
    const VfHipConvertScale::Kernel kk = h->kernel == VfHipConvertScale::K_SAME ? h->same_fallback : h->kernel;
    const bool taps = kk == VfHipConvertScale::K_TAPS && p.in_w >= 8;
    // strips of output rows per lane (k_cs_taps_strip) when the shape allows and the launch still fills the chip
    int rows = 1;
    if (taps && h->taps_adjacent && p.hscale_on && p.rx == 0 && p.ry == 0 && p.rw == p.out_w && p.rh == p.out_h) {
      rows = h->strip_rows;                   // (8-row strips measured the same as 4: profiles/r02x_taps_strip_ab.txt)
      if ((size_t) grid.x * ((p.out_h + rows - 1) / rows) * n_frames < (size_t) h->strip_fill * 4 * h->dev->n_cu) rows = 1;      // the launch must still give that many waves per SIMD
    }
    if (rows > 1) {
      // 1-D launch in XCD-aware block order (cs_xcd_block)
      p.xg_x = (int) grid.x; p.xg_y = ((p.out_h + rows - 1) / rows + 3) / 4;
      const long long nb = (long long) p.xg_x * p.xg_y * n_frames;
      if (nb > 0x7fffff00ll) return set_error (VFHIP_ERR_INVALID, "convertscale: too many blocks in one batch");
      p.xg_n = (int) nb; p.xg_chunk = (p.xg_n + 7) / 8;
      // n / d for n < 2^31 as (umulhi (m, n) + n) >> l with l = ceil (log2 d), m = floor (2^32 (2^l - d) / d) + 1 (Granlund & Montgomery)
      auto fastdiv = [] (uint32_t d, uint32_t &m, uint32_t &l) {
        l = 0; while ((1ull << l) < d) l++;
        m = (uint32_t) ((((1ull << l) - d) << 32) / d + 1);
      };
      fastdiv ((uint32_t) (p.xg_x * p.xg_y), p.xg_m_per, p.xg_l_per);
      fastdiv ((uint32_t) p.xg_x, p.xg_m_x, p.xg_l_x);
      dim3 sg ((unsigned) (8 * p.xg_chunk));
      const bool i420 = p.in_fmt == VFHIP_FORMAT_I420;
#define VF_STRIP(I, C, V) hipLaunchKernelGGL ((k_cs_taps_strip<I, C, V, 4>), sg, dim3 (64, 4), 0, s, p)
      if (i420) { if (p.vfirst) VF_STRIP (true, false, true); else VF_STRIP (true, false, false); }
      else if (p.cosited) { if (p.vfirst) VF_STRIP (false, true, true); else VF_STRIP (false, true, false); }
      else { if (p.vfirst) VF_STRIP (false, false, true); else VF_STRIP (false, false, false); }
#undef VF_STRIP
    }
    else if (taps && p.in_fmt == VFHIP_FORMAT_I420) hipLaunchKernelGGL (k_cs_taps<true>, grid, dim3 (64, 4), 0, s, p);
    else if (taps) hipLaunchKernelGGL (k_cs_taps<false>, grid, dim3 (64, 4), 0, s, p);
    else {
      // RGB -> RGB bilinear down-scales: k_cs_rgb_taps_strip under the same launch-size rule
      const bool rgb_in = p.in_fmt == VFHIP_FORMAT_BGRA || p.in_fmt == VFHIP_FORMAT_RGBA;
      const size_t strips = ((size_t) p.out_h + 3) / 4;
      if (rgb_in && !p.nearest && h->method == VFHIP_SCALE_BILINEAR && h->taps_adjacent && h->strip_rows > 1 && p.in_w >= 2 &&
          p.rx == 0 && p.ry == 0 && p.rw == p.out_w && p.rh == p.out_h && !(((uintptr_t) p.in[0] | (uintptr_t) p.is[0] | (uintptr_t) in_pitch) & 3) &&
          (size_t) grid.x * strips * n_frames >= (size_t) h->strip_fill * 4 * h->dev->n_cu) {
        dim3 sg (grid.x, (unsigned) ((strips + 3) / 4), (unsigned) n_frames);
        const int swap = (p.in_fmt == VFHIP_FORMAT_RGBA) != (p.out_rgba != 0) ? 1 : 0;
        if (p.vfirst) hipLaunchKernelGGL ((k_cs_rgb_taps_strip<true, 4>), sg, dim3 (64, 4), 0, s, p, swap);
        else hipLaunchKernelGGL ((k_cs_rgb_taps_strip<false, 4>), sg, dim3 (64, 4), 0, s, p, swap);
      }
      else hipLaunchKernelGGL (k_cs_generic, grid, dim3 (64, 4), 0, s, p);
    }
  }
  VFHIP_CHECK_HIP (hipGetLastError ());
  return VFHIP_OK;
}

int vfhip_convertscale_process_device_batch (VfHipConvertScale *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, void *stream)
{
  int rc = validate_frames (h, in0, out0);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return launch_device (h, in0, out0, in_frame_pitch, out_frame_pitch, n_frames, stream ? (hipStream_t) stream : h->st.s_compute);
}

int vfhip_convertscale_process_device (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out, void *stream)
{
  return vfhip_convertscale_process_device_batch (h, in, out, 0, 0, 1, stream);
}

int vfhip_convertscale_submit (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out)
{
  int rc = validate_frames (h, in, out);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return flights_submit (h->st, h->fl, &h->out, in, out,
      [h] (const VfHipFrame *di, VfHipFrame *dout, hipStream_t s) { return launch_device (h, di, dout, 0, 0, 1, s); });
}

int vfhip_convertscale_wait (VfHipConvertScale *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null handle");
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return flights_wait (h->st, h->fl);
}

int vfhip_convertscale_in_flight (VfHipConvertScale *h)
{
  if (!h) return 0;
  std::lock_guard<std::mutex> lk (h->mu);
  return h->fl.count;
}

int vfhip_convertscale_process (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out)
{
  int rc = validate_frames (h, in, out);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "frames submitted with vfhip_convertscale_submit are still in flight");
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  VfHipFrame din, dout;
  if ((rc = upload_frame (h->st, 0, in, &din))) return rc;                 // pinned staging + async H2D on the h2d stream
  if ((rc = output_frame (h->st, 1, &h->out, out, &dout))) return rc;
  VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if ((rc = launch_device (h, &din, &dout, 0, 0, 1, h->st.s_compute))) return rc;
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  return download_frame (h->st, 1, &dout, out);                              // async D2H on the d2h stream, then sync
}

void vfhip_convertscale_cleanup (VfHipConvertScale *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  flights_abandon (h->st, h->fl);
  free_tables (h);
  for (auto &b : h->st.slots) {
    if (b.host) (void) hipHostFree (b.host);
    if (b.devp) (void) hipFree (b.devp);
  }
  h->st.slots.clear ();
  h->configured = false;
}

void vfhip_convertscale_free (VfHipConvertScale *h)
{
  if (!h) return;
  vfhip_convertscale_cleanup (h);
  h->st.destroy ();
  delete h;
}

}  // extern "C"
