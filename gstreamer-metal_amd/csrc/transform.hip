// csrc/transform.hip — vfhip_transform_* : flip / rotate (8 methods) + crop (SURVEY.md §8f "next" item 2).
// Mirrors MetalTransformRenderer (reference transform/metaltransformrenderer.{h,m}); restates transformVertex /
// transformFragment{RGBA,NV12,I420} (transform/metaltransform_shaders.h:40-120) and the crop-into-UV-matrix folding
// (metaltransformrenderer.m:265-293) in `metal` numerics.  One kernel: sample at the transformed texcoord, out-of-range
// -> opaque black, 8-bit target, store epilogue for any of the four formats (the reference's RGBA->YUV pass fused).
#include "vfhip_internal.h"
#include "metal_common.h"
#include <algorithm>
#include <cmath>
#include <vector>

using namespace vfhip;

namespace vfhip {

struct TransformKParams {
  metal::Img in;
  metal::OutImg out;
  float m0, m1, m2, m3, offx, offy;
  size_t in_pitch, out_pitch;      // batch: frame blockIdx.z at base + z * pitch
};

__global__ __launch_bounds__ (256) void k_transform (const TransformKParams pp)
{
  TransformKParams p = pp;
  p.in = metal::img_at (pp.in, blockIdx.z * pp.in_pitch); p.out = metal::out_at (pp.out, blockIdx.z * pp.out_pitch);
  const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
  if (2 * bx >= p.out.w || 2 * by >= p.out.h) return;
  uint32_t q[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++) {
      const int x = min (2 * bx + dx, p.out.w - 1), y = min (2 * by + dy, p.out.h - 1);
      float tx = ((float) x + 0.5f) / (float) p.out.w, ty = ((float) y + 0.5f) / (float) p.out.h;
      tx -= 0.5f; ty -= 0.5f;
      const float ux = p.m0 * tx + p.m2 * ty, uy = p.m1 * tx + p.m3 * ty;
      tx = ux + (0.5f + p.offx); ty = uy + (0.5f + p.offy);
      if (tx < 0.0f || tx > 1.0f || ty < 0.0f || ty > 1.0f) q[dy][dx] = 0xff000000u;
      else q[dy][dx] = metal::quant_rgba8 (metal::sample_rgba (p.in, tx, ty, true));
    }
  metal::store_block (p.out, bx, by, q);
}

// k_transform_perm: an RGB frame, no crop, one of the eight methods — when the host has PROVED (tr_build_perm) that every tap of the sampler lands
// within 4e-4 of a texel centre, the four-tap interpolation rounds back to that texel in 8 bits (the two lerps move the value by at most
// 2 * 4e-4 of the 0..1 range = 0.2 LSB), so the frame is a permutation of the input's pixels: src = (A[x], B[y]), or (B[y], A[x]) for the methods
// that swap the axes.  Four pixels of a row per lane, one 16-byte store.  k_transform computes the same bytes with four gathers, sixteen
// conversions and twelve interpolations per pixel (6.7 us per 1080p frame).
struct TransformPermParams {
  const uint8_t *in; uint8_t *out;
  int is, os, ow, oh, swap_axes, swap_rb;
  const int *a, *b;                 // a[ow]: per output column, b[oh]: per output row
  size_t in_pitch, out_pitch;
};

__global__ __launch_bounds__ (256) void k_transform_perm (const TransformPermParams p)
{
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  const int xq = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (4 * xq >= p.ow || y >= p.oh) return;
  const uint8_t *in = p.in + (size_t) blockIdx.z * p.in_pitch;
  const int4 ax = *(reinterpret_cast<const int4 *> (p.a) + xq);
  const int by = p.b[y];
  const int a[4] = { ax.x, ax.y, ax.z, ax.w };
  v4u v;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int sx = p.swap_axes ? by : a[k], sy = p.swap_axes ? a[k] : by;
    uint32_t t = *reinterpret_cast<const uint32_t *> (in + (size_t) sy * p.is + 4 * (size_t) sx);
    if (p.swap_rb) t = __builtin_amdgcn_perm (0u, t, 0x03000102u);
    v[k] = t;
  }
  __builtin_nontemporal_store (v, reinterpret_cast<v4u *> (p.out + (size_t) blockIdx.z * p.out_pitch + (size_t) y * p.os) + xq);
}

}  // namespace vfhip

struct VfHipTransform {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo in {}, out {};
  Flights fl;                       // pipelined host path (submit / wait)
  // k_transform_perm: per method the two index tables (a[out_w] then b[out_h], ints) in ONE device buffer built at configure, or nothing when the
  // proof fails for that method (then, and with any crop, k_transform runs)
  int *d_perm = nullptr;
  bool perm_ok[8] = { false, false, false, false, false, false, false, false };
};

// UV matrix of the eight methods, column-major [m00 m10 m01 m11]
static const float kTransformMat[8][4] = {
  {  1,  0,  0,  1 }, {  0, -1,  1,  0 }, { -1,  0,  0, -1 }, {  0,  1, -1,  0 },
  { -1,  0,  0,  1 }, {  1,  0,  0, -1 }, {  0,  1,  1,  0 }, {  0, -1, -1,  0 },
};

// Proof and tables for k_transform_perm, method m, no crop.  The per-pixel expressions of k_transform are evaluated here exactly as the kernel
// evaluates them (same float operations in the same order; this file is compiled with -ffp-contract=off for host and device alike).  Without a
// crop the UV matrix has one non-zero entry per row, so the sampler's u depends on the output column only (or, for the four methods that swap
// the axes, on the output row only) — the product with the zero entry is +-0 and does not change the sum — and likewise v.
static bool tr_build_perm (int m, int iw, int ih, int ow, int oh, std::vector<int> &a, std::vector<int> &b)
{
  const float *t = kTransformMat[m];
  const float m0 = t[0] * 1.0f, m1 = t[1] * 1.0f, m2 = t[2] * 1.0f, m3 = t[3] * 1.0f;
  const float offx = t[0] * 0.0f + t[2] * 0.0f + 0.0f, offy = t[1] * 0.0f + t[3] * 0.0f + 0.0f;
  const bool swap = t[0] == 0.0f;
  const float eps = 4e-4f;
  auto centre = [] (int i, int n) { float c = ((float) i + 0.5f) / (float) n; c -= 0.5f; return c; };
  auto tap = [&] (float coord, int n, int *idx) {            // metal::lin_taps; the texel the interpolation rounds back to
    if (coord < 0.0f || coord > 1.0f) return false;          // (k_transform paints such a pixel black)
    const float x = coord * (float) n - 0.5f, fl = floorf (x), f = x - fl;
    const int i = (int) fl, i0 = std::min (std::max (i, 0), n - 1), i1 = std::min (std::max (i + 1, 0), n - 1);
    if (i0 == i1 || f <= eps) { *idx = i0; return true; }
    if (f >= 1.0f - eps) { *idx = i1; return true; }
    return false;
  };
  a.assign ((size_t) ow, 0); b.assign ((size_t) oh, 0);
  // the other axis' centre enters multiplied by zero: evaluate with both of its extremes and insist on one answer
  const float tx_lo = centre (0, ow), tx_hi = centre (ow - 1, ow), ty_lo = centre (0, oh), ty_hi = centre (oh - 1, oh);
  for (int x = 0; x < ow; x++) {
    const float tx = centre (x, ow);
    int i_lo, i_hi;
    if (!swap) {                                              // u (x) -> source column
      const float u_lo = (m0 * tx + m2 * ty_lo) + (0.5f + offx), u_hi = (m0 * tx + m2 * ty_hi) + (0.5f + offx);
      if (!tap (u_lo, iw, &i_lo) || !tap (u_hi, iw, &i_hi) || i_lo != i_hi) return false;
    } else {                                                  // v (x) -> source row
      const float v_lo = (m1 * tx + m3 * ty_lo) + (0.5f + offy), v_hi = (m1 * tx + m3 * ty_hi) + (0.5f + offy);
      if (!tap (v_lo, ih, &i_lo) || !tap (v_hi, ih, &i_hi) || i_lo != i_hi) return false;
    }
    a[(size_t) x] = i_lo;
  }
  for (int y = 0; y < oh; y++) {
    const float ty = centre (y, oh);
    int i_lo, i_hi;
    if (!swap) {                                              // v (y) -> source row
      const float v_lo = (m1 * tx_lo + m3 * ty) + (0.5f + offy), v_hi = (m1 * tx_hi + m3 * ty) + (0.5f + offy);
      if (!tap (v_lo, ih, &i_lo) || !tap (v_hi, ih, &i_hi) || i_lo != i_hi) return false;
    } else {                                                  // u (y) -> source column
      const float u_lo = (m0 * tx_lo + m2 * ty) + (0.5f + offx), u_hi = (m0 * tx_hi + m2 * ty) + (0.5f + offx);
      if (!tap (u_lo, iw, &i_lo) || !tap (u_hi, iw, &i_hi) || i_lo != i_hi) return false;
    }
    b[(size_t) y] = i_lo;
  }
  return true;
}

static int tr_launch (VfHipTransform *h, const VfHipFrame *in, VfHipFrame *out, const VfHipTransformParams *prm, hipStream_t s,
    int n_frames = 1, size_t in_pitch = 0, size_t out_pitch = 0)
{
  TransformKParams p {};
  p.in_pitch = in_pitch; p.out_pitch = out_pitch;
  p.in = metal::make_img (in); p.out = metal::make_out (out);
  const float cl = (float) prm->crop_left / (float) h->in.width, cr = (float) prm->crop_right / (float) h->in.width;
  const float ct = (float) prm->crop_top / (float) h->in.height, cb = (float) prm->crop_bottom / (float) h->in.height;
  const float sx = 1.0f - cl - cr, sy = 1.0f - ct - cb, ox = (cl - cr) * 0.5f, oy = (ct - cb) * 0.5f;
  const float *t = kTransformMat[prm->method & 7];
  p.m0 = t[0] * sx; p.m1 = t[1] * sx; p.m2 = t[2] * sy; p.m3 = t[3] * sy;
  p.offx = t[0] * ox + t[2] * oy + 0.0f; p.offy = t[1] * ox + t[3] * oy + 0.0f;
  const int m = prm->method & 7;
  if (h->d_perm && h->perm_ok[m] && !(prm->crop_left | prm->crop_right | prm->crop_top | prm->crop_bottom) && getenv ("VFHIP_TR_GENERAL") == nullptr &&
      !(((uintptr_t) in->data[0] | (uintptr_t) in->stride[0] | (uintptr_t) in_pitch) & 3) &&
      !(((uintptr_t) out->data[0] | (uintptr_t) out->stride[0] | (uintptr_t) out_pitch) & 15)) {
    TransformPermParams q {};
    q.in = (const uint8_t *) in->data[0]; q.out = (uint8_t *) out->data[0]; q.is = in->stride[0]; q.os = out->stride[0];
    q.ow = h->out.width; q.oh = h->out.height; q.swap_axes = kTransformMat[m][0] == 0.0f; q.swap_rb = h->in.format != h->out.format;
    q.a = h->d_perm + (size_t) m * ((size_t) q.ow + (((size_t) q.oh + 3) & ~(size_t) 3)); q.b = q.a + q.ow;
    q.in_pitch = in_pitch; q.out_pitch = out_pitch;
    dim3 grid ((unsigned) ((q.ow / 4 + 63) / 64), (unsigned) ((q.oh + 3) / 4), (unsigned) n_frames);
    hipLaunchKernelGGL (k_transform_perm, grid, dim3 (64, 4), 0, s, q);
    VFHIP_CHECK_HIP (hipGetLastError ());
    return VFHIP_OK;
  }
  // (4 x 2 pixels per lane with metal::store_quad's wide stores was tried here as in the filter and the overlay: byte-identical and SLOWER — 8.7 vs
  // 6.7 us per BGRA 1080p frame, 8.1 vs 7.5 NV12: the four-tap sampler's gathers dominate, and lanes four pixels apart spread them further)
  const int bw = (h->out.width + 1) / 2, bh = (h->out.height + 1) / 2;
  dim3 grid ((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4), (unsigned) n_frames);
  hipLaunchKernelGGL (k_transform, grid, dim3 (64, 4), 0, s, p);
  VFHIP_CHECK_HIP (hipGetLastError ());
  return VFHIP_OK;
}

static int tr_check (VfHipTransform *h, const VfHipFrame *in, const VfHipFrame *out, const VfHipTransformParams *prm)
{
  if (!h || !prm) return set_error (VFHIP_ERR_INVALID, "null argument");
  if (!h->configured) return set_error (VFHIP_ERR_NOT_CONFIGURED, "transform: process before configure");
  if (prm->method < 0 || prm->method > 7) return set_error (VFHIP_ERR_INVALID, "bad transform method %d", prm->method);
  if (prm->crop_top < 0 || prm->crop_bottom < 0 || prm->crop_left < 0 || prm->crop_right < 0)
    return set_error (VFHIP_ERR_INVALID, "negative crop");
  int rc = check_frame (in, &h->in, "input");
  if (rc) return rc;
  return check_frame (out, &h->out, "output");
}

extern "C" {

VfHipTransform *vfhip_transform_new (int device)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  VfHipTransform *h = new (std::nothrow) VfHipTransform ();
  if (!h) { set_error (VFHIP_ERR_NOMEM, "out of memory"); return nullptr; }
  h->dev = d;
  if (h->st.init (d) != VFHIP_OK) { delete h; return nullptr; }
  return h;
}

int vfhip_transform_configure (VfHipTransform *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out)
{
  if (!h || !in || !out) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "configure with %d submitted frame(s) still in flight: wait for them first", h->fl.count);
  if (in->width <= 0 || in->height <= 0 || in->width > 32768 || in->height > 32768 || out->width <= 0 || out->height <= 0 ||
      out->width > 32768 || out->height > 32768)
    return set_error (VFHIP_ERR_INVALID, "bad frame size");
  if (in->format < VFHIP_FORMAT_BGRA || in->format > VFHIP_FORMAT_I420 || out->format < VFHIP_FORMAT_BGRA || out->format > VFHIP_FORMAT_I420)
    return set_error (VFHIP_ERR_UNSUPPORTED, "transform: format not supported");
  // a failed configure leaves the handle unconfigured: nothing is published before the last step that can fail has succeeded
  h->configured = false;
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  if (h->d_perm) { (void) hipStreamSynchronize (h->st.s_compute); (void) hipFree (h->d_perm); h->d_perm = nullptr; }
  for (bool &ok : h->perm_ok) ok = false;
  // RGB frames whose rows take four pixels per lane: the permutation tables of the methods for which the proof holds (tr_build_perm)
  const bool rgb = (in->format == VFHIP_FORMAT_BGRA || in->format == VFHIP_FORMAT_RGBA) && (out->format == VFHIP_FORMAT_BGRA || out->format == VFHIP_FORMAT_RGBA);
  if (rgb && !(out->width & 3)) {
    const size_t per = (size_t) out->width + (((size_t) out->height + 3) & ~(size_t) 3);          // (both parts multiples of 4 ints: every method's a[] starts 16-byte aligned)
    std::vector<int> all (8 * per, 0), a, b;
    bool ok[8] = { false, false, false, false, false, false, false, false }, any = false;
    for (int m = 0; m < 8; m++) {
      // (a method that swaps the axes maps a W x H frame onto H x W; the reference scales whatever the sizes are — only the provable cases come here)
      if (!tr_build_perm (m, in->width, in->height, out->width, out->height, a, b)) continue;
      std::copy (a.begin (), a.end (), all.begin () + (size_t) m * per);
      std::copy (b.begin (), b.end (), all.begin () + (size_t) m * per + (size_t) out->width);
      ok[m] = any = true;
    }
    if (any) {
      int *d_perm = nullptr;
      VFHIP_CHECK_HIP (dev_malloc (&d_perm, all.size () * sizeof (int)));
      const hipError_t e = upload_in_stream (d_perm, all.data (), all.size () * sizeof (int), h->st.s_compute);
      if (e != hipSuccess) {                       // a table that may hold anything must never reach k_transform_perm (its entries are source indices)
        (void) hipFree (d_perm);
        return set_error (VFHIP_ERR_HIP, "transform: uploading the permutation tables failed: %s", hipGetErrorString (e));
      }
      h->d_perm = d_perm;
      for (int m = 0; m < 8; m++) h->perm_ok[m] = ok[m];
    }
  }
  h->in = *in; h->out = *out; h->configured = true;
  return VFHIP_OK;
}

int vfhip_transform_process (VfHipTransform *h, const VfHipFrame *in, VfHipFrame *out, const VfHipTransformParams *prm)
{
  int rc = tr_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "frames submitted with vfhip_transform_submit are still in flight");
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  VfHipFrame din, dout;
  if ((rc = upload_frame (h->st, 0, in, &din))) return rc;
  if ((rc = output_frame (h->st, 1, &h->out, out, &dout))) return rc;
  VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if ((rc = tr_launch (h, &din, &dout, prm, h->st.s_compute))) return rc;
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  return download_frame (h->st, 1, &dout, out);
}

int vfhip_transform_process_device (VfHipTransform *h, const VfHipFrame *in, VfHipFrame *out, const VfHipTransformParams *prm, void *stream)
{
  int rc = tr_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return tr_launch (h, in, out, prm, stream ? (hipStream_t) stream : h->st.s_compute);
}

int vfhip_transform_submit (VfHipTransform *h, const VfHipFrame *in, VfHipFrame *out, const VfHipTransformParams *prm)
{
  int rc = tr_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  const VfHipTransformParams p = *prm;
  return flights_submit (h->st, h->fl, &h->out, in, out,
      [h, &p] (const VfHipFrame *di, VfHipFrame *dout, hipStream_t s) { return tr_launch (h, di, dout, &p, s); });
}

int vfhip_transform_wait (VfHipTransform *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null handle");
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return flights_wait (h->st, h->fl);
}

int vfhip_transform_in_flight (VfHipTransform *h)
{
  if (!h) return 0;
  std::lock_guard<std::mutex> lk (h->mu);
  return h->fl.count;
}

int vfhip_transform_process_device_batch (VfHipTransform *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, const VfHipTransformParams *prm, void *stream)
{
  int rc = tr_check (h, in0, out0, prm);
  if (rc) return rc;
  if (n_frames < 1 || n_frames > 65535) return set_error (VFHIP_ERR_INVALID, "n_frames %d outside 1..65535", n_frames);
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return tr_launch (h, in0, out0, prm, stream ? (hipStream_t) stream : h->st.s_compute, n_frames, in_frame_pitch, out_frame_pitch);
}

void vfhip_transform_cleanup (VfHipTransform *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  flights_abandon (h->st, h->fl);
  for (auto &b : h->st.slots) { if (b.host) (void) hipHostFree (b.host); if (b.devp) (void) hipFree (b.devp); }
  h->st.slots.clear ();
  if (h->d_perm) { (void) hipFree (h->d_perm); h->d_perm = nullptr; }
  h->configured = false;
}

void vfhip_transform_free (VfHipTransform *h)
{
  if (!h) return;
  vfhip_transform_cleanup (h);
  h->st.destroy ();
  delete h;
}

}  // extern "C"
