// csrc/transform.hip — vfhip_transform_* : flip / rotate (8 methods) + crop (SURVEY.md §8f "next" item 2).
// Mirrors MetalTransformRenderer (reference transform/metaltransformrenderer.{h,m}); restates transformVertex /
// transformFragment{RGBA,NV12,I420} (transform/metaltransform_shaders.h:40-120) and the crop-into-UV-matrix folding
// (metaltransformrenderer.m:265-293) in `metal` numerics.  One kernel: sample at the transformed texcoord, out-of-range
// -> opaque black, 8-bit target, store epilogue for any of the four formats (the reference's RGBA->YUV pass fused).
#include "vfhip_internal.h"
#include "metal_common.h"

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

}  // namespace vfhip

struct VfHipTransform {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo in {}, out {};
  Flights fl;                       // pipelined host path (submit / wait)
};

// UV matrix of the eight methods, column-major [m00 m10 m01 m11]
static const float kTransformMat[8][4] = {
  {  1,  0,  0,  1 }, {  0, -1,  1,  0 }, { -1,  0,  0, -1 }, {  0,  1, -1,  0 },
  { -1,  0,  0,  1 }, {  1,  0,  0, -1 }, {  0,  1,  1,  0 }, {  0, -1, -1,  0 },
};

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
