// csrc/overlay.hip — vfhip_overlay_* : a still image (logo / watermark) blended over the video.
// Mirrors MetalOverlayRenderer (reference overlay/metaloverlayrenderer.{h,m}) and restates overlayFragmentRGBA / NV12 / I420
// (overlay/metaloverlay_shaders.h:60-151, `metal` numerics): the video is sampled 1:1, the image bilinearly inside its
// rectangle, rgb = mix (video.rgb, image.rgb, image.a * alpha), one 8-bit render target, then the output format.  The
// reference's render pass + RGBA->YUV pass are one kernel here (2x2 pixel blocks per lane, metal::store_block).
// The image comes from csrc/host_parsers.hip (PNG) or csrc/host_jpeg.hip (JPEG), chosen by the file's first bytes like the
// reference's ImageIO loader does, and is premultiplied on load because the reference's decoder does so (metaloverlayrenderer.m:214-219) and
// its shader then mixes the premultiplied colour as if it were straight — kept, it is what the reference renders.
#include "vfhip_internal.h"
#include "metal_common.h"

using namespace vfhip;

namespace vfhip {

struct OverlayKParams {
  metal::Img in, ov;               // ov.p[0] == nullptr: no image
  metal::OutImg out;
  float x, y, w, h, alpha;
  size_t in_pitch, out_pitch;      // batch: frame blockIdx.z at base + z * pitch
};

__global__ __launch_bounds__ (256) void k_overlay (const OverlayKParams pp)
{
  OverlayKParams p = pp;
  p.in = metal::img_at (pp.in, blockIdx.z * pp.in_pitch); p.out = metal::out_at (pp.out, blockIdx.z * pp.out_pitch);
  const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
  if (2 * bx >= p.out.w || 2 * by >= p.out.h) return;
  uint32_t q[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++) {
      const int x = min (2 * bx + dx, p.out.w - 1), y = min (2 * by + dy, p.out.h - 1);
      metal::F4 v = metal::fetch_1to1 (p.in, x, y, true);
      if (p.ov.p[0]) {
        const float tu = ((float) x + 0.5f) / (float) p.out.w, tv = ((float) y + 0.5f) / (float) p.out.h;
        const float px = tu * (float) p.out.w, py = tv * (float) p.out.h;
        if (px >= p.x && px < p.x + p.w && py >= p.y && py < p.y + p.h) {
          const metal::F4 o = metal::sample_rgba (p.ov, (px - p.x) / p.w, (py - p.y) / p.h, true);
          const float a = o.a * p.alpha;
          v.r = v.r + (o.r - v.r) * a; v.g = v.g + (o.g - v.g) * a; v.b = v.b + (o.b - v.b) * a;
        }
      }
      q[dy][dx] = metal::quant_rgba8 (v);
    }
  metal::store_block (p.out, bx, by, q);
}

// k_overlay_quad: the same per-pixel operations with 4 x 2 pixels per lane (metal::fetch_quad / store_quad: window loads for a 4:2:0 input's
// linear chroma, 16-byte RGB rows, dword luma / chroma stores) for frames that meet their alignment contract — NV12 1080p 8.0 -> see DESIGN §5.3.
// Only the lanes under the image sample it (a per-lane branch like k_overlay's).
__global__ __launch_bounds__ (256) void k_overlay_quad (const OverlayKParams pp)
{
  OverlayKParams p = pp;
  p.in = metal::img_at (pp.in, blockIdx.z * pp.in_pitch); p.out = metal::out_at (pp.out, blockIdx.z * pp.out_pitch);
  const int xq = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
  if (4 * xq >= p.out.w || 2 * by >= p.out.h) return;
  metal::F4 c[2][4];
  metal::fetch_quad (p.in, xq, by, c);
  uint32_t q[2][4];
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 4; dx++) {
      const int x = 4 * xq + dx, y = 2 * by + dy;
      metal::F4 v = c[dy][dx];
      if (p.ov.p[0]) {
        const float tu = ((float) x + 0.5f) / (float) p.out.w, tv = ((float) y + 0.5f) / (float) p.out.h;
        const float px = tu * (float) p.out.w, py = tv * (float) p.out.h;
        if (px >= p.x && px < p.x + p.w && py >= p.y && py < p.y + p.h) {
          const metal::F4 o = metal::sample_rgba (p.ov, (px - p.x) / p.w, (py - p.y) / p.h, true);
          const float a = o.a * p.alpha;
          v.r = v.r + (o.r - v.r) * a; v.g = v.g + (o.g - v.g) * a; v.b = v.b + (o.b - v.b) * a;
        }
      }
      q[dy][dx] = metal::quant_rgba8 (v);
    }
  metal::store_quad (p.out, xq, by, q);
}

}  // namespace vfhip

struct VfHipOverlay {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo in {}, out {};
  Flights fl;                       // pipelined host path (submit / wait)
  uint8_t *d_img = nullptr;        // RGBA8, tight rows
  int img_w = 0, img_h = 0;
};

static int ov_launch (VfHipOverlay *h, const VfHipFrame *in, VfHipFrame *out, const VfHipOverlayParams *prm, hipStream_t s,
    int n_frames = 1, size_t in_pitch = 0, size_t out_pitch = 0)
{
  OverlayKParams p {};
  p.in = metal::make_img (in); p.out = metal::make_out (out);
  p.in_pitch = in_pitch; p.out_pitch = out_pitch;
  if (h->d_img) {
    p.ov.p[0] = h->d_img; p.ov.s[0] = h->img_w * 4; p.ov.w = h->img_w; p.ov.h = h->img_h; p.ov.fmt = VFHIP_FORMAT_RGBA;
    p.x = prm->x; p.y = prm->y; p.alpha = prm->alpha;
    p.w = prm->width > 0.0f ? prm->width : (float) h->img_w;        // 0 = the image's own size (metaloverlayrenderer.m:268-269)
    p.h = prm->height > 0.0f ? prm->height : (float) h->img_h;
  }
  if (getenv ("VFHIP_OV_BLOCKS") == nullptr && metal::quad_frame_ok (in, in_pitch, false) && metal::quad_frame_ok (out, out_pitch, true)) {      // (knob: A/B and tests)
    dim3 grid ((unsigned) ((h->out.width / 4 + 63) / 64), (unsigned) ((h->out.height / 2 + 3) / 4), (unsigned) n_frames);
    hipLaunchKernelGGL (k_overlay_quad, grid, dim3 (64, 4), 0, s, p);
  } else {
    const int bw = (h->out.width + 1) / 2, bh = (h->out.height + 1) / 2;
    dim3 grid ((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4), (unsigned) n_frames);
    hipLaunchKernelGGL (k_overlay, grid, dim3 (64, 4), 0, s, p);
  }
  VFHIP_CHECK_HIP (hipGetLastError ());
  return VFHIP_OK;
}

static int ov_check (VfHipOverlay *h, const VfHipFrame *in, const VfHipFrame *out, const VfHipOverlayParams *prm)
{
  if (!h || !prm) return set_error (VFHIP_ERR_INVALID, "null argument");
  if (!h->configured) return set_error (VFHIP_ERR_NOT_CONFIGURED, "overlay: process before configure");
  int rc = check_frame (in, &h->in, "input");
  if (rc) return rc;
  return check_frame (out, &h->out, "output");
}

static void ov_drop_image (VfHipOverlay *h)
{
  (void) hipSetDevice (h->dev->ordinal);
  (void) hipStreamSynchronize (h->st.s_compute);       // a frame in flight may still read it
  if (h->d_img) (void) hipFree (h->d_img);
  h->d_img = nullptr; h->img_w = h->img_h = 0;
}

extern "C" {

VfHipOverlay *vfhip_overlay_new (int device)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  VfHipOverlay *h = new (std::nothrow) VfHipOverlay ();
  if (!h) { set_error (VFHIP_ERR_NOMEM, "out of memory"); return nullptr; }
  h->dev = d;
  if (h->st.init (d) != VFHIP_OK) { delete h; return nullptr; }
  return h;
}

int vfhip_overlay_configure (VfHipOverlay *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out)
{
  if (!h || !in || !out) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "configure with %d submitted frame(s) still in flight: wait for them first", h->fl.count);
  if (in->width <= 0 || in->height <= 0 || in->width > 32768 || in->height > 32768 || in->width != out->width || in->height != out->height)
    return set_error (VFHIP_ERR_INVALID, "overlay: bad or differing frame sizes");
  if (in->format < VFHIP_FORMAT_BGRA || in->format > VFHIP_FORMAT_I420 || out->format < VFHIP_FORMAT_BGRA || out->format > VFHIP_FORMAT_I420)
    return set_error (VFHIP_ERR_UNSUPPORTED, "overlay: format not supported");
  h->in = *in; h->out = *out; h->configured = true;
  return VFHIP_OK;
}

int vfhip_overlay_set_image (VfHipOverlay *h, const uint8_t *rgba, int width, int height)
{
  if (!h || !rgba || width <= 0 || height <= 0 || width > 16384 || height > 16384) return set_error (VFHIP_ERR_INVALID, "bad image");
  std::lock_guard<std::mutex> lk (h->mu);
  ov_drop_image (h);
  const size_t bytes = (size_t) width * height * 4;
  VFHIP_CHECK_HIP (dev_malloc (&h->d_img, bytes + 256));
  hipError_t e = upload_in_stream (h->d_img, rgba, bytes, h->st.s_compute);
  if (e != hipSuccess) { (void) hipFree (h->d_img); h->d_img = nullptr; return set_error (VFHIP_ERR_HIP, "image upload failed: %s", hipGetErrorString (e)); }
  h->img_w = width; h->img_h = height;
  return VFHIP_OK;
}

int vfhip_overlay_load_image (VfHipOverlay *h, const char *path)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null argument");
  if (!path || !*path) { vfhip_overlay_clear_image (h); return VFHIP_OK; }       // -loadImageFromFile: with an empty path clears
  std::vector<uint8_t> px;
  int w = 0, hh = 0;
  int rc = decode_image (path, px, &w, &hh);       // PNG or JPEG
  if (rc) return rc;
  // premultiplied, like the bytes CoreGraphics hands the reference (kCGImageAlphaPremultipliedLast); its exact rounding is
  // unpinned — round to nearest here
  for (size_t i = 0; i < px.size (); i += 4) {
    const unsigned a = px[i + 3];
    px[i] = (uint8_t) ((px[i] * a + 127) / 255); px[i + 1] = (uint8_t) ((px[i + 1] * a + 127) / 255); px[i + 2] = (uint8_t) ((px[i + 2] * a + 127) / 255);
  }
  return vfhip_overlay_set_image (h, px.data (), w, hh);
}

void vfhip_overlay_clear_image (VfHipOverlay *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  ov_drop_image (h);
}

int vfhip_overlay_image_size (VfHipOverlay *h, int *width, int *height)
{
  if (!h) return 0;
  std::lock_guard<std::mutex> lk (h->mu);
  if (width) *width = h->img_w;
  if (height) *height = h->img_h;
  return h->d_img != nullptr;
}

int vfhip_overlay_process (VfHipOverlay *h, const VfHipFrame *in, VfHipFrame *out, const VfHipOverlayParams *prm)
{
  int rc = ov_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "frames submitted with vfhip_overlay_submit are still in flight");
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  VfHipFrame din, dout;
  if ((rc = upload_frame (h->st, 0, in, &din))) return rc;
  if ((rc = output_frame (h->st, 1, &h->out, out, &dout))) return rc;
  VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if ((rc = ov_launch (h, &din, &dout, prm, h->st.s_compute))) return rc;
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  return download_frame (h->st, 1, &dout, out);
}

int vfhip_overlay_submit (VfHipOverlay *h, const VfHipFrame *in, VfHipFrame *out, const VfHipOverlayParams *prm)
{
  int rc = ov_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  const VfHipOverlayParams p = *prm;
  return flights_submit (h->st, h->fl, &h->out, in, out,
      [h, &p] (const VfHipFrame *di, VfHipFrame *dout, hipStream_t s) { return ov_launch (h, di, dout, &p, s); });
}

int vfhip_overlay_wait (VfHipOverlay *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null handle");
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return flights_wait (h->st, h->fl);
}

int vfhip_overlay_in_flight (VfHipOverlay *h)
{
  if (!h) return 0;
  std::lock_guard<std::mutex> lk (h->mu);
  return h->fl.count;
}

int vfhip_overlay_process_device_batch (VfHipOverlay *h, const VfHipFrame *in0, VfHipFrame *out0, size_t in_frame_pitch, size_t out_frame_pitch,
    int n_frames, const VfHipOverlayParams *prm, void *stream)
{
  int rc = ov_check (h, in0, out0, prm);
  if (rc) return rc;
  if (n_frames < 1 || n_frames > 65535) return set_error (VFHIP_ERR_INVALID, "n_frames %d outside 1..65535", n_frames);
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return ov_launch (h, in0, out0, prm, stream ? (hipStream_t) stream : h->st.s_compute, n_frames, in_frame_pitch, out_frame_pitch);
}

int vfhip_overlay_process_device (VfHipOverlay *h, const VfHipFrame *in, VfHipFrame *out, const VfHipOverlayParams *prm, void *stream)
{
  return vfhip_overlay_process_device_batch (h, in, out, 0, 0, 1, prm, stream);
}

void vfhip_overlay_cleanup (VfHipOverlay *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  flights_abandon (h->st, h->fl);
  for (auto &b : h->st.slots) { if (b.host) (void) hipHostFree (b.host); if (b.devp) (void) hipFree (b.devp); }
  h->st.slots.clear ();
  h->configured = false;               // the image survives, like the reference's texture (cleanup drops frame resources only)
}

void vfhip_overlay_free (VfHipOverlay *h)
{
  if (!h) return;
  vfhip_overlay_cleanup (h);
  { std::lock_guard<std::mutex> lk (h->mu); ov_drop_image (h); }
  h->st.destroy ();
  delete h;
}

}  // extern "C"
