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

using namespace vfhip;

namespace vfhip {

struct DeintParams {
  metal::Img cur, prev;      // prev.p[0] == nullptr: no history
  metal::OutImg out;
  int method, tff;
  float threshold;
};

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

__global__ __launch_bounds__ (256) void k_deinterlace (const DeintParams p)
{
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

}  // namespace vfhip

struct VfHipDeinterlace {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo info {};
  // host path: two upload slots used alternately, the other one is the previous frame
  int cur_slot = 0;
  bool has_prev = false;
  VfHipFrame prev_dev {};           // device-side previous input frame (host path: staging slot; device path: hist buffer)
  void *hist = nullptr; size_t hist_bytes = 0;
};

static int deint_launch (VfHipDeinterlace *h, const VfHipFrame *cur, const VfHipFrame *prev, VfHipFrame *out,
    const VfHipDeinterlaceParams *prm, hipStream_t s)
{
  DeintParams p {};
  p.cur = metal::make_img (cur);
  if (prev) p.prev = metal::make_img (prev);
  p.out = metal::make_out (out);
  p.method = prm->method; p.tff = prm->top_field_first != 0; p.threshold = prm->motion_threshold;
  const int bw = (h->info.width + 1) / 2, bh = (h->info.height + 1) / 2;
  dim3 grid ((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4));
  hipLaunchKernelGGL (k_deinterlace, grid, dim3 (64, 4), 0, s, p);
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

int vfhip_deinterlace_process (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  int rc = deint_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  VfHipFrame din, dout;
  const int slot = h->cur_slot;                         // slots 0/1 alternate: no device copy for the history
  if ((rc = upload_frame (h->st, slot, in, &din))) return rc;
  if ((rc = alloc_device_frame (h->st, 2, &h->info, &dout))) return rc;
  VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if ((rc = deint_launch (h, &din, h->has_prev ? &h->prev_dev : nullptr, &dout, prm, h->st.s_compute))) return rc;
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  rc = download_frame (h->st, 2, &dout, out);
  h->prev_dev = din; h->has_prev = true; h->cur_slot = 1 - slot;
  return rc;
}

int vfhip_deinterlace_process_device (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out,
    const VfHipDeinterlaceParams *prm, void *stream)
{
  int rc = deint_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  hipStream_t s = stream ? (hipStream_t) stream : h->st.s_compute;
  if ((rc = deint_launch (h, in, h->has_prev ? &h->prev_dev : nullptr, out, prm, s))) return rc;
  // stream-ordered copy of this input into the history buffer (the reference blits _inputRGBA -> _prevFrameRGBA, :394-405)
  size_t total = 0, off[VFHIP_MAX_PLANES] = { 0 };
  const int np = format_n_planes (in->info.format);
  for (int p = 0; p < np; p++) { off[p] = total; total += (frame_plane_bytes (in, p) + 255) / 256 * 256; }
  if (h->hist_bytes < total) {
    if (h->hist) (void) hipFree (h->hist);
    h->hist = nullptr; h->hist_bytes = 0;
    VFHIP_CHECK_HIP (hipMalloc (&h->hist, total));
    h->hist_bytes = total;
  }
  h->prev_dev = *in;
  for (int p = 0; p < np; p++) {
    h->prev_dev.data[p] = (uint8_t *) h->hist + off[p];
    VFHIP_CHECK_HIP (hipMemcpyAsync (h->prev_dev.data[p], in->data[p], frame_plane_bytes (in, p), hipMemcpyDeviceToDevice, s));
  }
  h->has_prev = true;
  return VFHIP_OK;
}

void vfhip_deinterlace_cleanup (VfHipDeinterlace *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  if (h->hist) (void) hipFree (h->hist);
  h->hist = nullptr; h->hist_bytes = 0;
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
