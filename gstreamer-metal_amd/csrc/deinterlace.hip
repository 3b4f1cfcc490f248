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
  if (h->info.format == VFHIP_FORMAT_NV12 || h->info.format == VFHIP_FORMAT_I420) {
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
    for (int k = 0; k < 2; k++) VFHIP_CHECK_HIP (hipMalloc (&h->hist[k], total));
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
