// csrc/vfhip_core.hip — device singleton, error strings, pinned staging, raw memory helpers.
// Replaces reference common/vfmetaldevice.{h,m} and common/vfmetaltextureutil.{h,m} (SURVEY.md §2 #2,#3).
#include "vfhip_internal.h"
#include <cstdlib>
#include <map>

namespace vfhip {

static thread_local char g_err[512] = "";

int set_error (int code, const char *fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (g_err, sizeof (g_err), fmt, ap);
  va_end (ap);
  return code;
}

static std::once_flag g_count_once;
static int g_count = 0;
static hipError_t g_count_err = hipSuccess;
static const int kMaxDevices = 64;
static Device g_devices[kMaxDevices];
static std::once_flag g_dev_once[kMaxDevices];
static int g_dev_status[kMaxDevices];

static int device_count ()
{
  std::call_once (g_count_once, [] {
    int n = 0;
    g_count_err = hipGetDeviceCount (&n);
    g_count = g_count_err == hipSuccess ? n : 0;
  });
  if (g_count_err != hipSuccess)
    return set_error (VFHIP_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString (g_count_err));
  if (g_count <= 0)
    return set_error (VFHIP_ERR_NO_DEVICE, "no HIP device visible (libvfhip has no CPU fallback)");
  return g_count;
}

int resolve_device (int device)
{
  int n = device_count ();
  if (n < 0) return n;
  if (device < 0) {
    const char *e = getenv ("VFHIP_DEVICE");
    device = e ? atoi (e) : 0;
  }
  if (device < 0 || device >= n || device >= kMaxDevices)
    return set_error (VFHIP_ERR_INVALID, "device ordinal %d out of range (0..%d)", device, n - 1);
  return device;
}

Device *get_device (int device)
{
  int d = resolve_device (device);
  if (d < 0) return nullptr;
  std::call_once (g_dev_once[d], [d] {
    Device &D = g_devices[d];
    hipError_t e = hipGetDeviceProperties (&D.props, d);
    if (e != hipSuccess) { g_dev_status[d] = set_error (VFHIP_ERR_HIP, "hipGetDeviceProperties(%d): %s", d, hipGetErrorString (e)); return; }
    D.ordinal = d;
    D.n_cu = D.props.multiProcessorCount;
    g_dev_status[d] = VFHIP_OK;
  });
  if (g_dev_status[d] != VFHIP_OK) return nullptr;
  return &g_devices[d];
}

// ---- staging ---------------------------------------------------------------------------------------
int Staging::init (Device *d)
{
  dev = d;
  VFHIP_CHECK_HIP (hipSetDevice (d->ordinal));
  VFHIP_CHECK_HIP (hipStreamCreateWithFlags (&s_h2d, hipStreamNonBlocking));
  VFHIP_CHECK_HIP (hipStreamCreateWithFlags (&s_compute, hipStreamNonBlocking));
  VFHIP_CHECK_HIP (hipStreamCreateWithFlags (&s_d2h, hipStreamNonBlocking));
  VFHIP_CHECK_HIP (hipEventCreateWithFlags (&ev_h2d, hipEventDisableTiming));
  VFHIP_CHECK_HIP (hipEventCreateWithFlags (&ev_compute, hipEventDisableTiming));
  for (auto &e : ev_done) VFHIP_CHECK_HIP (hipEventCreateWithFlags (&e, hipEventDisableTiming));
  return VFHIP_OK;
}

int Staging::ensure_slot (size_t slot, size_t bytes)
{
  if (slots.size () <= slot) slots.resize (slot + 1);
  Buf &b = slots[slot];
  if (b.bytes >= bytes) return VFHIP_OK;
  VFHIP_CHECK_HIP (hipSetDevice (dev->ordinal));
  if (b.host) (void) hipHostFree (b.host);
  if (b.devp) (void) hipFree (b.devp);
  b = Buf ();
  VFHIP_CHECK_HIP (hipHostMalloc (&b.host, bytes, hipHostMallocDefault));
  VFHIP_CHECK_HIP (dev_malloc (&b.devp, bytes));
  b.bytes = bytes;
  return VFHIP_OK;
}

void Staging::destroy ()
{
  if (!dev) return;
  (void) hipSetDevice (dev->ordinal);
  for (auto &b : slots) {
    if (b.host) (void) hipHostFree (b.host);
    if (b.devp) (void) hipFree (b.devp);
  }
  slots.clear ();
  if (ev_h2d) (void) hipEventDestroy (ev_h2d);
  if (ev_compute) (void) hipEventDestroy (ev_compute);
  for (auto &e : ev_done) { if (e) (void) hipEventDestroy (e); e = nullptr; }
  if (s_h2d) (void) hipStreamDestroy (s_h2d);
  if (s_compute) (void) hipStreamDestroy (s_compute);
  if (s_d2h) (void) hipStreamDestroy (s_d2h);
  ev_h2d = ev_compute = nullptr;
  s_h2d = s_compute = s_d2h = nullptr;
}

// ---- plane geometry ----------------------------------------------------------------------------------
bool format_is_yuv (int f) { return f >= VFHIP_FORMAT_NV12; }

int format_n_planes (int f)
{
  switch (f) {
    case VFHIP_FORMAT_BGRA: case VFHIP_FORMAT_RGBA: case VFHIP_FORMAT_UYVY: case VFHIP_FORMAT_YUY2: return 1;
    case VFHIP_FORMAT_NV12: return 2;
    case VFHIP_FORMAT_I420: return 3;
    default: return set_error (VFHIP_ERR_INVALID, "unknown format %d", f);
  }
}

int plane_width_bytes (int f, int plane, int w)
{
  switch (f) {
    case VFHIP_FORMAT_BGRA: case VFHIP_FORMAT_RGBA: return 4 * w;
    case VFHIP_FORMAT_UYVY: case VFHIP_FORMAT_YUY2: return 4 * ((w + 1) / 2);
    case VFHIP_FORMAT_NV12: return plane == 0 ? w : 2 * ((w + 1) / 2);
    case VFHIP_FORMAT_I420: return plane == 0 ? w : (w + 1) / 2;
    default: return -1;
  }
}

int plane_height (int f, int plane, int h)
{
  if ((f == VFHIP_FORMAT_NV12 || f == VFHIP_FORMAT_I420) && plane > 0) return (h + 1) / 2;
  return h;
}

static inline size_t align_up (size_t v, size_t a) { return (v + a - 1) / a * a; }

static int device_layout (const VfHipVideoInfo *info, size_t off[VFHIP_MAX_PLANES], int stride[VFHIP_MAX_PLANES], size_t *total)
{
  int np = format_n_planes (info->format);
  if (np < 0) return np;
  size_t o = 0;
  for (int p = 0; p < np; p++) {
    stride[p] = (int) align_up ((size_t) plane_width_bytes (info->format, p, info->width), 16);
    off[p] = o;
    o += align_up ((size_t) stride[p] * plane_height (info->format, p, info->height), 256);
  }
  *total = o + 256;   // tail slack: vector loads of the last row may touch up to 16 bytes past the row end
  return np;
}

int alloc_device_frame (Staging &st, size_t slot, const VfHipVideoInfo *info, VfHipFrame *df)
{
  size_t off[VFHIP_MAX_PLANES], total; int stride[VFHIP_MAX_PLANES];
  int np = device_layout (info, off, stride, &total);
  if (np < 0) return np;
  int rc = st.ensure_slot (slot, total);
  if (rc) return rc;
  memset (df, 0, sizeof (*df));
  df->info = *info;
  for (int p = 0; p < np; p++) { df->data[p] = (uint8_t *) st.slots[slot].devp + off[p]; df->stride[p] = stride[p]; }
  return VFHIP_OK;
}

// true when `p` is pinned host memory the HIP runtime knows (hipHostMalloc / hipHostRegister, e.g. a buffer from the
// plugin's pinned GstAllocator): such planes are DMA'd in place, without the staging memcpy
static bool is_pinned_host (const void *p)
{
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes (&a, p) != hipSuccess) { (void) hipGetLastError (); return false; }
  return a.type == hipMemoryTypeHost;
}

// device-resident frames (VFHIP_FRAME_FLAG_DEVICE) are used in place; alignment is what the kernels' 2- and 4-byte
// accesses need
static int check_device_frame (const VfHipFrame *f, const char *what)
{
  const int np = format_n_planes (f->info.format);
  if (np < 0) return np;
  const uintptr_t need = (f->info.format == VFHIP_FORMAT_BGRA || f->info.format == VFHIP_FORMAT_RGBA) ? 3u : 1u;
  for (int p = 0; p < np; p++) {
    if (!f->data[p]) return set_error (VFHIP_ERR_INVALID, "%s plane %d is NULL", what, p);
    if (f->stride[p] < plane_width_bytes (f->info.format, p, f->info.width)) return set_error (VFHIP_ERR_INVALID, "%s plane %d: short stride", what, p);
    if (((uintptr_t) f->data[p] | (uintptr_t) f->stride[p]) & need)
      return set_error (VFHIP_ERR_INVALID, "%s device plane %d is not %d-byte aligned", what, p, (int) need + 1);
  }
  return VFHIP_OK;
}

int output_frame (Staging &st, size_t slot, const VfHipVideoInfo *info, const VfHipFrame *out, VfHipFrame *df)
{
  if (out && (out->flags & VFHIP_FRAME_FLAG_DEVICE)) {
    int rc = check_device_frame (out, "output");
    if (rc) return rc;
    *df = *out;
    return VFHIP_OK;
  }
  return alloc_device_frame (st, slot, info, df);
}

int upload_frame (Staging &st, size_t slot, const VfHipFrame *host, VfHipFrame *df)
{
  if (host->flags & VFHIP_FRAME_FLAG_DEVICE) {            // already on the GPU: nothing to copy
    int rc = check_device_frame (host, "input");
    if (rc) return rc;
    *df = *host;
    VFHIP_CHECK_HIP (hipEventRecord (st.ev_h2d, st.s_h2d));
    return VFHIP_OK;
  }
  size_t off[VFHIP_MAX_PLANES], total; int stride[VFHIP_MAX_PLANES];
  int np = device_layout (&host->info, off, stride, &total);
  if (np < 0) return np;
  int rc = st.ensure_slot (slot, total);
  if (rc) return rc;
  Staging::Buf &b = st.slots[slot];
  memset (df, 0, sizeof (*df));
  df->info = host->info; df->flags = host->flags;
  for (int p = 0; p < np; p++) {
    if (!host->data[p]) return set_error (VFHIP_ERR_INVALID, "input plane %d is NULL", p);
    int wb = plane_width_bytes (host->info.format, p, host->info.width);
    int ph = plane_height (host->info.format, p, host->info.height);
    if (host->stride[p] < wb) return set_error (VFHIP_ERR_INVALID, "plane %d stride %d < row bytes %d", p, host->stride[p], wb);
    df->data[p] = (uint8_t *) b.devp + off[p]; df->stride[p] = stride[p];
    const uint8_t *src = (const uint8_t *) host->data[p];
    if (is_pinned_host (src)) {          // zero-copy host side: async 2-D DMA straight from the caller's pinned plane
      VFHIP_CHECK_HIP (hipMemcpy2DAsync (df->data[p], (size_t) stride[p], src, (size_t) host->stride[p], (size_t) wb, (size_t) ph,
                                         hipMemcpyHostToDevice, st.s_h2d));
      continue;
    }
    uint8_t *dst = (uint8_t *) b.host + off[p];
    if (host->stride[p] == stride[p]) memcpy (dst, src, (size_t) stride[p] * (ph - 1) + wb);
    else for (int y = 0; y < ph; y++) memcpy (dst + (size_t) y * stride[p], src + (size_t) y * host->stride[p], wb);
    VFHIP_CHECK_HIP (hipMemcpyAsync (df->data[p], dst, (size_t) stride[p] * (ph - 1) + wb, hipMemcpyHostToDevice, st.s_h2d));
  }
  VFHIP_CHECK_HIP (hipEventRecord (st.ev_h2d, st.s_h2d));
  return VFHIP_OK;
}

int download_begin (Staging &st, size_t slot, VfHipFrame *host, bool staged[VFHIP_MAX_PLANES], hipEvent_t done)
{
  for (int p = 0; p < VFHIP_MAX_PLANES; p++) staged[p] = false;
  if (host->flags & VFHIP_FRAME_FLAG_DEVICE) {            // the kernel wrote the caller's device frame in place (output_frame)
    VFHIP_CHECK_HIP (hipEventRecord (done, st.s_compute));
    return VFHIP_OK;
  }
  size_t off[VFHIP_MAX_PLANES], total; int stride[VFHIP_MAX_PLANES];
  int np = device_layout (&host->info, off, stride, &total);
  if (np < 0) return np;
  Staging::Buf &b = st.slots[slot];
  VFHIP_CHECK_HIP (hipStreamWaitEvent (st.s_d2h, st.ev_compute, 0));
  for (int p = 0; p < np; p++) {
    if (!host->data[p]) return set_error (VFHIP_ERR_INVALID, "output plane %d is NULL", p);
    int wb = plane_width_bytes (host->info.format, p, host->info.width);
    int ph = plane_height (host->info.format, p, host->info.height);
    if (host->stride[p] < wb) return set_error (VFHIP_ERR_INVALID, "output plane %d stride %d < row bytes %d", p, host->stride[p], wb);
    const uint8_t *dsrc = (const uint8_t *) b.devp + off[p];
    if (is_pinned_host (host->data[p])) {
      VFHIP_CHECK_HIP (hipMemcpy2DAsync (host->data[p], (size_t) host->stride[p], dsrc, (size_t) stride[p], (size_t) wb, (size_t) ph,
                                         hipMemcpyDeviceToHost, st.s_d2h));
    } else {
      staged[p] = true;
      VFHIP_CHECK_HIP (hipMemcpyAsync ((uint8_t *) b.host + off[p], dsrc, (size_t) stride[p] * (ph - 1) + wb, hipMemcpyDeviceToHost, st.s_d2h));
    }
  }
  VFHIP_CHECK_HIP (hipEventRecord (done, st.s_d2h));
  return VFHIP_OK;
}

int download_finish (Staging &st, size_t slot, VfHipFrame *host, const bool staged[VFHIP_MAX_PLANES], hipEvent_t done)
{
  VFHIP_CHECK_HIP (hipEventSynchronize (done));
  if (host->flags & VFHIP_FRAME_FLAG_DEVICE) return VFHIP_OK;
  size_t off[VFHIP_MAX_PLANES], total; int stride[VFHIP_MAX_PLANES];
  int np = device_layout (&host->info, off, stride, &total);
  if (np < 0) return np;
  Staging::Buf &b = st.slots[slot];
  for (int p = 0; p < np; p++) {
    if (!staged[p]) continue;
    int wb = plane_width_bytes (host->info.format, p, host->info.width);
    int ph = plane_height (host->info.format, p, host->info.height);
    const uint8_t *src = (const uint8_t *) b.host + off[p];
    uint8_t *dst = (uint8_t *) host->data[p];
    for (int y = 0; y < ph; y++) memcpy (dst + (size_t) y * host->stride[p], src + (size_t) y * stride[p], wb);
  }
  return VFHIP_OK;
}

int download_frame (Staging &st, size_t slot, const VfHipFrame *df, VfHipFrame *host)
{
  (void) df;
  bool staged[VFHIP_MAX_PLANES];
  int rc = download_begin (st, slot, host, staged, st.ev_done[0]);
  if (rc) return rc;
  return download_finish (st, slot, host, staged, st.ev_done[0]);
}

size_t frame_plane_bytes (const VfHipFrame *f, int plane)
{
  return (size_t) f->stride[plane] * plane_height (f->info.format, plane, f->info.height);
}

}  // namespace vfhip

using namespace vfhip;

extern "C" {

int vfhip_abi_version (void) { return VFHIP_ABI_VERSION; }
const char *vfhip_last_error_string (void) { return g_err; }
int vfhip_device_count (void) { return device_count (); }
int vfhip_device_init (int device) { Device *d = get_device (device); return d ? d->ordinal : (resolve_device (device) < 0 ? resolve_device (device) : VFHIP_ERR_HIP); }

int vfhip_device_name (int device, char *buf, size_t buflen)
{
  Device *d = get_device (device);
  if (!d) return VFHIP_ERR_NO_DEVICE;
  if (!buf || !buflen) return set_error (VFHIP_ERR_INVALID, "null buffer");
  snprintf (buf, buflen, "%s (%s, %d CUs)", d->props.name, d->props.gcnArchName, d->n_cu);
  return VFHIP_OK;
}

int vfhip_device_synchronize (int device)
{
  Device *d = get_device (device);
  if (!d) return VFHIP_ERR_NO_DEVICE;
  VFHIP_CHECK_HIP (hipSetDevice (d->ordinal));
  VFHIP_CHECK_HIP (hipDeviceSynchronize ());
  return VFHIP_OK;
}

void *vfhip_pinned_alloc (int device, size_t bytes)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  void *p = nullptr;
  if (hipSetDevice (d->ordinal) != hipSuccess || hipHostMalloc (&p, bytes, hipHostMallocDefault) != hipSuccess) {
    set_error (VFHIP_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
}
void vfhip_pinned_free (void *p) { if (p) (void) hipHostFree (p); }

int vfhip_host_register (void *p, size_t bytes)
{
  if (device_count () < 0) return VFHIP_ERR_NO_DEVICE;
  VFHIP_CHECK_HIP (hipHostRegister (p, bytes, hipHostRegisterDefault));
  return VFHIP_OK;
}
int vfhip_host_unregister (void *p)
{
  if (device_count () < 0) return VFHIP_ERR_NO_DEVICE;
  VFHIP_CHECK_HIP (hipHostUnregister (p));
  return VFHIP_OK;
}

void *vfhip_device_malloc (int device, size_t bytes)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  void *p = nullptr;
  if (hipSetDevice (d->ordinal) != hipSuccess || dev_malloc (&p, bytes) != hipSuccess) {
    set_error (VFHIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
}
void vfhip_device_free (int device, void *p)
{
  Device *d = get_device (device);
  if (d && p) { (void) hipSetDevice (d->ordinal); (void) hipFree (p); }
}
int vfhip_memcpy_h2d (int device, void *dst, const void *src, size_t bytes)
{
  Device *d = get_device (device);
  if (!d) return VFHIP_ERR_NO_DEVICE;
  VFHIP_CHECK_HIP (hipSetDevice (d->ordinal));
  VFHIP_CHECK_HIP (hipMemcpy (dst, src, bytes, hipMemcpyHostToDevice));
  return VFHIP_OK;
}
int vfhip_memcpy_d2h (int device, void *dst, const void *src, size_t bytes)
{
  Device *d = get_device (device);
  if (!d) return VFHIP_ERR_NO_DEVICE;
  VFHIP_CHECK_HIP (hipSetDevice (d->ordinal));
  VFHIP_CHECK_HIP (hipMemcpy (dst, src, bytes, hipMemcpyDeviceToHost));
  return VFHIP_OK;
}

int vfhip_format_n_planes (int format) { return format_n_planes (format); }
int vfhip_plane_width_bytes (int format, int plane, int width) { return plane_width_bytes (format, plane, width); }
int vfhip_plane_height (int format, int plane, int height) { return plane_height (format, plane, height); }

}  // extern "C"

namespace vfhip {
int check_frame (const VfHipFrame *f, const VfHipVideoInfo *want, const char *what)
{
  if (!f) return set_error (VFHIP_ERR_INVALID, "%s frame is NULL", what);
  if (want && (f->info.format != want->format || f->info.width != want->width || f->info.height != want->height))
    return set_error (VFHIP_ERR_INVALID, "%s frame does not match the configured caps", what);
  const int np = format_n_planes (f->info.format);
  if (np < 0) return np;
  if (f->info.width <= 0 || f->info.height <= 0) return set_error (VFHIP_ERR_INVALID, "%s frame has no size", what);
  for (int p = 0; p < np; p++)
    if (!f->data[p] || f->stride[p] < plane_width_bytes (f->info.format, p, f->info.width))
      return set_error (VFHIP_ERR_INVALID, "%s plane %d: null pointer or short stride", what, p);
  return VFHIP_OK;
}
}  // namespace vfhip
