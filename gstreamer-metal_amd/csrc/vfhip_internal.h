// csrc/vfhip_internal.h — shared internals of libvfhip (device singleton, errors, staging).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include "vfhip_host.h"

namespace vfhip {

#define VFHIP_CHECK_HIP(expr)                                                                   \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return vfhip::set_error (VFHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString (_e), __FILE__, __LINE__); \
  } while (0)

// every device allocation of the library.  With $VFHIP_DEBUG_POISON set the new memory is filled with 0xA5: a kernel that reads an intermediate or
// staging byte nothing wrote then fails its parity test on every run instead of only when the allocator hands back dirty memory (fresh
// allocations are zeroed by the driver, recycled ones are not) — the GPU tests and fuzzers are run once with it (tools/gpu_check.sh).
template <typename T> static inline hipError_t dev_malloc (T **p, size_t bytes)
{
  hipError_t e = hipMalloc (reinterpret_cast<void **> (p), bytes);
  static const bool poison = getenv ("VFHIP_DEBUG_POISON") != nullptr;
  if (e == hipSuccess && poison) {
    e = hipMemset (*p, 0xA5, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize ();                  // (hipMemset returns before the fill has run)
    if (e != hipSuccess) { (void) hipFree (*p); *p = nullptr; }        // an error never leaves the caller with memory it does not know about
  }
  return e;
}

// host tables / images that kernels of stream `s` read (tap tables, the filter's LUT, the overlay image): copied IN that stream and waited for, so the
// copy is ordered — and its result made visible — like any other work of the queue the kernels run in (a blocking hipMemcpy is a null-stream
// operation and the handles' streams are non-blocking; the host-side wait orders it as well, this merely keeps every byte the kernels read on
// their own queue).  Introduced while chasing two one-off fuzz mismatches that later evidence puts on the checker's side.
static inline hipError_t upload_in_stream (void *dst, const void *src, size_t bytes, hipStream_t s)
{
  hipError_t e = hipMemcpyAsync (dst, src, bytes, hipMemcpyHostToDevice, s);
  return e != hipSuccess ? e : hipStreamSynchronize (s);
}

// One per GPU ordinal, created once (std::call_once) — the HIP analogue of VfMetalDevice +sharedDevice
// (reference common/vfmetaldevice.m:30-38).
struct Device {
  int ordinal = -1;
  hipDeviceProp_t props;
  int n_cu = 0;
};
int resolve_device (int device);             // <0 → $VFHIP_DEVICE or 0; returns ordinal or negative status
Device *get_device (int device);             // nullptr on failure (error string set)

// Stream trio + pinned staging buffers owned by one renderer handle (reference: one MTLCommandQueue
// per renderer, convertscale/metalconvertscalerenderer.m:112; VfMetalTextureCache slot reuse,
// common/vfmetaltextureutil.m:64-114).
struct Staging {
  Device *dev = nullptr;
  hipStream_t s_h2d = nullptr, s_compute = nullptr, s_d2h = nullptr;
  hipEvent_t ev_h2d = nullptr, ev_compute = nullptr;
  hipEvent_t ev_done[2] = { nullptr, nullptr };    // one per frame in flight (pipelined submit / wait entry points)
  struct Buf { void *host = nullptr; void *devp = nullptr; size_t bytes = 0; };
  std::vector<Buf> slots;                    // slot-indexed like the reference's texture cache
  int init (Device *d);
  int ensure_slot (size_t slot, size_t bytes);   // (re)allocates pinned host + device buffers for a slot
  void destroy ();
};

// plane geometry
int format_n_planes (int format);
int plane_width_bytes (int format, int plane, int width);
int plane_height (int format, int plane, int height);
size_t frame_plane_bytes (const VfHipFrame *f, int plane);   // stride * plane_height
bool format_is_yuv (int format);

// upload host frame planes into a contiguous device image (tight layout with 16-byte aligned strides);
// fills `dev_frame` with device pointers/strides.  Asynchronous on st.s_h2d; records st.ev_h2d.
int upload_frame (Staging &st, size_t slot, const VfHipFrame *host, VfHipFrame *dev_frame);
// allocate a device image for an output frame of the given info (aligned strides)
int alloc_device_frame (Staging &st, size_t slot, const VfHipVideoInfo *info, VfHipFrame *dev_frame);
// the device image a kernel should write: the caller's own frame when it is device-resident (VFHIP_FRAME_FLAG_DEVICE),
// else a staging image like alloc_device_frame
int output_frame (Staging &st, size_t slot, const VfHipVideoInfo *info, const VfHipFrame *out, VfHipFrame *dev_frame);
// download a device image into host frame planes (honours the host strides); synchronises
int download_frame (Staging &st, size_t slot, const VfHipFrame *dev_frame, VfHipFrame *host);
// the same in two halves for the pipelined entry points: _begin enqueues the copies behind st.ev_compute and records
// `done`; _finish waits for `done` and moves staged planes (pageable destinations) into the caller's frame
int download_begin (Staging &st, size_t slot, VfHipFrame *host, bool staged[VFHIP_MAX_PLANES], hipEvent_t done);
int download_finish (Staging &st, size_t slot, VfHipFrame *host, const bool staged[VFHIP_MAX_PLANES], hipEvent_t done);

// ---- pipelined host path shared by the renderers' _submit / _wait entry points -----------------------------------------
// Up to two frames in flight per handle; flight k uses staging slots 2k (input image) and 2k + 1 (output image).  submit
// enqueues upload -> kernel(s) -> download on the handle's three streams and returns; wait blocks until the OLDEST frame's
// output is complete (and moves staged planes of a pageable destination into the caller's frame).
struct Flights {
  struct F { VfHipFrame out; bool staged[VFHIP_MAX_PLANES]; } f[2];
  int head = 0, count = 0;
};

template <class Launch>                     // Launch: int (const VfHipFrame *dev_in, VfHipFrame *dev_out, hipStream_t)
static inline int flights_submit (Staging &st, Flights &fl, const VfHipVideoInfo *out_info, const VfHipFrame *in, VfHipFrame *out, Launch launch)
{
  if (fl.count >= 2) return set_error (VFHIP_ERR_INVALID, "two frames are already in flight: call the handle's _wait first");
  const int k = (fl.head + fl.count) & 1;
  VfHipFrame din, dout;
  int rc;
  // pageable planes are copied into the pinned staging slot here (CPU work that overlaps the previous frame's GPU work);
  // pinned planes are DMA'd in place and must stay valid until the frame's wait returns
  if ((rc = upload_frame (st, (size_t) 2 * k, in, &din))) return rc;
  if ((rc = output_frame (st, (size_t) 2 * k + 1, out_info, out, &dout))) return rc;
  VFHIP_CHECK_HIP (hipStreamWaitEvent (st.s_compute, st.ev_h2d, 0));
  if ((rc = launch (&din, &dout, st.s_compute))) return rc;
  VFHIP_CHECK_HIP (hipEventRecord (st.ev_compute, st.s_compute));
  fl.f[k].out = *out;
  if ((rc = download_begin (st, (size_t) 2 * k + 1, &fl.f[k].out, fl.f[k].staged, st.ev_done[k]))) return rc;
  fl.count++;
  return VFHIP_OK;
}

static inline int flights_wait (Staging &st, Flights &fl)
{
  if (fl.count == 0) return set_error (VFHIP_ERR_INVALID, "no frame in flight");
  const int k = fl.head;
  fl.head ^= 1; fl.count--;
  return download_finish (st, (size_t) 2 * k + 1, &fl.f[k].out, fl.f[k].staged, st.ev_done[k]);
}

// abandoned frames (cleanup with a non-empty pipeline): let the GPU finish with the staging buffers before they go
static inline void flights_abandon (Staging &st, Flights &fl)
{
  if (!fl.count) return;
  (void) hipStreamSynchronize (st.s_h2d); (void) hipStreamSynchronize (st.s_compute); (void) hipStreamSynchronize (st.s_d2h);
  fl.count = 0; fl.head = 0;
}

}  // namespace vfhip

namespace vfhip {
// shared argument checks of the element entry points
int check_frame (const VfHipFrame *f, const VfHipVideoInfo *want, const char *what);
}  // namespace vfhip
