// csrc/vfhip_internal.h — shared internals of libvfhip (device singleton, errors, staging).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "../../include/vfhip.h"

namespace vfhip {

int set_error (int code, const char *fmt, ...) __attribute__ ((format (printf, 2, 3)));

#define VFHIP_CHECK_HIP(expr)                                                                   \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return vfhip::set_error (VFHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString (_e), __FILE__, __LINE__); \
  } while (0)

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

}  // namespace vfhip

namespace vfhip {
// PNG -> straight RGBA8 (image_png.hip; host code, zlib)
int decode_png (const char *path, std::vector<uint8_t> &rgba, int *width, int *height);
// shared argument checks of the element entry points
int check_frame (const VfHipFrame *f, const VfHipVideoInfo *want, const char *what);
}  // namespace vfhip
