/* include/vfhip.h — C ABI of libvfhip: the MI355X (gfx950) replacement for everything below the
 * `void *renderer` pointer of the reference's four hot-path elements.
 *
 * Plain C, plain pointers and sizes, no GStreamer / torch / HIP types.  The GStreamer element
 * shells (gstreamer-metal_amd/gst/) translate GstVideoInfo/GstVideoFrame into the PODs below,
 * exactly where the reference passed them to its Objective-C renderer objects:
 *
 *   reference interface (file:line)                                   replaced by
 *   ---------------------------------------------------------------   ---------------------------
 *   VfMetalDevice +sharedDevice         common/vfmetaldevice.m:30-38   vfhip_device_*
 *   VfMetalTextureCache -uploadPlane    common/vfmetaltextureutil.m:64-114   pinned staging pool inside *_process
 *   MetalConvertScaleRenderer           convertscale/metalconvertscalerenderer.h:35-50   vfhip_convertscale_*
 *   MetalVideoFilterRenderer            videofilter/metalvideofilterrenderer.h:30-70     vfhip_videofilter_*
 *   MetalCompositorRenderer             compositor/metalcomprenderer.h:29-63             vfhip_compositor_*
 *   MetalDeinterlaceRenderer            deinterlace/metaldeinterlacerenderer.h:29-54     vfhip_deinterlace_*
 *
 * Conventions (mirroring the reference's BOOL/nil conventions):
 *   - every function returns VFHIP_OK (0) or a negative VfHipStatus; vfhip_last_error_string()
 *     gives the thread-local message; nothing aborts;
 *   - *_new() returns NULL on failure (reference: -init returning nil);
 *   - frames are borrowed for the duration of the call; `*_process` is synchronous (output fully
 *     written on return); `*_process_device*` is asynchronous on the given HIP stream and takes
 *     device pointers;
 *   - a handle is single-caller (internally mutex-guarded); many handles per GPU/process coexist;
 *   - there is NO CPU fallback: without a HIP device every entry point fails with
 *     VFHIP_ERR_NO_DEVICE.
 */
#ifndef VFHIP_H
#define VFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFHIP_ABI_VERSION 1
#define VFHIP_MAX_PLANES 4

typedef enum {
  VFHIP_OK = 0,
  VFHIP_ERR_INVALID = -1,          /* bad argument */
  VFHIP_ERR_UNSUPPORTED = -2,      /* format / mode combination not implemented */
  VFHIP_ERR_NOT_CONFIGURED = -3,   /* process before configure (reference: GST_FLOW_NOT_NEGOTIATED) */
  VFHIP_ERR_HIP = -4,              /* a HIP runtime call failed (message has hipGetErrorString) */
  VFHIP_ERR_NOMEM = -5,
  VFHIP_ERR_NO_DEVICE = -6,        /* no usable gfx950 device / HIP runtime */
  VFHIP_ERR_IO = -7                /* LUT file could not be read / parsed */
} VfHipStatus;

/* Same six formats as the reference's pad templates (convertscale/gstvfmetalconvertscale.m:48-60). */
typedef enum {
  VFHIP_FORMAT_BGRA = 0,
  VFHIP_FORMAT_RGBA = 1,
  VFHIP_FORMAT_NV12 = 2,
  VFHIP_FORMAT_I420 = 3,
  VFHIP_FORMAT_UYVY = 4,
  VFHIP_FORMAT_YUY2 = 5
} VfHipFormat;

/* The reference only distinguishes 601 / 709 (common/vfmetaltextureutil.m:35-41); GStreamer's CPU
 * path also has BT.2020 (default at >= 2160 lines), which the gst-exact numerics need. */
typedef enum {
  VFHIP_MATRIX_BT601 = 0,
  VFHIP_MATRIX_BT709 = 1,
  VFHIP_MATRIX_BT2020 = 2
} VfHipColorMatrix;

typedef enum {
  VFHIP_CHROMA_SITE_CENTER = 0,      /* GST_VIDEO_CHROMA_SITE_NONE / JPEG: not co-sited */
  VFHIP_CHROMA_SITE_H_COSITED = 1    /* GST_VIDEO_CHROMA_SITE_MPEG2 */
} VfHipChromaSite;

/* Two sets of arithmetic exist for this path (SURVEY.md finding 3). */
typedef enum {
  VFHIP_NUMERICS_GST_EXACT = 0,  /* integer arithmetic of GStreamer 1.14 videoconvert+videoscale, bit-exact (default) */
  VFHIP_NUMERICS_METAL = 1       /* float arithmetic of the reference's Metal shaders */
} VfHipNumerics;

typedef struct {
  int32_t format;        /* VfHipFormat */
  int32_t width;
  int32_t height;
  int32_t color_matrix;  /* VfHipColorMatrix (YUV formats) */
  int32_t chroma_site;   /* VfHipChromaSite (4:2:0 formats) */
  int32_t reserved[3];
} VfHipVideoInfo;

#define VFHIP_FRAME_FLAG_TFF 0x1u   /* GST_VIDEO_BUFFER_FLAG_TFF (deinterlace/gstvfmetaldeinterlace.m:176-184) */

typedef struct {
  VfHipVideoInfo info;
  void *data[VFHIP_MAX_PLANES];      /* plane base pointers (host for *_process, device for *_process_device) */
  int32_t stride[VFHIP_MAX_PLANES];  /* bytes per row of each plane */
  uint32_t flags;
  uint32_t reserved;
} VfHipFrame;

/* ---- device layer (reference: VfMetalDevice, common/vfmetaldevice.m) --------------------------- */
int vfhip_abi_version (void);
const char *vfhip_last_error_string (void);
int vfhip_device_count (void);                       /* < 0: VfHipStatus */
int vfhip_device_init (int device);                  /* thread-safe singleton per ordinal; device < 0: $VFHIP_DEVICE or 0 */
int vfhip_device_name (int device, char *buf, size_t buflen);
int vfhip_device_synchronize (int device);
/* pinned host memory for GstAllocator / buffer pools, and registration of foreign buffers */
void *vfhip_pinned_alloc (int device, size_t bytes);
void vfhip_pinned_free (void *p);
int vfhip_host_register (void *p, size_t bytes);
int vfhip_host_unregister (void *p);
/* raw device memory + copies (tests, bench, device-resident frame rings) */
void *vfhip_device_malloc (int device, size_t bytes);
void vfhip_device_free (int device, void *p);
int vfhip_memcpy_h2d (int device, void *dst, const void *src, size_t bytes);
int vfhip_memcpy_d2h (int device, void *dst, const void *src, size_t bytes);
/* number of planes / plane geometry helpers shared by shells, tests and bench */
int vfhip_format_n_planes (int format);
int vfhip_plane_width_bytes (int format, int plane, int width);
int vfhip_plane_height (int format, int plane, int height);

/* ---- convertscale (reference: MetalConvertScaleRenderer) ------------------------------------------ */
typedef enum {
  VFHIP_SCALE_BILINEAR = 0,      /* VF_METAL_SCALE_BILINEAR, convertscale/metalconvertscalerenderer.h:30-33 */
  VFHIP_SCALE_NEAREST = 1
} VfHipScaleMethod;

typedef struct VfHipConvertScale VfHipConvertScale;

VfHipConvertScale *vfhip_convertscale_new (int device);
/* -configureWithInputInfo:outputInfo:method:addBorders:borderColor: (metalconvertscalerenderer.m:226-330);
 * border_color is ARGB like the element property. */
int vfhip_convertscale_configure (VfHipConvertScale *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out,
    int method, int add_borders, uint32_t border_color, int numerics);
/* -processFrame:output: (metalconvertscalerenderer.m:332-512): host frames, synchronous */
int vfhip_convertscale_process (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out);
/* device-resident frames, asynchronous on `stream` (a hipStream_t, NULL = the handle's own compute stream) */
int vfhip_convertscale_process_device (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out, void *stream);
/* n_frames frames laid out at a constant pitch: plane p of frame k lives at data[p] + k * pitch */
int vfhip_convertscale_process_device_batch (VfHipConvertScale *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, void *stream);
/* name of the kernel variant the current configuration dispatches to (for profiles / tests) */
const char *vfhip_convertscale_kernel_name (VfHipConvertScale *h);
void vfhip_convertscale_cleanup (VfHipConvertScale *h);   /* -cleanup: drop GPU resources, keep the handle */
void vfhip_convertscale_free (VfHipConvertScale *h);

#ifdef __cplusplus
}
#endif
#endif /* VFHIP_H */
