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
  VFHIP_NUMERICS_GST_EXACT = 0,  /* integer arithmetic of GStreamer 1.14 videoconvert+videoscale, bit-exact (default).  The few
                                  * cells of the format matrix whose GStreamer arithmetic is not pinned still
                                  * configure, run the `metal` arithmetic, and say so: vfhip_convertscale_numerics_in_effect ()
                                  * returns VFHIP_NUMERICS_METAL and the element posts a warning */
  VFHIP_NUMERICS_METAL = 1,      /* float arithmetic of the reference's Metal shaders */
  VFHIP_NUMERICS_GST_EXACT_STRICT = 2   /* gst-exact or nothing: configure refuses an unpinned cell with VFHIP_ERR_UNSUPPORTED */
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
/* data[] are DEVICE pointers on the handle's GPU (a `memory:HIPMemory` GstBuffer of an upstream / downstream vfhip
 * element): the synchronous *_process / _composite entry points then skip the PCIe copy on that
 * side — no upload for such an input, the kernel writes such an output in place — and still return only when the
 * output is complete.  RGBA / BGRA planes must be 4-byte aligned, 4:2:0 and packed-YUV planes 2-byte aligned. */
#define VFHIP_FRAME_FLAG_DEVICE 0x2u

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
/* PNG decoder used by the overlay image and PNG LUT loaders (host only; every colour type and bit depth, Adam7, tRNS; 16-bit samples keep their high byte):
 * straight RGBA8, row 0 first, malloc'ed — release with vfhip_image_free */
int vfhip_image_decode_png (const char *path, uint8_t **rgba, int *width, int *height);
/* the overlay's image loader: PNG as above, or sequential / progressive Huffman JPEG (greyscale, YCbCr 4:4:4 / 4:2:2 / 4:2:0, restart intervals;
 * libjpeg's integer arithmetic; alpha 255), chosen by the file's first bytes.  The reference loads both through ImageIO
 * (overlay/metaloverlayrenderer.m:166-245). */
int vfhip_image_decode (const char *path, uint8_t **rgba, int *width, int *height);
void vfhip_image_free (uint8_t *rgba);

int vfhip_format_n_planes (int format);
int vfhip_plane_width_bytes (int format, int plane, int width);
int vfhip_plane_height (int format, int plane, int height);

/* ---- convertscale (reference: MetalConvertScaleRenderer) ------------------------------------------ */
typedef enum {
  VFHIP_SCALE_BILINEAR = 0,      /* VF_METAL_SCALE_BILINEAR, convertscale/metalconvertscalerenderer.h:30-33 */
  VFHIP_SCALE_NEAREST = 1,
  /* additive (the reference has no bicubic; north_star names it): GStreamer's `videoscale method=catrom`, bit-exact, for
   * numerics = gst-exact on every cell of the 6 x 6 format matrix whose 2-tap arithmetic is pinned (RGB, 4:2:0 and packed
   * 4:2:2 outputs; borders on chroma-sample boundaries), for lines at least as long as the filter (ceil(4 * max(1, in/out))
   * taps, at most 64); anything else -> VFHIP_ERR_UNSUPPORTED at configure */
  VFHIP_SCALE_BICUBIC = 2
} VfHipScaleMethod;

typedef struct VfHipConvertScale VfHipConvertScale;

VfHipConvertScale *vfhip_convertscale_new (int device);
/* -configureWithInputInfo:outputInfo:method:addBorders:borderColor: (metalconvertscalerenderer.m:226-330);
 * border_color is ARGB like the element property. */
int vfhip_convertscale_configure (VfHipConvertScale *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out,
    int method, int add_borders, uint32_t border_color, int numerics);
/* -processFrame:output: (metalconvertscalerenderer.m:332-512): host frames, synchronous */
int vfhip_convertscale_process (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out);
/* the arithmetic family the configured cell really runs: VFHIP_NUMERICS_GST_EXACT or VFHIP_NUMERICS_METAL (differs from the
 * requested one exactly when gst-exact was asked for on an unpinned cell); < 0: not configured */
int vfhip_convertscale_numerics_in_effect (VfHipConvertScale *h);
/* device-resident frames, asynchronous on `stream` (a hipStream_t, NULL = the handle's own compute stream) */
/* Pipelined variant of _process for host frames: _submit enqueues upload -> kernel -> download of one frame on the handle's
 * three streams and returns at once (pageable planes are copied into pinned staging before it returns; pinned planes and
 * the whole output frame stay borrowed until the frame's _wait returns); at most two frames may be in flight; _wait blocks
 * until the OLDEST submitted frame's output is complete.  With two frames in flight the upload of frame n+1 overlaps the
 * kernel and the download of frame n (PCIe is full duplex).  cleanup / configure require an empty pipeline. */
int vfhip_convertscale_submit (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out);
int vfhip_convertscale_wait (VfHipConvertScale *h);
int vfhip_convertscale_in_flight (VfHipConvertScale *h);
int vfhip_convertscale_process_device (VfHipConvertScale *h, const VfHipFrame *in, VfHipFrame *out, void *stream);
/* n_frames frames laid out at a constant pitch: plane p of frame k lives at data[p] + k * pitch */
int vfhip_convertscale_process_device_batch (VfHipConvertScale *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, void *stream);
/* name of the kernel variant the current configuration dispatches to (for profiles / tests) */
const char *vfhip_convertscale_kernel_name (VfHipConvertScale *h);
void vfhip_convertscale_cleanup (VfHipConvertScale *h);   /* -cleanup: drop GPU resources, keep the handle */
void vfhip_convertscale_free (VfHipConvertScale *h);

/* ---- deinterlace (reference: MetalDeinterlaceRenderer, deinterlace/metaldeinterlacerenderer.h:29-54) ------- */
typedef enum {
  VFHIP_DEINTERLACE_BOB = 0,       /* VF_METAL_DEINTERLACE_BOB */
  VFHIP_DEINTERLACE_WEAVE = 1,
  VFHIP_DEINTERLACE_LINEAR = 2,    /* identical to bob in the reference (metaldeinterlace_shaders.h:148) */
  VFHIP_DEINTERLACE_GREEDYH = 3
} VfHipDeinterlaceMethod;

typedef struct {                   /* DeinterlaceParams, metaldeinterlacerenderer.h:36-40 */
  int32_t method;
  int32_t top_field_first;
  float motion_threshold;          /* greedyh: RGB euclidean distance in unorm units */
  int32_t reserved;
} VfHipDeinterlaceParams;

typedef struct VfHipDeinterlace VfHipDeinterlace;
VfHipDeinterlace *vfhip_deinterlace_new (int device);
int vfhip_deinterlace_configure (VfHipDeinterlace *h, const VfHipVideoInfo *info);            /* -configureWithInfo: (resets history) */
int vfhip_deinterlace_process (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *params);
/* device frames, asynchronous; the previous input frame is kept in an internal device buffer (stream-ordered copy) */
int vfhip_deinterlace_process_device (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out,
    const VfHipDeinterlaceParams *params, void *stream);
/* batch of n consecutive frames of ONE stream (frame k at data[p] + k * pitch): the history of frame k is frame k-1 of the
 * batch, of frame 0 the handle's stored history; afterwards the history is the last frame of the batch */
int vfhip_deinterlace_process_device_batch (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, const VfHipDeinterlaceParams *params, void *stream);
/* pipelined host path, like vfhip_convertscale_submit / _wait; frames are deinterlaced in submission order */
int vfhip_deinterlace_submit (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *params);
int vfhip_deinterlace_wait (VfHipDeinterlace *h);
int vfhip_deinterlace_in_flight (VfHipDeinterlace *h);
int vfhip_deinterlace_reset (VfHipDeinterlace *h);                                            /* drop the 1-frame history */
void vfhip_deinterlace_cleanup (VfHipDeinterlace *h);
void vfhip_deinterlace_free (VfHipDeinterlace *h);

/* ---- videofilter (reference: MetalVideoFilterRenderer, videofilter/metalvideofilterrenderer.h:30-70) -------- */
typedef struct {                   /* VideoFilterParams, metalvideofilterrenderer.h:30-46 */
  float brightness, contrast, saturation;
  float hue;                       /* radians */
  float gamma, sharpness, sepia, noise, vignette;
  int32_t invert;
  int32_t chroma_key_enabled;
  float chroma_key_r, chroma_key_g, chroma_key_b;
  float chroma_key_tolerance, chroma_key_smoothness;
  uint32_t frame_index;
} VfHipVideoFilterParams;

typedef struct VfHipVideoFilter VfHipVideoFilter;
VfHipVideoFilter *vfhip_videofilter_new (int device);
int vfhip_videofilter_configure (VfHipVideoFilter *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out);
int vfhip_videofilter_process (VfHipVideoFilter *h, const VfHipFrame *in, VfHipFrame *out, const VfHipVideoFilterParams *params);
int vfhip_videofilter_process_device (VfHipVideoFilter *h, const VfHipFrame *in, VfHipFrame *out,
    const VfHipVideoFilterParams *params, void *stream);
/* batch: frame k at data[p] + k * pitch, filtered with params->frame_index + k */
int vfhip_videofilter_process_device_batch (VfHipVideoFilter *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, const VfHipVideoFilterParams *params, void *stream);
/* pipelined host path, like vfhip_convertscale_submit / _wait (the parameters are copied at submit) */
int vfhip_videofilter_submit (VfHipVideoFilter *h, const VfHipFrame *in, VfHipFrame *out, const VfHipVideoFilterParams *params);
int vfhip_videofilter_wait (VfHipVideoFilter *h);
int vfhip_videofilter_in_flight (VfHipVideoFilter *h);
int vfhip_videofilter_load_lut (VfHipVideoFilter *h, const char *path);                       /* -loadLUTFromFile: (.cube, or a .png of size^2 x size slices) */
int vfhip_videofilter_set_lut (VfHipVideoFilter *h, const float *rgba, int size);             /* size^3 RGBA32F, R fastest */
void vfhip_videofilter_clear_lut (VfHipVideoFilter *h);
int vfhip_videofilter_lut_size (VfHipVideoFilter *h);
void vfhip_videofilter_cleanup (VfHipVideoFilter *h);
void vfhip_videofilter_free (VfHipVideoFilter *h);

/* ---- compositor (reference: MetalCompositorRenderer, compositor/metalcomprenderer.h:29-63) ------------------ */
typedef enum { VFHIP_BLEND_SOURCE = 0, VFHIP_BLEND_OVER = 1, VFHIP_BLEND_ADD = 2 } VfHipBlendMode;
typedef enum { VFHIP_BG_CHECKER = 0, VFHIP_BG_BLACK = 1, VFHIP_BG_WHITE = 2, VFHIP_BG_TRANSPARENT = 3 } VfHipBackground;

typedef struct {                   /* MetalPadInput, metalcomprenderer.h:43-49 */
  VfHipFrame frame;
  int32_t xpos, ypos, width, height;
  double alpha;
  int32_t blend_mode;
  int32_t reserved;
} VfHipPadInput;

typedef struct VfHipCompositor VfHipCompositor;
VfHipCompositor *vfhip_compositor_new (int device);
int vfhip_compositor_configure (VfHipCompositor *h, const VfHipVideoInfo *out);               /* -configureWithWidth:height:format: */
int vfhip_compositor_composite (VfHipCompositor *h, const VfHipPadInput *inputs, int count, int background, VfHipFrame *out);
int vfhip_compositor_composite_device (VfHipCompositor *h, const VfHipPadInput *inputs, int count, int background,
    VfHipFrame *out, void *stream);
/* batch: pad i's frame k at inputs[i].frame.data[p] + k * pad_frame_pitch[i] (0 = the same frame every time), output frame k
 * at out0->data[p] + k * out_frame_pitch; at most 16 pads */
int vfhip_compositor_composite_device_batch (VfHipCompositor *h, const VfHipPadInput *inputs, const size_t *pad_frame_pitch, int count,
    int background, VfHipFrame *out0, size_t out_frame_pitch, int n_frames, void *stream);
/* pipelined host path, two deep like the other elements' _submit / _wait: submit enqueues the uploads of every pad, the
 * kernel and the download and returns; wait blocks until the OLDEST output is complete.  Pad frames and the output frame
 * must stay valid (mapped) until the wait of their submit returns. */
int vfhip_compositor_submit (VfHipCompositor *h, const VfHipPadInput *inputs, int count, int background, VfHipFrame *out);
int vfhip_compositor_wait (VfHipCompositor *h);
int vfhip_compositor_in_flight (VfHipCompositor *h);
void vfhip_compositor_cleanup (VfHipCompositor *h);
void vfhip_compositor_free (VfHipCompositor *h);

/* ---- transform: flip / rotate / crop (reference: MetalTransformRenderer, transform/metaltransformrenderer.{h,m}) --
 * SURVEY.md §8f "next" item 2.  Output size == input size (the element is a GstVideoFilter with identical caps). */
typedef enum {
  VFHIP_TRANSFORM_IDENTITY = 0,   /* nick "none"                 (transform/gstvfmetaltransform.m:84-91) */
  VFHIP_TRANSFORM_90R = 1,        /* "clockwise" */
  VFHIP_TRANSFORM_180 = 2,        /* "rotate-180" */
  VFHIP_TRANSFORM_90L = 3,        /* "counterclockwise" */
  VFHIP_TRANSFORM_HORIZ = 4,      /* "horizontal-flip" */
  VFHIP_TRANSFORM_VERT = 5,       /* "vertical-flip" */
  VFHIP_TRANSFORM_UL_LR = 6,      /* "upper-left-diagonal" */
  VFHIP_TRANSFORM_UR_LL = 7       /* "upper-right-diagonal" */
} VfHipTransformMethod;

typedef struct {                  /* TransformParams */
  int32_t method;
  int32_t crop_top, crop_bottom, crop_left, crop_right;
  int32_t reserved[3];
} VfHipTransformParams;

typedef struct VfHipTransform VfHipTransform;
VfHipTransform *vfhip_transform_new (int device);
int vfhip_transform_configure (VfHipTransform *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out);
int vfhip_transform_process (VfHipTransform *h, const VfHipFrame *in, VfHipFrame *out, const VfHipTransformParams *params);
int vfhip_transform_process_device (VfHipTransform *h, const VfHipFrame *in, VfHipFrame *out,
    const VfHipTransformParams *params, void *stream);
int vfhip_transform_process_device_batch (VfHipTransform *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, const VfHipTransformParams *params, void *stream);
int vfhip_transform_submit (VfHipTransform *h, const VfHipFrame *in, VfHipFrame *out, const VfHipTransformParams *params);
int vfhip_transform_wait (VfHipTransform *h);
int vfhip_transform_in_flight (VfHipTransform *h);
void vfhip_transform_cleanup (VfHipTransform *h);
void vfhip_transform_free (VfHipTransform *h);

/* ---- overlay: a still image blended over the video (reference: MetalOverlayRenderer, overlay/metaloverlayrenderer.{h,m}) --
 * SURVEY.md §8f "next" item 4.  Output size == input size.  Without an image the frame only passes through the 8-bit
 * render target (the element is in passthrough then, gstvfmetaloverlay.m:94-99). */
typedef struct {
  float x, y;                 /* top-left corner of the image in frame pixels (OverlayParams, metaloverlayrenderer.h:30-36) */
  float width, height;        /* drawn size in pixels; <= 0: the image's own size */
  float alpha;                /* opacity 0..1, multiplies the image's alpha */
} VfHipOverlayParams;

typedef struct VfHipOverlay VfHipOverlay;
VfHipOverlay *vfhip_overlay_new (int device);
int vfhip_overlay_configure (VfHipOverlay *h, const VfHipVideoInfo *in, const VfHipVideoInfo *out);
int vfhip_overlay_load_image (VfHipOverlay *h, const char *path);     /* -loadImageFromFile: (PNG or JPEG; NULL / "" clears) */
int vfhip_overlay_set_image (VfHipOverlay *h, const uint8_t *rgba, int width, int height);   /* bytes as the shader sees them */
void vfhip_overlay_clear_image (VfHipOverlay *h);
int vfhip_overlay_image_size (VfHipOverlay *h, int *width, int *height);                     /* returns 1 when an image is loaded */
int vfhip_overlay_process (VfHipOverlay *h, const VfHipFrame *in, VfHipFrame *out, const VfHipOverlayParams *params);
int vfhip_overlay_process_device (VfHipOverlay *h, const VfHipFrame *in, VfHipFrame *out, const VfHipOverlayParams *params, void *stream);
int vfhip_overlay_process_device_batch (VfHipOverlay *h, const VfHipFrame *in0, VfHipFrame *out0, size_t in_frame_pitch,
    size_t out_frame_pitch, int n_frames, const VfHipOverlayParams *params, void *stream);
int vfhip_overlay_submit (VfHipOverlay *h, const VfHipFrame *in, VfHipFrame *out, const VfHipOverlayParams *params);
int vfhip_overlay_wait (VfHipOverlay *h);
int vfhip_overlay_in_flight (VfHipOverlay *h);
void vfhip_overlay_cleanup (VfHipOverlay *h);
void vfhip_overlay_free (VfHipOverlay *h);

#ifdef __cplusplus
}
#endif
#endif /* VFHIP_H */
