/* gst/gstvfhip.h — glue shared by the vfhip* element shells: GstVideoInfo / GstVideoFrame -> the C-ABI PODs of
 * include/vfhip.h, the plugin-wide debug category, and the small compatibility layer that lets the same sources
 * build against GStreamer 1.14 (this container) and >= 1.20 (what the reference requires, README.md:158).
 *
 * The shells keep the reference's element classes, pad templates, property names, nicks, ranges and defaults
 * (SURVEY.md §8b "Element-level API that must stay identical"); everything below `void *renderer` is libvfhip. */
#ifndef GST_VFHIP_H
#define GST_VFHIP_H

#include <gst/gst.h>
#include <gst/video/video.h>
#include "../../include/vfhip.h"

G_BEGIN_DECLS

GST_DEBUG_CATEGORY_EXTERN (gst_vfhip_debug);

/* element registration without GST_ELEMENT_REGISTER_* (absent before 1.20) */
gboolean gst_vfhip_convertscale_register (GstPlugin * plugin);
gboolean gst_vfhip_videofilter_register (GstPlugin * plugin);
gboolean gst_vfhip_deinterlace_register (GstPlugin * plugin);
gboolean gst_vfhip_compositor_register (GstPlugin * plugin);
gboolean gst_vfhip_transform_register (GstPlugin * plugin);
gboolean gst_vfhip_overlay_register (GstPlugin * plugin);

/* pinned host memory for GstBuffers (gstvfhipallocator.c) */
GstAllocator *gst_vfhip_pinned_allocator_get (void);
struct _GstBaseTransform;
gboolean gst_vfhip_propose_allocation (struct _GstBaseTransform * trans, GstQuery * decide_query, GstQuery * query,
    gboolean (*parent) (struct _GstBaseTransform *, GstQuery *, GstQuery *));
gboolean gst_vfhip_decide_allocation (struct _GstBaseTransform * trans, GstQuery * query,
    gboolean (*parent) (struct _GstBaseTransform *, GstQuery *));

/* hipHostRegister of recurring upstream system memories (gstvfhipallocator.c) */
typedef struct { guint registered, reused; gboolean disabled; } GstVfHipPinStats;
void gst_vfhip_pin_foreign_memory (GstBuffer * buf, GstVfHipPinStats * stats);

/* async-depth=1: one frame in flight across buffers (gstvfhipasync.c) */
typedef struct
{
  gint depth;                                   /* the async-depth property */
  int (*submit) (struct _GstBaseTransform * trans, const VfHipFrame * in, VfHipFrame * out);   /* libvfhip _submit with the element's parameters */
  int (*wait) (struct _GstBaseTransform * trans);                                              /* libvfhip _wait */
  struct { GstBuffer *inbuf, *outbuf; GstVideoFrame in, out; } pending[2];
  guint n, head;
  GstVfHipPinStats pin;
} GstVfHipAsync;
GstFlowReturn gst_vfhip_async_drain (struct _GstBaseTransform * trans, GstVfHipAsync * a, gboolean push);
GstFlowReturn gst_vfhip_async_generate_output (struct _GstBaseTransform * trans, GstBuffer ** outbuf, GstVfHipAsync * a, const GstVideoInfo * in_info,
    const GstVideoInfo * out_info, gboolean ready, GstFlowReturn (*parent) (struct _GstBaseTransform *, GstBuffer **));
gboolean gst_vfhip_async_sink_event (struct _GstBaseTransform * trans, GstEvent * event, GstVfHipAsync * a, gboolean (*parent) (struct _GstBaseTransform *, GstEvent *));
gboolean gst_vfhip_async_query (struct _GstBaseTransform * trans, GstPadDirection direction, GstQuery * query, GstVfHipAsync * a, const GstVideoInfo * out_info,
    gboolean (*parent) (struct _GstBaseTransform *, GstPadDirection, GstQuery *));
GParamSpec *gst_vfhip_async_depth_pspec (void);

/* device-resident buffers: caps feature memory:HIPMemory (gstvfhipmemory.c) */
#define GST_CAPS_FEATURE_MEMORY_HIP "memory:HIPMemory"
#define GST_MAP_VFHIP ((GstMapFlags) (GST_MAP_FLAG_LAST << 3))     /* map a device GstMemory to its DEVICE pointer */
/* template caps: both memories, device memory first */
#define GST_VFHIP_CAPS(formats) GST_VIDEO_CAPS_MAKE_WITH_FEATURES (GST_CAPS_FEATURE_MEMORY_HIP, formats) "; " GST_VIDEO_CAPS_MAKE (formats)
GstAllocator *gst_vfhip_device_allocator_get (gint device);
gint gst_vfhip_element_device (gpointer element);
GstMapFlags gst_vfhip_map_flag (GstBuffer * buf, gint device);
gboolean gst_vfhip_is_device_memory (GstMemory * mem);
gboolean gst_vfhip_caps_has_hip_feature (GstCaps * caps);
GstCaps *gst_vfhip_caps_both_memories (GstCaps * caps);
GstCaps *gst_vfhip_filter_transform_caps (struct _GstBaseTransform * trans, GstPadDirection direction, GstCaps * caps, GstCaps * filter);
GstFlowReturn gst_vfhip_filter_transform (struct _GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer * outbuf);

/* new, additive properties every vfhip element has */
#define GST_VFHIP_DEFAULT_DEVICE_ID (-1)       /* -1: $VFHIP_DEVICE, else GPU 0 */

static inline gint
gst_vfhip_format (GstVideoFormat f)
{
  switch (f) {
    case GST_VIDEO_FORMAT_BGRA: return VFHIP_FORMAT_BGRA;
    case GST_VIDEO_FORMAT_RGBA: return VFHIP_FORMAT_RGBA;
    case GST_VIDEO_FORMAT_NV12: return VFHIP_FORMAT_NV12;
    case GST_VIDEO_FORMAT_I420: return VFHIP_FORMAT_I420;
    case GST_VIDEO_FORMAT_UYVY: return VFHIP_FORMAT_UYVY;
    case GST_VIDEO_FORMAT_YUY2: return VFHIP_FORMAT_YUY2;
    default: return -1;
  }
}

/* matrix AND chroma siting come from the negotiated GstVideoInfo (SURVEY.md §8c rule 1): caps without
 * colorimetry got GStreamer's by-height default when the info was parsed */
static inline void
gst_vfhip_info (const GstVideoInfo * gi, VfHipVideoInfo * vi)
{
  memset (vi, 0, sizeof (*vi));
  vi->format = gst_vfhip_format (GST_VIDEO_INFO_FORMAT (gi));
  vi->width = GST_VIDEO_INFO_WIDTH (gi);
  vi->height = GST_VIDEO_INFO_HEIGHT (gi);
  switch (gi->colorimetry.matrix) {
    case GST_VIDEO_COLOR_MATRIX_BT709: vi->color_matrix = VFHIP_MATRIX_BT709; break;
    case GST_VIDEO_COLOR_MATRIX_BT2020: vi->color_matrix = VFHIP_MATRIX_BT2020; break;
    default: vi->color_matrix = VFHIP_MATRIX_BT601; break;
  }
  vi->chroma_site = ((gi->chroma_site & GST_VIDEO_CHROMA_SITE_H_COSITED) && !(gi->chroma_site & GST_VIDEO_CHROMA_SITE_V_COSITED))
      ? VFHIP_CHROMA_SITE_H_COSITED : VFHIP_CHROMA_SITE_CENTER;
}

/* frames are borrowed: the caller (base class or shell) maps before and unmaps after the libvfhip call */
static inline void
gst_vfhip_frame (GstVideoFrame * gf, VfHipFrame * vf)
{
  guint p;
  memset (vf, 0, sizeof (*vf));
  gst_vfhip_info (&gf->info, &vf->info);
  for (p = 0; p < GST_VIDEO_FRAME_N_PLANES (gf) && p < VFHIP_MAX_PLANES; p++) {
    vf->data[p] = GST_VIDEO_FRAME_PLANE_DATA (gf, p);
    vf->stride[p] = GST_VIDEO_FRAME_PLANE_STRIDE (gf, p);
  }
  if (gf->buffer && GST_BUFFER_FLAG_IS_SET (gf->buffer, GST_VIDEO_BUFFER_FLAG_TFF))
    vf->flags |= VFHIP_FRAME_FLAG_TFF;
  /* mapped with GST_MAP_VFHIP from a memory:HIPMemory buffer: the plane pointers are device pointers */
  if (gf->buffer && (gf->map[0].flags & GST_MAP_VFHIP) && gst_vfhip_is_device_memory (gst_buffer_peek_memory (gf->buffer, 0)))
    vf->flags |= VFHIP_FRAME_FLAG_DEVICE;
}

G_END_DECLS
#endif
