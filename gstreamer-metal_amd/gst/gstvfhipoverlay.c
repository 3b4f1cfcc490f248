/* gst/gstvfhipoverlay.c — `vfhipoverlay`: a still image (logo / watermark) over the video on an MI355X
 * (SURVEY.md §8f "next" item 4).
 *
 * Drop-in for the reference's vfmetaloverlay (overlay/gstvfmetaloverlay.{h,m}): GstVideoFilter subclass with identical
 * in/out caps { BGRA, RGBA, NV12, I420 }, properties location, x, y, width, height (0 = the image's own size), alpha,
 * relative-x / relative-y (fractions of the frame size that override x / y when >= 0) with the reference's ranges and
 * defaults (:375-420), passthrough until an image is loaded (:94-99), position resolved per frame from one snapshot taken
 * under the object lock (:176-200).  Images: PNG and JPEG (libvfhip's decoders; the reference reads them through ImageIO). */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include <gst/video/gstvideofilter.h>
#include "gstvfhip.h"

/* the element's own debug category, like the reference's (overlay/gstvfmetaloverlay.m); shared helpers log to `vfhip` */
GST_DEBUG_CATEGORY_STATIC (gst_vfhip_overlay_debug);
#define GST_CAT_DEFAULT gst_vfhip_overlay_debug
#define VFHIP_OV_FORMATS "{ BGRA, RGBA, NV12, I420 }"

typedef struct
{
  GstVideoFilter parent;
  VfHipOverlay *renderer;
  gint device_id, x, y, width, height;
  gdouble alpha, relative_x, relative_y;
  gchar *location;
  gboolean image_loaded, image_dirty;
  GstVfHipAsync async;                          /* async-depth=1 (gstvfhipasync.c) */
} GstVfHipOverlay;
typedef struct
{
  GstVideoFilterClass parent_class;
} GstVfHipOverlayClass;

enum { PROP_0, PROP_LOCATION, PROP_X, PROP_Y, PROP_WIDTH, PROP_HEIGHT, PROP_ALPHA, PROP_RELATIVE_X, PROP_RELATIVE_Y, PROP_DEVICE_ID, PROP_ASYNC_DEPTH };

static GstStaticPadTemplate ov_sink_template = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_OV_FORMATS)));
static GstStaticPadTemplate ov_src_template = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_OV_FORMATS)));

G_DEFINE_TYPE (GstVfHipOverlay, gst_vfhip_overlay, GST_TYPE_VIDEO_FILTER);
#define OV(obj) ((GstVfHipOverlay *) (obj))

static gboolean
ov_ensure_renderer (GstVfHipOverlay * self)
{
  if (!self->renderer && !(self->renderer = vfhip_overlay_new (self->device_id))) {
    GST_ERROR_OBJECT (self, "no HIP renderer: %s", vfhip_last_error_string ());
    return FALSE;
  }
  return TRUE;
}

/* (re)load the image named by `location`; streaming thread or state change, never under the object lock */
static void
ov_load_image (GstVfHipOverlay * self)
{
  gchar *path;
  gboolean loaded = FALSE;
  GST_OBJECT_LOCK (self);
  path = g_strdup (self->location);
  self->image_dirty = FALSE;
  GST_OBJECT_UNLOCK (self);
  if (ov_ensure_renderer (self)) {
    if (path && path[0]) {
      if (vfhip_overlay_load_image (self->renderer, path) == VFHIP_OK)
        loaded = TRUE;
      else
        GST_WARNING_OBJECT (self, "failed to load overlay image %s: %s", path, vfhip_last_error_string ());
    } else
      vfhip_overlay_clear_image (self->renderer);
  }
  g_free (path);
  GST_OBJECT_LOCK (self);
  self->image_loaded = loaded;
  GST_OBJECT_UNLOCK (self);
  gst_base_transform_set_passthrough (GST_BASE_TRANSFORM (self), !loaded);
}

static gboolean
ov_set_info (GstVideoFilter * filter, GstCaps * incaps, GstVideoInfo * in_info, GstCaps * outcaps, GstVideoInfo * out_info)
{
  GstVfHipOverlay *self = OV (filter);
  VfHipVideoInfo in, out;
  (void) incaps; (void) outcaps;
  GST_DEBUG_OBJECT (filter, "caps %" GST_PTR_FORMAT " -> %" GST_PTR_FORMAT, incaps, outcaps);
  if (!ov_ensure_renderer (self))
    return FALSE;
  gst_vfhip_info (in_info, &in);
  gst_vfhip_info (out_info, &out);
  if (vfhip_overlay_configure (self->renderer, &in, &out) != VFHIP_OK) {
    GST_ERROR_OBJECT (self, "configure failed: %s", vfhip_last_error_string ());
    return FALSE;
  }
  return TRUE;
}

static void
ov_params (GstVfHipOverlay * self, int frame_w, int frame_h, VfHipOverlayParams * p)
{
  gdouble rx, ry;
  gint ax, ay;
  GST_OBJECT_LOCK (self);                        /* one consistent snapshot per frame */
  p->alpha = (float) self->alpha;
  p->width = (float) self->width;
  p->height = (float) self->height;
  rx = self->relative_x; ry = self->relative_y;
  ax = self->x; ay = self->y;
  GST_OBJECT_UNLOCK (self);
  p->x = rx >= 0.0 ? (float) (rx * frame_w) : (float) ax;      /* relative overrides absolute */
  p->y = ry >= 0.0 ? (float) (ry * frame_h) : (float) ay;
}

static GstFlowReturn
ov_transform_frame (GstVideoFilter * filter, GstVideoFrame * in, GstVideoFrame * out)
{
  GstVfHipOverlay *self = OV (filter);
  VfHipOverlayParams p;
  VfHipFrame vin, vout;
  if (!self->renderer)
    return GST_FLOW_ERROR;
  ov_params (self, GST_VIDEO_FRAME_WIDTH (in), GST_VIDEO_FRAME_HEIGHT (in), &p);
  gst_vfhip_frame (in, &vin);
  gst_vfhip_frame (out, &vout);
  if (vfhip_overlay_process (self->renderer, &vin, &vout, &p) != VFHIP_OK) {
    GST_WARNING_OBJECT (self, "HIP processing failed: %s", vfhip_last_error_string ());
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

/* a new `location` while streaming is picked up before the next buffer is transformed */
static void
ov_before_transform (GstBaseTransform * trans, GstBuffer * buffer)
{
  GstVfHipOverlay *self = OV (trans);
  gboolean dirty;
  (void) buffer;
  GST_OBJECT_LOCK (self);
  dirty = self->image_dirty;
  GST_OBJECT_UNLOCK (self);
  if (dirty)
    ov_load_image (self);
}

static void
ov_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstVfHipOverlay *self = OV (object);
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_LOCATION:
      g_free (self->location);
      self->location = g_value_dup_string (value);
      self->image_dirty = TRUE;
      break;
    case PROP_X: self->x = g_value_get_int (value); break;
    case PROP_Y: self->y = g_value_get_int (value); break;
    case PROP_WIDTH: self->width = g_value_get_int (value); break;
    case PROP_HEIGHT: self->height = g_value_get_int (value); break;
    case PROP_ALPHA: self->alpha = g_value_get_double (value); break;
    case PROP_RELATIVE_X: self->relative_x = g_value_get_double (value); break;
    case PROP_RELATIVE_Y: self->relative_y = g_value_get_double (value); break;
    case PROP_DEVICE_ID: self->device_id = g_value_get_int (value); break;
    case PROP_ASYNC_DEPTH: self->async.depth = g_value_get_int (value); break;
    default:
      GST_OBJECT_UNLOCK (self);
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec);
      return;
  }
  GST_OBJECT_UNLOCK (self);
}

static void
ov_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstVfHipOverlay *self = OV (object);
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_LOCATION: g_value_set_string (value, self->location); break;
    case PROP_X: g_value_set_int (value, self->x); break;
    case PROP_Y: g_value_set_int (value, self->y); break;
    case PROP_WIDTH: g_value_set_int (value, self->width); break;
    case PROP_HEIGHT: g_value_set_int (value, self->height); break;
    case PROP_ALPHA: g_value_set_double (value, self->alpha); break;
    case PROP_RELATIVE_X: g_value_set_double (value, self->relative_x); break;
    case PROP_RELATIVE_Y: g_value_set_double (value, self->relative_y); break;
    case PROP_DEVICE_ID: g_value_set_int (value, self->device_id); break;
    case PROP_ASYNC_DEPTH: g_value_set_int (value, self->async.depth); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (self);
}

static gboolean
ov_start (GstBaseTransform * trans)
{
  ov_load_image (OV (trans));                    /* image (or its absence) decides passthrough, like the reference's start */
  return TRUE;
}

static gboolean
ov_stop (GstBaseTransform * trans)
{
  gst_vfhip_async_drain (trans, &OV (trans)->async, FALSE);          /* the streaming thread has stopped: frames in flight are dropped */
  if (OV (trans)->renderer)
    vfhip_overlay_cleanup (OV (trans)->renderer);
  return TRUE;
}

static void
ov_finalize (GObject * object)
{
  GstVfHipOverlay *self = OV (object);
  if (self->renderer)
    vfhip_overlay_free (self->renderer);
  self->renderer = NULL;
  g_free (self->location);
  G_OBJECT_CLASS (gst_vfhip_overlay_parent_class)->finalize (object);
}

static gboolean
ov_propose_allocation (GstBaseTransform * trans, GstQuery * decide_query, GstQuery * query)
{
  return gst_vfhip_propose_allocation (trans, decide_query, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_overlay_parent_class)->propose_allocation);
}

static gboolean
ov_decide_allocation (GstBaseTransform * trans, GstQuery * query)
{
  return gst_vfhip_decide_allocation (trans, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_overlay_parent_class)->decide_allocation);
}


/* ---- async-depth=1 (gstvfhipasync.c) -------------------------------------------------------------------------------- */
static int
ov_async_submit (GstBaseTransform * trans, const VfHipFrame * in, VfHipFrame * out)
{
  VfHipOverlayParams p;
  ov_params (OV (trans), in->info.width, in->info.height, &p);
  return vfhip_overlay_submit (OV (trans)->renderer, in, out, &p);
}

static int
ov_async_wait (GstBaseTransform * trans)
{
  return vfhip_overlay_wait (OV (trans)->renderer);
}

static GstFlowReturn
ov_generate_output (GstBaseTransform * trans, GstBuffer ** outbuf)
{
  GstVideoFilter *f = GST_VIDEO_FILTER_CAST (trans);
  return gst_vfhip_async_generate_output (trans, outbuf, &OV (trans)->async, &f->in_info, &f->out_info, f->negotiated && OV (trans)->renderer != NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_overlay_parent_class)->generate_output);
}

static gboolean
ov_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  return gst_vfhip_async_sink_event (trans, event, &OV (trans)->async, GST_BASE_TRANSFORM_CLASS (gst_vfhip_overlay_parent_class)->sink_event);
}

static gboolean
ov_query (GstBaseTransform * trans, GstPadDirection direction, GstQuery * query)
{
  GstVideoFilter *f = GST_VIDEO_FILTER_CAST (trans);
  return gst_vfhip_async_query (trans, direction, query, &OV (trans)->async, f->negotiated ? &f->out_info : NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_overlay_parent_class)->query);
}

static void
gst_vfhip_overlay_class_init (GstVfHipOverlayClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *bc = GST_BASE_TRANSFORM_CLASS (klass);
  const GParamFlags f = (GParamFlags) (G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS);
  oc->set_property = ov_set_property;
  oc->get_property = ov_get_property;
  oc->finalize = ov_finalize;
  bc->start = GST_DEBUG_FUNCPTR (ov_start);
  bc->stop = GST_DEBUG_FUNCPTR (ov_stop);
  bc->before_transform = GST_DEBUG_FUNCPTR (ov_before_transform);
  bc->propose_allocation = GST_DEBUG_FUNCPTR (ov_propose_allocation);
  bc->decide_allocation = GST_DEBUG_FUNCPTR (ov_decide_allocation);
  GST_VIDEO_FILTER_CLASS (klass)->set_info = GST_DEBUG_FUNCPTR (ov_set_info);
  GST_VIDEO_FILTER_CLASS (klass)->transform_frame = GST_DEBUG_FUNCPTR (ov_transform_frame);
  bc->transform_caps = GST_DEBUG_FUNCPTR (gst_vfhip_filter_transform_caps);
  bc->transform = GST_DEBUG_FUNCPTR (gst_vfhip_filter_transform);
  bc->generate_output = GST_DEBUG_FUNCPTR (ov_generate_output);
  bc->sink_event = GST_DEBUG_FUNCPTR (ov_sink_event);
  bc->query = GST_DEBUG_FUNCPTR (ov_query);

  g_object_class_install_property (oc, PROP_LOCATION, g_param_spec_string ("location", "Location", "Path to overlay image file (PNG or JPEG)", NULL, f));
  g_object_class_install_property (oc, PROP_X, g_param_spec_int ("x", "X Position", "Overlay X position in pixels", 0, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PROP_Y, g_param_spec_int ("y", "Y Position", "Overlay Y position in pixels", 0, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PROP_WIDTH, g_param_spec_int ("width", "Width", "Overlay width in pixels (0 = original image width)", 0, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PROP_HEIGHT, g_param_spec_int ("height", "Height", "Overlay height in pixels (0 = original image height)", 0, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PROP_ALPHA, g_param_spec_double ("alpha", "Alpha", "Overlay opacity (0.0 = transparent, 1.0 = opaque)", 0.0, 1.0, 1.0, f));
  g_object_class_install_property (oc, PROP_RELATIVE_X, g_param_spec_double ("relative-x", "Relative X",
          "Overlay X position as fraction of video width (-1 = use pixel x)", -1.0, 1.0, -1.0, f));
  g_object_class_install_property (oc, PROP_RELATIVE_Y, g_param_spec_double ("relative-y", "Relative Y",
          "Overlay Y position as fraction of video height (-1 = use pixel y)", -1.0, 1.0, -1.0, f));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "GPU ordinal to run on (-1: $VFHIP_DEVICE, else 0)", -1, 63, GST_VFHIP_DEFAULT_DEVICE_ID, f));

  g_object_class_install_property (oc, PROP_ASYNC_DEPTH, gst_vfhip_async_depth_pspec ());
  gst_element_class_add_static_pad_template (ec, &ov_sink_template);
  gst_element_class_add_static_pad_template (ec, &ov_src_template);
  gst_element_class_set_static_metadata (ec, "HIP Video Overlay", "Filter/Effect/Video",
      "MI355X-accelerated image overlay (logo / watermark) on video", "vfhip");
  GST_DEBUG_CATEGORY_INIT (gst_vfhip_overlay_debug, "vfhipoverlay", 0, "vfhipoverlay element");
}

static void
gst_vfhip_overlay_init (GstVfHipOverlay * self)
{
  self->alpha = 1.0;
  self->relative_x = self->relative_y = -1.0;
  self->device_id = GST_VFHIP_DEFAULT_DEVICE_ID;
  self->async.submit = ov_async_submit;
  self->async.wait = ov_async_wait;
  gst_base_transform_set_passthrough (GST_BASE_TRANSFORM (self), TRUE);
}

gboolean
gst_vfhip_overlay_register (GstPlugin * plugin)
{
  gboolean ok = gst_element_register (plugin, "vfhipoverlay", GST_RANK_NONE, gst_vfhip_overlay_get_type ());
#ifdef VFHIP_REGISTER_VFMETAL_NAMES
  ok &= gst_element_register (plugin, "vfmetaloverlay", GST_RANK_NONE, gst_vfhip_overlay_get_type ());
#endif
  return ok;
}
