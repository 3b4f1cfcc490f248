/* gst/gstvfhiptransform.c — `vfhiptransform`: flip / rotate / crop on an MI355X (SURVEY.md §8f "next" item 2).
 *
 * Drop-in for the reference's vfmetaltransform (transform/gstvfmetaltransform.{h,m}): GstVideoFilter subclass with
 * identical in/out caps { BGRA, RGBA, NV12, I420 }, property method with the eight videoflip-style nicks
 * (:84-91) and crop-top / crop-bottom / crop-left / crop-right in pixels (0 .. G_MAXINT, default 0, :329-357),
 * passthrough while method == none and all crops are 0 (:113-126). */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include <gst/video/gstvideofilter.h>
#include "gstvfhip.h"

/* the element's own debug category, like the reference's (transform/gstvfmetaltransform.m); shared helpers log to `vfhip` */
GST_DEBUG_CATEGORY_STATIC (gst_vfhip_transform_debug);
#define GST_CAT_DEFAULT gst_vfhip_transform_debug
#define VFHIP_TR_FORMATS "{ BGRA, RGBA, NV12, I420 }"

typedef struct
{
  GstVideoFilter parent;
  VfHipTransform *renderer;
  gint device_id, method, crop_top, crop_bottom, crop_left, crop_right;
  GstVfHipAsync async;                          /* async-depth=1 (gstvfhipasync.c) */
} GstVfHipTransform;
typedef struct
{
  GstVideoFilterClass parent_class;
} GstVfHipTransformClass;

enum { PROP_0, PROP_METHOD, PROP_CROP_TOP, PROP_CROP_BOTTOM, PROP_CROP_LEFT, PROP_CROP_RIGHT, PROP_DEVICE_ID, PROP_ASYNC_DEPTH };

static GstStaticPadTemplate tr_sink_template = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_TR_FORMATS)));
static GstStaticPadTemplate tr_src_template = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_TR_FORMATS)));

static GType
tr_method_type (void)
{
  static gsize t = 0;
  static const GEnumValue v[] = {
    {VFHIP_TRANSFORM_IDENTITY, "Identity (no rotation)", "none"},
    {VFHIP_TRANSFORM_90R, "Rotate clockwise 90 degrees", "clockwise"},
    {VFHIP_TRANSFORM_180, "Rotate 180 degrees", "rotate-180"},
    {VFHIP_TRANSFORM_90L, "Rotate counter-clockwise 90 degrees", "counterclockwise"},
    {VFHIP_TRANSFORM_HORIZ, "Flip horizontally", "horizontal-flip"},
    {VFHIP_TRANSFORM_VERT, "Flip vertically", "vertical-flip"},
    {VFHIP_TRANSFORM_UL_LR, "Flip across upper left/lower right diagonal", "upper-left-diagonal"},
    {VFHIP_TRANSFORM_UR_LL, "Flip across upper right/lower left diagonal", "upper-right-diagonal"},
    {0, NULL, NULL}
  };
  if (g_once_init_enter (&t))
    g_once_init_leave (&t, g_enum_register_static ("GstVfHipTransformMethod", v));
  return (GType) t;
}

G_DEFINE_TYPE (GstVfHipTransform, gst_vfhip_transform, GST_TYPE_VIDEO_FILTER);
#define TR(obj) ((GstVfHipTransform *) (obj))

static void
tr_update_passthrough (GstVfHipTransform * self)
{
  gboolean idle;
  GST_OBJECT_LOCK (self);
  idle = self->method == VFHIP_TRANSFORM_IDENTITY && !self->crop_top && !self->crop_bottom && !self->crop_left && !self->crop_right;
  GST_OBJECT_UNLOCK (self);
  gst_base_transform_set_passthrough (GST_BASE_TRANSFORM (self), idle);
}

static gboolean
tr_set_info (GstVideoFilter * filter, GstCaps * incaps, GstVideoInfo * in_info, GstCaps * outcaps, GstVideoInfo * out_info)
{
  GstVfHipTransform *self = TR (filter);
  VfHipVideoInfo in, out;
  (void) incaps; (void) outcaps;
  GST_DEBUG_OBJECT (filter, "caps %" GST_PTR_FORMAT " -> %" GST_PTR_FORMAT, incaps, outcaps);
  if (!self->renderer && !(self->renderer = vfhip_transform_new (self->device_id))) {
    GST_ERROR_OBJECT (self, "no HIP renderer: %s", vfhip_last_error_string ());
    return FALSE;
  }
  gst_vfhip_info (in_info, &in);
  gst_vfhip_info (out_info, &out);
  if (vfhip_transform_configure (self->renderer, &in, &out) != VFHIP_OK) {
    GST_ERROR_OBJECT (self, "configure failed: %s", vfhip_last_error_string ());
    return FALSE;
  }
  return TRUE;
}

static GstFlowReturn
tr_transform_frame (GstVideoFilter * filter, GstVideoFrame * in, GstVideoFrame * out)
{
  GstVfHipTransform *self = TR (filter);
  VfHipTransformParams p;
  VfHipFrame vin, vout;
  if (!self->renderer)
    return GST_FLOW_ERROR;
  memset (&p, 0, sizeof (p));
  GST_OBJECT_LOCK (self);
  p.method = self->method;
  p.crop_top = self->crop_top; p.crop_bottom = self->crop_bottom; p.crop_left = self->crop_left; p.crop_right = self->crop_right;
  GST_OBJECT_UNLOCK (self);
  gst_vfhip_frame (in, &vin);
  gst_vfhip_frame (out, &vout);
  if (vfhip_transform_process (self->renderer, &vin, &vout, &p) != VFHIP_OK) {
    GST_WARNING_OBJECT (self, "HIP processing failed: %s", vfhip_last_error_string ());
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

static void
tr_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstVfHipTransform *self = TR (object);
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_METHOD: self->method = g_value_get_enum (value); break;
    case PROP_CROP_TOP: self->crop_top = g_value_get_int (value); break;
    case PROP_CROP_BOTTOM: self->crop_bottom = g_value_get_int (value); break;
    case PROP_CROP_LEFT: self->crop_left = g_value_get_int (value); break;
    case PROP_CROP_RIGHT: self->crop_right = g_value_get_int (value); break;
    case PROP_DEVICE_ID: self->device_id = g_value_get_int (value); break;
    case PROP_ASYNC_DEPTH: self->async.depth = g_value_get_int (value); break;
    default:
      GST_OBJECT_UNLOCK (self);
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec);
      return;
  }
  GST_OBJECT_UNLOCK (self);
  tr_update_passthrough (self);
}

static void
tr_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstVfHipTransform *self = TR (object);
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_METHOD: g_value_set_enum (value, self->method); break;
    case PROP_CROP_TOP: g_value_set_int (value, self->crop_top); break;
    case PROP_CROP_BOTTOM: g_value_set_int (value, self->crop_bottom); break;
    case PROP_CROP_LEFT: g_value_set_int (value, self->crop_left); break;
    case PROP_CROP_RIGHT: g_value_set_int (value, self->crop_right); break;
    case PROP_DEVICE_ID: g_value_set_int (value, self->device_id); break;
    case PROP_ASYNC_DEPTH: g_value_set_int (value, self->async.depth); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (self);
}

static gboolean
tr_start (GstBaseTransform * trans)
{
  tr_update_passthrough (TR (trans));
  return TRUE;
}

static gboolean
tr_stop (GstBaseTransform * trans)
{
  gst_vfhip_async_drain (trans, &TR (trans)->async, FALSE);          /* the streaming thread has stopped: frames in flight are dropped */
  if (TR (trans)->renderer)
    vfhip_transform_cleanup (TR (trans)->renderer);
  return TRUE;
}

static void
tr_finalize (GObject * object)
{
  GstVfHipTransform *self = TR (object);
  if (self->renderer)
    vfhip_transform_free (self->renderer);
  self->renderer = NULL;
  G_OBJECT_CLASS (gst_vfhip_transform_parent_class)->finalize (object);
}

static gboolean
tr_propose_allocation (GstBaseTransform * trans, GstQuery * decide_query, GstQuery * query)
{
  return gst_vfhip_propose_allocation (trans, decide_query, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_transform_parent_class)->propose_allocation);
}

static gboolean
tr_decide_allocation (GstBaseTransform * trans, GstQuery * query)
{
  return gst_vfhip_decide_allocation (trans, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_transform_parent_class)->decide_allocation);
}


/* ---- async-depth=1 (gstvfhipasync.c) -------------------------------------------------------------------------------- */
static void
tr_params (GstVfHipTransform * self, VfHipTransformParams * p)
{
  memset (p, 0, sizeof (*p));
  GST_OBJECT_LOCK (self);
  p->method = self->method;
  p->crop_top = self->crop_top; p->crop_bottom = self->crop_bottom; p->crop_left = self->crop_left; p->crop_right = self->crop_right;
  GST_OBJECT_UNLOCK (self);
}

static int
tr_async_submit (GstBaseTransform * trans, const VfHipFrame * in, VfHipFrame * out)
{
  VfHipTransformParams p;
  tr_params (TR (trans), &p);
  return vfhip_transform_submit (TR (trans)->renderer, in, out, &p);
}

static int
tr_async_wait (GstBaseTransform * trans)
{
  return vfhip_transform_wait (TR (trans)->renderer);
}

static GstFlowReturn
tr_generate_output (GstBaseTransform * trans, GstBuffer ** outbuf)
{
  GstVideoFilter *f = GST_VIDEO_FILTER_CAST (trans);
  return gst_vfhip_async_generate_output (trans, outbuf, &TR (trans)->async, &f->in_info, &f->out_info, f->negotiated && TR (trans)->renderer != NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_transform_parent_class)->generate_output);
}

static gboolean
tr_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  return gst_vfhip_async_sink_event (trans, event, &TR (trans)->async, GST_BASE_TRANSFORM_CLASS (gst_vfhip_transform_parent_class)->sink_event);
}

static gboolean
tr_query (GstBaseTransform * trans, GstPadDirection direction, GstQuery * query)
{
  GstVideoFilter *f = GST_VIDEO_FILTER_CAST (trans);
  return gst_vfhip_async_query (trans, direction, query, &TR (trans)->async, f->negotiated ? &f->out_info : NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_transform_parent_class)->query);
}

static void
gst_vfhip_transform_class_init (GstVfHipTransformClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *bc = GST_BASE_TRANSFORM_CLASS (klass);
  const GParamFlags f = (GParamFlags) (G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS);
  oc->set_property = tr_set_property;
  oc->get_property = tr_get_property;
  oc->finalize = tr_finalize;
  bc->start = GST_DEBUG_FUNCPTR (tr_start);
  bc->stop = GST_DEBUG_FUNCPTR (tr_stop);
  bc->propose_allocation = GST_DEBUG_FUNCPTR (tr_propose_allocation);
  bc->decide_allocation = GST_DEBUG_FUNCPTR (tr_decide_allocation);
  GST_VIDEO_FILTER_CLASS (klass)->set_info = GST_DEBUG_FUNCPTR (tr_set_info);
  GST_VIDEO_FILTER_CLASS (klass)->transform_frame = GST_DEBUG_FUNCPTR (tr_transform_frame);
  /* memory:HIPMemory on either pad (gstvfhipmemory.c): same video caps in both memories, device buffers mapped in place */
  GST_BASE_TRANSFORM_CLASS (klass)->transform_caps = GST_DEBUG_FUNCPTR (gst_vfhip_filter_transform_caps);
  GST_BASE_TRANSFORM_CLASS (klass)->transform = GST_DEBUG_FUNCPTR (gst_vfhip_filter_transform);
  GST_BASE_TRANSFORM_CLASS (klass)->generate_output = GST_DEBUG_FUNCPTR (tr_generate_output);
  GST_BASE_TRANSFORM_CLASS (klass)->sink_event = GST_DEBUG_FUNCPTR (tr_sink_event);
  GST_BASE_TRANSFORM_CLASS (klass)->query = GST_DEBUG_FUNCPTR (tr_query);

  g_object_class_install_property (oc, PROP_METHOD, g_param_spec_enum ("method", "Method", "Flip/rotation method", tr_method_type (),
          VFHIP_TRANSFORM_IDENTITY, f));
  g_object_class_install_property (oc, PROP_CROP_TOP, g_param_spec_int ("crop-top", "Crop Top", "Pixels to crop from the top edge", 0, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PROP_CROP_BOTTOM, g_param_spec_int ("crop-bottom", "Crop Bottom", "Pixels to crop from the bottom edge", 0, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PROP_CROP_LEFT, g_param_spec_int ("crop-left", "Crop Left", "Pixels to crop from the left edge", 0, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PROP_CROP_RIGHT, g_param_spec_int ("crop-right", "Crop Right", "Pixels to crop from the right edge", 0, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "GPU ordinal to run on (-1: $VFHIP_DEVICE, else 0)", -1, 63, GST_VFHIP_DEFAULT_DEVICE_ID, f));

  g_object_class_install_property (oc, PROP_ASYNC_DEPTH, gst_vfhip_async_depth_pspec ());
  gst_element_class_add_static_pad_template (ec, &tr_sink_template);
  gst_element_class_add_static_pad_template (ec, &tr_src_template);
  gst_element_class_set_static_metadata (ec, "HIP Video Transform", "Filter/Effect/Video",
      "MI355X-accelerated video flip, rotation and crop", "vfhip");
  GST_DEBUG_CATEGORY_INIT (gst_vfhip_transform_debug, "vfhiptransform", 0, "vfhiptransform element");
}

static void
gst_vfhip_transform_init (GstVfHipTransform * self)
{
  self->method = VFHIP_TRANSFORM_IDENTITY;
  self->device_id = GST_VFHIP_DEFAULT_DEVICE_ID;
  self->async.submit = tr_async_submit;
  self->async.wait = tr_async_wait;
}

gboolean
gst_vfhip_transform_register (GstPlugin * plugin)
{
  gboolean ok = gst_element_register (plugin, "vfhiptransform", GST_RANK_NONE, gst_vfhip_transform_get_type ());
#ifdef VFHIP_REGISTER_VFMETAL_NAMES
  ok &= gst_element_register (plugin, "vfmetaltransform", GST_RANK_NONE, gst_vfhip_transform_get_type ());
#endif
  return ok;
}
