/* gst/gstvfhipdeinterlace.c — `vfhipdeinterlace`: bob / weave / linear / greedy-H deinterlacing on an MI355X.
 *
 * Drop-in for the reference's vfmetaldeinterlace (deinterlace/gstvfmetaldeinterlace.{h,m}): GstVideoFilter subclass,
 * templates { BGRA, RGBA, NV12, I420 } (:43-55), properties method {bob, weave, linear, greedyh}, field-layout {auto,
 * top-field-first, bottom-field-first}, motion-threshold 0..1 default 0.1 (:323-339); in auto mode the field order
 * comes from GST_VIDEO_BUFFER_FLAG_TFF on the input buffer (:176-184).  The one-frame history lives in libvfhip and
 * is dropped on caps change and on stop, like the reference's _prevFrameRGBA. */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include <gst/video/gstvideofilter.h>
#include "gstvfhip.h"

/* the element's own debug category, like the reference's (deinterlace/gstvfmetaldeinterlace.m:351-352); shared helpers log to `vfhip` */
GST_DEBUG_CATEGORY_STATIC (gst_vfhip_deinterlace_debug);
#define GST_CAT_DEFAULT gst_vfhip_deinterlace_debug
#define VFHIP_DI_FORMATS "{ BGRA, RGBA, NV12, I420 }"

typedef struct
{
  GstVideoFilter parent;
  VfHipDeinterlace *renderer;
  gint device_id, method, field_layout;
  gdouble motion_threshold;
  GstVfHipAsync async;                          /* async-depth=1 (gstvfhipasync.c) */
} GstVfHipDeinterlace;

typedef struct
{
  GstVideoFilterClass parent_class;
} GstVfHipDeinterlaceClass;

enum { PROP_0, PROP_METHOD, PROP_FIELD_LAYOUT, PROP_MOTION_THRESHOLD, PROP_DEVICE_ID, PROP_ASYNC_DEPTH };
enum { FIELDS_AUTO = 0, FIELDS_TFF = 1, FIELDS_BFF = 2 };

static GstStaticPadTemplate di_sink_template = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_DI_FORMATS)));
static GstStaticPadTemplate di_src_template = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_DI_FORMATS)));

static GType
di_method_type (void)
{
  static gsize t = 0;
  static const GEnumValue v[] = {
    {VFHIP_DEINTERLACE_BOB, "Bob (line doubling from one field)", "bob"},
    {VFHIP_DEINTERLACE_WEAVE, "Weave (merge fields of consecutive frames)", "weave"},
    {VFHIP_DEINTERLACE_LINEAR, "Linear interpolation", "linear"},
    {VFHIP_DEINTERLACE_GREEDYH, "Greedy-H motion adaptive", "greedyh"},
    {0, NULL, NULL}
  };
  if (g_once_init_enter (&t))
    g_once_init_leave (&t, g_enum_register_static ("GstVfHipDeinterlaceMethod", v));
  return (GType) t;
}

static GType
di_fields_type (void)
{
  static gsize t = 0;
  static const GEnumValue v[] = {
    {FIELDS_AUTO, "Auto detect from buffer flags", "auto"},
    {FIELDS_TFF, "Top field first", "top-field-first"},
    {FIELDS_BFF, "Bottom field first", "bottom-field-first"},
    {0, NULL, NULL}
  };
  if (g_once_init_enter (&t))
    g_once_init_leave (&t, g_enum_register_static ("GstVfHipDeinterlaceFieldLayout", v));
  return (GType) t;
}

G_DEFINE_TYPE (GstVfHipDeinterlace, gst_vfhip_deinterlace, GST_TYPE_VIDEO_FILTER);
#define DI(obj) ((GstVfHipDeinterlace *) (obj))

static gboolean
di_set_info (GstVideoFilter * filter, GstCaps * incaps, GstVideoInfo * in_info, GstCaps * outcaps, GstVideoInfo * out_info)
{
  GstVfHipDeinterlace *self = DI (filter);
  VfHipVideoInfo info;
  (void) incaps; (void) outcaps; (void) out_info;
  GST_DEBUG_OBJECT (filter, "caps %" GST_PTR_FORMAT " -> %" GST_PTR_FORMAT, incaps, outcaps);
  if (!self->renderer && !(self->renderer = vfhip_deinterlace_new (self->device_id))) {
    GST_ERROR_OBJECT (self, "no HIP renderer: %s", vfhip_last_error_string ());
    return FALSE;
  }
  gst_vfhip_info (in_info, &info);
  if (vfhip_deinterlace_configure (self->renderer, &info) != VFHIP_OK) {
    GST_ERROR_OBJECT (self, "configure failed: %s", vfhip_last_error_string ());
    return FALSE;
  }
  return TRUE;
}

/* one snapshot of the properties per frame; field order from the property, else from the buffer's TFF flag (reference :176-184) */
static void
di_params (GstVfHipDeinterlace * self, const VfHipFrame * in, VfHipDeinterlaceParams * p)
{
  gint layout;
  memset (p, 0, sizeof (*p));
  GST_OBJECT_LOCK (self);
  layout = self->field_layout;
  p->method = self->method;
  p->motion_threshold = (float) self->motion_threshold;
  GST_OBJECT_UNLOCK (self);
  if (layout == FIELDS_TFF) p->top_field_first = 1;
  else if (layout == FIELDS_BFF) p->top_field_first = 0;
  else p->top_field_first = (in->flags & VFHIP_FRAME_FLAG_TFF) != 0;
}

static GstFlowReturn
di_transform_frame (GstVideoFilter * filter, GstVideoFrame * in, GstVideoFrame * out)
{
  GstVfHipDeinterlace *self = DI (filter);
  VfHipDeinterlaceParams p;
  VfHipFrame vin, vout;
  if (!self->renderer) {
    GST_WARNING_OBJECT (self, "no HIP renderer");
    return GST_FLOW_ERROR;
  }
  gst_vfhip_frame (in, &vin);
  gst_vfhip_frame (out, &vout);
  di_params (self, &vin, &p);
  if (vfhip_deinterlace_process (self->renderer, &vin, &vout, &p) != VFHIP_OK) {
    GST_WARNING_OBJECT (self, "HIP processing failed: %s", vfhip_last_error_string ());
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

static void
di_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstVfHipDeinterlace *self = DI (object);
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_METHOD: self->method = g_value_get_enum (value); break;
    case PROP_FIELD_LAYOUT: self->field_layout = g_value_get_enum (value); break;
    case PROP_MOTION_THRESHOLD: self->motion_threshold = g_value_get_double (value); break;
    case PROP_DEVICE_ID: self->device_id = g_value_get_int (value); break;
    case PROP_ASYNC_DEPTH: self->async.depth = g_value_get_int (value); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (self);
}

static void
di_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstVfHipDeinterlace *self = DI (object);
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_METHOD: g_value_set_enum (value, self->method); break;
    case PROP_FIELD_LAYOUT: g_value_set_enum (value, self->field_layout); break;
    case PROP_MOTION_THRESHOLD: g_value_set_double (value, self->motion_threshold); break;
    case PROP_DEVICE_ID: g_value_set_int (value, self->device_id); break;
    case PROP_ASYNC_DEPTH: g_value_set_int (value, self->async.depth); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (self);
}

static gboolean
di_stop (GstBaseTransform * trans)
{
  GstVfHipDeinterlace *self = DI (trans);
  gst_vfhip_async_drain (trans, &self->async, FALSE);          /* the streaming thread has stopped: frames in flight are dropped */
  if (self->renderer)
    vfhip_deinterlace_cleanup (self->renderer);     /* also forgets the previous frame */
  return TRUE;
}

static void
di_finalize (GObject * object)
{
  GstVfHipDeinterlace *self = DI (object);
  if (self->renderer)
    vfhip_deinterlace_free (self->renderer);
  self->renderer = NULL;
  G_OBJECT_CLASS (gst_vfhip_deinterlace_parent_class)->finalize (object);
}


static gboolean
de_propose_allocation (GstBaseTransform * trans, GstQuery * decide_query, GstQuery * query)
{
  return gst_vfhip_propose_allocation (trans, decide_query, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_deinterlace_parent_class)->propose_allocation);
}

static gboolean
de_decide_allocation (GstBaseTransform * trans, GstQuery * query)
{
  return gst_vfhip_decide_allocation (trans, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_deinterlace_parent_class)->decide_allocation);
}


/* ---- async-depth=1 (gstvfhipasync.c) -------------------------------------------------------------------------------- */
static int
di_async_submit (GstBaseTransform * trans, const VfHipFrame * in, VfHipFrame * out)
{
  VfHipDeinterlaceParams p;
  di_params (DI (trans), in, &p);
  return vfhip_deinterlace_submit (DI (trans)->renderer, in, out, &p);
}

static int
di_async_wait (GstBaseTransform * trans)
{
  return vfhip_deinterlace_wait (DI (trans)->renderer);
}

static GstFlowReturn
di_generate_output (GstBaseTransform * trans, GstBuffer ** outbuf)
{
  GstVideoFilter *f = GST_VIDEO_FILTER_CAST (trans);
  return gst_vfhip_async_generate_output (trans, outbuf, &DI (trans)->async, &f->in_info, &f->out_info, f->negotiated && DI (trans)->renderer != NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_deinterlace_parent_class)->generate_output);
}

static gboolean
di_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  return gst_vfhip_async_sink_event (trans, event, &DI (trans)->async, GST_BASE_TRANSFORM_CLASS (gst_vfhip_deinterlace_parent_class)->sink_event);
}

static gboolean
di_query (GstBaseTransform * trans, GstPadDirection direction, GstQuery * query)
{
  GstVideoFilter *f = GST_VIDEO_FILTER_CAST (trans);
  return gst_vfhip_async_query (trans, direction, query, &DI (trans)->async, f->negotiated ? &f->out_info : NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_deinterlace_parent_class)->query);
}

static void
gst_vfhip_deinterlace_class_init (GstVfHipDeinterlaceClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  oc->set_property = di_set_property;
  oc->get_property = di_get_property;
  oc->finalize = di_finalize;
  GST_BASE_TRANSFORM_CLASS (klass)->propose_allocation = GST_DEBUG_FUNCPTR (de_propose_allocation);
  GST_BASE_TRANSFORM_CLASS (klass)->decide_allocation = GST_DEBUG_FUNCPTR (de_decide_allocation);
  GST_BASE_TRANSFORM_CLASS (klass)->stop = GST_DEBUG_FUNCPTR (di_stop);
  GST_VIDEO_FILTER_CLASS (klass)->set_info = GST_DEBUG_FUNCPTR (di_set_info);
  GST_VIDEO_FILTER_CLASS (klass)->transform_frame = GST_DEBUG_FUNCPTR (di_transform_frame);
  /* memory:HIPMemory on either pad (gstvfhipmemory.c): same video caps in both memories, device buffers mapped in place */
  GST_BASE_TRANSFORM_CLASS (klass)->transform_caps = GST_DEBUG_FUNCPTR (gst_vfhip_filter_transform_caps);
  GST_BASE_TRANSFORM_CLASS (klass)->transform = GST_DEBUG_FUNCPTR (gst_vfhip_filter_transform);
  GST_BASE_TRANSFORM_CLASS (klass)->generate_output = GST_DEBUG_FUNCPTR (di_generate_output);
  GST_BASE_TRANSFORM_CLASS (klass)->sink_event = GST_DEBUG_FUNCPTR (di_sink_event);
  GST_BASE_TRANSFORM_CLASS (klass)->query = GST_DEBUG_FUNCPTR (di_query);

  g_object_class_install_property (oc, PROP_METHOD, g_param_spec_enum ("method", "Method", "Deinterlacing algorithm",
          di_method_type (), VFHIP_DEINTERLACE_BOB, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_FIELD_LAYOUT, g_param_spec_enum ("field-layout", "Field Layout",
          "Field order (top-first or bottom-first)", di_fields_type (), FIELDS_AUTO, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_MOTION_THRESHOLD, g_param_spec_double ("motion-threshold", "Motion Threshold",
          "Motion detection threshold for greedy-H method (0.0 to 1.0)", 0.0, 1.0, 0.1, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "GPU ordinal to run on (-1: $VFHIP_DEVICE, else 0)", -1, 63, GST_VFHIP_DEFAULT_DEVICE_ID, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  g_object_class_install_property (oc, PROP_ASYNC_DEPTH, gst_vfhip_async_depth_pspec ());
  gst_element_class_add_static_pad_template (ec, &di_sink_template);
  gst_element_class_add_static_pad_template (ec, &di_src_template);
  gst_element_class_set_static_metadata (ec, "HIP Video Deinterlace", "Filter/Effect/Video/Deinterlace",
      "MI355X-accelerated deinterlacing (bob, weave, linear, greedy-H)", "vfhip");
  GST_DEBUG_CATEGORY_INIT (gst_vfhip_deinterlace_debug, "vfhipdeinterlace", 0, "vfhipdeinterlace element");
}

static void
gst_vfhip_deinterlace_init (GstVfHipDeinterlace * self)
{
  self->method = VFHIP_DEINTERLACE_BOB;
  self->field_layout = FIELDS_AUTO;
  self->motion_threshold = 0.1;
  self->device_id = GST_VFHIP_DEFAULT_DEVICE_ID;
  self->async.submit = di_async_submit;
  self->async.wait = di_async_wait;
}

gboolean
gst_vfhip_deinterlace_register (GstPlugin * plugin)
{
  gboolean ok = gst_element_register (plugin, "vfhipdeinterlace", GST_RANK_NONE, gst_vfhip_deinterlace_get_type ());
#ifdef VFHIP_REGISTER_VFMETAL_NAMES
  ok &= gst_element_register (plugin, "vfmetaldeinterlace", GST_RANK_NONE, gst_vfhip_deinterlace_get_type ());
#endif
  return ok;
}
