/* gst/gstvfhipconvertscale.c — `vfhipconvertscale`: colourspace conversion + scaling on an MI355X.
 *
 * Drop-in for the reference's vfmetalconvertscale (convertscale/gstvfmetalconvertscale.{h,m}): GstBaseTransform
 * subclass, sink/src templates { BGRA, RGBA, NV12, I420, UYVY, YUY2 } (:48-60), properties method {bilinear,
 * nearest} / add-borders / border-color with the same ranges and defaults (:510-526), rank NONE, passthrough when
 * format and size are unchanged (:279-280), DAR-preserving fixation (:160-248).  The renderer behind it is
 * libvfhip's vfhip_convertscale_* (include/vfhip.h) instead of MetalConvertScaleRenderer.
 *
 * Additive properties: device-id (GPU ordinal, -1 = $VFHIP_DEVICE or 0) and numerics {gst-exact, metal, gst-exact-strict}.
 * Unlike the reference, property changes after negotiation reconfigure under the object lock
 * (the reference reconfigures unlocked from the application thread, :392-403). */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include <gst/base/gstbasetransform.h>
#include "gstvfhip.h"

/* the element's own debug category, like the reference's (convertscale/gstvfmetalconvertscale.m:538-539); shared helpers log to `vfhip` */
GST_DEBUG_CATEGORY_STATIC (gst_vfhip_convertscale_debug);
#define GST_CAT_DEFAULT gst_vfhip_convertscale_debug
#define VFHIP_CS_FORMATS "{ BGRA, RGBA, NV12, I420, UYVY, YUY2 }"

typedef struct
{
  GstBaseTransform parent;
  VfHipConvertScale *renderer;
  gint device_id;
  GstVideoInfo in_info, out_info;
  gboolean negotiated, passthrough;
  gint method, numerics;
  gboolean warn_substituted;      /* gst-exact requested, metal arithmetic configured: warning not posted yet */
  gboolean add_borders;
  guint border_color;
  /* async-depth=1: one frame stays in flight across chain calls (cs_generate_output) */
  GstVfHipPinStats pin;
  GstVfHipAsync async;                          /* async-depth=1: one frame stays in flight across chain calls */
  gboolean reconfigure_pending;
} GstVfHipConvertScale;

typedef struct
{
  GstBaseTransformClass parent_class;
} GstVfHipConvertScaleClass;

enum { PROP_0, PROP_METHOD, PROP_ADD_BORDERS, PROP_BORDER_COLOR, PROP_DEVICE_ID, PROP_NUMERICS, PROP_ASYNC_DEPTH };

static GstStaticPadTemplate cs_sink_template = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_CS_FORMATS)));
static GstStaticPadTemplate cs_src_template = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_CS_FORMATS)));

static GType
scale_method_type (void)
{
  static gsize t = 0;
  static const GEnumValue v[] = {
    {VFHIP_SCALE_BILINEAR, "Bilinear interpolation", "bilinear"},
    {VFHIP_SCALE_NEAREST, "Nearest-neighbor", "nearest"},
    {VFHIP_SCALE_BICUBIC, "Bicubic (Catmull-Rom: GStreamer's videoscale method=catrom, bit-exact; RGB outputs)", "bicubic"},
    {0, NULL, NULL}
  };
  if (g_once_init_enter (&t))
    g_once_init_leave (&t, g_enum_register_static ("GstVfHipScaleMethod", v));
  return (GType) t;
}

GType
gst_vfhip_numerics_get_type (void)
{
  static gsize t = 0;
  static const GEnumValue v[] = {
    {VFHIP_NUMERICS_GST_EXACT, "Integer arithmetic of GStreamer's CPU videoconvert + videoscale (bit-exact)", "gst-exact"},
    {VFHIP_NUMERICS_METAL, "Float arithmetic of the vfmetal shaders", "metal"},
    {VFHIP_NUMERICS_GST_EXACT_STRICT, "gst-exact, and refuse caps whose GStreamer arithmetic is not pinned (plain gst-exact warns and runs the metal arithmetic there)", "gst-exact-strict"},
    {0, NULL, NULL}
  };
  if (g_once_init_enter (&t))
    g_once_init_leave (&t, g_enum_register_static ("GstVfHipNumerics", v));
  return (GType) t;
}

G_DEFINE_TYPE (GstVfHipConvertScale, gst_vfhip_convertscale, GST_TYPE_BASE_TRANSFORM);
#define CS(obj) ((GstVfHipConvertScale *) (obj))

/* any of the six formats at any size, in either memory, converts to any other: drop what we can change, keep the rest
 * (framerate ...) */
static GstCaps *
cs_transform_caps (GstBaseTransform * trans, GstPadDirection direction, GstCaps * caps, GstCaps * filter)
{
  static const gchar *formats[] = { "BGRA", "RGBA", "NV12", "I420", "UYVY", "YUY2" };
  GstCaps *res = gst_caps_new_empty (), *both;
  guint i, k, n = gst_caps_get_size (caps);
  (void) trans; (void) direction;
  for (i = 0; i < n; i++) {
    GstStructure *st = gst_structure_copy (gst_caps_get_structure (caps, i));
    GValue list = G_VALUE_INIT, one = G_VALUE_INIT;
    gst_structure_remove_fields (st, "format", "width", "height", "pixel-aspect-ratio", "colorimetry", "chroma-site", NULL);
    g_value_init (&list, GST_TYPE_LIST);
    g_value_init (&one, G_TYPE_STRING);
    for (k = 0; k < G_N_ELEMENTS (formats); k++) {
      g_value_set_string (&one, formats[k]);
      gst_value_list_append_value (&list, &one);
    }
    gst_structure_set_value (st, "format", &list);
    g_value_unset (&one);
    g_value_unset (&list);
    gst_structure_set (st, "width", GST_TYPE_INT_RANGE, 1, G_MAXINT, "height", GST_TYPE_INT_RANGE, 1, G_MAXINT, NULL);
    gst_caps_append_structure (res, st);
  }
  both = gst_vfhip_caps_both_memories (res);
  gst_caps_unref (res);
  res = both;
  if (filter) {
    GstCaps *tmp = gst_caps_intersect_full (res, filter, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (res);
    res = tmp;
  }
  return res;
}

static void
par_of (const GstStructure * st, gint * n, gint * d)
{
  const GValue *v = gst_structure_get_value (st, "pixel-aspect-ratio");
  *n = *d = 1;
  if (v && GST_VALUE_HOLDS_FRACTION (v)) {
    *n = gst_value_get_fraction_numerator (v);
    *d = gst_value_get_fraction_denominator (v);
  }
}

/* keep the format when possible; a missing output dimension follows the input display aspect ratio */
static GstCaps *
cs_fixate_caps (GstBaseTransform * trans, GstPadDirection direction, GstCaps * caps, GstCaps * othercaps)
{
  GstStructure *in, *out;
  const gchar *fmt;
  gint iw = 0, ih = 0, ipn, ipd, opn, opd, darn, dard, w = 0, h = 0;
  gboolean have_w, have_h;
  (void) direction;
  othercaps = gst_caps_make_writable (gst_caps_truncate (othercaps));
  in = gst_caps_get_structure (caps, 0);
  out = gst_caps_get_structure (othercaps, 0);
  if ((fmt = gst_structure_get_string (in, "format")))
    gst_structure_fixate_field_string (out, "format", fmt);
  gst_structure_get_int (in, "width", &iw);
  gst_structure_get_int (in, "height", &ih);
  par_of (in, &ipn, &ipd);
  par_of (out, &opn, &opd);
  if (!gst_util_fraction_multiply (iw, ih, ipn, ipd, &darn, &dard)) { darn = iw; dard = ih; }
  have_w = gst_structure_get_int (out, "width", &w);
  have_h = gst_structure_get_int (out, "height", &h);
  if (!have_w && !have_h) {
    gst_structure_fixate_field_nearest_int (out, "width", iw);
    gst_structure_get_int (out, "width", &w);
    have_w = TRUE;
  }
  if (have_w && !have_h) {
    h = (gint) gst_util_uint64_scale_int (w, dard * opn, darn * opd);
    gst_structure_fixate_field_nearest_int (out, "height", MAX (h, 1));
  } else if (!have_w && have_h) {
    w = (gint) gst_util_uint64_scale_int (h, darn * opd, dard * opn);
    gst_structure_fixate_field_nearest_int (out, "width", MAX (w, 1));
  }
  GST_DEBUG_OBJECT (trans, "fixated to %" GST_PTR_FORMAT, othercaps);
  return gst_caps_fixate (othercaps);
}

/* with the object lock held */
static gboolean
cs_configure_locked (GstVfHipConvertScale * self)
{
  VfHipVideoInfo in, out;
  gst_vfhip_info (&self->in_info, &in);
  gst_vfhip_info (&self->out_info, &out);
  if (vfhip_convertscale_configure (self->renderer, &in, &out, self->method, self->add_borders, self->border_color, self->numerics) != VFHIP_OK) {
    GST_ERROR_OBJECT (self, "configure failed: %s", vfhip_last_error_string ());
    return FALSE;
  }
  /* gst-exact on a cell without pinned GStreamer arithmetic runs the shader arithmetic: say so (cs_post_substitution) */
  self->warn_substituted = self->numerics == VFHIP_NUMERICS_GST_EXACT &&
      vfhip_convertscale_numerics_in_effect (self->renderer) == VFHIP_NUMERICS_METAL;
  return TRUE;
}

/* outside the object lock (posts a bus message) */
static void
cs_post_substitution (GstVfHipConvertScale * self)
{
  if (!self->warn_substituted)
    return;
  self->warn_substituted = FALSE;
  GST_ELEMENT_WARNING (self, STREAM, NOT_IMPLEMENTED,
      ("numerics=gst-exact: GStreamer's arithmetic for this format / colorimetry / border combination is not pinned; running the vfmetal shader arithmetic (kernel %s)",
          vfhip_convertscale_kernel_name (self->renderer)),
      ("set numerics=gst-exact-strict to refuse such caps instead, or numerics=metal to ask for the shader arithmetic"));
}

static gboolean
cs_ensure_renderer (GstVfHipConvertScale * self)
{
  if (!self->renderer) {
    self->renderer = vfhip_convertscale_new (self->device_id);
    if (!self->renderer)
      GST_ERROR_OBJECT (self, "no HIP renderer: %s", vfhip_last_error_string ());
  }
  return self->renderer != NULL;
}

/* Output caps that do not say which matrix / chroma siting they mean: take what `videoconvert ! videoscale` would.
 * There videoconvert runs at the INPUT size, so GStreamer's by-height defaults of its output are those of the input
 * height (not of the scaled frame), and it copies colorimetry / chroma-site from YUV input caps that carry them when
 * the output is YUV too (chroma-site only when the subsampling is unchanged).  Without this a 720p -> 360p NV12 -> UYVY
 * conversion would re-matrix bt709 -> bt601, which the CPU pipeline does not do. */
static void
cs_effective_out_info (GstCaps * incaps, GstCaps * outcaps, const GstVideoInfo * in, GstVideoInfo * out)
{
  const GstStructure *si = gst_caps_get_structure (incaps, 0), *so = gst_caps_get_structure (outcaps, 0);
  const gboolean both_yuv = GST_VIDEO_INFO_IS_YUV (in) && GST_VIDEO_INFO_IS_YUV (out);
  GstVideoInfo at_in_size;
  gst_video_info_init (&at_in_size);
  gst_video_info_set_format (&at_in_size, GST_VIDEO_INFO_FORMAT (out), GST_VIDEO_INFO_WIDTH (in), GST_VIDEO_INFO_HEIGHT (in));
  if (!gst_structure_has_field (so, "colorimetry"))
    out->colorimetry = (both_yuv && gst_structure_has_field (si, "colorimetry")) ? in->colorimetry : at_in_size.colorimetry;
  if (!gst_structure_has_field (so, "chroma-site") && GST_VIDEO_INFO_IS_YUV (out)) {
    const gboolean same_sub = both_yuv && in->finfo->w_sub[1] == out->finfo->w_sub[1] && in->finfo->h_sub[1] == out->finfo->h_sub[1];
    out->chroma_site = (same_sub && gst_structure_has_field (si, "chroma-site")) ? in->chroma_site : at_in_size.chroma_site;
  }
}

static gboolean
cs_set_caps (GstBaseTransform * trans, GstCaps * incaps, GstCaps * outcaps)
{
  GstVfHipConvertScale *self = CS (trans);
  gboolean same, ok = TRUE;
  if (!gst_video_info_from_caps (&self->in_info, incaps) || !gst_video_info_from_caps (&self->out_info, outcaps)) {
    GST_ERROR_OBJECT (self, "unparsable caps");
    return FALSE;
  }
  cs_effective_out_info (incaps, outcaps, &self->in_info, &self->out_info);
  same = GST_VIDEO_INFO_FORMAT (&self->in_info) == GST_VIDEO_INFO_FORMAT (&self->out_info) &&
      GST_VIDEO_INFO_WIDTH (&self->in_info) == GST_VIDEO_INFO_WIDTH (&self->out_info) &&
      GST_VIDEO_INFO_HEIGHT (&self->in_info) == GST_VIDEO_INFO_HEIGHT (&self->out_info);
  gst_base_transform_set_passthrough (trans, same);
  GST_DEBUG_OBJECT (self, "%" GST_PTR_FORMAT " -> %" GST_PTR_FORMAT "%s", incaps, outcaps, same ? " (passthrough)" : "");
  GST_OBJECT_LOCK (self);
  self->negotiated = TRUE;
  self->passthrough = same;
  if (!same)
    ok = cs_ensure_renderer (self) && cs_configure_locked (self);
  GST_OBJECT_UNLOCK (self);
  cs_post_substitution (self);
  return ok;
}

static gboolean
cs_get_unit_size (GstBaseTransform * trans, GstCaps * caps, gsize * size)
{
  GstVideoInfo info;
  (void) trans;
  if (!gst_video_info_from_caps (&info, caps))
    return FALSE;
  *size = GST_VIDEO_INFO_SIZE (&info);
  return TRUE;
}

static GstFlowReturn
cs_transform (GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstVfHipConvertScale *self = CS (trans);
  GstVideoFrame in, out;
  VfHipFrame vin, vout;
  int rc;
  gint dev;
  if (!self->negotiated)
    return GST_FLOW_NOT_NEGOTIATED;
  if (!self->renderer) {
    GST_WARNING_OBJECT (self, "no HIP renderer");
    return GST_FLOW_ERROR;
  }
  cs_post_substitution (self);                              /* after a property change reconfigured the renderer */
  gst_vfhip_pin_foreign_memory (inbuf, &self->pin);         /* recurring pageable upstream memory: page-lock it in place */
  dev = gst_vfhip_element_device (self);
  if (!gst_video_frame_map (&in, &self->in_info, inbuf, (GstMapFlags) (GST_MAP_READ | gst_vfhip_map_flag (inbuf, dev))))
    return GST_FLOW_ERROR;
  if (!gst_video_frame_map (&out, &self->out_info, outbuf, (GstMapFlags) (GST_MAP_WRITE | gst_vfhip_map_flag (outbuf, dev)))) {
    gst_video_frame_unmap (&in);
    return GST_FLOW_ERROR;
  }
  gst_vfhip_frame (&in, &vin);
  gst_vfhip_frame (&out, &vout);
  rc = vfhip_convertscale_process (self->renderer, &vin, &vout);
  gst_video_frame_unmap (&out);
  gst_video_frame_unmap (&in);
  if (rc != VFHIP_OK) {
    GST_WARNING_OBJECT (self, "HIP processing failed: %s", vfhip_last_error_string ());
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

/* ---- async-depth=1 (gstvfhipasync.c) -------------------------------------------------------------------------------- */
static int
cs_async_submit (GstBaseTransform * trans, const VfHipFrame * in, VfHipFrame * out)
{
  return vfhip_convertscale_submit (CS (trans)->renderer, in, out);
}

static int
cs_async_wait (GstBaseTransform * trans)
{
  return vfhip_convertscale_wait (CS (trans)->renderer);
}

static GstFlowReturn
cs_generate_output (GstBaseTransform * trans, GstBuffer ** outbuf)
{
  GstVfHipConvertScale *self = CS (trans);
  gboolean reconf;
  GST_OBJECT_LOCK (self);
  reconf = self->reconfigure_pending;
  self->reconfigure_pending = FALSE;
  GST_OBJECT_UNLOCK (self);
  if (reconf) {                                               /* a property changed: the renderer reconfigures on an empty pipeline */
    GstFlowReturn ret = gst_vfhip_async_drain (trans, &self->async, TRUE);
    if (ret != GST_FLOW_OK)
      return ret;
    GST_OBJECT_LOCK (self);
    if (self->negotiated && self->renderer && !self->passthrough)
      cs_configure_locked (self);
    GST_OBJECT_UNLOCK (self);
  }
  return gst_vfhip_async_generate_output (trans, outbuf, &self->async, &self->in_info, &self->out_info, self->negotiated && self->renderer != NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_convertscale_parent_class)->generate_output);
}

static gboolean
cs_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  return gst_vfhip_async_sink_event (trans, event, &CS (trans)->async, GST_BASE_TRANSFORM_CLASS (gst_vfhip_convertscale_parent_class)->sink_event);
}

static gboolean
cs_query (GstBaseTransform * trans, GstPadDirection direction, GstQuery * query)
{
  GstVfHipConvertScale *self = CS (trans);
  return gst_vfhip_async_query (trans, direction, query, &self->async, self->negotiated ? &self->out_info : NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_convertscale_parent_class)->query);
}

static void
cs_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstVfHipConvertScale *self = CS (object);
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_METHOD: self->method = g_value_get_enum (value); break;
    case PROP_ADD_BORDERS: self->add_borders = g_value_get_boolean (value); break;
    case PROP_BORDER_COLOR: self->border_color = g_value_get_uint (value); break;
    case PROP_NUMERICS: self->numerics = g_value_get_enum (value); break;
    case PROP_DEVICE_ID: self->device_id = g_value_get_int (value); break;      /* takes effect for the next renderer */
    case PROP_ASYNC_DEPTH: self->async.depth = g_value_get_int (value); break;
    default:
      GST_OBJECT_UNLOCK (self);
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec);
      return;
  }
  if (id != PROP_DEVICE_ID && id != PROP_ASYNC_DEPTH && self->negotiated && self->renderer && !self->passthrough) {
    if (self->async.depth > 0)
      self->reconfigure_pending = TRUE;         /* frames may be in flight: the streaming thread reconfigures (cs_generate_output) */
    else
      cs_configure_locked (self);
  }
  GST_OBJECT_UNLOCK (self);
}

static void
cs_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstVfHipConvertScale *self = CS (object);
  switch (id) {
    case PROP_METHOD: g_value_set_enum (value, self->method); break;
    case PROP_ADD_BORDERS: g_value_set_boolean (value, self->add_borders); break;
    case PROP_BORDER_COLOR: g_value_set_uint (value, self->border_color); break;
    case PROP_NUMERICS: g_value_set_enum (value, self->numerics); break;
    case PROP_DEVICE_ID: g_value_set_int (value, self->device_id); break;
    case PROP_ASYNC_DEPTH: g_value_set_int (value, self->async.depth); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
}

static GstStateChangeReturn
cs_change_state (GstElement * element, GstStateChange transition)
{
  GstVfHipConvertScale *self = CS (element);
  GstStateChangeReturn ret = GST_ELEMENT_CLASS (gst_vfhip_convertscale_parent_class)->change_state (element, transition);
  if (transition == GST_STATE_CHANGE_PAUSED_TO_READY) {     /* drop GPU resources, keep the handle */
    gst_vfhip_async_drain (GST_BASE_TRANSFORM (self), &self->async, FALSE);      /* the streaming thread has stopped: frames still in flight are dropped */
    GST_OBJECT_LOCK (self);
    if (self->renderer)
      vfhip_convertscale_cleanup (self->renderer);
    self->negotiated = FALSE;
    GST_OBJECT_UNLOCK (self);
  }
  return ret;
}

static void
cs_finalize (GObject * object)
{
  GstVfHipConvertScale *self = CS (object);
  if (self->renderer)
    vfhip_convertscale_free (self->renderer);
  self->renderer = NULL;
  G_OBJECT_CLASS (gst_vfhip_convertscale_parent_class)->finalize (object);
}


static gboolean
co_propose_allocation (GstBaseTransform * trans, GstQuery * decide_query, GstQuery * query)
{
  return gst_vfhip_propose_allocation (trans, decide_query, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_convertscale_parent_class)->propose_allocation);
}

static gboolean
co_decide_allocation (GstBaseTransform * trans, GstQuery * query)
{
  return gst_vfhip_decide_allocation (trans, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_convertscale_parent_class)->decide_allocation);
}

static void
gst_vfhip_convertscale_class_init (GstVfHipConvertScaleClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *bc = GST_BASE_TRANSFORM_CLASS (klass);
  oc->set_property = cs_set_property;
  oc->get_property = cs_get_property;
  oc->finalize = cs_finalize;
  bc->propose_allocation = GST_DEBUG_FUNCPTR (co_propose_allocation);
  bc->decide_allocation = GST_DEBUG_FUNCPTR (co_decide_allocation);
  ec->change_state = GST_DEBUG_FUNCPTR (cs_change_state);
  bc->transform_caps = GST_DEBUG_FUNCPTR (cs_transform_caps);
  bc->fixate_caps = GST_DEBUG_FUNCPTR (cs_fixate_caps);
  bc->set_caps = GST_DEBUG_FUNCPTR (cs_set_caps);
  bc->get_unit_size = GST_DEBUG_FUNCPTR (cs_get_unit_size);
  bc->transform = GST_DEBUG_FUNCPTR (cs_transform);
  bc->generate_output = GST_DEBUG_FUNCPTR (cs_generate_output);
  bc->sink_event = GST_DEBUG_FUNCPTR (cs_sink_event);
  bc->query = GST_DEBUG_FUNCPTR (cs_query);
  bc->passthrough_on_same_caps = FALSE;        /* decided in set_caps */

  g_object_class_install_property (oc, PROP_METHOD, g_param_spec_enum ("method", "Method", "Scaling interpolation method",
          scale_method_type (), VFHIP_SCALE_BILINEAR, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_ADD_BORDERS, g_param_spec_boolean ("add-borders", "Add Borders",
          "Add letterbox/pillarbox borders to preserve aspect ratio", FALSE, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_BORDER_COLOR, g_param_spec_uint ("border-color", "Border Color",
          "Border color in ARGB format (default: opaque black 0xFF000000)", 0, G_MAXUINT32, 0xFF000000u,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "GPU ordinal to run on (-1: $VFHIP_DEVICE, else 0); independent streams shard across GPUs by this", -1, 63,
          GST_VFHIP_DEFAULT_DEVICE_ID, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_NUMERICS, g_param_spec_enum ("numerics", "Numerics",
          "Arithmetic family: bit-exact GStreamer CPU videoconvert+videoscale, or the vfmetal shaders' float maths",
          gst_vfhip_numerics_get_type (), VFHIP_NUMERICS_GST_EXACT, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  g_object_class_install_property (oc, PROP_ASYNC_DEPTH, gst_vfhip_async_depth_pspec ());

  gst_element_class_add_static_pad_template (ec, &cs_sink_template);
  gst_element_class_add_static_pad_template (ec, &cs_src_template);
  gst_element_class_set_static_metadata (ec, "HIP Video Convert and Scale", "Filter/Converter/Video/Scaler",
      "MI355X-accelerated video format conversion and scaling", "vfhip");
  GST_DEBUG_CATEGORY_INIT (gst_vfhip_convertscale_debug, "vfhipconvertscale", 0, "vfhipconvertscale element");
}

static void
gst_vfhip_convertscale_init (GstVfHipConvertScale * self)
{
  self->method = VFHIP_SCALE_BILINEAR;
  self->add_borders = FALSE;
  self->border_color = 0xFF000000u;
  self->numerics = VFHIP_NUMERICS_GST_EXACT;
  self->device_id = GST_VFHIP_DEFAULT_DEVICE_ID;
  self->async.depth = 0;
  self->async.submit = cs_async_submit;
  self->async.wait = cs_async_wait;
  self->negotiated = FALSE;
  self->renderer = NULL;                 /* created on first negotiation so that device-id is honoured */
}

gboolean
gst_vfhip_convertscale_register (GstPlugin * plugin)
{
  gboolean ok = gst_element_register (plugin, "vfhipconvertscale", GST_RANK_NONE, gst_vfhip_convertscale_get_type ());
#ifdef VFHIP_REGISTER_VFMETAL_NAMES
  ok &= gst_element_register (plugin, "vfmetalconvertscale", GST_RANK_NONE, gst_vfhip_convertscale_get_type ());
#endif
  return ok;
}
