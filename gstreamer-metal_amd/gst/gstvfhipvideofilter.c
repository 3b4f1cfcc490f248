/* gst/gstvfhipvideofilter.c — `vfhipvideofilter`: the 15-property video filter on an MI355X.
 *
 * Drop-in for the reference's vfmetalvideofilter (videofilter/gstvfmetalvideofilter.{h,m}): GstVideoFilter subclass,
 * templates { BGRA, RGBA, NV12, I420 } (:53-65), the 15 properties with identical names, ranges and defaults
 * (:435-533), passthrough while every property sits at its default (:116-138), parameter snapshot under the object
 * lock with hue * pi, ARGB key colour -> float rgb and a per-frame counter for the noise hash (:184-205).
 * lut-file accepts .cube only (PNG needs an image decoder: SURVEY.md §2 #7). */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include <math.h>
#include <gst/video/gstvideofilter.h>
#include "gstvfhip.h"

/* the element's own debug category, like the reference's (videofilter/gstvfmetalvideofilter.m:546-547); shared helpers log to `vfhip` */
GST_DEBUG_CATEGORY_STATIC (gst_vfhip_videofilter_debug);
#define GST_CAT_DEFAULT gst_vfhip_videofilter_debug
#define VFHIP_VF_FORMATS "{ BGRA, RGBA, NV12, I420 }"

typedef struct
{
  GstVideoFilter parent;
  VfHipVideoFilter *renderer;
  gint device_id;
  gdouble brightness, contrast, saturation, hue, gamma, sharpness, sepia, noise, vignette;
  gboolean invert, chroma_key_enabled;
  guint chroma_key_color;
  gdouble chroma_key_tolerance, chroma_key_smoothness;
  gchar *lut_file;
  guint frame_count;
  GstVfHipAsync async;                          /* async-depth=1 (gstvfhipasync.c) */
} GstVfHipVideoFilter;

typedef struct
{
  GstVideoFilterClass parent_class;
} GstVfHipVideoFilterClass;

enum
{
  PROP_0, PROP_BRIGHTNESS, PROP_CONTRAST, PROP_SATURATION, PROP_HUE, PROP_GAMMA, PROP_SHARPNESS, PROP_SEPIA, PROP_INVERT,
  PROP_NOISE, PROP_VIGNETTE, PROP_CHROMA_KEY_ENABLED, PROP_CHROMA_KEY_COLOR, PROP_CHROMA_KEY_TOLERANCE,
  PROP_CHROMA_KEY_SMOOTHNESS, PROP_LUT_FILE, PROP_DEVICE_ID, PROP_ASYNC_DEPTH
};

static GstStaticPadTemplate vf_sink_template = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_VF_FORMATS)));
static GstStaticPadTemplate vf_src_template = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_VF_FORMATS)));

G_DEFINE_TYPE (GstVfHipVideoFilter, gst_vfhip_videofilter, GST_TYPE_VIDEO_FILTER);
#define VF(obj) ((GstVfHipVideoFilter *) (obj))
#define NEAR(a, b) (fabs ((a) - (b)) < 1e-6)

static gboolean
vf_ensure_renderer (GstVfHipVideoFilter * self)
{
  if (!self->renderer) {
    self->renderer = vfhip_videofilter_new (self->device_id);
    if (!self->renderer) {
      GST_ERROR_OBJECT (self, "no HIP renderer: %s", vfhip_last_error_string ());
      return FALSE;
    }
    if (self->lut_file && self->lut_file[0] && vfhip_videofilter_load_lut (self->renderer, self->lut_file) != VFHIP_OK)
      GST_WARNING_OBJECT (self, "failed to load LUT %s: %s", self->lut_file, vfhip_last_error_string ());
  }
  return TRUE;
}

static void
vf_update_passthrough (GstVfHipVideoFilter * self)
{
  gboolean idle;
  GST_OBJECT_LOCK (self);
  idle = NEAR (self->brightness, 0.0) && NEAR (self->contrast, 1.0) && NEAR (self->saturation, 1.0) && NEAR (self->hue, 0.0) &&
      NEAR (self->gamma, 1.0) && NEAR (self->sharpness, 0.0) && NEAR (self->sepia, 0.0) && !self->invert && NEAR (self->noise, 0.0) &&
      NEAR (self->vignette, 0.0) && !self->chroma_key_enabled && (!self->lut_file || !self->lut_file[0]);
  GST_OBJECT_UNLOCK (self);
  gst_base_transform_set_passthrough (GST_BASE_TRANSFORM (self), idle);
}

static gboolean
vf_set_info (GstVideoFilter * filter, GstCaps * incaps, GstVideoInfo * in_info, GstCaps * outcaps, GstVideoInfo * out_info)
{
  GstVfHipVideoFilter *self = VF (filter);
  VfHipVideoInfo in, out;
  (void) incaps; (void) outcaps;
  GST_DEBUG_OBJECT (filter, "caps %" GST_PTR_FORMAT " -> %" GST_PTR_FORMAT, incaps, outcaps);
  if (!vf_ensure_renderer (self))
    return FALSE;
  gst_vfhip_info (in_info, &in);
  gst_vfhip_info (out_info, &out);
  if (vfhip_videofilter_configure (self->renderer, &in, &out) != VFHIP_OK) {
    GST_ERROR_OBJECT (self, "configure failed: %s", vfhip_last_error_string ());
    return FALSE;
  }
  return TRUE;
}

/* one consistent snapshot of the properties per frame */
static void
vf_params (GstVfHipVideoFilter * self, VfHipVideoFilterParams * p)
{
  guint key;
  memset (p, 0, sizeof (*p));
  GST_OBJECT_LOCK (self);
  p->brightness = (float) self->brightness;
  p->contrast = (float) self->contrast;
  p->saturation = (float) self->saturation;
  p->hue = (float) (self->hue * G_PI);          /* property is -1..1 half-turns */
  p->gamma = (float) self->gamma;
  p->sharpness = (float) self->sharpness;
  p->sepia = (float) self->sepia;
  p->noise = (float) self->noise;
  p->vignette = (float) self->vignette;
  p->invert = self->invert;
  p->chroma_key_enabled = self->chroma_key_enabled;
  key = self->chroma_key_color;
  p->chroma_key_tolerance = (float) self->chroma_key_tolerance;
  p->chroma_key_smoothness = (float) self->chroma_key_smoothness;
  p->frame_index = self->frame_count++;
  GST_OBJECT_UNLOCK (self);
  p->chroma_key_r = ((key >> 16) & 0xff) / 255.0f;
  p->chroma_key_g = ((key >> 8) & 0xff) / 255.0f;
  p->chroma_key_b = (key & 0xff) / 255.0f;
}

static GstFlowReturn
vf_transform_frame (GstVideoFilter * filter, GstVideoFrame * in, GstVideoFrame * out)
{
  GstVfHipVideoFilter *self = VF (filter);
  VfHipVideoFilterParams p;
  VfHipFrame vin, vout;
  if (!self->renderer) {
    GST_WARNING_OBJECT (self, "no HIP renderer");
    return GST_FLOW_ERROR;
  }
  vf_params (self, &p);
  gst_vfhip_frame (in, &vin);
  gst_vfhip_frame (out, &vout);
  if (vfhip_videofilter_process (self->renderer, &vin, &vout, &p) != VFHIP_OK) {
    GST_WARNING_OBJECT (self, "HIP processing failed: %s", vfhip_last_error_string ());
    return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

static void
vf_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstVfHipVideoFilter *self = VF (object);
  gchar *lut = NULL;
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_BRIGHTNESS: self->brightness = g_value_get_double (value); break;
    case PROP_CONTRAST: self->contrast = g_value_get_double (value); break;
    case PROP_SATURATION: self->saturation = g_value_get_double (value); break;
    case PROP_HUE: self->hue = g_value_get_double (value); break;
    case PROP_GAMMA: self->gamma = g_value_get_double (value); break;
    case PROP_SHARPNESS: self->sharpness = g_value_get_double (value); break;
    case PROP_SEPIA: self->sepia = g_value_get_double (value); break;
    case PROP_INVERT: self->invert = g_value_get_boolean (value); break;
    case PROP_NOISE: self->noise = g_value_get_double (value); break;
    case PROP_VIGNETTE: self->vignette = g_value_get_double (value); break;
    case PROP_CHROMA_KEY_ENABLED: self->chroma_key_enabled = g_value_get_boolean (value); break;
    case PROP_CHROMA_KEY_COLOR: self->chroma_key_color = g_value_get_uint (value); break;
    case PROP_CHROMA_KEY_TOLERANCE: self->chroma_key_tolerance = g_value_get_double (value); break;
    case PROP_CHROMA_KEY_SMOOTHNESS: self->chroma_key_smoothness = g_value_get_double (value); break;
    case PROP_DEVICE_ID: self->device_id = g_value_get_int (value); break;
    case PROP_ASYNC_DEPTH: self->async.depth = g_value_get_int (value); break;
    case PROP_LUT_FILE:
      g_free (self->lut_file);
      self->lut_file = g_value_dup_string (value);
      lut = g_strdup (self->lut_file);
      break;
    default:
      GST_OBJECT_UNLOCK (self);
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec);
      return;
  }
  GST_OBJECT_UNLOCK (self);
  if (id == PROP_LUT_FILE && self->renderer) {      /* file I/O and the upload happen outside the lock */
    if (lut && lut[0]) {
      if (vfhip_videofilter_load_lut (self->renderer, lut) != VFHIP_OK)
        GST_WARNING_OBJECT (self, "failed to load LUT %s: %s", lut, vfhip_last_error_string ());
    } else {
      vfhip_videofilter_clear_lut (self->renderer);
    }
  }
  g_free (lut);
  vf_update_passthrough (self);
}

static void
vf_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstVfHipVideoFilter *self = VF (object);
  GST_OBJECT_LOCK (self);
  switch (id) {
    case PROP_BRIGHTNESS: g_value_set_double (value, self->brightness); break;
    case PROP_CONTRAST: g_value_set_double (value, self->contrast); break;
    case PROP_SATURATION: g_value_set_double (value, self->saturation); break;
    case PROP_HUE: g_value_set_double (value, self->hue); break;
    case PROP_GAMMA: g_value_set_double (value, self->gamma); break;
    case PROP_SHARPNESS: g_value_set_double (value, self->sharpness); break;
    case PROP_SEPIA: g_value_set_double (value, self->sepia); break;
    case PROP_INVERT: g_value_set_boolean (value, self->invert); break;
    case PROP_NOISE: g_value_set_double (value, self->noise); break;
    case PROP_VIGNETTE: g_value_set_double (value, self->vignette); break;
    case PROP_CHROMA_KEY_ENABLED: g_value_set_boolean (value, self->chroma_key_enabled); break;
    case PROP_CHROMA_KEY_COLOR: g_value_set_uint (value, self->chroma_key_color); break;
    case PROP_CHROMA_KEY_TOLERANCE: g_value_set_double (value, self->chroma_key_tolerance); break;
    case PROP_CHROMA_KEY_SMOOTHNESS: g_value_set_double (value, self->chroma_key_smoothness); break;
    case PROP_LUT_FILE: g_value_set_string (value, self->lut_file); break;
    case PROP_DEVICE_ID: g_value_set_int (value, self->device_id); break;
    case PROP_ASYNC_DEPTH: g_value_set_int (value, self->async.depth); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (self);
}

static gboolean
vf_start (GstBaseTransform * trans)
{
  VF (trans)->frame_count = 0;
  vf_update_passthrough (VF (trans));
  return TRUE;
}

static gboolean
vf_stop (GstBaseTransform * trans)
{
  gst_vfhip_async_drain (trans, &VF (trans)->async, FALSE);          /* the streaming thread has stopped: frames in flight are dropped */
  GstVfHipVideoFilter *self = VF (trans);
  if (self->renderer)
    vfhip_videofilter_cleanup (self->renderer);
  self->frame_count = 0;
  return TRUE;
}

static void
vf_finalize (GObject * object)
{
  GstVfHipVideoFilter *self = VF (object);
  if (self->renderer)
    vfhip_videofilter_free (self->renderer);
  self->renderer = NULL;
  g_free (self->lut_file);
  G_OBJECT_CLASS (gst_vfhip_videofilter_parent_class)->finalize (object);
}

#define DPROP(id, name, nick, blurb, lo, hi, def) \
  g_object_class_install_property (oc, id, g_param_spec_double (name, nick, blurb, lo, hi, def, \
          G_PARAM_READWRITE | GST_PARAM_CONTROLLABLE | G_PARAM_STATIC_STRINGS))


static gboolean
vi_propose_allocation (GstBaseTransform * trans, GstQuery * decide_query, GstQuery * query)
{
  return gst_vfhip_propose_allocation (trans, decide_query, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_videofilter_parent_class)->propose_allocation);
}

static gboolean
vi_decide_allocation (GstBaseTransform * trans, GstQuery * query)
{
  return gst_vfhip_decide_allocation (trans, query, GST_BASE_TRANSFORM_CLASS (gst_vfhip_videofilter_parent_class)->decide_allocation);
}


/* ---- async-depth=1 (gstvfhipasync.c) -------------------------------------------------------------------------------- */
static int
vf_async_submit (GstBaseTransform * trans, const VfHipFrame * in, VfHipFrame * out)
{
  VfHipVideoFilterParams p;
  vf_params (VF (trans), &p);
  return vfhip_videofilter_submit (VF (trans)->renderer, in, out, &p);
}

static int
vf_async_wait (GstBaseTransform * trans)
{
  return vfhip_videofilter_wait (VF (trans)->renderer);
}

static GstFlowReturn
vf_generate_output (GstBaseTransform * trans, GstBuffer ** outbuf)
{
  GstVideoFilter *f = GST_VIDEO_FILTER_CAST (trans);
  return gst_vfhip_async_generate_output (trans, outbuf, &VF (trans)->async, &f->in_info, &f->out_info, f->negotiated && VF (trans)->renderer != NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_videofilter_parent_class)->generate_output);
}

static gboolean
vf_sink_event (GstBaseTransform * trans, GstEvent * event)
{
  return gst_vfhip_async_sink_event (trans, event, &VF (trans)->async, GST_BASE_TRANSFORM_CLASS (gst_vfhip_videofilter_parent_class)->sink_event);
}

static gboolean
vf_query (GstBaseTransform * trans, GstPadDirection direction, GstQuery * query)
{
  GstVideoFilter *f = GST_VIDEO_FILTER_CAST (trans);
  return gst_vfhip_async_query (trans, direction, query, &VF (trans)->async, f->negotiated ? &f->out_info : NULL,
      GST_BASE_TRANSFORM_CLASS (gst_vfhip_videofilter_parent_class)->query);
}

static void
gst_vfhip_videofilter_class_init (GstVfHipVideoFilterClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstBaseTransformClass *bc = GST_BASE_TRANSFORM_CLASS (klass);
  GstVideoFilterClass *fc = GST_VIDEO_FILTER_CLASS (klass);
  oc->set_property = vf_set_property;
  oc->get_property = vf_get_property;
  oc->finalize = vf_finalize;
  bc->propose_allocation = GST_DEBUG_FUNCPTR (vi_propose_allocation);
  bc->decide_allocation = GST_DEBUG_FUNCPTR (vi_decide_allocation);
  bc->start = GST_DEBUG_FUNCPTR (vf_start);
  bc->stop = GST_DEBUG_FUNCPTR (vf_stop);
  fc->set_info = GST_DEBUG_FUNCPTR (vf_set_info);
  fc->transform_frame = GST_DEBUG_FUNCPTR (vf_transform_frame);
  /* memory:HIPMemory on either pad (gstvfhipmemory.c): same video caps in both memories, device buffers mapped in place */
  GST_BASE_TRANSFORM_CLASS (klass)->transform_caps = GST_DEBUG_FUNCPTR (gst_vfhip_filter_transform_caps);
  GST_BASE_TRANSFORM_CLASS (klass)->transform = GST_DEBUG_FUNCPTR (gst_vfhip_filter_transform);
  GST_BASE_TRANSFORM_CLASS (klass)->generate_output = GST_DEBUG_FUNCPTR (vf_generate_output);
  GST_BASE_TRANSFORM_CLASS (klass)->sink_event = GST_DEBUG_FUNCPTR (vf_sink_event);
  GST_BASE_TRANSFORM_CLASS (klass)->query = GST_DEBUG_FUNCPTR (vf_query);

  DPROP (PROP_BRIGHTNESS, "brightness", "Brightness", "Brightness adjustment (-1.0 to 1.0)", -1.0, 1.0, 0.0);
  DPROP (PROP_CONTRAST, "contrast", "Contrast", "Contrast adjustment (0.0 to 2.0, 1.0 = normal)", 0.0, 2.0, 1.0);
  DPROP (PROP_SATURATION, "saturation", "Saturation", "Color saturation (0.0 = grayscale, 1.0 = normal, 2.0 = oversaturated)", 0.0, 2.0, 1.0);
  DPROP (PROP_HUE, "hue", "Hue", "Hue rotation (-1.0 to 1.0, mapped to -180 to +180 degrees)", -1.0, 1.0, 0.0);
  DPROP (PROP_GAMMA, "gamma", "Gamma", "Gamma correction (0.01 to 10.0, 1.0 = normal)", 0.01, 10.0, 1.0);
  DPROP (PROP_SHARPNESS, "sharpness", "Sharpness", "Sharpness adjustment (-1.0 = maximum blur, 0.0 = none, 1.0 = maximum sharpen)", -1.0, 1.0, 0.0);
  DPROP (PROP_SEPIA, "sepia", "Sepia", "Sepia tone mix amount (0.0 = none, 1.0 = full sepia)", 0.0, 1.0, 0.0);
  g_object_class_install_property (oc, PROP_INVERT, g_param_spec_boolean ("invert", "Invert", "Invert all colors (negative image)", FALSE,
          G_PARAM_READWRITE | GST_PARAM_CONTROLLABLE | G_PARAM_STATIC_STRINGS));
  DPROP (PROP_NOISE, "noise", "Noise", "Film grain / noise amount (0.0 = none, 1.0 = maximum)", 0.0, 1.0, 0.0);
  DPROP (PROP_VIGNETTE, "vignette", "Vignette", "Vignette darkness (0.0 = none, 1.0 = maximum darkening at edges)", 0.0, 1.0, 0.0);
  g_object_class_install_property (oc, PROP_CHROMA_KEY_ENABLED, g_param_spec_boolean ("chroma-key-enabled", "Chroma Key Enabled",
          "Enable chroma key (green screen) removal", FALSE, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_CHROMA_KEY_COLOR, g_param_spec_uint ("chroma-key-color", "Chroma Key Color",
          "Chroma key color in ARGB format (default: green 0xFF00FF00)", 0, G_MAXUINT32, 0xFF00FF00u, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  DPROP (PROP_CHROMA_KEY_TOLERANCE, "chroma-key-tolerance", "Chroma Key Tolerance", "Color distance threshold for chroma key (0.0 to 1.0)", 0.0, 1.0, 0.2);
  DPROP (PROP_CHROMA_KEY_SMOOTHNESS, "chroma-key-smoothness", "Chroma Key Smoothness", "Edge softness for chroma key transition (0.0 to 1.0)", 0.0, 1.0, 0.1);
  g_object_class_install_property (oc, PROP_LUT_FILE, g_param_spec_string ("lut-file", "LUT File",
          "Path to a .cube 3D LUT file for color grading (.png LUTs are not supported by vfhip)", NULL, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "GPU ordinal to run on (-1: $VFHIP_DEVICE, else 0)", -1, 63, GST_VFHIP_DEFAULT_DEVICE_ID, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));

  g_object_class_install_property (oc, PROP_ASYNC_DEPTH, gst_vfhip_async_depth_pspec ());
  gst_element_class_add_static_pad_template (ec, &vf_sink_template);
  gst_element_class_add_static_pad_template (ec, &vf_src_template);
  gst_element_class_set_static_metadata (ec, "HIP Video Filter", "Filter/Effect/Video",
      "MI355X-accelerated single-pass colour adjustments, sharpen/blur, chroma key, vignette, grain and 3D LUT", "vfhip");
  GST_DEBUG_CATEGORY_INIT (gst_vfhip_videofilter_debug, "vfhipvideofilter", 0, "vfhipvideofilter element");
}

static void
gst_vfhip_videofilter_init (GstVfHipVideoFilter * self)
{
  self->contrast = self->saturation = self->gamma = 1.0;
  self->chroma_key_color = 0xFF00FF00u;
  self->chroma_key_tolerance = 0.2;
  self->chroma_key_smoothness = 0.1;
  self->device_id = GST_VFHIP_DEFAULT_DEVICE_ID;
  self->async.submit = vf_async_submit;
  self->async.wait = vf_async_wait;
}

gboolean
gst_vfhip_videofilter_register (GstPlugin * plugin)
{
  gboolean ok = gst_element_register (plugin, "vfhipvideofilter", GST_RANK_NONE, gst_vfhip_videofilter_get_type ());
#ifdef VFHIP_REGISTER_VFMETAL_NAMES
  ok &= gst_element_register (plugin, "vfmetalvideofilter", GST_RANK_NONE, gst_vfhip_videofilter_get_type ());
#endif
  return ok;
}
