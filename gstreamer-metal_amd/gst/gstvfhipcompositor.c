/* gst/gstvfhipcompositor.c — `vfhipcompositor`: N-input alpha / z-order compositor on an MI355X.
 *
 * Drop-in for the reference's vfmetalcompositor (compositor/gstvfmetalcompositor.{h,m}, gstvfmetalcompositorpad.m):
 * GstChildProxy, rank PRIMARY + 2 (:177-178), request pads sink_%u { BGRA, RGBA, NV12, I420 } and the same src template
 * (:65-78), element properties background {checker, black, white, transparent} and zero-size-is-unscaled (:1035-1051),
 * pad properties xpos, ypos, width, height, alpha, operator {source, over, add}, sizing-policy {none, keep-aspect-ratio}
 * (gstvfmetalcompositorpad.m:282-315) plus GstVideoAggregatorPad's own zorder, repeat-after-eos and max-last-buffer-repeat.
 * Inputs keep their own sizes (update_caps does not intersect pad sizes, :394-458); the output size is the bounding
 * box of the positioned pads, BGRA preferred, highest input frame rate (:460-540).
 *
 * ONE implementation, on every GStreamer version: the element sits directly on GstAggregator (gst-plugins-base 1.14 has
 * no GstVideoAggregator) and carries the part of GstVideoAggregator the reference relies on (:171-174, :574-684) itself
 * — see comp_fill_queues (): output frames are cut on the OUTPUT frame rate from the src segment position; every pad
 * contributes the buffer whose running-time interval covers the output frame, a slower pad's buffer is held and shown
 * again, buffers that end before the output frame are dropped, a pad that reached EOS disappears once its last buffer
 * has run out (or stays with repeat-after-eos), EOS goes downstream when every pad is done.  Pads of different frame
 * rates therefore composite like in the reference.  Obscured-pad culling (gstvfmetalcompositorpad.m:180-255), pointer
 * navigation (:706-787), async-depth and memory:HIPMemory are part of the same code.
 *
 * From GstVideoAggregator too, since the reference inherits them from that base class: QoS (the QOS events of a late sink in a non-live pipeline make
 * aggregate () skip output frames that are already late — not composited, their time passes, a QoS message is posted: comp_update_qos / comp_qos_jitter)
 * and the pad property max-last-buffer-repeat (comp_cap_repeat).
 * NOT carried over: the GstVideoAggregatorPad C API on the request pads — an application that casts them to GstVideoAggregatorPad, rather than setting
 * properties by name, does not drop in (this element has no such base class on 1.14). */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include "gstvfhip.h"

GST_DEBUG_CATEGORY_STATIC (gst_vfhip_compositor_debug);
#define GST_CAT_DEFAULT gst_vfhip_compositor_debug

#include <gst/base/gstaggregator.h>
#include <stdlib.h>

#define VFHIP_COMP_FORMATS "{ BGRA, RGBA, NV12, I420 }"
enum { SIZING_NONE = 0, SIZING_KEEP_ASPECT = 1 };

typedef struct
{
  GstAggregatorPad parent;
  GstVideoInfo info;
  gboolean have_info;
  gint xpos, ypos, width, height;
  gdouble alpha;
  gint op, sizing_policy;
  guint zorder;
  gboolean repeat_after_eos;
  GstClockTime max_last_buffer_repeat;   /* GstVideoAggregatorPad's property of the same name: how long a pad that is not at EOS keeps showing its last buffer */
  /* the buffer this pad currently shows and its running-time interval (GstVideoAggregatorPad's buffer / start_time / end_time) */
  GstBuffer *cur;
  GstClockTime cur_start, cur_end;
} GstVfHipCompositorPad;
typedef struct
{
  GstAggregatorPadClass parent_class;
} GstVfHipCompositorPadClass;

enum { PAD_PROP_0, PAD_PROP_XPOS, PAD_PROP_YPOS, PAD_PROP_WIDTH, PAD_PROP_HEIGHT, PAD_PROP_ALPHA, PAD_PROP_OPERATOR, PAD_PROP_SIZING_POLICY, PAD_PROP_ZORDER, PAD_PROP_REPEAT_AFTER_EOS, PAD_PROP_MAX_LAST_BUFFER_REPEAT };

static GType
comp_enum (const gchar * name, const GEnumValue * v, gsize * once)
{
  if (g_once_init_enter (once))
    g_once_init_leave (once, g_enum_register_static (name, v));
  return (GType) *once;
}

static GType
comp_operator_type (void)
{
  static gsize t = 0;
  static const GEnumValue v[] = { {VFHIP_BLEND_SOURCE, "Source", "source"}, {VFHIP_BLEND_OVER, "Over", "over"}, {VFHIP_BLEND_ADD, "Add", "add"}, {0, NULL, NULL} };
  return comp_enum ("GstVfHipCompositorOperator", v, &t);
}

static GType
comp_sizing_type (void)
{
  static gsize t = 0;
  static const GEnumValue v[] = {
    {SIZING_NONE, "None: image is scaled to fill configured destination rectangle without padding or keeping the aspect ratio", "none"},
    {SIZING_KEEP_ASPECT, "Keep Aspect Ratio: image is scaled to fit destination rectangle with preserved aspect ratio", "keep-aspect-ratio"},
    {0, NULL, NULL}
  };
  return comp_enum ("GstVfHipCompositorSizingPolicy", v, &t);
}

static GType
comp_background_type (void)
{
  static gsize t = 0;
  static const GEnumValue v[] = {
    {VFHIP_BG_CHECKER, "Checker pattern", "checker"}, {VFHIP_BG_BLACK, "Black", "black"}, {VFHIP_BG_WHITE, "White", "white"},
    {VFHIP_BG_TRANSPARENT, "Transparent Background to enable further compositing", "transparent"}, {0, NULL, NULL}
  };
  return comp_enum ("GstVfHipCompositorBackground", v, &t);
}

G_DEFINE_TYPE (GstVfHipCompositorPad, gst_vfhip_compositor_pad, GST_TYPE_AGGREGATOR_PAD);
#define CPAD(o) ((GstVfHipCompositorPad *) (o))

static void
cpad_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstVfHipCompositorPad *pad = CPAD (object);
  GstObject *agg;
  GST_OBJECT_LOCK (pad);
  switch (id) {
    case PAD_PROP_XPOS: pad->xpos = g_value_get_int (value); break;
    case PAD_PROP_YPOS: pad->ypos = g_value_get_int (value); break;
    case PAD_PROP_WIDTH: pad->width = g_value_get_int (value); break;
    case PAD_PROP_HEIGHT: pad->height = g_value_get_int (value); break;
    case PAD_PROP_ALPHA: pad->alpha = g_value_get_double (value); break;
    case PAD_PROP_OPERATOR: pad->op = g_value_get_enum (value); break;
    case PAD_PROP_SIZING_POLICY: pad->sizing_policy = g_value_get_enum (value); break;
    case PAD_PROP_ZORDER: pad->zorder = g_value_get_uint (value); break;
    case PAD_PROP_REPEAT_AFTER_EOS: pad->repeat_after_eos = g_value_get_boolean (value); break;
    case PAD_PROP_MAX_LAST_BUFFER_REPEAT: pad->max_last_buffer_repeat = g_value_get_uint64 (value); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (pad);
  if (id != PAD_PROP_ALPHA && id != PAD_PROP_OPERATOR && id != PAD_PROP_ZORDER && id != PAD_PROP_REPEAT_AFTER_EOS && id != PAD_PROP_MAX_LAST_BUFFER_REPEAT && (agg = gst_object_get_parent (GST_OBJECT (pad)))) {
    gst_pad_mark_reconfigure (GST_AGGREGATOR (agg)->srcpad);          /* the output bounding box may have changed */
    gst_object_unref (agg);
  }
}

static void
cpad_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstVfHipCompositorPad *pad = CPAD (object);
  GST_OBJECT_LOCK (pad);
  switch (id) {
    case PAD_PROP_XPOS: g_value_set_int (value, pad->xpos); break;
    case PAD_PROP_YPOS: g_value_set_int (value, pad->ypos); break;
    case PAD_PROP_WIDTH: g_value_set_int (value, pad->width); break;
    case PAD_PROP_HEIGHT: g_value_set_int (value, pad->height); break;
    case PAD_PROP_ALPHA: g_value_set_double (value, pad->alpha); break;
    case PAD_PROP_OPERATOR: g_value_set_enum (value, pad->op); break;
    case PAD_PROP_SIZING_POLICY: g_value_set_enum (value, pad->sizing_policy); break;
    case PAD_PROP_ZORDER: g_value_set_uint (value, pad->zorder); break;
    case PAD_PROP_REPEAT_AFTER_EOS: g_value_set_boolean (value, pad->repeat_after_eos); break;
    case PAD_PROP_MAX_LAST_BUFFER_REPEAT: g_value_set_uint64 (value, pad->max_last_buffer_repeat); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
  GST_OBJECT_UNLOCK (pad);
}

static void
cpad_finalize (GObject * object)
{
  gst_buffer_replace (&CPAD (object)->cur, NULL);
  G_OBJECT_CLASS (gst_vfhip_compositor_pad_parent_class)->finalize (object);
}

static void
gst_vfhip_compositor_pad_class_init (GstVfHipCompositorPadClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  const GParamFlags f = (GParamFlags) (G_PARAM_READWRITE | GST_PARAM_CONTROLLABLE | G_PARAM_STATIC_STRINGS);
  oc->set_property = cpad_set_property;
  oc->get_property = cpad_get_property;
  oc->finalize = cpad_finalize;
  g_object_class_install_property (oc, PAD_PROP_XPOS, g_param_spec_int ("xpos", "X Position", "X Position of the picture", G_MININT, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PAD_PROP_YPOS, g_param_spec_int ("ypos", "Y Position", "Y Position of the picture", G_MININT, G_MAXINT, 0, f));
  g_object_class_install_property (oc, PAD_PROP_WIDTH, g_param_spec_int ("width", "Width", "Width of the picture", G_MININT, G_MAXINT, -1, f));
  g_object_class_install_property (oc, PAD_PROP_HEIGHT, g_param_spec_int ("height", "Height", "Height of the picture", G_MININT, G_MAXINT, -1, f));
  g_object_class_install_property (oc, PAD_PROP_ALPHA, g_param_spec_double ("alpha", "Alpha", "Alpha of the picture", 0.0, 1.0, 1.0, f));
  g_object_class_install_property (oc, PAD_PROP_OPERATOR, g_param_spec_enum ("operator", "Operator",
          "Blending operator to use for blending this pad over the previous ones", comp_operator_type (), VFHIP_BLEND_OVER, f));
  g_object_class_install_property (oc, PAD_PROP_SIZING_POLICY, g_param_spec_enum ("sizing-policy", "Sizing policy",
          "Sizing policy to use for image scaling", comp_sizing_type (), SIZING_NONE, f));
  g_object_class_install_property (oc, PAD_PROP_ZORDER, g_param_spec_uint ("zorder", "Z-Order", "Z Order of the picture", 0, G_MAXUINT, 0, f));
  g_object_class_install_property (oc, PAD_PROP_REPEAT_AFTER_EOS, g_param_spec_boolean ("repeat-after-eos", "Repeat After EOS",
          "Repeat the last frame after EOS until all pads are EOS", FALSE, f));
  g_object_class_install_property (oc, PAD_PROP_MAX_LAST_BUFFER_REPEAT, g_param_spec_uint64 ("max-last-buffer-repeat", "Max Last Buffer Repeat",
          "Repeat last buffer for time (in ns, -1=until EOS), behaviour on EOS is not affected", 0, G_MAXUINT64, GST_CLOCK_TIME_NONE, f));
}

static void
gst_vfhip_compositor_pad_init (GstVfHipCompositorPad * pad)
{
  pad->width = pad->height = -1;
  pad->alpha = 1.0;
  pad->op = VFHIP_BLEND_OVER;
  pad->sizing_policy = SIZING_NONE;
  pad->cur_start = pad->cur_end = GST_CLOCK_TIME_NONE;
  pad->max_last_buffer_repeat = GST_CLOCK_TIME_NONE;
  gst_video_info_init (&pad->info);
}

typedef struct
{
  GstAggregator parent;
  VfHipCompositor *renderer;
  gint device_id, background;
  gboolean zero_size_is_unscaled;
  gboolean ignore_inactive_pads;   /* handed to GstAggregator where it has the notion (>= 1.20), kept for the getter otherwise */
  GstVideoInfo out_info;
  gboolean have_out_info, out_is_device;
  guint64 n_frames;                             /* output frames since the last (re)start of the time line */
  GstClockTime ts_offset;                       /* segment time of output frame 0: frame k ends at ts_offset + (k + 1) / fps, no drift */
  GstVfHipPinStats pin;                         /* recurring pageable input memories are page-locked in place */
  /* async-depth=1: the composite submitted last (vfhip_compositor_submit); its input buffers and its output buffer stay
   * mapped until the next aggregate() has submitted its own and waits for this one */
  gint pref_w, pref_h, pref_fn, pref_fd;        /* what update_src_caps found: bounding box of the pads, highest input frame rate */
  gint async_depth;
  gboolean have_pending;
  struct { GstBuffer *outbuf; GstVideoFrame out; guint n; GstBuffer **bufs; GstVideoFrame *frames; } pending;
  /* QoS like GstVideoAggregator (gst_video_aggregator_update_qos / _do_qos): what the last QOS event from downstream said (object lock), and the
   * counts its QoS messages carry */
  gdouble qos_proportion;
  GstClockTime qos_earliest;
  guint64 qos_processed, qos_dropped;
} GstVfHipCompositor;
typedef struct
{
  GstAggregatorClass parent_class;
} GstVfHipCompositorClass;

enum { PROP_0, PROP_BACKGROUND, PROP_ZERO_SIZE_IS_UNSCALED, PROP_IGNORE_INACTIVE_PADS, PROP_DEVICE_ID, PROP_ASYNC_DEPTH };

static GstStaticPadTemplate comp_src_template = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_COMP_FORMATS)));
/* sink pads also take memory:HIPMemory buffers from upstream vfhip elements (gstvfhipmemory.c) */
static GstStaticPadTemplate comp_sink_template = GST_STATIC_PAD_TEMPLATE ("sink_%u", GST_PAD_SINK, GST_PAD_REQUEST,
    GST_STATIC_CAPS (GST_VFHIP_CAPS (VFHIP_COMP_FORMATS)));

static void comp_child_proxy_init (gpointer g_iface, gpointer iface_data);
G_DEFINE_TYPE_WITH_CODE (GstVfHipCompositor, gst_vfhip_compositor, GST_TYPE_AGGREGATOR,
    G_IMPLEMENT_INTERFACE (GST_TYPE_CHILD_PROXY, comp_child_proxy_init));
#define COMP(o) ((GstVfHipCompositor *) (o))

/* destination rectangle of a pad (same rules as the >= 1.16 variant above / the reference :202-325) */
static void
comp_pad_rect (GstVfHipCompositor * self, GstVfHipCompositorPad * cpad, gint out_par_n, gint out_par_d, gint * w, gint * h, gint * xoff, gint * yoff)
{
  gint pw, ph;
  guint dn, dd;
  *w = *h = *xoff = *yoff = 0;
  if (!cpad->have_info)
    return;
  if (self->zero_size_is_unscaled) {
    pw = cpad->width <= 0 ? GST_VIDEO_INFO_WIDTH (&cpad->info) : cpad->width;
    ph = cpad->height <= 0 ? GST_VIDEO_INFO_HEIGHT (&cpad->info) : cpad->height;
  } else {
    pw = cpad->width < 0 ? GST_VIDEO_INFO_WIDTH (&cpad->info) : cpad->width;
    ph = cpad->height < 0 ? GST_VIDEO_INFO_HEIGHT (&cpad->info) : cpad->height;
  }
  if (pw == 0 || ph == 0)
    return;
  if (!gst_video_calculate_display_ratio (&dn, &dd, pw, ph, GST_VIDEO_INFO_PAR_N (&cpad->info), GST_VIDEO_INFO_PAR_D (&cpad->info), out_par_n, out_par_d))
    return;
  if (cpad->sizing_policy == SIZING_NONE) {
    if (ph % dn == 0) pw = gst_util_uint64_scale_int (ph, dn, dd);
    else if (pw % dd == 0) ph = gst_util_uint64_scale_int (pw, dd, dn);
    else pw = gst_util_uint64_scale_int (ph, dn, dd);
  } else {
    gint fn, fd, tn, td, num, den;
    if (!gst_util_fraction_multiply (GST_VIDEO_INFO_WIDTH (&cpad->info), GST_VIDEO_INFO_HEIGHT (&cpad->info),
            GST_VIDEO_INFO_PAR_N (&cpad->info), GST_VIDEO_INFO_PAR_D (&cpad->info), &fn, &fd)) fn = fd = -1;
    if (!gst_util_fraction_multiply (pw, ph, out_par_n, out_par_d, &tn, &td)) tn = td = -1;
    if (fn != tn || fd != td) {
      GstVideoRectangle src, dst, res;
      if (fn == -1 || !gst_util_fraction_multiply (fn, fd, out_par_d, out_par_n, &num, &den))
        return;
      src.x = src.y = 0; src.w = pw; src.h = gst_util_uint64_scale_int (pw, den, num);
      if (src.h == 0)
        return;
      dst.x = dst.y = 0; dst.w = pw; dst.h = ph;
      gst_video_sink_center_rect (src, dst, &res, TRUE);
      *xoff = res.x; *yoff = res.y; pw = res.w; ph = res.h;
    }
  }
  *w = pw; *h = ph;
}

static gboolean
comp_sink_event (GstAggregator * agg, GstAggregatorPad * pad, GstEvent * event)
{
  if (GST_EVENT_TYPE (event) == GST_EVENT_CAPS) {
    GstCaps *caps;
    GstVfHipCompositorPad *cpad = CPAD (pad);
    gst_event_parse_caps (event, &caps);
    GST_OBJECT_LOCK (pad);
    cpad->have_info = gst_video_info_from_caps (&cpad->info, caps);
    GST_OBJECT_UNLOCK (pad);
    gst_pad_mark_reconfigure (agg->srcpad);
  }
  return GST_AGGREGATOR_CLASS (gst_vfhip_compositor_parent_class)->sink_event (agg, pad, event);
}

static gboolean
comp_sink_query (GstAggregator * agg, GstAggregatorPad * pad, GstQuery * query)
{
  if (GST_QUERY_TYPE (query) == GST_QUERY_CAPS) {        /* every pad may carry its own size and format */
    GstCaps *filter, *tmpl = gst_pad_get_pad_template_caps (GST_PAD (pad)), *res;
    gst_query_parse_caps (query, &filter);
    res = filter ? gst_caps_intersect_full (filter, tmpl, GST_CAPS_INTERSECT_FIRST) : gst_caps_ref (tmpl);
    gst_query_set_caps_result (query, res);
    gst_caps_unref (res); gst_caps_unref (tmpl);
    return TRUE;
  }
  if (GST_QUERY_TYPE (query) == GST_QUERY_ACCEPT_CAPS) {
    GstCaps *caps, *tmpl = gst_pad_get_pad_template_caps (GST_PAD (pad));
    gst_query_parse_accept_caps (query, &caps);
    gst_query_set_accept_caps_result (query, gst_caps_is_subset (caps, tmpl));
    gst_caps_unref (tmpl);
    return TRUE;
  }
  return GST_AGGREGATOR_CLASS (gst_vfhip_compositor_parent_class)->sink_query (agg, pad, query);
}

/* output size = bounding box of the positioned pads; BGRA preferred; highest input frame rate (reference :394-540) */
static GstFlowReturn
comp_update_src_caps (GstAggregator * agg, GstCaps * caps, GstCaps ** ret)
{
  GstVfHipCompositor *self = COMP (agg);
  gint bw = -1, bh = -1, fn = -1, fd = -1;
  gdouble best = 0.0;
  GList *l;
  GstCaps *tmpl;
  GST_OBJECT_LOCK (agg);
  for (l = GST_ELEMENT (agg)->sinkpads; l; l = l->next) {
    GstVfHipCompositorPad *cpad = CPAD (l->data);
    gint w, h, xo, yo;
    gdouble fps = 0.0;
    comp_pad_rect (self, cpad, 1, 1, &w, &h, &xo, &yo);
    if (w == 0 || h == 0)
      continue;
    bw = MAX (bw, w + MAX (cpad->xpos + 2 * xo, 0));
    bh = MAX (bh, h + MAX (cpad->ypos + 2 * yo, 0));
    if (GST_VIDEO_INFO_FPS_D (&cpad->info) != 0)
      gst_util_fraction_to_double (GST_VIDEO_INFO_FPS_N (&cpad->info), GST_VIDEO_INFO_FPS_D (&cpad->info), &fps);
    if (fps > best) { best = fps; fn = GST_VIDEO_INFO_FPS_N (&cpad->info); fd = GST_VIDEO_INFO_FPS_D (&cpad->info); }
  }
  GST_OBJECT_UNLOCK (agg);
  if (bw <= 0 || bh <= 0)
    return GST_AGGREGATOR_FLOW_NEED_DATA;                 /* no pad has caps yet */
  if (fn <= 0 || fd <= 0) { fn = 25; fd = 1; }
  /* like the reference's _update_caps / _fixate_caps (gstvfmetalcompositor.m:394-540): the pads' bounding box and the highest
   * input frame rate are what the output PREFERS; a downstream caps filter may still ask for another size (fixate_src_caps
   * takes the nearest one) */
  self->pref_w = bw; self->pref_h = bh; self->pref_fn = fn; self->pref_fd = fd;
  tmpl = gst_static_pad_template_get_caps (&comp_src_template);
  *ret = gst_caps_copy (tmpl);
  gst_caps_unref (tmpl);
  if (caps) {
    GstCaps *tmp = gst_caps_intersect (*ret, caps);
    gst_caps_unref (*ret);
    *ret = tmp;
  }
  return GST_FLOW_OK;
}

static GstCaps *
comp_fixate_src_caps (GstAggregator * agg, GstCaps * caps)
{
  GstVfHipCompositor *self = COMP (agg);
  GstStructure *s;
  caps = gst_caps_make_writable (gst_caps_truncate (caps));
  s = gst_caps_get_structure (caps, 0);
  gst_structure_fixate_field_string (s, "format", "BGRA");
  if (self->pref_w > 0 && self->pref_h > 0) {
    gst_structure_fixate_field_nearest_int (s, "width", self->pref_w);
    gst_structure_fixate_field_nearest_int (s, "height", self->pref_h);
    gst_structure_fixate_field_nearest_fraction (s, "framerate", self->pref_fn, self->pref_fd);
  }
  if (gst_structure_has_field (s, "pixel-aspect-ratio"))
    gst_structure_fixate_field_nearest_fraction (s, "pixel-aspect-ratio", 1, 1);
  return gst_caps_fixate (caps);
}

static GstFlowReturn comp_finish_pending (GstVfHipCompositor * self, gboolean push);

static gboolean
comp_negotiated_src_caps (GstAggregator * agg, GstCaps * caps)
{
  GstVfHipCompositor *self = COMP (agg);
  VfHipVideoInfo out;
  (void) comp_finish_pending (self, TRUE);                 /* a renegotiation mid-stream: the composite in flight belongs to the old caps */
  if (!gst_video_info_from_caps (&self->out_info, caps))
    return FALSE;
  self->have_out_info = TRUE;
  self->out_is_device = gst_vfhip_caps_has_hip_feature (caps);
  if (!self->renderer && !(self->renderer = vfhip_compositor_new (self->device_id))) {
    GST_ERROR_OBJECT (self, "no HIP renderer: %s", vfhip_last_error_string ());
    return FALSE;
  }
  gst_vfhip_info (&self->out_info, &out);
  if (vfhip_compositor_configure (self->renderer, &out) != VFHIP_OK) {
    GST_ERROR_OBJECT (self, "configure failed: %s", vfhip_last_error_string ());
    return FALSE;
  }
  return GST_AGGREGATOR_CLASS (gst_vfhip_compositor_parent_class)->negotiated_src_caps (agg, caps);
}

typedef struct { GstVfHipCompositorPad *pad; guint order; } PadRef;
static int
padref_cmp (const void *a, const void *b)
{
  const PadRef *x = a, *y = b;
  if (x->pad->zorder != y->pad->zorder) return x->pad->zorder < y->pad->zorder ? -1 : 1;
  return x->order < y->order ? -1 : (x->order > y->order);
}

/* completes the composite in flight: waits for it, releases its inputs and pushes (or, with push = FALSE, drops) its output */
static GstFlowReturn
comp_finish_pending (GstVfHipCompositor * self, gboolean push)
{
  GstFlowReturn flow = GST_FLOW_OK;
  GstBuffer *outbuf;
  guint i;
  gint rc;
  if (!self->have_pending)
    return GST_FLOW_OK;
  self->have_pending = FALSE;
  rc = vfhip_compositor_wait (self->renderer);
  gst_video_frame_unmap (&self->pending.out);
  for (i = 0; i < self->pending.n; i++) {
    if (self->pending.frames[i].buffer) gst_video_frame_unmap (&self->pending.frames[i]);
    if (self->pending.bufs[i]) gst_buffer_unref (self->pending.bufs[i]);
  }
  g_free (self->pending.frames); g_free (self->pending.bufs);
  outbuf = self->pending.outbuf;
  memset (&self->pending, 0, sizeof (self->pending));
  if (rc != VFHIP_OK) {
    GST_ERROR_OBJECT (self, "HIP compositing failed: %s", vfhip_last_error_string ());
    gst_buffer_unref (outbuf);
    return GST_FLOW_ERROR;
  }
  if (push)
    flow = gst_aggregator_finish_buffer (GST_AGGREGATOR (self), outbuf);
  else
    gst_buffer_unref (outbuf);
  return flow;
}

/* Which buffer does every pad show during the output frame [out_start, out_end) (running time)?  GstVideoAggregator's
 * gst_video_aggregator_fill_queues restated on GstAggregator 1.14 (the reference inherits it: gstvfmetalcompositor.m:171-174):
 *   - a queued buffer that overlaps the output frame becomes the pad's current buffer and leaves the queue;
 *   - one that starts at or after the end of the output frame stays queued, the current buffer is shown again
 *     (a pad slower than the output);
 *   - one that ended before the output frame is dropped and the pad is asked for more (a pad faster than the output);
 *   - a pad at EOS keeps its last buffer while that still runs, then disappears (unless repeat-after-eos);
 * -> GST_FLOW_OK, GST_AGGREGATOR_FLOW_NEED_DATA (wait for the pads that were asked for more) or GST_FLOW_EOS (all done). */
/* max-last-buffer-repeat: a pad that is NOT at EOS shows its last buffer for at most that long past the buffer's end (EOS is repeat-after-eos's business) */
static void
comp_cap_repeat (GstVfHipCompositorPad * cpad, GstClockTime out_start)
{
  if (cpad->cur && GST_CLOCK_TIME_IS_VALID (cpad->max_last_buffer_repeat) && GST_CLOCK_TIME_IS_VALID (cpad->cur_end) && GST_CLOCK_TIME_IS_VALID (out_start) &&
      out_start > cpad->cur_end && out_start - cpad->cur_end > cpad->max_last_buffer_repeat) {
    GST_DEBUG_OBJECT (cpad, "last buffer repeated for more than %" GST_TIME_FORMAT ": dropped", GST_TIME_ARGS (cpad->max_last_buffer_repeat));
    gst_buffer_replace (&cpad->cur, NULL);
    cpad->cur_start = cpad->cur_end = GST_CLOCK_TIME_NONE;
  }
}

static GstFlowReturn
comp_fill_queues (GstVfHipCompositor * self, PadRef * refs, guint n, GstClockTime out_start, GstClockTime out_end, GstClockTime out_dur)
{
  gboolean eos = TRUE, need_more = FALSE;
  guint i;
  for (i = 0; i < n; i++) {
    GstVfHipCompositorPad *cpad = refs[i].pad;
    GstAggregatorPad *apad = GST_AGGREGATOR_PAD (cpad);
    for (;;) {
      GstBuffer *buf = gst_aggregator_pad_peek_buffer (apad);
      GstClockTime start, end, dur;
      if (!buf) {
        if (gst_aggregator_pad_is_eos (apad)) {
          if (cpad->cur && GST_CLOCK_TIME_IS_VALID (cpad->cur_end) && cpad->cur_end > out_start)
            eos = FALSE;                                   /* its last buffer still covers this frame */
          else if (!(cpad->cur && cpad->repeat_after_eos)) {
            gst_buffer_replace (&cpad->cur, NULL);
            cpad->cur_start = cpad->cur_end = GST_CLOCK_TIME_NONE;
          }
        } else {
          eos = FALSE;                                     /* nothing queued yet (timeout / just asked for more) */
          comp_cap_repeat (cpad, out_start);
        }
        break;
      }
      start = GST_BUFFER_PTS_IS_VALID (buf) ? GST_BUFFER_PTS (buf) : GST_BUFFER_DTS (buf);
      if (!GST_CLOCK_TIME_IS_VALID (start)) {              /* untimed buffer: shown from now on */
        gst_buffer_replace (&cpad->cur, buf);
        cpad->cur_start = cpad->cur_end = GST_CLOCK_TIME_NONE;
        gst_buffer_unref (buf);
        gst_aggregator_pad_drop_buffer (apad);
        eos = FALSE;
        break;
      }
      if (GST_BUFFER_DURATION_IS_VALID (buf))
        dur = GST_BUFFER_DURATION (buf);
      else if (cpad->have_info && GST_VIDEO_INFO_FPS_N (&cpad->info) > 0)
        dur = gst_util_uint64_scale (GST_SECOND, GST_VIDEO_INFO_FPS_D (&cpad->info), GST_VIDEO_INFO_FPS_N (&cpad->info));
      else
        dur = out_dur;
      end = start + dur;
      if (apad->segment.format == GST_FORMAT_TIME) {
        guint64 cs, ce;
        if (!gst_segment_clip (&apad->segment, GST_FORMAT_TIME, start, end, &cs, &ce)) {
          GST_DEBUG_OBJECT (cpad, "buffer outside of its segment: dropped");
          gst_buffer_unref (buf);
          gst_aggregator_pad_drop_buffer (apad);
          need_more = TRUE;
          continue;
        }
        start = gst_segment_to_running_time (&apad->segment, GST_FORMAT_TIME, cs);
        end = gst_segment_to_running_time (&apad->segment, GST_FORMAT_TIME, ce);
      }
      if (cpad->cur && GST_CLOCK_TIME_IS_VALID (cpad->cur_end) && cpad->cur_end > end) {
        GST_DEBUG_OBJECT (cpad, "buffer from the past: dropped");
        gst_buffer_unref (buf);
        gst_aggregator_pad_drop_buffer (apad);
        need_more = TRUE;
        continue;
      }
      if (end >= out_start && start < out_end) {           /* this frame's buffer */
        GST_LOG_OBJECT (cpad, "taking buffer %" GST_TIME_FORMAT " - %" GST_TIME_FORMAT " for output %" GST_TIME_FORMAT, GST_TIME_ARGS (start), GST_TIME_ARGS (end), GST_TIME_ARGS (out_start));
        gst_buffer_replace (&cpad->cur, buf);
        cpad->cur_start = start; cpad->cur_end = end;
        gst_buffer_unref (buf);
        gst_aggregator_pad_drop_buffer (apad);
        eos = FALSE;
        break;
      }
      if (start >= out_end) {                              /* for a later output frame: stays queued, the current buffer repeats */
        GST_LOG_OBJECT (cpad, "keeping buffer %" GST_TIME_FORMAT " for later, repeating the current one", GST_TIME_ARGS (start));
        gst_buffer_unref (buf);
        comp_cap_repeat (cpad, out_start);
        eos = FALSE;
        break;
      }
      GST_DEBUG_OBJECT (cpad, "buffer %" GST_TIME_FORMAT " - %" GST_TIME_FORMAT " is too old for output %" GST_TIME_FORMAT ": dropped", GST_TIME_ARGS (start), GST_TIME_ARGS (end), GST_TIME_ARGS (out_start));
      gst_buffer_unref (buf);
      gst_aggregator_pad_drop_buffer (apad);
      need_more = TRUE;
    }
  }
  (void) self;
  if (need_more)
    return GST_AGGREGATOR_FLOW_NEED_DATA;
  return eos ? GST_FLOW_EOS : GST_FLOW_OK;
}

static void
comp_drop_current_buffers (GstVfHipCompositor * self)
{
  GList *l, *pads = NULL;
  GST_OBJECT_LOCK (self);
  for (l = GST_ELEMENT (self)->sinkpads; l; l = l->next)
    pads = g_list_prepend (pads, gst_object_ref (l->data));
  GST_OBJECT_UNLOCK (self);
  for (l = pads; l; l = l->next) {
    GstVfHipCompositorPad *cpad = CPAD (l->data);
    gst_buffer_replace (&cpad->cur, NULL);
    cpad->cur_start = cpad->cur_end = GST_CLOCK_TIME_NONE;
    gst_object_unref (l->data);
  }
  g_list_free (pads);
  self->n_frames = 0;
  self->ts_offset = GST_CLOCK_TIME_NONE;
}

/* QoS (GstVideoAggregator's): a QOS event from downstream says how late the frame with `timestamp` was; in a non-live pipeline output frames whose
 * running time lies before `earliest` are not composited at all — their time still passes, their inputs are still consumed — and a QoS message
 * tells the application */
static void
comp_reset_qos (GstVfHipCompositor * self)
{
  GST_OBJECT_LOCK (self);
  self->qos_proportion = 0.5;
  self->qos_earliest = GST_CLOCK_TIME_NONE;
  GST_OBJECT_UNLOCK (self);
  self->qos_processed = self->qos_dropped = 0;
}

static void
comp_update_qos (GstVfHipCompositor * self, gdouble proportion, GstClockTimeDiff diff, GstClockTime timestamp)
{
  const gboolean live = GST_CLOCK_TIME_IS_VALID (gst_aggregator_get_latency (GST_AGGREGATOR (self)));
  GST_OBJECT_LOCK (self);
  self->qos_proportion = proportion;
  if (!live && GST_CLOCK_TIME_IS_VALID (timestamp)) {
    if (diff > 0) {
      const gint fn = GST_VIDEO_INFO_FPS_N (&self->out_info), fd = GST_VIDEO_INFO_FPS_D (&self->out_info);
      self->qos_earliest = timestamp + 2 * diff + ((fn > 0 && fd > 0) ? gst_util_uint64_scale_int_round (GST_SECOND, fd, fn) : 0);
    } else
      self->qos_earliest = (diff < 0 && (GstClockTime) (-diff) > timestamp) ? 0 : timestamp + diff;
  } else
    self->qos_earliest = GST_CLOCK_TIME_NONE;
  GST_OBJECT_UNLOCK (self);
}

/* > 0: the output frame starting at running time `rt` is late by that much */
static GstClockTimeDiff
comp_qos_jitter (GstVfHipCompositor * self, GstClockTime rt, gdouble * proportion)
{
  GstClockTime earliest;
  GST_OBJECT_LOCK (self);
  earliest = self->qos_earliest;
  *proportion = self->qos_proportion;
  GST_OBJECT_UNLOCK (self);
  if (!GST_CLOCK_TIME_IS_VALID (rt) || !GST_CLOCK_TIME_IS_VALID (earliest))
    return -1;
  return GST_CLOCK_DIFF (rt, earliest);
}

static GstFlowReturn
comp_aggregate (GstAggregator * agg, gboolean timeout)
{
  GstVfHipCompositor *self = COMP (agg);
  GList *l;
  guint n = 0, i, used = 0;
  GstClockTime out_start, out_end, out_start_rt, out_end_rt, out_dur;
  gint fps_n, fps_d;
  GstFlowReturn sel;
  GstSegment *seg = &GST_AGGREGATOR_PAD (agg->srcpad)->segment;      /* the output segment lives on the src pad since 1.14 */
  PadRef *refs;
  GstBuffer **bufs, *outbuf;
  GstVideoFrame *frames, out;
  VfHipPadInput *pads;
  VfHipFrame vout;
  GstAllocator *alloc;
  gboolean covered = FALSE, keep = FALSE;
  gint rc;
  (void) timeout;
  if (!self->renderer || !self->have_out_info)
    return GST_FLOW_NOT_NEGOTIATED;
  GST_OBJECT_LOCK (agg);
  n = GST_ELEMENT (agg)->numsinkpads;
  refs = g_new0 (PadRef, MAX (n, 1));
  for (l = GST_ELEMENT (agg)->sinkpads, i = 0; l && i < n; l = l->next, i++) {
    refs[i].pad = gst_object_ref (l->data);
    refs[i].order = i;
  }
  GST_OBJECT_UNLOCK (agg);
  qsort (refs, n, sizeof (PadRef), padref_cmp);            /* draw order = zorder, ties in pad creation order */
  bufs = g_new0 (GstBuffer *, MAX (n, 1));
  frames = g_new0 (GstVideoFrame, MAX (n, 1));
  pads = g_new0 (VfHipPadInput, MAX (n, 1));
  /* this output frame: [position, ts_offset + (n_frames + 1) / fps) of the src segment — cut on the OUTPUT frame rate, whatever
   * the inputs run at (GstVideoAggregator's time line: gst_video_aggregator_aggregate) */
  fps_n = GST_VIDEO_INFO_FPS_N (&self->out_info); fps_d = GST_VIDEO_INFO_FPS_D (&self->out_info);
  if (fps_n <= 0 || fps_d <= 0) { fps_n = 25; fps_d = 1; }
  if (!GST_CLOCK_TIME_IS_VALID (seg->position) || seg->position < seg->start)
    seg->position = seg->start;
  out_start = seg->position;
  if (self->n_frames == 0 || !GST_CLOCK_TIME_IS_VALID (self->ts_offset))
    self->ts_offset = out_start;
  out_end = self->ts_offset + gst_util_uint64_scale_round (self->n_frames + 1, GST_SECOND * fps_d, fps_n);
  if (GST_CLOCK_TIME_IS_VALID (seg->stop))
    out_end = MIN (out_end, seg->stop);
  out_dur = gst_util_uint64_scale_round (1, GST_SECOND * fps_d, fps_n);
  out_start_rt = gst_segment_to_running_time (seg, GST_FORMAT_TIME, out_start);
  out_end_rt = gst_segment_to_running_time (seg, GST_FORMAT_TIME, out_end);
  sel = (out_end <= out_start && GST_CLOCK_TIME_IS_VALID (seg->stop)) ? GST_FLOW_EOS :
      comp_fill_queues (self, refs, n, out_start_rt, out_end_rt, out_dur);
  if (sel != GST_FLOW_OK) {
    if (sel == GST_FLOW_EOS) {
      rc = comp_finish_pending (self, TRUE);               /* the last frame of an async-depth=1 stream */
      if (rc == GST_FLOW_OK) rc = GST_FLOW_EOS;
    } else
      rc = sel;                                            /* GST_AGGREGATOR_FLOW_NEED_DATA: aggregate () runs again when the pads have data */
    goto done;
  }
  {
    gdouble proportion;
    const GstClockTimeDiff jitter = comp_qos_jitter (self, out_start_rt, &proportion);
    if (jitter > 0) {
      GstMessage *msg = gst_message_new_qos (GST_OBJECT_CAST (self), FALSE, out_start_rt, gst_segment_to_stream_time (seg, GST_FORMAT_TIME, out_start),
          out_start, out_end - out_start);
      self->qos_dropped++;
      gst_message_set_qos_values (msg, jitter, proportion, 1000000);
      gst_message_set_qos_stats (msg, GST_FORMAT_BUFFERS, self->qos_processed, self->qos_dropped);
      gst_element_post_message (GST_ELEMENT_CAST (self), msg);
      GST_DEBUG_OBJECT (self, "output frame %" GST_TIME_FORMAT " is late by %" GST_TIME_FORMAT ": not composited", GST_TIME_ARGS (out_start), GST_TIME_ARGS (jitter));
      seg->position = out_end;
      self->n_frames++;
      rc = GST_FLOW_OK;
      goto done;
    }
    self->qos_processed++;
  }
  for (i = 0; i < n; i++)                                  /* this composite holds its own reference (it may stay in flight: async-depth) */
    bufs[i] = refs[i].pad->cur ? gst_buffer_ref (refs[i].pad->cur) : NULL;
  for (i = 0; i < n; i++) {
    GstVfHipCompositorPad *cpad = refs[i].pad;
    gint w, h, xo, yo;
    guint j;
    gboolean obscured = FALSE;
    if (!bufs[i] || !cpad->have_info || cpad->alpha == 0.0)
      continue;
    /* a pad completely behind an opaque later one is never seen: not mapped, not uploaded, not drawn (the reference's
     * pad_obscures_rectangle, gstvfmetalcompositor.m:329-358; here the later pad must also REPLACE what is under it —
     * operator source or over — since an `add` pad lets the lower one through) */
    comp_pad_rect (self, cpad, GST_VIDEO_INFO_PAR_N (&self->out_info), GST_VIDEO_INFO_PAR_D (&self->out_info), &w, &h, &xo, &yo);
    for (j = i + 1; j < n && !obscured; j++) {
      GstVfHipCompositorPad *up = refs[j].pad;
      gint uw, uh, uxo, uyo;
      if (!bufs[j] || !up->have_info || up->alpha != 1.0 || GST_VIDEO_INFO_HAS_ALPHA (&up->info) || up->op == VFHIP_BLEND_ADD)
        continue;
      comp_pad_rect (self, up, GST_VIDEO_INFO_PAR_N (&self->out_info), GST_VIDEO_INFO_PAR_D (&self->out_info), &uw, &uh, &uxo, &uyo);
      obscured = up->xpos + uxo <= cpad->xpos + xo && up->ypos + uyo <= cpad->ypos + yo &&
          up->xpos + uxo + uw >= cpad->xpos + xo + w && up->ypos + uyo + uh >= cpad->ypos + yo + h;
    }
    if (obscured) {
      GST_LOG_OBJECT (cpad, "obscured by a later opaque pad: skipped");
      continue;
    }
    gst_vfhip_pin_foreign_memory (bufs[i], &self->pin);
    if (!gst_video_frame_map (&frames[i], &cpad->info, bufs[i], (GstMapFlags) (GST_MAP_READ | gst_vfhip_map_flag (bufs[i], gst_vfhip_element_device (self)))))
      continue;
    comp_pad_rect (self, cpad, GST_VIDEO_INFO_PAR_N (&self->out_info), GST_VIDEO_INFO_PAR_D (&self->out_info), &w, &h, &xo, &yo);
    gst_vfhip_frame (&frames[i], &pads[used].frame);
    pads[used].xpos = cpad->xpos + xo; pads[used].ypos = cpad->ypos + yo; pads[used].width = w; pads[used].height = h;
    pads[used].alpha = cpad->alpha; pads[used].blend_mode = cpad->op;
    if (cpad->alpha == 1.0 && !GST_VIDEO_INFO_HAS_ALPHA (&cpad->info) && pads[used].xpos <= 0 && pads[used].ypos <= 0 &&
        pads[used].xpos + w >= GST_VIDEO_INFO_WIDTH (&self->out_info) && pads[used].ypos + h >= GST_VIDEO_INFO_HEIGHT (&self->out_info))
      covered = TRUE;                                      /* background invisible: TRANSPARENT like the reference (:649-651) */
    used++;
  }
  alloc = self->out_is_device ? gst_vfhip_device_allocator_get (gst_vfhip_element_device (self)) : gst_vfhip_pinned_allocator_get ();
  outbuf = gst_buffer_new_allocate (alloc, GST_VIDEO_INFO_SIZE (&self->out_info), NULL);
  gst_object_unref (alloc);
  if (!outbuf || !gst_video_frame_map (&out, &self->out_info, outbuf, (GstMapFlags) (GST_MAP_WRITE | gst_vfhip_map_flag (outbuf, gst_vfhip_element_device (self))))) {
    if (outbuf) gst_buffer_unref (outbuf);
    rc = GST_FLOW_ERROR;
    goto done;
  }
  gst_vfhip_frame (&out, &vout);
  if (self->async_depth > 0) {
    /* submit this composite, THEN complete the previous one: its download overlaps this one's uploads and kernel */
    rc = vfhip_compositor_submit (self->renderer, pads, (int) used, (covered && used > 0) ? VFHIP_BG_TRANSPARENT : self->background, &vout);
    if (rc != VFHIP_OK) {
      GST_ERROR_OBJECT (self, "HIP compositing failed: %s", vfhip_last_error_string ());
      gst_video_frame_unmap (&out);
      gst_buffer_unref (outbuf);
      rc = GST_FLOW_ERROR;
      goto done;
    }
    GST_BUFFER_PTS (outbuf) = out_start;
    GST_BUFFER_DURATION (outbuf) = out_end - out_start;
    seg->position = out_end;
    self->n_frames++;
    rc = comp_finish_pending (self, TRUE);
    self->pending.outbuf = outbuf; self->pending.out = out; self->pending.n = n; self->pending.bufs = bufs; self->pending.frames = frames;
    self->have_pending = TRUE;
    keep = TRUE;                                           /* inputs and output stay mapped until their wait */
    goto done;
  }
  rc = vfhip_compositor_composite (self->renderer, pads, (int) used, (covered && used > 0) ? VFHIP_BG_TRANSPARENT : self->background, &vout);
  gst_video_frame_unmap (&out);
  if (rc != VFHIP_OK) {
    GST_ERROR_OBJECT (self, "HIP compositing failed: %s", vfhip_last_error_string ());
    gst_buffer_unref (outbuf);
    rc = GST_FLOW_ERROR;
    goto done;
  }
  GST_BUFFER_PTS (outbuf) = out_start;
  GST_BUFFER_DURATION (outbuf) = out_end - out_start;
  seg->position = out_end;
  self->n_frames++;
  rc = gst_aggregator_finish_buffer (agg, outbuf);
done:
  for (i = 0; i < n; i++) {
    if (!keep && frames[i].buffer) gst_video_frame_unmap (&frames[i]);
    if (!keep && bufs[i]) gst_buffer_unref (bufs[i]);
    gst_object_unref (refs[i].pad);
  }
  if (!keep) { g_free (frames); g_free (bufs); }
  g_free (pads); g_free (refs);
  return (GstFlowReturn) rc;
}

/* Pointer navigation events travelling upstream go to the sink pads whose picture lies under the pointer, with the
 * coordinates translated into that input's own pixels (reference _src_event / src_pad_mouse_event,
 * gstvfmetalcompositor.m:704-787).  The event's structure is read directly ("event" = mouse-move, mouse-button-press,
 * mouse-button-release, mouse-scroll), so this needs nothing newer than GStreamer 1.14. */
static gboolean
comp_src_event (GstAggregator * agg, GstEvent * event)
{
  GstVfHipCompositor *self = COMP (agg);
  const GstStructure *st;
  const gchar *kind;
  gdouble px, py;
  GList *l, *pads = NULL;
  gboolean res = FALSE;
  if (GST_EVENT_TYPE (event) == GST_EVENT_QOS) {
    GstQOSType type;
    gdouble proportion;
    GstClockTimeDiff diff;
    GstClockTime timestamp;
    gst_event_parse_qos (event, &type, &proportion, &diff, &timestamp);
    comp_update_qos (self, proportion, diff, timestamp);
    return GST_AGGREGATOR_CLASS (gst_vfhip_compositor_parent_class)->src_event (agg, event);      /* ... and on to the sources */
  }
  if (GST_EVENT_TYPE (event) != GST_EVENT_NAVIGATION || !(st = gst_event_get_structure (event)) ||
      !(kind = gst_structure_get_string (st, "event")) || !g_str_has_prefix (kind, "mouse-") ||
      !gst_structure_get_double (st, "pointer_x", &px) || !gst_structure_get_double (st, "pointer_y", &py) || !self->have_out_info)
    return GST_AGGREGATOR_CLASS (gst_vfhip_compositor_parent_class)->src_event (agg, event);
  GST_OBJECT_LOCK (agg);
  for (l = GST_ELEMENT (agg)->sinkpads; l; l = l->next)
    pads = g_list_prepend (pads, gst_object_ref (l->data));
  GST_OBJECT_UNLOCK (agg);
  for (l = pads; l; l = l->next) {
    GstVfHipCompositorPad *cpad = CPAD (l->data);
    gint w, h, xo, yo;
    if (cpad->have_info) {
      gdouble x0, y0;
      comp_pad_rect (self, cpad, GST_VIDEO_INFO_PAR_N (&self->out_info), GST_VIDEO_INFO_PAR_D (&self->out_info), &w, &h, &xo, &yo);
      x0 = cpad->xpos + xo; y0 = cpad->ypos + yo;
      if (w > 0 && h > 0 && px >= x0 && px < x0 + w && py >= y0 && py < y0 + h) {
        GstStructure *copy = gst_structure_copy (st);
        gst_structure_set (copy, "pointer_x", G_TYPE_DOUBLE, (px - x0) * GST_VIDEO_INFO_WIDTH (&cpad->info) / (gdouble) w,
            "pointer_y", G_TYPE_DOUBLE, (py - y0) * GST_VIDEO_INFO_HEIGHT (&cpad->info) / (gdouble) h, NULL);
        res |= gst_pad_push_event (GST_PAD (cpad), gst_event_new_navigation (copy));
      }
    }
    gst_object_unref (l->data);
  }
  g_list_free (pads);
  gst_event_unref (event);
  return res;
}

/* upstream of every sink pad is offered the device allocator (memory:HIPMemory caps) or the pinned host one, plus video meta */
static gboolean
comp_propose_allocation (GstAggregator * agg, GstAggregatorPad * pad, GstQuery * decide_query, GstQuery * query)
{
  GstCaps *caps = NULL;
  GstAllocator *a;
  GstAllocationParams params;
  (void) pad; (void) decide_query;
  gst_query_parse_allocation (query, &caps, NULL);
  a = gst_vfhip_caps_has_hip_feature (caps) ? gst_vfhip_device_allocator_get (gst_vfhip_element_device (agg)) : gst_vfhip_pinned_allocator_get ();
  gst_allocation_params_init (&params);
  params.align = 63;
  gst_query_add_allocation_param (query, a, &params);
  gst_query_add_allocation_meta (query, GST_VIDEO_META_API_TYPE, NULL);
  gst_object_unref (a);
  return TRUE;
}

static GstFlowReturn
comp_flush (GstAggregator * agg)
{
  (void) comp_finish_pending (COMP (agg), FALSE);          /* a flushed frame is completed and dropped */
  comp_drop_current_buffers (COMP (agg));                  /* the time line restarts at the new segment */
  comp_reset_qos (COMP (agg));
  return GST_FLOW_OK;
}

static gboolean
comp_stop (GstAggregator * agg)
{
  GstVfHipCompositor *self = COMP (agg);
  (void) comp_finish_pending (self, FALSE);
  if (self->renderer)
    vfhip_compositor_cleanup (self->renderer);
  comp_drop_current_buffers (self);
  comp_reset_qos (self);
  self->have_out_info = FALSE;
  return TRUE;
}

static void
comp_set_property (GObject * object, guint id, const GValue * value, GParamSpec * pspec)
{
  GstVfHipCompositor *self = COMP (object);
  switch (id) {
    case PROP_BACKGROUND: self->background = g_value_get_enum (value); break;
    case PROP_ZERO_SIZE_IS_UNSCALED: self->zero_size_is_unscaled = g_value_get_boolean (value); break;
    case PROP_IGNORE_INACTIVE_PADS:
      /* the reference forwards this to its base class (gstvfmetalcompositor.m:929-933): a live aggregator then stops waiting out its
       * latency deadline for pads that never delivered a buffer.  It changes WHEN a frame is composed, never what is drawn (a pad without
       * a buffer is skipped either way); base classes older than 1.20 have no such notion and keep waiting */
      self->ignore_inactive_pads = g_value_get_boolean (value);
#if GST_CHECK_VERSION (1, 20, 0)
      gst_aggregator_set_ignore_inactive_pads (GST_AGGREGATOR (object), self->ignore_inactive_pads);
#else
      if (self->ignore_inactive_pads)
        GST_INFO_OBJECT (object, "ignore-inactive-pads: this GStreamer's GstAggregator (< 1.20) always waits for its latency deadline");
#endif
      break;
    case PROP_DEVICE_ID: self->device_id = g_value_get_int (value); break;
    case PROP_ASYNC_DEPTH: self->async_depth = g_value_get_int (value); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
}

static void
comp_get_property (GObject * object, guint id, GValue * value, GParamSpec * pspec)
{
  GstVfHipCompositor *self = COMP (object);
  switch (id) {
    case PROP_BACKGROUND: g_value_set_enum (value, self->background); break;
    case PROP_ZERO_SIZE_IS_UNSCALED: g_value_set_boolean (value, self->zero_size_is_unscaled); break;
    case PROP_IGNORE_INACTIVE_PADS: g_value_set_boolean (value, self->ignore_inactive_pads); break;
    case PROP_DEVICE_ID: g_value_set_int (value, self->device_id); break;
    case PROP_ASYNC_DEPTH: g_value_set_int (value, self->async_depth); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec); break;
  }
}

static void
comp_finalize (GObject * object)
{
  GstVfHipCompositor *self = COMP (object);
  if (self->renderer)
    vfhip_compositor_free (self->renderer);
  self->renderer = NULL;
  G_OBJECT_CLASS (gst_vfhip_compositor_parent_class)->finalize (object);
}

static GObject *
comp_child_by_index (GstChildProxy * proxy, guint index)
{
  GObject *obj;
  GST_OBJECT_LOCK (proxy);
  obj = g_list_nth_data (GST_ELEMENT_CAST (proxy)->sinkpads, index);
  if (obj)
    gst_object_ref (obj);
  GST_OBJECT_UNLOCK (proxy);
  return obj;
}

static guint
comp_children_count (GstChildProxy * proxy)
{
  guint n;
  GST_OBJECT_LOCK (proxy);
  n = GST_ELEMENT_CAST (proxy)->numsinkpads;
  GST_OBJECT_UNLOCK (proxy);
  return n;
}

static void
comp_child_proxy_init (gpointer g_iface, gpointer iface_data)
{
  GstChildProxyInterface *iface = g_iface;
  (void) iface_data;
  iface->get_child_by_index = comp_child_by_index;
  iface->get_children_count = comp_children_count;
}

static GstPad *
comp_request_new_pad (GstElement * element, GstPadTemplate * templ, const gchar * name, const GstCaps * caps)
{
  GstPad *pad = GST_ELEMENT_CLASS (gst_vfhip_compositor_parent_class)->request_new_pad (element, templ, name, caps);
  if (pad)
    gst_child_proxy_child_added (GST_CHILD_PROXY (element), G_OBJECT (pad), GST_OBJECT_NAME (pad));
  return pad;
}

static void
comp_release_pad (GstElement * element, GstPad * pad)
{
  gst_child_proxy_child_removed (GST_CHILD_PROXY (element), G_OBJECT (pad), GST_OBJECT_NAME (pad));
  GST_ELEMENT_CLASS (gst_vfhip_compositor_parent_class)->release_pad (element, pad);
}

static void
gst_vfhip_compositor_class_init (GstVfHipCompositorClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);
  GstAggregatorClass *ac = GST_AGGREGATOR_CLASS (klass);
  oc->set_property = comp_set_property;
  oc->get_property = comp_get_property;
  oc->finalize = comp_finalize;
  ec->request_new_pad = GST_DEBUG_FUNCPTR (comp_request_new_pad);
  ec->release_pad = GST_DEBUG_FUNCPTR (comp_release_pad);
  ac->sink_event = GST_DEBUG_FUNCPTR (comp_sink_event);
  ac->sink_query = GST_DEBUG_FUNCPTR (comp_sink_query);
  ac->update_src_caps = GST_DEBUG_FUNCPTR (comp_update_src_caps);
  ac->fixate_src_caps = GST_DEBUG_FUNCPTR (comp_fixate_src_caps);
  ac->negotiated_src_caps = GST_DEBUG_FUNCPTR (comp_negotiated_src_caps);
  ac->aggregate = GST_DEBUG_FUNCPTR (comp_aggregate);
  ac->stop = GST_DEBUG_FUNCPTR (comp_stop);
  ac->flush = GST_DEBUG_FUNCPTR (comp_flush);
  ac->src_event = GST_DEBUG_FUNCPTR (comp_src_event);
  ac->propose_allocation = GST_DEBUG_FUNCPTR (comp_propose_allocation);

  g_object_class_install_property (oc, PROP_BACKGROUND, g_param_spec_enum ("background", "Background", "Background type",
          comp_background_type (), VFHIP_BG_CHECKER, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_ZERO_SIZE_IS_UNSCALED, g_param_spec_boolean ("zero-size-is-unscaled", "Zero size is unscaled",
          "If TRUE, then input video is unscaled in that dimension if width or height is 0 (for backwards compatibility)", TRUE,
          G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_IGNORE_INACTIVE_PADS, g_param_spec_boolean ("ignore-inactive-pads", "Ignore inactive pads",
          "Avoid timing out waiting for inactive pads", FALSE, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_DEVICE_ID, g_param_spec_int ("device-id", "Device ID",
          "GPU ordinal to run on (-1: $VFHIP_DEVICE, else 0)", -1, 63, GST_VFHIP_DEFAULT_DEVICE_ID, G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
  g_object_class_install_property (oc, PROP_ASYNC_DEPTH, gst_vfhip_async_depth_pspec ());

  gst_element_class_add_static_pad_template_with_gtype (ec, &comp_src_template, GST_TYPE_AGGREGATOR_PAD);
  gst_element_class_add_static_pad_template_with_gtype (ec, &comp_sink_template, gst_vfhip_compositor_pad_get_type ());
  gst_element_class_set_static_metadata (ec, "HIP Compositor", "Filter/Editor/Video/Compositor",
      "MI355X-accelerated compositing of multiple video streams", "vfhip");
  GST_DEBUG_CATEGORY_INIT (gst_vfhip_compositor_debug, "vfhipcompositor", 0, "vfhip compositor (reference: gstvfmetalcompositor.m:1068-1069)");
}

static void
gst_vfhip_compositor_init (GstVfHipCompositor * self)
{
  self->background = VFHIP_BG_CHECKER;
  self->zero_size_is_unscaled = TRUE;
  self->device_id = GST_VFHIP_DEFAULT_DEVICE_ID;
  self->ts_offset = GST_CLOCK_TIME_NONE;
  self->qos_proportion = 0.5;
  self->qos_earliest = GST_CLOCK_TIME_NONE;
  gst_video_info_init (&self->out_info);
}

gboolean
gst_vfhip_compositor_register (GstPlugin * plugin)
{
  gboolean ok = gst_element_register (plugin, "vfhipcompositor", GST_RANK_PRIMARY + 2, gst_vfhip_compositor_get_type ());
#ifdef VFHIP_REGISTER_VFMETAL_NAMES
  ok &= gst_element_register (plugin, "vfmetalcompositor", GST_RANK_PRIMARY + 2, gst_vfhip_compositor_get_type ());
#endif
  return ok;
}

