/* gst/gstvfhipasync.c — async-depth=1 for the GstBaseTransform-based vfhip elements.
 *
 * The reference's processFrame is synchronous: upload, render, read back, return — the GPU idles while the CPU copies and
 * vice versa.  With async-depth=1 an element keeps one frame in flight: chain(n) submits frame n
 * (libvfhip's _submit / _wait: upload, kernel and download queued on the handle's three streams) and pushes frame n-1, so
 * frame n's upload overlaps frame n-1's kernel and download.  Cost: one frame of latency (reported in the latency query);
 * frames still leave in order, and EOS / new caps / a segment / a gap / flush / state changes drain the pipeline first. */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include <gst/base/gstbasetransform.h>
#include "gstvfhip.h"

#define GST_CAT_DEFAULT gst_vfhip_debug

static GstBuffer *
async_finish_oldest (GstBaseTransform * trans, GstVfHipAsync * a)
{
  const guint k = a->head;
  GstBuffer *out = a->pending[k].outbuf;
  const int rc = a->wait (trans);
  gst_video_frame_unmap (&a->pending[k].out);
  gst_video_frame_unmap (&a->pending[k].in);
  gst_buffer_unref (a->pending[k].inbuf);
  a->pending[k].inbuf = a->pending[k].outbuf = NULL;
  a->head ^= 1;
  a->n--;
  if (rc != VFHIP_OK) {
    GST_WARNING_OBJECT (trans, "HIP processing failed: %s", vfhip_last_error_string ());
    gst_buffer_unref (out);
    return NULL;
  }
  return out;
}

/* streaming thread (or a stopped element) only */
GstFlowReturn
gst_vfhip_async_drain (GstBaseTransform * trans, GstVfHipAsync * a, gboolean push)
{
  GstFlowReturn ret = GST_FLOW_OK;
  while (a->n > 0) {
    GstBuffer *out = async_finish_oldest (trans, a);
    if (!out)
      ret = GST_FLOW_ERROR;
    else if (push && ret == GST_FLOW_OK)
      ret = gst_pad_push (GST_BASE_TRANSFORM_SRC_PAD (trans), out);
    else
      gst_buffer_unref (out);
  }
  return ret;
}

/* GstBaseTransform::generate_output for an element whose transform is not passthrough and whose renderer is configured
 * (`ready`); `parent` is the base class's generate_output, used while async-depth is 0 */
GstFlowReturn
gst_vfhip_async_generate_output (GstBaseTransform * trans, GstBuffer ** outbuf, GstVfHipAsync * a, const GstVideoInfo * in_info,
    const GstVideoInfo * out_info, gboolean ready, GstFlowReturn (*parent) (GstBaseTransform *, GstBuffer **))
{
  GstBaseTransformClass *bclass = GST_BASE_TRANSFORM_GET_CLASS (trans);
  GstBuffer *inbuf, *out = NULL;
  GstFlowReturn ret;
  VfHipFrame vin, vout;
  guint k;
  gint dev;
  if (a->depth < 1 || gst_base_transform_is_passthrough (trans)) {
    if (a->n && (ret = gst_vfhip_async_drain (trans, a, TRUE)) != GST_FLOW_OK)
      return ret;
    return parent (trans, outbuf);
  }
  *outbuf = NULL;
  inbuf = trans->queued_buf;
  trans->queued_buf = NULL;
  if (!inbuf)
    return GST_FLOW_OK;                                       /* second call of the chain loop: nothing more this time */
  if (!ready) {
    gst_buffer_unref (inbuf);
    return GST_FLOW_NOT_NEGOTIATED;
  }
  if (bclass->before_transform)
    bclass->before_transform (trans, inbuf);
  if ((ret = bclass->prepare_output_buffer (trans, inbuf, &out)) != GST_FLOW_OK || !out) {
    gst_buffer_unref (inbuf);
    return ret != GST_FLOW_OK ? ret : GST_FLOW_ERROR;
  }
  k = (a->head + a->n) & 1;
  gst_vfhip_pin_foreign_memory (inbuf, &a->pin);
  dev = gst_vfhip_element_device (trans);
  if (!gst_video_frame_map (&a->pending[k].in, (GstVideoInfo *) in_info, inbuf, (GstMapFlags) (GST_MAP_READ | gst_vfhip_map_flag (inbuf, dev)))) {
    gst_buffer_unref (inbuf); gst_buffer_unref (out);
    return GST_FLOW_ERROR;
  }
  if (!gst_video_frame_map (&a->pending[k].out, (GstVideoInfo *) out_info, out, (GstMapFlags) (GST_MAP_WRITE | gst_vfhip_map_flag (out, dev)))) {
    gst_video_frame_unmap (&a->pending[k].in);
    gst_buffer_unref (inbuf); gst_buffer_unref (out);
    return GST_FLOW_ERROR;
  }
  gst_vfhip_frame (&a->pending[k].in, &vin);
  gst_vfhip_frame (&a->pending[k].out, &vout);
  if (a->submit (trans, &vin, &vout) != VFHIP_OK) {
    GST_WARNING_OBJECT (trans, "HIP submit failed: %s", vfhip_last_error_string ());
    gst_video_frame_unmap (&a->pending[k].out);
    gst_video_frame_unmap (&a->pending[k].in);
    gst_buffer_unref (inbuf); gst_buffer_unref (out);
    return GST_FLOW_ERROR;
  }
  a->pending[k].inbuf = inbuf;
  a->pending[k].outbuf = out;
  a->n++;
  if (a->n == 2) {                                            /* frame n is on its way: hand out frame n-1 */
    *outbuf = async_finish_oldest (trans, a);
    if (!*outbuf)
      return GST_FLOW_ERROR;
  }
  return GST_FLOW_OK;
}

/* GstBaseTransform::sink_event: serialized events travel behind every frame before them */
gboolean
gst_vfhip_async_sink_event (GstBaseTransform * trans, GstEvent * event, GstVfHipAsync * a, gboolean (*parent) (GstBaseTransform *, GstEvent *))
{
  switch (GST_EVENT_TYPE (event)) {
    case GST_EVENT_EOS:
    case GST_EVENT_CAPS:
    case GST_EVENT_SEGMENT:
    case GST_EVENT_GAP:
      gst_vfhip_async_drain (trans, a, TRUE);
      break;
    case GST_EVENT_FLUSH_STOP:
      gst_vfhip_async_drain (trans, a, FALSE);
      break;
    default:
      break;
  }
  return parent (trans, event);
}

/* GstBaseTransform::query: the frame kept in flight is one frame of latency */
gboolean
gst_vfhip_async_query (GstBaseTransform * trans, GstPadDirection direction, GstQuery * query, GstVfHipAsync * a, const GstVideoInfo * out_info,
    gboolean (*parent) (GstBaseTransform *, GstPadDirection, GstQuery *))
{
  gboolean ok = parent (trans, direction, query);
  if (ok && GST_QUERY_TYPE (query) == GST_QUERY_LATENCY && direction == GST_PAD_SRC && a->depth > 0 && out_info && GST_VIDEO_INFO_FPS_N (out_info) > 0) {
    gboolean live;
    GstClockTime min, max;
    const GstClockTime frame = gst_util_uint64_scale (GST_SECOND, GST_VIDEO_INFO_FPS_D (out_info), GST_VIDEO_INFO_FPS_N (out_info));
    gst_query_parse_latency (query, &live, &min, &max);
    min += frame;
    if (GST_CLOCK_TIME_IS_VALID (max))
      max += frame;
    gst_query_set_latency (query, live, min, max);
  }
  return ok;
}

GParamSpec *
gst_vfhip_async_depth_pspec (void)
{
  return g_param_spec_int ("async-depth", "Async depth",
      "Frames kept in flight across buffers: 0 = synchronous like the reference, 1 = the upload of frame n overlaps the kernel and "
      "download of frame n-1 (adds one frame of latency)", 0, 1, 0, (GParamFlags) (G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS));
}
