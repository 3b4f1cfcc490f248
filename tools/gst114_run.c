/* tools/gst114_run.c — golden-vector generator helper (NOT product code, never runs on the GPU box).
 *
 * Pushes the raw frames of one input file through a GStreamer pipeline described on the
 * command line, with explicit input caps (so colorimetry / chroma-site can be pinned), and
 * writes the raw output frames to a file.  Used only by tools/gen_goldens.py in the build
 * container, where GStreamer 1.14.0 lives under /opt/conda.
 *
 *   gst114_run <in.raw> <frame-bytes> "<in caps>" "<middle of pipeline>" "<out caps>" <out.raw>
 *
 * pipeline: appsrc caps=<in caps> ! <middle> ! <out caps> ! filesink
 */
#include <gst/gst.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main (int argc, char **argv)
{
  if (argc != 7) { fprintf (stderr, "usage: %s in frame_bytes incaps middle outcaps out\n", argv[0]); return 2; }
  gst_init (NULL, NULL);
  const char *in = argv[1]; size_t fb = strtoull (argv[2], NULL, 10);
  gchar *desc = g_strdup_printf ("appsrc name=src format=time ! %s ! %s ! filesink location=%s", argv[4], argv[5], argv[6]);
  GError *err = NULL;
  GstElement *pipe = gst_parse_launch (desc, &err);
  if (!pipe) { fprintf (stderr, "parse: %s\n", err ? err->message : "?"); return 1; }
  GstElement *src = gst_bin_get_by_name (GST_BIN (pipe), "src");
  GstCaps *caps = gst_caps_from_string (argv[3]);
  if (!caps) { fprintf (stderr, "bad caps %s\n", argv[3]); return 1; }
  g_object_set (src, "caps", caps, NULL);
  gst_element_set_state (pipe, GST_STATE_PLAYING);
  FILE *f = fopen (in, "rb"); if (!f) { perror (in); return 1; }
  guint64 n = 0;
  for (;;) {
    void *mem = g_malloc (fb);
    if (fread (mem, 1, fb, f) != fb) { g_free (mem); break; }
    GstBuffer *buf = gst_buffer_new_wrapped (mem, fb);
    GST_BUFFER_PTS (buf) = n * GST_SECOND; GST_BUFFER_DURATION (buf) = GST_SECOND; n++;
    GstFlowReturn ret; g_signal_emit_by_name (src, "push-buffer", buf, &ret);
    gst_buffer_unref (buf);
    if (ret != GST_FLOW_OK) { fprintf (stderr, "push failed %d\n", ret); return 1; }
  }
  fclose (f);
  GstFlowReturn ret; g_signal_emit_by_name (src, "end-of-stream", &ret);
  GstBus *bus = gst_element_get_bus (pipe);
  GstMessage *msg = gst_bus_timed_pop_filtered (bus, 60 * GST_SECOND, GST_MESSAGE_EOS | GST_MESSAGE_ERROR);
  int rc = 0;
  if (!msg) { fprintf (stderr, "timeout\n"); rc = 1; }
  else if (GST_MESSAGE_TYPE (msg) == GST_MESSAGE_ERROR) {
    GError *e = NULL; gchar *dbg = NULL; gst_message_parse_error (msg, &e, &dbg);
    fprintf (stderr, "error: %s\n%s\n", e->message, dbg ? dbg : ""); rc = 1;
  }
  gst_element_set_state (pipe, GST_STATE_NULL);
  return rc;
}
